/*
 * gen6d_hip.h — C ABI of libgen6d_hip.so: the hand-written gfx950 kernels behind the Gen6D inference hot path
 * (detector score-map correlation, selector viewpoint similarity, refiner feature-volume pose update).
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors' data_ptr()); nothing is allocated,
 *     freed or synchronised inside; every call enqueues on the given hipStream_t and returns immediately;
 *   - return value: 0 on success, a negative G6D_E* code otherwise (never throws);
 *   - activations are CHANNELS-LAST fp32: [N][D][H][W][C] with an explicit channel stride `ld` (>= C, multiple of 4),
 *     so a layer can read from / write into a channel slice of a wider (concatenated) buffer;
 *   - conv weights are [Cout][taps][Cin] (the reference's [Cout][Cin][kd][kh][kw] permuted once at load time).
 *
 * The reference has no FFI: each entry point replaces a PyTorch op sequence of the reference, cited per function
 * as file:line relative to liuyuan-pal/Gen6D.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 */
#ifndef GEN6D_HIP_H
#define GEN6D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* g6d_stream_t; /* hipStream_t */

enum {
  G6D_OK = 0,
  G6D_EINVAL = -1,   /* bad shape / null pointer / misaligned pointer */
  G6D_ENOSPC = -2,   /* workspace too small */
  G6D_ELAUNCH = -3   /* hipLaunch error (hipGetLastError != success) */
};

int g6d_abi_version(void);
/* last HIP error string of this thread ("" if none) */
const char* g6d_last_error(void);

/* Launch-policy knobs.  The library reads NO environment variable: every dispatch decision that tools/ and tests want to force (a kernel
 * variant for an A/B run, a constant of a split model for a sweep) is a named knob with the product default — conv_patch, tile_policy,
 * split_target, patch_pipe, corr_slots, sel_rowq, conv1_mfma, w43_split_max / _gain / w43_chunk_us, wino_debug, conv_wino43, wino_wide,
 * wino_split_max / _gain / _fix / _per, wino16_2w, conv_wino, conv_wino16, wino_min_work, w43_map, conv_pm, gemv_mfma, c16_ablate, conv16_halo, conv_narrow (gen6d_amd/csrc/common.hip lists meanings
 * and defaults).  Process-wide; reads and writes are relaxed atomics (a launch racing with g6d_set_knob sees the old or the new value):
 * set them before the launches they are meant to steer.  g6d_set_knob returns G6D_EINVAL for an unknown name; g6d_get_knob returns
 * -1e300 for one. */
int g6d_set_knob(const char* name, double value);
double g6d_get_knob(const char* name);
void g6d_reset_knobs(void);

/* Profiling aid: launches the empty kernel `g6d_marker_kernel` so a kernel trace can be cut to a region of interest. */
int g6d_marker(int id, g6d_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Generic implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32), 1-D/2-D/3-D, stride, zero padding.
 * Replaces every dense conv / linear contraction of the hot path:
 *   detector  F.conv2d(que_x, ref_x) correlation            network/detector.py:222-224
 *             score/scale/offset heads                        network/detector.py:164-184,248-255
 *   selector  corr_conv_list (1,3,3) Conv3d stacks            network/selector.py:27-69,187
 *             corr_feats_conv, score_process, mlps, heads     network/selector.py:71-104,197-214
 *   refiner   RefineFeatureNet convs                          network/refiner.py:24-51
 *             RefineVolumeEncodingNet 3x3x3 convs             network/refiner.py:88-143
 * Fused into the operand loader: optional elementwise multiplier (selector query x reference product,
 * selector.py:183-186, never materialised), optional per-channel affine (the preceding InstanceNorm,
 * applied as x*scale+shift; padding stays exactly zero as in the reference where padding follows the norm) and ReLU.
 * Fused into the epilogue: bias, activation, per-(group,channel) sum / sum-of-squares for the FOLLOWING
 * InstanceNorm (fp64 accumulators).
 *   out[m][co] = act( bias[co] + sum_{tap,ci} X(m,tap,ci) * weight[co][tap][ci] )
 *   X = relu?( (in * mul?) * in_scale + in_shift ), 0 outside the input
 * ---------------------------------------------------------------------------------------------------------------- */
/* Workspace (g6d_conv_igemm, g6d_corr2d_patch, g6d_wino_conv3x3): layers whose grid would not fill 256 CUs split their
 * reduction over more blocks.  The workspace holds G6D_WORKSPACE_COUNTER_BYTES of per-tile arrival counters followed by the
 * partial tiles; the block of a tile that arrives last adds the partials and runs the epilogue (no second kernel at any
 * split count).  Contract: 16-byte aligned, the first
 * G6D_WORKSPACE_COUNTER_BYTES are ZERO before the first call (every call leaves them zero), and the buffer is not shared
 * by launches that may run concurrently (one per stream).  NULL / 0 disables splitting. */
#define G6D_WORKSPACE_COUNTER_BYTES 16384

typedef struct G6dConv {
  const float* in;        /* [N][Di][Hi][Wi][ld_in] */
  const float* mul;       /* optional [Hi][Wi][Cin] (dense), broadcast over N and Di (one per image group with mul_group_images); NULL = none */
  const float* in_scale;  /* optional [in_affine_groups][Cin]; NULL = none */
  const float* in_shift;  /* same shape as in_scale */
  const float* weight;    /* [Cout][kd*kh*kw][Cin] */
  const float* bias;      /* [Cout] or NULL */
  float* out;             /* [N*Do*Ho*Wo][ld_out] */
  double* stats;          /* optional [stat_groups][Cout][2] (sum, sumsq), caller-zeroed; NULL = none */
  float* workspace;       /* split scratch (see "Workspace" below); may be NULL if workspace_bytes == 0 */
  size_t workspace_bytes;
  int32_t N, Di, Hi, Wi, Cin, ld_in;
  int32_t Do, Ho, Wo, Cout, ld_out;
  int32_t kd, kh, kw;
  int32_t sd, sh, sw;
  int32_t pd, ph, pw;
  int32_t in_relu;              /* apply ReLU after the input affine */
  int32_t in_affine_per_n;      /* 0: one affine table for all N; k >= 1: image n uses table n / k (1 = a table per image; k = the
                                   hypothesis count D when N = qn * D images of a batched selector call share one table per query) */
  int32_t out_act;              /* 0 none, 1 ReLU, 2 LeakyReLU(0.1) */
  int32_t stat_rows_per_group;  /* output rows per statistics group (0 = all rows in one group) */
  int32_t split_k;              /* 0 = choose automatically; 1 = never split; >1 = force */
  int32_t math_mode;            /* 0 = fp32 MFMA (default, the parity path); 1 = bf16, 2 = fp16 operands with fp32 accumulation:
                                   v_mfma_f32_32x32x16_{bf16,f16}, opt-in speed mode graded separately (BASELINE configs[2], [4]) */
  const float* weight_wino;     /* optional: the same filters transformed for Winograd F(2x2,3x3), [kd][Cin/8][16][Cout][8] in the
                                   layout of g6d_wino_conv3x3 (per depth tap for 3x3x3).  When set and the layer is eligible
                                   (stride 1, "same" padding, maps >= 6x6, Cin % 8 == 0, Cout % 32 == 0, fp32), the launch runs on
                                   the Winograd kernel (2.25x fewer multiplications) with the same prologue / epilogue semantics;
                                   otherwise `weight` is used.  NULL = never. */
  /* Optional InstanceNorm finalisation inside the launch (needs `stats`): the block that finishes last turns the completed
     (sum, sumsq) table into the affine of the FOLLOWING InstanceNorm — scale = 1/sqrt(var + eps), shift = -mean * scale with
     mean = sum / fin_count, var = sumsq / fin_count - mean^2 — instead of a separate g6d_stats_finalize launch.
     fin_scale / fin_shift: [fin_groups][Cout] floats; fin_counter: one int32, ZERO before the launch (the launch leaves it zero).
     NULL fin_scale = off.  Not for statistics that are still to be summed over ranks. */
  float* fin_scale;
  float* fin_shift;
  int32_t* fin_counter;
  double fin_count, fin_eps;
  int32_t fin_groups;
  /* Query batching (round 3; reference API: network/selector.py:165-175 takes [qn,...] queries).  N = qn * k images, image
     n = q * k + d belongs to query q:
       in_image_mod = k      image n READS input image n % k — `in` holds k images (the selector's reference cache, shared by the
                             queries of the batch instead of being replicated); 0 = off (`in` holds N images)
       mul_group_images = k  `mul` is [ceil(N / k)][Hi][Wi][Cin] and image n is multiplied by mul[n / k] (the query's own feature
                             map); 0 = one multiplier for all images
     together with in_affine_per_n = k and stat_rows_per_group = k * Do * Ho * Wo every query keeps its own InstanceNorm. */
  int32_t in_image_mod;
  int32_t mul_group_images;
  int32_t reserved_;
  const void* weight_wino16;    /* optional, used when math_mode != 0: the Winograd-domain filters ROUNDED to the operand type of
                                   math_mode (bf16 / fp16), [kd][Cin/16][16][Cout][16] 16-bit values in the layout of
                                   g6d_wino16_conv3x3_multi's U16 (per depth tap for 3x3x3).  Eligible layers (as weight_wino, and
                                   Cin % 16 == 0, Cout % 64 == 0) then run on the 16-bit Winograd kernel instead of the direct kernels
                                   with 16-bit operands.  NULL = never. */
  const float* weight_wino43;   /* optional (ABI v8): the same filters transformed for Winograd F(4x4,3x3) in the layout of
                                   g6d_wino43_conv3x3_multi's U43, [kd * Cin/8][2][Cout/64][18][2][4][16][4] (per depth tap for 3x3x3).  When set
                                   and the layer is eligible (fp32, stride 1, "same" padding, maps >= 8x8, Cin % 8 == 0, Cout % 64 ==
                                   0, no multiplier) the launch runs on the F(4x4,3x3) kernel: 4x fewer multiplications than the direct
                                   form at ~5x the fp32 error of F(2x2,3x3) — set it only on layers whose parity budget has the room
                                   (the refiner's 32^3 volume layers); takes precedence over weight_wino.  NULL = never. */
} G6dConv;

int g6d_conv_igemm(const G6dConv* desc, g6d_stream_t stream);
/* Kernel family g6d_conv_igemm will run `desc` on (no launch): 0 generic implicit GEMM, 1 LDS-patch kernel, 2 Winograd F(2x2,3x3) kernel,
   3 Winograd F(4x4,3x3) kernel */
int g6d_conv_plan(const G6dConv* desc);
/* sizeof(G6dConv) as compiled into the library: bindings check their struct layout against it */
int g6d_sizeof_conv_desc(void);

/* Stride-1 2-D cross-correlation without bias for Cout <= 32 with input-patch reuse in LDS: the detector's
 * F.conv2d(que_x0, ref_x0, padding=7) (network/detector.py:224), where the generic kernel is bound by re-loading the
 * activation tile for every one of the 225 taps.  in [H][W][ld_in], wgt [Cout][kh*kw][Cin], out [H*W][ld_out],
 * "same" zero padding (kh, kw odd, kw <= 31).  workspace: split scratch (see "Workspace"). */
int g6d_corr2d_patch(const float* in, int H, int W, int Cin, int ld_in, const float* wgt, int Cout, int kh, int kw,
                     float* out, int ld_out, float* workspace, size_t workspace_bytes, int math_mode /* as G6dConv.math_mode */,
                     g6d_stream_t stream);
/* The same correlation for up to 4 map sizes in one launch: the scales of the detector's image pyramid against the same reference
 * filters, N maps (the queries of a batch, network/detector.py:291-304 takes [qn,H,W,3]) per size; the tiles of all maps form one
 * flat work list (fewer splits, one launch).  Buffers within 2^30 floats of each other. */
typedef struct G6dCorrSeg {
  const float* in;        /* [N][H][W][ld_in] */
  float* out;             /* [N][H*W][ld_out] */
  int32_t H, W, ld_in, ld_out;
  int32_t N, reserved_;   /* N >= 1 maps of this size, dense one after the other */
} G6dCorrSeg;
int g6d_corr2d_patch_multi(const G6dCorrSeg* segs, int nseg, int Cin, const float* wgt, int Cout, int kh, int kw,
                           float* workspace, size_t workspace_bytes, int math_mode, g6d_stream_t stream);
/* The same correlation with 16-bit matrix-core operands on its own kernel (math_mode 1 = bf16, 2 = fp16; fp32 maps in and out, fp32
 * accumulation): the reference filters are rounded and laid out per unit on the host, once per object — w16 =
 * [(Cin/32) * kh units][kw][32 co][40 x 16 bit]: unit u holds the 32 channels of chunk u / kh at filter row u % kh, every (tap, co)
 * row is 32 channels + 8 values of zero padding (rows co >= Cout zero), and >= 1 KB of padding follows the last unit.  All kw weight
 * tiles of a unit are staged at once, so a wave runs its 2 kw MFMAs without a barrier (g6d_corr2d_patch_multi with math_mode != 0
 * keeps one hand-over per tap).  Cin % 32 == 0, Cout <= 32, odd kh, odd kw <= 15.  Replaces network/detector.py:222-224 in the
 * reduced-precision mode. */
int g6d_corr2d_patch16_multi(const G6dCorrSeg* segs, int nseg, int Cin, const void* w16, int Cout, int kh, int kw,
                             float* workspace, size_t workspace_bytes, int math_mode, g6d_stream_t stream);

/* The 15x15 level of the same correlation (network/detector.py:222-224) on the Winograd kernel of g6d_wino_conv3x3: the filter is
 * cut into kblocks x kblocks (= 5 x 5) blocks of 3x3 taps, out = sum_b conv3x3(in shifted by (3bi-6, 3bj-6), w_b), and the 25 blocks
 * accumulate in the F(2x2,3x3) transform domain: 2.25x fewer multiplications than g6d_corr2d_patch.  Maps as above (all with the
 * same ld_in); U = the sub-filter banks transformed on the host like g6d_wino_conv3x3's U, CHUNK-major [Cin/8][25][16][Cout][8]
 * (row c*25 + b = 8-channel chunk c of block b = 5*bi + bj, which holds w[:, 3bi..3bi+2, 3bj..3bj+2, :]: the 25 shifted visits of a
 * chunk are consecutive, so the window stays in L2); Cin % 8 == 0, Cout % 32 == 0. */
int g6d_corr2d_wino_multi(const G6dCorrSeg* segs, int nseg, int Cin, const float* U, int Cout, int kblocks, float* workspace,
                          size_t workspace_bytes, g6d_stream_t stream);

/* The same 15x15 correlation with the 25 blocks accumulated in the F(4x4,3x3) transform domain (ABI v8; kernel and filter layout of
 * g6d_wino43_conv3x3_multi): 225 taps cost 25 * 36 / 16 = 56.25 multiplications per output (F(2x2,3x3): 100).  U43 CHUNK-major
 * [Cin/8 * 25][2][Cout/CB][18][CB/32][4][16][4], row c*25 + b = chunk c of block b; Cin % 8 == 0, Cout % 32 == 0.
 * kblocks = 3 (ABI v9): a 9x9 "same" correlation as 3x3 blocks (rows c*9 + b, shifts 3bi-3, 3bj-3) — the 7x7 level of the detector
 * (F.conv2d(que_x1, ref_x1, padding=3)) with its filters zero-extended by one tap on every side: 20.25 multiplications instead of 49. */
int g6d_corr2d_wino43_multi(const G6dCorrSeg* segs, int nseg, int Cin, const float* U43, int Cout, int kblocks, float* workspace,
                            size_t workspace_bytes, g6d_stream_t stream);

/* InstanceNorm finalisation: stats[g][c] = (sum, sumsq) over `count` elements ->
 * scale = 1/sqrt(var+eps), shift = -mean*scale (biased variance; torch InstanceNorm{1,2,3}d, eps 1e-5,
 * network/selector.py:28-77, network/refiner.py:27-50,93-133). n = groups*channels. */
int g6d_stats_finalize(const double* stats, int n, double count, double eps, float* scale, float* shift, g6d_stream_t stream);

/* y = pool2x2?( relu?( x*scale[c] + shift[c] ) ) on [N][H][W][C] channels-last rows (D folded into N).
 * pool: 0 none, 1 = 2x2 max (MaxPool3d (1,2,2), network/selector.py:34,41,56), 2 = full-window mean over HxW
 * (AvgPool3d (1,4,4), selector.py:76).  scale/shift may be NULL (identity); affine_per_n = k >= 1: image n uses table n / k
 * (as G6dConv.in_affine_per_n), 0: one table. */
int g6d_affine_act_pool(const float* in, int ld_in, const float* scale, const float* shift, int affine_per_n, int relu,
                        int pool, int N, int H, int W, int C, float* out, int ld_out, g6d_stream_t stream);

/* Bilinear up-sampling by an integer factor (align_corners=False, F.interpolate in network/refiner.py:74-75) of
 * relu?(x*scale+shift) on [N][H][W][C] -> [N][H*f][W*f][C] written with channel stride ld_out. */
int g6d_upsample_bilinear(const float* in, int ld_in, const float* scale, const float* shift, int affine_per_n,
                          int N, int H, int W, int C, int factor, float* out, int ld_out, g6d_stream_t stream);

/* VGG trunk glue for an NCHW caller (a trunk whose convolutions run elsewhere, e.g. tools/library_trunk.py's MIOpen A/B trunk; reference
 * pretrain_models.py:86-104 with BatchNorm folded): out = maxpool2x2?( relu?( in + bias[c] ) ) in one pass.  The product trunk
 * (g6d_vgg_conv1_pool_nhwc + g6d_wino_conv3x3* / g6d_wino43_conv3x3_multi) fuses this into its epilogues and does not call it. */
int g6d_bias_relu_pool_nchw(const float* in, const float* bias, int N, int C, int H, int W, int relu, int pool, float* out,
                            g6d_stream_t stream);

/* First trunk layer fused (reference network/pretrain_models.py:17-25,66-72: vgg11_bn features[0..3] with BatchNorm
 * folded): out = maxpool2x2( relu( conv3x3_pad1(in, w) + bias ) ).  in [N][3][H][W], w_oihw [64][3][3][3], bias [64],
 * out [N][64][H/2][W/2] (floor); Cin must be 3 and Cout 64.  Replaces the library convolution of that layer, its two
 * layout transposes and the g6d_bias_relu_pool_nchw pass over the full-resolution result. */
int g6d_vgg_conv1_pool(const float* in, int N, int H, int W, const float* w_oihw, const float* bias, int Cin, int Cout,
                       float* out, g6d_stream_t stream);

/* The same layer with a channels-last result [N][H/2][W/2][64] (input of g6d_wino_conv3x3). */
int g6d_vgg_conv1_pool_nhwc(const float* in, int N, int H, int W, const float* w_oihw, const float* bias, int Cin, int Cout,
                            float* out, g6d_stream_t stream);
/* The same on an image in [0,1]: torchvision Normalize ((x - mean[c]) / std[c]; mean, std: HOST arrays of 3 floats) applied
 * while the input tile is staged — replaces the two elementwise passes of img_norm in front of every trunk call. */
int g6d_vgg_conv1_pool_nhwc_norm(const float* in, int N, int H, int W, const float* w_oihw, const float* bias, int Cin, int Cout,
                                 const float* mean_host, const float* std_host, float* out, g6d_stream_t stream);
/* g6d_vgg_conv1_pool_nhwc(_norm) with a 16-BIT channels-last result [N][H/2][W/2][64] (ABI v11; math_mode 1 = bf16, 2 = fp16): the first
 * layer of the reduced-precision mode's 16-bit activation path (its output feeds g6d_conv16_direct_multi).  fp32 arithmetic, rounded
 * once in the epilogue.  math_mode 3: fp16 hi / lo pairs [N][H/2][W/2][2][64] (hi = rn16(v), lo = rn16(v - hi)), the input format of
 * g6d_conv16_direct_multi's math_mode 3 (the fp32 path's trunk).  mean_host / std_host: NULL = the input is already normalised. */
int g6d_vgg_conv1_pool_nhwc16(const float* in, int N, int H, int W, const float* w_oihw, const float* bias, int Cin, int Cout,
                              const float* mean_host, const float* std_host, void* out16, int math_mode, g6d_stream_t stream);


/* The 3x3 layers 64->128 ... 512->512 of the VGG-11-BN trunks (reference network/pretrain_models.py:9-31,61-72:
 * vgg11_bn features[4..28], BatchNorm folded) as Winograd F(2x2,3x3) on fp32 MFMA with the trunk's bias, ReLU and 2x2
 * max-pool fused into the epilogue; channels-last in and out.
 *   in   [N][H][W][ld_in], Cin % 8 == 0;  U = filters transformed on the host, [Cin/8][16][Cout][8] with
 *        U[c][4a+b][co][k ^ (co & 8 ? 4 : 0)] = (G g G^T)[a][b] of filter (co, 8c+k), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
 *        (the two 4-channel halves of a row are swapped for co & 8: bank-conflict-free LDS image, the copy is lane-linear)
 *   y = conv3x3_pad1(in) + bias[co];  relu != 0: y = max(y, 0)
 *   out_full (optional) [N][H][W][ld_full] = y;  out_pool (optional) [N][H/2][W/2][ld_pool] = maxpool2x2(y) (floor)
 *   workspace (optional): small maps split the channel chunks over more blocks (see "Workspace")
 * One trunk layer = one launch: convolution, bias, ReLU, 2x2 max-pool and the channels-last layout in the kernel's epilogue. */
int g6d_wino_conv3x3(const float* in, int N, int H, int W, int Cin, int ld_in, const float* U, const float* bias, int Cout,
                     int relu, float* out_full, int ld_full, float* out_pool, int ld_pool, float* workspace,
                     size_t workspace_bytes, g6d_stream_t stream);

/* The same layer over up to 4 map sizes in one launch: the scales of the detector's image pyramid (network/detector.py:236-241)
 * share every trunk layer's filters, and the quarters of all segments form one flat work list, so the small scales fill the
 * blocks the large ones leave over.  All segments give the same kinds of output (out_full / out_pool both set or both NULL
 * across segments), and their buffers lie within 2^30 floats of each other (allocate them from one buffer). */
typedef struct G6dWinoSeg {
  const float* in;        /* [N][H][W][ld_in] */
  float* out_full;        /* [N][H][W][ld_full] or NULL */
  float* out_pool;        /* [N][H/2][W/2][ld_pool] or NULL */
  int32_t N, H, W, ld_in, ld_full, ld_pool;
} G6dWinoSeg;
int g6d_wino_conv3x3_multi(const G6dWinoSeg* segs, int nseg, int Cin, const float* U, const float* bias, int Cout, int relu,
                           float* workspace, size_t workspace_bytes, g6d_stream_t stream);

/* Reduced-precision variant of g6d_wino_conv3x3_multi (opt-in speed mode as G6dConv.math_mode: 1 = bf16, 2 = fp16 operands, fp32
 * accumulation, fp32 activations in and out) on v_mfma_f32_32x32x16_{bf16,f16}: a chunk is 16 input channels, U16 = the filters
 * transformed AND rounded on the host, [Cin/16][16][Cout][16] 16-bit values (32 bytes per (a,b,co) row; rows with co & 8 carry their
 * two 16-byte halves swapped, as in g6d_wino_conv3x3's U); Cin % 16 == 0, Cout % 64 == 0. */
int g6d_wino16_conv3x3_multi(const G6dWinoSeg* segs, int nseg, int Cin, const void* U16, const float* bias, int Cout, int relu,
                             int math_mode, float* workspace, size_t workspace_bytes, g6d_stream_t stream);

/* Direct (implicit-GEMM) 3x3 / 3x3x3 "same" convolution on 16-BIT ACTIVATIONS (ABI v11, round 6): the kernel family of the VGG trunks
 * (reference network/pretrain_models.py:9-31,66-72; the detector's image pyramid network/detector.py:188-197,236-241; the refiner's
 * crops network/refiner.py:64-78).  v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulation; the activation tile of a K step goes
 * global -> LDS by DMA (zero padding = the buffer's out-of-range zero fill).
 *   math_mode 1 = bf16, 2 = fp16   the reduced-precision mode (BASELINE configs[2] / [4]): `in`, W16 and 16-bit outputs hold that type
 *   math_mode 3                    fp32-CLASS arithmetic on the fp16 matrix cores, used by the fp32 path: every operand is a PAIR of fp16
 *                                  values hi = rn16(x), lo = rn16(x - hi) (22+ significand bits; the MFMA keeps fp16 subnormals) and
 *                                  acc += a_hi b_hi + a_hi b_lo + a_lo b_hi — 3 MFMAs of 32 cycles per 16 channels against 8 x 64 cycles
 *                                  on the fp32 matrix cores.  Activations [pixel][2][C]: the hi plane, then the lo plane (ld >= 2 C).
 *   in     [N][D][H][W][ld_in] 16-bit channels-last (D = 1 for the 2-D layers), rows 16-byte aligned; Cin % 64 == 0 (pairs: % 32)
 *   W16    w_layout 0: [Cout][kd*9][Cin] rows, tap = (kz*3 + ky)*3 + kx (math_mode 1 / 2; both operand tiles staged through LDS)
 *          w_layout 1: FRAGMENT-major, read straight into registers one K step ahead (K step = BK = 64 channels, pairs 32, of one tap;
 *            steps ordered channel slice outermost, taps innermost):
 *            [Cout/128][Cin/BK * kd*9 steps][BK/16 ks][planes][4 groups j][64 lanes][8 values], value e of lane l =
 *            filter (co = 128 tile + 32 j + (l & 31), tap, ci = BK slice + 16 ks + 8 (l >> 5) + e); planes = 1, or 2 (hi, lo) for pairs
 *          Cout % 128 == 0 (fragment-major, 2-D: also Cout = 64, packed as one 128-channel tile whose upper half is zero);
 *          bias [Cout] fp32 or NULL
 *   y = acc_scale * conv(in, W16) + bias  (acc_scale: the filters may carry an exact power-of-two scale that keeps their lo parts
 *       normal; 0 = 1);  relu != 0: y = max(y, 0)
 *   out_full [N][D][H][W][ld_full] = y        element type full_type: 0 = not written, 1 = the 16-bit type (modes 1 / 2), 2 = fp32,
 *                                             3 = fp16 pairs [pixel][2][Cout] (mode 3; ld_full >= 2 Cout)
 *   out_pool [N][H/2][W/2][ld_pool] = maxpool2x2(y) (2-D only, H and W even)     element type pool_type, same coding
 *   stats (optional) [groups][Cout][2] fp64, zeroed by the caller: sum and sum of squares of y (fp32 values, before rounding) per
 *        (image group, channel) are ADDED; group of a pixel = (its index in [N][D][H][W]) / stat_rows_per_group (0 = one group)
 * Up to 4 map sizes per launch (the scales of the detector's pyramid) like g6d_wino_conv3x3_multi; all segments share D and the
 * kinds of output.  No workspace: the K loop is never split. */
typedef struct G6dConv16Seg {
  const void* in;
  void* out_full;
  void* out_pool;
  int32_t N, D, H, W, ld_in, ld_full, ld_pool;
} G6dConv16Seg;
int g6d_conv16_direct_multi(const G6dConv16Seg* segs, int nseg, int Cin, const void* W16, int w_layout, float acc_scale, const float* bias,
                            int Cout, int kd, int relu, int full_type, int pool_type, int math_mode, double* stats, int stat_rows_per_group,
                            g6d_stream_t stream);

/* The selector's query x reference product (network/selector.py:183-186) in the 16-bit activation format of g6d_conv16_direct_multi (ABI v12):
 * out[(q D + d) P + px][plane][c] = split16((ref[d][px][c] * que[q][px][c]) * scale[q][c] + shift[q][c]); ref [D][P][C], que [qn][P][C], scale /
 * shift [qn][C] fp32; out 16-bit: [qn D P][C] for math_mode 1 (bf16) / 2 (fp16), [qn D P][2][C] fp16 hi / lo pairs for math_mode 3.  C % 8 == 0. */
int g6d_product_split16(const float* ref, const float* que, const float* scale, const float* shift, void* out, int qn, int D, int P, int C,
                        int math_mode, g6d_stream_t stream);

/* g6d_affine_act_pool (pool 0 / 1) with the result in the 16-bit activation format of g6d_conv16_direct_multi (ABI v12): [N][Ho][Wo][C] for
 * math_mode 1 (bf16) / 2 (fp16), [N][Ho][Wo][2][C] fp16 hi / lo pairs for math_mode 3 — the InstanceNorm affine + ReLU (+ 2x2 max-pool) between two
 * convs of the selector's stacks (network/selector.py:27-77) as a pass of its own, so that the next conv runs on the direct 16-bit kernel. */
int g6d_affine_split16(const float* in, int ld_in, const float* scale, const float* shift, int affine_per_n, int relu, int pool, int N, int H, int W,
                       int C, void* out, int math_mode, g6d_stream_t stream);

/* The detector's K x K correlation (network/detector.py:188-197,222-224: query feature map x the 32 reference-centre features) on 16-bit
 * activations, halo-patch kernel with the K^2 taps of a slice split over the eight waves of a block (ABI v12; csrc/conv16_direct.hip,
 * corr16_kernel).  segs[i]: in = [N][H][W][ld_in] of the mode's 16-bit type (mode 3: [pixel][2][Cin] fp16 hi / lo pairs), out_full = fp32
 * [N][H][W][ld_full] (ld_full >= 32), D = 1; out = acc_scale * sum_{ky,kx,c} in[y + ky - K/2][x + kx - K/2][c] * w[r][ky K + kx][c] with zero
 * padding.  W16 (host: ops.corr16_pack): [Cin / S][K K][2][64 lanes][8 values] 16-bit — S = 32 channels per slice and fragment f = the
 * slice's 16-channel group (modes 1 / 2), or S = 16 and f = hi / lo plane (mode 3); lane l of a fragment holds reference r = l & 31,
 * channels S c + (16 f for modes 1 / 2) + 8 (l >> 5) + e.  Cout = 32, k in {7, 15}, Cin % 32 == 0, up to 4 map sizes per launch. */
int g6d_corr16_multi(const G6dConv16Seg* segs, int nseg, int Cin, const void* W16, float acc_scale, int Cout, int k, int math_mode,
                     g6d_stream_t stream);

/* g6d_wino_conv3x3_multi on the Winograd F(4x4,3x3) kernel (ABI v8, fp32 on v_mfma_f32_16x16x4_f32): 36 multiplications per 16
 * outputs — 4x fewer than the direct form, 1.78x fewer than F(2x2,3x3) — with the interpolation points (0, +-3/4, +-3/2, inf), whose
 * fp32 error is ~1.3e-6 of the output range at Cin = 512 (F(2x2,3x3): 2.5e-7; the textbook points 0, +-1, +-2: 4.6e-6).  Meant for the
 * layers whose parity budget has that room: the detector's image pyramid (network/pretrain_models.py:17-25, network/detector.py:188-197,
 * 236-241; headline detector parity 1.5e-6 of range against a 1e-4 bar).
 * U43 = the filters transformed on the host in fp64, [Cin/8][2][Cout/CB][18][CB/32][4][16][4] (CB = 64; 32 when Cout % 64):
 * U43[c][b / 3][blk][3a + b % 3][np][kg][lt][2 par + s] = (G g G^T)[a][b] of filter (co = CB blk + 32 np + 16 par + lt, ci = 8c + 2kg + s),
 * (a, b) = position in the 6x6 transform domain.  Per chunk of 8 input channels the two column halves (b < 3, b >= 3) are the units
 * the kernel stages in LDS: each half of a CB-channel block is one contiguous run (18 CB/32 KB) in the order of its LDS image, of which
 * lane (kg, lt) reads the 16 bytes [position][np][kg][lt] — its B operands for two 16-channel tiles.  Cin % 8 == 0, Cout % 64 == 0;
 * segments, bias, relu, outputs and workspace as g6d_wino_conv3x3_multi. */
int g6d_wino43_conv3x3_multi(const G6dWinoSeg* segs, int nseg, int Cin, const float* U43, const float* bias, int Cout, int relu,
                             float* workspace, size_t workspace_bytes, g6d_stream_t stream);

/* In-place L2 normalisation over C of channels-last rows x[rows][ld] (F.normalize eps 1e-12, network/selector.py:118,
 * network/refiner.py:69-71,165).  16-byte accesses when x is 16-byte aligned and ld % 4 == 0 (then C % 4 == 0 is required),
 * scalar accesses otherwise (the [qn][7] rows of the regressor output). */
int g6d_l2norm_rows(float* x, int rows, int C, int ld, g6d_stream_t stream);

/* NCHW (backbone output) -> channels-last, optionally L2-normalised over C (F.normalize eps 1e-12,
 * network/selector.py:118, network/refiner.py:69-71). out [N][H][W][ld_out]. */
int g6d_nchw_to_nhwc(const float* in, int N, int C, int H, int W, int l2norm, float* out, int ld_out, g6d_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Selector similarity (network/selector.py:183-186,192-195 and the InstanceNorm3d(512) at :28,49,63).  The
 * query x reference product [C][D][HW] is never materialised.
 *   que   [HW][C]            L2-normalised query features (channels-last)
 *   refs  [D][HW][C]         reference cache, D = rfn*an with d = r*an + a
 * g6d_selector_ref_sums   (load time)  r1[hw][c] = sum_d refs, r2[hw][c] = sum_d refs^2      (fp64)
 * g6d_selector_prod_affine(query time) InstanceNorm affine of the product from r1/r2:
 *                                      mean_c = sum_hw q r1 / (D*HW), E[x^2]_c = sum_hw q^2 r2 / (D*HW)
 *                                      scale = 1/sqrt(var+eps), shift = -mean*scale           (exact algebra)
 * g6d_selector_scan       (query time) one coalesced streaming pass over refs, wavefront shuffle reductions:
 *                                      score_map[d][hw] = sum_c que*ref ;  vps[d] = sum_hw S^2 / max_hw S
 * ---------------------------------------------------------------------------------------------------------------- */
int g6d_selector_ref_sums(const float* refs, int D, int HW, int C, double* r1, double* r2, g6d_stream_t stream);
int g6d_selector_prod_affine(const float* que, const double* r1, const double* r2, int D, int HW, int C, double eps,
                             float* scale, float* shift, g6d_stream_t stream);
int g6d_selector_scan(const float* que, const float* refs, int D, int HW, int C, float* score_map, float* vps,
                      g6d_stream_t stream);
/* g6d_selector_scan + g6d_selector_prod_affine for all (<= 3) pyramid levels of a BATCH of qn <= 8 queries in ONE launch (+ a tiny
 * reduction launch): the reference cache is streamed once per batch.  Per level l: que[l] [qn][HW_l][C], refs[l] [D][HW_l][C],
 * r1[l] / r2[l] [HW_l][C], HW[l] <= 1024, score_maps[l] [qn][D][HW_l] (written; required); vps [qn][nlev][D], scale / shift
 * [qn][nlev][C]; Dg = global hypothesis count of the InstanceNorm statistics (= D unless the references are sharded over ranks). */
int g6d_selector_levels(int nlev, int qn, const float* const* que, const float* const* refs, const double* const* r1,
                        const double* const* r2, const int* HW, int D, int Dg, int C, double eps, float* const* score_maps,
                        float* vps, float* scale, float* shift, g6d_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Refiner feature-volume construction (network/refiner.py:183-206,208-247; network/operator.py:4-17), fused:
 * rotate the sn^3 grid by R_in, project every voxel into each reference view and the query view, bilinear-sample
 * (zeros padding, align_corners=False, coordinates normalised with the INPUT IMAGE size h_in x w_in), and reduce over
 * the references in registers (mean, unbiased std).  All pointers are device pointers.
 *   feats     [rfn+1][fh][fw][C]  channels-last feature maps; view rfn is the query
 *   projs     [rfn+1][12]         row-major 3x4 K*[R|t] per view (query last)
 *   rot_in    [9]                 row-major R_in (grid row-vector v is mapped to v @ R_in)
 *   lin       [sn]                torch.linspace(-1,1,sn)
 *   mean_in   [sn^3][2C]          cat[mean over refs, query sample]   (input of mean_embed)
 *   std       [sn^3][C]           1 <= rfn <= 8
 * ---------------------------------------------------------------------------------------------------------------- */
int g6d_refiner_volume(const float* feats, const float* projs, const float* rot_in, const float* lin, int rfn, int fh,
                       int fw, int C, int h_in, int w_in, int sn, float* mean_in, float* std, g6d_stream_t stream);
/* The same with intrinsics and poses given separately (ref_Ks [rfn][3][3], ref_poses [rfn][3][4], K_in [3][3], pose_in [3][4]):
 * the projections K @ pose of network/refiner.py:208-226 are formed inside the kernel and the rotation is read from pose_in.
 * batch >= 1 queries in one launch (the reference's forward takes [qn,...], refiner.py:249-269): every operand and both outputs
 * carry a leading [batch] axis (feats [batch][rfn+1][fh][fw][C], ... mean_in [batch][sn^3][2C], std [batch][sn^3][C]). */
int g6d_refiner_volume_kp(const float* feats, const float* ref_Ks, const float* ref_poses, const float* K_in, const float* pose_in,
                          const float* lin, int rfn, int fh, int fw, int C, int h_in, int w_in, int sn, float* mean_in, float* stdv,
                          int batch, g6d_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Detector score assembly (network/detector.py:225-229,243-245,207-216): for one detection scale, take the three raw
 * correlation maps (level l at 1/(8*2^l) resolution, channels-last [h_l*w_l][rfn]), nearest-upsample levels 1,2 to
 * level-0 size, normalise ((x-mu_l)/sigma_l), clip to +-clip, bilinear-resize (align_corners=False) to (hs,ws) and
 * write channels [3*scale_idx .. 3*scale_idx+2] of stacked [hs*ws][rfn][nch].  batch >= 1 queries (network/detector.py:291-304
 * takes [qn,H,W,3]): s_l [batch][h_l*w_l][rfn], stacked [batch][hs*ws][rfn][nch].
 * ---------------------------------------------------------------------------------------------------------------- */
int g6d_detector_assemble(const float* s0, const float* s1, const float* s2, int hc, int wc, int rfn,
                          const float* mu_sigma /* HOST pointer: {mu0,sigma0,mu1,sigma1,mu2,sigma2} */, float clip, int hs, int ws,
                          int scale_idx, int nch, float* stacked, int batch, g6d_stream_t stream);

/* score_conv + max over references (network/detector.py:159-163,246-247): per (pixel, ref) MLP nch->64 (ReLU) ->64,
 * then max over rfn.  w0 [64][nch], b0[64], w1[64][64], b1[64]; out [P][64]. */
int g6d_detector_score_mlp_max(const float* stacked, int P, int rfn, int nch, const float* w0, const float* b0,
                               const float* w1, const float* b1, float* out, g6d_stream_t stream);

/* arg-max + decode (network/detector.py:84-121): scores [hs*ws], offset [hs*ws][2], scale [hs*ws] (channels-last);
 * result[0..1] = position (x,y) px, result[2] = 2^scale, result[3..4] = (x,y) cell of the peak (as float).
 * batch >= 1 queries: the maps of query b start b*hs*ws rows after those of query 0, result [batch][5]. */
int g6d_detector_decode(const float* scores, int ld_s, const float* offset, int ld_o, const float* scale, int ld_c,
                        int hs, int ws, int pool_ratio, float* result, int batch, g6d_stream_t stream);
/* (ABI v10) The detector's image pyramid in one launch: F.interpolate(que_imgs, size=(h_k, w_k), mode='bilinear') (align_corners False;
 * network/detector.py:236-241) of src [planes = N*3][H][W] for up to 4 destination sizes -> dsts[k] [planes][hs[k]][ws[k]] (dense).
 * Source index as ATen: scale = in / out, src = scale * (dst + 0.5) - 0.5 clamped at 0. */
int g6d_resize_bilinear_pyramid(const float* src, int planes, int H, int W, int nscale, const int* hs, const int* ws,
                                float* const* dsts, g6d_stream_t stream);
/* (ABI v10) Stream-ordered zero fill of `bytes` bytes of device memory at a 16-byte aligned address (an own kernel: the runtime's memset node
 * lost writes inside captured graphs). */
int g6d_zero_bytes(void* ptr, size_t bytes, g6d_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Selector tail helpers (network/selector.py:201-214, network/attention.py:4-17,50-68).
 * ---------------------------------------------------------------------------------------------------------------- */
/* The tail helpers take `batch` >= 1 queries (selector.py:165-175 takes [qn,...]): row blocks of the queries follow each other.
 * InstanceNorm2d(3) of vps [batch][3][D] over D, written to channels c_off..c_off+2 of feats [batch*D][ld] (selector.py:201-202) */
int g6d_vps_norm(const float* vps, int D, float* feats, int ld, int c_off, int batch, g6d_stream_t stream);
/* x[b*rfn+r][c] = max_a in[(b*rfn+r)*an+a][c] + embed[r][c]   (selector.py:204-205) */
int g6d_max_an_add(const float* in, int ld_in, int rfn, int an, int C, const float* embed, float* out, int ld_out, int batch,
                   g6d_stream_t stream);
/* multi-head attention over n tokens with the reference's head split c -> (d=c/heads, head=c%heads), scale
 * 1/sqrt(C/heads): q,k,v [batch*n][ld] -> out [batch*n][ld_out], attention among the n tokens of one query
 * (attention.py:4-17,60-64) */
int g6d_attention(const float* q, const float* k, const float* v, int ld, int n, int C, int heads, float* out,
                  int ld_out, int batch, g6d_stream_t stream);
/* LayerNorm over C per token with affine (attention.py:19-26): out may alias in */
int g6d_layernorm(const float* in, int ld_in, int n, int C, const float* gamma, const float* beta, float eps,
                  float* out, int ld_out, g6d_stream_t stream);
/* out = relu?(x*scale+shift) (+ residual) elementwise on [n][C] rows (selector.py:209); rows_per_group = k > 0: row r uses
 * table r / k of scale / shift [n/k][C] (one InstanceNorm1d per query of a batch), 0: one table */
int g6d_affine_act_add(const float* in, int ld_in, const float* scale, const float* shift, int relu,
                       const float* residual, int ld_res, int n, int C, float* out, int ld_out, int rows_per_group,
                       g6d_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Inter-stage image warps of Gen6DEstimator.predict: cv2.warpAffine in transformation_crop
 * (utils/base_utils.py:646-655), cv2.warpPerspective in look_at_crop / normalize_reference_views
 * (utils/database_utils.py:8-25,54-110) and the rotated reference copies (estimator.py:150-164).
 * src uint8 [sh][sw][ch] (device); hinv: HOST pointer, row-major 3x3 destination->source pixel map (inverse of the
 * cv2 `M`/`H`); dst [dh][dw][ch] uint8 (out_float=0, round-to-nearest) or float32 * out_scale (out_float=1).
 * Bilinear, zero outside the source.  Exact float weights: cv2 quantises them to 1/32 px (<= 1-2 grey levels apart).
 * ---------------------------------------------------------------------------------------------------------------- */
int g6d_warp_perspective(const unsigned char* src, int sh, int sw, int ch, const float* hinv, void* dst, int dh, int dw,
                         int out_float, float out_scale, g6d_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Device-resident inter-stage glue: what Gen6DEstimator.predict computes on the HOST between the
 * network calls, as single-launch kernels on device buffers (float32 in memory, float64 arithmetic), so that
 * detect -> crop -> select -> pose -> 3 x refine is one chain of launches without host synchronisation (hipGraph-capturable).
 * All pointers are DEVICE pointers.  3x3 matrices are row-major [9], poses row-major [3][4] = [12].
 * Every kernel takes a `batch` of queries (round 3): per-query operands and results are dense arrays with the query as leading axis
 * (det [batch][5], logits / angles [batch][rfn], que_K [batch][9], poses [batch][12], geo [batch][42+30*ref_num], ...); the reference
 * state (ref_poses, ref_Ks, center, norm, sub_poses, sub_Ks) is shared by the queries.
 * ---------------------------------------------------------------------------------------------------------------- */
/* det = result of g6d_detector_decode (x, y, 2^scale, ...) -> hinv[9]: destination->source map of
 * transformation_crop(que_img, position, 1/scale_r2q, 0, size) (estimator.py:184, utils/base_utils.py:646-655) */
int g6d_chain_crop_from_detection(const float* det, float size, float* hinv, int batch, g6d_stream_t stream);
/* arg-max viewpoint (first maximum) + estimate_pose_from_similarity_transform_compose (estimator.py:193-206,
 * utils/pose_utils.py:12-49,104-111): logits/angles [rfn], ref_poses [rfn][12], ref_Ks [rfn][9] -> pose_out[12],
 * sel_out[2] = (selected reference index, its in-plane angle) */
int g6d_chain_pose_from_selection(const float* det, const float* logits, const float* angles, int rfn, const float* ref_poses,
                                  const float* ref_Ks, const float* que_K, const float* center, float* pose_out, float* sel_out,
                                  int batch, g6d_stream_t stream);
/* Geometry of one refinement step before the network (network/refiner.py:275-313, utils/database_utils.py:54-139 on the
 * NormalizedDatabase): pose_in[12] in the database frame, que_K[9], norm[4] = (NormalizedDatabase.scale, offset xyz),
 * sub_poses [n_sub][12] / sub_Ks [n_sub][9] = normalised poses and intrinsics of the (FPS) reference subset, n_sub <= 128.
 * Writes ref_idx[ref_num] (the views most aligned with the warped input pose) and the record
 *   geo = K_warp[9] | pose_warp[12] | pose_rect[12] | ref_Ks[ref_num][9] | ref_poses[ref_num][12] | hinv[1+ref_num][9]
 * (42 + 30*ref_num floats; hinv[0] warps the query, hinv[1+k] reference k: inputs of g6d_warp_batch).
 * angle_step > 0 (radians; reference-feature caching): the in-plane alignment angle of every reference view is
 * snapped to multiples of angle_step, so that its aligned crop depends on (view, bucket) only; ref_bucket[ref_num] (optional)
 * receives round(angle / angle_step).  angle_step = 0: the reference's exact alignment. */
int g6d_chain_refine_prepare(const float* pose_in, const float* que_K, const float* norm, float size, float margin,
                             const float* sub_poses, const float* sub_Ks, int n_sub, int ref_num, float* geo, int* ref_idx,
                             float angle_step, int* ref_bucket, int batch, g6d_stream_t stream);
/* Refiner outputs (rotation[4] w-first, offset[2], log2 scale[1]) + the step's geo record -> refined pose in the database frame
 * (refiner.py:327-341: compose_sim_pose, pose_sim_to_pose_rigid with the polar factor of the SVD, un-rectify, denormalise). */
int g6d_chain_refine_update(const float* rot, const float* off, const float* scl, const float* geo, int geo_floats /* per query */,
                            const float* norm, float* pose_out, int batch, g6d_stream_t stream);
/* Batched g6d_warp_perspective with homographies and source selection in device memory, output in the layout the networks
 * take: dst [B][ch][dh][dw] float = rint(bilinear)/255 (the uint8 image cv2.warpPerspective would return, scaled to [0,1]);
 * image b reads `single` when idx == NULL or idx[b] < 0, else stack[idx[b]] (uint8 [.][sh][sw][ch]); hinv [B][9]. */
int g6d_warp_batch(const unsigned char* stack, const unsigned char* single, const int* idx, int B, int sh, int sw, int ch,
                   const float* hinv, float* dst, int dh, int dw, g6d_stream_t stream);

/* Small-batch linear layer, weight-streaming GEMV (network/refiner.py:153-166): out[b][o] = act(W[o].x[b] + bias[o]);
 * W [O][K] row-major, x [B][K], B <= 8; act as G6dConv.out_act. */
int g6d_linear_gemv(const float* x, int B, int K, const float* W, const float* bias, int O, int act, float* out,
                    g6d_stream_t stream);

/* (ABI v9) The same layer for ANY number of right-hand sides (network/refiner.py:249-269 takes [qn, ...] queries).  Since v10 ONE pass over
 * the weights serves up to 32 rows of x: groups of 2..16 with O % 8 == 0 and K % 2048 == 0 run 8 output rows x one 2048-float K slice per
 * block with one or two groups of 8 rows of x against the block's weight values; groups of 17..32 with O % 32 == 0 and K % 1024 == 0 run
 * as a GEMM on the matrix cores (32 weight rows x 256 k per wave).  Partial sums are joined in slice order through `workspace`
 * (G6D_WORKSPACE_COUNTER_BYTES of counters, zero at rest, then <= O * K / 32 floats); anything else takes the row-per-block kernel of
 * g6d_linear_gemv. */
int g6d_linear_gemv_batch(const float* x, int B, int K, const float* W, const float* bias, int O, int act, float* out, float* workspace,
                          size_t workspace_bytes, g6d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GEN6D_HIP_H */
