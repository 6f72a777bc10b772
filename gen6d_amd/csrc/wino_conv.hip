// 3x3 stride-1 pad-1 convolution as Winograd F(2x2,3x3) on the fp32 matrix cores of gfx950 — the seven 64->128 ...
// 512->512 layers of the VGG-11-BN trunks (reference network/pretrain_models.py:9-31,61-72, BatchNorm folded), with the
// bias, ReLU and 2x2 max-pool of the trunk fused into the epilogue, channels-last in and out.
//
//   Y(2x2) = A^T [ sum_ci U_ci (.) V_ci ] A,   U = G g G^T (host, once per checkpoint),  V = B^T d B (4x4 input tile)
//
// i.e. 16 independent GEMMs  D_ab[tile][co] = sum_ci V_ab[tile][ci] * U_ab[ci][co]  with 2.25x fewer multiplications
// than the direct form.  Mapping to v_mfma_f32_32x32x2_f32 (M = tiles, N = co, K = ci):
//
//   block  = 256 threads = 4 waves = 64 Winograd tiles (four "quarters" of 4x4 tiles = 8x8 output pixels each, consecutive
//            entries of the flat list of all quarters of all images) x 64
//            output channels; wave (wm, wn) owns 32 tiles x 32 channels for ALL 16 (a,b) positions: 16 accumulator
//            tiles = 256 accumulator registers per lane, one wave per SIMD.  Because a lane holds all 16 D_ab of its
//            (tile, co) elements, the output transform, bias, ReLU and the 2x2 max-pool (= max over the 4 outputs of a
//            tile) are pure per-lane register arithmetic: no LDS exchange in the epilogue.
//   K loop = chunks of 8 input channels: the RAW 10x10 input patch of every quarter (not its 3.2x larger transform) is
//            staged in LDS through registers (zero padding + image borders masked there); the input transform runs in
//            registers on the fragment each lane reads (16 ds_read_b128 -> 32 vector add/sub -> 16 A fragments).  The
//            pre-transformed filters are stored [chunk][ab][co][8], so a block's 32 KB per chunk is 16 contiguous 2 KB
//            runs that go global -> LDS directly (global_load_lds_dwordx4, no VGPR round trip: the registers are spent on
//            accumulators).  Two LDS stages, one barrier per chunk; per wave and chunk 64 MFMAs against 32 ds_read_b128.
//   split  = small maps do not fill 256 CUs with 64-tile blocks and a block's K loop is serial (2-3 us per chunk): the
//            channel chunks are then split over gridDim.z; partial OUTPUT tiles (the output transform is linear) go to a
//            workspace and the block of a tile that arrives last adds them and applies bias / ReLU / max-pool / statistics.
//   K trick = as in conv_igemm.hip: lane-half h reads channels 4h..4h+3 of the chunk with one ds_read_b128 per operand;
//            MFMA s consumes channel s (lanes 0-31) and 4+s (lanes 32-63) for both operands.
//
// Numerics: fp32 throughout; F(2x2,3x3) has transform constants {0, +-1, +-1/2} only (same error class as the MIOpen
// Winograd solver the trunk ran on before).
#include "g6d_common.h"
#include <stdlib.h>
#include <stdio.h>
#include <algorithm>
#include <type_traits>

#ifndef WINO_ABLATE
#define WINO_ABLATE 0   // profiling builds only (tools/wino_ablate.sh; results are wrong by construction): 1 = no barrier in the chunk
                        // loop, 2 = no global loads / LDS stores of the next chunk, 3 = no input transform (raw values as fragments),
                        // 4 = no fragment reads from LDS, 5 = all of them (matrix cores only), 6 = (16-bit kernel) no MFMA, 7 = no epilogue, 9 = launch + prologue only
#endif
#define WABL(n) (WINO_ABLATE == (n) || WINO_ABLATE == 5)

namespace {

// Raw patch image in LDS: position (q, py, px) of quarter q at ((q*101 + py*10 + px) * 8) floats, its two 4-channel halves
// swapped when (py >> 1) is odd; filter image [ab][co][8] with the halves swapped when (co & 8) (done once on the host, the
// direct-to-LDS copy is lane-linear).  Both layouts make every ds_read_b128 of the fragment loops bank-conflict free
// (checked exhaustively over the four 16-lane service groups of the instruction; plain 12-float rows were 3-way, plain
// [co][8] rows 2-way conflicted).
#ifndef WINO_M3_RAW_AUX
#define WINO_M3_RAW_AUX 0                   // cache policy of the raw pieces of the product layers (profiling builds: 2 = non-temporal)
#endif
#define WQ_PIX 101                         // position stride between quarters (10 x 10 used)
#define WRAW_LD 8                          // floats per raw patch position
#define WRAW_FLOATS (4 * WQ_PIX * WRAW_LD) // 3232

// The same kernel also runs the stride-1 3x3 / 3x3x3 layers of the selector, the refiner feature net and the refiner volume
// net when g6d_conv_igemm finds pre-transformed filters in G6dConv.weight_wino (template parameters):
//   MODE  operand prologue applied when the raw patch is written to LDS, as in conv_igemm.hip: 0 none, 1 InstanceNorm
//         affine(+ReLU) with one table, 2 one table per image group (image / aff_div; LDS holds the tables of the block's four
//         quarters), 3 elementwise multiplier (selector query x reference product) + affine, tables per quarter as in 2 — a
//         batch of queries shares the reference images (img_mod) while each query brings its own multiplier map (mul_div) and
//         table; zero padding stays exactly zero.  The affine tables sit in LDS (loaded once per block).
//   KD    3: 3x3x3 layers as a 2-D Winograd over (h, w) with the three depth taps folded into the reduction: chunk c =
//         (kd, 8 input channels) reads depth slice d + kd - 1 — the transform-domain accumulators are shared, so the
//         multiplication count drops from 27 to 12 per output.
//         25: a 15x15 "same" correlation (the detector's reference-as-filter level, network/detector.py:222-224) as 5x5 blocks of 3x3
//         sub-filters: out = sum_{bi,bj} conv3x3(in shifted by (3bi-6, 3bj-6), w[3bi..3bi+2, 3bj..3bj+2]) — chunk c = (block, 8 input
//         channels) reads the raw patch at the block's shift, all 25 blocks accumulate in the same transform-domain accumulators:
//         225 taps cost 25 * 16 / 4 = 100 multiplications per output (2.25x fewer, as for a single 3x3).
//   NWN   output-channel width of a block in 32s: 2 = 64 channels / 4 waves; 1 = 32 channels / 2 waves (two such blocks share a
//         CU), used when 64-wide blocks would leave CUs idle (e.g. the 32^3 x 64 volume layers: 128 -> 256 blocks).
// Epilogue additions for those layers: per-(group, channel) sum / sum of squares of the outputs for the following
// InstanceNorm (float partials per block, one fp64 atomic per channel and run of equal groups).

// One map size of a launch.  A launch may cover up to WINO_MAX_SEG sizes (the scales of the detector's image pyramid run the
// same layer): their quarters form ONE flat list, so the small scales fill the blocks the large ones leave over instead of
// being four under-filled launches.  Offsets are in floats from the launch's common base pointers.
#define WINO_MAX_SEG 4
struct WinoSeg { int qstart, N, H, W, QH, QW, in_off, full_off, pool_off, ld_in, ld_full, ld_pool; };

struct WinoArgs {
  const float* in; const float* U; const float* bias; float* out_full; float* out_pool;
  int N, H, W, Cin, ld_in, Cout, ld_full, ld_pool, relu;      // N = images x depth slices (every slice is a 2-D map); = seg[0]
  int QH, QW;
  int nseg, qtotal; WinoSeg seg[WINO_MAX_SEG];
  unsigned in_bytes;                              // extent of the input tensor(s) from `in` (< 2^31): bound of the buffer loads
  unsigned mul_bytes;                             // ... of the multiplier maps from `mul` (MODE 3)
  int splits, chunks_per_split; float* ws;       // splits > 1: tile counters + partial outputs (no bias / ReLU / pool)
  // conv-family extras (zero / null for the trunk)
  int D;                                         // depth slices per image (1 for 2-D layers); KD = 3 pads in depth
  const float* mul; const float* in_scale; const float* in_shift; int in_relu;
  int aff_div;                                   // MODE 2 / 3: image i uses affine table i / aff_div (0: one table)
  int img_mod, mul_div;                          // > 0: image i reads input image i % img_mod / multiplier map i / mul_div (G6dConv)
  double* stats; int stats_div;                  // [groups][Cout][2]; group of image i = i / stats_div (0: one group)
  G6dFin fin;                                    // fin.scale != NULL: the last block finalises the statistics
  int mm;                                        // 0: fp32 kernel; 1 / 2: wino16_conv3x3_kernel with bf16 / fp16 operands (U holds 16-bit filters)
};

__device__ __forceinline__ f32x4 ldg4(const float* __restrict__ base, int elem_off) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + ((unsigned)elem_off << 2));
}

typedef float f32x2w __attribute__((ext_vector_type(2)));
#define f32x2 f32x2w
__device__ __forceinline__ f32x2 lo2(const f32x4& x) { return __builtin_shufflevector(x, x, 0, 1); }
__device__ __forceinline__ f32x2 hi2(const f32x4& x) { return __builtin_shufflevector(x, x, 2, 3); }
__device__ __forceinline__ f32x4 cat2(f32x2 a, f32x2 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3); }
// a - b on two floats in one instruction (the compiler scalarises a two-float subtraction; the packed add takes the negation as
// an operand modifier).  Register-only, consumed a whole MFMA group later.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// quarter q of this block -> segment geometry, image, first output row / column, validity.  The quarters (8x8 output pixels)
// of all images of all segments form one flat list, four consecutive ones per block: no block-level padding on odd quarter
// counts, and small maps (<= 8x8 pixels = one quarter per image) simply put four images into a block.
struct QGeo { int n, oy0, ox0; bool valid; int H, W, ld_in, ld_full, ld_pool, in_off, full_off, pool_off; };
__device__ __forceinline__ QGeo quarter_of(const WinoArgs& p, int q, int nq = 4) {
  const int Q = blockIdx.x * nq + q;
  int sidx = 0;
#pragma unroll
  for (int k = 1; k < WINO_MAX_SEG; ++k) sidx = (k < p.nseg && Q >= p.seg[k].qstart) ? k : sidx;
  const WinoSeg& sg = p.seg[sidx];
  QGeo g;
  g.valid = Q < p.qtotal;
  const int Ql = g.valid ? Q - sg.qstart : 0;
  const int per = sg.QH * sg.QW;
  g.n = Ql / per;
  const int r = Ql - g.n * per;
  const int qy = r / sg.QW, qx = r - qy * sg.QW;
  g.oy0 = 8 * qy; g.ox0 = 8 * qx;
  g.H = sg.H; g.W = sg.W; g.ld_in = sg.ld_in; g.ld_full = sg.ld_full; g.ld_pool = sg.ld_pool;
  g.in_off = sg.in_off; g.full_off = sg.full_off; g.pool_off = sg.pool_off;
  return g;
}

// NWM = 32-tile rows of the block (quarters / 2): 2 = four quarters x 32*NWN channels (the default shape); 1 with NWN = 4 = two quarters x
// 128 channels — the same four waves, but one staged raw patch feeds twice the output channels: a layer's input is re-read Cout / 128
// instead of Cout / 64 times (trunk layers with Cout % 128 == 0).
template <int MODE, int KD, int NWN, int NWM = 2>
__global__ void __launch_bounds__(64 * NWM * NWN, 1) wino_conv3x3_kernel(const WinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int THREADS = 64 * NWM * NWN;
  constexpr int NQ = 2 * NWM;                                 // quarters per block
  constexpr int RAWF = NQ * WQ_PIX * WRAW_LD;                 // floats of the raw patch image
  constexpr int NPR = (200 * NQ + THREADS - 1) / THREADS;     // raw-patch pieces per thread and chunk (2, 4 or 7)
  constexpr int GL = 16 / NWM;                                // direct-to-LDS filter pieces per wave and chunk
  constexpr int WU_FLOATS = 16 * 32 * NWN * 8;                // [ab][co][8], lane-linear image of the global layout
  constexpr int WSTAGE = RAWF + WU_FLOATS + 4 * THREADS;   // raw patch, filter image, scratch row for the idle pieces of the last round
  constexpr int AFF0 = 2 * WSTAGE;                            // affine tables behind the stages: [G][Cin] scales, [G][Cin] shifts, [Cin] zeros
  constexpr int AFFG = MODE >= 2 ? NQ : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * (32 * NWN);
  const int nc8 = p.Cin >> 3;                                             // 8-channel chunks per depth tap
  const int c_first = blockIdx.z * p.chunks_per_split;                    // this block's slice of the (kd, chunk) list
  const int c_last = min(KD * nc8, c_first + p.chunks_per_split) - 1;

  // ---- raw patch loader: pieces idx = tid + THREADS*j < 800 = 4 quarters x 100 positions x 2 halves of the 8-channel chunk
  int poff[NPR], lsto[NPR], moff[MODE == 3 ? NPR : 1], aoff[MODE != 0 ? NPR : 1];
  bool pval[NPR], live[NPR];
  unsigned dbits = 0;                                          // KD = 3: bit 2j / 2j+1 = piece j has a slice below / above
  unsigned smask[KD == 25 ? NPR : 1];                          // KD = 25: bits 0-4 / 8-12 = row / column of the piece inside the image under block shift b
  int rstep[KD == 25 ? NPR : 1];                               //          floats per block step along y (3 rows of the piece's map)
#pragma unroll
  for (int j = 0; j < NPR; ++j) {
    const int idx = tid + THREADS * j;
    const int q = idx / 200, r = idx - q * 200, pp = r >> 1, half = r & 1;
    const int py = pp / 10, px = pp - py * 10;
    const QGeo g = quarter_of(p, q < NQ ? q : 0, NQ);
    const int n = g.n;
    const int iy = g.oy0 + py - 1, ix = g.ox0 + px - 1;
    pval[j] = (idx < 200 * NQ) & g.valid & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
    const int n_in = p.img_mod > 0 ? ((n / p.D) % p.img_mod) * p.D + n % p.D : n;      // query batches share the input images
    poff[j] = pval[j] ? g.in_off + ((n_in * g.H + iy) * g.W + ix) * g.ld_in + 4 * half : 0;
    live[j] = idx < 200 * NQ;
    lsto[j] = live[j] ? (q * WQ_PIX + pp) * WRAW_LD + 4 * (half ^ ((py >> 1) & 1)) : RAWF + WU_FLOATS + 4 * tid;
    if constexpr (MODE == 3) moff[j] = pval[j] ? (((p.mul_div > 0 ? n / p.mul_div : 0) * p.H + iy) * p.W + ix) * p.Cin + 4 * half : 0;
    if constexpr (MODE != 0) aoff[j] = (MODE >= 2 ? (q < NQ ? q : 0) * p.Cin : 0) + 4 * half;
    if constexpr (KD == 3) { const int dd = n % p.D; dbits |= (unsigned)(dd > 0) << (2 * j) | (unsigned)(dd < p.D - 1) << (2 * j + 1); }
    if constexpr (KD == 25) {
      unsigned m = 0;
#pragma unroll
      for (int b = 0; b < 5; ++b)
        m |= (unsigned)((unsigned)(iy + 3 * b - 6) < (unsigned)g.H) << b | (unsigned)((unsigned)(ix + 3 * b - 6) < (unsigned)g.W) << (8 + b);
      smask[j] = ((idx < 200 * NQ) & g.valid) ? m : 0u;
      rstep[j] = 3 * g.W * g.ld_in;
      poff[j] = g.in_off + ((n * g.H + iy) * g.W + ix) * g.ld_in + 4 * half;      // the UNSHIFTED position (may lie outside: used under smask only)
    }
  }
  // Operand prologues without a select and (2-D layers) without per-chunk address arithmetic: every piece comes through a
  // bounds-checked buffer load — a piece outside the image (zero padding, masked quarter) asks for an offset beyond the tensor and the
  // hardware returns zeros — and the InstanceNorm SHIFT of such a piece is read from a row of zeros behind the tables, so that
  // raw * scale + shift (and ReLU of it) is exactly zero there, as the reference's padding behind the norm is.  2-D layers: the lane
  // offset of a piece and the LDS offset of its shift are constants of the piece, the chunk's channel offset is the instruction's
  // scalar / immediate offset.  (Round 4: per piece and chunk a `v ? off : 0` select and 64-bit address per load, 4 multiplies, 4 FMAs
  // and 4 selects beside the MFMAs — the MODE 3 / MODE 2 instantiations ran at 44-56 % MfmaUtil against 66 % for MODE 0.)
  unsigned pboff[KD == 1 ? NPR : 1];                          // byte offsets of the pieces for the buffer loads (beyond the tensor: zero)
  unsigned mboff[MODE == 3 ? NPR : 1];                        // ... into the multiplier maps
  int shoff[MODE != 0 ? NPR : 1];                             // LDS offset of the piece's shift values: its table, or the zero row
  if constexpr (KD == 1) {
#pragma unroll
    for (int j = 0; j < NPR; ++j) pboff[j] = pval[j] ? (unsigned)poff[j] << 2 : 0x80000000u;
  }
  if constexpr (MODE == 3) {
#pragma unroll
    for (int j = 0; j < NPR; ++j) mboff[j] = pval[j] ? (unsigned)moff[j] << 2 : 0x80000000u;
  }
  if constexpr (MODE != 0) {
#pragma unroll
    for (int j = 0; j < NPR; ++j) shoff[j] = pval[j] ? AFF0 + AFFG * p.Cin + aoff[j] : AFF0 + 2 * AFFG * p.Cin + (aoff[j] & 4);
  }
  const int slice = p.H * p.W * p.ld_in;                     // KD = 3: one depth step
  const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  const auto mul_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MODE == 3 ? p.mul : p.in), 0, MODE == 3 ? p.mul_bytes : p.in_bytes, 0x00020000);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  f32x4 rp[NPR], rm[MODE == 3 ? NPR : 1];
  bool rv[NPR];                                               // validity of the piece for the chunk it was loaded for
  auto load_piece = [&](int j, int chunk) {
    // reduction order: depth taps outermost for KD = 3; for KD = 25 the 25 block shifts are INNERMOST (chunk-major): the shifted
    // windows of a block overlap almost completely, so the 25 visits of an 8-channel slice come while its lines are still in L2
    // (shift-major re-fetched the whole window from the fabric 25 times: 10.5 GB per launch against 0.33 GB of input)
    const int kd = KD == 25 ? chunk % 25 : (KD != 1 ? chunk / nc8 : 0), cc = KD == 25 ? chunk / 25 : (KD != 1 ? chunk - kd * nc8 : chunk);
    bool v = pval[j];
    int off = poff[j] + cc * 8;
    if constexpr (KD == 3) { v &= kd == 1 || ((dbits >> (2 * j + (kd >> 1))) & 1u) != 0; off += (kd - 1) * slice; }
    if constexpr (KD == 25) {                      // block (bi, bj) of the 15x15 filter: the patch shifted by (3bi - 6, 3bj - 6)
      const int bi = kd / 5, bj = kd - 5 * bi;
      v = ((smask[j] >> bi) & (smask[j] >> (8 + bj)) & 1u) != 0;
      off += (bi - 2) * rstep[j] + (bj - 2) * 3 * p.ld_in;
    }
    rv[j] = v;
    {
      // bounds-checked buffer load: pieces outside the image (zero padding, masked quarters) ask for an offset beyond the
      // tensor and get zeros from the hardware — no select when the piece goes to LDS; for 2-D layers the lane offset is a
      // constant of the piece and the chunk's channel offset is the instruction's scalar offset: no vector-ALU work at all
      u32x4 raw;
      if constexpr (KD == 1) raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, pboff[j], cc * 32, (MODE == 3 ? WINO_M3_RAW_AUX : 0)));
      else raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, v ? (unsigned)off << 2 : 0x80000000u, 0, 0));
      rp[j] = __builtin_bit_cast(f32x4, raw);
    }
    if constexpr (MODE == 3)      // (2-D layers only: g6d_wino_eligible)
      rm[j] = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(mul_rsrc, mboff[j], cc * 32, 0)));
  };
  auto load_raw = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < NPR; ++j) load_piece(j, chunk);
  };
  auto store_piece = [&](int j, int st, int chunk) {   // unconditional stores (a branch would serialise them behind one vmcnt(0) each);
    const int cc = KD == 25 ? chunk / 25 : (KD != 1 ? chunk % nc8 : chunk);       // the idle pieces of the last round go to a scratch row behind the stages
    {
      f32x4 v = rp[j];
      if constexpr (MODE != 0) {
        // packed: two v_pk_fma_f32 (+ two v_pk_mul_f32 for the multiplier) per piece; the shift comes from the zero row for a piece
        // outside the image (3-D layers: decided per chunk by the depth tap — one select on the LDS offset instead of four on the data)
        const int so = KD == 1 ? shoff[j] : (rv[j] ? shoff[j] : AFF0 + 2 * AFFG * p.Cin + (aoff[j] & 4));
        f32x4 sc = *reinterpret_cast<const f32x4*>(lds + AFF0 + aoff[j] + cc * 8);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(lds + so + cc * 8);
        f32x2w s0 = {sc[0], sc[1]}, s1 = {sc[2], sc[3]};
        if constexpr (MODE == 3) { s0 = s0 * f32x2w{rm[j][0], rm[j][1]}; s1 = s1 * f32x2w{rm[j][2], rm[j][3]}; }
        const f32x2w r0 = __builtin_elementwise_fma(f32x2w{v[0], v[1]}, s0, f32x2w{sh[0], sh[1]});
        const f32x2w r1 = __builtin_elementwise_fma(f32x2w{v[2], v[3]}, s1, f32x2w{sh[2], sh[3]});
        v = f32x4{r0.x, r0.y, r1.x, r1.y};
        if (p.in_relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      }
      f32x4* dst = reinterpret_cast<f32x4*>(__builtin_assume_aligned(lds + st * WSTAGE + lsto[j], 16));
      *dst = v;
    }
  };
  auto store_raw = [&](int st, int chunk) {
#pragma unroll
    for (int j = 0; j < NPR; ++j) store_piece(j, st, chunk);
  };
  if constexpr (MODE != 0) {                  // InstanceNorm affine tables -> LDS: [G][Cin] scales, then [G][Cin] shifts
    constexpr int G = MODE >= 2 ? NQ : 1;
    for (int i = tid; i < G * p.Cin; i += THREADS) {
      int g = 0;
      if constexpr (MODE >= 2) g = p.aff_div > 0 ? (quarter_of(p, i / p.Cin, NQ).n / p.D) / p.aff_div : 0;
      const int c = MODE >= 2 ? i % p.Cin : i;
      lds[AFF0 + i] = p.in_scale[g * p.Cin + c];
      lds[AFF0 + G * p.Cin + i] = p.in_shift[g * p.Cin + c];
    }
    for (int i = tid; i < p.Cin; i += THREADS) lds[AFF0 + 2 * G * p.Cin + i] = 0.f;      // the shifts of the pieces outside the image
    __syncthreads();
  }
  // ---- filter tiles: wave w moves (ab, half) pairs idx = 8w .. 8w+7, 1 KB (32 co x 32 B) per instruction
  // Direct-to-LDS copy in inline asm: with the builtin, hipcc books the copy on the LDS counter as well and then waits
  // lgkmcnt(0) in front of every fragment use (it cannot count mixed event types), which exposes the LDS latency of the
  // fragment requests just issued.  M0 = LDS byte address of the wave's 1 KB destination, each lane lands at +16*lane.
  const float* ubase = p.U + (size_t)n0 * 8;                 // wave-uniform; each lane adds 16 bytes x lane
  const unsigned lane16 = lane * 16;
  const unsigned lds_addr0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
  auto glds = [&](int chunk, int st, int idx) {
    const int ab = idx / NWN, h = idx % NWN;
    // wave-uniform source: a 64-bit base per chunk plus a 32-bit piece offset (two scalar adds per piece)
    const char* g = reinterpret_cast<const char*>(ubase) + (size_t)chunk * ((size_t)p.Cout * 512) + (unsigned)((ab * p.Cout + h * 32) * 32);
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr0 + 4u * (unsigned)(st * WSTAGE + RAWF + (ab * 32 * NWN + h * 32) * 8));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane16), "s"(g), "s"(dst) : "memory");
  };
  auto load_u = [&](int chunk, int st) {
#pragma unroll
    for (int k = 0; k < GL; ++k) glds(chunk, st, wave * GL + k);
  };

  // ---- fragment bases
  const int tl = li & 15, ty = tl >> 2, tx = tl & 3;
  const int apos = ((2 * wm + (li >> 4)) * WQ_PIX + (2 * ty) * 10 + 2 * tx) * WRAW_LD;
  const int abase01 = apos + 4 * (lh ^ (ty & 1));            // patch rows 2ty, 2ty+1   ((py >> 1) & 1 == ty & 1)
  const int abase23 = apos + 4 * (lh ^ (ty & 1) ^ 1);        // patch rows 2ty+2, 2ty+3
  const int bbase = RAWF + (wn * 32 + li) * 8 + 4 * (lh ^ ((li >> 3) & 1));
  constexpr int USTRIDE = 32 * NWN * 8;                      // floats between (a,b) positions of the filter image

  f32x16 acc[16];
#pragma unroll
  for (int a = 0; a < 16; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  load_u(c_first, 0);
  load_raw(c_first);
  store_raw(0, c_first);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  if (WINO_ABLATE == 9) { if (lds[tid] == 12345.f && p.out_pool) p.out_pool[tid] = lds[tid + 1]; return; }   // launch + prologue only
  // Software pipeline (one wave per SIMD: nothing but the wave's own instruction order hides latency).  The 16 MFMAs of
  // the LAST (a,b) group of chunk c-1 are deferred across the barrier with their operands held in registers (vD, uD):
  // chunk c opens with them, and every request of the chunk — the direct-to-LDS filter copies and the raw loads of
  // chunk c+1, the 16 raw-tile reads and the first 8 filter fragments of chunk c — is issued in the shadow of those MFMAs,
  // two to three memory instructions behind each.  Groups 0-2 then run back to back (the fragments of group g+2 are
  // requested at the start of group g), and the chunk ends by preparing vD / uD of its own last group.
  f32x4 vD[4], uD[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { vD[j] = f32x4{0.f, 0.f, 0.f, 0.f}; uD[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // Input transform V = B^T d B of (a,b) group `grp` (row combination a = grp of the raw rows, then the four column combinations)
  // as 16 packed two-float adds.  On this matrix pipe fp32 MFMAs and vector-ALU instructions do not overlap (measured,
  // tools/ubench/mfma_fill.hip: every VALU instruction beside v_mfma_f32_32x32x2_f32 costs its 4 issue cycles, the first one in an
  // MFMA gap 14; a packed add costs the same as a scalar one; LDS reads, scalar instructions and loads are nearly free): the
  // kernel therefore keeps the VALU count minimal and in ONE burst per group instead of spreading it over the gaps.
  auto xform = [&](int grp, const f32x4 (&dd)[4][4], f32x4 (&vv)[4]) {
    if (WABL(3)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) vv[q] = dd[grp][q];
      return;
    }
    f32x2 rl[4], rh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 &a = dd[grp == 0 ? 0 : grp == 2 ? 2 : 1][q], &b = dd[grp == 0 ? 2 : grp == 1 ? 2 : grp == 2 ? 1 : 3][q];
      if (grp == 1) { rl[q] = pk_add(lo2(a), lo2(b)); rh[q] = pk_add(hi2(a), hi2(b)); }
      else { rl[q] = pk_sub(lo2(a), lo2(b)); rh[q] = pk_sub(hi2(a), hi2(b)); }
    }
    vv[0] = cat2(pk_sub(rl[0], rl[2]), pk_sub(rh[0], rh[2]));
    vv[1] = cat2(pk_add(rl[1], rl[2]), pk_add(rh[1], rh[2]));
    vv[2] = cat2(pk_sub(rl[2], rl[1]), pk_sub(rh[2], rh[1]));
    vv[3] = cat2(pk_sub(rl[1], rl[3]), pk_sub(rh[1], rh[3]));
  };

  for (int cc = c_first; cc <= c_last; ++cc) {
    const int c = cc - c_first;                   // stage parity counts from the block's first chunk
    const float* S = lds + (c & 1) * WSTAGE;
    const int cn = min(cc + 1, c_last);           // the last chunk re-requests itself into the idle stage: no branches
    f32x4 d[4][4], ub[3][4], v[4];
    auto rd_d = [&](int i, int j) { if (!WABL(4)) d[i][j] = *reinterpret_cast<const f32x4*>(S + (i < 2 ? abase01 : abase23) + (i * 10 + j) * WRAW_LD); };
    auto rd_u = [&](int g, int j) { if (!WABL(4)) ub[g % 3][j] = *reinterpret_cast<const f32x4*>(S + bbase + (g * 4 + j) * USTRIDE); };
    // deferred group of chunk c-1 (gaps 0-15): every request of the chunk is issued in its shadow — the direct-to-LDS filter
    // pieces and the raw loads of chunk c+1 (the pieces first: invisible to the compiler's load counting, they must be OLDER
    // than the loads it waits for), the 16 raw-tile reads and the filter fragments of groups 0 and 1 of chunk c
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      // consecutive MFMAs go to DIFFERENT accumulators (k & 3): instructions issued between two MFMAs on the same accumulator
      // stretch the dependent pair
      acc[12 + (k & 3)] = __builtin_amdgcn_mfma_f32_32x32x2f32(vD[k & 3][k >> 2], uD[k & 3][k >> 2], acc[12 + (k & 3)], 0, 0, 0);
      if (!WABL(2)) {
        if (k < GL) glds(cn, (c & 1) ^ 1, wave * GL + k);
        else if (k - GL < NPR) load_piece(k - GL, cn);
      }
      if (k < 12) {                                // LDS requests in the order of first use: rows 0, 2, fragments 0, row 1, fragments 1, row 3
        const int j0 = 2 * (k & 1);
        if (k < 2) { rd_d(0, j0); rd_d(0, j0 + 1); }
        else if (k < 4) { rd_d(2, j0); rd_d(2, j0 + 1); }
        else if (k < 6) { rd_u(0, j0); rd_u(0, j0 + 1); }
        else if (k < 8) { rd_d(1, j0); rd_d(1, j0 + 1); }
        else if (k < 10) { rd_u(1, j0); rd_u(1, j0 + 1); }
        else { rd_d(3, j0); rd_d(3, j0 + 1); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      xform(i, d, v);                              // one VALU burst, then 16 MFMAs with the remaining requests in their gaps
      if (i == 2 && !WABL(2)) store_raw((c & 1) ^ 1, cn);      // raw pieces of chunk c+1 -> LDS: requested ~40 gaps ago
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        acc[i * 4 + (m & 3)] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[m & 3][m >> 2], ub[i][m & 3][m >> 2], acc[i * 4 + (m & 3)], 0, 0, 0);
        if (i < 2 && m < 2) { rd_u(i + 2, 2 * m); rd_u(i + 2, 2 * m + 1); }
        if constexpr (GL == 16) {                    // 16 filter pieces fill the deferred group: the raw pieces follow in the first gaps of group 0
          if (i == 0 && m >= 2 && m - 2 < NPR && !WABL(2)) load_piece(m - 2, cn);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    xform(3, d, vD);                               // operands of this chunk's deferred group
#pragma unroll
    for (int q = 0; q < 4; ++q) uD[q] = ub[0][q];  // ub[3 % 3] holds the group-3 fragments
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the direct-to-LDS filter tiles of the next chunk have landed
    if (!WABL(1)) __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 16; ++k)
    acc[12 + (k & 3)] = __builtin_amdgcn_mfma_f32_32x32x2f32(vD[k & 3][k >> 2], uD[k & 3][k >> 2], acc[12 + (k & 3)], 0, 0, 0);

  if (WINO_ABLATE == 7) {                             // no epilogue: one conditional dummy store keeps the accumulators alive
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < 16; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[a][r];
    if (t == 12345.f && p.out_pool) p.out_pool[tid] = t;
    return;
  }
  // ---------------------------------------------------------------- epilogue: A^T D A, bias, ReLU, stores, 2x2 max-pool
  const int co = n0 + wn * 32 + li;
  // output transform first: Y[r] = the 2x2 outputs {y00, y01, y10, y11} of accumulator row r (no bias yet)
  f32x4 Y[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float sr[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sr[0][j] = acc[0 + j][r] + acc[4 + j][r] + acc[8 + j][r];
      sr[1][j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
    }
    Y[r] = f32x4{sr[0][0] + sr[0][1] + sr[0][2], sr[0][1] - sr[0][2] - sr[0][3], sr[1][0] + sr[1][1] + sr[1][2], sr[1][1] - sr[1][2] - sr[1][3]};
  }
  if (p.splits > 1) {
    // the output transform is linear: partial OUTPUT tiles go to the workspace as [split][tile][r][thread] 16-byte pieces, and
    // the block that arrives last adds them in split order (g6d_common.h) and carries on with bias / ReLU / pool / statistics
    constexpr int TILE = THREADS * 64;
    const int ntiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x;
    float* part = p.ws + G6D_WS_COUNTERS + (size_t)tile * TILE + tid * 4;
    const size_t zstride = (size_t)ntiles * TILE;
#pragma unroll
    for (int r = 0; r < 16; ++r) g6d_store_wt(part + blockIdx.z * zstride + r * (THREADS * 4), Y[r]);
    if (!g6d_split_arrive(reinterpret_cast<int*>(p.ws) + tile, p.splits, reinterpret_cast<int*>(lds))) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) Y[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z0 = 0; z0 < p.splits; z0 += 2) {
      f32x4 v[2][16];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[u][r] = *reinterpret_cast<const f32x4*>(part + (size_t)min(z0 + u, p.splits - 1) * zstride + r * (THREADS * 4));
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[r] += v[0][r];
      if (z0 + 1 < p.splits) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[r] += v[1][r];
      }
    }
  }
  const float bv = p.bias ? p.bias[co] : 0.f;
  const bool do_relu = p.relu != 0;
  const bool do_stats = p.stats != nullptr;
  const QGeo geo[2] = {quarter_of(p, 2 * wm, NQ), quarter_of(p, 2 * wm + 1, NQ)};      // the wave's two quarters
  float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};       // statistics of this lane's column, per quarter of the wave
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    // accumulator row r of the 32x32 tile = tile (r&3) + 8*(r>>2) + 4*lh of the wave's 32
    const QGeo& g = geo[r >> 3];
    const int tyy = lh + 2 * ((r >> 2) & 1), txx = r & 3;
    const int n = g.n; const bool qv = g.valid;
    const int Hp = g.H >> 1, Wp = g.W >> 1;
    float y[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        y[a][b] = Y[r][2 * a + b] + bv;
        if (do_relu) y[a][b] = fmaxf(y[a][b], 0.f);
      }
    const int oy = g.oy0 + 2 * tyy, ox = g.ox0 + 2 * txx;
    if (p.out_full && qv) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (oy + a < g.H && ox + b < g.W) {
            p.out_full[(size_t)g.full_off + ((size_t)(n * g.H + oy + a) * g.W + ox + b) * g.ld_full + co] = y[a][b];
            if (do_stats) { st1[r >> 3] += y[a][b]; st2[r >> 3] += y[a][b] * y[a][b]; }
          }
    }
    if (p.out_pool && qv) {
      const int py = oy >> 1, px = ox >> 1;
      if (py < Hp && px < Wp)
        p.out_pool[(size_t)g.pool_off + ((size_t)(n * Hp + py) * Wp + px) * g.ld_pool + co] = fmaxf(fmaxf(y[0][0], y[0][1]), fmaxf(y[1][0], y[1][1]));
    }
  }
  if (do_stats) {
    // lanes l and l+32 hold the same column; [quarter][column][2] float partials in LDS (the stages are free: every wave has
    // passed the last barrier of the K loop), then one fp64 atomic per column and run of quarters with the same group
    float* sred = lds;
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      st1[h] += __shfl_xor(st1[h], 32, 64);
      st2[h] += __shfl_xor(st2[h], 32, 64);
      if (lh == 0) {
        sred[((2 * wm + h) * 32 * NWN + wn * 32 + li) * 2] = st1[h];
        sred[((2 * wm + h) * 32 * NWN + wn * 32 + li) * 2 + 1] = st2[h];
      }
    }
    __syncthreads();
    if (tid < 32 * NWN) {
      double a1 = 0.0, a2 = 0.0;
      int cur = -1;
      for (int q = 0; q < NQ; ++q) {
        const QGeo qg = quarter_of(p, q, NQ);
        if (!qg.valid) break;
        const int g = p.stats_div > 0 ? (qg.n / p.D) / p.stats_div : 0;
        if (g != cur && cur >= 0) {
          double* st = p.stats + ((size_t)cur * p.Cout + n0 + tid) * 2;
          atomicAdd(st, a1); atomicAdd(st + 1, a2); a1 = a2 = 0.0;
        }
        cur = g;
        a1 += (double)sred[(q * 32 * NWN + tid) * 2];
        a2 += (double)sred[(q * 32 * NWN + tid) * 2 + 1];
      }
      if (cur >= 0) {
        double* st = p.stats + ((size_t)cur * p.Cout + n0 + tid) * 2;
        atomicAdd(st, a1); atomicAdd(st + 1, a2);
      }
    }
    if (p.fin.scale) {                          // InstanceNorm finalisation by the launch's last block (G6dConv.fin_*)
      __syncthreads();
      g6d_finalize_stats(p.fin, gridDim.x * gridDim.y, reinterpret_cast<int*>(lds));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Reduced-precision trunk (G6dConv.math_mode semantics: 1 = bf16, 2 = fp16 operands, fp32 accumulation; opt-in speed mode, graded
// separately): the same F(2x2,3x3) formulation on v_mfma_f32_32x32x16_{bf16,f16}.  One K step of that instruction is 16 input
// channels, so a chunk is a PAIR of the fp32 kernel's 8-channel chunks:
//   raw patch  two 8-channel planes, each in the conflict-free layout of the fp32 kernel; lane half h transforms plane h (all 8 of
//              its channels: two 16-byte reads per patch position), in fp32 registers, and rounds the transformed values to the
//              operand type when they are packed into the A operand (8 values = channels 16 c + 8 h .. + 7);
//   filters    transformed and ROUNDED ON THE HOST: U16[chunk][ab][co][16] in the operand type, 32 bytes per (ab, co) row — the same
//              row size as the fp32 image, so the direct-to-LDS copy, the swizzle (halves swapped for co & 8) and the fragment read
//              (one ds_read_b128 = the 8 channels of lane half h) are unchanged, and a chunk pair fits the 32 KB filter slot;
//   MFMAs      16 per chunk (one per (a,b) position) of 8 passes instead of 2 x 64 of 16 passes: the loop is bound by the input
//              transform (~200 vector instructions) and by streaming 58 KB per chunk into LDS, not by the matrix pipe.
// Epilogue as the fp32 trunk kernel (output transform, bias, ReLU, full / pooled fp32 outputs, chunk split with in-kernel hand-off).
// Epilogue shared by the 16-bit kernels (the fp32 kernel keeps its own, identical copy inline: it is scheduled by hand): output
// transform A^T D A, chunk-split hand-off, bias, ReLU, full / pooled stores, per-(group, channel) statistics and their finalisation.
template <int THREADS, int NWN>
__device__ __forceinline__ void wino_epilogue(const WinoArgs& p, f32x16 (&acc)[16], float* lds, int tid, int wm, int wn, int li, int lh, int n0) {
  const int co = n0 + wn * 32 + li;
  f32x4 Y[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float sr[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sr[0][j] = acc[0 + j][r] + acc[4 + j][r] + acc[8 + j][r];
      sr[1][j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
    }
    Y[r] = f32x4{sr[0][0] + sr[0][1] + sr[0][2], sr[0][1] - sr[0][2] - sr[0][3], sr[1][0] + sr[1][1] + sr[1][2], sr[1][1] - sr[1][2] - sr[1][3]};
  }
  if (p.splits > 1) {
    constexpr int TILE = THREADS * 64;
    const int ntiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x;
    float* part = p.ws + G6D_WS_COUNTERS + (size_t)tile * TILE + tid * 4;
    const size_t zstride = (size_t)ntiles * TILE;
#pragma unroll
    for (int r = 0; r < 16; ++r) g6d_store_wt(part + blockIdx.z * zstride + r * (THREADS * 4), Y[r]);
    if (!g6d_split_arrive(reinterpret_cast<int*>(p.ws) + tile, p.splits, reinterpret_cast<int*>(lds))) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) Y[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.splits; ++z) {
      f32x4 v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = *reinterpret_cast<const f32x4*>(part + (size_t)z * zstride + r * (THREADS * 4));
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[r] += v[r];
    }
  }
  const float bv = p.bias ? p.bias[co] : 0.f;
  const bool do_relu = p.relu != 0;
  const bool do_stats = p.stats != nullptr;
  const QGeo geo[2] = {quarter_of(p, 2 * wm), quarter_of(p, 2 * wm + 1)};
  float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const QGeo& g = geo[r >> 3];
    const int tyy = lh + 2 * ((r >> 2) & 1), txx = r & 3;
    const int n = g.n; const bool qv = g.valid;
    const int Hp = g.H >> 1, Wp = g.W >> 1;
    float y[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        y[a][b] = Y[r][2 * a + b] + bv;
        if (do_relu) y[a][b] = fmaxf(y[a][b], 0.f);
      }
    const int oy = g.oy0 + 2 * tyy, ox = g.ox0 + 2 * txx;
    if (p.out_full && qv) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (oy + a < g.H && ox + b < g.W) {
            p.out_full[(size_t)g.full_off + ((size_t)(n * g.H + oy + a) * g.W + ox + b) * g.ld_full + co] = y[a][b];
            if (do_stats) { st1[r >> 3] += y[a][b]; st2[r >> 3] += y[a][b] * y[a][b]; }
          }
    }
    if (p.out_pool && qv) {
      const int py = oy >> 1, px = ox >> 1;
      if (py < Hp && px < Wp)
        p.out_pool[(size_t)g.pool_off + ((size_t)(n * Hp + py) * Wp + px) * g.ld_pool + co] = fmaxf(fmaxf(y[0][0], y[0][1]), fmaxf(y[1][0], y[1][1]));
    }
  }
  if (do_stats) {
    float* sred = lds;
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      st1[h] += __shfl_xor(st1[h], 32, 64);
      st2[h] += __shfl_xor(st2[h], 32, 64);
      if (lh == 0) {
        sred[((2 * wm + h) * 32 * NWN + wn * 32 + li) * 2] = st1[h];
        sred[((2 * wm + h) * 32 * NWN + wn * 32 + li) * 2 + 1] = st2[h];
      }
    }
    __syncthreads();
    if (tid < 32 * NWN) {
      double a1 = 0.0, a2 = 0.0;
      int cur = -1;
      for (int q = 0; q < 4; ++q) {
        const QGeo qg = quarter_of(p, q);
        if (!qg.valid) break;
        const int g = p.stats_div > 0 ? (qg.n / p.D) / p.stats_div : 0;
        if (g != cur && cur >= 0) {
          double* st = p.stats + ((size_t)cur * p.Cout + n0 + tid) * 2;
          atomicAdd(st, a1); atomicAdd(st + 1, a2); a1 = a2 = 0.0;
        }
        cur = g;
        a1 += (double)sred[(q * 32 * NWN + tid) * 2];
        a2 += (double)sred[(q * 32 * NWN + tid) * 2 + 1];
      }
      if (cur >= 0) {
        double* st = p.stats + ((size_t)cur * p.Cout + n0 + tid) * 2;
        atomicAdd(st, a1); atomicAdd(st + 1, a2);
      }
    }
    if (p.fin.scale) {
      __syncthreads();
      g6d_finalize_stats(p.fin, gridDim.x * gridDim.y, reinterpret_cast<int*>(lds));
    }
  }
}

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ auto a8_dummy(T a, T b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));

template <int MM, int NWN>
__global__ void __launch_bounds__(128 * NWN, 1) wino16_conv3x3_kernel(const WinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int THREADS = 128 * NWN;
  constexpr int NPR = (1600 + THREADS - 1) / THREADS;         // 16-byte raw pieces per thread and chunk: 2 planes x 4 quarters x 100 x 2
  constexpr int WU_FLOATS = 16 * 32 * NWN * 8;                // [ab][co][16 x 2 bytes]
  constexpr int WSTAGE = 2 * WRAW_FLOATS + WU_FLOATS + 4 * THREADS;
  using hv4 = typename std::conditional<MM == 1, b16x4, h16x4>::type;
  using hv8 = typename std::conditional<MM == 1, bf16x8, f16x8>::type;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * (32 * NWN);
  const int nc16 = p.Cin >> 4;
  const int c_first = blockIdx.z * p.chunks_per_split;
  const int c_last = min(nc16, c_first + p.chunks_per_split) - 1;

  unsigned pboff[NPR]; int lsto[NPR];
#pragma unroll
  for (int j = 0; j < NPR; ++j) {
    const int idx = tid + THREADS * j;
    const int plane = idx / 800, r0 = idx - plane * 800;
    const int q = r0 / 200, r = r0 - q * 200, pp = r >> 1, half = r & 1;
    const int py = pp / 10, px = pp - py * 10;
    const QGeo g = quarter_of(p, q < 4 ? q : 0);
    const int iy = g.oy0 + py - 1, ix = g.ox0 + px - 1;
    const bool v = (idx < 1600) & g.valid & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
    pboff[j] = v ? (unsigned)(g.in_off + ((g.n * g.H + iy) * g.W + ix) * g.ld_in + 8 * plane + 4 * half) << 2 : 0x80000000u;
    lsto[j] = idx < 1600 ? plane * WRAW_FLOATS + (q * WQ_PIX + pp) * WRAW_LD + 4 * (half ^ ((py >> 1) & 1))
                         : 2 * WRAW_FLOATS + WU_FLOATS + 4 * tid;
  }
  const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  f32x4 rp[NPR];
  auto load_raw = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < NPR; ++j)
      rp[j] = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, pboff[j], chunk * 64, 0)));
  };
  auto store_raw = [&](int st) {
#pragma unroll
    for (int j = 0; j < NPR; ++j)
      *reinterpret_cast<f32x4*>(__builtin_assume_aligned(lds + st * WSTAGE + lsto[j], 16)) = rp[j];
  };
  const char* ubase = reinterpret_cast<const char*>(p.U) + (size_t)n0 * 32;       // 32 bytes per (ab, co) row
  const unsigned lane16 = lane * 16;
  const unsigned lds_addr0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
  auto glds = [&](int chunk, int st, int idx) {
    const int ab = idx / NWN, h = idx % NWN;
    const char* g = ubase + (size_t)chunk * ((size_t)p.Cout * 512) + (unsigned)((ab * p.Cout + h * 32) * 32);
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr0 + 4u * (unsigned)(st * WSTAGE + 2 * WRAW_FLOATS + (ab * 32 * NWN + h * 32) * 8));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane16), "s"(g), "s"(dst) : "memory");
  };
  auto load_u = [&](int chunk, int st) {
#pragma unroll
    for (int k = 0; k < 8; ++k) glds(chunk, st, wave * 8 + k);
  };
  const int tl = li & 15, ty = tl >> 2, tx = tl & 3;
  const int apos = lh * WRAW_FLOATS + ((2 * wm + (li >> 4)) * WQ_PIX + (2 * ty) * 10 + 2 * tx) * WRAW_LD;     // lane half h reads plane h
  const int bbase = 2 * WRAW_FLOATS + (wn * 32 + li) * 8 + 4 * (lh ^ ((li >> 3) & 1));
  constexpr int USTRIDE = 32 * NWN * 8;

  f32x16 acc[16];
#pragma unroll
  for (int a = 0; a < 16; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  load_u(c_first, 0);
  load_raw(c_first);
  store_raw(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // V = B^T d B of (a,b) group `grp` on 4 channels (the fp32 kernel's transform)
  auto xform = [&](int grp, const f32x4 (&dd)[4][4], f32x4 (&vv)[4]) {
    f32x2 rl[4], rh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 &a = dd[grp == 0 ? 0 : grp == 2 ? 2 : 1][q], &b = dd[grp == 0 ? 2 : grp == 1 ? 2 : grp == 2 ? 1 : 3][q];
      if (grp == 1) { rl[q] = pk_add(lo2(a), lo2(b)); rh[q] = pk_add(hi2(a), hi2(b)); }
      else { rl[q] = pk_sub(lo2(a), lo2(b)); rh[q] = pk_sub(hi2(a), hi2(b)); }
    }
    vv[0] = cat2(pk_sub(rl[0], rl[2]), pk_sub(rh[0], rh[2]));
    vv[1] = cat2(pk_add(rl[1], rl[2]), pk_add(rh[1], rh[2]));
    vv[2] = cat2(pk_sub(rl[2], rl[1]), pk_sub(rh[2], rh[1]));
    vv[3] = cat2(pk_sub(rl[1], rl[3]), pk_sub(rh[1], rh[3]));
  };

  for (int cc = c_first; cc <= c_last; ++cc) {
    const int c = cc - c_first;
    const float* S = lds + (c & 1) * WSTAGE;
    if (cc < c_last && !WABL(2)) {                // the copies of the next chunk first (they must be older than the loads hipcc counts)
      load_u(cc + 1, (c & 1) ^ 1);
      load_raw(cc + 1);
    }
    hv4 vh[16][2];                                // transformed patch, rounded: [a*4 + b][4-channel half of the lane's plane]
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
      f32x4 d[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!WABL(4) || cc == c_first) d[i][j] = *reinterpret_cast<const f32x4*>(S + apos + 4 * (hs ^ (ty & 1) ^ (i >> 1)) + (i * 10 + j) * WRAW_LD);
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        f32x4 v[4];
        if (WABL(3)) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = d[grp][q];
        } else
        xform(grp, d, v);
#pragma unroll
        for (int q = 0; q < 4; ++q) vh[grp * 4 + q][hs] = __builtin_convertvector(v[q], hv4);
      }
    }
#pragma unroll
    for (int ab = 0; ab < 16; ++ab) {
      f32x4 uraw;
      if (WABL(4)) uraw = __builtin_bit_cast(f32x4, a8_dummy(vh[ab][0], vh[ab][1])); else
      uraw = *reinterpret_cast<const f32x4*>(S + bbase + ab * USTRIDE);
      const hv8 a8 = __builtin_shufflevector(vh[ab][0], vh[ab][1], 0, 1, 2, 3, 4, 5, 6, 7);
      const hv8 b8 = __builtin_bit_cast(hv8, uraw);
      if (WINO_ABLATE == 6) { acc[ab][0] += __builtin_bit_cast(f32x4, a8)[0] * __builtin_bit_cast(f32x4, b8)[1]; continue; }
      if constexpr (MM == 1) acc[ab] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[ab], 0, 0, 0);
      else acc[ab] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[ab], 0, 0, 0);
    }
    if (cc < c_last && !WABL(2)) store_raw((c & 1) ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!WABL(1)) __syncthreads();
  }

  wino_epilogue<THREADS, NWN>(p, acc, lds, tid, wm, wn, li, lh, n0);
}

// Two waves per SIMD for the 16-bit trunk (MODE 0, 2-D, un-split launches): with one 512-register wave per SIMD the chunk of
// wino16_conv3x3_kernel costs ~8x its matrix-core time because nothing overlaps (ablation table).  Here a block is 512
// threads = 8 waves with 128 accumulator registers each: the 16 (a,b) positions of a 32-tile x 32-channel output tile are split
// over two waves (a < 2 / a >= 2), the input transform is done ONCE per tile (the 2 x 2 kernel does it in both channel waves) by
// 512 work items (tile, 4-channel group, (a,b) half) that leave the rounded V in LDS in fragment order, and the MFMA phase reads
// V and U fragments with one ds_read_b128 each.  Per chunk: requests of chunk c+2 | transform of chunk c -> V | barrier | 8 MFMAs
// per wave | raw pieces of chunk c+1 -> LDS | barrier, with the requests two chunks ahead.  LDS: one raw stage (two planes) + 3
// filter stages + V = 155 KB.
// The two partial output transforms of a tile (the transform is linear in the (a,b) accumulators) are exchanged through LDS:
// each wave finishes the 8 accumulator rows (= one quarter) it owns.
template <int MM>
__global__ void __launch_bounds__(512, 1) wino16b_conv3x3_kernel(const WinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int RAWS = 2 * WRAW_FLOATS;                       // floats of the raw stage (two 8-channel planes)
  constexpr int US = 16 * 64 * 8;                             // floats of a filter stage ([ab][co][16 x 2 bytes])
  constexpr int RAW0 = 0, U0 = RAWS, V0 = U0 + 3 * US, SCR = V0 + US;      // ONE raw stage, THREE filter stages, V, scratch row
  using hv4 = typename std::conditional<MM == 1, b16x4, h16x4>::type;
  using hv8 = typename std::conditional<MM == 1, bf16x8, f16x8>::type;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.y * 64;
  const int nc16 = p.Cin >> 4;

  // ---- raw pieces: idx = tid + 512 j < 1600 = 2 planes x 4 quarters x 100 positions x 2 halves
  unsigned pboff[4]; int lsto[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + 512 * j;
    const int plane = idx / 800, r0 = idx - plane * 800;
    const int q = r0 / 200, r = r0 - q * 200, pp = r >> 1, half = r & 1;
    const int py = pp / 10, px = pp - py * 10;
    const QGeo g = quarter_of(p, q < 4 ? q : 0);
    const int iy = g.oy0 + py - 1, ix = g.ox0 + px - 1;
    const bool v = (idx < 1600) & g.valid & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
    pboff[j] = v ? (unsigned)(g.in_off + ((g.n * g.H + iy) * g.W + ix) * g.ld_in + 8 * plane + 4 * half) << 2 : 0x80000000u;
    lsto[j] = idx < 1600 ? plane * WRAW_FLOATS + (q * WQ_PIX + pp) * WRAW_LD + 4 * (half ^ ((py >> 1) & 1)) : SCR - RAW0 + 4 * (tid & 63);
  }
  const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // Requests run TWO chunks ahead (a chunk's work is shorter than a first-touch load: one chunk of lead left the kernel waiting on
  // memory, ablation 2): the raw pieces of chunk c+2 wait in the register set of c's parity, the filter tiles in the third stage.
  f32x4 rp[2][4];
  auto load_raw = [&](auto S, int chunk) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      rp[decltype(S)::value][j] = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, pboff[j], chunk * 64, 0)));
  };
  auto store_raw = [&](auto S) {       // (the idle pieces of the last round land in the scratch row)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(__builtin_assume_aligned(lds + RAW0 + lsto[j], 16)) = rp[decltype(S)::value][j];
  };
  // ---- filter tiles: 32 pieces of 1 KB per chunk, wave w moves pieces 4w .. 4w+3
  const char* ubase = reinterpret_cast<const char*>(p.U) + (size_t)n0 * 32;
  const unsigned lane16 = lane * 16;
  const unsigned lds_addr0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
  auto load_u = [&](int chunk, int st) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = wave * 4 + k, ab = idx >> 1, h = idx & 1;
      const char* g = ubase + (size_t)chunk * ((size_t)p.Cout * 512) + (unsigned)((ab * p.Cout + h * 32) * 32);
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr0 + 4u * (unsigned)(U0 + st * US + (ab * 64 + h * 32) * 8));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(lane16), "s"(g), "s"(dst) : "memory");
    }
  };
  // ---- transform phase: wave w = (quarter qa, (a,b) half ha); lane = (tile of the quarter, 4-channel group g of the 16)
  const int qa = wave & 3, ha = wave >> 2;
  const int tl = lane & 15, ty = tl >> 2, tx = tl & 3, g4 = lane >> 4;
  const int araw = (g4 >> 1) * WRAW_FLOATS + (qa * WQ_PIX + (2 * ty) * 10 + 2 * tx) * WRAW_LD;
  const int ahalf = g4 & 1;
  // V image: [ab][tile 0..63][16 halfs]; this item's 8 bytes of row (ab, qa*16 + tl) at halfs 4*g4
  // (rows of tiles with bit 3 set carry their two 8-half groups swapped, like the filter rows: the 16-lane groups of the fragment
  // reads then cover all banks)
  const int vwr = V0 * 4 + ((8 * ha) * 64 + qa * 16 + tl) * 32 + 8 * (g4 ^ (2 * ((tl >> 3) & 1)));     // bytes from the LDS base, + ab_local * 2048
  // ---- MFMA phase: wave w = (wa = (a,b) half, wm, wn); lane = (tile / channel li, k half lh)
  const int wa = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int vrd = V0 * 4 + ((8 * wa) * 64 + wm * 32 + li) * 32 + 16 * (lh ^ ((li >> 3) & 1));          // bytes, + ab_local * 2048
  const int urd = (wn * 32 + li) * 8 + 4 * (lh ^ ((li >> 3) & 1)) + (8 * wa) * 512;     // floats inside a filter stage, + ab_local * 512

  f32x16 acc[8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
  load_u(0, 0);
  load_raw(P0{}, 0);
  store_raw(P0{});                                  // chunk 0 -> LDS (waits for its pieces)
  if (nc16 > 1) { load_u(1, 1); load_raw(P1{}, 1); }    // chunk 1: filter stage 1, register set 1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  char* const ldsb = reinterpret_cast<char*>(lds);
  // one chunk: requests of c+2 | transform c -> V | barrier | 8 MFMAs | raw pieces of c+1 (set of c+1's parity) -> LDS | barrier
  auto chunk_step = [&](auto PAR, int c) {
    using NXT = std::integral_constant<int, decltype(PAR)::value ^ 1>;
    const float* R = lds + RAW0;
    const float* Ub = lds + U0 + (c % 3) * US;
    const bool more = c + 2 < nc16;
    if (more && !WABL(2)) {                         // the filter pieces first: they must be OLDER than the raw loads the compiler counts
      load_u(c + 2, (c + 2) % 3);
      load_raw(PAR, c + 2);                         // set of c's parity: its previous content (chunk c) went to LDS an iteration ago
    }
    // transform of the item's 4 channels: rows 0,1,2 (ha = 0: groups a = 0, 1) or 1,2,3 (ha = 1: groups a = 2, 3); the half is
    // wave-uniform: a scalar branch, no per-lane selects
    auto transform = [&](auto HA) {
      constexpr int H = decltype(HA)::value;
      f32x4 d[3][4];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = i + H;
          d[i][j] = *reinterpret_cast<const f32x4*>(R + araw + 4 * (ahalf ^ ((ty + (row >> 1)) & 1)) + (row * 10 + j) * WRAW_LD);
        }
#pragma unroll
      for (int gl = 0; gl < 2; ++gl) {
        // H = 0: a=0: d0 - d2, a=1: d1 + d2;   H = 1 (rows 1,2,3 in d[0..2]): a=2: d2 - d1 = d[1] - d[0], a=3: d1 - d3 = d[0] - d[2]
        f32x2 rl[4], rh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4& x = H == 0 ? (gl == 0 ? d[0][q] : d[1][q]) : (gl == 0 ? d[1][q] : d[0][q]);
          const f32x4& y = H == 0 ? d[2][q] : (gl == 0 ? d[0][q] : d[2][q]);
          if (H == 0 && gl == 1) { rl[q] = pk_add(lo2(x), lo2(y)); rh[q] = pk_add(hi2(x), hi2(y)); }
          else { rl[q] = pk_sub(lo2(x), lo2(y)); rh[q] = pk_sub(hi2(x), hi2(y)); }
        }
        f32x4 vv[4];
        vv[0] = cat2(pk_sub(rl[0], rl[2]), pk_sub(rh[0], rh[2]));
        vv[1] = cat2(pk_add(rl[1], rl[2]), pk_add(rh[1], rh[2]));
        vv[2] = cat2(pk_sub(rl[2], rl[1]), pk_sub(rh[2], rh[1]));
        vv[3] = cat2(pk_sub(rl[1], rl[3]), pk_sub(rh[1], rh[3]));
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<hv4*>(__builtin_assume_aligned(ldsb + vwr + (gl * 4 + q) * 2048, 8)) = __builtin_convertvector(vv[q], hv4);
      }
    };
    if (!WABL(3) || c == 0) { if (ha == 0) transform(std::integral_constant<int, 0>{}); else transform(std::integral_constant<int, 1>{}); }
    if (!WABL(1)) __syncthreads();                  // V complete; every read of the raw stage done
    hv8 a8 = {}, b8 = {};
#pragma unroll
    for (int ab = 0; ab < 8; ++ab) {
      if (!WABL(4) || (c == 0 && ab == 0)) {
        a8 = *reinterpret_cast<const hv8*>(__builtin_assume_aligned(ldsb + vrd + ab * 2048, 16));
        b8 = __builtin_bit_cast(hv8, *reinterpret_cast<const f32x4*>(Ub + urd + ab * 512));
      }
      if (WINO_ABLATE == 6) { acc[ab][0] += (float)a8[0] * (float)b8[1]; continue; }
      if constexpr (MM == 1) acc[ab] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[ab], 0, 0, 0);
      else acc[ab] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[ab], 0, 0, 0);
    }
    if (c + 1 < nc16 && !WABL(2)) {
      // chunk c+1 (requested an iteration ago) into the raw stage; everything older than the four raw loads of this iteration has
      // landed then — its filter tiles too (the compiler's own wait before the stores asks for the same)
      if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      store_raw(NXT{});
    }
    if (!WABL(1)) __syncthreads();
  };
  for (int c = 0; c < nc16; c += 2) {
    chunk_step(P0{}, c);
    if (c + 1 < nc16) chunk_step(P1{}, c + 1);
  }

  // ---------------------------------------------------------------- epilogue
  // partial output transform over the wave's 8 positions: A^T = [1 1 1 0; 0 1 -1 -1] — rows a = 0,1 (wa = 0) or a = 2,3 (wa = 1)
  f32x4 Y[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float s0[4], s1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (wa == 0) { s0[j] = acc[j][r] + acc[4 + j][r]; s1[j] = acc[4 + j][r]; }
      else { s0[j] = acc[j][r]; s1[j] = -acc[j][r] - acc[4 + j][r]; }
    }
    Y[r] = f32x4{s0[0] + s0[1] + s0[2], s0[1] - s0[2] - s0[3], s1[0] + s1[1] + s1[2], s1[1] - s1[2] - s1[3]};
  }
  // the rows this wave does not own go to its slot (8 rows x 64 lanes x 16 bytes; the stages are free behind the loop's last barrier)
  f32x4* const X = reinterpret_cast<f32x4*>(lds);
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) X[(wave * 8 + rr) * 64 + lane] = Y[(wa == 0 ? 8 : 0) + rr];
  __syncthreads();
  f32x4 Yo[8];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) Yo[rr] = Y[(wa == 0 ? 0 : 8) + rr] + X[((wave ^ 4) * 8 + rr) * 64 + lane];

  const int co = n0 + wn * 32 + li;
  const float bv = p.bias ? p.bias[co] : 0.f;
  const bool do_relu = p.relu != 0;
  const QGeo g = quarter_of(p, 2 * wm + wa);              // accumulator rows 8 wa .. 8 wa + 7 = the tiles of that quarter
  if (!g.valid) return;
  const int Hp = g.H >> 1, Wp = g.W >> 1, n = g.n;
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int tyy = lh + 2 * ((rr >> 2) & 1), txx = rr & 3;
    float y[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        y[a][b] = Yo[rr][2 * a + b] + bv;
        if (do_relu) y[a][b] = fmaxf(y[a][b], 0.f);
      }
    const int oy = g.oy0 + 2 * tyy, ox = g.ox0 + 2 * txx;
    if (p.out_full) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (oy + a < g.H && ox + b < g.W)
            p.out_full[(size_t)g.full_off + ((size_t)(n * g.H + oy + a) * g.W + ox + b) * g.ld_full + co] = y[a][b];
    }
    if (p.out_pool) {
      const int py = oy >> 1, px = ox >> 1;
      if (py < Hp && px < Wp)
        p.out_pool[(size_t)g.pool_off + ((size_t)(n * Hp + py) * Wp + px) * g.ld_pool + co] = fmaxf(fmaxf(y[0][0], y[0][1]), fmaxf(y[1][0], y[1][1]));
    }
  }
}

// The conv family on the 16-bit kernel (G6dConv.weight_wino16, math_mode 1 / 2): the stride-1 3x3 / 3x3x3 layers of the selector, the
// refiner feature net and the volume net with the operand prologues of the fp32 kernel — MODE 0 none, 1 InstanceNorm affine(+ReLU)
// with one table, 2 one table per image group (tables of the block's four quarters in LDS), 3 query x reference multiplier + tables
// per quarter, shared input images / per-group multiplier maps of a query batch — applied in fp32 when the raw piece is written to
// LDS (zero padding stays exactly zero), KD = 3 with the depth taps folded into the reduction, statistics / finalisation epilogue.
// Same chunk (16 channels = two planes), filters and MFMA scheme as wino16_conv3x3_kernel; loads are plain (selected) global loads.
template <int MM, int MODE, int KD>
__global__ void __launch_bounds__(256, 1) wino16_conv_kernel(const WinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NWN = 2, THREADS = 256, NPR = 7;
  constexpr int WU_FLOATS = 16 * 32 * NWN * 8;
  constexpr int WSTAGE = 2 * WRAW_FLOATS + WU_FLOATS + 4 * THREADS;
  constexpr int AFF0 = 2 * WSTAGE;
  using hv4 = typename std::conditional<MM == 1, b16x4, h16x4>::type;
  using hv8 = typename std::conditional<MM == 1, bf16x8, f16x8>::type;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * (32 * NWN);
  const int nc16 = p.Cin >> 4;
  const int c_first = blockIdx.z * p.chunks_per_split;
  const int c_last = min(KD * nc16, c_first + p.chunks_per_split) - 1;

  int poff[NPR], lsto[NPR], moff[MODE == 3 ? NPR : 1], aoff[MODE != 0 ? NPR : 1];
  bool pval[NPR];
  unsigned dbits = 0;
#pragma unroll
  for (int j = 0; j < NPR; ++j) {
    const int idx = tid + THREADS * j;
    const int plane = idx / 800, r0 = idx - plane * 800;
    const int q = r0 / 200, r = r0 - q * 200, pp = r >> 1, half = r & 1;
    const int py = pp / 10, px = pp - py * 10;
    const QGeo g = quarter_of(p, q < 4 ? q : 0);
    const int n = g.n;
    const int iy = g.oy0 + py - 1, ix = g.ox0 + px - 1;
    pval[j] = (idx < 1600) & g.valid & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
    const int n_in = p.img_mod > 0 ? ((n / p.D) % p.img_mod) * p.D + n % p.D : n;
    poff[j] = pval[j] ? g.in_off + ((n_in * g.H + iy) * g.W + ix) * g.ld_in + 8 * plane + 4 * half : 0;
    lsto[j] = idx < 1600 ? plane * WRAW_FLOATS + (q * WQ_PIX + pp) * WRAW_LD + 4 * (half ^ ((py >> 1) & 1))
                         : 2 * WRAW_FLOATS + WU_FLOATS + 4 * tid;
    if constexpr (MODE == 3) moff[j] = pval[j] ? (((p.mul_div > 0 ? n / p.mul_div : 0) * p.H + iy) * p.W + ix) * p.Cin + 8 * plane + 4 * half : 0;
    if constexpr (MODE != 0) aoff[j] = idx < 1600 ? (MODE >= 2 ? q * p.Cin : 0) + 8 * plane + 4 * half : 0;
    if constexpr (KD == 3) { const int dd = n % p.D; dbits |= (unsigned)(dd > 0) << (2 * j) | (unsigned)(dd < p.D - 1) << (2 * j + 1); }
  }
  const int slice = p.H * p.W * p.ld_in;
  f32x4 rp[NPR], rm[MODE == 3 ? NPR : 1];
  bool rv[NPR];
  auto load_raw = [&](int chunk) {
    const int kd = KD == 3 ? chunk / nc16 : 0, cc = KD == 3 ? chunk - kd * nc16 : chunk;
#pragma unroll
    for (int j = 0; j < NPR; ++j) {
      bool v = pval[j];
      int off = poff[j] + cc * 16;
      if constexpr (KD == 3) { v &= kd == 1 || ((dbits >> (2 * j + (kd >> 1))) & 1u) != 0; off += (kd - 1) * slice; }
      rv[j] = v;
      rp[j] = ldg4(p.in, v ? off : 0);
      if constexpr (MODE == 3) rm[j] = ldg4(p.mul, v ? moff[j] + cc * 16 : 0);
    }
  };
  auto store_raw = [&](int st, int chunk) {
    const int cc = KD == 3 ? chunk % nc16 : chunk;
#pragma unroll
    for (int j = 0; j < NPR; ++j) {
      f32x4 v = rp[j];
      if constexpr (MODE == 3) v *= rm[j];
      if constexpr (MODE != 0) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(lds + AFF0 + aoff[j] + cc * 16);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(lds + AFF0 + (MODE >= 2 ? 4 : 1) * p.Cin + aoff[j] + cc * 16);
        v = v * sc + sh;
        if (p.in_relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      }
      *reinterpret_cast<f32x4*>(__builtin_assume_aligned(lds + st * WSTAGE + lsto[j], 16)) = rv[j] ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if constexpr (MODE != 0) {
    constexpr int G = MODE >= 2 ? 4 : 1;
    for (int i = tid; i < G * p.Cin; i += THREADS) {
      int g = 0;
      if constexpr (MODE >= 2) g = p.aff_div > 0 ? (quarter_of(p, i / p.Cin).n / p.D) / p.aff_div : 0;
      const int c = MODE >= 2 ? i % p.Cin : i;
      lds[AFF0 + i] = p.in_scale[g * p.Cin + c];
      lds[AFF0 + G * p.Cin + i] = p.in_shift[g * p.Cin + c];
    }
    __syncthreads();
  }
  const char* ubase = reinterpret_cast<const char*>(p.U) + (size_t)n0 * 32;
  const unsigned lane16 = lane * 16;
  const unsigned lds_addr0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
  auto glds = [&](int chunk, int st, int idx) {
    const int ab = idx / NWN, h = idx % NWN;
    const char* g = ubase + (size_t)chunk * ((size_t)p.Cout * 512) + (unsigned)((ab * p.Cout + h * 32) * 32);
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr0 + 4u * (unsigned)(st * WSTAGE + 2 * WRAW_FLOATS + (ab * 32 * NWN + h * 32) * 8));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane16), "s"(g), "s"(dst) : "memory");
  };
  auto load_u = [&](int chunk, int st) {
#pragma unroll
    for (int k = 0; k < 8; ++k) glds(chunk, st, wave * 8 + k);
  };
  const int tl = li & 15, ty = tl >> 2, tx = tl & 3;
  const int apos = lh * WRAW_FLOATS + ((2 * wm + (li >> 4)) * WQ_PIX + (2 * ty) * 10 + 2 * tx) * WRAW_LD;
  const int bbase = 2 * WRAW_FLOATS + (wn * 32 + li) * 8 + 4 * (lh ^ ((li >> 3) & 1));
  constexpr int USTRIDE = 32 * NWN * 8;

  f32x16 acc[16];
#pragma unroll
  for (int a = 0; a < 16; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  load_u(c_first, 0);
  load_raw(c_first);
  store_raw(0, c_first);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto xform = [&](int grp, const f32x4 (&dd)[4][4], f32x4 (&vv)[4]) {
    f32x2 rl[4], rh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 &a = dd[grp == 0 ? 0 : grp == 2 ? 2 : 1][q], &b = dd[grp == 0 ? 2 : grp == 1 ? 2 : grp == 2 ? 1 : 3][q];
      if (grp == 1) { rl[q] = pk_add(lo2(a), lo2(b)); rh[q] = pk_add(hi2(a), hi2(b)); }
      else { rl[q] = pk_sub(lo2(a), lo2(b)); rh[q] = pk_sub(hi2(a), hi2(b)); }
    }
    vv[0] = cat2(pk_sub(rl[0], rl[2]), pk_sub(rh[0], rh[2]));
    vv[1] = cat2(pk_add(rl[1], rl[2]), pk_add(rh[1], rh[2]));
    vv[2] = cat2(pk_sub(rl[2], rl[1]), pk_sub(rh[2], rh[1]));
    vv[3] = cat2(pk_sub(rl[1], rl[3]), pk_sub(rh[1], rh[3]));
  };

  for (int cc = c_first; cc <= c_last; ++cc) {
    const int c = cc - c_first;
    const float* S = lds + (c & 1) * WSTAGE;
    if (cc < c_last) {
      load_u(cc + 1, (c & 1) ^ 1);
      load_raw(cc + 1);
    }
    hv4 vh[16][2];
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
      f32x4 d[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          d[i][j] = *reinterpret_cast<const f32x4*>(S + apos + 4 * (hs ^ (ty & 1) ^ (i >> 1)) + (i * 10 + j) * WRAW_LD);
#pragma unroll
      for (int grp = 0; grp < 4; ++grp) {
        f32x4 v[4];
        xform(grp, d, v);
#pragma unroll
        for (int q = 0; q < 4; ++q) vh[grp * 4 + q][hs] = __builtin_convertvector(v[q], hv4);
      }
    }
#pragma unroll
    for (int ab = 0; ab < 16; ++ab) {
      const f32x4 uraw = *reinterpret_cast<const f32x4*>(S + bbase + ab * USTRIDE);
      const hv8 a8 = __builtin_shufflevector(vh[ab][0], vh[ab][1], 0, 1, 2, 3, 4, 5, 6, 7);
      const hv8 b8 = __builtin_bit_cast(hv8, uraw);
      if constexpr (MM == 1) acc[ab] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[ab], 0, 0, 0);
      else acc[ab] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[ab], 0, 0, 0);
    }
    if (cc < c_last) store_raw((c & 1) ^ 1, cc + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  wino_epilogue<THREADS, NWN>(p, acc, lds, tid, wm, wn, li, lh, n0);
}

// ---- launch: tile width, split over the chunks, instantiation
template <int MODE, int KD, int NWN, int NWM = 2>
int wino_launch_t(WinoArgs& a, long long blocks, hipStream_t stream) {
  constexpr int THREADS = 64 * NWM * NWN;
  const size_t lds_bytes = (2 * (size_t)(2 * NWM * WQ_PIX * WRAW_LD + 16 * 32 * NWN * 8 + 4 * THREADS) + (MODE == 0 ? 0 : (MODE >= 2 ? 4 * NWM : 2) * a.Cin + a.Cin)) * sizeof(float);
  g6d_allow_lds(reinterpret_cast<const void*>(&wino_conv3x3_kernel<MODE, KD, NWN, NWM>), 160 * 1024);
  hipLaunchKernelGGL((wino_conv3x3_kernel<MODE, KD, NWN, NWM>), dim3((unsigned)blocks, a.Cout / (32 * NWN), a.splits), dim3(THREADS), lds_bytes,
                     stream, a);
  return g6d_check_launch("wino_conv3x3");
}

template <int MODE, int KD>
int wino_launch_w(WinoArgs& a, long long blocks, int nwn, hipStream_t stream) {
  return nwn == 1 ? wino_launch_t<MODE, KD, 1>(a, blocks, stream) : wino_launch_t<MODE, KD, 2>(a, blocks, stream);
}

// Fills the geometry / split fields of `a` and launches.  kd = 1 or 3; mode as the kernel's MODE.
int wino_run(WinoArgs& a, int mode, int kd, float* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (a.nseg == 0) {          // one map size: the launch's own fields
    a.nseg = 1;
    a.seg[0] = WinoSeg{0, a.N, a.H, a.W, 0, 0, 0, 0, 0, a.ld_in, a.ld_full, a.ld_pool};
  }
  long long quarters = 0, in_extent = 0;
  double out_elems = 0.0;
  for (int k = 0; k < a.nseg; ++k) {
    WinoSeg& g = a.seg[k];
    const long long n_in = a.img_mod > 0 ? std::min<long long>(g.N, (long long)a.img_mod * a.D) : g.N;     // images actually read
    in_extent = std::max(in_extent, (long long)g.in_off + n_in * g.H * g.W * g.ld_in);
    g.QH = (g.H + 7) / 8; g.QW = (g.W + 7) / 8;
    g.qstart = (int)quarters;
    quarters += (long long)g.N * g.QH * g.QW;
    out_elems += (double)g.N * g.H * g.W;
  }
  a.QH = a.seg[0].QH; a.QW = a.seg[0].QW;
  // Block shape: four quarters x 64 (32) channels, or — trunk layers with Cout % 128 == 0 — two quarters x 128 channels: the same four
  // waves and accumulators, half the raw patch and twice the filter tile per chunk, and the layer's input is re-read Cout / 128 times
  const int we = (int)g6d_knob(G6D_KNOB_WINO_WIDE);         // (tests force both shapes): 0 = never, 2 = whenever eligible
  const bool wide_on = we != 0, wide_all = we == 2;
  // (measured per layer, tools/wino_split_probe.py and BATCH=8 tools/layer_table.py with G6D_WINO_WIDE=0/1: -1.5 ... -6 % where the input is
  // large against the filter bank — the detector's pyramid, the first layers of the crops' trunks; +1 ... 2 % where the doubled filter
  // stream per block dominates: 16x16 and 8x8 maps of a few crops)
  const bool wide = wide_on && !a.mm && mode == 0 && kd == 1 && (a.Cout & 127) == 0 && (wide_all || in_extent >= 6ll * 16 * a.Cin * a.Cout);
  const int nq = wide ? 2 : 4;
  const long long blocks = (quarters + nq - 1) / nq;
  if (blocks > 0x3fffffffll) { g6d_set_error("wino_conv3x3: grid too large"); return G6D_EINVAL; }
  a.qtotal = (int)quarters;
  if (in_extent * 4 >= (1ll << 31)) { g6d_set_error("wino_conv3x3: input tensor exceeds 2^31 bytes"); return G6D_EINVAL; }
  a.in_bytes = (unsigned)(in_extent * 4);
  // 32-channel (two-wave) blocks only for channel counts that are not multiples of 64: two of them share a CU, so they do not
  // spread a small grid over more CUs — the split over the chunks below does
  const int nwn = wide ? 4 : ((a.Cout & 63) ? 1 : 2);
  // Split of the (kd, chunk) list over gridDim.z.  One 64-channel block per CU is resident (512 registers per lane, ~100 KB
  // of LDS) and runs a serial loop of ~2.7 us per chunk, so a launch takes ceil(grid / 256) rounds of (chunks per block)
  // steps: small grids leave CUs idle and grids just above a multiple of 256 pay a nearly empty last round.  Pick the split
  // count with the smallest modelled time: rounds x block time + the hand-off (partial images written and read back, the
  // serial re-read of a tile's sp slabs of 64 KB by its last block).  The constants can be overridden for measurements.
  const int nchunks = a.mm ? kd * (a.Cin / 16) : kd * (a.Cin / 8);  // the 16-bit kernels' chunk is a pair of 8-channel chunks
  int splits = 1;
  const long long grid2 = blocks * (a.Cout / (32 * nwn));
  const int slots = 256 * (nwn == 1 ? 2 : 1);
  const int split_max = (int)g6d_knob(G6D_KNOB_WINO_SPLIT_MAX);
  const double m_gain = g6d_knob(G6D_KNOB_WINO_SPLIT_GAIN);
  const double m_fix = g6d_knob(G6D_KNOB_WINO_SPLIT_FIX);
  const double m_per = g6d_knob(G6D_KNOB_WINO_SPLIT_PER);
  const size_t room = workspace && workspace_bytes > G6D_WS_COUNTER_BYTES ? workspace_bytes - G6D_WS_COUNTER_BYTES : 0;
  const double tile_bytes = (double)grid2 * (wide ? 256 : 128 * nwn) * 64 * sizeof(float);     // one partial image, padded to whole tiles
  if (room > 0 && grid2 <= G6D_WS_COUNTERS && split_max > 1 && nchunks >= 4) {
    const double out_bytes = out_elems * a.Cout * sizeof(float);
    double best = 1e30;
    for (int sp = 1; sp <= split_max && sp <= nchunks / 2; ++sp) {
      if ((double)sp * tile_bytes > (double)room) break;
      const int cps_ = (nchunks + sp - 1) / sp, real = (nchunks + cps_ - 1) / cps_;
      if (real != sp) continue;
      const double rounds = (double)((grid2 * sp + slots - 1) / slots);
      double t = rounds * (cps_ * (a.mm ? 1.2 : 2.7) + 4.0);                    // us: chunks + prologue / epilogue of a block
      if (sp > 1) t += m_fix + m_per * sp + (2 * sp + 1) * out_bytes / 2.5e6;  // hand-off; partial traffic at 2.5 TB/s chip-wide
      if (t < best * m_gain) { best = t; splits = sp; }                        // a further split must buy >= 15 %
    }
  }
  const int cps = (nchunks + splits - 1) / splits;
  splits = (nchunks + cps - 1) / cps;
  a.splits = splits; a.chunks_per_split = cps; a.ws = workspace;
  const bool debug = g6d_knob(G6D_KNOB_WINO_DEBUG) == 1;
  if (debug) fprintf(stderr, "wino %d seg, N=%d %dx%dx%d->%d kd=%d mode=%d: grid %lld x %d splits of %d chunks\n", a.nseg, a.N, a.H, a.W, a.Cin,
                     a.Cout, kd, mode, grid2, splits, cps);
  if (a.mm) {                                   // 16-bit kernels: the trunk one (MODE 0, 2-D, buffer loads) or the conv-family one
    if (nwn != 2) { g6d_set_error("wino16: Cout % 64 == 0 expected"); return G6D_EINVAL; }
    const size_t stage16 = (size_t)(2 * WRAW_FLOATS + 16 * 64 * 8 + 4 * 256);
    const size_t lds16 = (2 * stage16 + (mode == 0 ? 0 : (mode >= 2 ? 8 : 2) * a.Cin)) * sizeof(float);
    if (lds16 > 160 * 1024) { g6d_set_error("wino16: affine tables do not fit LDS"); return G6D_EINVAL; }
    auto go_trunk = [&](auto V) {
      constexpr int MM = decltype(V)::value;
      g6d_allow_lds(reinterpret_cast<const void*>(&wino16_conv3x3_kernel<MM, 2>), 160 * 1024);
      hipLaunchKernelGGL((wino16_conv3x3_kernel<MM, 2>), dim3((unsigned)blocks, a.Cout / 64, a.splits), dim3(256), lds16, stream, a);
    };
    auto go_conv = [&](auto V, auto M, auto K) {
      constexpr int MM = decltype(V)::value, MODE = decltype(M)::value, KD = decltype(K)::value;
      g6d_allow_lds(reinterpret_cast<const void*>(&wino16_conv_kernel<MM, MODE, KD>), 160 * 1024);
      hipLaunchKernelGGL((wino16_conv_kernel<MM, MODE, KD>), dim3((unsigned)blocks, a.Cout / 64, a.splits), dim3(256), lds16, stream, a);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    auto by_mm = [&](auto M, auto K) { if (a.mm == 1) go_conv(I1{}, M, K); else go_conv(I2{}, M, K); };
    // knob wino16_2w (tests run both kernels): 0 = the one-wave-per-SIMD kernel only
    const bool two_waves = g6d_knob(G6D_KNOB_WINO16_2W) != 0 && mode == 0 && kd == 1 && !a.stats && a.splits == 1;
    auto go_trunk2 = [&](auto V) {
      constexpr int MM = decltype(V)::value;
      const size_t lds2 = (size_t)(2 * WRAW_FLOATS + 4 * 16 * 64 * 8 + 256) * sizeof(float);
      g6d_allow_lds(reinterpret_cast<const void*>(&wino16b_conv3x3_kernel<MM>), 160 * 1024);
      hipLaunchKernelGGL((wino16b_conv3x3_kernel<MM>), dim3((unsigned)blocks, a.Cout / 64, 1), dim3(512), lds2, stream, a);
    };
    if (two_waves) { if (a.mm == 1) go_trunk2(I1{}); else go_trunk2(I2{}); }
    else if (mode == 0 && kd == 1 && !a.stats) { if (a.mm == 1) go_trunk(I1{}); else go_trunk(I2{}); }
    else if (kd == 3) { if (mode == 0) by_mm(I0{}, I3{}); else if (mode == 1) by_mm(I1{}, I3{}); else if (mode == 2) by_mm(I2{}, I3{});
                        else { g6d_set_error("wino16: no multiplier prologue for 3x3x3"); return G6D_EINVAL; } }
    else { if (mode == 0) by_mm(I0{}, I1{}); else if (mode == 1) by_mm(I1{}, I1{}); else if (mode == 2) by_mm(I2{}, I1{}); else by_mm(I3{}, I1{}); }
    return g6d_check_launch("wino16_conv");
  }
  if (kd == 25) return wino_launch_w<0, 25>(a, blocks, nwn, stream);
  if (kd == 3) return mode == 2 ? wino_launch_w<2, 3>(a, blocks, nwn, stream)
                    : mode == 1 ? wino_launch_w<1, 3>(a, blocks, nwn, stream) : wino_launch_w<0, 3>(a, blocks, nwn, stream);
  if (mode == 3) return wino_launch_w<3, 1>(a, blocks, nwn, stream);
  if (mode == 2) return wino_launch_w<2, 1>(a, blocks, nwn, stream);
  if (mode == 1) return wino_launch_w<1, 1>(a, blocks, nwn, stream);
  if (wide) return wino_launch_t<0, 1, 4, 1>(a, blocks, stream);
  return wino_launch_w<0, 1>(a, blocks, nwn, stream);
}

}  // namespace

// in [N][H][W][ld_in] channels-last (Cin % 8 == 0), U = pre-transformed filters [Cin/8][16][Cout][8] (see
// gen6d_amd/network/backbone.py: winograd_filters), bias [Cout] or NULL, Cout % 64 == 0.
//   y = conv3x3(in, pad 1) + bias; if relu: y = max(y, 0)
//   out_full (optional) [N][H][W][ld_full]   <- y
//   out_pool (optional) [N][H/2][W/2][ld_pool] <- 2x2 max-pool of y (floor, as F.max_pool2d)
//   workspace (optional): split partial outputs for grids that would not fill the chip (splits*N*H*W*Cout floats)
// Replaces features[4..27] of vgg11_bn (conv + folded BatchNorm + ReLU + MaxPool), network/pretrain_models.py:17-25,66-72.
extern "C" int g6d_wino_conv3x3(const float* in, int N, int H, int W, int Cin, int ld_in, const float* U, const float* bias,
                                int Cout, int relu, float* out_full, int ld_full, float* out_pool, int ld_pool,
                                float* workspace, size_t workspace_bytes, g6d_stream_t stream) {
  if (!in || !U || (!out_full && !out_pool) || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 7) || (ld_in & 3) || ld_in < Cin ||
      Cout <= 0 || (Cout & 63) || (out_full && ld_full < Cout) || (out_pool && (ld_pool < Cout || H < 2 || W < 2)) ||
      !g6d_aligned16(in) || !g6d_aligned16(U) || (long long)N * H * W * ld_in >= (1ll << 29)) {
    g6d_set_error("wino_conv3x3: bad args (Cin % 8 == 0, Cout % 64 == 0, 16-byte aligned operands)"); return G6D_EINVAL;
  }
  WinoArgs a = {};
  a.in = in; a.U = U; a.bias = bias; a.out_full = out_full; a.out_pool = out_pool;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.ld_in = ld_in; a.Cout = Cout; a.ld_full = ld_full; a.ld_pool = ld_pool; a.relu = relu;
  a.D = 1;
  return wino_run(a, 0, 1, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

// The same layer over up to 4 map sizes in ONE launch (G6dWinoSeg, include/gen6d_hip.h): the scales of the detector's image
// pyramid (network/detector.py:236-241) run every trunk layer with the same filters.
extern "C" int g6d_wino_conv3x3_multi(const G6dWinoSeg* segs, int nseg, int Cin, const float* U, const float* bias, int Cout, int relu,
                                      float* workspace, size_t workspace_bytes, g6d_stream_t stream) {
  if (!segs || nseg < 1 || nseg > WINO_MAX_SEG || !U || Cin <= 0 || (Cin & 7) || Cout <= 0 || (Cout & 63) || !g6d_aligned16(U)) {
    g6d_set_error("wino_conv3x3_multi: bad args (1..4 segments, Cin % 8 == 0, Cout % 64 == 0)"); return G6D_EINVAL;
  }
  WinoArgs a = {};
  const bool want_full = segs[0].out_full != nullptr, want_pool = segs[0].out_pool != nullptr;
  if (!want_full && !want_pool) { g6d_set_error("wino_conv3x3_multi: no output"); return G6D_EINVAL; }
  // common base pointers: the lowest address of each kind; the kernel addresses with 32-bit float offsets from them
  const float* in0 = segs[0].in; float* f0 = segs[0].out_full; float* p0 = segs[0].out_pool;
  for (int k = 0; k < nseg; ++k) {
    const G6dWinoSeg& g = segs[k];
    if (!g.in || (g.out_full != nullptr) != want_full || (g.out_pool != nullptr) != want_pool || g.N <= 0 || g.H <= 0 || g.W <= 0 ||
        (g.ld_in & 3) || g.ld_in < Cin || (want_full && g.ld_full < Cout) || (want_pool && (g.ld_pool < Cout || g.H < 2 || g.W < 2)) ||
        !g6d_aligned16(g.in)) {
      g6d_set_error("wino_conv3x3_multi: bad segment (all segments give the same kinds of output)"); return G6D_EINVAL;
    }
    if (g.in < in0) in0 = g.in;
    if (want_full && g.out_full < f0) f0 = g.out_full;
    if (want_pool && g.out_pool < p0) p0 = g.out_pool;
  }
  a.in = in0; a.U = U; a.bias = bias; a.out_full = f0; a.out_pool = p0;
  a.Cin = Cin; a.Cout = Cout; a.relu = relu; a.D = 1;
  a.nseg = nseg;
  for (int k = 0; k < nseg; ++k) {
    const G6dWinoSeg& g = segs[k];
    const long long io = g.in - in0, fo = want_full ? g.out_full - f0 : 0, po = want_pool ? g.out_pool - p0 : 0;
    if (io + (long long)g.N * g.H * g.W * g.ld_in >= (1ll << 29) || fo + (long long)g.N * g.H * g.W * g.ld_full >= (1ll << 31) ||
        po + (long long)g.N * g.H * g.W * g.ld_pool >= (1ll << 31)) {
      g6d_set_error("wino_conv3x3_multi: segments must lie within 2^30 floats of each other (allocate them from one buffer)"); return G6D_EINVAL;
    }
    a.seg[k] = WinoSeg{0, g.N, g.H, g.W, 0, 0, (int)io, (int)fo, (int)po, g.ld_in, g.ld_full, g.ld_pool};
  }
  a.N = segs[0].N; a.H = segs[0].H; a.W = segs[0].W; a.ld_in = segs[0].ld_in; a.ld_full = segs[0].ld_full; a.ld_pool = segs[0].ld_pool;
  return wino_run(a, 0, 1, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

// Reduced-precision variant of g6d_wino_conv3x3_multi (math_mode 1 = bf16, 2 = fp16 operands, fp32 accumulation and outputs): U16 =
// the filters transformed AND rounded on the host, [Cin/16][16][Cout][16] 16-bit values in the layout of g6d_wino_conv3x3's U with a
// chunk of 16 input channels (rows with co & 8 carry their two 16-byte halves swapped); Cin % 16 == 0, Cout % 64 == 0.
extern "C" int g6d_wino16_conv3x3_multi(const G6dWinoSeg* segs, int nseg, int Cin, const void* U16, const float* bias, int Cout, int relu,
                                        int math_mode, float* workspace, size_t workspace_bytes, g6d_stream_t stream) {
  if (!segs || nseg < 1 || nseg > WINO_MAX_SEG || !U16 || Cin <= 0 || (Cin & 15) || Cout <= 0 || (Cout & 63) || !g6d_aligned16(U16) ||
      (math_mode != 1 && math_mode != 2)) {
    g6d_set_error("wino16_conv3x3_multi: bad args (1..4 segments, Cin % 16 == 0, Cout % 64 == 0, math_mode 1 or 2)"); return G6D_EINVAL;
  }
  WinoArgs a = {};
  const bool want_full = segs[0].out_full != nullptr, want_pool = segs[0].out_pool != nullptr;
  if (!want_full && !want_pool) { g6d_set_error("wino16_conv3x3_multi: no output"); return G6D_EINVAL; }
  const float* in0 = segs[0].in; float* f0 = segs[0].out_full; float* p0 = segs[0].out_pool;
  for (int k = 0; k < nseg; ++k) {
    const G6dWinoSeg& g = segs[k];
    if (!g.in || (g.out_full != nullptr) != want_full || (g.out_pool != nullptr) != want_pool || g.N <= 0 || g.H <= 0 || g.W <= 0 ||
        (g.ld_in & 3) || g.ld_in < Cin || (want_full && g.ld_full < Cout) || (want_pool && (g.ld_pool < Cout || g.H < 2 || g.W < 2)) ||
        !g6d_aligned16(g.in)) {
      g6d_set_error("wino16_conv3x3_multi: bad segment"); return G6D_EINVAL;
    }
    if (g.in < in0) in0 = g.in;
    if (want_full && g.out_full < f0) f0 = g.out_full;
    if (want_pool && g.out_pool < p0) p0 = g.out_pool;
  }
  a.in = in0; a.U = reinterpret_cast<const float*>(U16); a.bias = bias; a.out_full = f0; a.out_pool = p0;
  a.Cin = Cin; a.Cout = Cout; a.relu = relu; a.D = 1; a.nseg = nseg; a.mm = math_mode;
  for (int k = 0; k < nseg; ++k) {
    const G6dWinoSeg& g = segs[k];
    const long long io = g.in - in0, fo = want_full ? g.out_full - f0 : 0, po = want_pool ? g.out_pool - p0 : 0;
    if (io + (long long)g.N * g.H * g.W * g.ld_in >= (1ll << 29) || fo + (long long)g.N * g.H * g.W * g.ld_full >= (1ll << 31) ||
        po + (long long)g.N * g.H * g.W * g.ld_pool >= (1ll << 31)) {
      g6d_set_error("wino16_conv3x3_multi: segments must lie within 2^29 floats of each other (allocate them from one buffer)"); return G6D_EINVAL;
    }
    a.seg[k] = WinoSeg{0, g.N, g.H, g.W, 0, 0, (int)io, (int)fo, (int)po, g.ld_in, g.ld_full, g.ld_pool};
  }
  a.N = segs[0].N; a.H = segs[0].H; a.W = segs[0].W; a.ld_in = segs[0].ld_in; a.ld_full = segs[0].ld_full; a.ld_pool = segs[0].ld_pool;
  return wino_run(a, 0, 1, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

// The detector's 15x15 reference-as-filter correlation (network/detector.py:222-224) on the Winograd kernel: 5x5 blocks of 3x3
// sub-filters accumulated in the transform domain (KD = 25 above) — 2.25x fewer multiplications than the direct form of
// g6d_corr2d_patch.  Maps as in g6d_corr2d_patch_multi (up to 4 sizes, N maps each); U = the 25 sub-filter banks transformed like
// g6d_wino_conv3x3's, CHUNK-major: [Cin/8 * 25][16][Cout][8], row c * 25 + b = 8-channel chunk c of block b = 5 bi + bj, which holds
// w[:, 3bi..3bi+2, 3bj..3bj+2, 8c..8c+7] (include/gen6d_hip.h, backbone.winograd_corr_filters; the kernel walks kd = chunk % 25 innermost).
extern "C" int g6d_corr2d_wino_multi(const G6dCorrSeg* segs, int nseg, int Cin, const float* U, int Cout, int kblocks, float* workspace,
                                     size_t workspace_bytes, g6d_stream_t stream) {
  if (!segs || nseg < 1 || nseg > WINO_MAX_SEG || !U || kblocks != 5 || Cin <= 0 || (Cin & 7) || Cout <= 0 || (Cout & 31) || !g6d_aligned16(U)) {
    g6d_set_error("corr2d_wino_multi: bad args (1..4 map sizes, 15x15 = 5 blocks, Cin % 8 == 0, Cout % 32 == 0)"); return G6D_EINVAL;
  }
  WinoArgs a = {};
  const float* in0 = segs[0].in; float* f0 = segs[0].out;
  for (int k = 0; k < nseg; ++k) {
    const G6dCorrSeg& g = segs[k];
    if (!g.in || !g.out || g.N <= 0 || g.H <= 0 || g.W <= 0 || (g.ld_in & 3) || g.ld_in < Cin || g.ld_in != segs[0].ld_in || g.ld_out < Cout ||
        !g6d_aligned16(g.in)) {
      g6d_set_error("corr2d_wino_multi: bad map (all maps share ld_in)"); return G6D_EINVAL;
    }
    if (g.in < in0) in0 = g.in;
    if (g.out < f0) f0 = g.out;
  }
  a.in = in0; a.U = U; a.bias = nullptr; a.out_full = f0; a.out_pool = nullptr;
  a.Cin = Cin; a.Cout = Cout; a.relu = 0; a.D = 1; a.nseg = nseg;
  for (int k = 0; k < nseg; ++k) {
    const G6dCorrSeg& g = segs[k];
    const long long io = g.in - in0, fo = g.out - f0;
    if (io + (long long)g.N * g.H * g.W * g.ld_in >= (1ll << 29) || fo + (long long)g.N * g.H * g.W * g.ld_out >= (1ll << 31)) {
      g6d_set_error("corr2d_wino_multi: maps must lie within 2^29 floats of each other (allocate them from one buffer)"); return G6D_EINVAL;
    }
    a.seg[k] = WinoSeg{0, g.N, g.H, g.W, 0, 0, (int)io, (int)fo, 0, g.ld_in, g.ld_out, 0};
  }
  a.N = segs[0].N; a.H = segs[0].H; a.W = segs[0].W; a.ld_in = segs[0].ld_in; a.ld_full = segs[0].ld_out; a.ld_pool = 0;
  return wino_run(a, 0, kblocks * kblocks, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

// ---- the conv family on the Winograd kernel (called by g6d_conv_igemm when G6dConv.weight_wino is set)
// Eligible: kernel (1,3,3) on 2-D maps or (3,3,3), stride 1, "same" padding, Cin % 8 == 0, Cout % 32 == 0, maps of at least
// 6x6 (4x4 maps would fill a quarter of an 8x8 output quarter: the direct kernels are better there), fp32 math, no LeakyReLU,
// statistics groups = whole images or one group, no forced split.
bool g6d_wino_eligible(const G6dConv& d) {
  const bool on = g6d_knob(G6D_KNOB_CONV_WINO) != 0, on16 = g6d_knob(G6D_KNOB_CONV_WINO16) != 0;
  if (!on) return false;
  if (d.math_mode != 0) {     // 16-bit kernel: host-rounded 16-bit filters, chunks of 16 channels, 64-channel blocks
    if (!on16 || !d.weight_wino16 || (d.Cin & 15) || (d.Cout & 63) || !g6d_aligned16(d.weight_wino16)) return false;
  } else if (!d.weight_wino) return false;
  const bool k2 = d.kd == 1 && d.Di == 1 && d.pd == 0, k3 = d.kd == 3 && d.pd == 1;
  if (!(k2 || k3) || d.kh != 3 || d.kw != 3 || d.ph != 1 || d.pw != 1 || d.sd != 1 || d.sh != 1 || d.sw != 1) return false;
  if ((d.Cin & 7) || (d.Cout & 31) || d.Hi < 6 || d.Wi < 6 || d.out_act > 1 || d.split_k > 1) return false;
  if (d.mul && (!k2 || !d.in_scale)) return false;
  if ((d.in_image_mod > 0 || d.mul_group_images > 0) && !k2) return false;
  if (d.stats && d.stat_rows_per_group > 0 && d.stat_rows_per_group % (d.Do * d.Ho * d.Wo)) return false;   // groups = runs of whole images
  if (d.in_scale && d.Cin > 1024) return false;                          // affine tables of the block's four quarters in LDS
  if ((d.math_mode == 0 && !g6d_aligned16(d.weight_wino)) || (d.in_scale && ((d.Cin & 3) != 0))) return false;
  if (d.math_mode != 0 && d.mul && d.kd != 1) return false;
  // Profitability (measured per layer, profiles/r02_layer_table.md): a block pays ~4 us of prologue / output transform and the
  // kernel holds a whole SIMD per wave, so layers with a short reduction (K = kd*Cin < 128: 64-channel inputs) or little total work
  // (M * K * Cout < 1.5e8: the 7-image 8x8 / 16x16 feature-net layers, the 8^3 volume layer) stay on the direct kernels.
  // (knob wino_min_work: tests send small shapes down this path with 0, which disables the rule; -1 = the built-in thresholds.)
  const double mw = g6d_knob(G6D_KNOB_WINO_MIN_WORK);
  // The 16-bit kernel competes with direct kernels that already run 16-bit MFMAs at twice the fp32 rate while its own transforms
  // stay fp32 work: measured per layer (profiles/r03_layer_table_fp16.md) it wins from K >= 192 and ~9e8 multiply-adds upwards.
  const double min_work = mw >= 0 ? mw : (d.math_mode ? 9e8 : 1.5e8);
  const double M = (double)d.N * d.Di * d.Hi * d.Wi, K = (double)d.kd * d.Cin;
  if (min_work > 0 && (K < (d.math_mode ? 192 : 128) || M * K * d.Cout < min_work)) return false;
  if ((long long)d.N * d.Di * ((d.Hi + 7) / 8) * ((d.Wi + 7) / 8) >= (1ll << 31)) return false;           // quarter list
  return (long long)(d.in_image_mod > 0 ? d.in_image_mod : d.N) * d.Di * d.Hi * d.Wi * d.ld_in < (1ll << 29);      // 2^31 bytes: the bound of the buffer loads
}

int g6d_wino_launch(const G6dConv& d, hipStream_t stream) {
  WinoArgs a = {};
  a.in = d.in; a.U = d.math_mode ? reinterpret_cast<const float*>(d.weight_wino16) : d.weight_wino; a.mm = d.math_mode;
  a.bias = d.bias; a.out_full = d.out; a.out_pool = nullptr;
  a.D = d.Di; a.N = d.N * d.Di; a.H = d.Hi; a.W = d.Wi; a.Cin = d.Cin; a.ld_in = d.ld_in; a.Cout = d.Cout; a.ld_full = d.ld_out;
  a.ld_pool = 0; a.relu = d.out_act == 1;
  a.mul = d.mul; a.in_scale = d.in_scale; a.in_shift = d.in_shift; a.in_relu = d.in_relu;
  a.mul_bytes = d.mul ? (unsigned)((long long)(d.mul_group_images > 0 ? (d.N + d.mul_group_images - 1) / d.mul_group_images : 1) * d.Hi * d.Wi * d.Cin * 4) : 0u;      // (< 2^31: g6d_conv_igemm)
  a.stats = d.stats; a.stats_div = d.stat_rows_per_group > 0 ? d.stat_rows_per_group / (d.Do * d.Ho * d.Wo) : 0;
  a.aff_div = d.in_affine_per_n; a.img_mod = d.in_image_mod; a.mul_div = d.mul_group_images > 0 ? d.mul_group_images * d.Di : 0;
  if (d.fin_scale)
    a.fin = G6dFin{d.fin_scale, d.fin_shift, reinterpret_cast<int*>(d.fin_counter), d.stats, 1.0 / d.fin_count, d.fin_eps, d.fin_groups * d.Cout};
  const int mode = d.mul ? 3 : (!d.in_scale ? 0 : (d.in_affine_per_n ? 2 : 1));
  return wino_run(a, mode, d.kd, d.workspace, d.workspace_bytes, stream);
}
