// conv16_direct.hip — direct (implicit-GEMM) 3x3 / 3x3x3 convolution on 16-bit ACTIVATIONS: the reduced-precision mode's own kernel
// family (round 6; BASELINE configs[2] "bf16" / configs[4] "fp16 MFMA convs").
//
// Replaces, in the reduced-precision mode only, the VGG trunks' 3x3 layers (reference network/pretrain_models.py:9-31,66-72; detector
// pyramid detector.py:188-197,236-241; refiner crops refiner.py:64-78) and the first convs of the refiner's 32^3 volume net
// (refiner.py:88-143).  Until round 5 that mode kept fp32 activations in HBM and LDS and rounded operands when fragments left LDS:
// its 16-bit Winograd kernels were bound by LDS WRITES of fp32 data.  Here activations are fp16 / bf16 NHWC in HBM:
//   * both operand tiles of a K step go global -> LDS by DMA (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write, no
//     conversion), 16 bytes = 8 channels per lane; zero padding and ragged tile edges are the buffer's out-of-range zero fill;
//   * K step = 64 input channels of ONE tap (128 B per pixel row and per filter row): the LDS image [row][8 slots of 16 B] is lane-linear
//     as the DMA demands, and bank-conflict free for the ds_read_b128 fragment reads because the lane that FILLS slot s of row r fetches
//     channel group s ^ ((r >> 1) & 7) — the swizzle sits on the source address (cdna_hip_programming.md rule 21);
//   * v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulation; block = 128 output pixels x 128 output channels, 4 waves of 64 x 64 (2 x 2
//     accumulator tiles), two LDS stages of 32 KB, two blocks per CU (the second block's MFMAs cover this block's barrier / DMA waits);
//   * a tile is TH x TW pixels (TH TW = 128, TW a power of two chosen per map width) of the "tall image" [N D H rows][W columns], so 2x2
//     max-pool windows never straddle tiles and small maps pack several images into one tile;
//   * epilogue through LDS: bias, ReLU, then 16-byte stores of the full map (16-bit or fp32) and / or the 2x2 max-pooled map, and
//     optionally per-(image group, channel) sum / sum of squares in fp64 for the InstanceNorm that follows (volume net).
// K order: input-channel slice outermost, taps innermost — the nine shifted reads of a 16 KB slice hit L1 / L2.
#include "g6d_common.h"

namespace {

constexpr int C16_BM = 128, C16_BN = 128, C16_BK = 64;
constexpr int C16_STAGE = (C16_BM + C16_BN) * C16_BK * 2;      // 32 KB
constexpr int C16_LDS = 2 * C16_STAGE;                         // 64 KB
constexpr int C16_EP_LD = 68;                                  // floats per pixel row of the epilogue tile (64 channels + 4: conflict-free b128 reads)
constexpr unsigned C16_OOB = 0x80000000u;

struct C16Seg {
  const char* in; char* full; char* pool;
  int H, W, DH;            // map height / width, rows per image group (D * H)
  int rows;                // N * D * H rows of the tall image
  int ld_in, ld_full, ld_pool;
  int tw_log2, tiles_x, tile0;
  unsigned in_bytes;       // extent of the input buffer as the descriptor sees it (incl. the back-shift)
  int back;                // bytes the descriptor base lies BEFORE the tensor (offset of tap (0,0,0) from the centre, negated)
  // halo-patch kernel (conv16h_kernel): its own tiling of the same segment
  int h_tw_log2, h_tiles_x, h_tile0, h_tpi;      // tile width, tiles per row, first tile, tiles per image (0: tiles of whole small images)
  int h_segh, h_bands;                            // rows per band (min(H, TH)), bands per tile (TH / h_segh: > 1 when a tile holds several images)
  int h_swa, h_swd;                               // LDS slot swizzle: ((pcol >> h_swa) + prow * h_swd) & (slots - 1)
};
struct C16Params {
  C16Seg seg[4];
  int nseg, Cin, Cout, kd, D, relu, full_type, pool_type;   // *_type: 0 none, 1 = 16-bit (the operand type), 2 = fp32
  const char* w; unsigned w_bytes; const float* bias;
  int ptiles, nN;
  double* stats; int stat_rows_per_group;                    // optional: [groups][Cout][2]
  float acc_scale;                                           // y = acc * acc_scale + bias (the filters may carry a power-of-two scale)
  int ablate;                                                // knob c16_ablate (timing experiments)
};

template <int MM> struct C16T;
template <> struct C16T<1> { typedef __bf16 T; typedef bf16x8 V; };
template <> struct C16T<2> { typedef _Float16 T; typedef f16x8 V; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t c16_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// Epilogue shared by both kernels: two passes of 64 channels through an fp32 LDS tile [128 px][68]; bias, ReLU, then 16-byte stores.
// Output element types (full_type / pool_type): 1 = the 16-bit type T, 2 = fp32, 3 = fp16 hi / lo PAIR [pixel][2][Cout] (hi = rn16(v),
// lo = rn16(v - hi): the input format of the MM = 3 kernel; ld counts 16-bit elements and holds both planes).
template <int MM, typename WriteTile>
__device__ __forceinline__ void c16_epilogue(const C16Params& p, const C16Seg& sg, char* lds, int tid, int nt, int g0, int x0, int tw_log2, int ylim,
                                             WriteTile write_tile) {
  // write_tile(h, ep, cbase): the waves that own channels [cbase, cbase + 64) of the block store acc * acc_scale + bias (+ ReLU) of the
  // tile's 128 pixels into ep[pixel * C16_EP_LD + channel - cbase];  ylim: tile rows below it lie inside the image
  typedef typename C16T<MM>::T T;
  typedef typename C16T<MM>::V V8;
  const int TW = 1 << tw_log2, W = sg.W;
  float* ep = reinterpret_cast<float*>(lds);
  double* red = reinterpret_cast<double*>(lds + C16_BM * C16_EP_LD * 4);       // [4 quarters][64 ch][2] statistics partials
  const int TH2 = (C16_BM >> tw_log2) >> 1, TW2 = TW >> 1;
  auto inside = [&](int px, int& g, int& x) {
    const int py = px >> tw_log2;
    g = g0 + py; x = x0 + (px & (TW - 1));
    return py < ylim && g < sg.rows && x < W;
  };
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    const int cbase = nt * C16_BN + 64 * h;
    write_tile(h, ep, cbase);
    __syncthreads();
    if (p.full_type == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int px = (tid >> 3) + 32 * j, ch = (tid & 7) * 8;
        int g, x;
        if (inside(px, g, x)) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(ep + px * C16_EP_LD + ch), v1 = *reinterpret_cast<const f32x4*>(ep + px * C16_EP_LD + ch + 4);
          V8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] = (T)v0[e]; o[4 + e] = (T)v1[e]; }
          *reinterpret_cast<V8*>(sg.full + (((long)g * W + x) * sg.ld_full + cbase + ch) * 2) = o;
        }
      }
    } else if (p.full_type == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int px = (tid >> 3) + 32 * j, ch = (tid & 7) * 8;
        int g, x;
        if (inside(px, g, x)) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(ep + px * C16_EP_LD + ch), v1 = *reinterpret_cast<const f32x4*>(ep + px * C16_EP_LD + ch + 4);
          V8 hi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hi[e] = (T)v0[e]; lo[e] = (T)(v0[e] - (float)hi[e]);
            hi[4 + e] = (T)v1[e]; lo[4 + e] = (T)(v1[e] - (float)hi[4 + e]);
          }
          char* o = sg.full + (((long)g * W + x) * sg.ld_full + cbase + ch) * 2;
          *reinterpret_cast<V8*>(o) = hi;
          *reinterpret_cast<V8*>(o + p.Cout * 2) = lo;
        }
      }
    } else if (p.full_type == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int px = (tid >> 4) + 16 * j, ch = (tid & 15) * 4;
        int g, x;
        if (inside(px, g, x))
          *reinterpret_cast<f32x4*>(sg.full + (((long)g * W + x) * sg.ld_full + cbase + ch) * 4) = *reinterpret_cast<const f32x4*>(ep + px * C16_EP_LD + ch);
      }
    }
    if (p.pool_type) {
      const int per = p.pool_type == 2 ? 4 : 8, chunks = 64 / per;      // channels per item, items per pooled pixel
      for (int it = tid; it < 32 * chunks; it += 256) {
        const int pp = it / chunks, ch = (it - pp * chunks) * per;
        const int pry = pp / TW2, prx = pp - pry * TW2;
        const int r00 = ((2 * pry) << tw_log2) + 2 * prx;
        const int g = g0 + 2 * pry, x = x0 + 2 * prx;
        if (pry < TH2 && 2 * pry < ylim && g < sg.rows && x < W) {
          float m[8];
#pragma unroll
          for (int e = 0; e < 8; e += 4) {
            if (e < per) {
              const f32x4 q0 = *reinterpret_cast<const f32x4*>(ep + r00 * C16_EP_LD + ch + e), q1 = *reinterpret_cast<const f32x4*>(ep + (r00 + 1) * C16_EP_LD + ch + e);
              const f32x4 q2 = *reinterpret_cast<const f32x4*>(ep + (r00 + TW) * C16_EP_LD + ch + e), q3 = *reinterpret_cast<const f32x4*>(ep + (r00 + TW + 1) * C16_EP_LD + ch + e);
#pragma unroll
              for (int u = 0; u < 4; ++u) m[e + u] = fmaxf(fmaxf(q0[u], q1[u]), fmaxf(q2[u], q3[u]));
            }
          }
          const long o = ((long)(g >> 1) * (W >> 1) + (x >> 1)) * sg.ld_pool + cbase + ch;
          if (p.pool_type == 1) {
            V8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (T)m[e];
            *reinterpret_cast<V8*>(sg.pool + o * 2) = ov;
          } else if (p.pool_type == 3) {
            V8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) { hi[e] = (T)m[e]; lo[e] = (T)(m[e] - (float)hi[e]); }
            *reinterpret_cast<V8*>(sg.pool + o * 2) = hi;
            *reinterpret_cast<V8*>(sg.pool + (o + p.Cout) * 2) = lo;
          } else {
            f32x4 ov = {m[0], m[1], m[2], m[3]};
            *reinterpret_cast<f32x4*>(sg.pool + o * 4) = ov;
          }
        }
      }
    }
    if (p.stats) {
      // per-(group, channel) sum / sum of squares of the fp32 results of the tile's VALID pixels (a tile never straddles groups)
      const int c = tid & 63, q = tid >> 6;
      float s1 = 0.f, s2 = 0.f;
      for (int px = 32 * q; px < 32 * q + 32; ++px) {
        int g, x;
        if (inside(px, g, x)) { const float v = ep[px * C16_EP_LD + c]; s1 += v; s2 += v * v; }
      }
      red[(q * 64 + c) * 2] = (double)s1; red[(q * 64 + c) * 2 + 1] = (double)s2;
      __syncthreads();
      if (tid < 128) {
        const int cc = tid >> 1, w = tid & 1;
        const double v = red[(0 * 64 + cc) * 2 + w] + red[(1 * 64 + cc) * 2 + w] + red[(2 * 64 + cc) * 2 + w] + red[(3 * 64 + cc) * 2 + w];
        const int grp = p.stat_rows_per_group > 0 ? (int)(((long)g0 * W) / p.stat_rows_per_group) : 0;
        atomicAdd(p.stats + ((long)grp * p.Cout + cbase + cc) * 2 + w, v);
      }
    }
    __syncthreads();
  }
}

// The 2 x 2 wave layout of conv16_kernel / conv16r_kernel: wave (wm, wn) holds pixels 64 wm .. + 63 x channels 64 wn .. + 63.
__device__ __forceinline__ void c16_write22(const C16Params& p, const f32x16 (&acc)[2][2], int lane, int wv, int h, float* ep, int cbase) {
  const int wm = wv >> 1, wn = wv & 1;
  if (wn != h) return;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt2 = 0; nt2 < 2; ++nt2) {
      const float bv = p.bias ? p.bias[cbase + 32 * nt2 + (lane & 31)] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int px = 64 * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = fmaf(acc[mt][nt2][r], p.acc_scale, bv);
        if (p.relu) v = fmaxf(v, 0.f);
        ep[px * C16_EP_LD + 32 * nt2 + (lane & 31)] = v;
      }
    }
}

template <int MM>
__global__ __launch_bounds__(256, 2) void conv16_kernel(const C16Params p) {
  typedef typename C16T<MM>::V V8;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // block id -> (pixel tile, channel tile): the nN channel tiles of a pixel tile run back to back on ONE XCD (ids = xcd mod 8)
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int nt = jj % p.nN, ptile = (jj / p.nN) * 8 + xcd;
  if (ptile >= p.ptiles) return;
  int si = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if (i < p.nseg && ptile >= p.seg[i].tile0) si = i;
  const C16Seg& sg = p.seg[si];
  const int t = ptile - sg.tile0;
  const int tw_log2 = sg.tw_log2, TW = 1 << tw_log2;
  const int tx = t % sg.tiles_x, ty = t / sg.tiles_x;
  const int g0 = ty * (C16_BM >> tw_log2), x0 = tx * TW;
  const int H = sg.H, W = sg.W, D = p.D;
  const int ntaps = 9 * p.kd;

  // ---- geometry of the four A rows this lane fills per K step: byte offset of the (shifted-back) centre pixel + tap validity mask
  unsigned a_off[4], a_mask[4], b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 32 * wv + 8 * i + (lane >> 3);
    const int slot = (lane & 7) ^ ((r >> 1) & 7);
    const int g = g0 + (r >> tw_log2), x = x0 + (r & (TW - 1));
    unsigned m = 0;
    if (g < sg.rows && x < W) {
      const int y = g % H, z = (g / H) % D;
      unsigned my = (y > 0 ? 1u : 0u) | 2u | (y < H - 1 ? 4u : 0u);            // ky = 0,1,2 valid
      unsigned mx = (x > 0 ? 1u : 0u) | 2u | (x < W - 1 ? 4u : 0u);
      unsigned m9 = 0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) if (my >> ky & 1) m9 |= mx << (3 * ky);
      if (p.kd == 3) {
        if (z > 0) m |= m9;
        m |= m9 << 9;
        if (z < D - 1) m |= m9 << 18;
      } else m = m9;
    }
    a_mask[i] = m;
    a_off[i] = (unsigned)(((long)g * W + x) * sg.ld_in * 2) + slot * 16;
    b_off[i] = (unsigned)((long)(nt * C16_BN + r) * ntaps * p.Cin * 2) + slot * 16;
  }
  const __amdgpu_buffer_rsrc_t rs_in = c16_rsrc(sg.in - sg.back, sg.in_bytes);
  const __amdgpu_buffer_rsrc_t rs_w = c16_rsrc(p.w, p.w_bytes);
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int nchunk = p.Cin / C16_BK, nk = nchunk * ntaps;
  // K-step state (scalar): tap = (kz, ky, kx), channel chunk
  auto issue = [&](int k, int stage) {
    const int c = k / ntaps, tap = k - c * ntaps;
    const int kz = tap / 9, r9 = tap - 9 * kz, ky = r9 / 3, kx = r9 - 3 * ky;
    const unsigned a_k = (unsigned)(((kz * H + ky) * W + kx) * sg.ld_in * 2 + c * (C16_BK * 2));
    const unsigned b_k = (unsigned)((tap * p.Cin + c * C16_BK) * 2);
    char* sa = lds + stage * C16_STAGE + (32 * wv) * 128;
    char* sb = sa + C16_BM * 128;
    if (!((p.ablate & 1) && k > 1)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned vo = (a_mask[i] >> tap & 1u) ? a_off[i] + a_k : C16_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(sa + i * 1024), 16, vo, 0, 0, 0);
      }
    }
    if (!((p.ablate & 2) && k > 1)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(sb + i * 1024), 16, b_off[i] + b_k, 0, 0, 0);
    }
  };

  const int wm = wv >> 1, wn = wv & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // fragment read offsets: row = tile row (lane & 31), logical slot 2 ks + (lane >> 5) -> physical slot ^ ((row >> 1) & 7)
  const int frow = lane & 31, fhalf = lane >> 5, fsw = (frow >> 1) & 7;
  const int a_rd = (64 * wm + frow) * 128, b_rd = C16_BM * 128 + (64 * wn + frow) * 128;

  issue(0, 0);
  for (int k = 0; k < nk; ++k) {
    const int stage = k & 1;
    if (k + 1 < nk) {
      issue(k + 1, stage ^ 1);
      if (p.ablate) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const char* st = lds + stage * C16_STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int sl = ((2 * ks + fhalf) ^ fsw) * 16;
      V8 a0 = *reinterpret_cast<const V8*>(st + a_rd + sl);
      V8 a1 = *reinterpret_cast<const V8*>(st + a_rd + 32 * 128 + sl);
      V8 b0 = *reinterpret_cast<const V8*>(st + b_rd + sl);
      V8 b1 = *reinterpret_cast<const V8*>(st + b_rd + 32 * 128 + sl);
      if constexpr (MM == 1) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
      } else {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave has read this stage: the next iteration may refill it
  }

  c16_epilogue<MM>(p, sg, lds, tid, nt, g0, x0, tw_log2, C16_BM >> tw_log2,
                   [&](int h, float* ep, int cbase) { c16_write22(p, acc, lane, wv, h, ep, cbase); });
}


// ---------------------------------------------------------------------------------------------------------------------------------
// conv16r_kernel: the same tile with the FILTER operand kept out of LDS.  Measured on the kernel above: a wave spends as long issuing
// its 8 LDS-DMA pieces per K step (~100 cycles each beside MFMAs) as in its 16 MFMAs.  The filters are therefore packed on the host in
// FRAGMENT order — [channel tile][K step][ks][plane][32-channel group j][lane][8 values]: one coalesced 1 KB run per wave instruction,
// L2-resident — and go straight into registers one K step ahead (plain global loads: ~6 cycles of issue); only the activation tile
// takes the DMA path (4 pieces per wave and step), LDS holds 16 KB per stage, and the fragment loop reads LDS for A only.
//   MM = 1 / 2  bf16 / fp16 operands, K step = 64 channels of one tap
//   MM = 3      fp32-class arithmetic on the fp16 matrix cores: every operand is a PAIR of fp16 values (hi = rn16(x), lo = rn16(x - hi);
//               gfx950's MFMA keeps fp16 subnormals, tools/ubench/mfma_f16_denorm.hip), activations [pixel][2][C] (hi plane, lo plane),
//               acc += a_hi b_hi + a_hi b_lo + a_lo b_hi (the dropped lo x lo term is 2^-22 relative): 3 MFMAs of 32 cycles per 16 channels
//               against 8 x 64 cycles on the fp32 matrix cores.  K step = 32 channels (two planes of 64-byte rows: the same 16 KB stage).
template <int MM> struct C16R {
  static constexpr int BK = MM == 3 ? 32 : 64, NP = MM == 3 ? 2 : 1, KS = BK / 16, ROWB = BK * 2, SL = ROWB / 16;
  static constexpr int STAGE = NP * C16_BM * ROWB;            // 16 KB
  static constexpr int RPI = 1024 / ROWB;                     // rows one DMA wave-instruction fills (8 / 16)
};
template <int MM> struct C16T3 { typedef typename C16T<MM == 3 ? 2 : MM>::T T; typedef typename C16T<MM == 3 ? 2 : MM>::V V; };

template <int MM>
__device__ __forceinline__ f32x16 c16_mfma(typename C16T3<MM>::V a, typename C16T3<MM>::V b, f32x16 c) {
  if constexpr (MM == 1) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <int MM, int NSTAGE>
__global__ __launch_bounds__(256, 2) void conv16r_kernel(const C16Params p) {
  typedef typename C16T3<MM>::V V8;
  typedef C16R<MM> R;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int nt = jj % p.nN, ptile = (jj / p.nN) * 8 + xcd;
  if (ptile >= p.ptiles) return;
  int si = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if (i < p.nseg && ptile >= p.seg[i].tile0) si = i;
  const C16Seg& sg = p.seg[si];
  const int t = ptile - sg.tile0;
  const int tw_log2 = sg.tw_log2, TW = 1 << tw_log2;
  const int tx = t % sg.tiles_x, ty = t / sg.tiles_x;
  const int g0 = ty * (C16_BM >> tw_log2), x0 = tx * TW;
  const int H = sg.H, W = sg.W, D = p.D;
  const int ntaps = 9 * p.kd;

  // ---- A rows this lane fills: instruction i of wave wv covers plane i / (4 / NP), rows 32 wv + RPI (i % (4 / NP)) + lane / SL
  constexpr int NR = 4 / R::NP;                                // distinct rows per lane (4 / 2)
  unsigned a_off[NR], a_mask[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = 32 * wv + R::RPI * i + lane / R::SL;
    const int sw = R::SL == 8 ? (r >> 1) & 7 : (r >> 2) & 3;
    const int slot = (lane & (R::SL - 1)) ^ sw;
    const int g = g0 + (r >> tw_log2), x = x0 + (r & (TW - 1));
    unsigned m = 0;
    if (g < sg.rows && x < W) {
      const int y = g % H, z = (g / H) % D;
      unsigned my = (y > 0 ? 1u : 0u) | 2u | (y < H - 1 ? 4u : 0u);
      unsigned mx = (x > 0 ? 1u : 0u) | 2u | (x < W - 1 ? 4u : 0u);
      unsigned m9 = 0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) if (my >> ky & 1) m9 |= mx << (3 * ky);
      if (p.kd == 3) {
        if (z > 0) m |= m9;
        m |= m9 << 9;
        if (z < D - 1) m |= m9 << 18;
      } else m = m9;
    }
    a_mask[i] = m;
    a_off[i] = (unsigned)(((long)g * W + x) * sg.ld_in * 2) + slot * 16;
  }
  const __amdgpu_buffer_rsrc_t rs_in = c16_rsrc(sg.in - sg.back, sg.in_bytes);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int nchunk = p.Cin / R::BK, nk = nchunk * ntaps;
  const int plane_bytes = p.Cin * 2;                            // MM = 3: the lo plane of a pixel follows its hi plane

  auto issue_a = [&](int k, int stage) {
    const int c = k / ntaps, tap = k - c * ntaps;
    const int kz = tap / 9, r9 = tap - 9 * kz, ky = r9 / 3, kx = r9 - 3 * ky;
    const unsigned a_k = (unsigned)(((kz * H + ky) * W + kx) * sg.ld_in * 2 + c * (R::BK * 2));
    char* sa = lds + stage * R::STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ri = i % NR, pl = i / NR;
      const unsigned vo = (a_mask[ri] >> tap & 1u) ? a_off[ri] + a_k + pl * plane_bytes : C16_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(sa + pl * (C16_BM * R::ROWB) + (32 * wv + R::RPI * ri) * R::ROWB), 16, vo, 0, 0, 0);
    }
  };
  // ---- B fragments of K step k: [ks][plane][t] 16 bytes per lane, one coalesced 1 KB run per wave instruction.  Hand-issued loads:
  // beside LDS-DMA requests hipcc waits vmcnt(0) for any load it tracks (the whole pipeline drained every other step, ISA checked), so
  // these are invisible to its wait-count pass and retired by the counted s_waitcnt at the end of the step that requested them.
  const int wm = wv >> 1, wn = wv & 1;
  constexpr int NB = R::KS * R::NP * 2;                        // 8
  constexpr int STEP_B = R::KS * R::NP * 4 * 1024;             // bytes of one K step of one channel tile
  const char* wtile = p.w + (long)nt * nk * STEP_B;             // (uniform)
  const unsigned b_voff = (2 * wn) * 1024 + lane * 16;
  auto load_b = [&](int k, V8 (&b)[NB]) {
    const char* ws = wtile + (long)k * STEP_B;
#pragma unroll
    for (int q = 0; q < R::KS * R::NP; ++q) {
      const char* wq = ws + q * 4096;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[2 * q]) : "v"(b_voff), "s"(wq) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(b[2 * q + 1]) : "v"(b_voff), "s"(wq) : "memory");
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int fsw = R::SL == 8 ? (frow >> 1) & 7 : (frow >> 2) & 3;
  const int a_rd = (64 * wm + frow) * R::ROWB;

  auto compute = [&](int stage, const V8 (&b)[NB]) {
    const char* st = lds + stage * R::STAGE;
#pragma unroll
    for (int ks = 0; ks < R::KS; ++ks) {
      const int sl = ((2 * ks + fhalf) ^ fsw) * 16;
      V8 a[R::NP][2];
#pragma unroll
      for (int pl = 0; pl < R::NP; ++pl) {
        a[pl][0] = *reinterpret_cast<const V8*>(st + pl * (C16_BM * R::ROWB) + a_rd + sl);
        a[pl][1] = *reinterpret_cast<const V8*>(st + pl * (C16_BM * R::ROWB) + a_rd + 32 * R::ROWB + sl);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
          acc[mt][n2] = c16_mfma<MM>(a[0][mt], b[(ks * R::NP) * 2 + n2], acc[mt][n2]);
          if constexpr (MM == 3) {
            acc[mt][n2] = c16_mfma<MM>(a[0][mt], b[(ks * R::NP + 1) * 2 + n2], acc[mt][n2]);      // hi x lo
            acc[mt][n2] = c16_mfma<MM>(a[1][mt], b[(ks * R::NP) * 2 + n2], acc[mt][n2]);          // lo x hi
          }
        }
    }
  };

  // ---- K loop, three LDS stages, ONE barrier per step.  Step k: request B(k+1) and the DMA of step k+2 (into the stage step k-1 read:
  // every wave passed the barrier that ended it), compute step k, then retire B(k+1) and DMA(k+1) with a counted wait that leaves only
  // this step's four DMA pieces in flight (requests return in order), and meet at the barrier — which both frees stage k % 3 and
  // publishes everybody's DMA(k+1).
  static_assert(NSTAGE == 3, "the stage arithmetic below assumes three stages");
  V8 b0[NB], b1[NB];
  load_b(0, b0);
  issue_a(0, 0);
  if (nk > 1) { issue_a(1, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  int stage = 0;
  auto step = [&](int k, const V8 (&bc)[NB], V8 (&bn)[NB]) {
    if (k + 1 < nk && !((p.ablate & 2) && k > 1)) load_b(k + 1, bn);
    const int s2 = stage >= 1 ? stage - 1 : 2;               // (stage + 2) % 3
    if (k + 2 < nk && !((p.ablate & 1) && k > 1)) issue_a(k + 2, s2);
    compute(stage, bc);
    if (k + 2 < nk && !p.ablate) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stage = stage == 2 ? 0 : stage + 1;
  };
#pragma unroll 1
  for (int k = 0; k < nk; k += 2) {
    step(k, b0, b1);
    if (k + 1 < nk) step(k + 1, b1, b0);
  }
  c16_epilogue<MM == 3 ? 2 : MM>(p, sg, lds, tid, nt, g0, x0, tw_log2, C16_BM >> tw_log2,
                                 [&](int h, float* ep, int cbase) { c16_write22(p, acc, lane, wv, h, ep, cbase); });
}


// ---------------------------------------------------------------------------------------------------------------------------------
// conv16h_kernel: the HALO-PATCH variant (2-D layers).  Measured on the two kernels above (tools/conv16_ablate.py): without the
// activation requests they run 15 % faster, without the filter requests 14 %, without both 38 % — a step's 32-48 KB of vector-memory
// traffic is what holds them at a third of the MFMA rate.  Here
//   * the activations of a 64-channel slice (pairs: 32) are staged ONCE per slice as the tile's halo patch — (TH + 2) x (TW + 2) pixels
//     (tiles of several small images: one band of H + 2 rows per image, so that zero padding between images stays zero) — and all nine
//     taps read shifted windows of it: 2.9 KB of activation traffic per tap instead of 16 KB, and ONE barrier per slice instead of two
//     per tap (between them the waves run free: the patch is read-only and the filters are private);
//   * the waves split the block's 128 output channels (wave w: all 128 pixels x channels 32 w .. 32 w + 31), so that every filter
//     fragment is requested by exactly one wave — fragment-major filters straight into registers, one tap ahead: 16 KB per tap and block;
//   * the patch rows are lane-linear for the DMA and swizzled by (patch row, patch column) so that the shifted ds_read_b128 fragment
//     reads of every tap are bank-conflict free (exhaustive search, tools/ubench/conv16_swizzle.py).
template <int MM>
__global__ __launch_bounds__(256, 2) void conv16h_kernel(const C16Params p) {
  typedef typename C16T3<MM>::V V8;
  typedef C16R<MM> R;
  constexpr int PROWS = 288, PATCHB = PROWS * R::ROWB, STAGE = R::NP * PATCHB;      // 36,864 B per stage in both modes
  constexpr int NPI = 9;                                       // DMA wave-instructions per wave and slice at most (288 rows)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int nt = jj % p.nN, ptile = (jj / p.nN) * 8 + xcd;
  if (ptile >= p.ptiles) return;
  int si = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if (i < p.nseg && ptile >= p.seg[i].h_tile0) si = i;
  const C16Seg& sg = p.seg[si];
  const int t = ptile - sg.h_tile0;
  const int tw_log2 = sg.h_tw_log2, TW = 1 << tw_log2, TH = C16_BM >> tw_log2, PW = TW + 2;
  const int H = sg.H, W = sg.W;
  int g0, x0, y0, ylim;
  if (sg.h_tpi > 0) {                                          // tiles inside one image
    const int n = t / sg.h_tpi, r = t - n * sg.h_tpi;
    const int ty = r / sg.h_tiles_x, tx = r - ty * sg.h_tiles_x;
    y0 = ty * TH; x0 = tx * TW; g0 = n * H + y0; ylim = min(TH, H - y0);
  } else {                                                     // tiles of TH / H whole images
    const int ty = t / sg.h_tiles_x, tx = t - ty * sg.h_tiles_x;
    y0 = 0; x0 = tx * TW; g0 = ty * TH; ylim = TH;
  }
  const int segh = sg.h_segh, bandr = segh + 2, P = sg.h_bands * bandr * PW;      // patch rows
  const int swa = sg.h_swa, swd = sg.h_swd;
  const int NI = (P + R::RPI - 1) / R::RPI;                    // DMA wave-instructions per plane

  // ---- patch rows this lane fills: instruction idx = wv + 4 i -> plane idx / NI, rows RPI (idx % NI) + lane / SL
  unsigned poff[NPI];
#pragma unroll
  for (int i = 0; i < NPI; ++i) {
    const int idx = wv + 4 * i, pl = idx / NI, ii = idx - pl * NI;
    const int q = ii * R::RPI + lane / R::SL;
    const int prow = q / PW, pcol = q - prow * PW;
    const int b = prow / bandr, lr = prow - b * bandr - 1;     // band, row inside the band's image rows (-1 .. segh)
    const int g = g0 + b * segh + lr, x = x0 + pcol - 1;
    const int yimg = y0 + lr;                                  // (whole-image tiles: y0 = 0, lr = the image row)
    const bool ok = pl < R::NP && q < P && yimg >= 0 && yimg < H && g < sg.rows && x >= 0 && x < W;
    const int slot = (lane & (R::SL - 1)) ^ (((pcol >> swa) + prow * swd) & (R::SL - 1));
    poff[i] = ok ? (unsigned)((((long)g * W + x) * sg.ld_in + pl * p.Cin) * 2 + slot * 16) : C16_OOB;
  }
  const __amdgpu_buffer_rsrc_t rs_in = c16_rsrc(sg.in, sg.in_bytes - sg.back);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int nchunk = p.Cin / R::BK, nk = nchunk * 9;
  auto issue_patch = [&](int c, int stage) {
    char* sa = lds + stage * STAGE;
    const unsigned ck = (unsigned)(c * (R::BK * 2));
    // ALWAYS NPI requests per wave (the counted wait of the step below depends on it): the surplus ones ask for an out-of-range offset
    // and write zeros into the stage's last KB, which is unused whenever there is a surplus
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int idx = wv + 4 * i;
      const bool live = idx < R::NP * NI;
      const int pl = live ? idx / NI : 0, ii = live ? idx - pl * NI : 0;
      const unsigned vo = (!live || poff[i] == C16_OOB) ? C16_OOB : poff[i] + ck;
      char* dst = live ? sa + pl * PATCHB + ii * 1024 : sa + STAGE - 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)dst, 16, vo, 0, 0, 0);
    }
  };
  // ---- filter fragments of step s = 9 c + tap: [ks][plane] 16 bytes per lane of THIS wave's 32 channels, hand-issued (see conv16r_kernel)
  constexpr int NB = R::KS * R::NP;                            // 4
  constexpr int STEP_B = R::KS * R::NP * 4 * 1024;
  const char* wtile = p.w + (long)nt * nk * STEP_B;
  const unsigned b_voff = wv * 1024 + lane * 16;
  auto load_b = [&](int s_, V8 (&b)[NB]) {
    const char* ws = wtile + (long)s_ * STEP_B;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const char* wq = ws + q * 4096;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[q]) : "v"(b_voff), "s"(wq) : "memory");
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  // ---- fragment geometry: m-tile mt = tile pixels 32 mt + (lane & 31) -> patch (row, column) of tap (0, 0)
  const int fhalf = lane >> 5;
  int fq[4], frow[4], fcol[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int r = 32 * mt + (lane & 31), py = r >> tw_log2, px = r & (TW - 1);
    const int b = py / segh, ly = py - b * segh;
    frow[mt] = b * bandr + ly; fcol[mt] = px; fq[mt] = frow[mt] * PW + px;
  }
  // Fragment reads run ONE MFMA GROUP AHEAD in a second register set (first version: the compiler kept two fragment registers and
  // every MFMA waited out the LDS latency of a read issued one instruction earlier — the ISA showed ds_read / s_waitcnt / v_mfma triples):
  // while the MFMAs of (tap, ks) issue, the fragments of (tap, ks + 1) — or of the next tap's ks = 0 inside the same slice — are in flight.
  struct Frag { V8 a[R::NP][4]; };
  auto frag_addr = [&](int ky, int kx, int (&abase)[4], int (&asw)[4]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      abase[mt] = (fq[mt] + ky * PW + kx) * R::ROWB;
      asw[mt] = ((fcol[mt] + kx) >> swa) + (frow[mt] + ky) * swd;
    }
  };
  auto frag_read = [&](Frag& f, int stage, const int (&abase)[4], const int (&asw)[4], int ks) {
    const char* st = lds + stage * STAGE;
#pragma unroll
    for (int pl = 0; pl < R::NP; ++pl)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        f.a[pl][mt] = *reinterpret_cast<const V8*>(st + pl * PATCHB + abase[mt] + ((((2 * ks + fhalf) ^ asw[mt]) & (R::SL - 1)) << 4));
  };
  auto mfmas = [&](const Frag& f, const V8 (&b)[NB], int ks) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      acc[mt] = c16_mfma<MM>(f.a[0][mt], b[ks * R::NP], acc[mt]);
      if constexpr (MM == 3) {
        acc[mt] = c16_mfma<MM>(f.a[0][mt], b[ks * R::NP + 1], acc[mt]);      // hi x lo
        acc[mt] = c16_mfma<MM>(f.a[1][mt], b[ks * R::NP], acc[mt]);          // lo x hi
      }
    }
  };

  // ---- K loop: slices outermost (one patch, one barrier each), the nine taps inside.  The filter fragments run TWO steps ahead in a
  // ring of three register sets (requests return in order, so before step s the wait allows what was requested after B(s): two filter
  // sets = 8 loads, plus the 9 patch pieces while they are younger than B(s)).
  V8 b0[NB], b1[NB], b2[NB];
  load_b(0, b0);
  if (nk > 1) load_b(1, b1);
  issue_patch(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  int stage = 0, tap = 0, ky = 0, kx = 0, c = 0;
  int abase[4], asw[4];
  Frag f0, f1;
  frag_addr(0, 0, abase, asw);
  frag_read(f0, 0, abase, asw, 0);
  // one step; `fc` holds the fragments of (this tap, ks = 0) on entry and of (next tap, ks = 0) on exit (KS even: the sets swap back)
  auto step = [&](int s_, const V8 (&bc)[NB], V8 (&bfar)[NB], Frag& fc, Frag& fn) {
    const bool more = s_ + 2 < nk;
    if (more) load_b(s_ + 2, bfar);
    if (tap == 0 && c + 1 < nchunk) issue_patch(c + 1, stage ^ 1);      // first tap of a slice: request the next slice's patch
    if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (tap < 3 && c + 1 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB + NPI) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    const bool last_tap = tap == 8;
    int nky = ky, nkx = kx + 1;
    if (nkx == 3) { nkx = 0; ++nky; }
    int nbase[4], nsw[4];
    frag_addr(last_tap ? 0 : nky, last_tap ? 0 : nkx, nbase, nsw);
#pragma unroll
    for (int ks = 0; ks < R::KS; ++ks) {
      Frag& cur = (ks & 1) ? fn : fc;
      Frag& nxt = (ks & 1) ? fc : fn;
      if (ks + 1 < R::KS) frag_read(nxt, stage, abase, asw, ks + 1);
      else if (!last_tap) frag_read(nxt, stage, nbase, nsw, 0);          // the next tap reads the same patch
      __builtin_amdgcn_sched_barrier(0);
      mfmas(cur, bc, ks);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { abase[mt] = nbase[mt]; asw[mt] = nsw[mt]; }
    ++tap; kx = nkx; ky = nky;
    if (last_tap) {
      tap = 0; ky = 0; kx = 0; ++c;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                            // every wave has read this patch and received its share of the next
      __builtin_amdgcn_sched_barrier(0);
      stage ^= 1;
      if (c < nchunk) frag_read(fc, stage, abase, asw, 0);     // (KS even: after the ks loop the "current" set is fc again)
    }
  };
  static_assert(R::KS % 2 == 0, "the fragment sets swap back after an even number of MFMA groups");
#pragma unroll 1
  for (int s_ = 0; s_ < nk; s_ += 3) {                        // (nk = 9 slices: a multiple of 3)
    step(s_, b0, b2, f0, f1);
    step(s_ + 1, b1, b0, f0, f1);
    step(s_ + 2, b2, b1, f0, f1);
  }
  // ---- epilogue: wave w holds channels 32 w .. 32 w + 31 of all 128 pixels
  c16_epilogue<MM == 3 ? 2 : MM>(p, sg, lds, tid, nt, g0, x0, tw_log2, ylim, [&](int h, float* ep, int cbase) {
    if ((wv >> 1) != h) return;
    const float bv = p.bias ? p.bias[cbase + 32 * (wv & 1) + (lane & 31)] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int px = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = fmaf(acc[mt][r], p.acc_scale, bv);
        if (p.relu) v = fmaxf(v, 0.f);
        ep[px * C16_EP_LD + 32 * (wv & 1) + (lane & 31)] = v;
      }
  });
}

// Tiling of one segment for the halo-patch kernel: the tile width (32 / 16 / 8 / 4) with the least overhang; false if none fits
// (a map lower than the tile must divide it: tiles of whole images)
bool c16_halo_tiling(const G6dConv16Seg& s, int rowb, C16Seg& o, int& tiles) {
  double best = 1e30;
  bool found = false;
  for (int tw = 32; tw >= 4; tw >>= 1) {
    const int th = C16_BM / tw, tx = (s.W + tw - 1) / tw;
    long nt; int tpi, segh, bands;
    if (s.H >= th) { tpi = tx * ((s.H + th - 1) / th); nt = (long)s.N * tpi; segh = th; bands = 1; }
    else {
      if (th % s.H) continue;
      tpi = 0; nt = (long)tx * (((long)s.N * s.H + th - 1) / th); segh = s.H; bands = th / s.H;
    }
    if (bands * (segh + 2) * (tw + 2) > 288) continue;
    const double waste = (double)nt * C16_BM / ((double)s.N * s.H * s.W);
    if (waste < best - 1e-9) {
      best = waste; found = true;
      int l2 = 0; while ((1 << l2) < tw) ++l2;
      o.h_tw_log2 = l2; o.h_tiles_x = tx; o.h_tpi = tpi; o.h_segh = segh; o.h_bands = bands;
      tiles = (int)nt;
      // conflict-free slot swizzles found by tools/ubench/conv16_swizzle.py: ((pcol >> a) + prow * d) & (slots - 1)
      if (rowb == 128) { o.h_swa = tw == 16 ? 0 : 1; o.h_swd = tw == 4 ? 2 : (tw == 8 ? 4 : (tw == 16 ? 1 : 0)); }
      else { o.h_swa = tw == 32 ? 2 : (tw == 16 ? 1 : 0); o.h_swd = tw <= 8 ? 1 : 0; }
    }
  }
  return found;
}

int c16_pick_tw(int W, int pool) {
  // tile width (power of two, 4..64; even for pooling) with the least column overhang, ties -> the wider tile
  int best = 8; double bw = 1e9;
  for (int tw = 64; tw >= (pool ? 4 : 4); tw >>= 1) {
    const double waste = (double)((W + tw - 1) / tw * tw) / W;
    if (waste < bw - 1e-9) { bw = waste; best = tw; }
  }
  return best;
}

}  // namespace

extern "C" int g6d_conv16_direct_multi(const G6dConv16Seg* segs, int nseg, int Cin, const void* W16, int w_layout, float acc_scale,
                                       const float* bias, int Cout, int kd, int relu, int full_type, int pool_type, int math_mode, double* stats,
                                       int stat_rows_per_group, g6d_stream_t stream) {
  if (!segs || nseg < 1 || nseg > 4 || !W16) { g6d_set_error("conv16_direct: 1..4 segments and filters expected"); return G6D_EINVAL; }
  if (math_mode < 1 || math_mode > 3 || (w_layout != 0 && w_layout != 1) || (math_mode == 3 && w_layout != 1)) {
    g6d_set_error("conv16_direct: math_mode 1 (bf16) / 2 (fp16) / 3 (fp16 hi-lo pairs, fragment-major filters only)"); return G6D_EINVAL;
  }
  const int bk = math_mode == 3 ? 32 : C16_BK, planes = math_mode == 3 ? 2 : 1;
  if (Cin % bk || Cout % C16_BN || (kd != 1 && kd != 3)) { g6d_set_error("conv16_direct: Cin % 64 (32 for pairs), Cout % 128, kd in {1,3} expected"); return G6D_EINVAL; }
  const int t16 = math_mode == 3 ? 3 : 1;                    // the 16-bit output coding of this mode
  auto type_ok = [&](int t) { return t == 0 || t == 2 || t == t16; };
  if (!type_ok(full_type) || !type_ok(pool_type) || (!full_type && !pool_type && !stats)) { g6d_set_error("conv16_direct: output types"); return G6D_EINVAL; }
  if (pool_type && kd != 1) { g6d_set_error("conv16_direct: pooling is 2-D only"); return G6D_EINVAL; }
  C16Params p = {};
  p.nseg = nseg; p.Cin = Cin; p.Cout = Cout; p.kd = kd; p.relu = relu; p.full_type = full_type; p.pool_type = pool_type;
  p.w = static_cast<const char*>(W16); p.bias = bias; p.stats = stats; p.stat_rows_per_group = stat_rows_per_group;
  p.acc_scale = acc_scale != 0.f ? acc_scale : 1.f;
  p.ablate = (int)g6d_knob(G6D_KNOB_C16_ABLATE);
  const long wb = (long)Cout * 9 * kd * Cin * 2 * planes;
  if (wb >= (1L << 31)) { g6d_set_error("conv16_direct: filters beyond 2 GB"); return G6D_EINVAL; }
  p.w_bytes = (unsigned)wb;
  p.nN = Cout / C16_BN;
  int tiles = 0, D0 = segs[0].D;
  for (int i = 0; i < nseg; ++i) {
    const G6dConv16Seg& s = segs[i];
    C16Seg& o = p.seg[i];
    if (!s.in || s.N < 1 || s.D < 1 || s.H < 1 || s.W < 1 || s.ld_in < planes * Cin || s.D != D0 || (kd == 1 && s.D != 1)) { g6d_set_error("conv16_direct: bad segment"); return G6D_EINVAL; }
    const int need_full = full_type == 3 ? 2 * Cout : Cout, need_pool = pool_type == 3 ? 2 * Cout : Cout;
    if ((full_type && (!s.out_full || s.ld_full < need_full)) || (pool_type && (!s.out_pool || s.ld_pool < need_pool || (s.H & 1) || (s.W & 1)))) {
      g6d_set_error("conv16_direct: outputs missing / odd map with pooling"); return G6D_EINVAL;
    }
    if (!g6d_aligned16(s.in) || (s.ld_in & 7) || (full_type && (!g6d_aligned16(s.out_full) || (s.ld_full & 7))) || (pool_type && (!g6d_aligned16(s.out_pool) || (s.ld_pool & 7)))) {
      g6d_set_error("conv16_direct: 16-byte aligned rows expected"); return G6D_EINVAL;
    }
    o.in = static_cast<const char*>(s.in); o.full = static_cast<char*>(s.out_full); o.pool = static_cast<char*>(s.out_pool);
    o.H = s.H; o.W = s.W; o.DH = s.D * s.H; o.rows = s.N * s.D * s.H;
    o.ld_in = s.ld_in; o.ld_full = s.ld_full; o.ld_pool = s.ld_pool;
    const int tw = c16_pick_tw(s.W, pool_type != 0);
    int l2 = 0; while ((1 << l2) < tw) ++l2;
    o.tw_log2 = l2; o.tiles_x = (s.W + tw - 1) / tw; o.tile0 = tiles;
    const int th = C16_BM / tw;
    if (stats && stat_rows_per_group > 0 && ((long)stat_rows_per_group % ((long)th * s.W) != 0)) { g6d_set_error("conv16_direct: statistics groups must be whole tile rows"); return G6D_EINVAL; }
    tiles += o.tiles_x * ((o.rows + th - 1) / th);
    // the descriptor starts (kd == 3 ? H W : 0) + W + 1 pixels before the tensor: tap offsets are then non-negative
    const long back = ((long)(kd == 3 ? s.H * s.W : 0) + s.W + 1) * s.ld_in * 2;
    const long ext = (long)o.rows * s.W * s.ld_in * 2 + back;
    if (ext >= (1L << 31)) { g6d_set_error("conv16_direct: a segment's input beyond 2 GB"); return G6D_EINVAL; }
    o.back = (int)back; o.in_bytes = (unsigned)ext;
  }
  p.D = D0;
  p.ptiles = tiles;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // fragment-major filters, 2-D layer: the halo-patch kernel when every segment has a tiling for it (knob conv16_halo = 0: never)
  if (w_layout == 1 && kd == 1 && g6d_knob(G6D_KNOB_CONV16_HALO) != 0) {
    bool ok = true;
    int htiles = 0;
    for (int i = 0; i < nseg && ok; ++i) {
      int nt_ = 0;
      ok = c16_halo_tiling(segs[i], math_mode == 3 ? 64 : 128, p.seg[i], nt_);
      p.seg[i].h_tile0 = htiles; htiles += nt_;
      if (stats && stat_rows_per_group > 0 && ok) {
        const C16Seg& o = p.seg[i];
        const int th = C16_BM >> o.h_tw_log2;
        // a tile must lie inside one statistics group: in-image tiles do when groups are whole images; tiles of whole images when the group is too
        ok = (stat_rows_per_group % (segs[i].H * segs[i].W) == 0) && (o.h_tpi > 0 || ((long)stat_rows_per_group % ((long)th * segs[i].W) == 0));
      }
    }
    if (ok) {
      p.ptiles = htiles;
      const int hblocks = (htiles + 7) / 8 * 8 * p.nN;
      constexpr int LDSH = 2 * 36864;
      if (math_mode == 1) {
        g6d_allow_lds(reinterpret_cast<const void*>(&conv16h_kernel<1>), LDSH);
        hipLaunchKernelGGL(conv16h_kernel<1>, dim3(hblocks), dim3(256), LDSH, st, p);
      } else if (math_mode == 2) {
        g6d_allow_lds(reinterpret_cast<const void*>(&conv16h_kernel<2>), LDSH);
        hipLaunchKernelGGL(conv16h_kernel<2>, dim3(hblocks), dim3(256), LDSH, st, p);
      } else {
        g6d_allow_lds(reinterpret_cast<const void*>(&conv16h_kernel<3>), LDSH);
        hipLaunchKernelGGL(conv16h_kernel<3>, dim3(hblocks), dim3(256), LDSH, st, p);
      }
      return g6d_check_launch("conv16h_direct");
    }
  }
  const int blocks = (tiles + 7) / 8 * 8 * p.nN;
  if (w_layout == 0) {
    if (math_mode == 1) {
      g6d_allow_lds(reinterpret_cast<const void*>(&conv16_kernel<1>), C16_LDS);
      hipLaunchKernelGGL(conv16_kernel<1>, dim3(blocks), dim3(256), C16_LDS, st, p);
    } else {
      g6d_allow_lds(reinterpret_cast<const void*>(&conv16_kernel<2>), C16_LDS);
      hipLaunchKernelGGL(conv16_kernel<2>, dim3(blocks), dim3(256), C16_LDS, st, p);
    }
    return g6d_check_launch("conv16_direct");
  }
  constexpr int NST = 3, LDSR = NST * 16384;                   // three 16 KB stages (>= the epilogue's 38.9 KB tile)
  if (math_mode == 1) {
    g6d_allow_lds(reinterpret_cast<const void*>(&conv16r_kernel<1, NST>), LDSR);
    hipLaunchKernelGGL((conv16r_kernel<1, NST>), dim3(blocks), dim3(256), LDSR, st, p);
  } else if (math_mode == 2) {
    g6d_allow_lds(reinterpret_cast<const void*>(&conv16r_kernel<2, NST>), LDSR);
    hipLaunchKernelGGL((conv16r_kernel<2, NST>), dim3(blocks), dim3(256), LDSR, st, p);
  } else {
    g6d_allow_lds(reinterpret_cast<const void*>(&conv16r_kernel<3, NST>), LDSR);
    hipLaunchKernelGGL((conv16r_kernel<3, NST>), dim3(blocks), dim3(256), LDSR, st, p);
  }
  return g6d_check_launch("conv16r_direct");
}
