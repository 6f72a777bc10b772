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

// Epilogue, one PASS = one fp32 LDS tile ep[NPX px][68] of 64 output channels (already holding acc * acc_scale + bias (+ ReLU) of tile
// pixels px0 .. px0 + NPX - 1, the NT threads synchronised): 16-byte stores of the full map and / or the 2x2 max-pooled map, optional
// per-(group, channel) statistics.  NPX is a multiple of two tile rows (pool windows stay inside a pass).
// Output element types (full_type / pool_type): 1 = the 16-bit type T, 2 = fp32, 3 = fp16 hi / lo PAIR [pixel][2][Cout] (hi = rn16(v),
// lo = rn16(v - hi): the input format of the MM = 3 kernels; ld counts 16-bit elements and holds both planes).
// NT = threads that share the pass (256: the whole block, NPX = 128; 64: one wave, tid = lane, no block-wide synchronisation inside).
template <int MM, int NT, int NPX, int CH = 64>
__device__ __forceinline__ void c16_epilogue_pass(const C16Params& p, const C16Seg& sg, const float* ep, double* red, int tid, int cbase, int g0, int x0,
                                                  int tw_log2, int ylim, int px0) {
  typedef typename C16T<MM>::T T;
  typedef typename C16T<MM>::V V8;
  static_assert(NT == 64 || (NPX == 128 && CH == 64), "the block-wide form handles whole tiles of 64 channels");
  constexpr int C8 = CH / 8, C4 = CH / 4;                                           // 16-byte items per pixel: 16-bit / fp32 output
  const int TW = 1 << tw_log2, W = sg.W;
  const int TW2 = TW >> 1, py0 = px0 >> tw_log2, PR2 = (NPX >> tw_log2) >> 1;      // first tile row of the pass, pooled rows of the pass
  auto inside = [&](int lp, int& g, int& x) {                                       // lp: pixel of the pass = row of ep
    const int px = px0 + lp, py = px >> tw_log2;
    g = g0 + py; x = x0 + (px & (TW - 1));
    return py < ylim && g < sg.rows && x < W;
  };
  if (p.full_type == 1) {
#pragma unroll
    for (int j = 0; j < NPX * C8 / NT; ++j) {
      const int lp = tid / C8 + (NT / C8) * j, ch = (tid % C8) * 8;
      int g, x;
      if (inside(lp, g, x)) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(ep + lp * C16_EP_LD + ch), v1 = *reinterpret_cast<const f32x4*>(ep + lp * C16_EP_LD + ch + 4);
        V8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = (T)v0[e]; o[4 + e] = (T)v1[e]; }
        *reinterpret_cast<V8*>(sg.full + (((long)g * W + x) * sg.ld_full + cbase + ch) * 2) = o;
      }
    }
  } else if (p.full_type == 3) {
#pragma unroll
    for (int j = 0; j < NPX * C8 / NT; ++j) {
      const int lp = tid / C8 + (NT / C8) * j, ch = (tid % C8) * 8;
      int g, x;
      if (inside(lp, g, x)) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(ep + lp * C16_EP_LD + ch), v1 = *reinterpret_cast<const f32x4*>(ep + lp * C16_EP_LD + ch + 4);
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
    for (int j = 0; j < NPX * C4 / NT; ++j) {
      const int lp = tid / C4 + (NT / C4) * j, ch = (tid % C4) * 4;
      int g, x;
      if (inside(lp, g, x))
        *reinterpret_cast<f32x4*>(sg.full + (((long)g * W + x) * sg.ld_full + cbase + ch) * 4) = *reinterpret_cast<const f32x4*>(ep + lp * C16_EP_LD + ch);
    }
  }
  if (p.pool_type) {
    const int per = p.pool_type == 2 ? 4 : 8, chunks = CH / per;      // channels per item, items per pooled pixel
    for (int it = tid; it < (NPX / 4) * chunks; it += NT) {
      const int pp = it / chunks, ch = (it - pp * chunks) * per;
      const int pry = pp / TW2, prx = pp - pry * TW2;
      const int r00 = ((2 * pry) << tw_log2) + 2 * prx;
      const int g = g0 + py0 + 2 * pry, x = x0 + 2 * prx;
      if (pry < PR2 && py0 + 2 * pry < ylim && g < sg.rows && x < W) {
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
    const int grp = p.stat_rows_per_group > 0 ? (int)(((long)g0 * W) / p.stat_rows_per_group) : 0;
    if constexpr (NT == 256) {
      const int c = tid & 63, q = tid >> 6;
      float s1 = 0.f, s2 = 0.f;
      for (int lp = 32 * q; lp < 32 * q + 32; ++lp) {
        int g, x;
        if (inside(lp, g, x)) { const float v = ep[lp * C16_EP_LD + c]; s1 += v; s2 += v * v; }
      }
      red[(q * 64 + c) * 2] = (double)s1; red[(q * 64 + c) * 2 + 1] = (double)s2;
      __syncthreads();
      if (tid < 128) {
        const int cc = tid >> 1, w = tid & 1;
        const double v = red[(0 * 64 + cc) * 2 + w] + red[(1 * 64 + cc) * 2 + w] + red[(2 * 64 + cc) * 2 + w] + red[(3 * 64 + cc) * 2 + w];
        atomicAdd(p.stats + ((long)grp * p.Cout + cbase + cc) * 2 + w, v);
      }
    } else {
      // one wave: lane = channel; fp32 partial sums of 32 pixels each, combined in fp64
      if (tid >= CH) return;
      double d1 = 0.0, d2 = 0.0;
      for (int q = 0; q < NPX / 32; ++q) {
        float s1 = 0.f, s2 = 0.f;
        for (int lp = 32 * q; lp < 32 * q + 32; ++lp) {
          int g, x;
          if (inside(lp, g, x)) { const float v = ep[lp * C16_EP_LD + tid]; s1 += v; s2 += v * v; }
        }
        d1 += (double)s1; d2 += (double)s2;
      }
      atomicAdd(p.stats + ((long)grp * p.Cout + cbase + tid) * 2, d1);
      atomicAdd(p.stats + ((long)grp * p.Cout + cbase + tid) * 2 + 1, d2);
    }
  }
}

// Epilogue of the per-tap kernels: two passes of 64 channels through ONE fp32 LDS tile.
template <int MM, typename WriteTile>
__device__ __forceinline__ void c16_epilogue(const C16Params& p, const C16Seg& sg, char* lds, int tid, int nt, int g0, int x0, int tw_log2, int ylim,
                                             WriteTile write_tile) {
  // write_tile(h, ep, cbase): the waves that own channels [cbase, cbase + 64) of the block store acc * acc_scale + bias (+ ReLU) of the
  // tile's 128 pixels into ep[pixel * C16_EP_LD + channel - cbase];  ylim: tile rows below it lie inside the image
  float* ep = reinterpret_cast<float*>(lds);
  double* red = reinterpret_cast<double*>(lds + C16_BM * C16_EP_LD * 4);       // [4 quarters][64 ch][2] statistics partials
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    const int cbase = nt * C16_BN + 64 * h;
    write_tile(h, ep, cbase);
    __syncthreads();
    c16_epilogue_pass<MM, 256, 128>(p, sg, ep, red, tid, cbase, g0, x0, tw_log2, ylim, 0);
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
// conv16w_kernel: the HALO-PATCH kernel (2-D layers).  Measured on the two kernels above (tools/conv16_ablate.py): without the
// activation requests they run 15 % faster, without the filter requests 14 %, without both 38 % — a step's 32-48 KB of vector-memory
// traffic and two barriers per 16 MFMAs hold them at a third of the MFMA rate.  Here
//   * the activations of a 32-channel slice are staged ONCE per slice as the tile's halo patch — (TH + 2) x (TW + 2) pixels (tiles of
//     several small images: one band of H + 2 rows per image, so that zero padding between images stays zero) — and all nine taps read
//     shifted windows of it: ~1.4 KB of activation traffic per tap instead of 8, ONE barrier per slice (between barriers the waves run
//     free: the patch is read-only and the filters are private);
//   * a wave owns 128 pixels x 64 output channels (4 x 2 accumulator tiles): an activation fragment read from LDS feeds two MFMAs and
//     a filter fragment four — 64 B/clk/CU of ds_read_b128 and 32 B/clk/CU of filter loads at the full MFMA rate (LDS delivers 256,
//     L2 ~56).  A block is four waves: WM pixel tiles x (4 / WM) channel groups — 128 px x 256 ch, or 2 tiles x 128 ch for Cout = 128;
//   * filters: fragment-major, straight into registers TWO taps ahead (ring of three sets, hand-issued loads and counted waits);
//   * a patch row holds 64 B (4 slots of 16 B), lane-linear for the DMA, the slot swizzled by (patch row, column) so that the shifted
//     ds_read_b128 fragment reads of every tap are bank-conflict free (tools/ubench/conv16_swizzle.py); a fragment address is
//     row base | swizzled slot, and the second 16-channel half of the slice is that address ^ 32: one VALU per read;
//   * the nine taps are unrolled (tap offsets are immediates); the epilogue is PRIVATE to a wave (its own fp32 LDS tile, no block
//     barrier): blocks and waves drift apart freely, one block's epilogue runs beside its neighbours' MFMAs.
//   MM = 1 / 2: bf16 / fp16, a wave owns 128 px x 64 ch.  MM = 3: fp16 hi / lo pairs (see conv16r_kernel), two planes per patch, three
//   MFMAs per product; a wave owns 128 px x 32 ch (64 accumulator registers: with the two-plane fragment and filter sets a 64-channel
//   wave needs > 256 registers = one block per CU, every prologue and epilogue exposed).  Two blocks per CU in every mode.
// timing experiments only (WRONG results), a compile-time switch (make FLAGS_conv16_direct=-DC16W_ABLATE=n): after the first slice, bit 0 = no
// patch requests, bit 1 = no filter requests, bit 2 = no fragment reads
#ifndef C16W_ABLATE
#define C16W_ABLATE 0
#endif
template <int MM> struct C16W {
  static constexpr int NP = MM == 3 ? 2 : 1, KS = 2, NI_MAX = 18;             // planes, 16-channel groups per slice, DMA pieces per plane at most
  static constexpr int PLANE = NI_MAX * 1024;                                   // 288 patch rows of 64 B
  static constexpr int NT2 = MM == 3 ? 1 : 2;                                   // 32-channel accumulator tiles per wave (a wave = 128 px x 32 NT2 ch)
  static constexpr int NBL = KS * NP * NT2;                                     // filter loads per tap and wave
  static constexpr int NPX = 64;                                                // pixels per epilogue pass
  static constexpr int EPW = NPX * C16_EP_LD * 4;                               // a wave's epilogue tile
};

struct C16Geom {      // one 128-pixel tile of the halo tiling (block-uniform)
  int si, g0, x0, y0, ylim, tw_log2, PW, segh, bandr, P, swa, swd, ni;
};
__device__ __forceinline__ C16Geom c16w_geom(const C16Params& p, int ptile) {
  C16Geom o;
  int si = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if (i < p.nseg && ptile >= p.seg[i].h_tile0) si = i;
  const C16Seg& sg = p.seg[si];
  const int t = ptile - sg.h_tile0;
  o.si = si; o.tw_log2 = sg.h_tw_log2;
  const int TW = 1 << o.tw_log2, TH = C16_BM >> o.tw_log2;
  o.PW = TW + 2;
  if (sg.h_tpi > 0) {                                          // tiles inside one image
    const int n = t / sg.h_tpi, r = t - n * sg.h_tpi;
    const int ty = r / sg.h_tiles_x, tx = r - ty * sg.h_tiles_x;
    o.y0 = ty * TH; o.x0 = tx * TW; o.g0 = n * sg.H + o.y0; o.ylim = min(TH, sg.H - o.y0);
  } else {                                                     // tiles of TH / H whole images
    const int ty = t / sg.h_tiles_x, tx = t - ty * sg.h_tiles_x;
    o.y0 = 0; o.x0 = tx * TW; o.g0 = ty * TH; o.ylim = TH;
  }
  o.segh = sg.h_segh; o.bandr = sg.h_segh + 2; o.P = sg.h_bands * o.bandr * o.PW;
  o.swa = sg.h_swa; o.swd = sg.h_swd; o.ni = (o.P + 15) >> 4;
  return o;
}

// s_waitcnt vmcnt(BASE + n) for a wave-uniform run-time n in [0, HI] (the instruction takes an immediate only): a compare chain
template <int BASE, int HI>
__device__ __forceinline__ void c16_wait_vm(int n) {
  if constexpr (HI == 0) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE) : "memory");
  } else {
    if (n >= HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + HI) : "memory");
    else c16_wait_vm<BASE, HI - 1>(n);
  }
}

template <int MM, int WM, int NT2 = C16W<MM>::NT2>
__global__ __launch_bounds__(256, 2) void conv16w_kernel(const C16Params p) {
  typedef typename C16T3<MM>::V V8;
  typedef C16W<MM> R;
  constexpr int WN = 4 / WM, NP = R::NP, KS = R::KS, NBL = KS * NP * NT2, CW = 32 * NT2;   // (NT2 = 1 in a 16-bit mode: 32-channel waves for Cout = 64)
  constexpr bool SINGLE = MM == 3 && WM == 2;                  // one patch stage instead of two (see the slice hand-over below)
  constexpr int TILE_B = NP * R::PLANE, STAGE = WM * TILE_B;
  constexpr int NPIT = (NP * R::NI_MAX + 3) / 4;               // DMA wave-instructions per wave, tile and slice at most
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block id -> (tile group, channel block): the channel blocks of a tile group run back to back on ONE XCD (ids = xcd mod 8)
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int cb = jj % p.nN, tg = (jj / p.nN) * 8 + xcd;
  if (tg * WM >= p.ptiles) return;
  const int wm = wv / WN, wn = wv % WN;
  const int nchunk = p.Cin >> 5;
  typedef __attribute__((address_space(3))) void* lds_ptr;

  // ---- the patch pieces this lane requests per slice: tile tl, piece k = wv + 4 i of its NP * ni pieces (plane k / ni, rows 16 (k % ni) + lane / 4)
  // A piece none of whose lanes lies inside the map (halo pixels outside the image, the tail of the last piece) is not requested: its LDS
  // rows are zeroed here, once, in both stages (the geometry is the same for every slice).
  unsigned poff[WM][NPIT], pmask[WM];
  int t_ni[WM], t_si[WM];
#pragma unroll
  for (int tl = 0; tl < WM; ++tl) {
    const C16Geom gm = c16w_geom(p, min(tg * WM + tl, p.ptiles - 1));
    const C16Seg& sg = p.seg[gm.si];
    const int npt = NP * gm.ni;
    t_ni[tl] = gm.ni; t_si[tl] = gm.si;
    pmask[tl] = 0;
#pragma unroll
    for (int i = 0; i < NPIT; ++i) {
      const int k = wv + 4 * i;
      const int pl = k >= gm.ni ? 1 : 0, ii = k - pl * gm.ni;
      const int q = ii * 16 + (lane >> 2);
      const int prow = q / gm.PW, pcol = q - prow * gm.PW;
      const int b = prow / gm.bandr, lr = prow - b * gm.bandr - 1;   // band, row inside the band's image rows (-1 .. segh)
      const int g = gm.g0 + b * gm.segh + lr, x = gm.x0 + pcol - 1;
      const int yimg = gm.y0 + lr;                                    // (whole-image tiles: y0 = 0, lr = the image row)
      const bool ok = k < npt && q < gm.P && yimg >= 0 && yimg < sg.H && g < sg.rows && x >= 0 && x < sg.W;
      const int slot = (lane & 3) ^ (((pcol >> gm.swa) + prow * gm.swd) & 3);
      poff[tl][i] = ok ? (unsigned)((((long)g * sg.W + x) * sg.ld_in + pl * p.Cin) * 2 + slot * 16) : C16_OOB;
      if (k < npt) {
        if (__builtin_amdgcn_ballot_w64(ok) != 0) pmask[tl] |= 1u << i;
        else {
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          char* dst = lds + tl * TILE_B + pl * R::PLANE + ii * 1024 + lane * 16;
          *reinterpret_cast<f32x4*>(dst) = z;
          if constexpr (!SINGLE) *reinterpret_cast<f32x4*>(dst + STAGE) = z;
        }
      }
    }
    pmask[tl] = __builtin_amdgcn_readfirstlane(pmask[tl]);
  }
  auto issue_patch = [&](int c, int stage) {
    if (c >= nchunk) return;                                           // (no request past the last slice: the waits count 0 pieces there)
    const unsigned ck = (unsigned)(c * 64);
#pragma unroll
    for (int tl = 0; tl < WM; ++tl) {
      const C16Seg& sg = p.seg[t_si[tl]];
      const __amdgpu_buffer_rsrc_t rs_in = c16_rsrc(sg.in, sg.in_bytes - sg.back);
      const int ni = t_ni[tl];
#pragma unroll
      for (int i = 0; i < NPIT; ++i) {
        const int k = wv + 4 * i;
        if (pmask[tl] >> i & 1) {
          const int pl = k >= ni ? 1 : 0, ii = k - pl * ni;
          const unsigned vo = poff[tl][i] == C16_OOB ? C16_OOB : poff[tl][i] + ck;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(lds + stage * STAGE + tl * TILE_B + pl * R::PLANE + ii * 1024), 16, vo, 0, 0, 0);
        }
      }
    }
  };

  // ---- filter fragments of (slice c, tap t): KS x NP x 2 pieces of 16 bytes per lane of THIS wave's 64 channels, hand-issued (see
  // conv16r_kernel).  Packed K steps hold 64 channels (pairs: 32): slice c is half (c & 1) of packed step (c >> 1) * 9 + t.
  const int chan0 = (cb * WN + wn) * CW;
  const int nkp = (MM == 3 ? nchunk : nchunk >> 1) * 9;
  const char* wtile = p.w + (long)(chan0 >> 7) * nkp * 16384;
  unsigned bvo[KS * NP];
#pragma unroll
  for (int q = 0; q < KS * NP; ++q) bvo[q] = q * 4096 + ((chan0 >> 5) & 3) * 1024 + lane * 16;
  // The requests walk the packed filters in step order with a running pointer: + 16 KB per tap; at a slice boundary + 16 KB for pairs,
  // and for the 16-bit modes (two 32-channel slices per packed step) + 8 KB - 8 x 16 KB into an odd slice, + 8 KB out of it.  Past the
  // last step the pointer stays on the first one (a harmless reload: the request count per step stays uniform).
  const char* wp = wtile;
  int wleft = nchunk * 9;
  auto load_b = [&](int c, int t, V8 (&b)[NBL]) {                // (c, t): the step being requested — requests are issued in step order
    const unsigned long wa = (unsigned long)(wleft > 0 ? wp : wtile);
    const char* ws = reinterpret_cast<const char*>((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)wa) |
                                                   ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(wa >> 32)) << 32));
#pragma unroll
    for (int q = 0; q < KS * NP; ++q) {
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[NT2 * q]) : "v"(bvo[q]), "s"(ws) : "memory");
      if constexpr (NT2 == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(b[NT2 * q + 1]) : "v"(bvo[q]), "s"(ws) : "memory");
    }
    --wleft;
    if (t < 8 || MM == 3) wp += 16384;
    else wp += (c & 1) ? 8192 : 8192 - 8 * 16384;
  };

  // ---- fragment geometry of this wave's tile: m-tile mt = tile pixels 32 mt + (lane & 31) -> patch row of tap (0, 0)
  const C16Geom gm = c16w_geom(p, min(tg * WM + wm, p.ptiles - 1));
  const int fhalf = lane >> 5;
  // slot swizzle of patch (row, column): ((column >> swa) + row * swd) & 3, and swd != 0 only with swa == 0 (c16_halo_tiling): with
  // fe = column + row * swd (swa == 0) or column (swd == 0) the swizzle of tap (ky, kx) is ((fe + kx) >> swa) + ky * swd
  int qb[4], fe[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int r = 32 * mt + (lane & 31), py = r >> gm.tw_log2, px = r & ((1 << gm.tw_log2) - 1);
    const int b = py / gm.segh, ly = py - b * gm.segh;
    const int frow = b * gm.bandr + ly;
    qb[mt] = (frow * gm.PW + px) * 64 + wm * TILE_B; fe[mt] = px + frow * gm.swd;
  }
  const int PW64 = gm.PW * 64, swa = gm.swa, swd = gm.swd;
  auto tap_addr = [&](int ky, int kx, int stage, int (&tb)[4]) {
    const int so = stage * STAGE + (ky * PW64 + kx * 64);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int sw = ((fe[mt] + kx) >> swa) + ky * swd;
      tb[mt] = qb[mt] + so + (((fhalf ^ sw) & 3) << 4);
    }
  };
  auto frag_read = [&](int addr, V8 (&a)[NP]) {
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) a[pl] = *reinterpret_cast<const V8*>(lds + pl * R::PLANE + addr);
  };
  f32x16 acc[4][NT2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NT2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // the MFMAs of one m-tile and one 16-channel group: b[(ks * NP + plane) * NT2 + n2];  pairs: hi x hi, hi x lo, lo x hi
  auto mfmas = [&](int mt, const V8 (&a)[NP], const V8 (&b)[NBL], int ks) {
#pragma unroll
    for (int term = 0; term < (MM == 3 ? 3 : 1); ++term) {
      const int pa = term == 2 ? 1 : 0, pb = term == 1 ? 1 : 0;
#pragma unroll
      for (int n2 = 0; n2 < NT2; ++n2) acc[mt][n2] = c16_mfma<MM>(a[pa], b[(ks * NP + pb) * NT2 + n2], acc[mt][n2]);
    }
  };

  // ---- K loop: slices outermost (one patch, one barrier each), the nine taps unrolled inside.  FILTER loads return in order among
  // themselves, so before the MFMAs of step (c, t) vmcnt(2 NBL) — the two younger filter sets — proves B(c, t) has arrived.  LDS-DMA
  // requests are NOT ordered against them (measured on corr16_kernel: a counted wait that budgeted the younger patch pieces let MFMAs
  // start on filters still in flight once the filters missed L2 — the DMA of L2-resident activations overtakes them): the wait therefore
  // never budgets for DMA pieces (while some are in flight it is merely stricter than needed), and the wave's own pieces of the next
  // patch are drained with vmcnt(0) before the slice's barrier.
  // ONE set of activation fragments, replaced m-tile by m-tile: right after the MFMAs of (m-tile, 16-channel group) have issued, the
  // same registers receive the m-tile's fragment of the NEXT group (the other half of the slice, or the next tap) — consumed eight
  // MFMAs later.
  V8 bs[3][NBL];
  V8 fa[4][NP];
  int tb[4], ntb[4];
  issue_patch(0, 0);
  load_b(0, 0, bs[0]);
  load_b(0, 1, bs[1]);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (lgkmcnt: the zeroed halo pieces)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  tap_addr(0, 0, 0, tb);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) frag_read(tb[mt], fa[mt]);
  int stage = 0;
#pragma unroll 1
  for (int c = 0; c < ((C16W_ABLATE & 8) ? 0 : nchunk); ++c) {      // (ablation bit 3: no K loop at all — what a block costs outside it)
    // (opaque to the optimiser: it would otherwise hoist the 36 per-(tap, m-tile) swizzle terms out of the slice loop and spill them)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) asm volatile("" : "+v"(fe[mt]), "+v"(qb[mt]));
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int t2 = (t + 2) % 9, c2 = c + (t + 2) / 9;
      if (!((C16W_ABLATE & 2) && c > 0)) load_b(c2, t2, bs[(t + 2) % 3]);
      if (t == 0 && !SINGLE && !((C16W_ABLATE & 1) && c > 0)) issue_patch(c + 1, stage ^ 1);
      if (C16W_ABLATE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NBL) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (t < 8) tap_addr((t + 1) / 3, (t + 1) % 3, stage, ntb);
      else tap_addr(0, 0, SINGLE ? 0 : stage ^ 1, ntb);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        mfmas(mt, fa[mt], bs[t % 3], 0);
        if (!((C16W_ABLATE & 4) && c > 0)) frag_read(tb[mt] ^ 32, fa[mt]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        mfmas(mt, fa[mt], bs[t % 3], 1);
        if (t < 8 && !((C16W_ABLATE & 4) && c > 0)) frag_read(ntb[mt], fa[mt]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) tb[mt] = ntb[mt];
      if (t == 8) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (vmcnt: this wave's pieces of the next patch, see above)
        __builtin_amdgcn_s_barrier();                          // every wave has read this patch and received its share of the next
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SINGLE) {
          // one LDS stage (the pair kernel's two-tile form: two stages would be 147 KB = one block per CU): the next patch is requested
          // only now, into the stage everybody has just left; the other block of the CU computes while it arrives
          issue_patch(c + 1, 0);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        } else stage ^= 1;
        if (c + 1 < nchunk) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) frag_read(tb[mt], fa[mt]);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the look-ahead filter requests past the end have landed
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue, private to the wave: its 128 px x 64 ch go through its own fp32 LDS tile in passes of NPX pixels
  if (tg * WM + wm >= p.ptiles) return;                         // (the odd tile of the last group)
  const C16Seg& sg = p.seg[gm.si];
  float* ep = reinterpret_cast<float*>(lds + wv * R::EPW);
  float bv[NT2];
#pragma unroll
  for (int n2 = 0; n2 < NT2; ++n2) bv[n2] = p.bias ? p.bias[chan0 + 32 * n2 + (lane & 31)] : 0.f;
  constexpr int MTP = R::NPX / 32;                              // m-tiles per pass
#pragma unroll
  for (int ps = 0; ps < 4 / MTP; ++ps) {
#pragma unroll
    for (int m = 0; m < MTP; ++m)
#pragma unroll
      for (int n2 = 0; n2 < NT2; ++n2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          float v = fmaf(acc[ps * MTP + m][n2][r], p.acc_scale, bv[n2]);
          if (p.relu) v = fmaxf(v, 0.f);
          ep[px * C16_EP_LD + 32 * n2 + (lane & 31)] = v;
        }
    c16_epilogue_pass<MM == 3 ? 2 : MM, 64, R::NPX, CW>(p, sg, ep, nullptr, lane, chan0, gm.g0, gm.x0, gm.tw_log2, gm.ylim, ps * R::NPX);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// product_split_kernel: the selector's query x reference product (reference network/selector.py:183-186 — every hypothesis image is the
// reference's feature map times the query's, InstanceNorm'ed) written ONCE in the 16-bit activation format of conv16w_kernel, so that the
// first conv of every level can run on it: out[q D + d][px][plane][c] = split16((ref[d][px][c] * que[q][px][c]) * scale[q][c] + shift[q][c]).
// HBM-bound: one 16-byte (pairs: two) store per 8 channels; the reference cache (D P C floats) is re-read per query out of L2 / MALL.
template <int MM>
__global__ void __launch_bounds__(256) product_split_kernel(const float* __restrict__ ref, const float* __restrict__ que, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, char* __restrict__ out, int D, int P, int C, long total) {
  // a work item = 8 channels of one (query, pixel) for a run of PS_RUN hypotheses: the query's values and tables are loaded once per run
  // (one reference load and one / two 16-byte stores per output instead of four loads)
  typedef typename C16T3<MM>::T T;
  typedef typename C16T3<MM>::V V8;
  constexpr int PS_RUN = 8;
  const int c8 = C >> 3, nrun = (D + PS_RUN - 1) / PS_RUN;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c8);
    long r = i / c8;
    const int px = (int)(r % P); r /= P;
    const int run = (int)(r % nrun), q = (int)(r / nrun);
    const int c = cg * 8;
    const float* qp = que + ((long)q * P + px) * C + c;
    const f32x4 q0 = *reinterpret_cast<const f32x4*>(qp), q1 = *reinterpret_cast<const f32x4*>(qp + 4);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + (long)q * C + c), s1 = *reinterpret_cast<const f32x4*>(scale + (long)q * C + c + 4);
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(shift + (long)q * C + c), t1 = *reinterpret_cast<const f32x4*>(shift + (long)q * C + c + 4);
    const int d1 = min(D, (run + 1) * PS_RUN);
#pragma unroll 2
    for (int d = run * PS_RUN; d < d1; ++d) {
      const float* rp = ref + ((long)d * P + px) * C + c;
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = fmaf(r0[e] * q0[e], s0[e], t0[e]); v[4 + e] = fmaf(r1[e] * q1[e], s1[e], t1[e]); }
      V8 hi;
#pragma unroll
      for (int e = 0; e < 8; ++e) hi[e] = (T)v[e];
      const long row = ((long)q * D + d) * P + px;
      if constexpr (MM == 3) {
        V8 lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) lo[e] = (T)(v[e] - (float)hi[e]);
        char* o = out + (row * 2 * C + c) * 2;
        *reinterpret_cast<V8*>(o) = hi;
        *reinterpret_cast<V8*>(o + (long)C * 2) = lo;
      } else {
        *reinterpret_cast<V8*>(out + (row * C + c) * 2) = hi;
      }
    }
  }
}

// affine_split16_kernel: y = pool2x2?(relu?(x * scale[g][c] + shift[g][c])) of an fp32 channels-last map written in the 16-bit activation
// format of conv16w_kernel (pairs on the fp32 path) — the InstanceNorm affine + ReLU (+ MaxPool) between two convs of the selector's stacks
// (reference network/selector.py:27-77), which the Winograd / implicit-GEMM kernels apply in their operand prologue and a DMA-staged
// kernel cannot.  HBM-bound elementwise pass; image n uses table n / per_n (0: one table).
template <int MM>
__global__ void __launch_bounds__(256) affine_split16_kernel(const float* __restrict__ in, int ld_in, const float* __restrict__ scale, const float* __restrict__ shift,
                                                            int per_n, int relu, int pool, int H, int W, int C, char* __restrict__ out, long total) {
  typedef typename C16T3<MM>::T T;
  typedef typename C16T3<MM>::V V8;
  const int c8 = C >> 3, Ho = pool ? H >> 1 : H, Wo = pool ? W >> 1 : W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % c8);
    const long r = i / c8;                                     // output pixel (n Ho + y) Wo + x
    const int x = (int)(r % Wo);
    const long ny = r / Wo;
    const int y = (int)(ny % Ho);
    const long n = ny / Ho;
    const int c = cg * 8;
    const long tb = (per_n > 0 ? n / per_n : 0) * C + c;
    f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
    if (scale) {
      s0 = *reinterpret_cast<const f32x4*>(scale + tb); s1 = *reinterpret_cast<const f32x4*>(scale + tb + 4);
      t0 = *reinterpret_cast<const f32x4*>(shift + tb); t1 = *reinterpret_cast<const f32x4*>(shift + tb + 4);
    }
    float v[8];
    const int np = pool ? 2 : 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = -3.0e38f;
    for (int dy = 0; dy < np; ++dy)
      for (int dx = 0; dx < np; ++dx) {
        const float* p_ = in + (((long)n * H + (pool ? 2 * y + dy : y)) * W + (pool ? 2 * x + dx : x)) * ld_in + c;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(p_), a1 = *reinterpret_cast<const f32x4*>(p_ + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u0 = fmaf(a0[e], s0[e], t0[e]), u1 = fmaf(a1[e], s1[e], t1[e]);
          if (relu) { u0 = fmaxf(u0, 0.f); u1 = fmaxf(u1, 0.f); }
          v[e] = fmaxf(v[e], u0); v[4 + e] = fmaxf(v[4 + e], u1);
        }
      }
    V8 hi;
#pragma unroll
    for (int e = 0; e < 8; ++e) hi[e] = (T)v[e];
    if constexpr (MM == 3) {
      V8 lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) lo[e] = (T)(v[e] - (float)hi[e]);
      char* o = out + (r * 2 * C + c) * 2;
      *reinterpret_cast<V8*>(o) = hi;
      *reinterpret_cast<V8*>(o + (long)C * 2) = lo;
    } else {
      *reinterpret_cast<V8*>(out + (r * C + c) * 2) = hi;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// corr16_kernel: the detector's K x K correlation (K = 15, 7; reference network/detector.py:188-197,222-224: the query's feature map
// correlated with the 32 reference-centre features) on 16-bit activations — the halo-patch scheme of conv16w_kernel with K^2 taps
// per patch.  Cout = 32 (the reference views), so a 128-pixel tile has ONE wave's worth of output channels: the block's CORR16_NW = 11
// computing waves (three per SIMD with the loader, 150 registers each) split the TAPS of every slice (wave w: taps w, w + NW, ...), each
// accumulates its partial 128 px x 32 ch sums over all slices, and the partials are added through LDS at the end.  One more wave
// requests the patches (see inside).
//   * patch: (TH + K - 1) x (TW + K - 1) rows of 64 B, double-buffered (2 x 43 KB at K = 15); a row holds TWO fragments at addr and
//     addr ^ 32: the two 16-channel groups of a 32-channel slice (MM = 1 / 2), or the hi and lo plane of a 16-channel slice (MM = 3:
//     fp16 pairs, fp32-class results) — the same LDS geometry, addressing and filter traffic in both arithmetics;
//   * filters: [slice][tap][2 fragments][64 lanes][8 values] (host: ops.corr16_pack), 2 KB per (slice, tap), straight into registers two
//     of the wave's taps ahead; a slice's K^2 x 2 KB are read by one wave each;
//   * per tap a wave issues 8 (MM = 3: 12) MFMAs, 8 ds_read_b128 (one fragment set, replaced m-tile by m-tile) and ~24 address VALUs.
struct Corr16Seg {
  const char* in; float* out;
  int H, W, rows, ld_in, ld_out;
  int tw_log2, tiles_x, tpi, tile0, swa, swd;
  unsigned in_bytes;
};
struct Corr16Params {
  Corr16Seg seg[4];
  int nseg, Cin, K, ptiles, nslice;
  const char* w;
  float acc_scale;
};
constexpr int CORR16_NW = 11;                                // computing waves per block (+ one loader wave)
constexpr int CORR16_STAGE = 42 * 1024;                      // 672 patch rows of 64 B (K = 15, 8 x 16 tile: 660)

template <int MM>
__global__ __launch_bounds__(64 * (CORR16_NW + 1), 1) void corr16_kernel(const Corr16Params p) {
  constexpr int NW = CORR16_NW;                              // computing waves (the taps of a slice are dealt out to them in turn)
  typedef typename C16T3<MM>::V V8;
  constexpr int NPL = CORR16_STAGE / 1024;                    // patch pieces per slice at most (42)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ptile = blockIdx.x;
  int si = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if (i < p.nseg && ptile >= p.seg[i].tile0) si = i;
  const Corr16Seg& sg = p.seg[si];
  const int t = ptile - sg.tile0;
  const int tw_log2 = sg.tw_log2, TW = 1 << tw_log2, TH = C16_BM >> tw_log2, K = p.K, R = K >> 1;
  const int PW = TW + K - 1, P = (TH + K - 1) * PW, NI = (P + 15) >> 4;
  const int n = t / sg.tpi, r_ = t - n * sg.tpi;
  const int ty = r_ / sg.tiles_x, tx = r_ - ty * sg.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int swa = sg.swa, swd = sg.swd;
  const int nslice = p.nslice;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  if (wv == NW) {
    // ---- THE LOADER WAVE.  LDS-DMA requests are not ordered against ordinary loads (a counted wait that budgeted a patch's pieces beside
    // the filter loads let MFMAs start on filters still in flight, once the filters missed L2: wrong tiles at random), and they share
    // the one vmcnt counter of their wave: the patches are therefore requested by a wave of their own, which does nothing else — its
    // vmcnt(0) before every barrier is exact — and the eight computing waves count filter loads only.
    // piece k: rows 16 k + lane / 4; physical slot lane & 3 holds logical slot (lane & 3) ^ swizzle — 16-bit modes: channel group L of the
    // 32-channel slice; pairs: plane L >> 1, channel half L & 1 of the 16-channel slice
    unsigned poff[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int q = k * 16 + (lane >> 2);
      const int prow = q / PW, pcol = q - prow * PW;
      const int y = y0 - R + prow, x = x0 - R + pcol;
      const bool ok = k < NI && q < P && y >= 0 && y < sg.H && x >= 0 && x < sg.W;
      const int L = (lane & 3) ^ (((pcol >> swa) + prow * swd) & 3);
      const long px = ((long)n * sg.H + y) * sg.W + x;
      const unsigned within = MM == 3 ? (unsigned)((L >> 1) * p.Cin * 2 + (L & 1) * 16) : (unsigned)(L * 16);
      poff[k] = ok ? (unsigned)(px * sg.ld_in * 2) + within : C16_OOB;
    }
    const __amdgpu_buffer_rsrc_t rs_in = c16_rsrc(sg.in, sg.in_bytes);
    auto issue_patch = [&](int c, int stage) {
      const unsigned ck = (unsigned)(c * (MM == 3 ? 32 : 64));
#pragma unroll
      for (int k = 0; k < NPL; ++k)
        if (k < NI) {
          const unsigned vo = poff[k] == C16_OOB ? C16_OOB : poff[k] + ck;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(lds + stage * CORR16_STAGE + k * 1024), 16, vo, 0, 0, 0);
        }
    };
    issue_patch(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // patch 0 is in place
#pragma unroll 1
    for (int c = 0; c < nslice; ++c) {
      if (c + 1 < nslice) issue_patch(c + 1, (c + 1) & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                            // slice c has been read by every wave, patch c + 1 is in place
    }
  } else {
  // ---- filters: the wave's taps of every slice in order; running pointer, + 8 taps per step, past the end a harmless reload
  const int T = K * K;
  const int ntw = (T - wv + NW - 1) / NW;                     // taps of this wave per slice
  const char* wp = p.w + (long)wv * 2048;
  int wtap = wv, wleft = nslice * ntw;
  const unsigned bvo = lane * 16;
  auto load_b = [&](V8 (&b)[2]) {
    const unsigned long wa = (unsigned long)(wleft > 0 ? wp : p.w);
    const char* ws = reinterpret_cast<const char*>((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)wa) |
                                                   ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(wa >> 32)) << 32));
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[0]) : "v"(bvo), "s"(ws) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(b[1]) : "v"(bvo), "s"(ws) : "memory");
    --wleft;
    wtap += NW;
    if (wtap < T) wp += NW * 2048;
    else { wp += (long)(T - wtap + NW + wv) * 2048; wtap = wv; }      // first tap of this wave in the next slice
  };

  // ---- fragment geometry: m-tile mt = tile pixels 32 mt + (lane & 31); patch row of tap (0, 0) = the pixel's own (row, column)
  const int fhalf = lane >> 5;
  int qb[4], fe[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int r = 32 * mt + (lane & 31), py = r >> tw_log2, px = r & (TW - 1);
    qb[mt] = (py * PW + px) * 64; fe[mt] = px + py * swd;
  }
  const int PW64 = PW * 64;
  auto tap_addr = [&](int ky, int kx, int stage, int (&tb)[4]) {
    const int so = stage * CORR16_STAGE + ky * PW64 + kx * 64;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int sw = ((fe[mt] + kx) >> swa) + ky * swd;
      tb[mt] = qb[mt] + so + (((fhalf ^ sw) & 3) << 4);
    }
  };
  V8 bs[3][2];
  V8 fa[4][2];
  int tb[4], ntb[4];
  load_b(bs[0]);
  load_b(bs[1]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  const int ky0 = wv / K, kx0 = wv - ky0 * K;                 // the wave's first tap of a slice
  int ky = ky0, kx = kx0;
  tap_addr(ky, kx, 0, tb);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    fa[mt][0] = *reinterpret_cast<const V8*>(lds + tb[mt]);
    fa[mt][1] = *reinterpret_cast<const V8*>(lds + (tb[mt] ^ 32));
  }
  int stage = 0, it = 0, c = 0;                               // tap index of the wave inside the slice, slice
  // One tap.  The filter ring runs through all slices without a break (period 3: the loop below is unrolled by three steps, slice
  // boundaries are run-time state); the filters of the wave's tap after next are requested first.  Filter loads return in order and are
  // the only requests of this wave: vmcnt(4) — the two younger sets — proves this step's filters have arrived.
  auto step = [&](const V8 (&bc)[2], V8 (&bfar)[2]) {
    const bool live = c < nslice, last = it == ntw - 1;
    load_b(bfar);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!live) return;                                        // (padding steps after the last slice: the step count is a multiple of three)
    int nky = ky, nkx = kx + NW;
    while (nkx >= K) { nkx -= K; ++nky; }
    if (last) { nky = ky0; nkx = kx0; }
    tap_addr(nky, nkx, last ? stage ^ 1 : stage, ntb);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if constexpr (MM == 3) {
        acc[mt] = c16_mfma<MM>(fa[mt][0], bc[0], acc[mt]);      // hi x hi
        acc[mt] = c16_mfma<MM>(fa[mt][0], bc[1], acc[mt]);      // hi x lo
        acc[mt] = c16_mfma<MM>(fa[mt][1], bc[0], acc[mt]);      // lo x hi
      } else {
        acc[mt] = c16_mfma<MM>(fa[mt][0], bc[0], acc[mt]);      // channels 0..15 of the slice
        acc[mt] = c16_mfma<MM>(fa[mt][1], bc[1], acc[mt]);      // channels 16..31
      }
      if (!last) {
        fa[mt][0] = *reinterpret_cast<const V8*>(lds + ntb[mt]);
        fa[mt][1] = *reinterpret_cast<const V8*>(lds + (ntb[mt] ^ 32));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    ky = nky; kx = nkx;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) tb[mt] = ntb[mt];
    ++it;
    if (last) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                          // every wave has read this patch; the loader has the next one in place
      __builtin_amdgcn_sched_barrier(0);
      stage ^= 1; it = 0; ++c;
      if (c < nslice) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          fa[mt][0] = *reinterpret_cast<const V8*>(lds + tb[mt]);
          fa[mt][1] = *reinterpret_cast<const V8*>(lds + (tb[mt] ^ 32));
        }
      }
    }
  };
  const int nsteps = nslice * ntw;
#pragma unroll 1
  for (int s_ = 0; s_ < nsteps; s_ += 3) {
    asm volatile("" : "+v"(fe[0]), "+v"(fe[1]), "+v"(fe[2]), "+v"(fe[3]));
    step(bs[0], bs[2]);
    step(bs[1], bs[0]);
    step(bs[2], bs[1]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // ---- the eight partial tiles are added through LDS, four at a time: [slot][128 px][33]
  float* red = reinterpret_cast<float*>(lds);
  float sum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sum[e] = 0.f;
#pragma unroll 1
  for (int rd = 0; rd < (NW + 3) / 4; ++rd) {
    if (wv < NW && (wv >> 2) == rd) {
      float* mine = red + (wv & 3) * (128 * 33);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 33 + (lane & 31)] = acc[mt][r];
    }
    __syncthreads();
    if (wv < 8)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int o = tid + 512 * e, px = o >> 5, ch = o & 31;
      const int ns = NW - 4 * rd < 4 ? NW - 4 * rd : 4;        // partial tiles written in this round
      for (int sl = 0; sl < ns; ++sl) sum[e] += red[sl * 128 * 33 + px * 33 + ch];
    }
    __syncthreads();
  }
  if (wv < 8)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int o = tid + 512 * e, px = o >> 5, ch = o & 31;
    const int y = y0 + (px >> tw_log2), x = x0 + (px & (TW - 1));
    if (y < sg.H && x < sg.W) sg.out[(((long)n * sg.H + y) * sg.W + x) * sg.ld_out + ch] = sum[e] * p.acc_scale;
  }
}


// Tiling of one segment for the halo-patch kernel: the tile width (32 / 16 / 8 / 4) with the least overhang; false if none fits
// (a map lower than the tile must divide it: tiles of whole images)
bool c16_halo_tiling(const G6dConv16Seg& s, C16Seg& o, int& tiles) {
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
      o.h_swa = tw == 32 ? 2 : (tw == 16 ? 1 : 0); o.h_swd = tw <= 8 ? 1 : 0;             // (64-byte patch rows: 4 slots)
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
  // Cout = 64 (the selector's first product layer): halo-patch kernel only, filters packed as one 128-channel tile whose upper half is zero
  const bool half_tile = Cout == 64 && w_layout == 1 && kd == 1;      // (32-channel waves: two pixel tiles x two channel groups per block)
  if (Cin % bk || (Cout % C16_BN && !half_tile) || (kd != 1 && kd != 3)) { g6d_set_error("conv16_direct: Cin % 64 (32 for pairs), Cout % 128 (64: fragment-major 2-D layers), kd in {1,3} expected"); return G6D_EINVAL; }
  const int t16 = math_mode == 3 ? 3 : 1;                    // the 16-bit output coding of this mode
  auto type_ok = [&](int t) { return t == 0 || t == 2 || t == t16; };
  if (!type_ok(full_type) || !type_ok(pool_type) || (!full_type && !pool_type && !stats)) { g6d_set_error("conv16_direct: output types"); return G6D_EINVAL; }
  if (pool_type && kd != 1) { g6d_set_error("conv16_direct: pooling is 2-D only"); return G6D_EINVAL; }
  C16Params p = {};
  p.nseg = nseg; p.Cin = Cin; p.Cout = Cout; p.kd = kd; p.relu = relu; p.full_type = full_type; p.pool_type = pool_type;
  p.w = static_cast<const char*>(W16); p.bias = bias; p.stats = stats; p.stat_rows_per_group = stat_rows_per_group;
  p.acc_scale = acc_scale != 0.f ? acc_scale : 1.f;
  p.ablate = (int)g6d_knob(G6D_KNOB_C16_ABLATE);
  const long wb = (long)(half_tile ? 128 : Cout) * 9 * kd * Cin * 2 * planes;
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
      ok = c16_halo_tiling(segs[i], p.seg[i], nt_);
      p.seg[i].h_tile0 = htiles; htiles += nt_;
      if (stats && stat_rows_per_group > 0 && ok) {
        const C16Seg& o = p.seg[i];
        const int th = C16_BM >> o.h_tw_log2;
        // a tile must lie inside one statistics group: in-image tiles do when groups are whole images; tiles of whole images when the group is too
        ok = (stat_rows_per_group % (segs[i].H * segs[i].W) == 0) && (o.h_tpi > 0 || ((long)stat_rows_per_group % ((long)th * segs[i].W) == 0));
      }
    }
    if (!ok && half_tile) { g6d_set_error("conv16_direct: Cout = 64 needs a halo tiling for every segment"); return G6D_EINVAL; }
    if (ok) {
      // a block = 4 waves of 128 px x 64 ch (pairs: 32 ch): one pixel tile x 256 (128) channels, or two pixel tiles x 128 channels
      const int cw = half_tile ? 32 : 32 * (math_mode == 3 ? C16W<3>::NT2 : C16W<2>::NT2);
      const int wm = Cout % (4 * cw) == 0 ? 1 : 2;
      p.ptiles = htiles;
      p.nN = Cout / (cw * (4 / wm));
      const int hblocks = ((htiles + wm - 1) / wm + 7) / 8 * 8 * p.nN;
      auto launch = [&](auto kern, int main_lds, int ep_lds) {
        const int bytes = main_lds > ep_lds ? main_lds : ep_lds;
        g6d_allow_lds(reinterpret_cast<const void*>(kern), bytes);
        hipLaunchKernelGGL(kern, dim3(hblocks), dim3(256), bytes, st, p);
      };
#define C16W_LAUNCH(MM_, WM_) launch(&conv16w_kernel<MM_, WM_>, ((MM_ == 3 && WM_ == 2) ? 1 : 2) * WM_ * C16W<MM_>::NP * C16W<MM_>::PLANE, 4 * C16W<MM_>::EPW)
      if (half_tile && math_mode != 3) {
        if (math_mode == 1) launch(&conv16w_kernel<1, 2, 1>, 2 * 2 * C16W<1>::PLANE, 4 * C16W<1>::EPW);
        else launch(&conv16w_kernel<2, 2, 1>, 2 * 2 * C16W<2>::PLANE, 4 * C16W<2>::EPW);
      }
      else if (math_mode == 1) { if (wm == 1) C16W_LAUNCH(1, 1); else C16W_LAUNCH(1, 2); }
      else if (math_mode == 2) { if (wm == 1) C16W_LAUNCH(2, 1); else C16W_LAUNCH(2, 2); }
      else { if (wm == 1) C16W_LAUNCH(3, 1); else C16W_LAUNCH(3, 2); }   // (two tiles per block: Cout = 64)
#undef C16W_LAUNCH
      return g6d_check_launch("conv16w_direct");
    }
  }
  if (half_tile) { g6d_set_error("conv16_direct: Cout = 64 runs on the halo-patch kernel only (knob conv16_halo, tile shapes)"); return G6D_EINVAL; }
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

extern "C" int g6d_corr16_multi(const G6dConv16Seg* segs, int nseg, int Cin, const void* W16, float acc_scale, int Cout, int k, int math_mode,
                                g6d_stream_t stream) {
  if (!segs || nseg < 1 || nseg > 4 || !W16) { g6d_set_error("corr16: 1..4 segments and filters expected"); return G6D_EINVAL; }
  if (math_mode < 1 || math_mode > 3 || Cout != 32 || (k != 15 && k != 7) || Cin % 32) {
    g6d_set_error("corr16: math_mode 1 (bf16) / 2 (fp16) / 3 (fp16 pairs), Cout = 32, k in {7, 15}, Cin % 32 == 0 expected"); return G6D_EINVAL;
  }
  const int planes = math_mode == 3 ? 2 : 1;
  Corr16Params p = {};
  p.nseg = nseg; p.Cin = Cin; p.K = k; p.w = static_cast<const char*>(W16); p.acc_scale = acc_scale != 0.f ? acc_scale : 1.f;
  p.nslice = Cin / (math_mode == 3 ? 16 : 32);
  int tiles = 0;
  for (int i = 0; i < nseg; ++i) {
    const G6dConv16Seg& s = segs[i];
    Corr16Seg& o = p.seg[i];
    if (!s.in || !s.out_full || s.N < 1 || s.D != 1 || s.H < 1 || s.W < 1 || s.ld_in < planes * Cin || s.ld_full < Cout || (s.ld_in & 7) ||
        !g6d_aligned16(s.in) || !g6d_aligned16(s.out_full)) { g6d_set_error("corr16: bad segment"); return G6D_EINVAL; }
    // tile width with the least overhang whose halo patch fits a stage (672 rows)
    double best = 1e30; int btw = 0;
    for (int tw = 32; tw >= 4; tw >>= 1) {
      const int th = C16_BM / tw;
      if ((th + k - 1) * (tw + k - 1) > CORR16_STAGE / 64) continue;
      const double waste = (double)((s.W + tw - 1) / tw * tw) * ((s.H + th - 1) / th * th) / ((double)s.W * s.H);
      if (waste < best - 1e-9) { best = waste; btw = tw; }
    }
    if (!btw) { g6d_set_error("corr16: no tile shape fits"); return G6D_EINVAL; }
    int l2 = 0; while ((1 << l2) < btw) ++l2;
    const int th = C16_BM / btw;
    o.in = static_cast<const char*>(s.in); o.out = static_cast<float*>(s.out_full);
    o.H = s.H; o.W = s.W; o.rows = s.N * s.H; o.ld_in = s.ld_in; o.ld_out = s.ld_full;
    o.tw_log2 = l2; o.tiles_x = (s.W + btw - 1) / btw; o.tpi = o.tiles_x * ((s.H + th - 1) / th); o.tile0 = tiles;
    o.swa = btw == 32 ? 2 : (btw == 16 ? 1 : 0); o.swd = btw <= 8 ? 1 : 0;
    const long ext = (long)s.N * s.H * s.W * s.ld_in * 2;
    if (ext >= (1L << 31)) { g6d_set_error("corr16: a segment's input beyond 2 GB"); return G6D_EINVAL; }
    o.in_bytes = (unsigned)ext;
    tiles += s.N * o.tpi;
  }
  p.ptiles = tiles;
  hipStream_t st = static_cast<hipStream_t>(stream);
  constexpr int LDSB = 2 * CORR16_STAGE;                       // (>= the 4 x 128 x 33 floats of the final reduction)
  if (math_mode == 1) {
    g6d_allow_lds(reinterpret_cast<const void*>(&corr16_kernel<1>), LDSB);
    hipLaunchKernelGGL(corr16_kernel<1>, dim3(tiles), dim3(64 * (CORR16_NW + 1)), LDSB, st, p);
  } else if (math_mode == 2) {
    g6d_allow_lds(reinterpret_cast<const void*>(&corr16_kernel<2>), LDSB);
    hipLaunchKernelGGL(corr16_kernel<2>, dim3(tiles), dim3(64 * (CORR16_NW + 1)), LDSB, st, p);
  } else {
    g6d_allow_lds(reinterpret_cast<const void*>(&corr16_kernel<3>), LDSB);
    hipLaunchKernelGGL(corr16_kernel<3>, dim3(tiles), dim3(64 * (CORR16_NW + 1)), LDSB, st, p);
  }
  return g6d_check_launch("corr16");
}

extern "C" int g6d_product_split16(const float* ref, const float* que, const float* scale, const float* shift, void* out, int qn, int D, int P, int C,
                                   int math_mode, g6d_stream_t stream) {
  if (!ref || !que || !scale || !shift || !out || qn < 1 || D < 1 || P < 1 || C < 8 || (C & 7) || math_mode < 1 || math_mode > 3 ||
      !g6d_aligned16(ref) || !g6d_aligned16(que) || !g6d_aligned16(scale) || !g6d_aligned16(shift) || !g6d_aligned16(out)) {
    g6d_set_error("product_split16: bad args (C % 8 == 0, 16-byte aligned pointers, math_mode 1..3)"); return G6D_EINVAL;
  }
  const long total = (long)qn * ((D + 7) / 8) * P * (C >> 3);                  // work items: runs of 8 hypotheses
  const int blocks = (int)((total + 255) / 256 < 256 * 64 ? (total + 255) / 256 : 256 * 64);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* o = static_cast<char*>(out);
  if (math_mode == 1) hipLaunchKernelGGL(product_split_kernel<1>, dim3(blocks), dim3(256), 0, st, ref, que, scale, shift, o, D, P, C, total);
  else if (math_mode == 2) hipLaunchKernelGGL(product_split_kernel<2>, dim3(blocks), dim3(256), 0, st, ref, que, scale, shift, o, D, P, C, total);
  else hipLaunchKernelGGL(product_split_kernel<3>, dim3(blocks), dim3(256), 0, st, ref, que, scale, shift, o, D, P, C, total);
  return g6d_check_launch("product_split16");
}

extern "C" int g6d_affine_split16(const float* in, int ld_in, const float* scale, const float* shift, int affine_per_n, int relu, int pool, int N, int H, int W,
                                  int C, void* out, int math_mode, g6d_stream_t stream) {
  if (!in || !out || N < 1 || H < 1 || W < 1 || C < 8 || (C & 7) || ld_in < C || (ld_in & 3) || (scale && !shift) || affine_per_n < 0 || math_mode < 1 ||
      math_mode > 3 || (pool && ((H | W) & 1)) || !g6d_aligned16(in) || !g6d_aligned16(out) || (scale && (!g6d_aligned16(scale) || !g6d_aligned16(shift)))) {
    g6d_set_error("affine_split16: bad args (C % 8 == 0, 16-byte aligned rows, even map with pooling, math_mode 1..3)"); return G6D_EINVAL;
  }
  const long total = (long)N * (pool ? H / 2 : H) * (pool ? W / 2 : W) * (C >> 3);
  const int blocks = (int)((total + 255) / 256 < 256 * 64 ? (total + 255) / 256 : 256 * 64);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* o = static_cast<char*>(out);
  if (math_mode == 1) hipLaunchKernelGGL(affine_split16_kernel<1>, dim3(blocks), dim3(256), 0, st, in, ld_in, scale, shift, affine_per_n, relu, pool, H, W, C, o, total);
  else if (math_mode == 2) hipLaunchKernelGGL(affine_split16_kernel<2>, dim3(blocks), dim3(256), 0, st, in, ld_in, scale, shift, affine_per_n, relu, pool, H, W, C, o, total);
  else hipLaunchKernelGGL(affine_split16_kernel<3>, dim3(blocks), dim3(256), 0, st, in, ld_in, scale, shift, affine_per_n, relu, pool, H, W, C, o, total);
  return g6d_check_launch("affine_split16");
}
