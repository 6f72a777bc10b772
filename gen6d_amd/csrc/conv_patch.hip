// 3x3 / 3x3x3 stride-1 "same" convolution with the input patch of a spatial output tile kept in LDS and reused by all
// taps (64 output channels per block).  Generalises corr_patch.hip: measured on conv_igemm.hip, what costs matrix-pipe
// time is the amount of data brought into the CU per MFMA; an implicit-GEMM tile re-loads its 128
// activation rows for every one of the 9 / 27 taps, while a spatial tile of 128 outputs only needs its halo box once
// per channel chunk:
//   3-D: outputs 2x8x8, patch 4x10x10 = 400 positions (instead of 27 x 128 = 3456 row loads per chunk)
//   2-D: outputs 1x8x16, patch 1x10x18 = 180 positions (instead of 9 x 128 = 1152); 8x8 maps: 2 images per tile
// The prologue (multiplier, InstanceNorm affine with one table or one per image, ReLU, zero padding) is applied once
// per patch element when it is written to LDS instead of once per tap.  Weight tiles [64 co][32 ci] are streamed per
// tap (double-buffered); the patch has ONE buffer (next chunk prefetched into registers), see the kernel body.
// Same fragment scheme (K-permuted ds_read_b128 + v_mfma_f32_32x32x2_f32), accumulator layout and epilogue (bias,
// activation, channel-slice store, fp64 InstanceNorm statistics, split-K partials) as conv_igemm.hip.
#include "g6d_common.h"
#include <stdlib.h>
#include <type_traits>

#define LDS_K 36
#define BN 64


namespace {

__device__ __forceinline__ f32x4 ldg(const float* __restrict__ base, int elem_off) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + ((unsigned)elem_off << 2));
}

// Output tile geometries (128 outputs each): images per tile x depth x height x width, and the kernel depth.
//   KIND 0: 2-D  1 x 1 x 8 x 16          KIND 1: 3-D  1 x 2 x 8 x 8 (kd = 3)
//   KIND 2: 2-D  2 images x 8 x 8   (8x8 maps of the selector; 4x4 maps measured slower than the generic kernel:
//           a 6x6 halo per 4x4 outputs more than doubles the loaded rows, they stay on conv_igemm.hip)
template <int KIND> struct TileGeo;
template <> struct TileGeo<0> { static constexpr int TN = 1, TD = 1, TH = 8, TW = 16, KD = 1; };
template <> struct TileGeo<1> { static constexpr int TN = 1, TD = 2, TH = 8, TW = 8, KD = 3; };
template <> struct TileGeo<2> { static constexpr int TN = 2, TD = 1, TH = 8, TW = 8, KD = 1; };

// MODE: 0 plain, 1 affine(+ReLU), 3 mul + affine.
// VAR: 0 = fp32 MFMA, software-pipelined tap loop (default); 1 = fp32, plain tap loop (G6D_PATCH_PIPE=0, A/B measurements);
//      2 / 3 = bf16 / fp16 operands (G6dConv.math_mode 1 / 2), plain tap loop with 4 MFMAs of K = 16 per tap.
template <int KIND, int MODE, int VAR>
__global__ void __launch_bounds__(256) conv_patch_kernel(const G6dConv p, const int M, const int tiles_d, const int tiles_h,
                                                         const int tiles_w, const int chunks_per_split,
                                                         const int total_chunks, const int splits) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using G = TileGeo<KIND>;
  constexpr int TN = G::TN, TD = G::TD, TH = G::TH, TW = G::TW, KD = G::KD, T = KD * 9;
  constexpr int PD = TD + KD - 1, PH = TH + 2, PW = TW + 2, NPOS = TN * PD * PH * PW;
  constexpr int NPL = (NPOS * 8 + 255) / 256;                  // 16-byte patch loads per thread per chunk
  constexpr int PATCH = NPOS * LDS_K, BT = BN * LDS_K;
  constexpr bool AFF = MODE != 0, MUL = MODE == 3;
  // ONE patch buffer (the next chunk's patch waits in registers during the taps and is written between two barriers
  // after the last tap): 76 KB per block for the 3-D tile instead of 134 KB, so that two blocks — or a block of another
  // stream's kernel — share a CU.  Weight tiles stay double-buffered.
  float* patch = lds;
  float* bt0 = lds + PATCH; float* bt1 = bt0 + BT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;                     // wave tile 64 (M) x 32 (N)
  const int li = lane & 31, lh = lane >> 5;
  const int seg = tid & 7;

  // tile coordinates
  int t = blockIdx.x;
  const int tw = t % tiles_w; t /= tiles_w;
  const int th = t % tiles_h; t /= tiles_h;
  const int td = t % tiles_d; const int n = (t / tiles_d) * TN;          // first image of the tile
  const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
  const int n0 = blockIdx.y * BN;
  const int D = p.Di, H = p.Hi, W = p.Wi, Cin = p.Cin, Cout = p.Cout;
  const int c_begin = blockIdx.z * chunks_per_split, c_end = min(total_chunks, c_begin + chunks_per_split);

  const float* __restrict__ gin = p.in;
  const float* __restrict__ gmul = p.mul;
  const float* __restrict__ gw = p.weight;
  const int relu = p.in_relu;

  // ---- patch loader state: position q = tid/8 + 32*j of the patch, j < NPL
  int poff[NPL], pmul[NPL]; bool pval[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int q = (tid >> 3) + 32 * j;
    const int px = q % PW, py = (q / PW) % PH, pz = (q / (PW * PH)) % PD, pn = q / (PW * PH * PD);
    const int iz = d0 + pz - (KD / 2), iy = h0 + py - 1, ix = w0 + px - 1;
    pval[j] = (q < NPOS) & (n + pn < p.N) & ((unsigned)iz < (unsigned)D) & ((unsigned)iy < (unsigned)H) &
              ((unsigned)ix < (unsigned)W);
    poff[j] = pval[j] ? ((((n + pn) * D + iz) * H + iy) * W + ix) * p.ld_in : 0;
    pmul[j] = pval[j] ? (iy * W + ix) * Cin : 0;
  }
  // one (scale, shift) table per image group: the images of a tile share a group (host check)
  const int aff_off = p.in_affine_per_n ? (n / p.in_affine_per_n) * Cin : 0;
  f32x4 rp[NPL], rmul[MUL ? NPL : 1], rsc = {1.f, 1.f, 1.f, 1.f}, rsh = {0.f, 0.f, 0.f, 0.f};
  bool vp[NPL];
  auto load_patch = [&](int chunk) {
    const int c = chunk * 32 + 4 * seg;
    const bool cv = (c < Cin) & (chunk < c_end);
    if constexpr (AFF) { rsc = ldg(p.in_scale, cv ? aff_off + c : 0); rsh = ldg(p.in_shift, cv ? aff_off + c : 0); }
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const bool v = pval[j] & cv;
      vp[j] = v;
      rp[j] = ldg(gin, v ? poff[j] + c : 0);
      if constexpr (MUL) rmul[j] = ldg(gmul, v ? pmul[j] + c : 0);
    }
  };
  auto store_patch = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int q = (tid >> 3) + 32 * j;
      f32x4 v = rp[j];
      if constexpr (MUL) v *= rmul[j];
      if constexpr (AFF) {
        v = v * rsc + rsh;
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      }
      v = vp[j] ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      if (q < NPOS) *reinterpret_cast<f32x4*>(dst + q * LDS_K + 4 * seg) = v;
    }
  };
  // ---- weight tile loader: rows brow + 32*j, j < 2
  f32x4 rb[2]; bool vb[2];
  const int brow = tid >> 3;
  auto load_b = [&](int chunk, int tap) {
    const int c = chunk * 32 + 4 * seg;
    const bool cv = (c < Cin) & (chunk < c_end);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int co = n0 + brow + 32 * j;
      vb[j] = cv & (co < Cout);
      rb[j] = ldg(gw, vb[j] ? (co * T + tap) * Cin + c : 0);
    }
  };
  auto store_b = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *reinterpret_cast<f32x4*>(dst + (brow + 32 * j) * LDS_K + 4 * seg) = vb[j] ? rb[j] : f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- fragment base: output o = wm*64 + mt*32 + li of the tile -> patch position at tap (0,0,0)
  int abase[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int o = wm * 64 + mt * 32 + li;
    const int ow = o % TW, oh = (o / TW) % TH, od = (o / (TW * TH)) % TD, on = o / (TW * TH * TD);
    abase[mt] = (((on * PD + od) * PH + oh) * PW + ow) * LDS_K + 4 * lh;
  }
  const int bfrag = (wn * 32 + li) * LDS_K + 4 * lh;

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  if (c_begin < c_end) {
    load_patch(c_begin); load_b(c_begin, 0);
    store_patch(patch); store_b(bt0);
    __syncthreads();
    int bcur = 0;
    constexpr bool PIPE = VAR == 0;
    if constexpr (!PIPE) {
      for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const float* P = patch;
        load_patch(chunk + 1);                          // masked beyond c_end; lands during the T taps below
#pragma unroll 1
        for (int tap = 0; tap < T; ++tap) {
          const float* B = bcur ? bt1 : bt0;
          float* Bn = bcur ? bt0 : bt1;
          const bool last = tap == T - 1;
          load_b(last ? chunk + 1 : chunk, last ? 0 : tap + 1);
          const int kz = tap / 9, ky = (tap - kz * 9) / 3, kx = tap - kz * 9 - ky * 3;
          const int toff = ((kz * PH + ky) * PW + kx) * LDS_K;
          f32x4 a[4][2], b[4];
#pragma unroll
          for (int kc = 0; kc < 4; ++kc) {
            a[kc][0] = *reinterpret_cast<const f32x4*>(P + abase[0] + toff + kc * 8);
            a[kc][1] = *reinterpret_cast<const f32x4*>(P + abase[1] + toff + kc * 8);
            b[kc] = *reinterpret_cast<const f32x4*>(B + bfrag + kc * 8);
          }
          if constexpr (VAR >= 2) {
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
              acc[0] = g6d_mfma_lowp<VAR - 1>(a[2 * kp][0], a[2 * kp + 1][0], b[2 * kp], b[2 * kp + 1], acc[0]);
              acc[1] = g6d_mfma_lowp<VAR - 1>(a[2 * kp][1], a[2 * kp + 1][1], b[2 * kp], b[2 * kp + 1], acc[1]);
            }
          } else {
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kc][0][s], b[kc][s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kc][1][s], b[kc][s], acc[1], 0, 0, 0);
              }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (last && chunk + 1 < c_end) {
            __syncthreads();                              // every wave is done with this chunk's patch
            store_patch(patch);
          }
          store_b(Bn);
          __syncthreads();
          bcur ^= 1;
        }
      }
    } else {
      // Software-pipelined tap loop.  A tap = 4 groups (8-channel slices kc) of 8 MFMAs.  The fragments of group kc+1 are
      // requested in front of the MFMAs of group kc (two register sets, static parity because a tap has 4 groups), and the
      // LAST group of every tap is deferred across the tap's barrier with its operands already in registers (set 1): the
      // next tap opens with those 8 MFMAs, in whose shadow the first fragments of the new weight tile — which could not be
      // requested before the barrier — arrive.  With one block per CU (the 32^3 layers: 256 tiles) nothing else hides that
      // latency: the plain loop above leaves the matrix pipe idle for ~400 of every 2048 + 400 cycles.
      f32x4 fa[2][2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { fa[i][0] = fa[i][1] = fb[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      auto rd = [&](int set, const float* P, const float* B, int toff, int kc) {
        fa[set][0] = *reinterpret_cast<const f32x4*>(P + abase[0] + toff + kc * 8);
        fa[set][1] = *reinterpret_cast<const f32x4*>(P + abase[1] + toff + kc * 8);
        fb[set] = *reinterpret_cast<const f32x4*>(B + bfrag + kc * 8);
      };
      auto mm = [&](int set) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][0][s2], fb[set][s2], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][1][s2], fb[set][s2], acc[1], 0, 0, 0);
        }
      };
      for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const float* P = patch;
        load_patch(chunk + 1);                          // masked beyond c_end; lands during the T taps below
#pragma unroll 1
        for (int tap = 0; tap < T; ++tap) {
          const float* B = bcur ? bt1 : bt0;
          float* Bn = bcur ? bt0 : bt1;
          const bool last = tap == T - 1;
          const int kz = tap / 9, ky = (tap - kz * 9) / 3, kx = tap - kz * 9 - ky * 3;
          const int toff = ((kz * PH + ky) * PW + kx) * LDS_K;
          rd(0, P, B, toff, 0);                                    // group 0 of this tap
          __builtin_amdgcn_sched_barrier(0);
          mm(1);                                                   // deferred group 3 of the previous tap (zeros at the start)
          load_b(last ? chunk + 1 : chunk, last ? 0 : tap + 1);
          __builtin_amdgcn_sched_barrier(0);
          rd(1, P, B, toff, 1);
          __builtin_amdgcn_sched_barrier(0);
          mm(0);
          __builtin_amdgcn_sched_barrier(0);
          rd(0, P, B, toff, 2);
          __builtin_amdgcn_sched_barrier(0);
          mm(1);
          __builtin_amdgcn_sched_barrier(0);
          rd(1, P, B, toff, 3);                                    // operands of the deferred group stay in set 1
          __builtin_amdgcn_sched_barrier(0);
          mm(0);
          __builtin_amdgcn_sched_barrier(0);
          if (last && chunk + 1 < c_end) {
            __syncthreads();                              // every wave has its last fragments of this chunk's patch in registers
            store_patch(patch);
          }
          store_b(Bn);
          __syncthreads();
          bcur ^= 1;
        }
      }
      mm(1);                                              // the deferred group of the very last tap
    }
  }

  // ---------------------------------------------------------------- epilogue
  const int col = n0 + wn * 32 + li;
  const bool cval = col < Cout;
  auto out_row = [&](int mt, int r, bool& ok) {
    const int o = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int ow = o % TW, oh = (o / TW) % TH, od = (o / (TW * TH)) % TD, on = o / (TW * TH * TD);
    const int d = d0 + od, h = h0 + oh, w = w0 + ow;
    ok = (n + on < p.N) & (d < D) & (h < H) & (w < W);
    return (((n + on) * D + d) * H + h) * W + w;
  };
  if (splits > 1) {                   // lane-linear partial tile; the block that arrives last adds them up (g6d_common.h)
    constexpr int TILE = 256 * 32;
    const int ntiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x;
    float* part = p.workspace + G6D_WS_COUNTERS + (size_t)tile * TILE + tid * 4;
    const size_t zstride = (size_t)ntiles * TILE;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      g6d_store_wt(part + blockIdx.z * zstride + k * 1024,
                   f32x4{acc[k >> 2][4 * (k & 3)], acc[k >> 2][4 * (k & 3) + 1], acc[k >> 2][4 * (k & 3) + 2], acc[k >> 2][4 * (k & 3) + 3]});
    if (!g6d_split_arrive(reinterpret_cast<int*>(p.workspace) + tile, splits, reinterpret_cast<int*>(lds))) return;
    __syncthreads();                                           // the flag word is read; lds is reused below
    f32x4 sum[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sum[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < splits; ++z) {            // 8 pieces in flight: no more registers than the tap loop needs
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(part + (size_t)z * zstride + k * 1024);
#pragma unroll
      for (int k = 0; k < 8; ++k) sum[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc[k >> 2][4 * (k & 3)] = sum[k][0]; acc[k >> 2][4 * (k & 3) + 1] = sum[k][1];
      acc[k >> 2][4 * (k & 3) + 2] = sum[k][2]; acc[k >> 2][4 * (k & 3) + 3] = sum[k][3];
    }
  }
  const bool do_stats = p.stats != nullptr;
  // statistics groups are whole images (or runs of whole images); the images of a tile share a group (host check)
  const int g0 = p.stat_rows_per_group > 0 ? n / (p.stat_rows_per_group / (p.Do * p.Ho * p.Wo)) : 0;
  float* sred = lds;
  if (do_stats) {
    for (int i = tid; i < BN * 2; i += 256) sred[i] = 0.f;
    __syncthreads();
  }
  const float bv = (p.bias && cval) ? p.bias[col] : 0.f;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      bool ok; const int row = out_row(mt, r, ok);
      const float v = apply_act(acc[mt][r] + bv, p.out_act);
      if (ok && cval) {
        p.out[(size_t)row * p.ld_out + col] = v;
        s1 += v; s2 += v * v;
      }
    }
  if (do_stats) {
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lh == 0) { atomicAdd(&sred[(wn * 32 + li) * 2], s1); atomicAdd(&sred[(wn * 32 + li) * 2 + 1], s2); }
    __syncthreads();
    if (tid < BN && n0 + tid < Cout) {
      double* st = p.stats + ((size_t)g0 * Cout + n0 + tid) * 2;
      atomicAdd(st, (double)sred[tid * 2]);
      atomicAdd(st + 1, (double)sred[tid * 2 + 1]);
    }
    if (p.fin_scale) {
      __syncthreads();
      g6d_finalize_stats(g6d_fin_of(p), gridDim.x * gridDim.y, reinterpret_cast<int*>(lds));
    }
  }
}

template <int KIND, int MODE>
int launch_patch(const G6dConv& d, int M, hipStream_t stream) {
  using G = TileGeo<KIND>;
  constexpr int NPOS = G::TN * (G::TD + G::KD - 1) * (G::TH + 2) * (G::TW + 2);
  const int tiles_n = (d.N + G::TN - 1) / G::TN;
  const int tiles_d = (d.Di + G::TD - 1) / G::TD, tiles_h = (d.Hi + G::TH - 1) / G::TH, tiles_w = (d.Wi + G::TW - 1) / G::TW;
  const int tiles = tiles_n * tiles_d * tiles_h * tiles_w, ntn = (d.Cout + BN - 1) / BN;
  const int total_chunks = (d.Cin + 31) / 32;
  int splits = 1;
  if (tiles * ntn < 200 && total_chunks >= 4 && d.workspace) {
    splits = (400 + tiles * ntn - 1) / (tiles * ntn);
    if (splits > total_chunks / 2) splits = total_chunks / 2;
    const size_t per = (size_t)tiles * ntn * 256 * 32 * sizeof(float);      // tile-padded partials (>= M * Cout)
    const size_t room = d.workspace_bytes > G6D_WS_COUNTER_BYTES ? d.workspace_bytes - G6D_WS_COUNTER_BYTES : 0;
    if ((size_t)splits * per > room) splits = (int)(room / per);
    if (splits < 2 || tiles * ntn > G6D_WS_COUNTERS) splits = 1;
  }
  const int cps = (total_chunks + splits - 1) / splits;
  splits = (total_chunks + cps - 1) / cps;
  const size_t lds_bytes = (size_t)(NPOS * LDS_K + 2 * BN * LDS_K) * sizeof(float);
  const bool pipe = g6d_knob(G6D_KNOB_PATCH_PIPE) != 0;
  const int var = d.math_mode == 1 ? 2 : d.math_mode == 2 ? 3 : (pipe ? 0 : 1);
  auto go = [&](auto V) {
    constexpr int VAR = decltype(V)::value;
    g6d_allow_lds(reinterpret_cast<const void*>(&conv_patch_kernel<KIND, MODE, VAR>), (int)lds_bytes);
    hipLaunchKernelGGL((conv_patch_kernel<KIND, MODE, VAR>), dim3(tiles, ntn, splits), dim3(256), lds_bytes, stream, d, M,
                       tiles_d, tiles_h, tiles_w, cps, total_chunks, splits);
  };
  switch (var) {
    case 0: go(std::integral_constant<int, 0>{}); break;
    case 1: go(std::integral_constant<int, 1>{}); break;
    case 2: go(std::integral_constant<int, 2>{}); break;
    default: go(std::integral_constant<int, 3>{}); break;
  }
  int rc = g6d_check_launch("conv_patch");
  return rc;
}

template <int KIND>
int launch_kind(const G6dConv& d, int M, hipStream_t stream) {
  if constexpr (KIND != 1) { if (d.mul) return launch_patch<KIND, 3>(d, M, stream); }
  return d.in_scale ? launch_patch<KIND, 1>(d, M, stream) : launch_patch<KIND, 0>(d, M, stream);
}

// tile kind with the best fill for this layer, or -1
int pick_kind(const G6dConv& d, double* eff_out, long long* tiles_out) {
  const bool k3 = d.kd == 3;
  int best = -1; double best_eff = 0; long long best_tiles = 0;
  for (int kind = 0; kind < 3; ++kind) {
    if (k3 != (kind == 1)) continue;
    const int TN = kind == 2 ? 2 : 1, TD = kind == 1 ? 2 : 1, TH = 8, TW = kind == 0 ? 16 : 8;
    // the TN images of a tile must share their statistics group and their affine table
    if (TN > 1 && d.stats && d.stat_rows_per_group > 0 && (d.stat_rows_per_group / (d.Do * d.Ho * d.Wo)) % TN) continue;
    if (TN > 1 && d.in_scale && d.in_affine_per_n % TN) continue;
    const long long tiles = (long long)((d.N + TN - 1) / TN) * ((d.Di + TD - 1) / TD) * ((d.Hi + TH - 1) / TH) *
                            ((d.Wi + TW - 1) / TW);
    const double eff = (double)d.N * d.Di * d.Hi * d.Wi / (double)(tiles * 128);
    if (eff > best_eff + 1e-9) { best = kind; best_eff = eff; best_tiles = tiles; }
  }
  *eff_out = best_eff; *tiles_out = best_tiles;
  return best;
}

}  // namespace

// Eligibility (checked by the caller g6d_conv_igemm): stride 1, kernel (1,3,3) or (3,3,3) with "same" padding,
// statistics groups = whole images or one group, enough well-filled tiles (times channel-chunk splits) for the chip.
bool g6d_conv_patch_eligible(const G6dConv& d) {
  const bool k2 = d.kd == 1 && d.kh == 3 && d.kw == 3 && d.pd == 0 && d.ph == 1 && d.pw == 1 && d.Di == 1;
  const bool k3 = d.kd == 3 && d.kh == 3 && d.kw == 3 && d.pd == 1 && d.ph == 1 && d.pw == 1;
  if (!(k2 || k3) || d.sd != 1 || d.sh != 1 || d.sw != 1) return false;
  if (d.mul && !k2) return false;
  if (d.in_image_mod > 0 || d.mul_group_images > 0) return false;      // batched selector product: Winograd / generic kernel
  const int per_image = d.Do * d.Ho * d.Wo;
  if (d.stats && d.stat_rows_per_group > 0 && d.stat_rows_per_group % per_image) return false;   // groups = runs of whole images
  if (d.split_k > 0) return false;                                   // forced split counts go to the generic kernel
  double eff; long long tiles;
  if (pick_kind(d, &eff, &tiles) < 0) return false;
  const int chunks = (d.Cin + 31) / 32;
  const long long unsplit = tiles * ((d.Cout + BN - 1) / BN);
  const bool can_split = unsplit < 200 && chunks >= 4 && d.workspace;            // over channel chunks (launch_patch)
  // A grid of >= 128 well-filled tiles runs unsplit.  Smaller grids need the chunk split to reach >= 200 blocks: with
  // fewer, the serial (chunks x taps) loop of a block is longer than the generic kernel's split-K path (16^3 layers:
  // 94 vs 70 us), while the 7-image 2-D layers of the refiner feature net gain 5-20 %.
  // Wide layers re-load the patch once per 64-channel slice, still fewer rows than 9 / 27 tile loads per output.
  const bool enough = unsplit >= 128 || (can_split && unsplit * (chunks / 2) >= 200);
  return enough && eff >= 0.85 && d.Cout <= 256;
}

int g6d_conv_patch_launch(const G6dConv& d, int M, hipStream_t stream) {
  double eff; long long tiles;
  switch (pick_kind(d, &eff, &tiles)) {
    case 0: return launch_kind<0>(d, M, stream);
    case 1: return launch_kind<1>(d, M, stream);
    default: return launch_kind<2>(d, M, stream);
  }
}
