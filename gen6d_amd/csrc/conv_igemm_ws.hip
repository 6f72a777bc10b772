// Warp-specialised variant of the implicit-GEMM convolution (same math, tiles, LDS layout and epilogue as
// conv_igemm.hip).
//
// Measured on conv_igemm.hip (profiles/r01_conv_microbench.md): the matrix pipe alone sustains ~123 TFLOP/s, the full
// kernel 84-92: the gap is the per-wave cost of generating addresses and issuing the global loads in the same
// instruction stream as the MFMAs; LDS traffic and the barrier are free.  Here the two jobs run in DIFFERENT waves:
//   waves 0-3 (consumers): LDS fragment reads + v_mfma_f32_32x32x2_f32 only;
//   waves 4-7 (producers): address generation, global loads (issued two K steps ahead into two register sets),
//                          prologue transform (multiplier / norm affine / ReLU / zero padding) and the LDS stores of K
//                          step t+1, while the consumers work on step t.
// Every SIMD hosts one consumer and one producer wave; they issue to different pipes, so the load path disappears
// behind the matrix work.  One barrier per K step as before.  Each producer owns a quarter of the activation rows and
// a quarter of the weight rows of the tile (8 lanes x 16 bytes per 128-byte row, 8 rows per wave instruction).
#include "g6d_common.h"
#include <type_traits>

#define LDS_K 36
#define BK 32

int g6d_splitk_reduce_launch(const float* ws, int splits, int M, int Cout, const float* bias, int act, float* out,
                             int ld_out, double* stats, int rows_per_group, hipStream_t stream);

namespace {

__device__ __forceinline__ f32x4 ldg(const float* __restrict__ base, int elem_off) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + ((unsigned)elem_off << 2));
}

// MODE: 0 = plain operand, 1 = affine(+ReLU) with one table, 3 = elementwise multiplier + affine.
template <int BM, int BN, int WGM, int WGN, int MODE>
__global__ void __launch_bounds__(512) conv_igemm_ws_kernel(const G6dConv p, const int M, const int T, const int nChunks,
                                                            const int itersPerSplit, const int totalIters,
                                                            const int splits) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int NA = BM / 32, NB = BN / 32;           // 8-row groups per producer wave (a quarter of the tile each)
  constexpr int STAGE = (BM + BN) * LDS_K;
  constexpr bool AFF = MODE != 0, MUL = MODE == 3;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int it_begin = blockIdx.z * itersPerSplit;
  const int it_end = min(totalIters, it_begin + itersPerSplit);
  const int Cout = p.Cout;

  // consumer coordinates
  const int wm = (wave & 3) / WGN, wn = (wave & 3) % WGN;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (it_begin < it_end) {
    if (producer) {
      // ------------------------------------------------------------------------------------------ producer waves
      const int L = wave - 4;                            // 0..3
      const int r8 = lane >> 3, lseg = lane & 7;
      const int Cin = p.Cin, khw = p.kh * p.kw;
      const float* __restrict__ gin = p.in;
      const float* __restrict__ gmul = p.mul;
      const float* __restrict__ gw = p.weight;
      const float* __restrict__ gsc = p.in_scale;
      const float* __restrict__ gsh = p.in_shift;
      const int relu = p.in_relu;

      int az0[NA], ay0[NA], ax0[NA], abase[NA], mbase[NA], arow[NA];
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        arow[j] = (L * NA + j) * 8 + r8;                 // row of the tile
        const int m = m0 + arow[j];
        if (m < M) {
          int ow = m % p.Wo; int t1 = m / p.Wo;
          int oh = t1 % p.Ho; int t2 = t1 / p.Ho;
          int od = t2 % p.Do; int n = t2 / p.Do;
          az0[j] = od * p.sd - p.pd; ay0[j] = oh * p.sh - p.ph; ax0[j] = ow * p.sw - p.pw;
          abase[j] = (((n * p.Di + az0[j]) * p.Hi + ay0[j]) * p.Wi + ax0[j]) * p.ld_in;
          mbase[j] = (ay0[j] * p.Wi + ax0[j]) * Cin;
        } else {
          az0[j] = -(1 << 28); ay0[j] = 0; ax0[j] = 0; abase[j] = 0; mbase[j] = 0;
        }
      }
      int boff[NB], brow[NB]; bool bval[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        brow[j] = (L * NB + j) * 8 + r8;
        const int co = n0 + brow[j];
        bval[j] = co < Cout;
        boff[j] = (bval[j] ? co : 0) * T * Cin;
      }

      // K position: taps fastest inside a channel chunk (as conv_igemm.hip)
      int cc = it_begin / T;
      int tap = it_begin - cc * T;
      int kz = tap / khw, ky = (tap - kz * khw) / p.kw, kx = tap - kz * khw - ky * p.kw;
      auto advance = [&]() {
        tap += 1; kx += 1; const bool w1 = kx == p.kw; kx = w1 ? 0 : kx;
        ky += w1; const bool w2 = ky == p.kh; ky = w2 ? 0 : ky;
        kz += w2; const bool w3 = kz == p.kd; kz = w3 ? 0 : kz;
        tap = w3 ? 0 : tap; cc += w3;
      };

      // two register sets: tile u lives in set u&1 from its load (issued during K step u-2) to its LDS store (step u-1)
      f32x4 ra[2][NA], rm[2][MUL ? NA : 1], rb[2][NB], rsc[2], rsh[2];
      bool va[2][NA], vb[2][NB];
      auto load_tile = [&](auto S) {
        constexpr int s = decltype(S)::value;
        const int cch = cc * BK + 4 * lseg;
        const bool cv = cch < Cin;
        const int toff = ((kz * p.Hi + ky) * p.Wi + kx) * p.ld_in + cch;
        const int moff = (ky * p.Wi + kx) * Cin + cch;
        const int woff = tap * Cin + cch;
        if constexpr (AFF) { rsc[s] = ldg(gsc, cv ? cch : 0); rsh[s] = ldg(gsh, cv ? cch : 0); }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const bool v = cv & ((unsigned)(az0[j] + kz) < (unsigned)p.Di) & ((unsigned)(ay0[j] + ky) < (unsigned)p.Hi) &
                         ((unsigned)(ax0[j] + kx) < (unsigned)p.Wi);
          va[s][j] = v;
          ra[s][j] = ldg(gin, v ? abase[j] + toff : 0);
          if constexpr (MUL) rm[s][j] = ldg(gmul, v ? mbase[j] + moff : 0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const bool v = bval[j] & cv;
          vb[s][j] = v;
          rb[s][j] = ldg(gw, v ? boff[j] + woff : 0);
        }
      };
      auto store_tile = [&](auto S, float* As, float* Bs) {
        constexpr int s = decltype(S)::value;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          f32x4 v = ra[s][j];
          if constexpr (MUL) v *= rm[s][j];
          if constexpr (AFF) {
            v = v * rsc[s] + rsh[s];
            if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
          }
          v = va[s][j] ? v : f32x4{0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(As + arow[j] * LDS_K + 4 * lseg) = v;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
          *reinterpret_cast<f32x4*>(Bs + brow[j] * LDS_K + 4 * lseg) = vb[s][j] ? rb[s][j] : f32x4{0.f, 0.f, 0.f, 0.f};
      };
      using S0 = std::integral_constant<int, 0>;
      using S1 = std::integral_constant<int, 1>;
      float* L0 = lds; float* L1 = lds + STAGE;

      load_tile(S0{});                                           // tile 0
      if (it_begin + 1 < it_end) { advance(); load_tile(S1{}); } // tile 1
      store_tile(S0{}, L0, L0 + BM * LDS_K);
      __syncthreads();
      // K step `it` (consumers on stage it&1): store tile it+1 (landed during the previous step) into the other stage,
      // then issue the loads of tile it+2 into the register set tile `it` has just vacated.
      for (int it = it_begin; it < it_end; it += 2) {
        if (it + 1 < it_end) store_tile(S1{}, L1, L1 + BM * LDS_K);
        if (it + 2 < it_end) { advance(); load_tile(S0{}); }
        __syncthreads();
        if (it + 1 < it_end) {
          if (it + 2 < it_end) store_tile(S0{}, L0, L0 + BM * LDS_K);
          if (it + 3 < it_end) { advance(); load_tile(S1{}); }
          __syncthreads();
        }
      }
    } else {
      // ------------------------------------------------------------------------------------------ consumer waves
      __syncthreads();                                  // tile 0 is in LDS
      for (int it = it_begin; it < it_end; ++it) {
        const int cur = (it - it_begin) & 1;
        const float* As = lds + cur * STAGE;
        const float* Bs = As + BM * LDS_K;
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
          f32x4 a[MT], b[NT];
#pragma unroll
          for (int i = 0; i < MT; ++i)
            a[i] = *reinterpret_cast<const f32x4*>(As + (wm * WM + i * 32 + li) * LDS_K + kc * 8 + 4 * lh);
#pragma unroll
          for (int j = 0; j < NT; ++j)
            b[j] = *reinterpret_cast<const f32x4*>(Bs + (wn * WN + j * 32 + li) * LDS_K + kc * 8 + 4 * lh);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
      }
    }
  }

  // ---------------------------------------------------------------- epilogue (consumer waves hold the accumulators)
  if (splits > 1) {
    if (producer) return;
    float* ws = p.workspace + (size_t)blockIdx.z * M * Cout;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * WN + j * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < M && col < Cout) ws[(size_t)row * Cout + col] = acc[i][j][r];
        }
      }
    return;
  }
  const bool do_stats = p.stats != nullptr;
  const int rpg = p.stat_rows_per_group;
  const int mlast = min(m0 + BM, M) - 1;
  const int g0 = rpg > 0 ? m0 / rpg : 0;
  const bool one_group = rpg <= 0 || (mlast / rpg) == g0;
  float* sred = lds;   // [BN][2], reused after the K loop
  if (do_stats && one_group) {
    for (int i = tid; i < BN * 2; i += 512) sred[i] = 0.f;
    __syncthreads();
  }
  if (!producer) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * WN + j * 32 + li;
      const bool cval = col < Cout;
      const float bv = (p.bias && cval) ? p.bias[col] : 0.f;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          float v = apply_act(acc[i][j][r] + bv, p.out_act);
          if (row < M && cval) {
            p.out[(size_t)row * p.ld_out + col] = v;
            if (do_stats) {
              if (one_group) { s1 += v; s2 += v * v; }
              else {
                double* st = p.stats + ((size_t)(row / rpg) * Cout + col) * 2;
                atomicAdd(st, (double)v); atomicAdd(st + 1, (double)v * v);
              }
            }
          }
        }
      }
      if (do_stats && one_group) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (lh == 0) {
          atomicAdd(&sred[(wn * WN + j * 32 + li) * 2], s1);
          atomicAdd(&sred[(wn * WN + j * 32 + li) * 2 + 1], s2);
        }
      }
    }
  }
  if (do_stats && one_group) {
    __syncthreads();
    if (tid < BN && n0 + tid < Cout) {
      double* st = p.stats + ((size_t)g0 * Cout + n0 + tid) * 2;
      atomicAdd(st, (double)sred[tid * 2]);
      atomicAdd(st + 1, (double)sred[tid * 2 + 1]);
    }
  }
}

template <int BM, int BN, int WGM, int WGN, int MODE>
int launch_ws_mode(const G6dConv& d, int M, int T, int nChunks, int splits, hipStream_t stream) {
  const int total = T * nChunks;
  const int ips = (total + splits - 1) / splits;
  splits = (total + ips - 1) / ips;
  dim3 grid((M + BM - 1) / BM, (d.Cout + BN - 1) / BN, splits);
  const size_t lds_bytes = 2 * (size_t)(BM + BN) * LDS_K * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_ws_kernel<BM, BN, WGM, WGN, MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_igemm_ws_kernel<BM, BN, WGM, WGN, MODE>), grid, dim3(512), lds_bytes, stream, d, M, T, nChunks,
                     ips, total, splits);
  int rc = g6d_check_launch("conv_igemm_ws");
  if (rc != G6D_OK) return rc;
  if (splits > 1)
    rc = g6d_splitk_reduce_launch(d.workspace, splits, M, d.Cout, d.bias, d.out_act, d.out, d.ld_out, d.stats,
                                  d.stat_rows_per_group, stream);
  return rc;
}

template <int BM, int BN, int WGM, int WGN>
int launch_ws_cfg(const G6dConv& d, int M, int T, int nChunks, int splits, hipStream_t stream) {
  if (d.mul) return launch_ws_mode<BM, BN, WGM, WGN, 3>(d, M, T, nChunks, splits, stream);
  if (!d.in_scale) return launch_ws_mode<BM, BN, WGM, WGN, 0>(d, M, T, nChunks, splits, stream);
  return launch_ws_mode<BM, BN, WGM, WGN, 1>(d, M, T, nChunks, splits, stream);
}

}  // namespace

// Called by g6d_conv_igemm for the tile shapes / prologue modes this variant covers (bn in {32, 64, 128}, bm = 128,
// no per-image affine table).
int g6d_conv_igemm_ws_launch(const G6dConv& d, int M, int T, int nChunks, int bn, int splits, hipStream_t stream) {
  if (bn == 32) return launch_ws_cfg<128, 32, 4, 1>(d, M, T, nChunks, splits, stream);
  if (bn == 64) return launch_ws_cfg<128, 64, 2, 2>(d, M, T, nChunks, splits, stream);
  return launch_ws_cfg<128, 128, 2, 2>(d, M, T, nChunks, splits, stream);
}
