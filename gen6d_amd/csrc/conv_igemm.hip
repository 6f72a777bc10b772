// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
//   out[m][co] = act(bias[co] + sum_{tap,ci} X(m,tap,ci) * W[co][tap][ci]),   m = (n,od,oh,ow) flattened
//
// GEMM view: M = N*Do*Ho*Wo output positions, N = Cout, K = taps*Cin.  Activations are channels-last, so the K axis
// of both operands is contiguous in memory (128-byte runs of 32 channels): a 256-thread workgroup stages a
// [BM x 32] activation tile and a [BN x 32] weight tile per K step through LDS with 16-byte loads, LDS rows padded to
// 36 floats so that the ds_read_b128 fragment reads (lane i -> row i) are bank-conflict free.
//
// Fragment trick: v_mfma_f32_32x32x2_f32 wants A[i][k], B[k][j] with k = lane>>5.  The K order inside an 8-wide
// chunk is free as long as A and B agree, so lane-half h reads the 4 consecutive k = 4h..4h+3 with ONE ds_read_b128
// per operand and feeds them to 4 consecutive MFMAs (MFMA s consumes k = s and k = 4+s).
//
// Pipeline: double-buffered LDS, register-staged global prefetch of K step t+1 issued before the MFMAs of step t,
// one barrier per K step.  The operand loader applies the fused prologue (elementwise multiplier, InstanceNorm
// affine, ReLU; zero padding stays exactly zero); the epilogue adds bias/activation and accumulates the
// InstanceNorm statistics of the output in fp64.  Small grids are split along K into a workspace and reduced by a
// second kernel that carries the same epilogue.
#include "g6d_common.h"

#define LDS_K 36
#define BK 32

namespace {

template <int BM, int BN, int WGM, int WGN>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const G6dConv p, const int M, const int T,
                                                         const int nChunks, const int itersPerSplit,
                                                         const int totalIters, const int splits) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int RA = BM / 32, RB = BN / 32;
  constexpr int STAGE = (BM + BN) * LDS_K;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int lrow = tid >> 3, lseg = tid & 7;

  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int it_begin = blockIdx.z * itersPerSplit;
  const int it_end = min(totalIters, it_begin + itersPerSplit);

  const int Cin = p.Cin, khw = p.kh * p.kw;
  const float* __restrict__ gin = p.in;
  const float* __restrict__ gmul = p.mul;
  const float* __restrict__ gw = p.weight;
  const bool has_aff = p.in_scale != nullptr;
  const bool per_n = p.in_affine_per_n != 0;

  // ---- per-thread rows of the activation tile: decode output position once
  int an_[RA], az0[RA], ay0[RA], ax0[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    int m = m0 + lrow + 32 * j;
    if (m < M) {
      int ow = m % p.Wo; int t1 = m / p.Wo;
      int oh = t1 % p.Ho; int t2 = t1 / p.Ho;
      int od = t2 % p.Do; int n = t2 / p.Do;
      an_[j] = n; az0[j] = od * p.sd - p.pd; ay0[j] = oh * p.sh - p.ph; ax0[j] = ow * p.sw - p.pw;
    } else {
      an_[j] = 0; az0[j] = -(1 << 28); ay0[j] = 0; ax0[j] = 0;   // never valid
    }
  }
  // ---- per-thread rows of the weight tile
  size_t boff[RB]; bool bval[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    int co = n0 + lrow + 32 * j;
    bval[j] = co < p.Cout;
    boff[j] = (size_t)(bval[j] ? co : 0) * T * Cin;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[RA], rm[RA], rb[RB], rsc[RA], rsh[RA];
  bool va[RA];

  int tap = it_begin / nChunks;
  int cc = it_begin - tap * nChunks;

  auto issue_loads = [&](int tap_, int cc_) {
    const int kz = tap_ / khw; const int r_ = tap_ - kz * khw;
    const int ky = r_ / p.kw; const int kx = r_ - ky * p.kw;
    const int c = cc_ * BK + 4 * lseg;
    const bool cv = c < Cin;
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int iz = az0[j] + kz, iy = ay0[j] + ky, ix = ax0[j] + kx;
      const bool v = cv && (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      va[j] = v;
      ra[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (v) {
        const size_t pos = ((size_t)(an_[j] * p.Di + iz) * p.Hi + iy) * p.Wi + ix;
        ra[j] = *reinterpret_cast<const f32x4*>(gin + pos * p.ld_in + c);
        if (gmul) rm[j] = *reinterpret_cast<const f32x4*>(gmul + ((size_t)iy * p.Wi + ix) * Cin + c);
        if (has_aff && per_n) {
          rsc[j] = *reinterpret_cast<const f32x4*>(p.in_scale + (size_t)an_[j] * Cin + c);
          rsh[j] = *reinterpret_cast<const f32x4*>(p.in_shift + (size_t)an_[j] * Cin + c);
        }
      }
    }
    if (has_aff && !per_n && cv) {
      rsc[0] = *reinterpret_cast<const f32x4*>(p.in_scale + c);
      rsh[0] = *reinterpret_cast<const f32x4*>(p.in_shift + c);
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      rb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (bval[j] && cv) rb[j] = *reinterpret_cast<const f32x4*>(gw + boff[j] + (size_t)tap_ * Cin + c);
    }
  };

  auto store_stage = [&](float* As, float* Bs) {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      f32x4 v = ra[j];
      if (va[j]) {
        if (gmul) v *= rm[j];
        if (has_aff) { const int q = per_n ? j : 0; v = v * rsc[q] + rsh[q]; }
        if (p.in_relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      }
      *reinterpret_cast<f32x4*>(As + (lrow + 32 * j) * LDS_K + 4 * lseg) = v;
    }
#pragma unroll
    for (int j = 0; j < RB; ++j)
      *reinterpret_cast<f32x4*>(Bs + (lrow + 32 * j) * LDS_K + 4 * lseg) = rb[j];
  };

  auto compute = [&](const float* As, const float* Bs) {
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
  };

  if (it_begin < it_end) {
    issue_loads(tap, cc);
    store_stage(lds, lds + BM * LDS_K);
    __syncthreads();
    for (int it = it_begin; it < it_end; ++it) {
      const int cur = (it - it_begin) & 1;
      float* As = lds + cur * STAGE;
      float* Bs = As + BM * LDS_K;
      float* An = lds + (cur ^ 1) * STAGE;
      float* Bn = An + BM * LDS_K;
      const bool more = it + 1 < it_end;
      if (more) {
        if (++cc == nChunks) { cc = 0; ++tap; }
        issue_loads(tap, cc);
      }
      compute(As, Bs);
      if (more) store_stage(An, Bn);
      __syncthreads();
    }
  }

  // ---------------------------------------------------------------- epilogue
  const int Cout = p.Cout;
  if (splits > 1) {
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
  float* sred = lds;   // [BN][2], reused after the K loop (all waves passed the final barrier)
  if (do_stats && one_group) {
    for (int i = tid; i < BN * 2; i += 256) sred[i] = 0.f;
    __syncthreads();
  }
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
  if (do_stats && one_group) {
    __syncthreads();
    if (tid < BN && n0 + tid < Cout) {
      double* st = p.stats + ((size_t)g0 * Cout + n0 + tid) * 2;
      atomicAdd(st, (double)sred[tid * 2]);
      atomicAdd(st + 1, (double)sred[tid * 2 + 1]);
    }
  }
}

// Sum split-K partials, then the same epilogue as above. Block = 64 columns x 4 row lanes, 32 rows per block.
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int Cout,
                                                            const float* __restrict__ bias, int act,
                                                            float* __restrict__ out, int ld_out, double* stats, int rpg) {
  __shared__ float sred[64 * 2];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + tx;
  const int r0 = blockIdx.x * 32;
  const int rlast = min(r0 + 32, M) - 1;
  const int g0 = rpg > 0 ? r0 / rpg : 0;
  const bool one_group = rpg <= 0 || (rlast / rpg) == g0;
  if (threadIdx.x < 128) sred[threadIdx.x] = 0.f;
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
  if (col < Cout) {
    const float bv = bias ? bias[col] : 0.f;
    for (int i = 0; i < 8; ++i) {
      const int row = r0 + ty + 4 * i;
      if (row >= M) break;
      float v = 0.f;
      for (int z = 0; z < splits; ++z) v += ws[((size_t)z * M + row) * Cout + col];
      v = apply_act(v + bv, act);
      out[(size_t)row * ld_out + col] = v;
      if (stats) {
        if (one_group) { s1 += v; s2 += v * v; }
        else {
          double* st = stats + ((size_t)(row / rpg) * Cout + col) * 2;
          atomicAdd(st, (double)v); atomicAdd(st + 1, (double)v * v);
        }
      }
    }
  }
  if (stats && one_group) {
    atomicAdd(&sred[tx * 2], s1);
    atomicAdd(&sred[tx * 2 + 1], s2);
    __syncthreads();
    if (ty == 0 && col < Cout) {
      double* st = stats + ((size_t)g0 * Cout + col) * 2;
      atomicAdd(st, (double)sred[tx * 2]);
      atomicAdd(st + 1, (double)sred[tx * 2 + 1]);
    }
  }
}

template <int BM, int BN, int WGM, int WGN>
int launch_cfg(const G6dConv& d, int M, int T, int nChunks, int splits, hipStream_t stream) {
  const int total = T * nChunks;
  const int ips = (total + splits - 1) / splits;
  splits = (total + ips - 1) / ips;
  dim3 grid((M + BM - 1) / BM, (d.Cout + BN - 1) / BN, splits);
  const size_t lds_bytes = 2 * (size_t)(BM + BN) * LDS_K * sizeof(float);
  static bool attr_done = false;   // per template instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WGM, WGN>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN>), grid, dim3(256), lds_bytes, stream, d, M, T, nChunks, ips,
                     total, splits);
  int rc = g6d_check_launch("conv_igemm");
  if (rc != G6D_OK) return rc;
  if (splits > 1) {
    dim3 g2((M + 31) / 32, (d.Cout + 63) / 64);
    hipLaunchKernelGGL(splitk_reduce_kernel, g2, dim3(256), 0, stream, d.workspace, splits, M, d.Cout, d.bias, d.out_act,
                       d.out, d.ld_out, d.stats, d.stat_rows_per_group);
    rc = g6d_check_launch("splitk_reduce");
  }
  return rc;
}

}  // namespace

extern "C" int g6d_conv_igemm(const G6dConv* desc, g6d_stream_t stream_) {
  if (!desc) return G6D_EINVAL;
  const G6dConv& d = *desc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!d.in || !d.weight || !d.out) { g6d_set_error("conv: null pointer"); return G6D_EINVAL; }
  if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.Do <= 0 || d.Ho <= 0 || d.Wo <= 0 || d.kd <= 0 || d.kh <= 0 || d.kw <= 0) {
    g6d_set_error("conv: bad shape"); return G6D_EINVAL;
  }
  if ((d.Cin & 3) || (d.ld_in & 3) || d.ld_in < d.Cin || d.ld_out < d.Cout) { g6d_set_error("conv: Cin/ld_in must be multiples of 4"); return G6D_EINVAL; }
  if (!g6d_aligned16(d.in) || !g6d_aligned16(d.weight) || (d.mul && !g6d_aligned16(d.mul)) ||
      (d.in_scale && (!g6d_aligned16(d.in_scale) || !d.in_shift || !g6d_aligned16(d.in_shift)))) {
    g6d_set_error("conv: operand pointers must be 16-byte aligned"); return G6D_EINVAL;
  }
  const long long Mll = (long long)d.N * d.Do * d.Ho * d.Wo;
  if (Mll > (1ll << 30)) { g6d_set_error("conv: M too large"); return G6D_EINVAL; }
  const int M = (int)Mll;
  const int T = d.kd * d.kh * d.kw;
  const int nChunks = (d.Cin + BK - 1) / BK;
  const int total = T * nChunks;

  // tile configuration
  int bm = 128, bn = d.Cout <= 32 ? 32 : (d.Cout <= 64 ? 64 : 128);
  if (M <= 64 && d.Cout > 32) { bm = 64; bn = 64; }
  const long long blocks = (long long)((M + bm - 1) / bm) * ((d.Cout + bn - 1) / bn);

  int splits = d.split_k;
  if (splits <= 0) {
    splits = 1;
    if (blocks < 384 && total >= 8) {
      splits = (int)((768 + blocks - 1) / blocks);
      if (splits > total / 4) splits = total / 4;
      if (splits > 64) splits = 64;
      if (splits < 1) splits = 1;
    }
  }
  if (splits > total) splits = total;
  if (splits > 1) {
    const size_t need = (size_t)splits * M * d.Cout * sizeof(float);
    if (!d.workspace || d.workspace_bytes < need) {
      if (d.split_k > 1) { g6d_set_error("conv: workspace too small for forced split_k"); return G6D_ENOSPC; }
      size_t per = (size_t)M * d.Cout * sizeof(float);
      splits = d.workspace ? (int)(d.workspace_bytes / per) : 1;
      if (splits < 2) splits = 1;
    }
  }
  if (bm == 64) return launch_cfg<64, 64, 2, 2>(d, M, T, nChunks, splits, stream);
  if (bn == 32) return launch_cfg<128, 32, 4, 1>(d, M, T, nChunks, splits, stream);
  if (bn == 64) return launch_cfg<128, 64, 2, 2>(d, M, T, nChunks, splits, stream);
  return launch_cfg<128, 128, 2, 2>(d, M, T, nChunks, splits, stream);
}
