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

// MODE: 0 = plain operand, 1 = affine(+ReLU) with one table, 2 = affine(+ReLU) with a table per batch index n,
//       3 = elementwise multiplier + affine (selector product).
template <int BM, int BN, int WGM, int WGN, int MODE>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const G6dConv p, const int M, const int T,
                                                         const int nChunks, const int itersPerSplit,
                                                         const int totalIters, const int splits) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int RA = BM / 32, RB = BN / 32;
  constexpr int STAGE = (BM + BN) * LDS_K;
  constexpr bool AFF = MODE != 0, PER_N = MODE == 2, MUL = MODE == 3;

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
  const float* __restrict__ gsc = p.in_scale;
  const float* __restrict__ gsh = p.in_shift;
  const int relu = p.in_relu;

  // ---- per-thread rows of the activation tile: decode the output position once.  abase is the (possibly virtual,
  //      i.e. out-of-range) element offset of input voxel (n, iz0, iy0, ix0); every tap adds a uniform offset.
  int az0[RA], ay0[RA], ax0[RA], abase[RA], mbase[RA], nbase[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    int m = m0 + lrow + 32 * j;
    if (m < M) {
      int ow = m % p.Wo; int t1 = m / p.Wo;
      int oh = t1 % p.Ho; int t2 = t1 / p.Ho;
      int od = t2 % p.Do; int n = t2 / p.Do;
      az0[j] = od * p.sd - p.pd; ay0[j] = oh * p.sh - p.ph; ax0[j] = ow * p.sw - p.pw;
      abase[j] = (((n * p.Di + az0[j]) * p.Hi + ay0[j]) * p.Wi + ax0[j]) * p.ld_in;
      mbase[j] = (ay0[j] * p.Wi + ax0[j]) * Cin;
      nbase[j] = n * Cin;
    } else {
      az0[j] = -(1 << 28); ay0[j] = 0; ax0[j] = 0; abase[j] = 0; mbase[j] = 0; nbase[j] = 0;   // never valid
    }
  }
  // ---- per-thread rows of the weight tile
  int boff[RB]; bool bval[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    int co = n0 + lrow + 32 * j;
    bval[j] = co < p.Cout;
    boff[j] = (bval[j] ? co : 0) * T * Cin;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[RA], rm[RA], rb[RB], rsc[PER_N ? RA : 1], rsh[PER_N ? RA : 1];
  bool va[RA], vb[RB];

  int tap = it_begin / nChunks;
  int cc = it_begin - tap * nChunks;

  // All loads are unconditional (clamped to element 0 when masked) so that they pipeline; masking happens in
  // store_stage.  A branch around a load makes hipcc wait for each one separately.
  auto issue_loads = [&](int tap_, int cc_) {
    const int kz = tap_ / khw; const int r_ = tap_ - kz * khw;
    const int ky = r_ / p.kw; const int kx = r_ - ky * p.kw;
    const int c = cc_ * BK + 4 * lseg;
    const bool cv = c < Cin;
    const int toff = ((kz * p.Hi + ky) * p.Wi + kx) * p.ld_in + c;
    const int moff = (ky * p.Wi + kx) * Cin + c;
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const bool v = cv && (unsigned)(az0[j] + kz) < (unsigned)p.Di && (unsigned)(ay0[j] + ky) < (unsigned)p.Hi &&
                     (unsigned)(ax0[j] + kx) < (unsigned)p.Wi;
      va[j] = v;
      ra[j] = *reinterpret_cast<const f32x4*>(gin + (v ? abase[j] + toff : 0));
      if constexpr (MUL) rm[j] = *reinterpret_cast<const f32x4*>(gmul + (v ? mbase[j] + moff : 0));
      if constexpr (PER_N) {
        rsc[j] = *reinterpret_cast<const f32x4*>(gsc + (cv ? nbase[j] + c : 0));
        rsh[j] = *reinterpret_cast<const f32x4*>(gsh + (cv ? nbase[j] + c : 0));
      }
    }
    if constexpr (AFF && !PER_N) {
      rsc[0] = *reinterpret_cast<const f32x4*>(gsc + (cv ? c : 0));
      rsh[0] = *reinterpret_cast<const f32x4*>(gsh + (cv ? c : 0));
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const bool v = bval[j] && cv;
      vb[j] = v;
      rb[j] = *reinterpret_cast<const f32x4*>(gw + (v ? boff[j] + tap_ * Cin + c : 0));
    }
  };

  auto store_stage = [&](float* As, float* Bs) {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      f32x4 v = ra[j];
      if constexpr (MUL) v *= rm[j];
      if constexpr (AFF) {
        v = v * rsc[PER_N ? j : 0] + rsh[PER_N ? j : 0];
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      }
      v = va[j] ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(As + (lrow + 32 * j) * LDS_K + 4 * lseg) = v;
    }
#pragma unroll
    for (int j = 0; j < RB; ++j)
      *reinterpret_cast<f32x4*>(Bs + (lrow + 32 * j) * LDS_K + 4 * lseg) = vb[j] ? rb[j] : f32x4{0.f, 0.f, 0.f, 0.f};
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
      // keep the consumers of the prefetched registers behind the MFMAs: otherwise hipcc hoists the masking/affine
      // (and with it the s_waitcnt vmcnt) above the matrix work and the global-load latency is exposed every K step
      __builtin_amdgcn_sched_barrier(0);
      compute(As, Bs);
      __builtin_amdgcn_sched_barrier(0);
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

// Sum split-K partials, then the same epilogue as above.  Thread = one row x 4 columns (16-byte loads, 4 splits in
// flight); block = 32 column-quads x 8 rows, looping over 4 row groups (32 rows x 128 columns per block).
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int Cout,
                                                            const float* __restrict__ bias, int act,
                                                            float* __restrict__ out, int ld_out, double* stats, int rpg) {
  __shared__ float sred[128 * 2];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = blockIdx.y * 128 + tx * 4;
  const int r0 = blockIdx.x * 32;
  const int rlast = min(r0 + 32, M) - 1;
  const int g0 = rpg > 0 ? r0 / rpg : 0;
  const bool one_group = rpg <= 0 || (rlast / rpg) == g0;
  sred[threadIdx.x] = 0.f;
  __syncthreads();
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  const bool vec = (Cout & 3) == 0;
  const size_t zstride = (size_t)M * Cout;
  if (col < Cout) {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 4; ++k) if (bias && col + k < Cout) bv[k] = bias[col + k];
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + ty + 8 * i;
      if (row >= M) break;
      const float* src = ws + (size_t)row * Cout + col;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (vec) {
        f32x4 a0 = v, a1 = v, a2 = v, a3 = v;
        int z = 0;
        for (; z + 4 <= splits; z += 4) {
          a0 += *reinterpret_cast<const f32x4*>(src + (size_t)z * zstride);
          a1 += *reinterpret_cast<const f32x4*>(src + (size_t)(z + 1) * zstride);
          a2 += *reinterpret_cast<const f32x4*>(src + (size_t)(z + 2) * zstride);
          a3 += *reinterpret_cast<const f32x4*>(src + (size_t)(z + 3) * zstride);
        }
        for (; z < splits; ++z) a0 += *reinterpret_cast<const f32x4*>(src + (size_t)z * zstride);
        v = (a0 + a1) + (a2 + a3);
      } else {
        for (int k = 0; k < 4; ++k)
          if (col + k < Cout) for (int z = 0; z < splits; ++z) v[k] += src[(size_t)z * zstride + k];
      }
      for (int k = 0; k < 4; ++k) {
        if (col + k >= Cout) break;
        const float o = apply_act(v[k] + bv[k], act);
        out[(size_t)row * ld_out + col + k] = o;
        if (stats) {
          if (one_group) { s1[k] += o; s2[k] += o * o; }
          else {
            double* st = stats + ((size_t)(row / rpg) * Cout + col + k) * 2;
            atomicAdd(st, (double)o); atomicAdd(st + 1, (double)o * o);
          }
        }
      }
    }
  }
  if (stats && one_group) {
    for (int k = 0; k < 4; ++k) { atomicAdd(&sred[(tx * 4 + k) * 2], s1[k]); atomicAdd(&sred[(tx * 4 + k) * 2 + 1], s2[k]); }
    __syncthreads();
    if (threadIdx.x < 128 && blockIdx.y * 128 + threadIdx.x < Cout) {
      double* st = stats + ((size_t)g0 * Cout + blockIdx.y * 128 + threadIdx.x) * 2;
      atomicAdd(st, (double)sred[threadIdx.x * 2]);
      atomicAdd(st + 1, (double)sred[threadIdx.x * 2 + 1]);
    }
  }
}

template <int BM, int BN, int WGM, int WGN, int MODE>
int launch_mode(const G6dConv& d, int M, int T, int nChunks, int splits, hipStream_t stream) {
  const int total = T * nChunks;
  const int ips = (total + splits - 1) / splits;
  splits = (total + ips - 1) / ips;
  dim3 grid((M + BM - 1) / BM, (d.Cout + BN - 1) / BN, splits);
  const size_t lds_bytes = 2 * (size_t)(BM + BN) * LDS_K * sizeof(float);
  static bool attr_done = false;   // per template instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WGM, WGN, MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN, MODE>), grid, dim3(256), lds_bytes, stream, d, M, T, nChunks,
                     ips, total, splits);
  int rc = g6d_check_launch("conv_igemm");
  if (rc != G6D_OK) return rc;
  if (splits > 1) {
    dim3 g2((M + 31) / 32, (d.Cout + 127) / 128);
    hipLaunchKernelGGL(splitk_reduce_kernel, g2, dim3(256), 0, stream, d.workspace, splits, M, d.Cout, d.bias, d.out_act,
                       d.out, d.ld_out, d.stats, d.stat_rows_per_group);
    rc = g6d_check_launch("splitk_reduce");
  }
  return rc;
}

template <int BM, int BN, int WGM, int WGN>
int launch_cfg(const G6dConv& d, int M, int T, int nChunks, int splits, hipStream_t stream) {
  if (d.mul) return launch_mode<BM, BN, WGM, WGN, 3>(d, M, T, nChunks, splits, stream);
  if (!d.in_scale) return launch_mode<BM, BN, WGM, WGN, 0>(d, M, T, nChunks, splits, stream);
  if (d.in_affine_per_n) return launch_mode<BM, BN, WGM, WGN, 2>(d, M, T, nChunks, splits, stream);
  return launch_mode<BM, BN, WGM, WGN, 1>(d, M, T, nChunks, splits, stream);
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
  if (d.mul && !d.in_scale) { g6d_set_error("conv: mul requires in_scale/in_shift"); return G6D_EINVAL; }
  if ((long long)d.N * d.Di * d.Hi * d.Wi * d.ld_in >= (1ll << 31) || (long long)d.Cout * d.kd * d.kh * d.kw * d.Cin >= (1ll << 31)) {
    g6d_set_error("conv: tensor exceeds 2^31 elements"); return G6D_EINVAL;
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
