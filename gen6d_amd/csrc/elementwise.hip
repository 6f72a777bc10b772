// HBM-bound glue kernels: InstanceNorm finalisation, affine/ReLU/pool, bilinear up-sampling, layout change with
// L2 normalisation, token-wise tails of the selector.  All channels-last, 16-byte accesses along C.
#include "g6d_common.h"

namespace {

__global__ void stats_finalize_kernel(const double* __restrict__ st, int n, double inv_count, double eps,
                                      float* __restrict__ scale, float* __restrict__ shift) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double mean = st[2 * i] * inv_count;
  double var = st[2 * i + 1] * inv_count - mean * mean;
  if (var < 0) var = 0;
  double rs = 1.0 / sqrt(var + eps);
  scale[i] = (float)rs;
  shift[i] = (float)(-mean * rs);
}

__device__ __forceinline__ f32x4 aff(f32x4 v, f32x4 sc, f32x4 sh, bool has, int relu) {
  if (has) v = v * sc + sh;
  if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
  return v;
}

// pool: 0 none, 1 = 2x2 max
__global__ void affine_act_pool_kernel(const float* __restrict__ in, int ld_in, const float* __restrict__ scale,
                                       const float* __restrict__ shift, int per_n, int relu, int pool, int N, int H,
                                       int W, int C, float* __restrict__ out, int ld_out) {
  const int C4 = C >> 2;
  const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
  const long long total = (long long)N * Ho * Wo * C4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C4) * 4; long long t = i / C4;
    int x = (int)(t % Wo); t /= Wo; int y = (int)(t % Ho); int n = (int)(t / Ho);
    const bool has = scale != nullptr;
    f32x4 sc = {1, 1, 1, 1}, sh = {0, 0, 0, 0};
    if (has) {
      const size_t o = (size_t)(per_n ? n / per_n : 0) * C + c;
      sc = *reinterpret_cast<const f32x4*>(scale + o); sh = *reinterpret_cast<const f32x4*>(shift + o);
    }
    f32x4 v;
    if (pool) {
      const float* b = in + (((size_t)n * H + 2 * y) * W + 2 * x) * ld_in + c;
      f32x4 v00 = aff(*reinterpret_cast<const f32x4*>(b), sc, sh, has, relu);
      f32x4 v01 = aff(*reinterpret_cast<const f32x4*>(b + ld_in), sc, sh, has, relu);
      f32x4 v10 = aff(*reinterpret_cast<const f32x4*>(b + (size_t)W * ld_in), sc, sh, has, relu);
      f32x4 v11 = aff(*reinterpret_cast<const f32x4*>(b + (size_t)W * ld_in + ld_in), sc, sh, has, relu);
      for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaxf(v00[k], v01[k]), fmaxf(v10[k], v11[k]));
    } else {
      v = aff(*reinterpret_cast<const f32x4*>(in + (((size_t)n * H + y) * W + x) * ld_in + c), sc, sh, has, relu);
    }
    *reinterpret_cast<f32x4*>(out + (((size_t)n * Ho + y) * Wo + x) * ld_out + c) = v;
  }
}

// pool == 2: mean over the whole HxW window -> [N][C]
__global__ void affine_act_avg_kernel(const float* __restrict__ in, int ld_in, const float* __restrict__ scale,
                                      const float* __restrict__ shift, int per_n, int relu, int N, int HW, int C,
                                      float* __restrict__ out, int ld_out) {
  const int C4 = C >> 2;
  const long long total = (long long)N * C4;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c = (int)(i % C4) * 4; int n = (int)(i / C4);
  const bool has = scale != nullptr;
  f32x4 sc = {1, 1, 1, 1}, sh = {0, 0, 0, 0};
  if (has) {
    const size_t o = (size_t)(per_n ? n / per_n : 0) * C + c;
    sc = *reinterpret_cast<const f32x4*>(scale + o); sh = *reinterpret_cast<const f32x4*>(shift + o);
  }
  f32x4 s = {0, 0, 0, 0};
  for (int p = 0; p < HW; ++p) s += aff(*reinterpret_cast<const f32x4*>(in + ((size_t)n * HW + p) * ld_in + c), sc, sh, has, relu);
  const float inv = 1.f / (float)HW;
  *reinterpret_cast<f32x4*>(out + (size_t)n * ld_out + c) = s * inv;
}

__global__ void upsample_bilinear_kernel(const float* __restrict__ in, int ld_in, const float* __restrict__ scale,
                                         const float* __restrict__ shift, int per_n, int N, int H, int W, int C, int f,
                                         float* __restrict__ out, int ld_out) {
  const int C4 = C >> 2, Ho = H * f, Wo = W * f;
  const long long total = (long long)N * Ho * Wo * C4;
  const float rs = 1.f / (float)f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C4) * 4; long long t = i / C4;
    int x = (int)(t % Wo); t /= Wo; int y = (int)(t % Ho); int n = (int)(t / Ho);
    float sy = rs * (y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
    float sx = rs * (x + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
    int y0 = (int)sy, x0 = (int)sx;
    int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const bool has = scale != nullptr;
    f32x4 sc = {1, 1, 1, 1}, sh = {0, 0, 0, 0};
    if (has) {
      const size_t o = (size_t)(per_n ? n / per_n : 0) * C + c;
      sc = *reinterpret_cast<const f32x4*>(scale + o); sh = *reinterpret_cast<const f32x4*>(shift + o);
    }
    const float* b = in + (size_t)n * H * W * ld_in + c;
    f32x4 v00 = aff(*reinterpret_cast<const f32x4*>(b + ((size_t)y0 * W + x0) * ld_in), sc, sh, has, 0);
    f32x4 v01 = aff(*reinterpret_cast<const f32x4*>(b + ((size_t)y0 * W + x1) * ld_in), sc, sh, has, 0);
    f32x4 v10 = aff(*reinterpret_cast<const f32x4*>(b + ((size_t)y1 * W + x0) * ld_in), sc, sh, has, 0);
    f32x4 v11 = aff(*reinterpret_cast<const f32x4*>(b + ((size_t)y1 * W + x1) * ld_in), sc, sh, has, 0);
    f32x4 v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    *reinterpret_cast<f32x4*>(out + (((size_t)n * Ho + y) * Wo + x) * ld_out + c) = v;
  }
}

// Trunk glue: y = maxpool2x2?( relu?( x + bias[c] ) ) on NCHW (the MIOpen convolutions of the VGG trunk run without
// bias; this replaces PyTorch's separate bias-add, clamp and max_pool2d passes with one).  VEC outputs per thread
// (4 with 16-byte accesses when the output width allows it, else 1).
template <int VEC>
__global__ void bias_relu_pool_nchw_kernel(const float* __restrict__ in, const float* __restrict__ bias, int C, int H,
                                           int W, int relu, int pool, float* __restrict__ out, long long total) {
  const int Wo = pool ? W / 2 : W, Ho = pool ? H / 2 : H;
  for (long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * VEC; i < total;
       i += (long long)gridDim.x * blockDim.x * VEC) {
    const int x = (int)(i % Wo); long long t = i / Wo;
    const int y = (int)(t % Ho); t /= Ho;
    const float b = bias[(int)(t % C)];
    const float* src = in + t * (long long)H * W;
    float v[VEC];
    if (pool) {
      const float* r0 = src + (size_t)(2 * y) * W + 2 * x;
      if constexpr (VEC == 4) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(r0), a1 = *reinterpret_cast<const f32x4*>(r0 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(r0 + W), b1 = *reinterpret_cast<const f32x4*>(r0 + W + 4);
        v[0] = fmaxf(fmaxf(a0[0], a0[1]), fmaxf(b0[0], b0[1])); v[1] = fmaxf(fmaxf(a0[2], a0[3]), fmaxf(b0[2], b0[3]));
        v[2] = fmaxf(fmaxf(a1[0], a1[1]), fmaxf(b1[0], b1[1])); v[3] = fmaxf(fmaxf(a1[2], a1[3]), fmaxf(b1[2], b1[3]));
      } else {
        v[0] = fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[W], r0[W + 1]));
      }
    } else {
      if constexpr (VEC == 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + (size_t)y * W + x);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
      } else {
        v[0] = src[(size_t)y * W + x];
      }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {                    // max(a,b)+c == max(a+c,b+c); relu(max) == max(relu)
      v[k] += b;
      if (relu) v[k] = fmaxf(v[k], 0.f);
    }
    if constexpr (VEC == 4) *reinterpret_cast<f32x4*>(out + i) = f32x4{v[0], v[1], v[2], v[3]};
    else out[i] = v[0];
  }
}

// In-place L2 normalisation of channels-last rows (F.normalize over C, eps 1e-12): one wave per position, 16-byte
// accesses along C.  Runs after the layout change, where the C axis is contiguous.
__global__ void __launch_bounds__(256) l2norm_rows_kernel(float* __restrict__ x, int rows, int C, int ld) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* p = x + (size_t)row * ld;
  float s = 0.f;
  if ((ld & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) {      // rows not 16-byte aligned (e.g. the [qn][7] regressor rows): scalar accesses
    for (int c = lane; c < C; c += 64) s += p[c] * p[c];
    const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int c = lane; c < C; c += 64) p[c] *= inv;
    return;
  }
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
  for (int c = lane * 4; c < C; c += 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
    *reinterpret_cast<f32x4*>(p + c) = v * inv;
  }
}

// LDS-tiled transpose: block = (64 positions x 64 channels) tile; reads coalesced along HW, writes along C.
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ in, int C, int HW,
                                                           float* __restrict__ out, int ld_out) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* src = in + (size_t)n * C * HW;
  for (int k = ty; k < 64; k += 4) {
    const int c = c0 + k, p = p0 + tx;
    tile[k][tx] = (c < C && p < HW) ? src[(size_t)c * HW + p] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 64; k += 4) {           // k = position, tx = channel
    const int p = p0 + k, c = c0 + tx;
    if (p < HW && c < C) {
      out[((size_t)n * HW + p) * ld_out + c] = tile[tx][k];
    }
  }
}

// InstanceNorm2d(3) of vps[ch][D] over D (biased var, eps 1e-5); one block per (channel, query of the batch): vps [batch][3][D],
// feats [batch * D][ld].
__global__ void __launch_bounds__(256) vps_norm_kernel(const float* __restrict__ vps, int D, float* __restrict__ feats,
                                                       int ld, int c_off) {
  __shared__ double red[2][4];
  const int ch = blockIdx.x;
  const float* v = vps + ((size_t)blockIdx.y * 3 + ch) * D;
  feats += (size_t)blockIdx.y * D * ld;
  double s1 = 0, s2 = 0;
  for (int i = threadIdx.x; i < D; i += 256) { double x = v[i]; s1 += x; s2 += x * x; }
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  double mean = s1 / D, var = s2 / D - mean * mean;
  if (var < 0) var = 0;
  float rs = (float)(1.0 / sqrt(var + 1e-5)), mu = (float)mean;
  for (int i = threadIdx.x; i < D; i += 256) feats[(size_t)i * ld + c_off + ch] = (v[i] - mu) * rs;
}

__global__ void max_an_add_kernel(const float* __restrict__ in, int ld_in, int rfn, int an, int C,
                                  const float* __restrict__ embed, float* __restrict__ out, int ld_out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rfn * C) return;
  int c = i % C, r = i / C;
  in += (size_t)blockIdx.y * rfn * an * ld_in; out += (size_t)blockIdx.y * rfn * ld_out;      // query of the batch; embed is shared
  float m = in[(size_t)(r * an) * ld_in + c];
  for (int a = 1; a < an; ++a) m = fmaxf(m, in[(size_t)(r * an + a) * ld_in + c]);
  out[(size_t)r * ld_out + c] = m + embed[(size_t)r * C + c];
}

// one wave per token
__global__ void __launch_bounds__(64) layernorm_kernel(const float* __restrict__ in, int ld_in, int C,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, float* __restrict__ out, int ld_out) {
  const int t = blockIdx.x, lane = threadIdx.x;
  const float* x = in + (size_t)t * ld_in;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[c];
  const float mean = wave_sum(s) / C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { float d = x[c] - mean; q += d * d; }
  const float rs = 1.f / sqrtf(wave_sum(q) / C + eps);
  for (int c = lane; c < C; c += 64) out[(size_t)t * ld_out + c] = (x[c] - mean) * rs * gamma[c] + beta[c];
}

__global__ void affine_act_add_kernel(const float* __restrict__ in, int ld_in, const float* __restrict__ scale,
                                      const float* __restrict__ shift, int relu, const float* __restrict__ res,
                                      int ld_res, int n, int C, float* __restrict__ out, int ld_out, int rows_per_group) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * C) return;
  int c = i % C, r = i / C;
  float v = in[(size_t)r * ld_in + c];
  const int g = rows_per_group > 0 ? r / rows_per_group : 0;          // one affine table per run of rows (query of the batch)
  if (scale) v = v * scale[(size_t)g * C + c] + shift[(size_t)g * C + c];
  if (relu) v = fmaxf(v, 0.f);
  if (res) v += res[(size_t)r * ld_res + c];
  out[(size_t)r * ld_out + c] = v;
}

// 8-head attention over n tokens, one wave per (head, query token). Channel c = d*heads + head.
__global__ void __launch_bounds__(64) attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, int ld, int n, int C, int heads,
                                                       float* __restrict__ out, int ld_out) {
  extern __shared__ float sm[];           // [dh] query + [n] probabilities
  const int dh = C / heads;
  float* qs = sm; float* pr = sm + dh;
  const int h = blockIdx.x % heads, i = blockIdx.x / heads, lane = threadIdx.x;
  // blockIdx.y = query of the batch: attention runs among the n tokens of one query
  q += (size_t)blockIdx.y * n * ld; k += (size_t)blockIdx.y * n * ld; v += (size_t)blockIdx.y * n * ld; out += (size_t)blockIdx.y * n * ld_out;
  for (int d = lane; d < dh; d += 64) qs[d] = q[(size_t)i * ld + d * heads + h];
  __syncthreads();
  const float inv = 1.f / sqrtf((float)dh);
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) {
    float s = 0.f;
    const float* kj = k + (size_t)j * ld + h;
    for (int d = 0; d < dh; ++d) s += qs[d] * kj[d * heads];
    s *= inv; pr[j] = s; mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) { float e = expf(pr[j] - mx); pr[j] = e; sum += e; }
  sum = wave_sum(sum);
  __syncthreads();
  const float rsum = 1.f / sum;
  for (int d = lane; d < dh; d += 64) {
    float acc = 0.f;
    for (int j = 0; j < n; ++j) acc += pr[j] * v[(size_t)j * ld + d * heads + h];
    out[(size_t)i * ld_out + d * heads + h] = acc * rsum;
  }
}


// The same attention with one 256-thread block per (query of the batch, head) for n <= 64 tokens and dh <= 64 (the selector: 64
// references, 8 heads of 64): Q, K, V of the head are gathered into LDS ONCE (the wave-per-token kernel above re-reads K and V for each
// of the n tokens with 32-byte strides: 114 us per launch at 16 queries, 0.35 % of a step), scores / softmax / P V run out of LDS.
//   thread (i = tid >> 2, c = tid & 3): token i, score columns / output channels c, c + 4, ... (16 each)
#define ATT_N 64
#define ATT_LD (ATT_N + 1)
__global__ void __launch_bounds__(256) attention_block_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, int ld, int n, int dh, int heads,
                                                              float* __restrict__ out, int ld_out) {
  __shared__ float Q[ATT_N][ATT_LD], K[ATT_N][ATT_LD], V[ATT_N][ATT_LD], P[ATT_N][ATT_LD];
  const int h = blockIdx.x, tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.y * n * ld;
  for (int e = tid; e < ATT_N * ATT_N; e += 256) {
    const int j = e / ATT_N, d = e - j * ATT_N;
    const bool ok = j < n && d < dh;
    const size_t o = base + (size_t)j * ld + d * heads + h;
    Q[j][d] = ok ? q[o] : 0.f; K[j][d] = ok ? k[o] : 0.f; V[j][d] = ok ? v[o] : 0.f;
  }
  __syncthreads();
  const int i = tid >> 2, c = tid & 3;
  const float inv = 1.f / sqrtf((float)dh);
  float sc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) sc[t] = 0.f;
  for (int d = 0; d < dh; ++d) {
    const float qv = Q[i][d];
#pragma unroll
    for (int t = 0; t < 16; ++t) sc[t] += qv * K[c + 4 * t][d];
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < 16; ++t) { sc[t] = c + 4 * t < n ? sc[t] * inv : -INFINITY; mx = fmaxf(mx, sc[t]); }
  mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) { const float e = c + 4 * t < n ? expf(sc[t] - mx) : 0.f; P[i][c + 4 * t] = e; sum += e; }
  sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64);
  __syncthreads();
  float ov[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) ov[t] = 0.f;
  for (int j = 0; j < n; ++j) {
    const float pv = P[i][j];
#pragma unroll
    for (int t = 0; t < 16; ++t) ov[t] += pv * V[j][c + 4 * t];
  }
  if (i < n) {
    const float rsum = 1.f / sum;
    float* o = out + ((size_t)blockIdx.y * n + i) * ld_out + h;
#pragma unroll
    for (int t = 0; t < 16; ++t)
      if (c + 4 * t < dh) o[(c + 4 * t) * heads] = ov[t] * rsum;
  }
}

// Weight-streaming GEMV: one block of 512 threads per output row, B <= 8 right-hand sides.  HBM-bound (the refiner's
// first FC layer reads 67 MB of weights for 33 MFLOP): every thread keeps four independent 16-byte weight loads in
// flight so that two resident blocks per CU cover the HBM latency-bandwidth product (~64 KB per CU).
#define GEMV_THREADS 512
// out[b][o] = act(bias[o] + sum_k x[b][k] * W[o][k]), B <= 8 rows of x (the refiner's regressor: B = 1, K = 32768 -> 512 is a
// 64 MB weight stream).  One block per output row; UNROLL 16-byte weight loads per thread are requested before the first is
// used (16 at K >= 32768: the whole row is in flight at once, one memory latency per block instead of four), streamed past
// the caches (read once); x is re-read by every block and stays in L2.
template <int UNROLL>
__global__ void __launch_bounds__(GEMV_THREADS) linear_gemv_kernel(const float* __restrict__ x, int B, int K,
                                                                   const float* __restrict__ W,
                                                                   const float* __restrict__ bias, int act,
                                                                   float* __restrict__ out, int O) {
  __shared__ float red[8][GEMV_THREADS / 64];
  const int o = blockIdx.x;
  const float* w = W + (size_t)o * K;
  float acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) acc[b] = 0.f;
  constexpr int STEP = GEMV_THREADS * 4;
  for (int k0 = threadIdx.x * 4; k0 < K; k0 += STEP * UNROLL) {
    f32x4 wv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int kk = k0 + u * STEP;
      wv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w + (kk < K ? kk : 0)));
    }
    if (B == 1) {
      f32x4 xv[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) { const int kk = k0 + u * STEP; xv[u] = *reinterpret_cast<const f32x4*>(x + (kk < K ? kk : 0)); }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (k0 + u * STEP < K) acc[0] += wv[u][0] * xv[u][0] + wv[u][1] * xv[u][1] + wv[u][2] * xv[u][2] + wv[u][3] * xv[u][3];
    } else {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int kk = k0 + u * STEP;
        if (kk >= K) continue;
#pragma unroll
        for (int b = 0; b < 8; ++b)
          if (b < B) {
            f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)b * K + kk);
            acc[b] += wv[u][0] * xv[0] + wv[u][1] * xv[1] + wv[u][2] * xv[2] + wv[u][3] * xv[3];
          }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    float s = wave_sum(acc[b]);
    if ((threadIdx.x & 63) == 0) red[b][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x < B) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < GEMV_THREADS / 64; ++i) s += red[threadIdx.x][i];
    if (bias) s += bias[o];
    out[(size_t)threadIdx.x * O + o] = apply_act(s, act);
  }
}

// Batched weight-streaming GEMV for 2 <= B <= 8 right-hand sides and long rows (the refiner's first FC layer with a batch of queries:
// K = 32768, O = 512, 67 MB of weights for B x 33 MFLOP).  linear_gemv_kernel gives every block ONE output row and lets it read all B
// rows of x: 512 blocks x 1 MB of x through L2 for 67 MB of HBM traffic — measured 49 us at B = 8 (0.17 of 8 TB/s) against 14.8 us
// at B = 1.  Here a block owns R = 8 output rows x one K slice of 1024 KV floats: a thread requests its KV 16-byte pieces of all B rows
// of x (L2) and then of all R weight rows (HBM, streamed past the caches) BEFORE it uses any of them — every load of the block is in
// flight at once (a first version that fetched the x pieces inside the FMA loop paid one exposed L2 round trip per piece: 41 us) —
// and runs R x B x 4 FMAs per x piece.  x through L2: O / R x B x K x 4 bytes = the size of the weight stream at B = 8.  The
// R x 8 partial sums of a wave are reduced with the transposing butterfly (63 shuffles instead of 64 x 6), the K slices of a row
// group meet through the workspace like the split launches of the conv kernels (g6d_split_arrive): partial sums written through, the
// block that arrives last adds them IN SLICE ORDER (deterministic) and applies bias / activation.
#define GEMVB_THREADS 256
#define GEMVB_R 8
// NH (round 5): up to NH groups of 8 right-hand sides against ONE pass over the weights — the block keeps its 8 x KL weight values in
// registers and runs the x rows of every group past them (group h + 1's rows are requested while group h is reduced); the partial sums
// of a group land in their own slab of the workspace, the last block of a row group adds and finishes all of them.
template <int KV, int NH>
__global__ void __launch_bounds__(GEMVB_THREADS, 2) linear_gemv_batch_kernel(const float* __restrict__ x, int B, int K, const float* __restrict__ W,
                                                                              const float* __restrict__ bias, int act, float* __restrict__ out,
                                                                              int O, float* __restrict__ ws) {
  constexpr int R = GEMVB_R, KL = GEMVB_THREADS * 4 * KV;
  __shared__ float red[GEMVB_THREADS / 64][R * 8];
  __shared__ int flag;
  const int og = blockIdx.x, ks = blockIdx.y, KS = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kbase = ks * KL + tid * 4;
  f32x4 xr[8][KV];
  auto load_x = [&](int h) {
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int v = 0; v < KV; ++v)
        xr[b][v] = 8 * h + b < B ? *reinterpret_cast<const f32x4*>(x + (size_t)(8 * h + b) * K + kbase + v * (GEMVB_THREADS * 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  load_x(0);
  f32x4 wv[R][KV];
#pragma unroll
  for (int v = 0; v < KV; ++v)
#pragma unroll
    for (int r = 0; r < R; ++r)
      wv[r][v] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + (size_t)(og * R + r) * K + kbase + v * (GEMVB_THREADS * 4)));
  int idx = 0;
#pragma unroll
  for (int sft = 0; sft < 6; ++sft) idx |= ((lane >> (5 - sft)) & 1) << sft;
  float part[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
  if (8 * h >= B) { part[h] = 0.f; continue; }
  float acc[R * 8];                       // value index i = r * 8 + b
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      float a = 0.f;
#pragma unroll
      for (int v = 0; v < KV; ++v)
        a += wv[r][v][0] * xr[b][v][0] + wv[r][v][1] * xr[b][v][1] + wv[r][v][2] * xr[b][v][2] + wv[r][v][3] * xr[b][v][3];
      acc[r * 8 + b] = a;
    }
  if (h + 1 < NH && 8 * (h + 1) < B) load_x(h + 1);      // the next group's rows travel while this one is reduced
  // transposing butterfly: after the step with lane bit `bit` every lane holds half as many values, each summed over that bit; at the
  // end lane L holds the wave's sum of value index bitrev6(L)
  int bit = 32;
#pragma unroll
  for (int nv = R * 8; nv > 1; nv >>= 1, bit >>= 1) {
    const bool hi = (lane & bit) != 0;
#pragma unroll
    for (int i = 0; i < nv / 2; ++i) {
      const float keep = hi ? acc[2 * i + 1] : acc[2 * i], send = hi ? acc[2 * i] : acc[2 * i + 1];
      acc[i] = keep + __shfl_xor(send, bit, 64);
    }
  }
  if (h > 0) __syncthreads();                            // the previous group's sums have been read
  red[wave][idx] = acc[0];
  __syncthreads();
  float pt = 0.f;
  if (tid < R * 8) {
#pragma unroll
    for (int w = 0; w < GEMVB_THREADS / 64; ++w) pt += red[w][tid];
  }
  part[h] = pt;
  }
  if (KS > 1) {
    float* slab = ws + G6D_WS_COUNTERS + (size_t)og * NH * KS * (R * 8);       // [group h][slice][R * 8]
#pragma unroll
    for (int h = 0; h < NH; ++h)
      if (tid < R * 8 && 8 * h < B) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(slab + (h * KS + ks) * (R * 8) + tid), "v"(part[h]) : "memory");
    if (!g6d_split_arrive(reinterpret_cast<int*>(ws) + og, KS, &flag)) return;
    if (tid < R * 8) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        if (8 * h >= B) continue;
        float pt = 0.f;
        for (int z = 0; z < KS; ++z) pt += slab[(h * KS + z) * (R * 8) + tid];
        part[h] = pt;
      }
    }
  }
  if (tid < R * 8) {
    const int r = tid >> 3, b = tid & 7, o = og * R + r;
#pragma unroll
    for (int h = 0; h < NH; ++h)
      if (8 * h + b < B) out[(size_t)(8 * h + b) * O + o] = apply_act(part[h] + (bias ? bias[o] : 0.f), act);
  }
}


// ---- the same layer on the matrix cores (round 5): with 2..32 right-hand sides the 8-rows-x-K-slice kernel above is bound by its
// vector-ALU work (512 FMAs + a 63-step butterfly per thread and group of 8 rows: 39 us for 16 right-hand sides against a 13 us weight
// stream).  As a GEMM it is M = O weight rows, N = right-hand sides (padded to 32), K = 32768: v_mfma_f32_32x32x2_f32 consumes 64 weight
// values per instruction — 7.8 us of matrix-pipe time for the 67 MB layer, so the stream from HBM is what is left.
//   block  = 4 waves x (32 weight rows, K range KW): wave w of block (rg, ns) owns k in [(4 ns + w) KW, +KW) of rows 32 rg .. 32 rg + 31
//   lane   = (i = lane & 31, h = lane >> 5): one 16-byte load of W[32 rg + i][k + 4h .. +3] (non-temporal: streamed once) and one of
//            x[i][k + 4h .. +3] (L2; zeros for i >= B) feed FOUR MFMAs (MFMA s consumes k + s and k + 4 + s on both sides: the same
//            permutation of K for both operands, as conv_igemm.hip); U = 8 such pairs are requested before the first is used
//   finish = the four waves add their accumulators through LDS, the NS blocks of a row group meet through the workspace (partials written
//            through, last block adds them in slice order: bit-equal from run to run), bias / activation, out[b][o] for b < B.
#define GEMVM_KW 256
template <int U>
__global__ void __launch_bounds__(256, 2) linear_gemv_mfma_kernel(const float* __restrict__ x, int B, int K, const float* __restrict__ W,
                                                                  const float* __restrict__ bias, int act, float* __restrict__ out, int O,
                                                                  float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) float red[3][16][64];
  __shared__ int flag;
  const int rg = blockIdx.x, ns = blockIdx.y, NS = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const size_t k0 = (size_t)(4 * ns + wave) * GEMVM_KW + 4 * h;
  const float* wp = W + (size_t)(32 * rg + i) * K + k0;
  const bool xv = i < B;
  const float* xp = x + (size_t)(xv ? i : 0) * K + k0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 1
  for (int k = 0; k < GEMVM_KW; k += 8 * U) {
    f32x4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = *reinterpret_cast<const f32x4*>(wp + k + 8 * u);      // (plain: the four loads that share a 128-byte line of a row meet in L1)
#pragma unroll
    for (int u = 0; u < U; ++u) b[u] = xv ? *reinterpret_cast<const f32x4*>(xp + k + 8 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][s_], b[u][s_], acc, 0, 0, 0);
  }
  // waves 1..3 -> LDS, wave 0 adds them in wave order
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += red[w][r][lane];
  }
  if (NS > 1) {
    float* slab = ws + G6D_WS_COUNTERS + (size_t)rg * NS * 1024;            // [slice][16 registers as 4 x 16 bytes][64 lanes]
    if (wave == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        g6d_store_wt(slab + (size_t)ns * 1024 + (q * 64 + lane) * 4, f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
    }
    if (!g6d_split_arrive(reinterpret_cast<int*>(ws) + rg, NS, &flag)) return;
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      for (int z = 0; z < NS; ++z) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(slab + (size_t)z * 1024 + (q * 64 + lane) * 4);
          acc[4 * q] += v[0]; acc[4 * q + 1] += v[1]; acc[4 * q + 2] += v[2]; acc[4 * q + 3] += v[3];
        }
      }
    }
  }
  if (wave == 0 && i < B) {
    // accumulator register r of lane (i, h): weight row (r & 3) + 8 (r >> 2) + 4 h of the group, right-hand side i
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = 32 * rg + (r & 3) + 8 * (r >> 2) + 4 * h;
      out[(size_t)i * O + o] = apply_act(acc[r] + (bias ? bias[o] : 0.f), act);
    }
  }
}

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  return (int)(g > 65535 * 16 ? 65535 * 16 : (g < 1 ? 1 : g));
}

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int g6d_stats_finalize(const double* stats, int n, double count, double eps, float* scale, float* shift,
                                  g6d_stream_t stream) {
  if (!stats || !scale || !shift || n <= 0 || count <= 0) { g6d_set_error("stats_finalize: bad args"); return G6D_EINVAL; }
  hipLaunchKernelGGL(stats_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, STREAM(stream), stats, n, 1.0 / count,
                     eps, scale, shift);
  return g6d_check_launch("stats_finalize");
}

extern "C" int g6d_affine_act_pool(const float* in, int ld_in, const float* scale, const float* shift, int per_n,
                                   int relu, int pool, int N, int H, int W, int C, float* out, int ld_out,
                                   g6d_stream_t stream) {
  if (!in || !out || (C & 3) || (ld_in & 3) || (ld_out & 3) || N <= 0 || H <= 0 || W <= 0 || pool < 0 || pool > 2 ||
      (pool == 1 && ((H | W) & 1)) || !g6d_aligned16(in) || !g6d_aligned16(out) || (scale && (!shift || !g6d_aligned16(scale) || !g6d_aligned16(shift)))) {
    g6d_set_error("affine_act_pool: bad args"); return G6D_EINVAL;
  }
  if (pool == 2) {
    long long total = (long long)N * (C / 4);
    hipLaunchKernelGGL(affine_act_avg_kernel, dim3(grid_for(total, 256)), dim3(256), 0, STREAM(stream), in, ld_in, scale,
                       shift, per_n, relu, N, H * W, C, out, ld_out);
  } else {
    long long total = (long long)N * (pool ? H / 2 : H) * (pool ? W / 2 : W) * (C / 4);
    hipLaunchKernelGGL(affine_act_pool_kernel, dim3(grid_for(total, 256)), dim3(256), 0, STREAM(stream), in, ld_in, scale,
                       shift, per_n, relu, pool, N, H, W, C, out, ld_out);
  }
  return g6d_check_launch("affine_act_pool");
}

extern "C" int g6d_upsample_bilinear(const float* in, int ld_in, const float* scale, const float* shift, int per_n,
                                     int N, int H, int W, int C, int factor, float* out, int ld_out, g6d_stream_t stream) {
  if (!in || !out || (C & 3) || (ld_in & 3) || (ld_out & 3) || factor < 1 || N <= 0 || !g6d_aligned16(in) ||
      !g6d_aligned16(out) || (scale && (!shift || !g6d_aligned16(scale) || !g6d_aligned16(shift)))) {
    g6d_set_error("upsample_bilinear: bad args"); return G6D_EINVAL;
  }
  long long total = (long long)N * H * factor * W * factor * (C / 4);
  hipLaunchKernelGGL(upsample_bilinear_kernel, dim3(grid_for(total, 256)), dim3(256), 0, STREAM(stream), in, ld_in, scale,
                     shift, per_n, N, H, W, C, factor, out, ld_out);
  return g6d_check_launch("upsample_bilinear");
}

extern "C" int g6d_bias_relu_pool_nchw(const float* in, const float* bias, int N, int C, int H, int W, int relu, int pool,
                                       float* out, g6d_stream_t stream) {
  if (!in || !bias || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (pool && (H < 2 || W < 2))) {   // odd sizes pool with floor, as F.max_pool2d
    g6d_set_error("bias_relu_pool_nchw: bad args"); return G6D_EINVAL;
  }
  const int Wo = pool ? W / 2 : W;
  const long long total = (long long)N * C * (pool ? H / 2 : H) * Wo;
  // 16-byte path: every row start (input and output) must stay 16-byte aligned
  const bool vec = (Wo & 3) == 0 && (W & 3) == 0 && g6d_aligned16(in) && g6d_aligned16(out);
  if (vec)
    hipLaunchKernelGGL(bias_relu_pool_nchw_kernel<4>, dim3(grid_for(total / 4, 256)), dim3(256), 0, STREAM(stream), in, bias,
                       C, H, W, relu, pool, out, total);
  else
    hipLaunchKernelGGL(bias_relu_pool_nchw_kernel<1>, dim3(grid_for(total, 256)), dim3(256), 0, STREAM(stream), in, bias, C,
                       H, W, relu, pool, out, total);
  return g6d_check_launch("bias_relu_pool_nchw");
}

extern "C" int g6d_nchw_to_nhwc(const float* in, int N, int C, int H, int W, int l2norm, float* out, int ld_out,
                                g6d_stream_t stream) {
  if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || ld_out < C ||
      (l2norm && ((C & 3) || (ld_out & 3) || !g6d_aligned16(out)))) {
    g6d_set_error("nchw_to_nhwc: bad args (l2norm needs C and ld_out multiples of 4)"); return G6D_EINVAL;
  }
  const int HW = H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((HW + 63) / 64, (C + 63) / 64, N), dim3(256), 0, STREAM(stream), in, C, HW,
                     out, ld_out);
  int rc = g6d_check_launch("nchw_to_nhwc");
  if (rc != G6D_OK || !l2norm) return rc;
  const int rows = N * HW;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, STREAM(stream), out, rows, C, ld_out);
  return g6d_check_launch("l2norm_rows");
}

extern "C" int g6d_l2norm_rows(float* x, int rows, int C, int ld, g6d_stream_t stream) {
  if (!x || rows <= 0 || C <= 0 || ld < C || (!(ld & 3) && g6d_aligned16(x) && (C & 3))) {
    g6d_set_error("l2norm_rows: bad args (C a multiple of 4 on the 16-byte aligned path)"); return G6D_EINVAL;
  }
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, STREAM(stream), x, rows, C, ld);
  return g6d_check_launch("l2norm_rows");
}

extern "C" int g6d_vps_norm(const float* vps, int D, float* feats, int ld, int c_off, int batch, g6d_stream_t stream) {
  if (!vps || !feats || D <= 0 || batch < 1 || batch > 65535) { g6d_set_error("vps_norm: bad args"); return G6D_EINVAL; }
  hipLaunchKernelGGL(vps_norm_kernel, dim3(3, batch), dim3(256), 0, STREAM(stream), vps, D, feats, ld, c_off);
  return g6d_check_launch("vps_norm");
}

extern "C" int g6d_max_an_add(const float* in, int ld_in, int rfn, int an, int C, const float* embed, float* out,
                              int ld_out, int batch, g6d_stream_t stream) {
  if (!in || !embed || !out || rfn <= 0 || an <= 0 || C <= 0 || batch < 1 || batch > 65535) { g6d_set_error("max_an_add: bad args"); return G6D_EINVAL; }
  hipLaunchKernelGGL(max_an_add_kernel, dim3((rfn * C + 255) / 256, batch), dim3(256), 0, STREAM(stream), in, ld_in, rfn, an, C,
                     embed, out, ld_out);
  return g6d_check_launch("max_an_add");
}

extern "C" int g6d_attention(const float* q, const float* k, const float* v, int ld, int n, int C, int heads, float* out,
                             int ld_out, int batch, g6d_stream_t stream) {
  if (!q || !k || !v || !out || n <= 0 || heads <= 0 || C % heads || batch < 1 || batch > 65535) { g6d_set_error("attention: bad args"); return G6D_EINVAL; }
  size_t lds = (size_t)(C / heads + n) * sizeof(float);
  if (lds > 60000) { g6d_set_error("attention: n too large"); return G6D_EINVAL; }
  if (n <= ATT_N && C / heads <= ATT_N)
    hipLaunchKernelGGL(attention_block_kernel, dim3(heads, batch), dim3(256), 0, STREAM(stream), q, k, v, ld, n, C / heads, heads, out, ld_out);
  else
    hipLaunchKernelGGL(attention_kernel, dim3(n * heads, batch), dim3(64), lds, STREAM(stream), q, k, v, ld, n, C, heads, out, ld_out);
  return g6d_check_launch("attention");
}

extern "C" int g6d_layernorm(const float* in, int ld_in, int n, int C, const float* gamma, const float* beta, float eps,
                             float* out, int ld_out, g6d_stream_t stream) {
  if (!in || !out || !gamma || !beta || n <= 0 || C <= 0) { g6d_set_error("layernorm: bad args"); return G6D_EINVAL; }
  hipLaunchKernelGGL(layernorm_kernel, dim3(n), dim3(64), 0, STREAM(stream), in, ld_in, C, gamma, beta, eps, out, ld_out);
  return g6d_check_launch("layernorm");
}

extern "C" int g6d_affine_act_add(const float* in, int ld_in, const float* scale, const float* shift, int relu,
                                  const float* residual, int ld_res, int n, int C, float* out, int ld_out,
                                  int rows_per_group, g6d_stream_t stream) {
  if (!in || !out || n <= 0 || C <= 0 || (scale && !shift) || rows_per_group < 0) { g6d_set_error("affine_act_add: bad args"); return G6D_EINVAL; }
  hipLaunchKernelGGL(affine_act_add_kernel, dim3((n * C + 255) / 256), dim3(256), 0, STREAM(stream), in, ld_in, scale,
                     shift, relu, residual, ld_res, n, C, out, ld_out, rows_per_group);
  return g6d_check_launch("affine_act_add");
}

extern "C" int g6d_linear_gemv_batch(const float* x, int B, int K, const float* W, const float* bias, int O, int act, float* out,
                                     float* workspace, size_t workspace_bytes, g6d_stream_t stream) {
  if (!x || !W || !out || B <= 0 || K <= 0 || (K & 3) || O <= 0 || !g6d_aligned16(x) || !g6d_aligned16(W)) {
    g6d_set_error("linear_gemv_batch: bad args"); return G6D_EINVAL;
  }
  constexpr int KL = GEMVB_THREADS * 4 * 2;                 // KV = 2: K slices of 2048 floats
  const bool rows_ok = O % GEMVB_R == 0 && K % KL == 0 && O / GEMVB_R <= G6D_WS_COUNTERS;
  const int KS = rows_ok ? K / KL : 0;
  // Measured (profiles/r05_gemv.md, 67 MB layer): vector-ALU kernel 22 us for <= 8 right-hand sides, 37 us for 16 (two groups per block),
  // 430 us for 32 (four groups: spills); matrix-core kernel 35-40 us for ANY count up to 32 (its 32-byte pieces of 32 weight rows per
  // load reach 1.9 TB/s; FC as a 1x1 conv on conv_igemm: 40-45 us).  So: groups of <= 16 on the vector-ALU kernel, 17..32 on the matrix
  // cores.  Knob gemv_mfma: 1 = that rule, 2 = matrix cores whenever eligible (tests), 0 = never (groups of <= 16).
  const int mf = (int)g6d_knob(G6D_KNOB_GEMV_MFMA);
  const int NSm = K / (4 * GEMVM_KW);
  const bool mfma_ok = mf != 0 && O % 32 == 0 && K % (4 * GEMVM_KW) == 0 && O / 32 <= G6D_WS_COUNTERS && workspace && g6d_aligned16(workspace) &&
                       workspace_bytes >= G6D_WS_COUNTER_BYTES + (size_t)(O / 32) * NSm * 1024 * sizeof(float);
  const int gmax = (mfma_ok && (mf == 2 || B > 16)) ? 32 : 16;
  for (int b0 = 0; b0 < B; b0 += gmax) {
    const int nb = B - b0 < gmax ? B - b0 : gmax;
    const int nh = nb <= 8 ? 1 : 2;
    const size_t need = G6D_WS_COUNTER_BYTES + (size_t)(O / GEMVB_R) * nh * (KS > 0 ? KS : 1) * GEMVB_R * 8 * sizeof(float);
    const float* xg = x + (size_t)b0 * K;
    float* og = out + (size_t)b0 * O;
    if (mfma_ok && nb >= 2 && (mf == 2 || nb > 16)) {
      hipLaunchKernelGGL((linear_gemv_mfma_kernel<8>), dim3(O / 32, NSm), dim3(256), 0, STREAM(stream), xg, nb, K, W, bias, act, og, O, workspace);
    } else if (nb >= 2 && rows_ok && KS >= 2 && workspace && workspace_bytes >= need && g6d_aligned16(workspace)) {
      const dim3 grid(O / GEMVB_R, KS);
      if (nh == 1) hipLaunchKernelGGL((linear_gemv_batch_kernel<2, 1>), grid, dim3(GEMVB_THREADS), 0, STREAM(stream), xg, nb, K, W, bias, act, og, O, workspace);
      else hipLaunchKernelGGL((linear_gemv_batch_kernel<2, 2>), grid, dim3(GEMVB_THREADS), 0, STREAM(stream), xg, nb, K, W, bias, act, og, O, workspace);
    } else if (nb > 8) {                     // (fallback kernels take <= 8 rows per launch)
      for (int c0 = 0; c0 < nb; c0 += 8) {
        const int nc = nb - c0 < 8 ? nb - c0 : 8;
        if (K >= GEMV_THREADS * 4 * 16)
          hipLaunchKernelGGL(linear_gemv_kernel<16>, dim3(O), dim3(GEMV_THREADS), 0, STREAM(stream), xg + (size_t)c0 * K, nc, K, W, bias, act, og + (size_t)c0 * O, O);
        else
          hipLaunchKernelGGL(linear_gemv_kernel<4>, dim3(O), dim3(GEMV_THREADS), 0, STREAM(stream), xg + (size_t)c0 * K, nc, K, W, bias, act, og + (size_t)c0 * O, O);
      }
    } else if (K >= GEMV_THREADS * 4 * 16) {
      hipLaunchKernelGGL(linear_gemv_kernel<16>, dim3(O), dim3(GEMV_THREADS), 0, STREAM(stream), xg, nb, K, W, bias, act, og, O);
    } else {
      hipLaunchKernelGGL(linear_gemv_kernel<4>, dim3(O), dim3(GEMV_THREADS), 0, STREAM(stream), xg, nb, K, W, bias, act, og, O);
    }
  }
  return g6d_check_launch("linear_gemv_batch");
}

extern "C" int g6d_linear_gemv(const float* x, int B, int K, const float* W, const float* bias, int O, int act,
                               float* out, g6d_stream_t stream) {
  if (!x || !W || !out || B <= 0 || B > 8 || K <= 0 || (K & 3) || O <= 0 || !g6d_aligned16(x) || !g6d_aligned16(W)) {
    g6d_set_error("linear_gemv: bad args"); return G6D_EINVAL;
  }
  if (K >= GEMV_THREADS * 4 * 16)
    hipLaunchKernelGGL(linear_gemv_kernel<16>, dim3(O), dim3(GEMV_THREADS), 0, STREAM(stream), x, B, K, W, bias, act, out, O);
  else
    hipLaunchKernelGGL(linear_gemv_kernel<4>, dim3(O), dim3(GEMV_THREADS), 0, STREAM(stream), x, B, K, W, bias, act, out, O);
  return g6d_check_launch("linear_gemv");
}
