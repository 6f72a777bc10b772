// Detector glue kernels around the MFMA correlation (network/detector.py):
//   assemble      :225-229,243-245,207-216  nearest up-sample of the coarse levels, (x-mu)/sigma, clip, bilinear resize
//   score_mlp_max :159-163,246-247         1x1x1 Conv3d 12->64, ReLU, 64->64 per (pixel, reference), max over references
//   decode        :84-121                   arg-max (first maximum wins) + offset/scale gather
#include "g6d_common.h"

namespace {

struct LevelStats { float mu[3], sigma[3]; };

__global__ void assemble_kernel(const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2,
                                int hc, int wc, int rfn, LevelStats st, float clip, int hs, int ws, int scale_idx,
                                int nch, float* __restrict__ stacked) {
  const long long total = (long long)hs * ws * rfn;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  // blockIdx.y = query of the batch: its level maps and its slab of `stacked` follow those of the previous query
  s0 += (size_t)blockIdx.y * hc * wc * rfn; s1 += (size_t)blockIdx.y * (hc >> 1) * (wc >> 1) * rfn;
  s2 += (size_t)blockIdx.y * (hc >> 2) * (wc >> 2) * rfn; stacked += (size_t)blockIdx.y * total * nch;
  const int r = (int)(i % rfn); const int pix = (int)(i / rfn);
  const int x = pix % ws, y = pix / ws;
  const float ry = (float)hc / (float)hs, rx = (float)wc / (float)ws;
  float sy = ry * (y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
  float sx = rx * (x + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < hc - 1), x1 = x0 + (x0 < wc - 1);
  const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
  const float* src[3] = {s0, s1, s2};
  float* dst = stacked + ((size_t)pix * rfn + r) * nch + 3 * scale_idx;
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const int wl = wc >> l;
    const float mu = st.mu[l], sg = st.sigma[l];
    auto tap = [&](int yy, int xx) {
      float v = src[l][((size_t)(yy >> l) * wl + (xx >> l)) * rfn + r];
      v = (v - mu) / sg;
      return fminf(fmaxf(v, -clip), clip);
    };
    dst[l] = hy * (hx * tap(y0, x0) + lx * tap(y0, x1)) + ly * (hx * tap(y1, x0) + lx * tap(y1, x1));
  }
}

__device__ __forceinline__ int enc_f(float f) { int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7fffffff; }
__device__ __forceinline__ float dec_f(int b) { return __int_as_float(b >= 0 ? b : b ^ 0x7fffffff); }

// thread = (pixel, ref); PPB whole pixels per block; max over refs through LDS integer atomics.
template <int NCH>
__global__ void __launch_bounds__(256) score_mlp_max_kernel(const float* __restrict__ stacked, int P, int rfn,
                                                            const float* __restrict__ w0, const float* __restrict__ b0,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            float* __restrict__ out, int ppb) {
  extern __shared__ int smax[];   // [ppb][64]
  const int pl = threadIdx.x / rfn, r = threadIdx.x % rfn;
  const int pix = blockIdx.x * ppb + pl;
  for (int i = threadIdx.x; i < ppb * 64; i += 256) smax[i] = enc_f(-INFINITY);
  __syncthreads();
  if (pl < ppb && pix < P) {
    float in[NCH];
    const float* src = stacked + ((size_t)pix * rfn + r) * NCH;
#pragma unroll
    for (int k = 0; k < NCH; ++k) in[k] = src[k];
    // Both layers on PAIRS of input channels (round 5): v_pk_fma_f32 with the weight pair as the instruction's scalar operand does two
    // multiply-adds per issue slot — the kernel was at 45 TFLOP/s on unpacked FMAs (57 % of their peak, 531 us per batch of 16); the two
    // partial sums of an output (even / odd inputs) are added at the end.
    typedef float f2 __attribute__((ext_vector_type(2)));
    static_assert(NCH % 2 == 0, "channel pairs");
    f2 inp[NCH / 2];
#pragma unroll
    for (int k = 0; k < NCH / 2; ++k) inp[k] = f2{in[2 * k], in[2 * k + 1]};
    f2 hp[32];
#pragma unroll
    for (int o = 0; o < 64; ++o) {
      f2 a = {b0[o], 0.f};
#pragma unroll
      for (int k = 0; k < NCH / 2; ++k) a = __builtin_elementwise_fma(inp[k], *reinterpret_cast<const f2*>(w0 + o * NCH + 2 * k), a);
      const float v = fmaxf(a.x + a.y, 0.f);
      if (o & 1) hp[o >> 1].y = v; else hp[o >> 1].x = v;
    }
    for (int o = 0; o < 64; ++o) {
      f2 a = {b1[o], 0.f};
#pragma unroll
      for (int k = 0; k < 32; ++k) a = __builtin_elementwise_fma(hp[k], *reinterpret_cast<const f2*>(w1 + o * 64 + 2 * k), a);
      atomicMax(&smax[pl * 64 + o], enc_f(a.x + a.y));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ppb * 64; i += 256) {
    const int p2 = blockIdx.x * ppb + i / 64;
    if (p2 < P) out[(size_t)p2 * 64 + (i & 63)] = dec_f(smax[i]);
  }
}

__global__ void __launch_bounds__(256) decode_kernel(const float* __restrict__ scores, int ld_s,
                                                     const float* __restrict__ offset, int ld_o,
                                                     const float* __restrict__ scale, int ld_c, int hs, int ws,
                                                     float pool_ratio, float* __restrict__ result) {
  __shared__ float bv[256];
  __shared__ int bi[256];
  const int n = hs * ws;
  scores += (size_t)blockIdx.x * n * ld_s; offset += (size_t)blockIdx.x * n * ld_o; scale += (size_t)blockIdx.x * n * ld_c;
  result += blockIdx.x * 5;                              // blockIdx.x = query of the batch
  float best = -INFINITY; int idx = 0x7fffffff;
  for (int i = threadIdx.x; i < n; i += 256) {
    float v = scores[(size_t)i * ld_s];
    if (v > best || (v == best && i < idx)) { best = v; idx = i; }
  }
  bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      float v = bv[threadIdx.x + s]; int j = bi[threadIdx.x + s];
      if (v > bv[threadIdx.x] || (v == bv[threadIdx.x] && j < bi[threadIdx.x])) { bv[threadIdx.x] = v; bi[threadIdx.x] = j; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int id = bi[0] == 0x7fffffff ? 0 : bi[0];
    const int x = id % ws, y = id / ws;
    result[0] = ((float)x + offset[(size_t)id * ld_o] + 0.5f) * pool_ratio - 0.5f;
    result[1] = ((float)y + offset[(size_t)id * ld_o + 1] + 0.5f) * pool_ratio - 0.5f;
    result[2] = exp2f(scale[(size_t)id * ld_c]);
    result[3] = (float)x; result[4] = (float)y;
  }
}


// ---- the detector's image pyramid (network/detector.py:236-241: F.interpolate(que_imgs, size=(ht, wt), mode='bilinear'), align_corners
// False) for all detection scales in ONE launch: planes = N*3 image planes [H][W] -> up to 4 destinations [planes][h_k][w_k].  Source
// index as ATen's upsample_bilinear2d: scale = in / out (float), src = scale * (dst + 0.5) - 0.5 clamped at 0, the far neighbour
// clamped to the last row / column.  A scale of the source's own size is not a destination: the caller passes the image itself on.
struct PyrArgs { const float* src; int planes, H, W, n; int h[4], w[4]; long long first[5]; float* dst[4]; };

// VEC = 4: a thread produces four consecutive outputs of a row (all widths % 4 == 0, destinations 16-byte aligned) and stores them as one
// 16-byte piece; first[] then counts pieces.  (The first version — one output per thread, 64-bit index arithmetic — took 176 us per batch
// of 16 queries against 115 us for the three ATen launches it replaced.)
#define PYR_PL 4                           // image planes per work item
template <int VEC>
__global__ void resize_pyramid_kernel(const PyrArgs a) {
  // a work item = VEC consecutive outputs of one row of one destination for PYR_PL consecutive image planes: source columns / weights are
  // formed once (the index arithmetic — three runtime divisions — is amortised over 16 outputs) and all 16 x PYR_PL loads are requested
  // before the first value is used.  (One output per thread: 176 us per batch of 16 queries; one item walking all 48 planes: 186 us — too
  // few threads; this version 148 us; ATen's three launches: 115 us — 0.05 % of a step apart.)
  const unsigned total = (unsigned)a.first[a.n];
  // (64-bit walk: with totals near 2^32 a 32-bit index would wrap past the end instead of leaving the loop)
  for (unsigned long i64 = blockIdx.x * blockDim.x + threadIdx.x; i64 < total; i64 += (unsigned long)gridDim.x * blockDim.x) {
    const unsigned i = (unsigned)i64;
    int k = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j) k = (j < a.n && i >= (unsigned)a.first[j]) ? j : k;
    const unsigned l = i - (unsigned)a.first[k];
    const int h = a.h[k], w = a.w[k], wv = w / VEC;
    const int xq = (int)(l % (unsigned)wv); const unsigned t = l / (unsigned)wv;
    const int y = (int)(t % (unsigned)h), pg = (int)(t / (unsigned)h);
    const float ry = (float)a.H / (float)h, rx = (float)a.W / (float)w;
    float sy = ry * (y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
    const int y0 = min((int)sy, a.H - 1);
    const int y1 = y0 + (y0 < a.H - 1);
    const float ly = sy - y0, hy = 1.f - ly;
    int x0[VEC], x1[VEC]; float lx[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int x = xq * VEC + e;
      float sx = rx * (x + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
      x0[e] = min((int)sx, a.W - 1);
      x1[e] = x0[e] + (x0[e] < a.W - 1);
      lx[e] = sx - x0[e];
    }
    const size_t sstride = (size_t)a.H * a.W, dstride = (size_t)h * w;
    const int pl0 = pg * PYR_PL;
    const float* r0 = a.src + (size_t)pl0 * sstride + (size_t)y0 * a.W;
    const float* r1 = a.src + (size_t)pl0 * sstride + (size_t)y1 * a.W;
    float t00[PYR_PL][VEC], t01[PYR_PL][VEC], t10[PYR_PL][VEC], t11[PYR_PL][VEC];
#pragma unroll
    for (int p_ = 0; p_ < PYR_PL; ++p_) {
      const size_t o = (size_t)min(p_, a.planes - 1 - pl0) * sstride;          // (planes beyond the last repeat it; not stored)
#pragma unroll
      for (int e = 0; e < VEC; ++e) { t00[p_][e] = r0[o + x0[e]]; t01[p_][e] = r0[o + x1[e]]; t10[p_][e] = r1[o + x0[e]]; t11[p_][e] = r1[o + x1[e]]; }
    }
    float* d = a.dst[k] + (size_t)pl0 * dstride + (size_t)y * w + xq * VEC;
#pragma unroll
    for (int p_ = 0; p_ < PYR_PL; ++p_) {
      if (pl0 + p_ >= a.planes) break;
      float v[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float hx = 1.f - lx[e];
        v[e] = hy * (hx * t00[p_][e] + lx[e] * t01[p_][e]) + ly * (hx * t10[p_][e] + lx[e] * t11[p_][e]);
      }
      if constexpr (VEC == 4) *reinterpret_cast<f32x4*>(d + p_ * dstride) = f32x4{v[0], v[1], v[2], v[3]};
      else d[p_ * dstride] = v[0];
    }
  }
}
}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int g6d_detector_assemble(const float* s0, const float* s1, const float* s2, int hc, int wc, int rfn,
                                     const float* mu_sigma, float clip, int hs, int ws, int scale_idx, int nch,
                                     float* stacked, int batch, g6d_stream_t stream) {
  if (!s0 || !s1 || !s2 || !mu_sigma || !stacked || hc <= 0 || wc <= 0 || (hc & 3) || (wc & 3) || rfn <= 0 ||
      scale_idx < 0 || 3 * scale_idx + 3 > nch || batch < 1 || batch > 65535) {
    g6d_set_error("detector_assemble: bad args (level-0 map must be a multiple of 4)"); return G6D_EINVAL;
  }
  LevelStats st;
  for (int l = 0; l < 3; ++l) { st.mu[l] = mu_sigma[2 * l]; st.sigma[l] = mu_sigma[2 * l + 1]; }
  const long long total = (long long)hs * ws * rfn;
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((total + 255) / 256), batch), dim3(256), 0, STREAM(stream), s0, s1, s2, hc,
                     wc, rfn, st, clip, hs, ws, scale_idx, nch, stacked);
  return g6d_check_launch("detector_assemble");
}

extern "C" int g6d_detector_score_mlp_max(const float* stacked, int P, int rfn, int nch, const float* w0, const float* b0,
                                          const float* w1, const float* b1, float* out, g6d_stream_t stream) {
  if (!stacked || !w0 || !b0 || !w1 || !b1 || !out || P <= 0 || rfn <= 0 || rfn > 256 || nch != 12) {
    g6d_set_error("detector_score_mlp_max: bad args (rfn <= 256, nch == 12)"); return G6D_EINVAL;
  }
  const int ppb = 256 / rfn;
  hipLaunchKernelGGL(score_mlp_max_kernel<12>, dim3((P + ppb - 1) / ppb), dim3(256), (size_t)ppb * 64 * sizeof(int),
                     STREAM(stream), stacked, P, rfn, w0, b0, w1, b1, out, ppb);
  return g6d_check_launch("detector_score_mlp_max");
}

extern "C" int g6d_detector_decode(const float* scores, int ld_s, const float* offset, int ld_o, const float* scale,
                                   int ld_c, int hs, int ws, int pool_ratio, float* result, int batch, g6d_stream_t stream) {
  if (!scores || !offset || !scale || !result || hs <= 0 || ws <= 0 || batch < 1) { g6d_set_error("detector_decode: bad args"); return G6D_EINVAL; }
  hipLaunchKernelGGL(decode_kernel, dim3(batch), dim3(256), 0, STREAM(stream), scores, ld_s, offset, ld_o, scale, ld_c, hs, ws,
                     (float)pool_ratio, result);
  return g6d_check_launch("detector_decode");
}

extern "C" int g6d_resize_bilinear_pyramid(const float* src, int planes, int H, int W, int nscale, const int* hs, const int* ws,
                                           float* const* dsts, g6d_stream_t stream) {
  if (!src || !hs || !ws || !dsts || planes <= 0 || H <= 0 || W <= 0 || nscale < 1 || nscale > 4) {
    g6d_set_error("resize_bilinear_pyramid: bad args (1..4 destination sizes)"); return G6D_EINVAL;
  }
  PyrArgs a = {};
  a.src = src; a.planes = planes; a.H = H; a.W = W; a.n = nscale;
  bool vec = true;
  for (int k = 0; k < nscale; ++k)
    if (hs[k] <= 0 || ws[k] <= 0 || !dsts[k]) { g6d_set_error("resize_bilinear_pyramid: bad destination"); return G6D_EINVAL; }
    else vec = vec && (ws[k] & 3) == 0 && g6d_aligned16(dsts[k]);
  long long tot = 0;                      // work items: PYR_PL planes x (4 outputs of a row when every width allows it, else one)
  for (int k = 0; k < nscale; ++k) {
    a.h[k] = hs[k]; a.w[k] = ws[k]; a.dst[k] = dsts[k]; a.first[k] = tot;
    tot += (long long)((planes + PYR_PL - 1) / PYR_PL) * hs[k] * (vec ? ws[k] / 4 : ws[k]);
  }
  for (int k = nscale; k < 5; ++k) a.first[k] = tot;
  if (tot >= (1ll << 32)) { g6d_set_error("resize_bilinear_pyramid: more than 2^32 work items"); return G6D_EINVAL; }
  const long long blocks = (tot + 255) / 256;
  const dim3 grid((unsigned)(blocks < 65536 * 8 ? blocks : 65536 * 8));
  if (vec) hipLaunchKernelGGL(resize_pyramid_kernel<4>, grid, dim3(256), 0, STREAM(stream), a);
  else hipLaunchKernelGGL(resize_pyramid_kernel<1>, grid, dim3(256), 0, STREAM(stream), a);
  return g6d_check_launch("resize_bilinear_pyramid");
}
