// Projective image warp on the device: replaces the cv2.warpPerspective / cv2.warpAffine calls that sit between the
// stages of Gen6DEstimator.predict (reference utils/base_utils.py:646-655 transformation_crop,
// utils/database_utils.py:8-25 look_at_crop, :54-110 normalize_reference_views, estimator.py:150-164).
//   dst(x,y) = bilinear( src, Hinv * (x, y, 1) ),  zero outside the source (cv2 BORDER_CONSTANT, INTER_LINEAR).
// Exact float bilinear weights; OpenCV quantises the weights to 1/32 pixel, so uint8 results may differ from cv2 by
// 1-2 grey levels (parity for this host glue is "unpinned": cv2 is not vendored in the reference).
#include "g6d_common.h"

namespace {

struct Mat3 { float m[9]; };

template <typename TOut>
__global__ void warp_perspective_kernel(const unsigned char* __restrict__ src, int sh, int sw, int ch, Mat3 hinv,
                                        TOut* __restrict__ dst, int dh, int dw, float out_scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dh * dw) return;
  const int x = i % dw, y = i / dw;
  const float X = hinv.m[0] * x + hinv.m[1] * y + hinv.m[2];
  const float Y = hinv.m[3] * x + hinv.m[4] * y + hinv.m[5];
  const float Wd = hinv.m[6] * x + hinv.m[7] * y + hinv.m[8];
  const float iw = Wd != 0.f ? 1.f / Wd : 0.f;
  float fx = X * iw, fy = Y * iw;
  fx = fminf(fmaxf(fx, -4.f), (float)sw + 4.f);
  fy = fminf(fmaxf(fy, -4.f), (float)sh + 4.f);
  const float x0f = floorf(fx), y0f = floorf(fy);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float ax = fx - x0f, ay = fy - y0f;
  const bool vx0 = (unsigned)x0 < (unsigned)sw, vx1 = (unsigned)(x0 + 1) < (unsigned)sw;
  const bool vy0 = (unsigned)y0 < (unsigned)sh, vy1 = (unsigned)(y0 + 1) < (unsigned)sh;
  const int xc0 = min(max(x0, 0), sw - 1), xc1 = min(max(x0 + 1, 0), sw - 1);
  const int yc0 = min(max(y0, 0), sh - 1), yc1 = min(max(y0 + 1, 0), sh - 1);
  const float w00 = (vx0 && vy0) ? (1.f - ax) * (1.f - ay) : 0.f, w01 = (vx1 && vy0) ? ax * (1.f - ay) : 0.f;
  const float w10 = (vx0 && vy1) ? (1.f - ax) * ay : 0.f, w11 = (vx1 && vy1) ? ax * ay : 0.f;
  for (int c = 0; c < ch; ++c) {
    const float v = w00 * src[((size_t)yc0 * sw + xc0) * ch + c] + w01 * src[((size_t)yc0 * sw + xc1) * ch + c] +
                    w10 * src[((size_t)yc1 * sw + xc0) * ch + c] + w11 * src[((size_t)yc1 * sw + xc1) * ch + c];
    if constexpr (sizeof(TOut) == 1) dst[(size_t)i * ch + c] = (TOut)fminf(fmaxf(rintf(v), 0.f), 255.f);
    else dst[(size_t)i * ch + c] = (TOut)(v * out_scale);
  }
}

}  // namespace

// src uint8 [sh][sw][ch]; hinv = HOST pointer to the row-major 3x3 map from destination to source pixels;
// dst [dh][dw][ch] uint8 (out_float = 0, rounded to nearest) or float32 scaled by out_scale (out_float = 1).
extern "C" int g6d_warp_perspective(const unsigned char* src, int sh, int sw, int ch, const float* hinv, void* dst, int dh,
                                    int dw, int out_float, float out_scale, g6d_stream_t stream) {
  if (!src || !hinv || !dst || sh <= 0 || sw <= 0 || ch <= 0 || ch > 4 || dh <= 0 || dw <= 0) {
    g6d_set_error("warp_perspective: bad args"); return G6D_EINVAL;
  }
  Mat3 m;
  for (int i = 0; i < 9; ++i) m.m[i] = hinv[i];
  const int n = dh * dw;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (out_float)
    hipLaunchKernelGGL(warp_perspective_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s, src, sh, sw, ch, m,
                       reinterpret_cast<float*>(dst), dh, dw, out_scale);
  else
    hipLaunchKernelGGL(warp_perspective_kernel<unsigned char>, dim3((n + 255) / 256), dim3(256), 0, s, src, sh, sw, ch, m,
                       reinterpret_cast<unsigned char*>(dst), dh, dw, 1.f);
  return g6d_check_launch("warp_perspective");
}
