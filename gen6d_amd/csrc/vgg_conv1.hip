// First trunk layer fused: conv 3x3 (3 -> 64 channels, pad 1) + folded-BatchNorm bias + ReLU + 2x2 max-pool, NCHW in
// and out (reference network/pretrain_models.py:17-25,66-72: features[0..3] of vgg11_bn).
//
// With 3 input channels the layer is 27 MACs per output: the library path (convolution, layout transposes, then the
// separate bias/ReLU/pool pass over the 167 MB full-resolution result) takes 113 us on the 704x928 detector scale.  Here
// a thread owns one POOLED output pixel and 16 output channels: its 4x4x3 input window sits in registers, the 27x64
// weights in LDS (read as wave-uniform 16-byte broadcasts), 4 conv positions x 16 channels accumulate in registers, and
// only the pooled map (1/4 of the conv output) is ever written.  fp32 FMA on the vector pipe; algorithmic bytes: 12 B read per
// input pixel, 64 B written per input pixel.
//
// The two phases of the kernel are plain inline functions of (thread id, block origin) so that the index arithmetic can
// be run thread by thread on the host: tests/test_conv1_emulation_cpu.py builds this file with -DG6D_CONV1_HOST_EMU
// and checks the emulation against torch (test infrastructure only; the library itself has no host compute path).
#ifndef G6D_CONV1_HOST_EMU
#include "g6d_common.h"
#include <stdlib.h>
#define G6D_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#include <cstddef>
#define G6D_HD inline
#endif

#define C1_PTX 32                      // pooled tile: 32 x 2 outputs; 256 threads = 64 pixels x 4 channel groups of 16
#define C1_PTY 2
#define C1_IW (2 * C1_PTX + 2)         // input tile incl. halo: 66 x 6
#define C1_IH (2 * C1_PTY + 2)
#define C1_IWP (C1_IW + 1)
#define C1_CIN 3
#define C1_COUT 64
#define C1_TAPS 27
#define C1_THREADS (C1_PTX * C1_PTY * (C1_COUT / 16))

// optional input normalisation (x - mean[c]) / std[c] applied while the tile is staged (torchvision Normalize of
// network/pretrain_models.py / dataset code: the images arrive in [0,1]); the zero padding stays zero, as in the reference
// where the padding follows the normalisation
struct Conv1Norm { float mean[C1_CIN], std[C1_CIN]; int on; };

struct Conv1Smem {
  float in[C1_CIN][C1_IH][C1_IWP];
  float w[C1_TAPS][C1_COUT];           // tap-major: the 16 channels of one pass are contiguous
  float b[C1_COUT];
};

// Phase 1: stage the zero-padded input tile of image n whose pooled origin is (px0, py0), and the weights.
G6D_HD void conv1_stage(Conv1Smem& s, int tid, const float* in, const float* w_oihw, const float* bias, int n, int H, int W,
                        int px0, int py0, const Conv1Norm& nm) {
  const int gx0 = 2 * px0 - 1, gy0 = 2 * py0 - 1;
  for (int i = tid; i < C1_CIN * C1_IH * C1_IW; i += C1_THREADS) {
    const int col = i % C1_IW, r = (i / C1_IW) % C1_IH, c = i / (C1_IW * C1_IH);
    const int gy = gy0 + r, gx = gx0 + col;
    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
    float v = ok ? in[((size_t)(n * C1_CIN + c) * H + gy) * W + gx] : 0.f;
    if (nm.on && ok) v = (v - nm.mean[c]) / nm.std[c];
    s.in[c][r][col] = v;
  }
  for (int i = tid; i < C1_TAPS * C1_COUT; i += C1_THREADS) {
    const int co = i % C1_COUT, t = i / C1_COUT;
    s.w[t][co] = w_oihw[co * C1_TAPS + t];          // OIHW: [co][ci][ky][kx] -> tap t = (ci*3 + ky)*3 + kx
  }
  if (tid < C1_COUT) s.b[tid] = bias[tid];
}

// Phase 2: thread (tx, ty, cg) computes pooled pixel (px0 + tx, py0 + ty) for the 16 channels of group cg.  A wavefront
// holds one cg (64 pixels), so its weight reads are wave-uniform; splitting the channels over threads instead of looping
// keeps the serial FMA chain of a thread at 1728 instead of 6912, which is what bounds the small 128x128 crops.
G6D_HD void conv1_compute(const Conv1Smem& s, int tid, float* out, int n, int Ho, int Wo, int px0, int py0, bool nhwc) {
  const int pix = tid % (C1_PTX * C1_PTY), cg = tid / (C1_PTX * C1_PTY);
  const int tx = pix % C1_PTX, ty = pix / C1_PTX;
  const int px = px0 + tx, py = py0 + ty;
  const bool live = px < Wo && py < Ho;
  float v[C1_CIN][4][4];
#pragma unroll
  for (int c = 0; c < C1_CIN; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[c][r][j] = s.in[c][2 * ty + r][2 * tx + j];
  {
    float acc[4][16];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[p][k] = 0.f;
#pragma unroll
    for (int t = 0; t < C1_TAPS; ++t) {
      const int c = t / 9, ky = (t % 9) / 3, kx = t % 3;
      float wv[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) wv[k] = s.w[t][cg * 16 + k];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float x = v[c][(p >> 1) + ky][(p & 1) + kx];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[p][k] = fmaf(x, wv[k], acc[p][k]);
      }
    }
    if (live) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int co = cg * 16 + k;
        float m = fmaxf(fmaxf(acc[0][k], acc[1][k]), fmaxf(acc[2][k], acc[3][k])) + s.b[co];   // max(a)+b == max(a+b)
        m = fmaxf(m, 0.f);                                                                     // relu(max) == max(relu)
        if (nhwc) out[((size_t)(n * Ho + py) * Wo + px) * C1_COUT + co] = m;      // 16 consecutive channels = 64 B per thread
        else out[((size_t)(n * C1_COUT + co) * Ho + py) * Wo + px] = m;
      }
    }
  }
}

#ifndef G6D_CONV1_HOST_EMU

namespace {

__global__ void __launch_bounds__(C1_THREADS) vgg_conv1_pool_kernel(const float* __restrict__ in,
                                                                    const float* __restrict__ w_oihw,
                                                                    const float* __restrict__ bias, int H, int W, int Ho,
                                                                    int Wo, float* __restrict__ out, int nhwc, const Conv1Norm nm) {
  __shared__ Conv1Smem s;
  const int px0 = blockIdx.x * C1_PTX, py0 = blockIdx.y * C1_PTY, n = blockIdx.z;
  conv1_stage(s, threadIdx.x, in, w_oihw, bias, n, H, W, px0, py0, nm);
  __syncthreads();
  conv1_compute(s, threadIdx.x, out, n, Ho, Wo, px0, py0, nhwc != 0);
}


// ---- round 3: the same layer on the matrix cores (channels-last result; the own trunk's first layer) ------------------------
// The vector-pipe kernel above runs at ~14 TFLOP/s (250 us per query over the detector's pyramid, 4 % of the batched step).  As a
// GEMM the layer is M = conv pixels, N = 64, K = 27 (padded to 28): 14 x v_mfma_f32_32x32x2_f32 per 32 pixels and 32 channels.
//   block  = 4 waves, pooled tile 32 x 8 (conv 64 x 16 + halo staged in LDS as three planes, normalised while staged);
//   wave   = 2 pooled rows = 8 groups of 8 pool windows; a group = 32 conv pixels: A row 4q + p = pixel p (dy = p >> 1, dx = p & 1) of
//            pool window q, so that the 4 accumulator registers r = 4j .. 4j+3 of a lane are the four pixels of window 2j + lh:
//            bias / ReLU / 2x2 max-pool are per-lane register arithmetic and a half-wave stores 32 consecutive channels;
//   A      = one ds_read_b32 per lane and K step (lane half lh supplies k = 2s + lh), shared by the two 32-channel N tiles;
//   B      = the lane's 2 x 14 filter values, in registers for the whole block.
#define M1_PTX 32
#define M1_PTY 8
#define M1_IW (2 * M1_PTX + 2)
#define M1_IH (2 * M1_PTY + 2)
#define M1_IWP 80                       // row pitch = 16 mod 32: the two conv rows of a pool window hit disjoint banks

// OT: element type of the channels-last result — float, or (round 6, the reduced-precision mode's 16-bit activation path: the input of
// g6d_conv16_direct_multi) _Float16 / __bf16, rounded once here.
// PAIR (round 6): fp16 hi / lo pairs [pixel][2][64] — hi = rn16(v), lo = rn16(v - hi) — the input format of the fp32 path's
// split-precision trunk kernel (g6d_conv16_direct_multi, math_mode 3).
template <typename OT, bool PAIR = false>
__global__ void __launch_bounds__(256) vgg_conv1_pool_mfma_kernel(const float* __restrict__ in, const float* __restrict__ w_oihw,
                                                                  const float* __restrict__ bias, int H, int W, int Ho, int Wo,
                                                                  OT* __restrict__ out, const Conv1Norm nm) {
  __shared__ float tile[C1_CIN][M1_IH][M1_IWP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int px0 = blockIdx.x * M1_PTX, py0 = blockIdx.y * M1_PTY, n = blockIdx.z;
  const int gx0 = 2 * px0 - 1, gy0 = 2 * py0 - 1;
  for (int i = tid; i < C1_CIN * M1_IH * M1_IW; i += 256) {
    const int col = i % M1_IW, r = (i / M1_IW) % M1_IH, c = i / (M1_IW * M1_IH);
    const int gy = gy0 + r, gx = gx0 + col;
    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
    float v = ok ? in[((size_t)(n * C1_CIN + c) * H + gy) * W + gx] : 0.f;
    if (nm.on && ok) v = (v - nm.mean[c]) / nm.std[c];
    tile[c][r][col] = v;
  }
  // filter values of this lane: K step s supplies k = 2s + lh (k = 27: zero padding of K), N tile t supplies channel li + 32 t
  float wb[2][14];
  int koff[14];
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k = 2 * s + lh;
    const int kk = k < C1_TAPS ? k : 0;
    const int c = kk / 9, ky = (kk % 9) / 3, kx = kk % 3;
    koff[s] = (c * M1_IH + ky) * M1_IWP + kx;
#pragma unroll
    for (int t = 0; t < 2; ++t) wb[t][s] = k < C1_TAPS ? w_oihw[(li + 32 * t) * C1_TAPS + k] : 0.f;
  }
  const float b0 = bias[li], b1 = bias[li + 32];
  __syncthreads();
  const float* tbase = &tile[0][0][0];
  const int q = li >> 2, p = li & 3;
  const int lane_off = (p >> 1) * M1_IWP + 2 * q + (p & 1);          // pixel p of pool window q inside a group
#pragma unroll 1
  for (int g = 0; g < 8; ++g) {
    const int ty = 2 * wave + (g >> 2), gx = g & 3;                    // pooled row of the tile, group of 8 windows along x
    const float* a = tbase + (2 * ty) * M1_IWP + 16 * gx + lane_off;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const float av = a[koff[s]];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[0][s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[1][s], acc1, 0, 0, 0);
    }
    const int py = py0 + ty;
    if (py < Ho) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int px = px0 + 8 * gx + 2 * j + lh;                      // accumulator rows 8j + 4lh .. +3 = window 2j + lh
        if (px < Wo) {
          const float m0 = fmaxf(fmaxf(acc0[4 * j], acc0[4 * j + 1]), fmaxf(acc0[4 * j + 2], acc0[4 * j + 3])) + b0;   // max(a)+b == max(a+b)
          const float m1 = fmaxf(fmaxf(acc1[4 * j], acc1[4 * j + 1]), fmaxf(acc1[4 * j + 2], acc1[4 * j + 3])) + b1;
          OT* o = out + ((size_t)(n * Ho + py) * Wo + px) * (PAIR ? 2 * C1_COUT : C1_COUT);
          const float r0 = fmaxf(m0, 0.f), r1 = fmaxf(m1, 0.f);               // relu(max) == max(relu)
          o[li] = (OT)r0; o[li + 32] = (OT)r1;
          if constexpr (PAIR) { o[C1_COUT + li] = (OT)(r0 - (float)(OT)r0); o[C1_COUT + li + 32] = (OT)(r1 - (float)(OT)r1); }
        }
      }
    }
  }
}

int conv1_launch(const float* in, int N, int H, int W, const float* w_oihw, const float* bias, int Cin, int Cout, float* out,
                 int nhwc, g6d_stream_t stream, const float* mean = nullptr, const float* stdv = nullptr) {
  if (!in || !w_oihw || !bias || !out || N <= 0 || N > 65535 || H < 2 || W < 2 || Cin != C1_CIN || Cout != C1_COUT ||
      (long long)N * Cout * (H / 2) * (W / 2) >= (1ll << 31)) {
    g6d_set_error("vgg_conv1_pool: bad args (3 -> 64 channels, H, W >= 2)"); return G6D_EINVAL;
  }
  const int Ho = H / 2, Wo = W / 2;
  Conv1Norm nm = {};
  if (mean && stdv) {
    for (int c = 0; c < C1_CIN; ++c) { nm.mean[c] = mean[c]; nm.std[c] = stdv[c]; }
    nm.on = 1;
  }
  // channels-last results take the matrix-core kernel (knob conv1_mfma = 0: the vector-pipe kernel)
  const bool use_mfma = g6d_knob(G6D_KNOB_CONV1_MFMA) != 0;
  if (nhwc && use_mfma) {
    hipLaunchKernelGGL(vgg_conv1_pool_mfma_kernel<float>, dim3((Wo + M1_PTX - 1) / M1_PTX, (Ho + M1_PTY - 1) / M1_PTY, N), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), in, w_oihw, bias, H, W, Ho, Wo, out, nm);
    return g6d_check_launch("vgg_conv1_pool_mfma");
  }
  hipLaunchKernelGGL(vgg_conv1_pool_kernel, dim3((Wo + C1_PTX - 1) / C1_PTX, (Ho + C1_PTY - 1) / C1_PTY, N), dim3(C1_THREADS),
                     0, reinterpret_cast<hipStream_t>(stream), in, w_oihw, bias, H, W, Ho, Wo, out, nhwc, nm);
  return g6d_check_launch("vgg_conv1_pool");
}

}  // namespace

// in [N][3][H][W], w_oihw [64][3][3][3] (BatchNorm folded), bias [64] -> out [N][64][H/2][W/2] (floor, as F.max_pool2d).
extern "C" int g6d_vgg_conv1_pool(const float* in, int N, int H, int W, const float* w_oihw, const float* bias, int Cin,
                                  int Cout, float* out, g6d_stream_t stream) {
  return conv1_launch(in, N, H, W, w_oihw, bias, Cin, Cout, out, 0, stream);
}

// Same layer with a channels-last result [N][H/2][W/2][64]: the input of g6d_wino_conv3x3 (the own trunk).
extern "C" int g6d_vgg_conv1_pool_nhwc(const float* in, int N, int H, int W, const float* w_oihw, const float* bias, int Cin,
                                       int Cout, float* out, g6d_stream_t stream) {
  return conv1_launch(in, N, H, W, w_oihw, bias, Cin, Cout, out, 1, stream);
}

// As g6d_vgg_conv1_pool_nhwc on an image in [0,1]: (x - mean[c]) / std[c] (HOST arrays of 3 floats: torchvision Normalize) is
// applied while the input tile is staged, so the normalised image is never written.
extern "C" int g6d_vgg_conv1_pool_nhwc_norm(const float* in, int N, int H, int W, const float* w_oihw, const float* bias, int Cin,
                                            int Cout, const float* mean_host, const float* std_host, float* out, g6d_stream_t stream) {
  if (!mean_host || !std_host) { g6d_set_error("vgg_conv1_pool_nhwc_norm: mean / std missing"); return G6D_EINVAL; }
  return conv1_launch(in, N, H, W, w_oihw, bias, Cin, Cout, out, 1, stream, mean_host, std_host);
}

// The same layer with a 16-BIT channels-last result (math_mode 1 = bf16, 2 = fp16; ABI v11): the first layer of the reduced-precision
// mode's 16-bit activation path, rounded once in the epilogue; math_mode 3: fp16 hi / lo pairs [N][H/2][W/2][2][64], the first layer of
// the fp32 path's split-precision trunk.  mean_host / std_host may be NULL (already normalised input).
extern "C" int g6d_vgg_conv1_pool_nhwc16(const float* in, int N, int H, int W, const float* w_oihw, const float* bias, int Cin, int Cout,
                                         const float* mean_host, const float* std_host, void* out16, int math_mode, g6d_stream_t stream) {
  if (!in || !w_oihw || !bias || !out16 || N <= 0 || N > 65535 || H < 2 || W < 2 || Cin != C1_CIN || Cout != C1_COUT ||
      math_mode < 1 || math_mode > 3 || (long long)N * Cout * (H / 2) * (W / 2) >= (1ll << 30)) {
    g6d_set_error("vgg_conv1_pool_nhwc16: bad args (3 -> 64 channels, H, W >= 2, math_mode 1 / 2 / 3)"); return G6D_EINVAL;
  }
  const int Ho = H / 2, Wo = W / 2;
  Conv1Norm nm = {};
  if (mean_host && std_host) {
    for (int c = 0; c < C1_CIN; ++c) { nm.mean[c] = mean_host[c]; nm.std[c] = std_host[c]; }
    nm.on = 1;
  }
  const dim3 grid((Wo + M1_PTX - 1) / M1_PTX, (Ho + M1_PTY - 1) / M1_PTY, N);
  if (math_mode == 1)
    hipLaunchKernelGGL(vgg_conv1_pool_mfma_kernel<__bf16>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, w_oihw, bias, H, W, Ho, Wo,
                       static_cast<__bf16*>(out16), nm);
  else if (math_mode == 2)
    hipLaunchKernelGGL(vgg_conv1_pool_mfma_kernel<_Float16>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, w_oihw, bias, H, W, Ho, Wo,
                       static_cast<_Float16*>(out16), nm);
  else
    hipLaunchKernelGGL((vgg_conv1_pool_mfma_kernel<_Float16, true>), grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, w_oihw, bias, H, W,
                       Ho, Wo, static_cast<_Float16*>(out16), nm);
  return g6d_check_launch("vgg_conv1_pool_mfma16");
}

#else   // ---- host emulation of the two phases, thread by thread (tests only) -------------------------------------

extern "C" int g6d_conv1_emulate(const float* in, int N, int H, int W, const float* w_oihw, const float* bias, float* out,
                                 int nhwc, const float* mean, const float* stdv) {
  const int Ho = H / 2, Wo = W / 2;
  Conv1Smem* s = new Conv1Smem;
  Conv1Norm nm = {};
  if (mean && stdv) { for (int c = 0; c < C1_CIN; ++c) { nm.mean[c] = mean[c]; nm.std[c] = stdv[c]; } nm.on = 1; }
  for (int n = 0; n < N; ++n)
    for (int by = 0; by < (Ho + C1_PTY - 1) / C1_PTY; ++by)
      for (int bx = 0; bx < (Wo + C1_PTX - 1) / C1_PTX; ++bx) {
        for (int tid = 0; tid < C1_THREADS; ++tid) conv1_stage(*s, tid, in, w_oihw, bias, n, H, W, bx * C1_PTX, by * C1_PTY, nm);
        for (int tid = 0; tid < C1_THREADS; ++tid) conv1_compute(*s, tid, out, n, Ho, Wo, bx * C1_PTX, by * C1_PTY, nhwc != 0);
      }
  delete s;
  return 0;
}

#endif
