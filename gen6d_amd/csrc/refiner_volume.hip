// Refiner feature-volume construction (network/refiner.py:183-206,208-247 + network/operator.py:4-17), fused.
//
// The reference builds a [rfn][C][sn^3] tensor with grid_sample (100 MB at C=128, sn=32, rfn=6) and then reduces it
// three times (mean, std, query).  Here each half-wavefront owns one voxel: 32 lanes x 4 channels = 128 channels, so
// every bilinear tap is one 512-byte coalesced read of a channels-last feature row (the 3.7 MB of feature maps stay
// L2 resident), the per-reference samples live in registers, and mean / unbiased std / query sample are written
// straight into the layouts the 3-D CNN consumes (cat[mean, query] and std).  HBM traffic = the 50 MB of outputs.
// The rfn+1 projections of a voxel are spread over the lanes of its half-wave (lane v = view v) and exchanged with
// shuffles: done redundantly by all 32 lanes they were ~600 VALU instructions per voxel, as long as the gathers.
#include "g6d_common.h"

#define MAX_RFN 8

namespace {


// Bilinear footprint of one view at one voxel: element offsets of the 4 taps (clamped into the map) and their weights
// (0 for taps outside the map: grid_sample padding_mode='zeros', align_corners=False).
struct Taps { int o00, o01, o10, o11; float w00, w01, w10, w11; };

__device__ __forceinline__ Taps project_view(int fh, int fw, int C, const float (&P)[12], float vx, float vy,
                                             float vz, float h_in, float w_in) {
  float X = vx * P[0] + vy * P[1] + vz * P[2] + P[3];
  float Y = vx * P[4] + vy * P[5] + vz * P[6] + P[7];
  float Z = vx * P[8] + vy * P[9] + vz * P[10] + P[11];
  if (Z < 1e-4f) Z = 1e-4f;
  float u = X / Z, v = Y / Z;
  float gx = ((u + 0.5f) / w_in - 0.5f) * 2.f;
  float gy = ((v + 0.5f) / h_in - 0.5f) * 2.f;
  float ix = ((gx + 1.f) * fw - 1.f) * 0.5f;
  float iy = ((gy + 1.f) * fh - 1.f) * 0.5f;
  float fx = floorf(ix), fy = floorf(iy);
  // keep the integer conversion in range for far-away projections
  fx = fminf(fmaxf(fx, -4.f), (float)fw + 4.f);
  fy = fminf(fmaxf(fy, -4.f), (float)fh + 4.f);
  int x0 = (int)fx, y0 = (int)fy;
  float wx1 = ix - fx, wy1 = iy - fy, wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy;
  const bool xv0 = (unsigned)x0 < (unsigned)fw, xv1 = (unsigned)(x0 + 1) < (unsigned)fw;
  const bool yv0 = (unsigned)y0 < (unsigned)fh, yv1 = (unsigned)(y0 + 1) < (unsigned)fh;
  const int xc0 = min(max(x0, 0), fw - 1), xc1 = min(max(x0 + 1, 0), fw - 1);
  const int yc0 = min(max(y0, 0), fh - 1), yc1 = min(max(y0 + 1, 0), fh - 1);
  Taps t;
  t.o00 = (yc0 * fw + xc0) * C; t.o01 = (yc0 * fw + xc1) * C;
  t.o10 = (yc1 * fw + xc0) * C; t.o11 = (yc1 * fw + xc1) * C;
  t.w00 = (yv0 && xv0) ? wx0 * wy0 : 0.f; t.w01 = (yv0 && xv1) ? wx1 * wy0 : 0.f;
  t.w10 = (yv1 && xv0) ? wx0 * wy1 : 0.f; t.w11 = (yv1 && xv1) ? wx1 * wy1 : 0.f;
  return t;
}

// the footprint computed by lane `src` of this half-wave
__device__ __forceinline__ Taps taps_from(const Taps& t, int src) {
  Taps r;
  r.o00 = __shfl(t.o00, src, 32); r.o01 = __shfl(t.o01, src, 32);
  r.o10 = __shfl(t.o10, src, 32); r.o11 = __shfl(t.o11, src, 32);
  r.w00 = __shfl(t.w00, src, 32); r.w01 = __shfl(t.w01, src, 32);
  r.w10 = __shfl(t.w10, src, 32); r.w11 = __shfl(t.w11, src, 32);
  return r;
}

// unconditional, clamped loads (branches around loads serialise them); out-of-range taps carry weight 0
__device__ __forceinline__ f32x4 gather_view(const float* __restrict__ fmap, const Taps& t, int c) {
  const f32x4 v00 = *reinterpret_cast<const f32x4*>(fmap + t.o00 + c);
  const f32x4 v01 = *reinterpret_cast<const f32x4*>(fmap + t.o01 + c);
  const f32x4 v10 = *reinterpret_cast<const f32x4*>(fmap + t.o10 + c);
  const f32x4 v11 = *reinterpret_cast<const f32x4*>(fmap + t.o11 + c);
  f32x4 acc = t.w00 * v00;
  acc += t.w01 * v01;
  acc += t.w10 * v10;
  acc += t.w11 * v11;
  return acc;
}

// Projections: either projs [rfn+1][3][4] (K @ pose per view, query last), or — projs == NULL — intrinsics and poses given
// separately (ref_Ks [rfn][3][3], ref_poses [rfn][3][4], K_in [3][3], pose_in [3][4]) and multiplied here, one view per lane;
// rot = 3x3 with row stride rot_ld (3: dense; 4: the rotation part of pose_in in place).
struct VolViews { const float* projs; const float* ref_Ks; const float* ref_poses; const float* K_in; const float* pose_in;
                  const float* rot; int rot_ld; };

__global__ void __launch_bounds__(256, 4) refiner_volume_kernel(const float* __restrict__ feats, const VolViews vw,
                                                             const float* __restrict__ lin, int rfn, int fh, int fw,
                                                             int C, float h_in, float w_in, int sn,
                                                             float* __restrict__ mean_in, float* __restrict__ stdv) {
  // blockIdx.y = query of the batch (g6d_refiner_volume_kp): its views, cameras and volumes follow those of the previous query
  const int bq = blockIdx.y;
  const float* __restrict__ rot = vw.rot + bq * 12;
  const int rl = vw.rot_ld;
  const int nvox = sn * sn * sn;
  feats += (size_t)bq * (rfn + 1) * fh * fw * C;
  mean_in += (size_t)bq * nvox * 2 * C; stdv += (size_t)bq * nvox * C;
  int half = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;         // one half-wave per voxel
  const int l32 = threadIdx.x & 31;
  const bool live = half < nvox;                                   // (whole half-waves; keep them for the shuffles)
  if (!live) half = nvox - 1;
  const int k = half % sn, j = (half / sn) % sn, i = half / (sn * sn);
  const float g0 = lin[i], g1 = lin[j], g2 = lin[k];
  const float vx = g0 * rot[0] + g1 * rot[rl] + g2 * rot[2 * rl];
  const float vy = g0 * rot[1] + g1 * rot[rl + 1] + g2 * rot[2 * rl + 1];
  const float vz = g0 * rot[2] + g1 * rot[rl + 2] + g2 * rot[2 * rl + 2];
  const size_t fsz = (size_t)fh * fw * C;
  const float inv_n = 1.f / (float)rfn, inv_n1 = 1.f / (float)(rfn > 1 ? rfn - 1 : 1);
  // lane v of the half-wave projects the voxel into view v (views 0..rfn-1 = references, view rfn = query); every
  // lane then fetches the 8 footprint values of each view with shuffles instead of redoing the 9 projections
  const int myview = l32 <= rfn ? l32 : rfn;
  float P[12];
  if (vw.projs) {
#pragma unroll
    for (int e = 0; e < 12; ++e) P[e] = vw.projs[myview * 12 + e];
  } else {
    const float* K = myview < rfn ? vw.ref_Ks + (bq * rfn + myview) * 9 : vw.K_in + bq * 9;
    const float* T = myview < rfn ? vw.ref_poses + (bq * rfn + myview) * 12 : vw.pose_in + bq * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) P[r * 4 + c] = K[r * 3] * T[c] + K[r * 3 + 1] * T[4 + c] + K[r * 3 + 2] * T[8 + c];
  }
  const Taps mine = project_view(fh, fw, C, P, vx, vy, vz, h_in, w_in);
  for (int c = l32 * 4; c < C; c += 128) {           // (uniform trip count per half-wave pair: C is a kernel argument)
    f32x4 s[MAX_RFN];
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < MAX_RFN; ++r)
      if (r < rfn) { s[r] = gather_view(feats + r * fsz, taps_from(mine, r), c); sum += s[r]; }
    const f32x4 mean = sum * inv_n;
    f32x4 var = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < MAX_RFN; ++r)
      if (r < rfn) { f32x4 d = s[r] - mean; var += d * d; }
    var = var * inv_n1;
    f32x4 sd = {sqrtf(var[0]), sqrtf(var[1]), sqrtf(var[2]), sqrtf(var[3])};
    const f32x4 q = gather_view(feats + rfn * fsz, taps_from(mine, rfn), c);
    if (!live) continue;
    *reinterpret_cast<f32x4*>(mean_in + (size_t)half * 2 * C + c) = mean;
    *reinterpret_cast<f32x4*>(mean_in + (size_t)half * 2 * C + C + c) = q;
    *reinterpret_cast<f32x4*>(stdv + (size_t)half * C + c) = sd;
  }
}

}  // namespace

extern "C" int g6d_refiner_volume(const float* feats, const float* projs, const float* rot_in,
                                  const float* lin, int rfn, int fh, int fw, int C, int h_in, int w_in, int sn,
                                  float* mean_in, float* stdv, g6d_stream_t stream) {
  if (!feats || !projs || !rot_in || !lin || !mean_in || !stdv || rfn < 1 || rfn > MAX_RFN || (C & 3) ||
      sn < 1 || sn > 256 || !g6d_aligned16(feats) || !g6d_aligned16(mean_in) || !g6d_aligned16(stdv)) {
    g6d_set_error("refiner_volume: bad args (1 <= rfn <= 8, C % 4 == 0)"); return G6D_EINVAL;
  }
  const long long threads = (long long)sn * sn * sn * 32;
  const VolViews vw = {projs, nullptr, nullptr, nullptr, nullptr, rot_in, 3};
  hipLaunchKernelGGL(refiner_volume_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), feats, vw, lin, rfn, fh, fw, C, (float)h_in, (float)w_in, sn,
                     mean_in, stdv);
  return g6d_check_launch("refiner_volume");
}

// The same with intrinsics and poses given separately: the projections ref_Ks[r] @ ref_poses[r], K_in @ pose_in of
// network/refiner.py:208-226 are formed inside the kernel (one view per lane), and the volume is rotated by the rotation part of
// pose_in read in place — no matrix-product, concatenation or copy launches in front of the kernel.
extern "C" int g6d_refiner_volume_kp(const float* feats, const float* ref_Ks, const float* ref_poses, const float* K_in,
                                     const float* pose_in, const float* lin, int rfn, int fh, int fw, int C, int h_in, int w_in, int sn,
                                     float* mean_in, float* stdv, int batch, g6d_stream_t stream) {
  if (!feats || !ref_Ks || !ref_poses || !K_in || !pose_in || !lin || !mean_in || !stdv || rfn < 1 || rfn > MAX_RFN || (C & 3) ||
      sn < 1 || sn > 256 || batch < 1 || batch > 65535 || !g6d_aligned16(feats) || !g6d_aligned16(mean_in) || !g6d_aligned16(stdv)) {
    g6d_set_error("refiner_volume_kp: bad args (1 <= rfn <= 8, C % 4 == 0)"); return G6D_EINVAL;
  }
  const long long threads = (long long)sn * sn * sn * 32;
  const VolViews vw = {nullptr, ref_Ks, ref_poses, K_in, pose_in, pose_in, 4};
  hipLaunchKernelGGL(refiner_volume_kernel, dim3((unsigned)((threads + 255) / 256), batch), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), feats, vw, lin, rfn, fh, fw, C, (float)h_in, (float)w_in, sn,
                     mean_in, stdv);
  return g6d_check_launch("refiner_volume_kp");
}
