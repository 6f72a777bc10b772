// Device-resident glue between the stages of Gen6DEstimator.predict: the 3x4 pose algebra and the
// warp parameters that the reference computes on the host between network calls (estimator.py:173-216,
// network/refiner.py:275-341, utils/pose_utils.py, utils/database_utils.py) as tiny single-thread float64 kernels reading
// and writing device buffers, plus a batched image warp that takes its homographies from device memory.  With these,
// detect -> crop -> select -> pose -> 3 x (look-at crop, reference selection + alignment, refiner, pose update) is one
// chain of launches without a host synchronisation, i.e. capturable in a hipGraph and runnable for several queries at
// once.  The arithmetic lives in pose_algebra.h, which also builds for the host (tests/test_pose_chain_cpu.py).
#include "g6d_common.h"
#include "pose_algebra.h"

namespace {

using namespace pa;

__device__ __forceinline__ M3 ld_m3(const float* p) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = p[i]; return r; }
__device__ __forceinline__ P34 ld_p34(const float* p) { P34 r; for (int i = 0; i < 12; ++i) r.m[i] = p[i]; return r; }
__device__ __forceinline__ void st_m3(float* p, const M3& a) { for (int i = 0; i < 9; ++i) p[i] = (float)a.m[i]; }
__device__ __forceinline__ void st_p34(float* p, const P34& a) { for (int i = 0; i < 12; ++i) p[i] = (float)a.m[i]; }

// detection -> inverse crop transform (estimator.py:184 transformation_crop(que_img, position, 1/scale_r2q, 0, size))
// Every chain kernel takes a batch of queries: blockIdx.x = query; per-query operands and results are dense arrays with the query as
// leading axis, the reference state (poses, intrinsics, centre, normalisation) is shared.
__global__ void crop_from_detection_kernel(const float* __restrict__ det, float size, float* __restrict__ hinv) {
  if (threadIdx.x != 0) return;
  det += 5 * blockIdx.x; hinv += 9 * blockIdx.x;
  st_m3(hinv, inv3(crop_transform(det[0], det[1], 1.0 / (double)det[2], 0.0, size)));
}

// arg-max viewpoint (first maximum) + estimate_pose_from_similarity_transform_compose (estimator.py:193-206)
__global__ void pose_from_selection_kernel(const float* __restrict__ det, const float* __restrict__ logits,
                                           const float* __restrict__ angles, int rfn, const float* __restrict__ ref_poses,
                                           const float* __restrict__ ref_Ks, const float* __restrict__ que_K,
                                           const float* __restrict__ center, float* __restrict__ pose_out,
                                           float* __restrict__ sel_out) {
  if (threadIdx.x != 0) return;
  const int q = blockIdx.x;
  det += 5 * q; logits += (size_t)rfn * q; angles += (size_t)rfn * q; que_K += 9 * q; pose_out += 12 * q; sel_out += 2 * q;
  int best = 0;
  for (int r = 1; r < rfn; ++r) if (logits[r] > logits[best]) best = r;
  const V3 c{center[0], center[1], center[2]};
  st_p34(pose_out, pose_from_similarity(det[0], det[1], det[2], angles[best], ld_p34(ref_poses + 12 * best), ld_m3(ref_Ks + 9 * best),
                                        ld_m3(que_K), c));
  sel_out[0] = (float)best; sel_out[1] = angles[best];
}

// Geometry record of one refinement step (floats): K_warp[9] | pose_warp[12] | pose_rect[12] | ref_Ks[R][9] | ref_poses[R][12] |
// hinv[1+R][9] (query first)                                                    -> G6D_REFINE_GEO_FLOATS(R) = 42 + 30 R
__global__ void __launch_bounds__(128) refine_prepare_kernel(const float* __restrict__ pose_in, const float* __restrict__ que_K,
                                                            const float* __restrict__ norm, float size, float margin,
                                                            const float* __restrict__ sub_poses, const float* __restrict__ sub_Ks,
                                                            int n_sub, int ref_num, float* __restrict__ geo, int* __restrict__ ref_idx,
                                                            float angle_step, int* __restrict__ ref_bucket) {
  __shared__ double corr[128];
  __shared__ RefinePrep g;
  __shared__ int sel[8];
  const int t = threadIdx.x;
  {
    const int q = blockIdx.x;                                    // query of the batch
    pose_in += 12 * q; que_K += 9 * q; geo += (size_t)(42 + 30 * ref_num) * q; ref_idx += ref_num * q;
    if (ref_bucket) ref_bucket += ref_num * q;
  }
  const V3 noff{norm[1], norm[2], norm[3]};
  if (t == 0) {
    g = refine_prepare(ld_p34(pose_in), ld_m3(que_K), norm[0], noff, size, margin);
    st_m3(geo, g.K_warp); st_p34(geo + 9, g.pose_warp); st_p34(geo + 21, g.pose_rect);
    st_m3(geo + 33 + 21 * ref_num, inv3(g.H));
  }
  __syncthreads();
  // select_reference_img_ids_refinement (database_utils.py:125-139): the ref_num views most aligned with the warped pose
  if (t < n_sub) corr[t] = view_cos(g.pose_warp, ld_p34(sub_poses + 12 * t));
  __syncthreads();
  if (t == 0) {
    for (int k = 0; k < ref_num; ++k) {
      int best = -1;
      for (int i = 0; i < n_sub; ++i) {
        bool used = false;
        for (int j = 0; j < k; ++j) used |= sel[j] == i;
        if (!used && (best < 0 || corr[i] > corr[best])) best = i;
      }
      sel[k] = best; ref_idx[k] = best;
    }
  }
  __syncthreads();
  // normalize_reference_views aligned with the input pose: one thread per selected view
  if (t < ref_num) {
    const int i = sel[t];
    M3 K_new, H; P34 pose_new;
    int bucket = 0;
    align_reference(ld_p34(sub_poses + 12 * i), ld_m3(sub_Ks + 9 * i), g.pose_warp, g.K_warp, size, margin, K_new, pose_new, H, angle_step, &bucket);
    if (ref_bucket) ref_bucket[t] = bucket;
    st_m3(geo + 33 + 9 * t, K_new);
    st_p34(geo + 33 + 9 * ref_num + 12 * t, pose_new);
    st_m3(geo + 33 + 21 * ref_num + 9 * (1 + t), inv3(H));
  }
}

// refiner outputs -> refined pose (refiner.py:327-341)
__global__ void refine_update_kernel(const float* __restrict__ rot, const float* __restrict__ off, const float* __restrict__ scl_,
                                     const float* __restrict__ geo, int geo_stride, const float* __restrict__ norm,
                                     float* __restrict__ pose_out) {
  if (threadIdx.x != 0) return;
  rot += 4 * blockIdx.x; off += 2 * blockIdx.x; scl_ += blockIdx.x; geo += (size_t)geo_stride * blockIdx.x; pose_out += 12 * blockIdx.x;
  RefinePrep g;
  g.K_warp = ld_m3(geo); g.pose_warp = ld_p34(geo + 9); g.pose_rect = ld_p34(geo + 21);
  const double q[4] = {rot[0], rot[1], rot[2], rot[3]};
  st_p34(pose_out, refine_update(q, off[0], off[1], scl_[0], g, norm[0], V3{norm[1], norm[2], norm[3]}));
}

// Batched projective warp, homographies and source selection in DEVICE memory: image b of the batch is
//   dst[b][c][y][x] = rint( bilinear( src_b, hinv[b] * (x, y, 1) ) ) / 255     (NCHW float in [0,1], as the networks take it;
// the rounding is the uint8 image the reference's cv2.warpPerspective would have produced), with
//   src_b = idx[b] < 0 ? single : stack + idx[b] * sh * sw * ch.
__global__ void __launch_bounds__(256) warp_batch_kernel(const unsigned char* __restrict__ stack, const unsigned char* __restrict__ single,
                                                        const int* __restrict__ idx, int sh, int sw, int ch,
                                                        const float* __restrict__ hinv, float* __restrict__ dst, int dh, int dw) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= dh * dw) return;
  const int sel = idx ? idx[b] : -1;
  const unsigned char* src = sel < 0 ? single : stack + (size_t)sel * sh * sw * ch;
  const float* h = hinv + 9 * b;
  const int x = i % dw, y = i / dw;
  const float X = h[0] * x + h[1] * y + h[2], Y = h[3] * x + h[4] * y + h[5], Wd = h[6] * x + h[7] * y + h[8];
  const float iw = Wd != 0.f ? 1.f / Wd : 0.f;
  float fx = fminf(fmaxf(X * iw, -4.f), (float)sw + 4.f), fy = fminf(fmaxf(Y * iw, -4.f), (float)sh + 4.f);
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
    dst[((size_t)(b * ch + c) * dh + y) * dw + x] = fminf(fmaxf(rintf(v), 0.f), 255.f) * (1.f / 255.f);
  }
}

}  // namespace

#define CHAIN_STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int g6d_chain_crop_from_detection(const float* det, float size, float* hinv, int batch, g6d_stream_t stream) {
  if (!det || !hinv || size <= 0 || batch < 1) { g6d_set_error("chain_crop_from_detection: bad args"); return G6D_EINVAL; }
  hipLaunchKernelGGL(crop_from_detection_kernel, dim3(batch), dim3(64), 0, CHAIN_STREAM(stream), det, size, hinv);
  return g6d_check_launch("chain_crop_from_detection");
}

extern "C" int g6d_chain_pose_from_selection(const float* det, const float* logits, const float* angles, int rfn,
                                             const float* ref_poses, const float* ref_Ks, const float* que_K, const float* center,
                                             float* pose_out, float* sel_out, int batch, g6d_stream_t stream) {
  if (!det || !logits || !angles || rfn <= 0 || !ref_poses || !ref_Ks || !que_K || !center || !pose_out || !sel_out || batch < 1) {
    g6d_set_error("chain_pose_from_selection: bad args"); return G6D_EINVAL;
  }
  hipLaunchKernelGGL(pose_from_selection_kernel, dim3(batch), dim3(64), 0, CHAIN_STREAM(stream), det, logits, angles, rfn, ref_poses,
                     ref_Ks, que_K, center, pose_out, sel_out);
  return g6d_check_launch("chain_pose_from_selection");
}

extern "C" int g6d_chain_refine_prepare(const float* pose_in, const float* que_K, const float* norm, float size, float margin,
                                        const float* sub_poses, const float* sub_Ks, int n_sub, int ref_num, float* geo,
                                        int* ref_idx, float angle_step, int* ref_bucket, int batch, g6d_stream_t stream) {
  if (!pose_in || !que_K || !norm || size <= 0 || !sub_poses || !sub_Ks || n_sub <= 0 || n_sub > 128 || ref_num <= 0 ||
      ref_num > 8 || ref_num > n_sub || !geo || !ref_idx || batch < 1) {
    g6d_set_error("chain_refine_prepare: bad args (n_sub <= 128, ref_num <= 8)"); return G6D_EINVAL;
  }
  hipLaunchKernelGGL(refine_prepare_kernel, dim3(batch), dim3(128), 0, CHAIN_STREAM(stream), pose_in, que_K, norm, size, margin, sub_poses,
                     sub_Ks, n_sub, ref_num, geo, ref_idx, angle_step, ref_bucket);
  return g6d_check_launch("chain_refine_prepare");
}

extern "C" int g6d_chain_refine_update(const float* rot, const float* off, const float* scl, const float* geo, int geo_floats,
                                       const float* norm, float* pose_out, int batch, g6d_stream_t stream) {
  if (!rot || !off || !scl || !geo || !norm || !pose_out || batch < 1 || geo_floats < 33) { g6d_set_error("chain_refine_update: bad args"); return G6D_EINVAL; }
  hipLaunchKernelGGL(refine_update_kernel, dim3(batch), dim3(64), 0, CHAIN_STREAM(stream), rot, off, scl, geo, geo_floats, norm, pose_out);
  return g6d_check_launch("chain_refine_update");
}

extern "C" int g6d_warp_batch(const unsigned char* stack, const unsigned char* single, const int* idx, int B, int sh, int sw, int ch,
                              const float* hinv, float* dst, int dh, int dw, g6d_stream_t stream) {
  if ((!stack && !single) || B <= 0 || sh <= 0 || sw <= 0 || ch <= 0 || ch > 4 || !hinv || !dst || dh <= 0 || dw <= 0 ||
      (!idx && !single)) {
    g6d_set_error("warp_batch: bad args"); return G6D_EINVAL;
  }
  hipLaunchKernelGGL(warp_batch_kernel, dim3((dh * dw + 255) / 256, B), dim3(256), 0, CHAIN_STREAM(stream), stack, single, idx, sh, sw,
                     ch, hinv, dst, dh, dw);
  return g6d_check_launch("warp_batch");
}
