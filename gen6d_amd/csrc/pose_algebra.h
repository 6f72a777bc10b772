// Pose / 2-D similarity algebra between the stages of Gen6DEstimator.predict, written once for the device kernels of
// pose_chain.hip and for the host build that tests/test_pose_chain_cpu.py checks against gen6d_amd/geometry.py and the
// golden vectors of the reference's own utils (tests/golden/geometry.npz).  float64 throughout, a few hundred flops per
// query.  Each function names the reference function whose contract it keeps.
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define PA_HD __host__ __device__ __forceinline__
#else
#define PA_HD inline
#endif

namespace pa {

struct M3 { double m[9]; };     // row-major 3x3
struct P34 { double m[12]; };   // row-major [R|t]
struct V3 { double x, y, z; };

PA_HD M3 eye3() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
PA_HD M3 mul(const M3& a, const M3& b) {
  M3 c;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
  return c;
}
PA_HD M3 tr(const M3& a) { return M3{{a.m[0], a.m[3], a.m[6], a.m[1], a.m[4], a.m[7], a.m[2], a.m[5], a.m[8]}}; }
PA_HD V3 mulv(const M3& a, const V3& v) {
  return V3{a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
            a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
PA_HD double det3(const M3& a) {
  return a.m[0] * (a.m[4] * a.m[8] - a.m[5] * a.m[7]) - a.m[1] * (a.m[3] * a.m[8] - a.m[5] * a.m[6]) +
         a.m[2] * (a.m[3] * a.m[7] - a.m[4] * a.m[6]);
}
PA_HD M3 inv3(const M3& a) {
  const double d = 1.0 / det3(a);
  M3 r;
  r.m[0] = (a.m[4] * a.m[8] - a.m[5] * a.m[7]) * d; r.m[1] = (a.m[2] * a.m[7] - a.m[1] * a.m[8]) * d; r.m[2] = (a.m[1] * a.m[5] - a.m[2] * a.m[4]) * d;
  r.m[3] = (a.m[5] * a.m[6] - a.m[3] * a.m[8]) * d; r.m[4] = (a.m[0] * a.m[8] - a.m[2] * a.m[6]) * d; r.m[5] = (a.m[2] * a.m[3] - a.m[0] * a.m[5]) * d;
  r.m[6] = (a.m[3] * a.m[7] - a.m[4] * a.m[6]) * d; r.m[7] = (a.m[1] * a.m[6] - a.m[0] * a.m[7]) * d; r.m[8] = (a.m[0] * a.m[4] - a.m[1] * a.m[3]) * d;
  return r;
}
PA_HD double norm3(const V3& v) { return sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
PA_HD V3 sub(const V3& a, const V3& b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PA_HD V3 add(const V3& a, const V3& b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PA_HD V3 scl(const V3& a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
PA_HD double f32(double v) { return (double)(float)v; }          // where the reference rounds to float32 on the way

PA_HD M3 rot_of(const P34& p) { return M3{{p.m[0], p.m[1], p.m[2], p.m[4], p.m[5], p.m[6], p.m[8], p.m[9], p.m[10]}}; }
PA_HD V3 trans_of(const P34& p) { return V3{p.m[3], p.m[7], p.m[11]}; }
PA_HD P34 make_pose(const M3& R, const V3& t) {
  return P34{{R.m[0], R.m[1], R.m[2], t.x, R.m[3], R.m[4], R.m[5], t.y, R.m[6], R.m[7], R.m[8], t.z}};
}
// base_utils.py:502-505
PA_HD P34 pose_inverse(const P34& p) { const M3 Rt = tr(rot_of(p)); return make_pose(Rt, scl(mulv(Rt, trans_of(p)), -1.0)); }
// base_utils.py:512-521: apply p0 first, then p1
PA_HD P34 pose_compose(const P34& p0, const P34& p1) {
  const M3 R1 = rot_of(p1);
  return make_pose(mul(R1, rot_of(p0)), add(mulv(R1, trans_of(p0)), trans_of(p1)));
}
PA_HD V3 pose_apply(const P34& p, const V3& x) { return add(mulv(rot_of(p), x), trans_of(p)); }
// base_utils.py:256-265 for one point: pixel (u, v) and depth; 0 < |depth| < 1e-4 clamped to 1e-4
PA_HD void project_point(const V3& x, const P34& p, const M3& K, double& u, double& v, double& depth) {
  const V3 c = mulv(K, pose_apply(p, x));
  double d = c.z;
  if (fabs(d) < 1e-4 && fabs(d) > 0) d = 1e-4;
  u = c.x / d; v = c.y / d; depth = d;
}
// dataset/database.py:399-410 (results are float32 in the reference)
PA_HD P34 normalize_pose(const P34& p, double scale, const V3& off) {
  const M3 R = rot_of(p);
  const V3 t = add(mulv(R, scl(off, -1.0)), scl(trans_of(p), scale));
  P34 r = make_pose(R, t);
  for (int i = 0; i < 12; ++i) r.m[i] = f32(r.m[i]);
  return r;
}
PA_HD P34 denormalize_pose(const P34& p, double scale, const V3& off) {
  const M3 R = rot_of(p);
  const V3 t = add(scl(mulv(R, off), 1.0 / scale), scl(trans_of(p), 1.0 / scale));
  P34 r = make_pose(R, t);
  for (int i = 0; i < 12; ++i) r.m[i] = f32(r.m[i]);
  return r;
}

PA_HD M3 rot_x(double a) { const double c = cos(a), s = sin(a); return M3{{1, 0, 0, 0, c, -s, 0, s, c}}; }
PA_HD M3 rot_y(double a) { const double c = cos(a), s = sin(a); return M3{{c, 0, s, 0, 1, 0, -s, 0, c}}; }
PA_HD M3 rot_z(double a) { const double c = cos(a), s = sin(a); return M3{{c, -s, 0, s, c, 0, 0, 0, 1}}; }
// base_utils.py:657-666: euler2mat(atan y,'sxyz') @ euler2mat(-atan x,'syxz')
PA_HD M3 look_at_rotation(double x, double y) { return mul(rot_x(atan2(y, 1.0)), rot_y(-atan2(x, 1.0))); }
// first angle of transforms3d.mat2euler(R, 'szyx') (pose_utils.py:96-99)
PA_HD double angle_about_z(const M3& R) {
  const double cy = sqrt(R.m[8] * R.m[8] + R.m[5] * R.m[5]);
  if (cy > 2.220446049250313e-16 * 4.0) return -atan2(R.m[1], R.m[0]);
  return atan2(R.m[3], R.m[4]);
}
// transforms3d.quaternions.quat2mat (w, x, y, z), used at pose_utils.py:239
PA_HD M3 quat2mat(double w, double x, double y, double z) {
  const double n = w * w + x * x + y * y + z * z;
  if (n < 2.220446049250313e-16) return eye3();
  const double s = 2.0 / n, X = x * s, Y = y * s, Z = z * s;
  const double wX = w * X, wY = w * Y, wZ = w * Z, xX = x * X, xY = x * Y, xZ = x * Z, yY = y * Y, yZ = y * Z, zZ = z * Z;
  return M3{{1.0 - (yY + zZ), xY - wZ, xZ + wY, xY + wZ, 1.0 - (xX + zZ), yZ - wX, xZ - wY, yZ + wX, 1.0 - (xX + yY)}};
}

// ---- 2-D similarities stored as 3x3 with last row (0,0,1)
PA_HD M3 sim2d(double scale, double angle, double ox, double oy) {
  const double c = cos(angle) * scale, s = sin(angle) * scale;
  return M3{{c, -s, ox, s, c, oy, 0, 0, 1}};
}
// transformation_crop's M (base_utils.py:646-653): centre on `position`, scale, rotate, move to the crop centre
PA_HD M3 crop_transform(double px, double py, double scale, double angle, double size) {
  M3 m = sim2d(1.0, 0.0, -px, -py);
  m = mul(sim2d(scale, 0.0, 0, 0), m);
  m = mul(sim2d(1.0, angle, 0, 0), m);
  return mul(sim2d(1.0, 0.0, size / 2, size / 2), m);
}

// pose_utils.py:55-61
PA_HD void let_me_look_at_2d(double cx, double cy, const M3& K, M3& R, double& f) {
  const double f_raw = (K.m[0] + K.m[4]) / 2;
  const double x = cx - K.m[2], y = cy - K.m[5];
  R = look_at_rotation(x / f_raw, y / f_raw);
  f = sqrt(x * x + y * y + f_raw * f_raw);
}
// pose_utils.py:51-53
PA_HD void let_me_look_at(const P34& pose, const M3& K, const V3& center, M3& R, double& f) {
  double u, v, d;
  project_point(center, pose, K, u, v, d);
  let_me_look_at_2d(u, v, K, R, f);
}
// database_utils.py:8-25 without the image warp: K_new, pose_new, pose_rect, H (source -> crop homography)
PA_HD void look_at_crop_params(const M3& K, const P34& pose, double posx, double posy, double angle, double scale, double h,
                               double w, M3& K_new, P34& pose_new, P34& pose_rect, M3& H) {
  M3 R_new; double f_new;
  let_me_look_at_2d(posx, posy, K, R_new, f_new);
  R_new = mul(rot_z(angle), R_new);
  f_new *= scale;
  K_new = M3{{f32(f_new), 0, f32(w / 2), 0, f32(f_new), f32(h / 2), 0, 0, 1}};
  H = mul(mul(K_new, R_new), inv3(K));
  M3 Rr = R_new;
  for (int i = 0; i < 9; ++i) Rr.m[i] = f32(Rr.m[i]);
  pose_rect = make_pose(Rr, V3{0, 0, 0});
  pose_new = pose_compose(pose, pose_rect);
}
// pose_utils.py:63-102 for one (reference, query) pair
PA_HD void scale_rotation_difference(const P34& ref_pose, const P34& que_pose, const M3& ref_K, const M3& que_K, const V3& center,
                                     double& scale, double& angle) {
  M3 Rr, Rq; double fr, fq;
  let_me_look_at(ref_pose, ref_K, center, Rr, fr);
  let_me_look_at(que_pose, que_K, center, Rq, fq);
  const M3 ref_rot = mul(Rr, rot_of(ref_pose)), que_rot = mul(Rq, rot_of(que_pose));
  const double ref_dist = norm3(sub(trans_of(pose_inverse(ref_pose)), center));
  const double que_dist = norm3(sub(trans_of(pose_inverse(que_pose)), center));
  scale = ref_dist / que_dist * fq / fr;
  angle = angle_about_z(mul(que_rot, tr(ref_rot)));
}

// pose_utils.py:104-111 -> :12-49: query pose from detection (position, scale) + selected view and in-plane angle
PA_HD P34 pose_from_similarity(double px, double py, double scale_r2q, double angle_r2q, const P34& ref_pose, const M3& ref_K,
                               const M3& que_K, const V3& center) {
  double rcx, rcy, rd;
  project_point(center, ref_pose, ref_K, rcx, rcy, rd);
  M3 m = sim2d(1.0, 0.0, -px, -py);
  m = mul(sim2d(1.0 / scale_r2q, 0.0, 0, 0), m);
  m = mul(sim2d(1.0, -angle_r2q, 0, 0), m);
  const M3 m_q2r = mul(sim2d(1.0, 0.0, rcx, rcy), m);
  const M3 m_r2q = inv3(m_q2r);
  const V3 ref_cam = trans_of(pose_inverse(ref_pose));
  const V3 qc = mulv(m_r2q, V3{rcx, rcy, 1.0});
  V3 qn = mulv(inv3(que_K), V3{qc.x, qc.y, 1.0});
  qn.x /= qn.z; qn.y /= qn.z;
  const double scale = sqrt(m_r2q.m[0] * m_r2q.m[4] - m_r2q.m[1] * m_r2q.m[3]);
  const double rotation = atan2(m_r2q.m[3], m_r2q.m[0]);
  const double que_f = (que_K.m[0] + que_K.m[4]) / 2, ref_f = (ref_K.m[0] + ref_K.m[4]) / 2;
  const double que_f_ = sqrt(que_f * que_f + (qn.x * qn.x + qn.y * qn.y) * que_f * que_f);
  const double que_dist = norm3(sub(ref_cam, center)) * que_f_ / ref_f / scale;
  const V3 ray{qn.x, qn.y, 1.0};
  const V3 cen3d = scl(ray, que_dist / norm3(ray));
  const M3 que_rot = mul(tr(look_at_rotation(qn.x, qn.y)), mul(rot_z(rotation), rot_of(ref_pose)));
  return make_pose(que_rot, sub(cen3d, mulv(que_rot, center)));
}

// pose_utils.py:237-244
PA_HD P34 compose_sim_pose(double scale, const double quat[4], double offx, double offy, const P34& in_pose, const V3& center) {
  const M3 rotation = quat2mat(quat[0], quat[1], quat[2], quat[3]);
  const V3 cin = pose_apply(in_pose, center);
  const V3 cque{cin.x + offx, cin.y + offy, cin.z};
  M3 A = rotation;
  for (int i = 0; i < 9; ++i) A.m[i] *= scale;
  return make_pose(A, sub(cque, mulv(A, cin)));
}
// Orthogonal polar factor U V^T of A (= the `U @ Vt` of np.linalg.svd, reflections included) and the mean singular value,
// by Newton's iteration X <- (X + X^-T) / 2: A = (U V^T)(V S V^T), so trace((U V^T)^T A) = sum(S).
PA_HD void polar3(const M3& A, M3& Q, double& mean_sv) {
  Q = A;
  for (int it = 0; it < 60; ++it) {
    const M3 Xit = tr(inv3(Q));
    double diff = 0;
    M3 N;
    for (int i = 0; i < 9; ++i) { N.m[i] = 0.5 * (Q.m[i] + Xit.m[i]); diff += fabs(N.m[i] - Q.m[i]); }
    Q = N;
    if (diff < 1e-15) break;
  }
  const M3 P = mul(tr(Q), A);
  mean_sv = (P.m[0] + P.m[4] + P.m[8]) / 3.0;
}
// pose_utils.py:217-235
PA_HD P34 pose_sim_to_pose_rigid(const P34& sim, const P34& pose_in, const M3& K_que, const M3& K_in, const V3& center) {
  const double f_que = (K_que.m[0] + K_que.m[4]) / 2, f_in = (K_in.m[0] + K_in.m[4]) / 2;
  const V3 cin = pose_apply(pose_in, center);
  M3 Q; double msv;
  polar3(rot_of(sim), Q, msv);
  const double depth_que = cin.z / msv * f_que / f_in;
  const V3 csim = pose_apply(sim, cin);
  const V3 cque = scl(csim, depth_que / csim.z);
  const M3 rotation = mul(Q, rot_of(pose_in));
  return make_pose(rotation, sub(cque, mulv(rotation, center)));
}

// ---- one refinement step, before the network (reference network/refiner.py:275-313 with the database normalised):
//      pose_in is in the DATABASE frame; (nscale, noff) = NormalizedDatabase.scale / .offset; centre 0, diameter 2.
struct RefinePrep { M3 K_warp; P34 pose_warp, pose_rect; M3 H; };
PA_HD RefinePrep refine_prepare(const P34& pose_in_db, const M3& que_K, double nscale, const V3& noff, double size, double margin) {
  const V3 center{0, 0, 0};
  const P34 in_pose = normalize_pose(pose_in_db, nscale, noff);
  M3 Rl; double new_f;
  let_me_look_at(in_pose, que_K, center, Rl, new_f);
  const double in_dist = norm3(sub(trans_of(pose_inverse(in_pose)), center));
  const double scale = size * (1 - margin) / 2.0 * in_dist / new_f;
  double px, py, pd;
  project_point(center, in_pose, que_K, px, py, pd);
  RefinePrep r;
  look_at_crop_params(que_K, in_pose, px, py, 0.0, scale, size, size, r.K_warp, r.pose_warp, r.pose_rect, r.H);
  return r;
}
// One reference view aligned with the warped query (utils/database_utils.py:54-110, rectify_rot with input pose/K)
// angle_step > 0 (round 3, reference-feature caching): the in-plane angle is snapped to multiples of angle_step (radians), so that
// the aligned crop of a view is a function of (view, bucket) only and its features can be cached across refinement steps and queries;
// *bucket receives round(angle / angle_step) (0 when angle_step <= 0).  angle_step = 0 is the reference's exact alignment.
PA_HD void align_reference(const P34& ref_pose, const M3& ref_K, const P34& pose_warp, const M3& K_warp, double size, double margin,
                           M3& K_new, P34& pose_new, M3& H, double angle_step = 0.0, int* bucket = nullptr) {
  const V3 center{0, 0, 0};
  double cx, cy, cd;
  project_point(center, ref_pose, ref_K, cx, cy, cd);
  const double dist = norm3(sub(trans_of(pose_inverse(ref_pose)), center));
  M3 Rl; double f_look;
  let_me_look_at(ref_pose, ref_K, center, Rl, f_look);
  const double scale = size * (1 - margin) / 2.0 * dist / f_look;
  double s, angle;
  scale_rotation_difference(ref_pose, pose_warp, ref_K, K_warp, center, s, angle);
  int b = 0;
  if (angle_step > 0) { b = (int)floor(angle / angle_step + 0.5); angle = b * angle_step; }
  if (bucket) *bucket = b;
  P34 rect;
  look_at_crop_params(ref_K, ref_pose, cx, cy, angle, scale, size, size, K_new, pose_new, rect, H);
}
// After the network (refiner.py:327-341): (quaternion, offset, log2 scale) -> refined pose in the database frame
PA_HD P34 refine_update(const double quat[4], double offx, double offy, double log2_scale, const RefinePrep& g, double nscale,
                        const V3& noff) {
  const V3 center{0, 0, 0};
  const P34 sim = compose_sim_pose(exp2(log2_scale), quat, offx, offy, g.pose_warp, center);
  P34 pr = pose_sim_to_pose_rigid(sim, g.pose_warp, g.K_warp, g.K_warp, center);
  pr = pose_compose(pr, pose_inverse(g.pose_rect));
  return denormalize_pose(pr, nscale, noff);
}
// cosine between camera directions seen from the (normalised) object centre (database_utils.py:27-52)
PA_HD double view_cos(const P34& a, const P34& b) {
  const V3 ca = trans_of(pose_inverse(a)), cb = trans_of(pose_inverse(b));
  return (ca.x * cb.x + ca.y * cb.y + ca.z * cb.z) / (norm3(ca) * norm3(cb));
}

}  // namespace pa
