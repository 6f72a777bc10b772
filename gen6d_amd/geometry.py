"""Host-side pose / 2-D similarity algebra between the stages of `Gen6DEstimator.predict`.

These are a few dozen flops per query on 3x4 matrices; they stay on the host in float64/float32 numpy exactly like the
reference.  Each function names the reference function whose contract it keeps (utils/base_utils.py,
utils/pose_utils.py, utils/database_utils.py, dataset/database.py).  The reference delegates Euler / quaternion
conversions to `transforms3d` (not vendored, unpinned in requirements.txt:4); the closed forms used here are the
standard ones for the conventions the reference calls: euler2mat(a,0,0,'syxz') = R_y(a), euler2mat(a,0,0,'sxyz') =
R_x(a), mat2euler(R,'szyx')[0] = angle about z of R = R_x R_y R_z(a), quat2mat(w,x,y,z).
"""
import numpy as np


# ------------------------------------------------------------------------------------------------ rigid poses
def pose_inverse(pose):
    """[R|t] -> [R^T | -R^T t] (base_utils.py:502-505)."""
    R = pose[:, :3].T
    return np.concatenate([R, -R @ pose[:, 3:]], -1)


def pose_compose(pose0, pose1):
    """apply pose0 first, then pose1 (base_utils.py:512-521)."""
    return np.concatenate([pose1[:, :3] @ pose0[:, :3], pose1[:, :3] @ pose0[:, 3:] + pose1[:, 3:]], 1)


def pose_apply(pose, pts):
    """pts [n,3] or [3] -> R pts + t (base_utils.py:523-524)."""
    return pts @ pose[:, :3].T + pose[:, 3]


def project_points(pts, pose, K):
    """[n,3] -> pixel coordinates [n,2] and depths [n] (base_utils.py:256-265; tiny |depth| clamped to 1e-4)."""
    p = (pts @ pose[:, :3].T + pose[:, 3:].T) @ K.T
    d = p[:, 2].copy()
    tiny = (np.abs(d) < 1e-4) & (np.abs(d) > 0)
    d[tiny] = 1e-4
    return p[:, :2] / d[:, None], d


def normalize_pose(pose, scale, offset):
    """pose for the object frame x' = scale*x + offset (dataset/database.py:399-404)."""
    R, t = pose[:3, :3], pose[:3, 3]
    return np.concatenate([R, (R @ -offset + scale * t)[:, None]], -1).astype(np.float32)


def denormalize_pose(pose, scale, offset):
    """inverse of normalize_pose (dataset/database.py:406-410)."""
    R, t = pose[:3, :3], pose[:3, 3]
    return np.concatenate([R, (R @ offset / scale + t / scale)[:, None]], -1).astype(np.float32)


# ------------------------------------------------------------------------------------------------ rotations
def rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float64)


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float64)


def look_at_rotation(point):
    """R with R @ (x, y, 1) parallel to the optical axis, for a point in normalised image coordinates
    (base_utils.py:657-666: euler2mat(atan y,'sxyz') @ euler2mat(-atan x,'syxz'))."""
    x, y = point
    return rot_x(np.arctan2(y, 1)) @ rot_y(-np.arctan2(x, 1))


def angle_about_z(R):
    """First angle of transforms3d.mat2euler(R, 'szyx'), i.e. a in R = R_x(c) R_y(b) R_z(a) (pose_utils.py:96-99)."""
    cy = np.sqrt(R[2, 2] ** 2 + R[1, 2] ** 2)
    if cy > np.finfo(float).eps * 4.0:
        return -np.arctan2(R[0, 1], R[0, 0])
    return np.arctan2(R[1, 0], R[1, 1])


def quat2mat(q):
    """(w,x,y,z) -> rotation matrix, normalising q (transforms3d.quaternions.quat2mat, used at pose_utils.py:239)."""
    w, x, y, z = np.asarray(q, np.float64)
    n = w * w + x * x + y * y + z * z
    if n < np.finfo(np.float64).eps:
        return np.eye(3)
    s = 2.0 / n
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ, xX, xY, xZ, yY, yZ, zZ = w * X, w * Y, w * Z, x * X, x * Y, x * Z, y * Y, y * Z, z * Z
    return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY],
                     [xY + wZ, 1.0 - (xX + zZ), yZ - wX],
                     [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])


# ------------------------------------------------------------------------------------------------ 2-D similarities [2,3]
def sim2d(scale=1.0, angle=0.0, offset=(0.0, 0.0)):
    c, s = np.cos(angle) * scale, np.sin(angle) * scale
    return np.array([[c, -s, offset[0]], [s, c, offset[1]]], np.float64)


def sim2d_compose(m0, m1):
    """apply m0 first, then m1 (base_utils.py:610-622)."""
    A = m1[:, :2] @ m0[:, :2]
    return np.concatenate([A, (m1[:, :2] @ m0[:, 2] + m1[:, 2])[:, None]], 1)


def sim2d_inverse(m):
    A = np.linalg.inv(m[:, :2])
    return np.concatenate([A, -A @ m[:, 2:]], 1)


def sim2d_apply(m, pts):
    return pts @ m[:, :2].T + m[:, 2:].T


def crop_transform(position, scale, angle, size, new_position=None):
    """M of `transformation_crop` (base_utils.py:646-653): centre on `position`, scale, rotate, move to the crop centre."""
    m = sim2d(offset=(-position[0], -position[1]))
    m = sim2d_compose(m, sim2d(scale=scale))
    m = sim2d_compose(m, sim2d(angle=angle))
    tgt = (size / 2, size / 2) if new_position is None else new_position
    return sim2d_compose(m, sim2d(offset=tgt))


# ------------------------------------------------------------------------------------------------ look-at rectification
def let_me_look_at_2d(image_center, K):
    """Rotation that brings pixel `image_center` onto the optical axis and the matching focal length
    (pose_utils.py:55-61)."""
    f_raw = (K[0, 0] + K[1, 1]) / 2
    c = np.asarray(image_center, np.float64) - K[:2, 2]
    return look_at_rotation(c / f_raw), np.sqrt(np.linalg.norm(c) ** 2 + f_raw ** 2)


def let_me_look_at(pose, K, obj_center):
    """(pose_utils.py:51-53)."""
    return let_me_look_at_2d(project_points(obj_center[None], pose, K)[0][0], K)


def look_at_crop_params(K, pose, position, angle, scale, h, w):
    """Everything `look_at_crop` computes except the image warp (database_utils.py:8-25):
    K_new, pose_new, pose_rect, H (source -> crop homography)."""
    R_new, f_new = let_me_look_at_2d(position, K)
    R_new = rot_z(angle) @ R_new
    f_new = f_new * scale
    K_new = np.array([[f_new, 0, w / 2], [0, f_new, h / 2], [0, 0, 1]], np.float32)
    H = K_new @ R_new @ np.linalg.inv(K)
    pose_rect = np.concatenate([R_new, np.zeros([3, 1])], 1).astype(np.float32)
    return K_new, pose_compose(pose, pose_rect), pose_rect, H


def scale_rotation_difference_from_cameras(ref_poses, que_poses, ref_Ks, que_Ks, center):
    """Relative scale and in-plane angle taking each reference view to the paired query view (pose_utils.py:63-102)."""
    def rect(poses, Ks):
        rots, fs = [], []
        for p, K in zip(poses, Ks):
            R, f = let_me_look_at(p, K, center)
            rots.append(R @ p[:, :3]); fs.append(f)
        cams = np.stack([pose_inverse(p)[:, 3] for p in poses])
        return np.stack(rots), np.asarray(fs), np.linalg.norm(cams - center[None], 2, 1)

    ref_rot, ref_f, ref_dist = rect(ref_poses, ref_Ks)
    que_rot, que_f, que_dist = rect(que_poses, que_Ks)
    rel = que_rot @ ref_rot.transpose(0, 2, 1)
    return ref_dist / que_dist * que_f / ref_f, np.asarray([angle_about_z(r) for r in rel])


# ------------------------------------------------------------------------------------------------ detection + selection -> pose
def estimate_pose_from_similarity_transform_compose(position, scale_r2q, angle_r2q, ref_pose, ref_K, que_K, object_center):
    """Pose of the query from the detected position/scale, the selected reference view and its in-plane angle
    (pose_utils.py:104-111 -> :12-49)."""
    ref_cen = project_points(object_center[None], ref_pose, ref_K)[0][0]
    m = sim2d(offset=(-position[0], -position[1]))
    m = sim2d_compose(m, sim2d(scale=1 / scale_r2q))
    m = sim2d_compose(m, sim2d(angle=-angle_r2q))
    m_q2r = sim2d_compose(m, sim2d(offset=ref_cen))
    m_r2q = sim2d_inverse(m_q2r)
    ref_cam = pose_inverse(ref_pose)[:, 3]
    que_cen = sim2d_apply(m_r2q, ref_cen[None])[0]
    que_cen_n = (np.linalg.inv(que_K) @ np.array([que_cen[0], que_cen[1], 1.0]))
    que_cen_n = que_cen_n[:2] / que_cen_n[2]
    scale = np.sqrt(np.linalg.det(m_r2q[:, :2]))
    rotation = np.arctan2(m_r2q[1, 0], m_r2q[0, 0])
    que_f, ref_f = (que_K[0, 0] + que_K[1, 1]) / 2, (ref_K[0, 0] + ref_K[1, 1]) / 2
    que_f_ = np.sqrt(que_f ** 2 + np.linalg.norm(que_cen_n * que_f) ** 2)
    que_dist = np.linalg.norm(ref_cam - object_center) * que_f_ / ref_f / scale
    ray = np.array([que_cen_n[0], que_cen_n[1], 1.0])
    que_cen3d = ray / np.linalg.norm(ray) * que_dist
    que_rot = look_at_rotation(que_cen_n).T @ (rot_z(rotation) @ ref_pose[:, :3])
    return np.concatenate([que_rot, (que_cen3d - que_rot @ object_center)[:, None]], 1)


# ------------------------------------------------------------------------------------------------ refiner residual -> pose
def compose_sim_pose(scale, quat, offset, in_pose, object_center):
    """Similarity transform (in the warped query camera frame) from the refiner outputs (pose_utils.py:237-244)."""
    rotation = quat2mat(quat)
    center_in = pose_apply(in_pose, object_center)
    center_que = center_in + np.concatenate([offset, np.zeros(1)])
    A = scale * rotation
    return np.concatenate([A, (center_que - A @ center_in)[:, None]], 1)


def pose_sim_to_pose_rigid(pose_sim_in_to_que, pose_in, K_que, K_in, center):
    """Nearest rigid pose to a similarity residual: SVD rotation, depth rescaled by the similarity scale
    (pose_utils.py:217-235)."""
    f_que, f_in = np.mean(np.diag(K_que)[:2]), np.mean(np.diag(K_in)[:2])
    center_in = pose_apply(pose_in, center)
    U, S, Vt = np.linalg.svd(pose_sim_in_to_que[:3, :3])
    depth_que = center_in[2] / np.mean(np.abs(S)) * f_que / f_in
    center_sim = pose_apply(pose_sim_in_to_que, center_in)
    center_que = center_sim / center_sim[2] * depth_que
    rotation = (U @ Vt) @ pose_in[:3, :3]
    return np.concatenate([rotation, (center_que - rotation @ center)[:, None]], 1)


# ------------------------------------------------------------------------------------------------ view selection
def sample_fps_points(points, sample_num, init_center=True):
    """Farthest-point sampling indices, started from the centroid (base_utils.py:558-586 with init_center=True,
    index_model=True): returns min(sample_num, n) - 1 indices."""
    n = points.shape[0]
    sample_num = min(n, sample_num)
    if not init_center:
        raise NotImplementedError("only the deterministic centroid start used at inference is implemented")
    cur = points.mean(0)
    dist = np.full(n, 1e8)
    out = []
    for _ in range(min(sample_num - 1, n - 1)):
        dist = np.minimum(dist, np.linalg.norm(cur[None] - points, 2, 1))
        k = int(np.argmax(dist))
        out.append(k)
        cur = points[k]
    return np.asarray(out, dtype=np.int64)


def view_correlation(que_poses, ref_poses, center):
    """cosine between camera directions seen from the object centre (database_utils.py:27-52, numpy branch)."""
    def dirs(poses):
        cams = np.stack([pose_inverse(p)[:, 3] for p in poses]) - center[None]
        return cams / np.linalg.norm(cams, 2, 1, keepdims=True)
    return dirs(que_poses) @ dirs(ref_poses).T
