"""Gen6DEstimator — drop-in for the reference's estimator.py:94-220 (`build(database, split_type)`,
`predict(que_img, que_K, pose_init=None) -> (pose [3,4], inter_results)`, `ref_info`, `cfg`, `name2estimator`).

The reference warps images on the host with OpenCV between the stages (transformation_crop, look_at_crop,
normalize_reference_views); here every warp is one `g6d_warp_perspective` launch on images that stay resident on the
GPU (reference images are uploaded once and cached per image id), and the 3x4 pose algebra stays in numpy
(`gen6d_amd/geometry.py`).  The database is any object with the reference's `BaseDatabase` protocol
(`get_image / get_K / get_pose / get_img_ids / get_mask`, dataset/database.py:30-54) plus the object meta the
reference looks up per database type (`object_center`, `object_diameter`, `object_vert`).
"""
import numpy as np
import torch

from . import geometry as G
from . import ops


# ------------------------------------------------------------------------------------------------ database helpers
def _reference_database_module(db):
    """The reference's own `dataset.database` module when this process runs under the reference repo (eval.py /
    predict.py import it) and `db` is one of its classes: its get_object_center / get_diameter / get_object_vert /
    get_database_split then resolve every reference database type exactly as the reference does
    (dataset/database.py:311-397).  Never imported from here — only picked up when already loaded."""
    import sys
    mod = sys.modules.get("dataset.database")
    base = getattr(mod, "BaseDatabase", None)
    return mod if (base is not None and isinstance(db, base)) else None


def _kind(db):
    return type(db).__name__


def get_object_center(db):
    """reference dataset/database.py:365-381, by attribute instead of isinstance so that any database with the
    BaseDatabase protocol works: LINEMOD / GSO / ShapeNet expose `object_center`, Custom `center`, GenMOP
    `meta_info.center`, NormalizedDatabase sits at the origin."""
    mod = _reference_database_module(db)
    if mod is not None:
        return np.asarray(mod.get_object_center(db), np.float32)
    if _kind(db) == "NormalizedDatabase" and not hasattr(db, "object_center"):
        return np.zeros(3, np.float32)
    for name in ("object_center", "center"):
        if hasattr(db, name):
            return np.asarray(getattr(db, name), np.float32)
    meta = getattr(db, "meta_info", None)
    if meta is not None and hasattr(meta, "center"):
        return np.asarray(meta.center, np.float32)
    raise AttributeError(f"{_kind(db)}: no object centre (object_center / center / meta_info.center)")


def get_diameter(db):
    """reference dataset/database.py:346-363: GenMOP / Custom / Normalized objects are already scaled to diameter 2,
    LINEMOD reads data/LINEMOD/<model>/distance.txt (cm), the rendered sets carry `object_diameter`."""
    mod = _reference_database_module(db)
    if mod is not None:
        return float(mod.get_diameter(db))
    for name in ("object_diameter", "diameter"):
        if hasattr(db, name):
            return float(getattr(db, name))
    kind = _kind(db)
    if kind in ("GenMOPDatabase", "CustomDatabase", "NormalizedDatabase") or hasattr(db, "meta_info"):
        return 2.0
    if kind == "LINEMODDatabase" or str(getattr(db, "database_name", "")).startswith("linemod/"):
        model = db.database_name.split("/")[-1]
        return float(np.loadtxt(f"data/LINEMOD/{model}/distance.txt")) / 100
    raise AttributeError(f"{kind}: no object diameter (object_diameter / diameter)")


def get_object_vert(db):
    """reference dataset/database.py:383-397: +z unless the database says otherwise."""
    mod = _reference_database_module(db)
    if mod is not None:
        try:
            return np.asarray(mod.get_object_vert(db), np.float32)
        except NotImplementedError:          # the reference has no entry for NormalizedDatabase: use the wrapped one
            pass
    if hasattr(db, "object_vert"):
        return np.asarray(db.object_vert, np.float32)
    inner = getattr(db, "database", None)
    if inner is not None:
        return get_object_vert(inner)
    return np.asarray((0.0, 0.0, 1.0), np.float32)


def get_database_split(db, split_type):
    """(ref_ids, que_ids) as reference dataset/database.py:311-325: 'all' -> every image on both sides,
    'linemod_test' / 'linemod_val' -> train.txt as references, test.txt (every 10th for _val) as queries.
    A database may override with its own `get_split(split_type)`."""
    mod = _reference_database_module(db)
    if mod is not None:
        return mod.get_database_split(db, split_type)
    if hasattr(db, "get_split"):
        return db.get_split(split_type)
    if split_type == "all":
        ids = list(db.get_img_ids())
        return ids, ids
    if split_type.startswith("linemod"):
        model = db.database_name.split("/")[1]
        ids = lambda fn: [str(int(l.split("/")[-1].split(".")[0])) for l in np.loadtxt(fn, dtype=str).tolist()]
        que = ids(f"data/LINEMOD/{model}/test.txt")
        return ids(f"data/LINEMOD/{model}/train.txt"), (que[::10] if split_type == "linemod_val" else que)
    raise NotImplementedError(split_type)


class NormalizedDatabase:
    """Object rescaled into the unit sphere at the origin (reference dataset/database.py:667-694)."""

    def __init__(self, database):
        self.database = database
        self.database_name = "norm/" + getattr(database, "database_name", "db")
        center, diameter = get_object_center(database), get_diameter(database)
        self.scale = 2 / diameter
        self.offset = -self.scale * center
        self.object_center = np.zeros(3, np.float32)
        self.object_diameter = 2.0
        self.object_vert = get_object_vert(database)

    def get_image(self, i): return self.database.get_image(i)
    def get_K(self, i): return self.database.get_K(i)
    def get_pose(self, i): return G.normalize_pose(self.database.get_pose(i), self.scale, self.offset)
    def get_img_ids(self): return self.database.get_img_ids()
    def get_mask(self, i): return self.database.get_mask(i)


def _root(db):
    while hasattr(db, "database") and _kind(db) == "NormalizedDatabase":
        db = db.database
    return db


class DeviceImageCache:
    """Per-estimator cache of what the reference re-reads from disk on every call: the uint8 HWC images (and masks)
    of a database, uploaded once and kept on the GPU, and the pose-independent FPS subset of
    select_reference_img_ids_refinement.  Entries are keyed by the database OBJECT (kept alive by the cache, so a
    CPython id can never be reused by another database) and, for the subset, by the exact id list; `clear()` is
    called by Gen6DEstimator.build."""

    def __init__(self, device):
        self.device = device
        self.clear()

    def clear(self):
        self._dbs, self._imgs, self._masks, self._subsets = {}, {}, {}, {}

    def _key(self, db):
        root = _root(db)
        self._dbs[id(root)] = root                     # strong reference: the id stays unique while cached
        return id(root)

    def holds(self, db):
        return id(_root(db)) in self._dbs

    def get(self, db, img_id):
        key = (self._key(db), img_id)
        if key not in self._imgs:
            self._imgs[key] = torch.from_numpy(np.ascontiguousarray(db.get_image(img_id))).to(self.device)
        return self._imgs[key]

    def get_mask(self, db, img_id):
        """uint8 [h,w,1] (0 / 255) on the device."""
        key = (self._key(db), img_id)
        if key not in self._masks:
            m = (np.asarray(db.get_mask(img_id)) > 0).astype(np.uint8) * 255
            self._masks[key] = torch.from_numpy(np.ascontiguousarray(m[..., None])).to(self.device)
        return self._masks[key]

    def subset(self, db, ref_ids, even, even_num, compute):
        key = (self._key(db), _kind(db), tuple(str(i) for i in ref_ids), bool(even), int(even_num))
        if key not in self._subsets:
            self._subsets[key] = compute()
        return self._subsets[key]


def select_reference_img_ids_fps(db, ref_ids_all, ref_num):
    """reference utils/database_utils.py:112-123 (deterministic branch)."""
    center = get_object_center(db)
    cams = np.asarray([G.pose_inverse(db.get_pose(i))[:, 3] - center for i in ref_ids_all])
    return np.asarray(ref_ids_all)[G.sample_fps_points(cams, ref_num + 1, True)]


def select_reference_img_ids_refinement(db, center, ref_ids, sel_pose, ref_num=6, even=False, even_num=128, cache=None):
    """reference utils/database_utils.py:125-139.  The reference recomputes the pose list and its FPS subset on every
    call; with a DeviceImageCache the (pose-independent) subset is computed once per (database, id list)."""
    ref_ids = np.asarray(ref_ids)

    def compute():
        ids = ref_ids
        poses = np.asarray([db.get_pose(i) for i in ids])
        if even:
            cams = np.asarray([G.pose_inverse(p)[:, 3] for p in poses])
            idx = G.sample_fps_points(cams, even_num + 1, True)
            ids, poses = ids[idx], poses[idx]
        return ids, poses

    ids, poses = cache.subset(db, ref_ids, even, even_num, compute) if cache is not None else compute()
    corr = G.view_correlation(sel_pose[None], poses, center)
    return ids[np.argsort(-corr[0])[:ref_num]]


def reference_view_params(db, ref_ids, size, margin, rectify_rot=True, input_pose=None, input_K=None, angle_step=0.0):
    """The geometry of normalize_reference_views (utils/database_utils.py:54-110) without the image warps: per reference view the
    new intrinsics, pose and source->crop homography.  angle_step > 0 (radians; reference-feature caching): the
    in-plane alignment angle is snapped to multiples of angle_step, so that the crop of a view depends on (view, bucket) only; the
    fourth result is then the bucket per view (None otherwise)."""
    center, diameter = get_object_center(db), get_diameter(db)
    poses = np.asarray([db.get_pose(i) for i in ref_ids])
    Ks = np.asarray([db.get_K(i) for i in ref_ids])
    cens = np.asarray([G.project_points(center[None], p, K)[0][0] for p, K in zip(poses, Ks)])
    cams = np.stack([G.pose_inverse(p)[:, 3] for p in poses], 0)
    dist = np.linalg.norm(cams - center[None], 2, 1)
    f_look = np.asarray([G.let_me_look_at(p, K, center)[1] for p, K in zip(poses, Ks)])
    scales = size * (1 - margin) / diameter * dist / f_look
    if not rectify_rot:
        angles = np.zeros(len(ref_ids), np.float32)
    elif input_K is not None and input_pose is not None:
        n = len(poses)
        _, angles = G.scale_rotation_difference_from_cameras(poses, np.repeat(input_pose[None], n, 0), Ks,
                                                             np.repeat(input_K[None], n, 0), center)
    else:
        v2 = np.asarray([(p[:, :3] @ get_object_vert(db))[:2] for p in poses])
        small = np.linalg.norm(v2, 2, 1) < 1e-5
        v2[small] += 1e-5 * np.sign(v2[small])
        angles = -np.arctan2(v2[:, 1], v2[:, 0]) - np.pi / 2
    buckets = None
    if angle_step > 0:
        buckets = np.floor(np.asarray(angles, np.float64) / angle_step + 0.5).astype(np.int64)
        angles = buckets * angle_step
    Ks_new, poses_new, Hs = [], [], []
    for k in range(len(ref_ids)):
        K_new, pose_new, _, H = G.look_at_crop_params(Ks[k], poses[k], cens[k], angles[k], scales[k], size, size)
        Ks_new.append(K_new); poses_new.append(pose_new); Hs.append(H)
    return np.stack(Ks_new, 0).astype(np.float32), np.stack(poses_new, 0).astype(np.float32), np.stack(Hs, 0), buckets


def normalize_reference_views(db, ref_ids, size, margin, cache, rectify_rot=True, input_pose=None, input_K=None,
                              with_masks=True):
    """Crop every reference view so that the object is centred, fills `size*(1-margin)` pixels and is upright (or aligned
    with `input_pose`).  Returns, in the reference's order (utils/database_utils.py:54-110): device images uint8
    [rfn,size,size,3], masks float32 [rfn,size,size] in [0,1] (None when with_masks=False — the per-query refiner path
    does not use them), Ks, poses, Hs."""
    Ks_new, poses_new, Hs, _ = reference_view_params(db, ref_ids, size, margin, rectify_rot, input_pose, input_K)
    imgs, masks = [], []
    for k, i in enumerate(ref_ids):
        imgs.append(ops.warp_perspective(cache.get(db, i), Hs[k], size, size))
        if with_masks:
            masks.append(ops.warp_perspective(cache.get_mask(db, i), Hs[k], size, size, out_float=True)[..., 0])
    return torch.stack(imgs, 0), torch.stack(masks, 0) if with_masks else None, Ks_new, poses_new, Hs


class Gen6DEstimator:
    default_cfg = {
        "ref_resolution": 128, "ref_view_num": 64, "det_ref_view_num": 32,
        "selector": None, "detector": None, "refiner": None, "refine_iter": 3,
    }

    def __init__(self, cfg, modules=None):
        """cfg as in configs/gen6d_pretrain.yaml.  `modules` = dict(detector=…, selector=…, refiner=…) of already
        constructed networks bypasses checkpoint loading (used with synthetic weights)."""
        self.cfg = {**self.default_cfg, **cfg}
        self.ref_info = {}
        if modules is not None:
            self.detector, self.selector, self.refiner = modules["detector"], modules["selector"], modules.get("refiner")
        else:
            self.detector = self._load_module(self.cfg["detector"])
            self.selector = self._load_module(self.cfg["selector"])
            self.refiner = self._load_module(self.cfg["refiner"]) if self.cfg["refiner"] is not None else None
        self.device = self.detector.device_()
        self.cache = DeviceImageCache(self.device)

    @staticmethod
    def _load_module(cfg_path):
        """YAML -> network -> data/model/<name>/model_best.pth (reference estimator.py:117-125)."""
        import yaml
        from .network import name2network
        with open(cfg_path) as f:
            cfg = yaml.load(f, Loader=yaml.FullLoader)
        net = name2network[cfg["network"]](cfg)
        state = torch.load(f'data/model/{cfg["name"]}/model_best.pth', map_location="cpu")
        net.load_state_dict(state["network_state_dict"])
        return net.cuda().eval()

    def build(self, database, split_type):
        """Select, normalise and rotate the reference views and load them into the networks
        (reference estimator.py:139-171)."""
        self.cache.clear()                                     # a new object: nothing of the previous one may survive
        center, vert = get_object_center(database), get_object_vert(database)
        ref_ids_all, _ = get_database_split(database, split_type)
        ref_ids = select_reference_img_ids_fps(database, ref_ids_all, self.cfg["ref_view_num"])
        size = self.cfg["ref_resolution"]
        ref_imgs, ref_masks, ref_Ks, ref_poses, ref_Hs = normalize_reference_views(database, ref_ids, size, 0.05, self.cache)
        angles = [-np.pi / 2, -np.pi / 4, 0, np.pi / 4, np.pi / 2]
        an = self.selector.cfg["selector_angle_num"]
        if an != 5:
            angles = list(np.linspace(-np.pi, np.pi, an, endpoint=False))
        rots = []
        for a in angles:                                        # in-plane rotated copies about the crop centre
            M = G.sim2d_compose(G.sim2d_compose(G.sim2d(offset=(-size / 2, -size / 2)), G.sim2d(angle=a)),
                                G.sim2d(offset=(size / 2, size / 2)))
            Hr = np.concatenate([M, [[0, 0, 1]]], 0)
            rots.append(torch.stack([ops.warp_perspective(self.cache.get(database, i), Hr @ ref_Hs[k], size, size)
                                     for k, i in enumerate(ref_ids)], 0))
        ref_imgs_rots = torch.stack(rots, 0)                    # an,rfn,h,w,3 uint8 on the device
        with torch.no_grad():
            det = ref_imgs[:self.cfg["det_ref_view_num"]].float().div_(255).permute(0, 3, 1, 2).contiguous()
            self.detector.load_impl(det)
            f = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(self.device)
            self.selector.extract_ref_feats(ref_imgs_rots.float().div_(255).permute(0, 1, 4, 2, 3).contiguous(),
                                            f(ref_poses), f(center), f(vert))
        self.ref_info = {"imgs": ref_imgs.cpu().numpy(), "ref_imgs": ref_imgs_rots.cpu().numpy(), "masks": ref_masks.cpu().numpy(),
                         "Ks": ref_Ks, "poses": ref_poses, "center": center}
        if self.refiner is not None:
            self.refiner.image_cache = self.cache
            self.refiner.load_ref_imgs(database, ref_ids_all)

    def predict(self, que_img, que_K, pose_init=None):
        """que_img uint8 [H,W,3], que_K [3,3] -> pose [3,4], intermediate results (reference estimator.py:173-216)."""
        inter = {}
        que_dev = (que_img if torch.is_tensor(que_img) else torch.from_numpy(np.ascontiguousarray(que_img))).to(self.device)
        if pose_init is None:
            with torch.no_grad():
                x = que_dev.float().div_(255).permute(2, 0, 1)[None].contiguous()
                det = self.detector.detect_impl(x)
                position = det["positions"][0].cpu().numpy()
                scale_r2q = float(det["scales"][0])
                size = self.cfg["ref_resolution"]
                M = G.crop_transform(position, 1 / scale_r2q, 0, size)
                crop = ops.warp_perspective(que_dev, M, size, size)
                logits, angles = self.selector.compute_view_point_feats(crop.float().div_(255).permute(2, 0, 1)[None].contiguous())
                ref_idx = int(torch.argmax(logits, 1)[0])
                angle_r2q = float(angles[0, ref_idx])
            inter.update(det_position=position, det_scale_r2q=scale_r2q, det_que_img=crop.cpu().numpy(),
                         sel_angle_r2q=angle_r2q, sel_scores=logits[0].cpu().numpy(), sel_ref_idx=ref_idx)
            pose_pr = G.estimate_pose_from_similarity_transform_compose(
                position, scale_r2q, angle_r2q, self.ref_info["poses"][ref_idx], self.ref_info["Ks"][ref_idx], que_K,
                self.ref_info["center"])
        else:
            pose_pr = pose_init
        if self.refiner is not None:
            poses = [pose_pr]
            for _ in range(self.cfg["refine_iter"]):
                pose_pr = self.refiner.refine_que_imgs(que_dev, que_K, pose_pr, size=128, ref_num=6, ref_even=True)
                poses.append(pose_pr)
            inter["refine_poses"] = poses
        return np.asarray(pose_pr, np.float32), inter


    # ------------------------------------------------------------------ device-resident variants (gen6d_amd/chain.py)
    def device_chain(self):
        """The whole of `predict` as one chain of launches on reference state resident in HBM (built lazily after `build`)."""
        from .chain import DeviceChain
        if getattr(self, "_chain", None) is None or self._chain_info is not self.ref_info:
            self._chain, self._chain_info = DeviceChain(self), self.ref_info
        return self._chain

    def predict_device(self, que_img, que_K, use_feat_cache=False):
        """`predict` without host round trips between the stages: que_img uint8 [H,W,3] (numpy or device tensor), que_K [3,3]
        -> (pose [3,4] float32 numpy, inter) with one synchronisation at the end.  use_feat_cache: reference-crop features are reused
        across refinement steps and queries (needs the refiner's cfg ref_feat_cache_deg > 0; one tiny read-back per step)."""
        chain = self.device_chain()
        img = que_img if torch.is_tensor(que_img) else torch.from_numpy(np.ascontiguousarray(que_img))
        K = torch.from_numpy(np.ascontiguousarray(que_K, dtype=np.float32)).to(self.device)
        out = chain.query(img.to(self.device), K, use_feat_cache=use_feat_cache)
        det, sel = out["det"].cpu().numpy(), out["sel"].cpu().numpy()
        inter = {"det_position": det[:2], "det_scale_r2q": float(det[2]), "det_que_img": None, "sel_angle_r2q": float(sel[1]),
                 "sel_scores": out["logits"].cpu().numpy(), "sel_ref_idx": int(sel[0]),
                 "refine_poses": [p.cpu().numpy() for p in out["refine_poses"]]}
        return out["pose"].cpu().numpy().astype(np.float32), inter

    def predict_many(self, que_imgs, que_Ks, lanes=3, batch=1):
        """Several queries in flight at once (one captured hipGraph of the whole chain per lane, `batch` queries per graph sharing
        every launch), one synchronisation at the end: [(pose, inter)] in query order (BASELINE configs[4]: batched multi-query
        stream)."""
        return self.device_chain().predict_many(que_imgs, que_Ks, lanes, batch)


name2estimator = {"gen6d": Gen6DEstimator}
