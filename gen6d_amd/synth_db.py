"""Procedural database with the reference's `BaseDatabase` protocol (dataset/database.py:30-54) for tests and demos:
a textured unit-diameter sphere-ish object rendered analytically from cameras on a Fibonacci sphere.  No dataset is
available offline, so `Gen6DEstimator.build/predict` is exercised on this."""
import numpy as np

from . import synth


class SyntheticDatabase:
    def __init__(self, n_views=96, size=(240, 320), radius=4.0, focal=330.0, seed=7, name="synthetic/blob"):
        self.database_name = name
        self.h, self.w = size
        self.object_center = np.array([0.05, -0.02, 0.03], np.float32)
        self.object_diameter = 1.0
        self.object_vert = np.array([0.0, 0.0, 1.0], np.float32)
        poses, _ = synth.fibonacci_cameras(n_views, radius=radius, focal=focal, size=min(size))
        c = self.object_center.astype(np.float64)
        for p in poses:                                  # look at the object centre instead of the origin
            p[:, 3] -= p[:, :3] @ c
        self.poses = poses
        self.K = np.array([[focal, 0, self.w / 2], [0, focal, self.h / 2], [0, 0, 1]], np.float32)
        rng = np.random.RandomState(seed)
        self._dirs = rng.randn(24, 3); self._dirs /= np.linalg.norm(self._dirs, axis=1, keepdims=True)
        self._cols = rng.rand(24, 3)
        self._bg = rng.rand(3)
        self._cache = {}

    def get_img_ids(self): return [str(i) for i in range(len(self.poses))]
    def get_K(self, i): return self.K.copy()
    def get_pose(self, i): return self.poses[int(i)].copy()

    def get_image(self, i):
        """Ray-cast a sphere of diameter 1 whose albedo is a smooth function of the surface direction."""
        i = int(i)
        if i not in self._cache:
            R, t = self.poses[i][:, :3].astype(np.float64), self.poses[i][:, 3].astype(np.float64)
            ys, xs = np.mgrid[0:self.h, 0:self.w]
            rays = np.stack([(xs - self.K[0, 2]) / self.K[0, 0], (ys - self.K[1, 2]) / self.K[1, 1], np.ones_like(xs, float)], -1)
            rays /= np.linalg.norm(rays, axis=-1, keepdims=True)
            cc = R @ self.object_center + t                            # sphere centre in camera coordinates
            b = rays @ cc
            disc = b * b - (cc @ cc - 0.25)
            hit = disc > 0
            tt = b - np.sqrt(np.where(hit, disc, 0))
            nrm = ((rays * tt[..., None] - cc) @ R) * 2.0              # object-frame unit normal
            wgt = np.exp(6.0 * (nrm @ self._dirs.T))                   # soft Voronoi colouring
            col = (wgt @ self._cols) / wgt.sum(-1, keepdims=True)
            shade = 0.55 + 0.45 * np.clip(-(nrm @ R.T) @ np.array([0, 0, 1.0]), 0, 1)
            img = np.where(hit[..., None], col * shade[..., None], self._bg[None, None] * (0.6 + 0.4 * ys[..., None] / self.h))
            self._cache[i] = (img * 255).round().clip(0, 255).astype(np.uint8)
        return self._cache[i]

    def get_mask(self, i):
        return np.ones((self.h, self.w), bool)

    def get_split(self, split_type):
        ids = self.get_img_ids()
        return ids[: len(ids) * 3 // 4], ids[len(ids) * 3 // 4:]
