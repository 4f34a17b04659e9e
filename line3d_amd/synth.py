"""Seeded synthetic Line3D scenes: helix of cameras looking at a box of
random 3-D segments; every view observes the projected segments with pixel noise.

No reference code involved: the reference has no data generator.  The camera convention is the
reference's (view.cc:24-34): x_cam = R X + t, C = -R^T t, P = K [R|t]; K is built the way the
drivers build it (main_vsfm.cpp:232-241): [[f,0,w/2],[0,f,h/2],[0,0,1]].
"""
from __future__ import annotations

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


class SplitMix64:
    """Counter-based SplitMix64: output i is mix(seed + (i+1)*gamma). Vectorised, portable."""

    def __init__(self, seed: int):
        self.seed = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        self.ctr = 0

    def u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(self.ctr + 1, self.ctr + n + 1, dtype=np.uint64)
            z = self.seed + idx * _GAMMA
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
        self.ctr += n
        return z

    def uniform(self, n: int) -> np.ndarray:
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def normal(self, n: int) -> np.ndarray:
        m = (n + 1) // 2
        u1 = 1.0 - self.uniform(m)          # (0,1]
        u2 = self.uniform(m)
        r = np.sqrt(-2.0 * np.log(u1))
        out = np.concatenate([r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)])
        return out[:n]

    def permutation(self, n: int) -> np.ndarray:
        return np.argsort(self.u64(n), kind="stable")


def _look_at(C: np.ndarray, target: np.ndarray) -> np.ndarray:
    z = target - C
    z /= np.linalg.norm(z)
    up = np.array([0.0, 1.0, 0.0])
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z])  # rows: world -> camera


class Scene:
    """views: list of dicts {id,K,R,t,width,height,segments(float32 Sx4),sims{id:sim},gt(int S)}."""

    def __init__(self, views, segs3d, params):
        self.views = views
        self.segs3d = segs3d
        self.params = params


def make_scene(n_views: int, n_segments: int, n_neighbors: int, seed: int = 1234,
               noise_px: float = 0.5, width: int = 1920, height: int = 1080, f: float = 1500.0,
               first_id: int = 0, step: float = 0.12) -> Scene:
    rng = SplitMix64(seed)
    K = np.array([[f, 0.0, width / 2.0], [0.0, f, height / 2.0], [0.0, 0.0, 1.0]])

    # cameras on a helix around the origin
    cams = []
    jit = rng.normal(6 * n_views).reshape(n_views, 6)
    for i in range(n_views):
        th = step * i
        turn = int(th // (2.0 * np.pi))
        r = 4.0 + 0.35 * turn
        h = 0.3 * np.sin(0.7 * i) + 0.25 * turn
        C = np.array([r * np.cos(th), h, r * np.sin(th)]) + 0.02 * jit[i, :3]
        R = _look_at(C, 0.05 * jit[i, 3:])
        t = -R @ C
        cams.append((R, t))

    # pool of 3-D segments; keep those visible (both endpoints) in every view so that each view
    # has exactly n_segments observations of the same 3-D lines
    pool = int(n_segments * 1.5) + 64
    u = rng.uniform(pool * 3).reshape(pool, 3)
    start = np.stack([2.0 * u[:, 0] - 1.0, 1.2 * u[:, 1] - 0.6, 2.0 * u[:, 2] - 1.0], axis=1)
    d = rng.normal(pool * 3).reshape(pool, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    length = 0.1 + 0.3 * np.abs(2.0 * rng.uniform(pool) - 1.0)
    end = start + d * length[:, None]

    def project(R, t, X):
        x = (K @ (R @ X.T + t[:, None])).T
        return x[:, :2] / x[:, 2:3], x[:, 2]

    visible = np.ones(pool, dtype=bool)
    for R, t in cams:
        for X in (start, end):
            p, z = project(R, t, X)
            visible &= (z > 0.1) & (p[:, 0] >= 1.0) & (p[:, 0] < width - 1.0) & (p[:, 1] >= 1.0) & (p[:, 1] < height - 1.0)
    keep = np.nonzero(visible)[0][:n_segments]
    if len(keep) < n_segments:
        raise RuntimeError("synthetic pool too small: %d < %d" % (len(keep), n_segments))
    start, end = start[keep], end[keep]

    views = []
    half = n_neighbors // 2
    for i, (R, t) in enumerate(cams):
        p1, _ = project(R, t, start)
        p2, _ = project(R, t, end)
        noise = noise_px * rng.normal(4 * n_segments).reshape(n_segments, 4)
        perm = rng.permutation(n_segments)
        segs = np.concatenate([p1, p2], axis=1)[perm] + noise
        sims = {}
        for j in range(max(0, i - half), min(n_views, i + half + 1)):
            if j != i:
                sims[first_id + j] = 1.0 / (1.0 + abs(i - j))
        views.append(dict(id=first_id + i, K=K.copy(), R=R.copy(), t=t.copy(), width=width, height=height,
                          segments=np.ascontiguousarray(segs, dtype=np.float32), sims=sims, gt=perm.copy()))
    params = dict(n_views=n_views, n_segments=n_segments, n_neighbors=n_neighbors, seed=seed,
                  noise_px=noise_px, width=width, height=height, f=f, step=step)
    return Scene(views, np.concatenate([start, end], axis=1), params)


def make_scene_from_poses(centers, targets, n_segments: int, seed: int = 77, noise_px: float = 0.5, width: int = 1920,
                          height: int = 1080, f: float = 1500.0, all_neighbors: bool = True) -> Scene:
    """Cameras at arbitrary poses (centre, look-at point) observing one pool of 3-D segments near the origin: opposing
    cameras, forward motion (epipole inside the image), ... -- geometries the helix of make_scene never produces.  Every
    view is every other view's neighbour."""
    rng = SplitMix64(seed)
    K = np.array([[f, 0.0, width / 2.0], [0.0, f, height / 2.0], [0.0, 0.0, 1.0]])
    cams = []
    for C, T in zip(centers, targets):
        C = np.asarray(C, float)
        R = _look_at(C, np.asarray(T, float))
        cams.append((R, -R @ C))
    pool = n_segments * 3 + 64
    u = rng.uniform(pool * 3).reshape(pool, 3)
    start = np.stack([1.6 * u[:, 0] - 0.8, 1.0 * u[:, 1] - 0.5, 1.6 * u[:, 2] - 0.8], axis=1)
    d = rng.normal(pool * 3).reshape(pool, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    end = start + d * (0.1 + 0.3 * np.abs(2.0 * rng.uniform(pool) - 1.0))[:, None]

    def project(R, t, X):
        x = (K @ (R @ X.T + t[:, None])).T
        return x[:, :2] / x[:, 2:3], x[:, 2]

    visible = np.ones(pool, dtype=bool)
    for R, t in cams:
        for X in (start, end):
            p, z = project(R, t, X)
            visible &= (z > 0.1) & (p[:, 0] >= 1.0) & (p[:, 0] < width - 1.0) & (p[:, 1] >= 1.0) & (p[:, 1] < height - 1.0)
    keep = np.nonzero(visible)[0][:n_segments]
    if len(keep) < n_segments:
        raise RuntimeError("synthetic pool too small: %d < %d" % (len(keep), n_segments))
    start, end = start[keep], end[keep]
    views = []
    for i, (R, t) in enumerate(cams):
        p1, _ = project(R, t, start)
        p2, _ = project(R, t, end)
        noise = noise_px * rng.normal(4 * n_segments).reshape(n_segments, 4)
        perm = rng.permutation(n_segments)
        segs = np.concatenate([p1, p2], axis=1)[perm] + noise
        sims = {j: 1.0 / (1.0 + abs(i - j)) for j in range(len(cams)) if j != i} if all_neighbors else {}
        views.append(dict(id=i, K=K.copy(), R=R.copy(), t=t.copy(), width=width, height=height,
                          segments=np.ascontiguousarray(segs, dtype=np.float32), sims=sims, gt=perm.copy()))
    params = dict(n_views=len(cams), n_segments=n_segments, n_neighbors=len(cams) - 1, seed=seed, noise_px=noise_px,
                  width=width, height=height, f=f)
    return Scene(views, np.concatenate([start, end], axis=1), params)


def pair_work(scene: Scene) -> int:
    """Stage-1 segment pairs as the reference schedules them: each mutual
    view pair is evaluated once, from the view processed first (ascending id)."""
    ids = [v["id"] for v in scene.views]
    S = {v["id"]: len(v["segments"]) for v in scene.views}
    nb = {v["id"]: set(v["sims"].keys()) for v in scene.views}
    done = set()
    total = 0
    for a in sorted(ids):
        for b in sorted(nb[a]):
            if (b, a) in done:      # already matched from b's side and mutual
                continue
            total += S[a] * S[b]
            if a in nb[b]:
                done.add((a, b))
    return total
