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
               first_id: int = 0, step: float = 0.12, pp=(0.0, 0.0), seg_len=(0.1, 0.4), pool_factor: float = 1.5, turn_period: int = 0) -> Scene:
    """pp: offset of the principal point from the image centre in pixels; seg_len: range of the 3-D segment lengths (the box is 2 x 1.2 x 2 units at
    distance ~4: 0.1-0.4 projects to 40-150 pixels at f = 1500; 0.003-0.008 to 1-3 pixels; 1.5-2.5 spans the image); turn_period > 0: the helix's radius
    and height repeat every that many turns instead of growing without bound (2048 views are 39 turns: at radius 17 the whole box projects into the
    middle of the image and a view keeps 10-17 M matches instead of 4 M, profiles/r6_cfg5_2048_default_generator.txt) -- defaults reproduce every committed
    golden scene bit for bit."""
    rng = SplitMix64(seed)
    K = np.array([[f, 0.0, width / 2.0 + pp[0]], [0.0, f, height / 2.0 + pp[1]], [0.0, 0.0, 1.0]])

    # cameras on a helix around the origin
    cams = []
    jit = rng.normal(6 * n_views).reshape(n_views, 6)
    for i in range(n_views):
        th = step * i
        turn = int(th // (2.0 * np.pi))
        if turn_period > 0:
            turn = turn % turn_period
        r = 4.0 + 0.35 * turn
        h = 0.3 * np.sin(0.7 * i) + 0.25 * turn
        C = np.array([r * np.cos(th), h, r * np.sin(th)]) + 0.02 * jit[i, :3]
        R = _look_at(C, 0.05 * jit[i, 3:])
        t = -R @ C
        cams.append((R, t))

    # pool of 3-D segments; keep those visible (both endpoints) in every view so that each view
    # has exactly n_segments observations of the same 3-D lines
    pool = int(n_segments * pool_factor) + 64
    u = rng.uniform(pool * 3).reshape(pool, 3)
    start = np.stack([2.0 * u[:, 0] - 1.0, 1.2 * u[:, 1] - 0.6, 2.0 * u[:, 2] - 1.0], axis=1)
    d = rng.normal(pool * 3).reshape(pool, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    length = seg_len[0] + (seg_len[1] - seg_len[0]) * np.abs(2.0 * rng.uniform(pool) - 1.0)
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
                  noise_px=noise_px, width=width, height=height, f=f, step=step, pp=tuple(pp), seg_len=tuple(seg_len), turn_period=turn_period)
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


def make_scene_scattered(n_views: int, n_segments: int, seed: int = 4242, n_worldpoints: int = 900, twins: float = 0.12, noise_px: float = 0.5,
                         width: int = 1920, height: int = 1080, f: float = 1500.0) -> Scene:
    """Cameras SCATTERED around the box, in no order (view ids do not follow the geometry), each seeing only the part of the segment pool that
    projects into its image (ragged views: at most n_segments, often fewer); neighbours are not given -- every view carries the ids of the world
    points it sees (`worldpoints`), so Line3D::addImage builds the view similarities (line3D.cc:95-217, 1874-1935) and findVisualNeighbors picks the
    neighbourhoods (line3D.cc:476-549): top-N by shared points, hence NOT mutual; `twins` of the cameras stand a few centimetres beside another one
    and fall under min_baseline (line3D.cc:504-529: rejected against the view and against every neighbour already taken).  load_scene_worldpoints()."""
    rng = SplitMix64(seed)
    K = np.array([[f, 0.0, width / 2.0], [0.0, f, height / 2.0], [0.0, 0.0, 1.0]])
    u = rng.uniform(6 * n_views).reshape(n_views, 6)
    cams, centers = [], []
    for i in range(n_views):
        if i > 0 and u[i, 5] < twins:
            C = centers[int(u[i, 4] * i)] + 0.12 * (u[i, :3] - 0.5)                        # a twin: baseline <= 0.1 < min_baseline 0.25
        else:
            az, el, r = 2.0 * np.pi * u[i, 0], 0.9 * (u[i, 1] - 0.5), 3.9 + 1.6 * u[i, 2]
            C = np.array([r * np.cos(az) * np.cos(el), r * np.sin(el), r * np.sin(az) * np.cos(el)])
        centers.append(C)
        R = _look_at(C.copy(), 0.5 * (u[i, 3:] - 0.5) * np.array([1.0, 0.6, 1.0]))
        cams.append((R, -R @ C))
    pool = (5 * n_segments) // 4 + 64                 # (about three quarters of the pool project into a view: some views reach n_segments, some do not)
    q = rng.uniform(pool * 3).reshape(pool, 3)
    start = np.stack([5.2 * q[:, 0] - 2.6, 2.8 * q[:, 1] - 1.4, 5.2 * q[:, 2] - 2.6], axis=1)
    d = rng.normal(pool * 3).reshape(pool, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    end = start + d * (0.1 + 0.3 * np.abs(2.0 * rng.uniform(pool) - 1.0))[:, None]
    w = rng.uniform(n_worldpoints * 3).reshape(n_worldpoints, 3)
    wpts = np.stack([5.2 * w[:, 0] - 2.6, 2.8 * w[:, 1] - 1.4, 5.2 * w[:, 2] - 2.6], axis=1)

    def project(R, t, X):
        x = (K @ (R @ X.T + t[:, None])).T
        return x[:, :2] / x[:, 2:3], x[:, 2]

    def inside(p, z):
        return (z > 0.1) & (p[:, 0] >= 1.0) & (p[:, 0] < width - 1.0) & (p[:, 1] >= 1.0) & (p[:, 1] < height - 1.0)

    views = []
    for i, (R, t) in enumerate(cams):
        p1, z1 = project(R, t, start)
        p2, z2 = project(R, t, end)
        vis = np.nonzero(inside(p1, z1) & inside(p2, z2))[0]
        vis = vis[rng.permutation(len(vis))][:n_segments]
        noise = noise_px * rng.normal(4 * len(vis)).reshape(len(vis), 4)
        segs = np.concatenate([p1[vis], p2[vis]], axis=1) + noise
        pw, zw = project(R, t, wpts)
        seen = np.nonzero(inside(pw, zw) & (rng.uniform(n_worldpoints) < 0.75))[0]              # (a tracker misses a quarter of what is in view)
        views.append(dict(id=i, K=K.copy(), R=R.copy(), t=t.copy(), width=width, height=height, segments=np.ascontiguousarray(segs, dtype=np.float32),
                          sims={}, worldpoints=seen.astype(np.uint32), gt=vis.copy()))
    params = dict(n_views=n_views, n_segments=n_segments, seed=seed, n_worldpoints=n_worldpoints, twins=twins, noise_px=noise_px, width=width, height=height, f=f, kind="scattered")
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
