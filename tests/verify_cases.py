"""Seeded candidate lists for the K_verify_matches pin (tests/test_oracle_pins.py, tests/golden/make_golden_verify.py): one source view with S
segments, N neighbour cameras, per source segment a (camera, target)-sorted list of candidates as compute_pairwise_matches packs them
(cudawrapper.cu:958-1003) -- clusters of hypotheses at nearly the same depths (they support each other through the gate), outliers, several
candidates of one camera in a row (the per-camera maximum), candidates of the hypothesis's own camera (skipped), projections behind a camera."""
import numpy as np

F32 = np.float32


def make_case(seed, S=40, N=5, m_max=30, spatial_k=0.02, sigma_p=2.5, sigma_a=10.0):
    rng = np.random.default_rng(seed)
    Ks = np.array([[1500.0, 0, 960.0], [0, 1500.0, 540.0], [0, 0, 1.0]])

    def look_at(C):
        z = -C / np.linalg.norm(C)
        x = np.cross([0.0, 1.0, 0.0], z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        return np.stack([x, y, z])
    C_src = np.array([4.0, 0.2, 0.1])
    R_src = look_at(C_src)
    RtKinv = (R_src.T @ np.linalg.inv(Ks)).astype(F32)
    src = np.empty((S, 4), F32)
    src[:, 0:2] = (rng.random((S, 2)) * [1600, 900] + [150, 90]).astype(F32)
    src[:, 2:4] = src[:, 0:2] + rng.normal(0, 60, (S, 2)).astype(F32)
    P = np.empty((N, 3, 4), F32)
    cams = []
    for c in range(N):
        th = 0.25 * (c + 1) * (1 if c % 2 else -1)
        Cc = np.array([4.0 * np.cos(th), 0.3 * c - 0.5, 4.0 * np.sin(th)])
        Rc = look_at(Cc)
        cams.append((Rc, Cc))
        P[c] = (Ks @ np.concatenate([Rc, (-Rc @ Cc)[:, None]], 1)).astype(F32)

    def ray(p):
        r = RtKinv.astype(np.float64) @ np.array([p[0], p[1], 1.0])
        return r / np.linalg.norm(r)

    data, depths, offsets, tgt_by_cam = [], [], np.zeros((S, 2), np.int32), [[] for _ in range(N)]
    for s in range(S):
        m = int(rng.integers(0, m_max))
        base1, base2 = rng.uniform(2.5, 5.5), rng.uniform(2.5, 5.5)
        rows = []
        for _ in range(m):
            c = int(rng.integers(0, N))
            kind = rng.integers(0, 5)
            f1 = 1.0 + (rng.normal(0, 0.004) if kind <= 2 else rng.normal(0, 0.2))          # cluster vs outlier
            f2 = 1.0 + (rng.normal(0, 0.004) if kind <= 2 else rng.normal(0, 0.2))
            d1, d2 = base1 * f1, base2 * f2
            if kind == 4 and rng.random() < 0.2:
                d1 = -d1                                                                   # behind the source camera
            X1, X2 = C_src + d1 * ray(src[s, 0:2]), C_src + d2 * ray(src[s, 2:4])
            q = []
            for X in (X1, X2):
                x = P[c].astype(np.float64) @ np.array([X[0], X[1], X[2], 1.0])
                q += [x[0] / x[2] + rng.normal(0, 1.0), x[1] / x[2] + rng.normal(0, 1.0)] if abs(x[2]) > 1e-9 else [0.0, 0.0]
            rows.append((c, q, d1, d2, rng.uniform(2, 6), rng.uniform(2, 6)))
        rows.sort(key=lambda r: r[0])
        offsets[s] = (len(data), len(rows))
        for c, q, d1, d2, d3, d4 in rows:
            tgt_by_cam[c].append(q)
            data.append((s, c, len(tgt_by_cam[c]) - 1, 0.0))
            depths.append((d1, d2, d3, d4))
    cam_off = np.zeros((N, 2), np.int32)
    tgt = []
    for c in range(N):
        cam_off[c] = (len(tgt), len(tgt_by_cam[c]))
        tgt += tgt_by_cam[c]
    return dict(matches_data=np.array(data, F32).reshape(-1, 4), matches_depths=np.array(depths, F32).reshape(-1, 4), match_offsets=offsets,
                camera_offsets=cam_off, src_segs=src, RtKinv=RtKinv, C_src=C_src.astype(F32), tgt_segs=np.array(tgt, F32).reshape(-1, 4), P=P,
                sigma_p=F32(sigma_p), sigma_a=F32(sigma_a), spatial_k=F32(spatial_k))


CASES = [dict(seed=1), dict(seed=2, S=60, N=8, m_max=50), dict(seed=3, spatial_k=0.0), dict(seed=4, S=25, N=3, m_max=80, spatial_k=0.05),
         dict(seed=5, S=80, N=12, m_max=40, sigma_p=1.0, sigma_a=5.0)]
