"""Inputs for the device-function pins (tests/test_oracle_pins.py, tests/golden/make_golden_devfn.py): the same seeded random
and adversarial cases are fired through the reference's own functions (oracle/_ref/libdevfn_ref.so, prefix l3dref_) and the
oracle's restatements (oracle/libl3d_oracle.so, prefix l3do_devfn_); the three kernel bodies (SPLICED) through
oracle/_spliced/libkernels_spliced.so.  Points are float32 xyz triples."""
import ctypes as C

import numpy as np

F32 = np.float32


def _pts2d(rng, n, scale=2000.0):
    p = np.empty((n, 3), F32)
    p[:, :2] = (rng.random((n, 2)) * scale - scale * 0.1).astype(F32)
    p[:, 2] = 1.0
    return p


def overlap_cases(rng, n):
    """Four points per case (src_p1, src_p2, q1, q2) as D_segment_overlap_2D meets them -- the q's are intersections of epipolar
    lines with the source line, i.e. (nearly) collinear with it -- plus the hard spots: exactly collinear lattice points, points
    ON endpoints (dot = 0 against EPS_G), zero-length and sub-pixel segments (the `< 1.0f` tests), reversed and nested
    intervals, huge and tiny coordinates."""
    out = [np.empty((n, 3), F32) for _ in range(4)]
    kinds = rng.integers(0, 8, n)
    o = (rng.random((n, 2)) * 1500).astype(F32)
    ang = rng.random(n) * 2 * np.pi
    u = np.stack([np.cos(ang), np.sin(ang)], 1)
    lattice_dirs = np.array([[1, 0], [0, 1], [1, 1], [1, -1], [3, 4], [-5, 12], [2, 1], [7, -24]], np.float64)
    specials = np.array([0.0, 1.0, 0.5, 0.25, 2.0, -1.0, 1.0 - 2.0 ** -24, 1.0 + 2.0 ** -23, 0.999999, 1.000001, 3.0, 100.0])
    for i in range(n):
        k = kinds[i]
        if k <= 1:        # generic: random parameters along a random line, small perpendicular noise (float rounding scale)
            t = rng.normal(0, 80, 4)
            noise = rng.normal(0, 1e-3 if k == 0 else 0.0, (4, 2))
            P = o[i] + t[:, None] * u[i] + noise
        elif k == 2:      # integer lattice, exactly collinear: dots and lengths are exact in float
            d = lattice_dirs[rng.integers(0, len(lattice_dirs))]
            t = rng.integers(-40, 41, 4).astype(np.float64)
            P = np.round(o[i]) + t[:, None] * d
        elif k == 3:      # shared endpoints / duplicates
            d = lattice_dirs[rng.integers(0, len(lattice_dirs))]
            t = rng.integers(-10, 11, 2).astype(np.float64)
            t = np.array([t[0], t[1], t[rng.integers(0, 2)], rng.integers(-10, 11)], np.float64)
            if rng.random() < 0.3:
                t[3] = t[rng.integers(0, 3)]
            P = np.round(o[i]) + t[:, None] * d
        elif k == 4:      # lengths around the 1-pixel threshold, axis aligned (exact) or not
            L = specials[rng.integers(0, len(specials), 2)]
            a = rng.integers(-5, 6, 2).astype(np.float64)
            t = np.array([a[0], a[0] + L[0], a[1], a[1] + L[1]])
            d = np.array([1.0, 0.0]) if rng.random() < 0.5 else u[i]
            P = np.round(o[i]) + t[:, None] * d
        elif k == 5:      # nested / containing / disjoint intervals with a random scale
            s = 10.0 ** rng.uniform(-2, 4)
            t = np.sort(rng.random(4)) * s
            perm = [[0, 3, 1, 2], [1, 2, 0, 3], [0, 1, 2, 3], [0, 2, 1, 3], [3, 0, 2, 1], [2, 1, 3, 0]][rng.integers(0, 6)]
            P = o[i] + t[perm][:, None] * u[i]
        elif k == 6:      # not collinear at all (the function is still deterministic there)
            P = rng.random((4, 2)) * 1000
        else:             # degenerate / extreme magnitudes
            s = 10.0 ** rng.uniform(-20, 15)
            P = (rng.random((4, 2)) - 0.5) * s
            if rng.random() < 0.3:
                P[:] = P[0]
        for j in range(4):
            out[j][i, :2] = P[j].astype(F32)
            out[j][i, 2] = 1.0
    return out


def point_on_segment_cases(rng, n):
    sp1, sp2, q1, _ = overlap_cases(rng, n)
    return sp1, sp2, q1


def lines_and_points(rng, n):
    line = rng.normal(0, 1, (n, 3)).astype(F32)
    line[:, 2] *= 500
    k = rng.integers(0, 10, n)
    line[k == 0, 0] = 0.0                      # horizontal / vertical / degenerate lines
    line[k == 1, 1] = 0.0
    line[k == 2, :2] = 0.0                     # 0/0 and x/0: NaN / inf must come out alike
    line[k == 3] *= F32(1e-20)
    p = _pts2d(rng, n)
    return line, p


def hom_points(rng, n):
    p = rng.normal(0, 100, (n, 3)).astype(F32)
    k = rng.integers(0, 8, n)
    p[k == 0, 2] = 0.0
    p[k == 1, 2] = F32(1e-12)                  # |z| > EPS_G is false at exactly EPS_G (as float)
    p[k == 2, 2] = np.nextafter(F32(1e-12), F32(1))
    p[k == 3, 2] = F32(-1e-12)
    p[k == 4, 2] = F32(1e-30)
    return p


def points3d(rng, n, scale=3.0):
    return rng.normal(0, scale, (n, 3)).astype(F32)


def angle_cases(rng, n):
    P1, P2, Q1, Q2 = (points3d(rng, n) for _ in range(4))
    k = rng.integers(0, 8, n)
    par = k == 0                               # parallel / antiparallel / nearly parallel: dot at and around +-1 (the clamp)
    s = rng.choice([1.0, -1.0, 2.5, -0.3], n).astype(F32)
    Q2[par] = Q1[par] + (P2[par] - P1[par]) * s[par, None]
    near = k == 1
    Q2[near] = Q1[near] + (P2[near] - P1[near]) + rng.normal(0, 1e-4, (int(near.sum()), 3)).astype(F32)
    perp = k == 2                              # right angles: the 90-degree fold
    d = P2[perp] - P1[perp]
    r = rng.normal(0, 1, d.shape).astype(F32)
    Q2[perp] = Q1[perp] + np.cross(d, r).astype(F32)
    zero = k == 3                              # zero-length direction: NaN from normalize
    Q2[zero] = Q1[zero]
    return P1, P2, Q1, Q2


def collinearity_cases(rng, n):
    """Two segments per case (p1, p2, q1, q2) and the squared sigma, as K_collinearity meets them: near-collinear pairs at perpendicular
    offsets around sigma (the affinity threshold 0.5 is at d = 1.18 sigma), along-the-line arrangements on both sides of the overlap
    check (gaps, touching end points -- the dots against -EPS_G --, nested and crossing intervals), exact lattice cases, and unrelated or
    degenerate segments."""
    p1, p2, q1, q2 = overlap_cases(rng, n)                      # (nearly) collinear quadruples incl. lattice / shared end points / extremes
    sig = rng.choice(np.array([1.0, 2.5, 5.0, 10.0, 0.3], F32), n)
    k = rng.integers(0, 6, n)
    nrm = np.stack([-(p2[:, 1] - p1[:, 1]), p2[:, 0] - p1[:, 0]], 1).astype(np.float64)
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = np.divide(nrm, ln, out=np.zeros_like(nrm), where=ln > 0)
    off = (rng.random(n) * 3.0 * sig)[:, None] * nrm            # second segment shifted sideways by 0 .. 3 sigma
    tilt = (rng.normal(0, 0.5, n) * sig)[:, None] * nrm
    m = k <= 2
    q1[m, :2] = (q1[m, :2] + off[m]).astype(F32)
    q2[m, :2] = (q2[m, :2] + off[m] + (tilt[m] if True else 0)).astype(F32)
    thr = k == 3                                                 # right at the threshold: d = sqrt(2 ln 2) sigma
    d0 = (np.sqrt(2.0 * np.log(2.0)) * sig * (1.0 + rng.normal(0, 1e-6, n)))[:, None] * nrm
    q1[thr, :2] = (q1[thr, :2] + d0[thr]).astype(F32)
    q2[thr, :2] = (q2[thr, :2] + d0[thr]).astype(F32)
    return p1, p2, q1, q2, (sig * sig).astype(F32)


def confidence_cases(rng, n):
    """D_hypothesis_confidence as K_verify_matches calls it: a source segment p1 p2, the hypothesis P1 P2 and a witness Q1 Q2 (3-D), the camera
    centre, the witness's target segment, (sigma_p, sigma_a, spatial_k).  Witnesses near the hypothesis (inside, at and outside the
    gate k * depth), parallel / tilted / reversed directions (the angle term), target segments near the source line (the distance term),
    and the gate switched off (spatial_k = 0)."""
    p1, p2 = _pts2d(rng, n), _pts2d(rng, n)
    C = points3d(rng, n, 2.0)
    P1, P2 = points3d(rng, n, 4.0), points3d(rng, n, 4.0)
    par = np.empty((n, 3), F32)
    par[:, 0] = rng.choice(np.array([2.5, 1.0, 5.0], F32), n)
    par[:, 1] = rng.choice(np.array([10.0, 5.0, 20.0], F32), n)
    par[:, 2] = rng.choice(np.array([0.005, 0.02, 0.0, 0.1], F32), n)
    k = rng.integers(0, 6, n)
    depth1 = np.linalg.norm(C - P1, axis=1)
    depth2 = np.linalg.norm(C - P2, axis=1)
    u = rng.normal(0, 1, (n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = rng.normal(0, 1, (n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.where(k == 0, rng.random(n) * 2.0,                                         # inside .. twice the gate radius
                 np.where(k == 1, 1.0 + rng.normal(0, 1e-6, n), rng.random(n) * 0.5))   # right at it / well inside
    kk = np.where(par[:, 2] > 0, par[:, 2], 0.01)
    Q1 = (P1 + (f * kk * depth1)[:, None] * u).astype(F32)
    Q2 = (P2 + (f * kk * depth2)[:, None] * v).astype(F32)
    far = k == 5
    Q1[far] = points3d(rng, int(far.sum()), 4.0)
    rev = k == 4                                                                       # reversed witness: angle folded at 90 degrees
    Q1[rev], Q2[rev] = Q2[rev].copy(), Q1[rev].copy()
    tgt = np.empty((n, 4), F32)
    a = rng.normal(0, 1, n); b = a + rng.normal(0, 1, n)
    off = rng.normal(0, 1, n) * par[:, 0] * 1.5
    d = (p2[:, :2] - p1[:, :2]).astype(np.float64)
    ln = np.linalg.norm(d, axis=1, keepdims=True); ln[ln == 0] = 1.0
    nrm = np.stack([-d[:, 1], d[:, 0]], 1) / ln
    tgt[:, 0:2] = (p1[:, :2] + a[:, None] * d + off[:, None] * nrm).astype(F32)
    tgt[:, 2:4] = (p1[:, :2] + b[:, None] * d + (off + rng.normal(0, 0.5, n))[:, None] * nrm).astype(F32)
    unrelated = rng.integers(0, 5, n) == 0
    tgt[unrelated] = (rng.random((int(unrelated.sum()), 4)) * 1500).astype(F32)
    return p1, p2, P1.astype(F32), P2.astype(F32), Q1, Q2, C.astype(F32), tgt, par


def pairwise_cases(rng, n):
    """The middle of K_pairwise_matches: source segment p1 p2, target segment q1 q2 and the four epipolar lines (of p1, p2 in the target image,
    of q1, q2 in the source image).  A line through a chosen point of the other segment's line and an "epipole" somewhere: the intersection
    parameters are drawn around [0, 1] (inside, at the ends, outside, reversed, far away), epipoles inside and far outside the image, plus
    lines parallel to the segment (intersection at infinity: the validity test), degenerate lines and sub-pixel segments."""
    p1, p2, q1, q2 = _pts2d(rng, n), _pts2d(rng, n), _pts2d(rng, n), _pts2d(rng, n)
    short = rng.integers(0, 12, n) == 0
    p2[short, :2] = p1[short, :2] + rng.normal(0, 0.6, (int(short.sum()), 2)).astype(F32)

    def line_through(a1, a2, t, epi):
        x = a1[:, :2].astype(np.float64) + t[:, None] * (a2[:, :2].astype(np.float64) - a1[:, :2])
        X = np.concatenate([x, np.ones((n, 1))], 1)
        E = np.concatenate([epi, np.ones((n, 1))], 1)
        return np.cross(X, E).astype(F32)

    def params():
        k = rng.integers(0, 6, n)
        t = rng.normal(0.5, 0.6, n)
        t = np.where(k == 0, rng.choice([0.0, 1.0], n) + rng.normal(0, 1e-4, n), t)
        t = np.where(k == 1, rng.normal(0.5, 30.0, n), t)
        return t
    far = rng.integers(0, 3, n) == 0
    ep_t = np.where(far[:, None], rng.normal(0, 1e5, (n, 2)), rng.random((n, 2)) * 2000)    # epipole in the target / source image
    ep_s = np.where(far[:, None], rng.normal(0, 1e5, (n, 2)), rng.random((n, 2)) * 2000)
    ta, tb = params(), params()
    swap = rng.integers(0, 4, n) == 0
    tb = np.where(swap, ta - np.abs(tb - ta), ta + np.abs(tb - ta) * 0.8)
    e1, e2 = line_through(q1, q2, ta, ep_t), line_through(q1, q2, tb, ep_t)
    tc, td = params(), params()
    e3, e4 = line_through(p1, p2, tc, ep_s), line_through(p1, p2, td, ep_s)
    k = rng.integers(0, 20, n)
    par = k == 0                                                   # parallel to the segment's line: z of the cross product ~ 0
    e1[par] = np.cross(q1[par].astype(np.float64), q2[par].astype(np.float64)).astype(F32) * F32(0.5)
    e1[par, 2] += F32(3.0)
    e3[k == 1] = 0.0                                               # degenerate line
    return p1, p2, q1, q2, e1, e2, e3, e4


def matrices(rng, n, stride):
    M = np.zeros((n, 3, stride), F32)
    M[:, :, :3] = rng.normal(0, 1, (n, 3, 3)).astype(F32)
    M[:, :, 3:] = F32(7777.0)                  # padding must never be read
    return M


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def make_inputs(seed, n):
    """The seeded cases of every pinned function: {name: (inputs tuple, output dtype/shape, extra int args)}."""
    rng = np.random.default_rng(seed)
    c = {}
    c["segment_overlap_2D"] = (overlap_cases(rng, n), (np.float32, (n,)), "segment_overlap_2D", None)
    c["point_on_segment_2D"] = (point_on_segment_cases(rng, n), (np.int32, (n,)), "point_on_segment_2D", None)
    p1, p2, _, _ = overlap_cases(rng, n)
    c["segment_length_2D"] = ((p1, p2), (F32, (n,)), "segment_length_2D", None)
    c["distance_p2l_2D"] = (lines_and_points(rng, n), (F32, (n,)), "distance_p2l_2D", None)
    c["angle_between_lines_deg_3D"] = (angle_cases(rng, n), (F32, (n,)), "angle_between_lines_deg_3D", None)
    c["normalize_hom_coords_2D"] = ((hom_points(rng, n),), (F32, (n, 3)), "normalize_hom_coords_2D", None)
    for stride in (3, 8):                      # the reference's host stride for RtKinv is 8 floats
        p = _pts2d(rng, n)
        M = matrices(rng, n, stride)
        c["get_ray_src_stride%d" % stride] = ((p, M), (F32, (n, 3)), "get_ray_src", stride)
        Cc, depth = points3d(rng, n), (rng.random(n) * 10).astype(F32)
        c["unproject_point_src_stride%d" % stride] = ((p, Cc, depth, M), (F32, (n, 3)), "unproject_point_src", stride)
    c["collinearity_pair"] = (collinearity_cases(rng, n), (F32, (n,)), "collinearity_pair", None)
    c["hypothesis_confidence"] = (confidence_cases(rng, n), (F32, (n,)), "hypothesis_confidence", None)
    c["pairwise_overlap"] = (pairwise_cases(rng, n), (F32, (n, 13)), "pairwise_overlap", None)
    v = points3d(rng, n)
    v[rng.integers(0, 10, n) == 0] = 0.0
    c["normalize3"] = ((v,), (F32, (n, 3)), "normalize3", None)
    c["cross3"] = ((points3d(rng, n), points3d(rng, n)), (F32, (n, 3)), "cross3", None)
    c["length3"] = ((points3d(rng, n, 1e3),), (F32, (n,)), "length3", None)
    c["dot3"] = ((points3d(rng, n), points3d(rng, n)), (F32, (n,)), "dot3", None)
    return {k: (tuple(np.ascontiguousarray(a) for a in ins), o, fn, st) for k, (ins, o, fn, st) in c.items()}


# the three kernel BODIES: reference lines inside builder-written function heads (oracle/_spliced/libkernels_spliced.so, corroboration);
# everything else is unmodified reference text (oracle/_ref/libdevfn_ref.so)
SPLICED = ("collinearity_pair", "hypothesis_confidence", "pairwise_overlap")


def run(lib, prefix, cases, only=None, skip=()):
    """Every pinned function of `lib` on `cases` (make_inputs); only / skip: case names.  Returns {name: (inputs tuple, output)}."""
    res = {}
    for name, (ins, (dt, shape), fname, stride) in cases.items():
        if (only is not None and name not in only) or name in skip:
            continue
        fn = getattr(lib, prefix + fname)
        fn.restype = None
        out = np.zeros(shape, dt)
        extra = () if stride is None else (C.c_int(stride),)
        fn(C.c_int(shape[0]), *[_fp(a) for a in ins], *extra, _fp(out))
        res[name] = (ins, out)
    consts = []
    for nm in ("eps_g", "min_overlap_lower", "min_overlap_upper", "collin_aff_t"):
        fn = getattr(lib, prefix + nm)
        fn.restype = C.c_float
        consts.append(fn())
    res["constants"] = ((), np.asarray(consts, F32))
    return res


def run_all(lib, prefix, seed, n):
    return run(lib, prefix, make_inputs(seed, n))


def run_reference(clean, spliced, cases):
    """The reference side of the pins: unmodified text out of oracle/_ref/libdevfn_ref.so (`clean`), the three kernel bodies out of
    oracle/_spliced/libkernels_spliced.so (`spliced`; None: left out)."""
    res = run(clean, "l3dref_", cases, skip=SPLICED)
    if spliced is not None:
        res.update({k: v for k, v in run(spliced, "l3dref_", cases, only=SPLICED).items() if k != "constants"})
    return res


def same_bits(a, b):
    """bit-equal, NaNs of any payload counted equal (0/0 has no defined payload across compilers)"""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype.kind != "f":
        return np.array_equal(a, b)
    ai, bi = a.view(np.uint32), b.view(np.uint32)
    return bool(np.all((ai == bi) | (np.isnan(a) & np.isnan(b))))


def collinear_segments(seed, n_lines=60, pieces=5, sigma=2.5):
    """An image's segments with planted collinearities for the K_collinearity pin: every line is cut into pieces -- with gaps, touching end points,
    overlaps (the conflict check) and sideways jitter of 0 .. 2 sigma -- plus unrelated segments.  float32 (S, 4)."""
    rng = np.random.default_rng(seed)
    segs = []
    for _ in range(n_lines):
        o = rng.random(2) * [1700, 900] + [100, 90]
        a = rng.random() * np.pi
        u, nrm = np.array([np.cos(a), np.sin(a)]), np.array([-np.sin(a), np.cos(a)])
        t = 0.0
        for _ in range(pieces):
            length = rng.uniform(15, 120)
            kind = rng.integers(0, 4)
            t += 0.0 if kind == 0 else (rng.uniform(1, 40) if kind <= 2 else -rng.uniform(1, 30))      # touching / gap / overlap
            j1, j2 = rng.uniform(-2, 2, 2) * sigma * (rng.random() < 0.7)
            segs.append(np.concatenate([o + t * u + j1 * nrm, o + (t + length) * u + j2 * nrm]))
            t += length
    for _ in range(n_lines):
        p = rng.random(2) * [1700, 900] + [100, 90]
        segs.append(np.concatenate([p, p + rng.normal(0, 50, 2)]))
    segs = np.array(segs, F32)
    return np.ascontiguousarray(segs[rng.permutation(len(segs))])
