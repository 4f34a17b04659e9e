"""CPU property test of the interval bounds behind level 2 of k_pair_mask (line3d_amd/csrc/l3d_kernels.hip: iou_bounds, the
rejects of round 1 and the ACCEPTS of round 4): a numpy float32 restatement of the bounds, tile by tile like the kernel (the margins
depend on the coordinate extents of a workgroup's 64 source and 256 target segments), is held against the oracle's exact test
(oracle/l3d_oracle.c: pairwise_overlap, the reference's cudawrapper.cu:569-588) on pairs built to sit ON the decision boundaries:

    bound says "reject"  =>  the exact test rejects          bound says "accept"  =>  the exact test accepts

The kernel evaluates the same formulas with FMAs and hardware reciprocals; the slack of the bounds (1e-3 on the thresholds, 1e-2 on
the conditioning) is four orders of magnitude above what that changes.  The GPU tests (tests/test_gpu_seam_parity.py:
test_pair_pretest_is_conservative and friends) hold the kernel itself to identical bit rows with the bounds on and off."""
import os

import numpy as np

import l3d_oracle_pipeline as op
from line3d_amd.synth import make_scene

F32 = np.float32
K_WEDGE_TAU, K_IOU_COND, K_IOU_SLACK = F32(1.0e-4), F32(1.0e-2), F32(1.0e-3)      # l3d_kernels.hip: kWedgeTau, kIouCond, kIouSlack
K_LINE_COND, K_EDGE_GUARD = F32(5.0e-7), F32(4.0)                                     # kLineCond; the end-point guard band of the accepts, in units of e
MIN_LOWER, MIN_UPPER = F32(0.10), F32(0.30)                                       # cudawrapper.h:45-46


def _line_apply(l, x, y):
    """l.x * x + l.y * y + l.z (one rounding per FMA in the kernel; here: double, rounded once)"""
    return (l[..., 0].astype(np.float64) * x + l[..., 1].astype(np.float64) * y + l[..., 2]).astype(F32)


def _iou_bounds(t1, r1, t2, r2, length, ext_over_len):
    """iou_bounds of l3d_kernels.hip, vectorised: (upper, lower); 2 / -1 = cannot tell"""
    with np.errstate(all="ignore"):
        e0 = (K_LINE_COND * ext_over_len * ext_over_len + F32(1e-6) * ext_over_len).astype(F32)
        e1 = (K_IOU_COND * (F32(1) + np.abs(t1)) * np.abs(r1) + F32(1e-6) * np.abs(t1) + e0).astype(F32)
        e2 = (K_IOU_COND * (F32(1) + np.abs(t2)) * np.abs(r2) + F32(1e-6) * np.abs(t2) + e0).astype(F32)
        e = np.maximum(e1, e2)
        ill = ~(e < F32(10))
        lo, hi = np.minimum(t1, t2), np.maximum(t1, t2)
        short = (hi - lo + F32(2) * e) * length < F32(1) - K_IOU_SLACK
        in0 = np.minimum(hi, F32(1)) - np.maximum(lo, F32(0))
        un0 = np.maximum(hi, F32(1)) - np.minimum(lo, F32(0))
        uni_lo = un0 - F32(2) * e
        upper = np.where(uni_lo > 0, (in0 + F32(2) * e) / uni_lo * F32(1 + 1e-5), F32(2)).astype(F32)
        edge = np.minimum(np.minimum(np.abs(lo), np.abs(lo - F32(1))), np.minimum(np.abs(hi), np.abs(hi - F32(1))))
        near_end = ~(edge > K_EDGE_GUARD * e)                                  # an intersection point on an end point, within the error: cannot tell
        sure = (e < F32(0.1)) & ((hi - lo - F32(2) * e) * length > F32(1) + K_IOU_SLACK) & (in0 - F32(2) * e > 0)
        lower = np.where(sure, (in0 - F32(2) * e) / (un0 + F32(2) * e) * F32(1 - 1e-5), F32(-1)).astype(F32)
        upper = np.where(near_end, F32(2), upper)
        lower = np.where(near_end, F32(-1), lower)
        upper = np.where(short, F32(0), upper)
        lower = np.where(short, F32(-1), lower)
        upper = np.where(ill, F32(2), upper)
        lower = np.where(ill, F32(-1), lower)
    return upper, lower


def _level2(src, tgt, F):
    """(reject, accept) of every (source, target) pair the way one k_pair_mask workgroup decides them: src (<= 64) x tgt (<= 256)"""
    src, tgt = src.astype(F32), tgt.astype(F32)
    Fm = F.reshape(3, 3).astype(F32)
    one = lambda a: np.concatenate([a, np.ones((len(a), 1), F32)], 1)
    p1, p2, q1, q2 = one(src[:, 0:2]), one(src[:, 2:4]), one(tgt[:, 0:2]), one(tgt[:, 2:4])
    epi_p1, epi_p2 = (p1 @ Fm.T).astype(F32), (p2 @ Fm.T).astype(F32)                  # F p   (lines in the target image)
    epi_q1, epi_q2 = (q1 @ Fm).astype(F32), (q2 @ Fm).astype(F32)                      # F^T q (lines in the source image)
    ext0 = np.abs(tgt[:, [0, 2]]).max(); ext1 = np.abs(tgt[:, [1, 3]]).max()           # the tile's targets
    ext2 = np.abs(src[:, [0, 2]]).max(); ext3 = np.abs(src[:, [1, 3]]).max()           # the block's sources
    ext = F32(max(ext0, ext1, ext2, ext3))
    marg = lambda l, ex, ey: (K_WEDGE_TAU * (np.abs(l[:, 0]) * ex + np.abs(l[:, 1]) * ey + np.abs(l[:, 2]))).astype(F32)
    with np.errstate(all="ignore"):
        se1 = (epi_p1 / marg(epi_p1, ext0, ext1)[:, None]).astype(F32); se2 = (epi_p2 / marg(epi_p2, ext0, ext1)[:, None]).astype(F32)
        te1 = (epi_q1 / marg(epi_q1, ext2, ext3)[:, None]).astype(F32); te2 = (epi_q2 / marg(epi_q2, ext2, ext3)[:, None]).astype(F32)
        S, T = len(src), len(tgt)
        # b: the target's lines at the source endpoints; a: the source's lines at the target endpoints   [S, T]
        b1 = _line_apply(te1[None, :, :], p1[:, None, 0], p1[:, None, 1]); b2 = _line_apply(te1[None, :, :], p2[:, None, 0], p2[:, None, 1])
        b3 = _line_apply(te2[None, :, :], p1[:, None, 0], p1[:, None, 1]); b4 = _line_apply(te2[None, :, :], p2[:, None, 0], p2[:, None, 1])
        a1 = _line_apply(se1[:, None, :], q1[None, :, 0], q1[None, :, 1]); a2 = _line_apply(se1[:, None, :], q2[None, :, 0], q2[None, :, 1])
        a3 = _line_apply(se2[:, None, :], q1[None, :, 0], q1[None, :, 1]); a4 = _line_apply(se2[:, None, :], q2[None, :, 0], q2[None, :, 1])
        rb1, rb2 = (F32(1) / (b1 - b2)).astype(F32), (F32(1) / (b3 - b4)).astype(F32)
        ra1, ra2 = (F32(1) / (a1 - a2)).astype(F32), (F32(1) / (a3 - a4)).astype(F32)
        ls = np.sqrt(((src[:, 0] - src[:, 2]) ** 2 + (src[:, 1] - src[:, 3]) ** 2).astype(F32)).astype(F32)[:, None] * np.ones((1, T), F32)
        lt = np.sqrt(((tgt[:, 0] - tgt[:, 2]) ** 2 + (tgt[:, 1] - tgt[:, 3]) ** 2).astype(F32)).astype(F32)[None, :] * np.ones((S, 1), F32)
        u1, l1 = _iou_bounds(b1 * rb1, rb1, b3 * rb2, rb2, ls, ext / ls)
        u2, l2 = _iou_bounds(a1 * ra1, ra1, a3 * ra2, ra2, lt, ext / lt)
        rej = (np.maximum(u1, u2) < MIN_UPPER - K_IOU_SLACK) | (np.minimum(u1, u2) < MIN_LOWER - K_IOU_SLACK)
        acc = (ls >= 1) & (lt >= 1) & (np.minimum(l1, l2) > MIN_LOWER + K_IOU_SLACK) & (np.maximum(l1, l2) > MIN_UPPER + K_IOU_SLACK)
    return rej, acc & ~rej


def _exact(lib, mv, cam, src, tgt):
    """the oracle's decision for every pair: the dense buffer holds (0, 0, 0, 0) where the overlap test fails (cudawrapper.cu:586-590)"""
    buf = op.pairwise_dense(lib, np.ascontiguousarray(src, F32), mv["RtKinv_src"], mv["C_src"], np.ascontiguousarray(tgt, F32), 0, len(tgt), cam,
                            mv["F"], mv["RtKinv"], mv["centers"])
    return np.any(buf.reshape(len(src), len(tgt), 4) != 0, axis=2)


def _boundary_targets(rng, src, F, n, w, h, transpose=False, only_kind=None, src_of=None):
    """target segments whose end points lie on the epipolar lines of the source points p(s1), p(s2), with (s1, s2) chosen so that the
    overlap of [s1, s2] with the source segment sits on / next to the thresholds 0.1 and 0.3, at the 1-pixel limit, or far outside
    (transpose: the roles swapped -- source segments built on the epipolar lines F^T q of points of the given target segments)"""
    Fm = F.reshape(3, 3).astype(np.float64)
    if transpose:
        Fm = Fm.T
    out = np.zeros((n, 4), F32)
    for i in range(n):
        s = src[src_of(i) if src_of else rng.integers(0, len(src))].astype(np.float64)
        kind = rng.integers(0, 11) if only_kind is None else only_kind
        d = float(rng.choice([0.0, 1e-6, 1e-5, 1e-4, 5e-4, 1e-3, 2e-3, 1e-2])) * float(rng.choice([-1, 1]))
        if only_kind is not None:
            d = 0.0
        thr = float(rng.choice([0.1, 0.3]))
        if kind == 0:      # inside the segment, length = thr (+d): overlap = thr
            a = rng.uniform(0, 1 - thr); s1, s2 = a, a + thr + d
        elif kind == 1:    # covers the segment, 1 / length = thr
            L = 1.0 / thr + d * 10; a = rng.uniform(-(L - 1), 0); s1, s2 = a, a + L
        elif kind == 2:    # sticks out beyond p2: [s1, s2] = [1 - i, 1 + o] with i / (1 + o) = thr
            o_ = rng.uniform(0.2, 3.0); i_ = min(1.0, thr * (1 + o_) + d); s1, s2 = 1 - i_, 1 + o_
        elif kind == 3:    # about a pixel long in the source image
            Ls = max(1e-3, np.hypot(s[0] - s[2], s[1] - s[3])); a = rng.uniform(0, 1); s1, s2 = a, a + (1.0 + d * 100) / Ls
        elif kind == 4:    # just outside / just touching an end
            s1, s2 = -rng.uniform(0, 2), d * 10
        elif kind >= 8:    # one end point of the pair ON an end point of the segment (+- d), the other far enough for a large overlap: the reference's
            # point-on-segment tests (cudawrapper.cu:135-141) can BOTH fail there on float noise, and the overlap drops to 0
            end = float(rng.integers(0, 2)); far = rng.uniform(0.35, 0.95) if rng.random() < 0.5 else rng.uniform(1.2, 3.0)
            s1 = end + d * float(rng.choice([1, 0.1, 10])); s2 = end + (far if end == 0 else -far)
        elif kind >= 6:    # ill-conditioned: both end points on (nearly) the same epipolar line -- the segment runs along the epipolar direction
            s1 = rng.uniform(-0.3, 1.3); s2 = s1 + float(rng.choice([0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2])) * float(rng.choice([-1, 1]))
        else:              # anywhere
            s1, s2 = rng.uniform(-1.5, 2.5, 2)
        if rng.random() < 0.5:
            s1, s2 = s2, s1
        q = []
        for sv in (s1, s2):
            p = np.array([(1 - sv) * s[0] + sv * s[2], (1 - sv) * s[1] + sv * s[3], 1.0])
            l = Fm @ p                                                         # epipolar line of p(s) in the target image
            n2 = np.hypot(l[0], l[1])
            if n2 < 1e-12:
                q.append((rng.uniform(0, w), rng.uniform(0, h)))
                continue
            # a point of the line near the image: foot of a random image point
            x0, y0 = rng.uniform(0, w), rng.uniform(0, h)
            dist = (l[0] * x0 + l[1] * y0 + l[2]) / n2
            q.append((x0 - dist * l[0] / n2, y0 - dist * l[1] / n2))
        out[i] = (q[0][0], q[0][1], q[1][0], q[1][1])
    return out


def _check(lib, mv, cam, src, tgt, tag):
    exact = _exact(lib, mv, cam, src, tgt)
    n_rej = n_acc = 0
    for s0 in range(0, len(src), 64):                                           # a workgroup: 64 sources x 256 targets
        for t0 in range(0, len(tgt), 256):
            rej, acc = _level2(src[s0:s0 + 64], tgt[t0:t0 + 256], mv["F"][cam])
            ex = exact[s0:s0 + 64, t0:t0 + 256]
            bad_r, bad_a = rej & ex, acc & ~ex
            assert not bad_r.any(), "%s: %d pairs rejected by the bound pass the exact test, first %s" % (tag, bad_r.sum(), np.argwhere(bad_r)[0] + (s0, t0))
            assert not bad_a.any(), "%s: %d pairs accepted by the bound fail the exact test, first %s" % (tag, bad_a.sum(), np.argwhere(bad_a)[0] + (s0, t0))
            n_rej += int(rej.sum()); n_acc += int(acc.sum())
    return n_rej, n_acc, int(exact.sum()), exact.size


def test_level2_bounds_never_contradict_the_exact_test(oracle_lib):
    rng = np.random.default_rng(4242)
    tot = np.zeros(4, np.int64)
    for seed, step in ((11, 0.12), (12, 0.02), (13, 0.45)):                     # ordinary, tiny and wide baselines
        sc = make_scene(8, 256, 4, seed=seed, step=step)
        o = op.OracleLine3D(matching_neighbors=4, use_collinearity=False)
        for v in sc.views:
            o.add_image_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
        o.matched = {}
        o.find_visual_neighbors()
        o.transform_geometry()
        vid = sorted(o.views)[3]
        for nb in o.visual_neighbors[vid]:
            o._fundamental(vid, nb)
        mv = o.marshal_view(vid)
        src = mv["src_segs"][:192]
        for cam in range(min(2, len(mv["F"]))):
            off, wdt = mv["offsets"][cam]
            real = mv["tgt_segs"][off:off + wdt]
            # the scene's own pairs, and pairs built on the boundaries
            tot += _check(oracle_lib, mv, cam, src, real, "scene %d cam %d (own segments)" % (seed, cam))
            built = _boundary_targets(rng, src, mv["F"][cam], 4096, 1920, 1080)
            tot += _check(oracle_lib, mv, cam, src, built, "scene %d cam %d (boundary targets)" % (seed, cam))
            # the same from the other side: source segments on the epipolar lines of points of the scene's target segments
            built_src = _boundary_targets(rng, real, mv["F"][cam], 512, 1920, 1080, transpose=True)
            tot += _check(oracle_lib, mv, cam, built_src, real, "scene %d cam %d (boundary sources)" % (seed, cam))
            tot += _check(oracle_lib, mv, cam, built_src, built[:768], "scene %d cam %d (boundary sources x boundary targets)" % (seed, cam))
            # sub-pixel and near-degenerate segments among them
            tiny = built.copy()
            tiny[:, 2:4] = tiny[:, 0:2] + rng.uniform(-1.5, 1.5, (len(tiny), 2)).astype(F32)
            tot += _check(oracle_lib, mv, cam, src, tiny[:512], "scene %d cam %d (pixel-sized targets)" % (seed, cam))
    n_rej, n_acc, n_exact, n_all = (int(x) for x in tot)
    # the bounds decide most pairs, in both directions (otherwise this test would hold vacuously)
    assert n_all > 7_000_000 and n_all - n_rej - n_acc < 0.02 * n_all and n_acc > 0.5 * n_exact and n_rej > 0.5 * (n_all - n_exact), (n_rej, n_acc, n_exact, n_all)


def test_level2_accepts_keep_away_from_coincident_end_points(oracle_lib):
    """True correspondences put an end point of the target ON the epipolar line of an end point of the source.  Where the float intersection point
    then lands within ~1e-7 px of the segment's end along the line and ~1e-3 px beside it, BOTH of the reference's point-on-segment tests fail
    (cudawrapper.cu:135-141: a dot product against 1e-12) and D_segment_overlap_2D returns 0 for a pair whose intervals overlap by a third:
    found by the diagnostic build of k_pair_mask (-DL3D_BOUND_CHECK) on 512 x 2000 x 12, three pairs in 1.2e10.  The accepts stay a guard band
    away from the end points; this test builds 1e5 such coincidences and holds the bounds to the exact test on every pair around them."""
    rng = np.random.default_rng(777)
    sc = make_scene(8, 256, 4, seed=21)
    o = op.OracleLine3D(matching_neighbors=4, use_collinearity=False)
    for v in sc.views:
        o.add_image_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
    o.matched = {}
    o.find_visual_neighbors()
    o.transform_geometry()
    vid = sorted(o.views)[4]
    for nb in o.visual_neighbors[vid]:
        o._fundamental(vid, nb)
    mv = o.marshal_view(vid)
    tot = np.zeros(4, np.int64)
    for cam in range(min(2, len(mv["F"]))):
        for rep in range(200):
            src = mv["src_segs"][rng.permutation(len(mv["src_segs"]))[:64]]
            built = _boundary_targets(rng, src, mv["F"][cam], 256, 1920, 1080, only_kind=8, src_of=lambda i: i % 64)
            tot += _check(oracle_lib, mv, cam, src, built, "cam %d rep %d (coincident end points)" % (cam, rep))
    assert tot[3] > 6_000_000 and tot[1] > 10_000, tot


def test_bounds_on_the_pairs_that_fooled_them():
    """tests/golden/endpoint_quirk_pairs.npz (make_golden_endpoint_quirk.py): the workgroup tiles around four pairs that the bounds decided against the
    exact test before they kept away from segment end points -- three of the 512 x 2000 x 12 scene the first accepts let through (the exact test rejects
    them: overlap 0), one of 256 x 4000 x 24 that the reject of rounds 1-3 dropped (the exact test keeps it: overlap 3791).  An intersection point on
    an end point of a segment, the point-on-segment tests flipping on float noise.  With the guard band all four are left to the exact test, and no pair
    of their tiles is decided against it."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "endpoint_quirk_pairs.npz"))
    verdicts = []
    for i in range(4):
        src, tgt, F, exact = g["src_%d" % i], g["tgt_%d" % i], g["F_%d" % i], g["exact_%d" % i]
        y, x = (int(v) for v in g["pair_%d" % i])
        verdicts.append(bool(exact[y, x]))
        rej, acc = _level2(src, tgt, F)
        assert not acc[y, x] and not rej[y, x], "case %d: the pair is decided by the bounds" % i
        assert not (rej & exact).any() and not (acc & ~exact).any(), "case %d" % i
    assert verdicts == [False, False, False, True]
