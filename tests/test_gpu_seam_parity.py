"""GPU parity of the three seam functions (+ similarity batch) against the CPU oracle, through the C ABI.
Bar: bit-exact (match ids, depths, confidences, median; collinearity weights; diffusion values)."""
import os
import numpy as np
import pytest

import l3d_oracle_pipeline as op

pytestmark = pytest.mark.gpu


def test_contract_math_bit_exact(gpu_ctx, oracle_lib):
    rng = np.random.default_rng(3)
    x = np.concatenate([-rng.random(20000) * 3.0, -rng.random(2000) * 90.0, rng.random(20000) * 2 - 1,
                        [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, -87.5, -86.9]]).astype(np.float32)
    e, ac, acd = gpu_ctx.test_contract_math(x)
    xc = np.clip(x, -1, 1)
    e_o = np.array([oracle_lib.l3do_test_expf(float(v)) for v in x], dtype=np.float32)
    ac_o = np.array([oracle_lib.l3do_test_acosf(float(v)) for v in xc], dtype=np.float32)
    acd_o = np.array([oracle_lib.l3do_test_acos(float(v)) for v in xc], dtype=np.float64)
    assert np.array_equal(e.view(np.uint32), e_o.view(np.uint32))
    assert np.array_equal(ac.view(np.uint32), ac_o.view(np.uint32))
    assert np.array_equal(acd.view(np.uint64), acd_o.view(np.uint64))


def test_collinearity_matches_oracle(gpu_ctx, oracle_lib, small_scene):
    for v in small_scene.views[:3]:
        segs = v["segments"]
        rel = op.collinearity(oracle_lib, segs, 2.0)
        ii, jj = np.nonzero(np.triu(rel > 0, 1))
        gi, gj, gw = gpu_ctx.compute_collinearity(segs, 2.0)
        assert np.array_equal(gi, ii.astype(np.int32)) and np.array_equal(gj, jj.astype(np.int32))
        assert np.array_equal(gw.view(np.uint32), rel[jj, ii].view(np.uint32))


def test_collinearity_structured(gpu_ctx, oracle_lib):
    # collinear but non-overlapping pieces of the same lines, plus overlapping and crossing ones
    segs = []
    for k in range(40):
        x0, y0, dx, dy = 50.0 + 13 * k, 30.0 + 7 * k, np.cos(0.1 * k), np.sin(0.1 * k)
        for a, b in ((0, 40), (55, 90), (30, 60), (120, 200)):
            segs.append([x0 + a * dx, y0 + a * dy, x0 + b * dx, y0 + b * dy])
    segs = np.array(segs, dtype=np.float32)
    rel = op.collinearity(oracle_lib, segs, 2.0)
    ii, jj = np.nonzero(np.triu(rel > 0, 1))
    assert len(ii) > 40
    gi, gj, gw = gpu_ctx.compute_collinearity(segs, 2.0)
    assert np.array_equal(gi, ii.astype(np.int32)) and np.array_equal(gj, jj.astype(np.int32))
    assert np.array_equal(gw.view(np.uint32), rel[jj, ii].view(np.uint32))


def test_collinearity_batch_equals_per_set_calls(gpu_ctx, oracle_lib, small_scene):
    """l3d_compute_collinearity_batch (what prepare() uses for all views at once): sets of different sizes, including an empty one
    and a single segment, against the per-set call and the oracle."""
    sets = [v["segments"] for v in small_scene.views[:4]]
    sets = [sets[0], sets[1][:1], np.zeros((0, 4), np.float32), sets[2][:77], sets[3]]
    got = gpu_ctx.compute_collinearity_batch(sets, 2.0)
    assert len(got) == len(sets)
    for segs, (gi, gj, gw) in zip(sets, got):
        if len(segs) < 2:
            assert len(gi) == 0
            continue
        si, sj, sw = gpu_ctx.compute_collinearity(segs, 2.0)
        assert np.array_equal(gi, si) and np.array_equal(gj, sj) and np.array_equal(gw.view(np.uint32), sw.view(np.uint32))
        rel = op.collinearity(oracle_lib, segs, 2.0)
        ii, jj = np.nonzero(np.triu(rel > 0, 1))
        assert np.array_equal(gi, ii.astype(np.int32)) and np.array_equal(gj, jj.astype(np.int32))
        assert np.array_equal(gw.view(np.uint32), rel[jj, ii].view(np.uint32))


def _run_view(ctx, tr, seg_range=None):
    mv = tr["marshal"]
    return ctx.compute_pairwise_matches(mv["src_segs"], mv["RtKinv_src"], mv["C_src"], mv["tgt_segs"], mv["offsets"],
                                        mv["F"], mv["RtKinv"], mv["centers"], mv["P"], mv["tbm"], tr["in_matches"],
                                        mv["l2g"], mv["k_upper"], mv["k_lower"], 3.5, 10.0, mv["spatial_k"],
                                        median_depth=1.0, seg_range=seg_range, want_best=True)


def test_pairwise_matches_bit_exact_per_view(gpu_ctx, small_oracle):
    total = 0
    for v in sorted(small_oracle.trace):
        tr = small_oracle.trace[v]
        got, med, _ = _run_view(gpu_ctx, tr)
        exp = tr["matches"]
        assert len(got) == len(exp), "view %d: %d vs %d kept" % (v, len(got), len(exp))
        assert got.tobytes() == exp.tobytes(), "view %d differs" % v
        assert np.float32(med) == np.float32(tr["median"])
        total += len(got)
    assert total > 1000


def test_pairwise_matches_segment_ranges_concatenate(gpu_ctx, small_oracle):
    """Sharding by source-segment range is exact: the ranges' outputs concatenate to the full output
    and the merged best-depth lists give the same median."""
    tr = small_oracle.trace[4]
    full, med, best = _run_view(gpu_ctx, tr)
    S = len(tr["marshal"]["src_segs"])
    parts, bests = [], []
    for s0, s1 in ((0, 97), (97, 200), (200, S)):
        m, _, b = _run_view(gpu_ctx, tr, (s0, s1))
        parts.append(m)
        bests.append(b)
    cat = np.concatenate(parts)
    assert cat.tobytes() == full.tobytes()
    allb = np.sort(np.concatenate(bests))
    assert np.array_equal(np.sort(best), allb)
    assert np.float32(allb[len(allb) // 2]) == np.float32(med)


def test_pairwise_matches_edge_cases(gpu_ctx, small_oracle):
    tr = small_oracle.trace[9]            # last view: nothing to match -> list returned untouched
    assert len(tr["marshal"]["tbm"]) == 0
    got, med, _ = _run_view(gpu_ctx, tr)
    assert got.tobytes() == tr["in_matches"].tobytes() and med == 1.0
    # empty source range
    tr = small_oracle.trace[2]
    got, med, best = _run_view(gpu_ctx, tr, (5, 5))
    assert len(got) == 0 and len(best) == 0


def test_rdd_matches_oracle(gpu_ctx, oracle_lib, small_oracle):
    A = small_oracle.affinity
    n = len(small_oracle.local2global)
    exp = op.rdd(oracle_lib, A, n, 10)
    got = gpu_ctx.replicator_dynamics_diffusion(A, n, 10)
    assert got.tobytes() == exp.tobytes()
    # toy matrix with ragged rows (positional product quirk is exercised when row/col lengths differ)
    rng = np.random.default_rng(5)
    e = []
    for i in range(30):
        for j in rng.choice(30, size=rng.integers(1, 6), replace=False):
            e.append((i, int(j), rng.random()))
    e = np.array(e, dtype=op.EDGE_DTYPE)
    rows = set(e["i"].tolist())
    e = np.concatenate([e, np.array([(k, k, 0.5) for k in range(30) if k not in rows], dtype=op.EDGE_DTYPE)])
    assert gpu_ctx.replicator_dynamics_diffusion(e, 30, 3).tobytes() == op.rdd(oracle_lib, e, 30, 3).tobytes()
    # a large unsymmetric list in random order: the host builds the column- and row-sorted matrices with a multi-threaded
    # ordering (sparsematrix.cc:81-86,157-167).  Keys are unique: with duplicate (i,j) entries the reference's diffusion
    # kernel has two threads writing the same slot (cudawrapper.cu:809-826), i.e. no defined result to compare with.
    n, E = 3000, 60000
    key = rng.choice(n * n, size=E, replace=False)
    e = np.zeros(E, dtype=op.EDGE_DTYPE)
    e["i"], e["j"] = key // n, key % n
    e["w"] = rng.random(E).astype(np.float32)
    assert gpu_ctx.replicator_dynamics_diffusion(e, n, 4).tobytes() == op.rdd(oracle_lib, e, n, 4).tobytes()


def test_similarity_batch_matches_oracle(gpu_ctx, small_oracle):
    from line3d_amd import capi
    o = small_oracle
    keys = sorted(o.best_match)
    hyp = np.zeros(len(keys), dtype=capi.HYP_DTYPE)
    for k, key in enumerate(keys):
        b = o.best_match[key]
        v = o.views[b["cam"]]
        hyp[k]["P1"], hyp[k]["P2"], hyp[k]["dir"] = b["seg3D"][0:3], b["seg3D"][3:6], b["seg3D"][6:9]
        hyp[k]["depth_p1"], hyp[k]["depth_p2"] = b["depths"]
        hyp[k]["k_lower"], hyp[k]["k_upper"], hyp[k]["median_depth"] = v.k_lower, v.k_upper, v.median_depth
    rng = np.random.default_rng(11)
    pairs = rng.integers(0, len(keys), size=(4000, 2)).astype(np.int32)
    idx = {key: k for k, key in enumerate(keys)}
    real = [(idx[s], idx[t]) for s in keys[:400] for t in o.potential.get(s, {}) if t in idx]
    pairs = np.concatenate([pairs, np.array(real, dtype=np.int32)])
    got = gpu_ctx.similarity_coll3D_batch(hyp, pairs, 10.0)
    exp = np.array([o._similarity(o.best_match[keys[a]], o.best_match[keys[b]]) for a, b in pairs], dtype=np.float32)
    assert (exp > 0).sum() > 100
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_against_committed_golden_vectors(gpu_ctx):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seam_small.npz"))
    for v in (0, 4):
        k = lambda name: g["v%d_%s" % (v, name)]  # noqa: E731
        sc = k("scalars")
        m, med = gpu_ctx.compute_pairwise_matches(k("src_segs"), k("RtKinv_src"), k("C_src"), k("tgt_segs"), k("offsets"), k("F"),
                                                  k("RtKinv"), k("centers"), k("P"), k("tbm"), k("in"), k("l2g"),
                                                  float(sc[0]), float(sc[1]), 3.5, 10.0, float(sc[2]))
        assert m.tobytes() == k("out").astype(op.MATCH_DTYPE).tobytes() and np.float32(med) == sc[3]
    gi, gj, gw = gpu_ctx.compute_collinearity(g["coll_segs"], 2.0)
    assert np.array_equal(gi, g["coll_i"]) and np.array_equal(gj, g["coll_j"]) and np.array_equal(gw, g["coll_w"])
    out = gpu_ctx.replicator_dynamics_diffusion(g["rdd_A"], int(g["rdd_n"]), 10)
    assert out.tobytes() == g["rdd_out"].astype(op.EDGE_DTYPE).tobytes()


def test_window_and_all_pairs_verify_agree_bitwise(gpu_ctx, small_oracle):
    """The depth-window search (default) and the all-pairs loop must produce identical kept lists."""
    for mode in (1, 0):
        gpu_ctx.set_verify_mode(mode)
        for v in (0, 3, 6):
            tr = small_oracle.trace[v]
            got, med, _ = _run_view(gpu_ctx, tr)
            assert got.tobytes() == tr["matches"].tobytes(), "mode %d view %d" % (mode, v)
    gpu_ctx.set_verify_mode(0)


def test_window_verify_stress_against_all_pairs(gpu_ctx):
    """Larger, noisier views (more near-threshold witnesses): window search vs all-pairs on the GPU, every
    confidence-derived output identical."""
    from line3d_amd.synth import make_scene
    for seed, noise in ((31, 0.5), (32, 2.0), (33, 0.05)):
        sc = make_scene(9, 700, 8, seed=seed, noise_px=noise)
        o = op.OracleLine3D(matching_neighbors=8, use_collinearity=False)
        for v in sc.views:
            o.add_image_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
        o.matched, o.potential = {}, {}
        o.find_visual_neighbors()
        o.transform_geometry()
        for n in o.visual_neighbors[4]:
            o._fundamental(4, n)
        mv = o.marshal_view(4)                                 # all 8 neighbours to be matched
        tr = dict(marshal=mv, in_matches=np.zeros(0, op.MATCH_DTYPE))
        gpu_ctx.set_verify_mode(1)
        a, ma, ba = _run_view(gpu_ctx, tr)
        gpu_ctx.set_verify_mode(0)
        b, mb, bb = _run_view(gpu_ctx, tr)
        assert len(a) > 500
        assert a.tobytes() == b.tobytes() and ma == mb and np.array_equal(ba, bb)


def test_pair_pretest_is_conservative(gpu_ctx):
    """The stage-1 filters (wedge test, depth-sign test) may only reject pairs the exact float sequence rejects: with
    none, either or both of them the number of raw candidates and every output must be identical (several noise
    levels and baselines -- tiny baselines make the triangulation rays nearly parallel --, ~10^8 pairs)."""
    from line3d_amd.synth import make_scene
    total_pairs = 0
    for seed, noise, S, step in ((41, 0.5, 1500, 0.12), (42, 3.0, 1200, 0.12), (43, 0.0, 1000, 0.12), (44, 10.0, 800, 0.12),
                                 (45, 0.5, 1000, 0.004), (46, 1.0, 1000, 0.0005), (47, 0.5, 1000, 0.45)):
        sc = make_scene(9, S, 8, seed=seed, noise_px=noise, step=step)
        o = op.OracleLine3D(matching_neighbors=8, use_collinearity=False)
        for v in sc.views:
            o.add_image_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
        o.matched, o.potential = {}, {}
        o.find_visual_neighbors()
        o.transform_geometry()
        for n in o.visual_neighbors[4]:
            o._fundamental(4, n)
        tr = dict(marshal=o.marshal_view(4), in_matches=np.zeros(0, op.MATCH_DTYPE))
        res = []
        for mask in (0, 1, 2, 3):
            gpu_ctx.set_pair_pretest(mask)
            m, med, best = _run_view(gpu_ctx, tr)
            st = gpu_ctx.last_stats()
            res.append((m.tobytes(), med, best.tobytes(), st[1]))
            total_pairs += st[0]
        gpu_ctx.set_pair_pretest(3)
        assert res[0] == res[1] == res[2] == res[3], "seed %d" % seed
        if step == 0.12:
            assert res[0][3] > 10000
    assert total_pairs > 1e8


def _sector_rejected(mv, cam, accepted):
    """Oracle-accepted pairs of (source view, local camera `cam`) whose target segment lies inside ONE same-sign sector of the
    source endpoints' epipolar lines -- the pairs a sector test without the e_d condition would drop (float64 geometry)."""
    o0, n = mv["offsets"][cam]
    s1 = mv["src_segs"].astype(np.float64)
    s2 = mv["tgt_segs"][o0:o0 + n].astype(np.float64)
    F = mv["F"][cam].astype(np.float64)
    one = lambda a: np.concatenate([a, np.ones((len(a), 1))], 1)
    e1, e2 = one(s1[:, :2]) @ F.T, one(s1[:, 2:]) @ F.T
    q1, q2 = one(s2[:, :2]), one(s2[:, 2:])
    a = np.stack([e1 @ q1.T, e1 @ q2.T, e2 @ q1.T, e2 @ q2.T])
    return int((accepted & ((a > 0).all(0) | (a < 0).all(0))).sum())


def test_pair_pretest_wrapping_epipolar_transfer(gpu_ctx, oracle_lib):
    """Cameras that face each other / move forward: the epipolar transfer of a source segment can wrap through infinity, the
    reference then keeps pairs whose target segment lies in a same-sign sector of the two epipolar lines (all four depths
    positive although one 3-D endpoint is behind the other camera, cudawrapper.cu:931 does not look).  The stage-1 filters
    must keep them too: every filter mask gives the oracle's kept list, bit for bit, and the scene provably contains such pairs."""
    from line3d_amd.synth import make_scene_from_poses
    O = (0.0, 0.0, 0.0)
    centers = [(0, 0, -4), (0.3, 0.1, 4), (-0.4, 0.2, 4.2), (0.1, 0.05, -3.0), (0.0, -0.1, -5.0), (4, 0.2, 0.3), (0.5, 0.3, -4.1)]
    sc = make_scene_from_poses(centers, [O] * len(centers), 500, seed=91)
    o = op.OracleLine3D(matching_neighbors=len(centers) - 1, use_collinearity=False)
    for v in sc.views:
        o.add_image_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
    o.matched, o.potential = {}, {}
    o.find_visual_neighbors()
    o.transform_geometry()
    wrapped = 0
    for src in (0, 1):
        for n in o.visual_neighbors[src]:
            o._fundamental(src, n)
        mv = o.marshal_view(src)
        for cam in range(len(mv["l2g"])):
            o0, n = mv["offsets"][cam]
            buf = op.pairwise_dense(oracle_lib, mv["src_segs"], mv["RtKinv_src"], mv["C_src"], mv["tgt_segs"], int(o0), int(n), cam,
                                    mv["F"], mv["RtKinv"], mv["centers"])
            wrapped += _sector_rejected(mv, cam, (buf > 0).all(axis=2))
        tr = dict(marshal=mv, in_matches=np.zeros(0, op.MATCH_DTYPE))
        exp, exp_med = op.compute_pairwise_matches(oracle_lib, mv["src_segs"], mv["RtKinv_src"], mv["C_src"], mv["tgt_segs"], mv["offsets"],
                                                   mv["F"], mv["RtKinv"], mv["centers"], mv["P"], mv["tbm"], tr["in_matches"], mv["l2g"],
                                                   mv["k_upper"], mv["k_lower"], 3.5, 10.0, mv["spatial_k"])[:2]
        for mask in (0, 1, 2, 3):
            gpu_ctx.set_pair_pretest(mask)
            m, med, _ = _run_view(gpu_ctx, tr)
            assert m.tobytes() == exp.tobytes(), "view %d, filter mask %d: differs from the oracle" % (src, mask)
            assert np.float32(med) == np.float32(exp_med)
        gpu_ctx.set_pair_pretest(3)
    assert wrapped > 20, "the scene does not exercise the wrapping case (%d pairs)" % wrapped


def test_window_verify_global_scratch_variant(gpu_ctx, small_oracle, small_scene):
    """With the LDS budget capped most segments take the global-scratch variant of the window kernel; results stay
    bit-identical (per-view seam call and the resident chain)."""
    gpu_ctx.set_verify_lds_budget(8 * 1024 + 256 * 6 * 4 + 4096)       # ~512 candidates fit
    try:
        for v in (1, 4, 7):
            tr = small_oracle.trace[v]
            got, med, _ = _run_view(gpu_ctx, tr)
            assert got.tobytes() == tr["matches"].tobytes(), "view %d" % v
        from line3d_amd.pipeline import Line3D, load_scene
        l = Line3D("", matchingNeighbors=6)
        l.keep_view_matches(True)
        load_scene(l, small_scene)
        l.compute3Dmodel(False)
        for v in sorted(small_oracle.trace):
            assert l.view_matches(v)[0].tobytes() == small_oracle.trace[v]["matches"].tobytes(), "chain view %d" % v
        l.close()
    finally:
        gpu_ctx.set_verify_lds_budget(0)


def test_sq_threshold_closed_form_equals_walk(gpu_ctx):
    """T(u) = largest float x with sqrtf(x) <= u: the closed form used by the verification kernels against the ulp walk,
    and both against the definition evaluated with numpy's correctly rounded float32 sqrt."""
    rng = np.random.default_rng(7)
    u = np.concatenate([np.exp(rng.uniform(-40, 40, 200000)), rng.uniform(0, 1, 50000), 2.0 ** rng.integers(-60, 60, 2000),
                        np.array([0.0, 1e-45, 1e-38, 1.17549435e-38, 1.0, 3.0, 1.8e19])]).astype(np.float32)
    walk, closed = gpu_ctx.test_sq_threshold(u)
    assert walk.tobytes() == closed.tobytes()
    sel = np.isfinite(closed) & (closed < 1e38) & (closed > 1e-36)
    T = closed[sel]
    assert np.all(np.sqrt(T) <= u[sel])
    assert np.all(np.sqrt(np.nextafter(T, np.float32(np.inf))) > u[sel])


def _literal_affinity(oracle_lib, seg_base, hyp, score, hyp_dense, best, pot, coll, sigma_a):
    """clusterSegments2D's three loops with the `used` set, literally (line3D.cc:996-1214), over dense segment ids."""
    import ctypes as C
    half = np.float32(0.5)

    def sim(a, b):
        def parts(h):
            return (np.concatenate([h["P1"], h["P2"], h["dir"]]).astype(np.float64), np.array([h["depth_p1"], h["depth_p2"]], np.float32),
                    np.array([h["k_lower"], h["k_upper"], h["median_depth"]], np.float32))
        s1, d1, c1 = parts(hyp[a])
        s2, d2, c2 = parts(hyp[b])
        return np.float32(oracle_lib.l3do_similarity_coll3D(s1.ctypes.data_as(C.POINTER(C.c_double)), d1.ctypes.data_as(C.POINTER(C.c_float)),
                                                            c1.ctypes.data_as(C.POINTER(C.c_float)), s2.ctypes.data_as(C.POINTER(C.c_double)),
                                                            d2.ctypes.data_as(C.POINTER(C.c_float)), c2.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(sigma_a)))

    used, node_of, node_hyp, A, n_cand = set(), {}, [], [], 0

    def node(h):
        if h not in node_of:
            node_of[h] = len(node_hyp)
            node_hyp.append(h)
        return node_of[h]

    def edge(a, b, w):
        na = node(a)
        nb = node(b)
        A.append((na, nb, w))
        A.append((nb, na, w))

    for si in range(len(hyp)):
        src = int(hyp_dense[si])
        for t in pot[src]:
            if (src, t) in used:
                continue
            used.add((src, t)); used.add((t, src))
            if best[t] >= 0:
                n_cand += 1
                w = np.float32(half * np.float32(score[si] + score[best[t]])) * sim(si, best[t])
                if w > np.float32(0.25):
                    edge(si, int(best[t]), w)
                for c, _ in coll[t]:
                    if (src, c) in used:
                        continue
                    used.add((src, c)); used.add((c, src))
                    if best[c] >= 0:
                        n_cand += 1
                        w = np.float32(half * np.float32(score[si] + score[best[c]])) * sim(si, best[c])
                        if w > np.float32(0.01):
                            edge(si, int(best[c]), w)
        for x, cw in coll[src]:
            if (src, x) in used:
                continue
            used.add((src, x)); used.add((x, src))
            if best[x] >= 0:
                n_cand += 1
                w = np.float32(np.float32(np.float32(cw) * half) * np.float32(score[si] + score[best[x]])) * sim(si, best[x])
                if w > np.float32(0.01):
                    edge(si, int(best[x]), w)
    return A, node_hyp, n_cand


@pytest.mark.parametrize("seed,one_way,chunk,blocks", [(1, 0.0, None, None), (2, 0.15, None, (300, 40)), (3, 0.15, "3", None), (4, 0.5, "1", (57, 1000000)), (5, 0.0, "2", (1000000, 7)),
                                                       (6, 0.1, None, (1, 1)),
                                                       # seeds 7-9: SYMMETRIC collinearity lists (what the reference always has, segments.h:94-95): the short-list path of the fill
                                                       (7, 0.0, None, None), (8, 0.15, "3", None), (9, 0.1, None, (1, 1))])
def test_affinity_fill_tables_against_the_literal_used_rule(gpu_ctx, oracle_lib, monkeypatch, seed, one_way, chunk, blocks):
    """l3d_affinity_fill on random flat tables -- clustered hypotheses so that many similarities pass, long target groups,
    targets without a hypothesis, collinearity lists that are NOT symmetric, potential correspondences recorded one way only
    (`one_way`: the schedule then waits for earlier views) -- against the reference's loops with a literal `used` set.
    Small passes (`chunk` targets) make groups and flattened entries straddle passes.  `blocks` = (candidate pairs, decision words) per block of
    sources: the fill enumerates a block of sources at a time (round 5: no count of the whole fill is bound to 31 bits) -- tiny budgets give one
    source per block, and the node numbering (rank of the 64-bit first-touch positions) and the edge list must not notice."""
    from line3d_amd.capi import HYP_DTYPE
    gpu_ctx.set_option("L3D_AFF_CHUNK", int(chunk) if chunk else 0)       # (a switch of the context; the environment is read once, at its creation)
    gpu_ctx.set_option("L3D_AFF_BLOCK", blocks[0] if blocks else 0)
    gpu_ctx.set_option("L3D_AFF_WORD_BLOCK", blocks[1] if blocks else 0)
    rng = np.random.default_rng(seed)
    V, S = 7, 40
    dense = seed in (6, 9)               # groups of more than 64 targets and collinearity lists of more than 64 entries (64-lane passes)
    p_sym = 1.0 if seed >= 7 else 0.8
    if dense:
        V, S = 4, 260
    seg_base = np.arange(V + 1, dtype=np.int32) * S
    nd = V * S
    has_hyp = rng.random(nd) < 0.8
    best = np.full(nd, -1, np.int32)
    best[has_hyp] = np.arange(int(has_hyp.sum()), dtype=np.int32)
    hyp_dense = np.nonzero(has_hyp)[0].astype(np.int32)
    nh = len(hyp_dense)
    view_hyp_begin = np.searchsorted(hyp_dense, seg_base).astype(np.int32)
    # hypotheses: a few 3-D lines, every hypothesis a noisy piece of one of them
    lines_p = rng.normal(size=(12, 3)); lines_d = rng.normal(size=(12, 3)); lines_d /= np.linalg.norm(lines_d, axis=1, keepdims=True)
    which = rng.integers(0, 12, nh)
    hyp = np.zeros(nh, HYP_DTYPE)
    for i in range(nh):
        a, b = np.sort(rng.uniform(-1, 1, 2))
        P1 = lines_p[which[i]] + a * lines_d[which[i]] + rng.normal(scale=0.002, size=3)
        P2 = lines_p[which[i]] + (b + 0.1) * lines_d[which[i]] + rng.normal(scale=0.002, size=3)
        dd = P2 - P1
        hyp[i]["P1"], hyp[i]["P2"], hyp[i]["dir"] = P1, P2, dd / np.linalg.norm(dd)
        hyp[i]["depth_p1"], hyp[i]["depth_p2"] = rng.uniform(2, 6, 2)
        hyp[i]["k_lower"], hyp[i]["k_upper"], hyp[i]["median_depth"] = 0.002, 0.01, 4.0
    score = rng.uniform(0.3, 1.0, nh).astype(np.float32)
    # potential correspondences: dense, both directions unless dropped one way
    pot = [set() for _ in range(nd)]
    for d in range(nd):
        for _ in range(rng.integers(0, 9)):
            tv = int(rng.integers(0, V))
            if tv == d // S:
                continue
            run = int(rng.integers(1, 6)) if not dense or rng.random() < 0.9 else int(rng.integers(70, 200))
            base = int(rng.integers(0, S - run))
            for t in range(base, base + run):                           # runs: long groups in one view
                pot[d].add(tv * S + t)
                if rng.random() >= one_way:
                    pot[tv * S + t].add(d)
    pot = [sorted(p) for p in pot]
    coll = [dict() for _ in range(nd)]
    for d in range(nd):
        v = d // S
        for _ in range(rng.integers(0, 5) if not dense or rng.random() < 0.9 else rng.integers(70, 150)):
            x = v * S + int(rng.integers(0, S))
            if x == d:
                continue
            w = np.float32(rng.uniform(0.05, 1.0))
            coll[d][x] = w
            if rng.random() < p_sym:                                     # (seeds 1-6: mostly, not always, symmetric: the general path)
                coll[x][d] = w
    coll = [sorted(c.items()) for c in coll]
    pot_start = np.zeros(nd + 1, np.int64); pot_start[1:] = np.cumsum([len(p) for p in pot])
    coll_start = np.zeros(nd + 1, np.int64); coll_start[1:] = np.cumsum([len(c) for c in coll])
    pot_tgt = np.array([t for p in pot for t in p], np.int32)
    coll_other = np.array([x for c in coll for x, _ in c], np.int32)
    coll_w = np.array([w for c in coll for _, w in c], np.float32)
    try:
        A, node_hyp, n_cand = gpu_ctx.affinity_fill(seg_base, view_hyp_begin, hyp, score, hyp_dense, best, pot_start, pot_tgt, coll_start, coll_other, coll_w, 10.0)
    finally:
        gpu_ctx.set_option("L3D_AFF_CHUNK", 0)
        gpu_ctx.set_option("L3D_AFF_BLOCK", 0)
        gpu_ctx.set_option("L3D_AFF_WORD_BLOCK", 0)
    eA, e_nodes, e_cand = _literal_affinity(oracle_lib, seg_base, hyp, score, hyp_dense, best, pot, coll, 10.0)
    assert n_cand == e_cand and len(eA) > 200
    assert node_hyp.tolist() == e_nodes
    assert A["i"].tolist() == [e[0] for e in eA] and A["j"].tolist() == [e[1] for e in eA]
    assert np.array_equal(A["w"].view(np.uint32), np.array([e[2] for e in eA], np.float32).view(np.uint32))
    if seed == 1:
        # tables that break the contract are refused with a message instead of being walked
        from line3d_amd.capi import L3DError
        args = [seg_base, view_hyp_begin, hyp, score, hyp_dense, best, pot_start, pot_tgt, coll_start, coll_other, coll_w, 10.0]

        def broken(idx, mutate):
            b = list(args)
            b[idx] = np.array(b[idx], copy=True)
            mutate(b[idx])
            with pytest.raises(L3DError, match="affinity fill"):
                gpu_ctx.affinity_fill(*b)
        broken(7, lambda x: x.__setitem__(0, nd))                          # target out of range
        broken(7, lambda x: x.__setitem__(slice(0, 2), x[1::-1].copy()) if pot_start[1] >= 2 else x.__setitem__(0, -1))   # out of order
        broken(9, lambda x: x.__setitem__(0, (int(x[0]) + S) % nd))        # collinear segment in another view
        broken(5, lambda x: x.__setitem__(int(hyp_dense[0]), 1))           # best[] disagrees with hyp_dense[]
        broken(6, lambda x: x.__setitem__(3, x[4] + 1))                    # CSR starts that do not ascend


@pytest.mark.parametrize("diffusion", [False, True])
def test_clustering_edges_order_and_diffusion_match_oracle(gpu_ctx, oracle_lib, diffusion):
    """l3d_clustering_edges: the list performClustering walks.  A random symmetric list with MANY tied weights (the stable
    order decides the segmentation), signed zeros included: without diffusion the stable ascending order of the input; with
    diffusion the oracle's replicator dynamics + the reference's map symmetrisation (line3D.cc:1275-1301), then the order."""
    rng = np.random.default_rng(11)
    n = 300
    pairs = set()
    while len(pairs) < 2500:
        a, b = (int(x) for x in rng.integers(0, n, 2))
        if a != b:
            pairs.add((min(a, b), max(a, b)))
    levels = np.array([0.0, -0.0, 0.011, 0.25, 0.2500001, 0.5, 0.75, 1.0], np.float32)
    A = []
    for a, b in sorted(pairs, key=lambda p: (p[0] * 7919 + p[1] * 104729) % 1000003):
        w = levels[rng.integers(0, len(levels))] if rng.random() < 0.6 else np.float32(rng.random())
        A.append((a, b, w)); A.append((b, a, w))
    A = np.array(A, dtype=op.EDGE_DTYPE)
    got = gpu_ctx.clustering_edges(A, n, perform_diffusion=diffusion)
    if diffusion:
        W = op.rdd(oracle_lib, A, n, 10)
        entries = {}
        for e in W:
            s1, s2, w12 = int(e["i"]), int(e["j"]), np.float32(e["w"])
            w21 = entries.get(s2, {}).get(s1, w12)
            w = min(w12, w21)
            entries.setdefault(s1, {})[s2] = w
            entries.setdefault(s2, {})[s1] = w
        exp = np.array([(a, b, entries[a][b]) for a in sorted(entries) for b in sorted(entries[a])], dtype=op.EDGE_DTYPE)
    else:
        exp = A
    exp = exp[np.argsort(exp["w"], kind="stable")]                      # (-0.0 == 0.0 for the comparison, as for CLEdge::operator<)
    assert got.tobytes() == exp.tobytes()
    assert len(np.unique(exp["w"])) <= len(exp) // 2                    # (every weight at least twice: the stable order matters)
    # a list without the transposed entries is refused by the device path, not guessed
    if diffusion:
        from line3d_amd.capi import L3DError
        with pytest.raises(L3DError):
            gpu_ctx.clustering_edges(A[::2], n, perform_diffusion=True)


def test_fit_clusters_against_the_oracle_line_fit(gpu_ctx):
    """l3d_fit_clusters (the line fit of processClusteredSegments on the device) against the oracle's align(): clusters of
    noisy pieces of 3-D lines seen from 4-9 cameras, a cluster of 300 members, one seen by two cameras only (no segment), one
    whose members are all the same segment, and a non-trivial inverse transformation.  End points within 1e-9 (the oracle takes
    the direction from numpy's SVD, the product from its own Jacobi iteration), segment counts and sweep structure identical."""
    from collections import OrderedDict
    from line3d_amd.capi import HYP_DTYPE
    rng = np.random.default_rng(17)
    th = 0.4
    Rinv = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]]) @ np.array([[1.0, 0, 0], [0, 0.8, -0.6], [0, 0.6, 0.8]])
    scale_inv, tneg = 2.5, np.array([0.3, -1.2, 0.7])
    hyps, cams, groups = [], [], []

    def piece(p0, d, a, b, cam, noise=0.002):
        hyps.append((p0 + a * d + rng.normal(scale=noise, size=3), p0 + b * d + rng.normal(scale=noise, size=3)))
        cams.append(cam)
        return len(hyps) - 1

    for g in range(40):
        p0, d = rng.normal(size=3), rng.normal(size=3)
        d /= np.linalg.norm(d)
        ncam = int(rng.integers(4, 10)) if g != 5 else 2
        m = int(rng.integers(4, 20)) if g != 7 else 300
        members = []
        for _ in range(m):
            a = rng.uniform(-1, 1)
            members.append(piece(p0, d, a, a + rng.uniform(0.2, 1.0), int(rng.integers(0, ncam)) + 10 * g))
        if g == 9:                                                       # every member the very same segment
            for k in members:
                hyps[k] = hyps[members[0]]
        groups.append(members)
    # hypotheses are numbered by (camera, segment): sort all of them by camera and renumber
    order = sorted(range(len(hyps)), key=lambda k: (cams[k], k))
    new_of = {old: new for new, old in enumerate(order)}
    hyp = np.zeros(len(hyps), HYP_DTYPE)
    hyp_cam = np.zeros(len(hyps), np.uint32)
    for new, old in enumerate(order):
        hyp[new]["P1"], hyp[new]["P2"] = hyps[old]
        hyp_cam[new] = cams[old]
    group_start, member_hyp = [0], []
    for members in groups:
        member_hyp += sorted(new_of[k] for k in members)
        group_start.append(len(member_hyp))
    got = gpu_ctx.fit_clusters(group_start, member_hyp, hyp, hyp_cam, Rinv, scale_inv, tneg)
    o = op.OracleLine3D(matching_neighbors=4)
    o.transf_Rinv, o.transf_scale_inv, o.transf_tneg = Rinv, scale_inv, tneg
    n_lines = 0
    for g, members in enumerate(groups):
        t3 = OrderedDict()
        for k in sorted(new_of[q] for q in members):
            t3[(int(hyp_cam[k]), k)] = (o.inverse_transform(hyp[k]["P1"]), o.inverse_transform(hyp[k]["P2"]))
        exp = o.align(t3)
        assert len(got[g]) == len(exp), g
        n_lines += len(exp)
        for (gs, ge), (es, ee) in zip(got[g], exp):
            assert np.allclose(gs, es, rtol=0, atol=1e-9) and np.allclose(ge, ee, rtol=0, atol=1e-9), g
    assert len(got[5]) == 0 and n_lines > 30


@pytest.mark.parametrize("diffusion", [False, True])
def test_clustering_edges_grouped_by_component(gpu_ctx, oracle_lib, diffusion):
    """l3d_clustering_edges_grouped: every group is one connected component of the list (scipy's labels), its edges in the stable
    ascending weight order of l3d_clustering_edges, the groups ascend by their smallest node id -- and the Felzenszwalb-Huttenlocher
    walk over the groups one after the other gives the partition of the oracle's walk over the whole sorted list."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    rng = np.random.default_rng(23)
    n = 600
    comp_of = rng.integers(0, 40, n)                                       # 40 planted components (some may split)
    pairs = set()
    while len(pairs) < 3000:
        a = int(rng.integers(0, n))
        same = np.nonzero(comp_of == comp_of[a])[0]
        b = int(rng.choice(same))
        if a != b:
            pairs.add((min(a, b), max(a, b)))
    levels = np.array([0.011, 0.25, 0.2500001, 0.5, 0.75, 1.0], np.float32)
    A = []
    for a, b in sorted(pairs, key=lambda p: (p[0] * 7919 + p[1] * 104729) % 1000003):
        w = levels[rng.integers(0, len(levels))] if rng.random() < 0.5 else np.float32(rng.random())
        A.append((a, b, w)); A.append((b, a, w))
    A = np.array(A, dtype=op.EDGE_DTYPE)
    flat = gpu_ctx.clustering_edges(A, n, perform_diffusion=diffusion)
    grouped, start = gpu_ctx.clustering_edges_grouped(A, n, perform_diffusion=diffusion)
    assert start[0] == 0 and start[-1] == len(A) and np.all(np.diff(start) > 0)
    nc, lab = connected_components(coo_matrix((np.ones(len(A)), (A["i"], A["j"])), shape=(n, n)).tocsr(), directed=False)
    touched = np.unique(A["i"])
    assert len(start) - 1 == len(np.unique(lab[touched]))
    firsts = []
    for g in range(len(start) - 1):
        e = grouped[start[g]:start[g + 1]]
        assert len(np.unique(lab[e["i"]])) == 1 and np.array_equal(lab[e["i"]], lab[e["j"]])
        members = np.nonzero(lab == lab[e["i"][0]])[0]
        firsts.append(int(members.min()))
        sel = flat[np.isin(flat["i"], members)]                             # the flat stable order, restricted to the component
        assert e.tobytes() == sel.tobytes(), g
    assert firsts == sorted(firsts)
    labels_flat = op.clustering(oracle_lib, flat, n, 1.0)
    labels_grp = op.clustering(oracle_lib, grouped, n, 1.0)
    # same partition (the oracle's walk over the concatenated groups is the per-group walk)
    def canon(l):
        first = {}
        return [first.setdefault(int(x), k) for k, x in enumerate(l)]
    assert canon(labels_flat) == canon(labels_grp)


def test_clustering_edges_grouped_long_chains(gpu_ctx):
    """Connected components on graphs that are hard for label propagation: one path of 60 000 nodes in scrambled numbering, 500
    paths of 100 nodes, isolated pairs -- every group one component, the order inside the groups the stable weight order."""
    rng = np.random.default_rng(31)
    n = 60000 + 500 * 100 + 2000
    perm = rng.permutation(n)
    edges = []
    chain = perm[:60000]
    edges += [(int(chain[k]), int(chain[k + 1])) for k in range(len(chain) - 1)]
    for p in range(500):
        c = perm[60000 + p * 100: 60000 + (p + 1) * 100]
        edges += [(int(c[k]), int(c[k + 1])) for k in range(99)]
    iso = perm[60000 + 50000:]
    edges += [(int(iso[2 * k]), int(iso[2 * k + 1])) for k in range(1000)]
    A = []
    for a, b in edges:
        w = np.float32(rng.choice([0.1, 0.5, 0.9])) if rng.random() < 0.7 else np.float32(rng.random())
        A.append((a, b, w)); A.append((b, a, w))
    A = np.array(A, dtype=op.EDGE_DTYPE)
    grouped, start = gpu_ctx.clustering_edges_grouped(A, n)
    assert len(start) - 1 == 1 + 500 + 1000 and start[-1] == len(A)
    sizes = np.diff(start)
    assert sorted(sizes.tolist())[-1] == 2 * 59999 and sorted(sizes.tolist())[0] == 2
    flat = gpu_ctx.clustering_edges(A, n)
    for g in (0, 1, len(start) - 2, int(np.argmax(sizes))):
        e = grouped[start[g]:start[g + 1]]
        members = np.unique(np.concatenate([e["i"], e["j"]]))
        sel = flat[np.isin(flat["i"], members)]
        assert e.tobytes() == sel.tobytes()


def _planted_graph(rng, n, n_comp, n_pairs):
    comp_of = rng.integers(0, n_comp, n)
    pairs = set()
    while len(pairs) < n_pairs:
        a = int(rng.integers(0, n))
        b = int(rng.choice(np.nonzero(comp_of == comp_of[a])[0]))
        if a != b:
            pairs.add((min(a, b), max(a, b)))
    levels = np.array([0.011, 0.25, 0.2500001, 0.5, 0.75, 1.0], np.float32)
    A = []
    for a, b in sorted(pairs, key=lambda p: (p[0] * 7919 + p[1] * 104729) % 1000003):
        w = levels[rng.integers(0, len(levels))] if rng.random() < 0.5 else np.float32(rng.random())
        A.append((a, b, w)); A.append((b, a, w))
    return np.array(A, dtype=op.EDGE_DTYPE)


@pytest.mark.parametrize("diffusion", [False, True])
@pytest.mark.parametrize("c", [1.0, 0.3])
def test_clustering_merge_loop_on_device(gpu_ctx, oracle_lib, diffusion, c):
    """l3d_perform_clustering_device (the merge loop of clustering.cc:21-40, one wave per connected component, state in LDS under
    local node numbers): labels BIT-EQUAL -- the same root ids -- to the oracle's sequential walk over the whole sorted list; nodes
    without an edge are their own cluster."""
    rng = np.random.default_rng(41)
    n = 900                                                               # (nodes 800..899 never appear in an edge)
    A = _planted_graph(rng, 800, 40, 4000)
    flat = gpu_ctx.clustering_edges(A, n, perform_diffusion=diffusion)
    want = op.clustering(oracle_lib, flat, n, c)
    got, n_comp = gpu_ctx.perform_clustering_device(A, n, c=c, perform_diffusion=diffusion)
    assert np.array_equal(got, want)
    assert np.array_equal(got[800:], np.arange(800, 900))
    assert n_comp == len(gpu_ctx.clustering_edges_grouped(A, n, perform_diffusion=diffusion)[1]) - 1
    assert len(np.unique(got)) > n_comp                                    # (the thresholds did split components)


def test_clustering_merge_loop_reference_vectors(gpu_ctx):
    """... and equal to the labels the reference's own clustering.cc produced (tests/golden/clustering_ref.npz, made from
    oracle/_ref/libclustering_ref.so)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "clustering_ref.npz"))
    cases = sorted({k.rsplit("_", 1)[0] for k in g.files})
    assert len(cases) == 8
    for case in cases:
        A = np.zeros(len(g[case + "_i"]), dtype=op.EDGE_DTYPE)
        A["i"], A["j"], A["w"] = g[case + "_i"], g[case + "_j"], g[case + "_w"]
        got, _ = gpu_ctx.perform_clustering_device(A, int(g[case + "_n"]), c=float(case.split("_")[1]))
        assert np.array_equal(got, g[case + "_labels"]), case


def test_clustering_merge_loop_big_components(gpu_ctx, oracle_lib):
    """Components beyond the LDS state (2048 nodes) take the same walk with their state in HBM: one path of 60 000 nodes in scrambled
    numbering, one dense blob of 3000 nodes, 500 paths of 100 nodes, 1000 pairs -- labels equal to the oracle's."""
    rng = np.random.default_rng(43)
    n = 60000 + 3000 + 500 * 100 + 2000 + 17
    perm = rng.permutation(n - 17)
    edges = []
    chain = perm[:60000]
    edges += [(int(chain[k]), int(chain[k + 1])) for k in range(len(chain) - 1)]
    blob = perm[60000:63000]
    for _ in range(20000):
        a, b = rng.choice(blob, 2, replace=False)
        edges.append((int(a), int(b)))
    edges += [(int(blob[k]), int(blob[k + 1])) for k in range(len(blob) - 1)]
    for p in range(500):
        cc = perm[63000 + p * 100: 63000 + (p + 1) * 100]
        edges += [(int(cc[k]), int(cc[k + 1])) for k in range(99)]
    iso = perm[63000 + 50000:]
    edges += [(int(iso[2 * k]), int(iso[2 * k + 1])) for k in range(1000)]
    edges = sorted({(min(a, b), max(a, b)) for a, b in edges}, key=lambda p: (p[0] * 7919 + p[1] * 104729) % 1000003)
    A = []
    for a, b in edges:
        w = np.float32(rng.choice([0.1, 0.5, 0.9])) if rng.random() < 0.7 else np.float32(rng.random())
        A.append((a, b, w)); A.append((b, a, w))
    A = np.array(A, dtype=op.EDGE_DTYPE)
    flat = gpu_ctx.clustering_edges(A, n)
    want = op.clustering(oracle_lib, flat, n, 1.0)
    got, n_comp = gpu_ctx.perform_clustering_device(A, n, c=1.0)
    assert n_comp == 1 + 1 + 500 + 1000
    assert np.array_equal(got, want)
    assert np.array_equal(got[n - 17:], np.arange(n - 17, n))


def test_clustering_merge_loop_without_components(gpu_ctx, oracle_lib, monkeypatch):
    """When the component labels do not converge (forced here: one hooking round on long paths) the whole list is walked as ONE
    component -- the plain sequential order, state in HBM: same labels."""
    rng = np.random.default_rng(47)
    n = 5000
    perm = rng.permutation(n)
    edges = [(int(perm[k]), int(perm[k + 1])) for k in range(n - 1) if k % 500 != 499]      # ten paths of 500 nodes
    A = []
    for a, b in edges:
        w = np.float32(rng.choice([0.1, 0.5, 0.9])) if rng.random() < 0.7 else np.float32(rng.random())
        A.append((a, b, w)); A.append((b, a, w))
    A = np.array(A, dtype=op.EDGE_DTYPE)
    want = op.clustering(oracle_lib, gpu_ctx.clustering_edges(A, n), n, 1.0)
    got, n_comp = gpu_ctx.perform_clustering_device(A, n, c=1.0)
    assert n_comp == 10 and np.array_equal(got, want)
    gpu_ctx.set_option("L3D_CC_MAX_ROUNDS", 1)
    try:
        got1, n_comp1 = gpu_ctx.perform_clustering_device(A, n, c=1.0)
    finally:
        gpu_ctx.set_option("L3D_CC_MAX_ROUNDS", 0)
    assert n_comp1 == 1 and np.array_equal(got1, want)


def test_fit_labelled_clusters_groups_like_process_clustered_segments(gpu_ctx):
    """l3d_fit_labelled_clusters: from (label, hypothesis) per node to the fitted clusters -- ascending label order, members in key
    order, only clusters with >= 4 members seen from >= 4 cameras (line3D.cc:1306-1340); the fits bit-equal to l3d_fit_clusters on the
    same tables."""
    from line3d_amd.capi import HYP_DTYPE
    rng = np.random.default_rng(19)
    n_hyp = 6000
    hyp = np.zeros(n_hyp, HYP_DTYPE)
    hyp_cam = np.sort(rng.integers(0, 40, n_hyp)).astype(np.uint32)            # hypotheses are numbered by (camera, segment)
    Rinv, scale_inv, tneg = np.eye(3), 1.0, np.zeros(3)
    nodes = rng.permutation(n_hyp)[:4000]                                      # node -> hypothesis (first-touch numbering: any order)
    n = len(nodes)
    labels = np.arange(n, dtype=np.int32)                                      # everybody alone ...
    free = list(rng.permutation(n))
    expected = {}
    for g in range(180):                                                       # ... except 180 planted clusters
        size = int(rng.integers(2, 30)) if g else 400
        members = [free.pop() for _ in range(size)]
        p0, d = rng.normal(size=3), rng.normal(size=3)
        d /= np.linalg.norm(d)
        for v in members:
            a = rng.uniform(-1, 1)
            hyp[nodes[v]]["P1"] = p0 + a * d + rng.normal(scale=0.002, size=3)
            hyp[nodes[v]]["P2"] = p0 + (a + rng.uniform(0.2, 1.0)) * d + rng.normal(scale=0.002, size=3)
        root = int(members[int(rng.integers(0, size))])                        # (a root is one of the cluster's nodes)
        labels[members] = root
        hs = sorted(int(nodes[v]) for v in members)
        if size >= 4 and len({int(hyp_cam[k]) for k in hs}) >= 4:
            expected[root] = hs
    gs, mh, fits = gpu_ctx.fit_labelled_clusters(labels, nodes, hyp, hyp_cam, Rinv, scale_inv, tneg)
    want_gs, want_mh = [0], []
    for root in sorted(expected):
        want_mh += expected[root]
        want_gs.append(len(want_mh))
    assert len(expected) > 100 and any(len(v) == 400 for v in expected.values())
    assert gs.tolist() == want_gs and mh.tolist() == want_mh
    ref = gpu_ctx.fit_clusters(want_gs, want_mh, hyp, hyp_cam, Rinv, scale_inv, tneg)
    assert len(ref) == len(fits) and sum(len(f) for f in fits) > 100
    for a, b in zip(fits, ref):
        assert len(a) == len(b)
        for (s1, e1), (s2, e2) in zip(a, b):
            assert s1.tobytes() == s2.tobytes() and e1.tobytes() == e2.tobytes()


def test_diffusion_equals_the_reference_kernels(gpu_ctx):
    """l3d_replicator_dynamics_diffusion against tests/golden/rdd_ref.npz: results of the REFERENCE's own K_sparseMat_row_normalization /
    K_sparseMat_diffusion_step (compiled from cudawrapper.cu:717-829, tests/golden/make_golden_rdd.py) -- bit for bit, 1 and 10 iterations,
    symmetric and asymmetric values, ties, values under L3D_EPS_G."""
    import rdd_cases as rc
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rdd_ref.npz"))
    for k, case in enumerate(rc.CASES):
        A = rc.make_list(**case)
        assert A.tobytes() == g["c%d_in" % k].tobytes()
        for iters in (1, 10):
            assert gpu_ctx.replicator_dynamics_diffusion(A, case["n"], iters).tobytes() == g["c%d_it%d" % (k, iters)].tobytes(), (k, iters)


def test_collinearity_equals_the_reference_kernel(gpu_ctx):
    """l3d_compute_collinearity against tests/golden/pairwise_ref.npz: the relation the REFERENCE's own K_collinearity (cudawrapper.cu:476-535,
    compiled from its text) gives for an image with planted collinear pieces -- the same non-zero pattern, weights within 1.2e-7 (one float ulp
    below 1: the product's expf is the numeric contract's, the reference build's glibc's)."""
    import devfn_cases as dc
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pairwise_ref.npz"))
    segs = dc.collinear_segments(31)
    gi, gj, gw = gpu_ctx.compute_collinearity(segs, 2.5)
    S = len(segs)
    got = np.zeros((S, S), np.float32)
    got[gi, gj] = gw
    got[gj, gi] = gw
    want = np.zeros((S, S), np.float32)
    want[g["coll_idx"][:, 0], g["coll_idx"][:, 1]] = g["coll_val"]
    assert (want > 0).sum() > 100 and np.array_equal(got > 0, want > 0) and np.max(np.abs(got - want)) <= 1.2e-7
