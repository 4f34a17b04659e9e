"""Parity at the literal sizes BASELINE.json states (the workload bench.py times and the shapes of configs[2] and configs[4]).

The oracle needs ~25 s for one whole 2000-segment view, so whole scenes are covered by properties -- every matching path
(resident chain, per-view seam calls, native sharded run, virtual ranks) gives the same bytes, a second pass is idempotent,
the conservative stage-1 filters and the depth-window search change nothing against the exact sequence / the all-pairs loop
on EVERY pair of the scene -- and the oracle pins slices of early, mid-chain, late and early-return views bit for bit
(reference: line3D.cc:620-648, cudawrapper.cu:858-1128)."""
import os

import numpy as np
import pytest

from helpers import digest_lists, oracle_view_slice

pytestmark = pytest.mark.gpu


def _run(scene, N, sync=False, pretest=None, verify_mode=None, native=False, S=2000):
    from line3d_amd.pipeline import Line3D, load_scene
    from line3d_amd import distributed as l3dist
    l = Line3D("", matchingNeighbors=N)
    l.keep_view_matches(True)
    l.set_sync_matching(sync)
    if pretest is not None:
        l.context().set_pair_pretest(pretest)
    if verify_mode is not None:
        l.context().set_verify_mode(verify_mode)
    load_scene(l, scene)
    l.prepare()
    if native:
        l3dist.match_views_chain_native(l, 0, 1, None, commit=True, n_segments=S, n_neighbors=N)
    else:
        l.match_views()
    lists = {v["id"]: l.view_matches(v["id"]) for v in scene.views}
    return l, lists


# ---------------------------------------------------------------------------------------------------------------------
# configs[1] / configs[3]: 64 views x 2000 segments x 12 neighbours, the scene bench.py times (seed 20260)
# ---------------------------------------------------------------------------------------------------------------------
V2, S2, N2 = 64, 2000, 12


@pytest.fixture(scope="module")
def cfg2_scene():
    from line3d_amd.synth import make_scene
    return make_scene(V2, S2, N2, seed=20260)


@pytest.fixture(scope="module")
def cfg2_chain(cfg2_scene):
    l, lists = _run(cfg2_scene, N2)
    yield l, lists
    l.close()


def test_config2_all_paths_agree_and_a_second_pass_is_idempotent(cfg2_scene, cfg2_chain):
    """63 chained views: the stage-1 ring (15 slots) wraps four times, the kept arena holds every view's list."""
    l, lists = cfg2_chain
    ref = digest_lists(lists)
    assert sum(len(m) for m, _ in lists.values()) > 1500000
    assert l.stats()["pairs"] == pytest.approx(1.452e9, rel=0.01)         # the workload of the bench line
    l.match_views()
    assert digest_lists({v["id"]: l.view_matches(v["id"]) for v in cfg2_scene.views}) == ref, "second pass"
    for name, kw in (("per-view seam calls", dict(sync=True)), ("native sharded run, world 1", dict(native=True))):
        l2, lists2 = _run(cfg2_scene, N2, **kw)
        l2.close()
        assert digest_lists(lists2) == ref, name


def test_config2_filters_and_window_search_change_nothing_on_the_whole_scene(cfg2_scene, cfg2_chain):
    """1.45e9 segment pairs through the conservative stage-1 filters (margins 1e-4 / 1e-2, l3d_kernels.hip) against the
    exact sequence alone, and every candidate through the depth-window search against the all-pairs loop of
    K_verify_matches (cudawrapper.cu:614-714): identical kept lists."""
    _l, lists = cfg2_chain
    ref = digest_lists(lists)
    for name, kw in (("no stage-1 filters", dict(pretest=0)), ("wedge test only", dict(pretest=1)), ("overlap bound only", dict(pretest=2)),
                     ("all-pairs verification", dict(verify_mode=1))):
        l2, lists2 = _run(cfg2_scene, N2, **kw)
        l2.close()
        assert digest_lists(lists2) == ref, name


@pytest.mark.parametrize("vid,lo,hi", [(7, 0, 256), (20, 400, 656), (33, 900, 1156), (48, 1300, 1556), (62, 1744, 2000), (63, 0, 2000)])
def test_config2_view_slices_against_the_oracle(cfg2_scene, cfg2_chain, vid, lo, hi):
    """256 source segments of an early view, a mid-chain view (6 cameras to match, 6 sources of reverse matches), the last
    computed view (1 camera to match, 11 sources) -- and the whole last view, which has nothing left to match: the
    reference returns its existing list untouched, LOCAL camera ids, confidence 0 (cudawrapper.cu:877-878)."""
    _l, lists = cfg2_chain
    exp, mv, existing = oracle_view_slice(cfg2_scene, lists, vid, lo, hi, N2, threads=1 if vid == 63 else None)   # (the early return ignores the range)
    got = lists[vid][0]
    got = got[(got["segID1"] >= lo) & (got["segID1"] < hi)]
    if vid == 63:
        assert len(mv["tbm"]) == 0 and len(existing) > 10000
    else:
        assert len(mv["tbm"]) == min(6, 63 - vid) and (vid < 12 or len(existing) > 10000)
    assert len(exp) > 500 and got.tobytes() == exp.tobytes()


def test_config2_mid_chain_slice_against_the_reference_kernels(cfg2_scene, cfg2_chain):
    """The same mid-chain slice (view 33, 128 source segments) with the REFERENCE's own K_pairwise_matches and K_verify_matches
    (oracle/_spliced/libkernels_spliced.so, compiled from cudawrapper.cu's text) inside the oracle's host code, glibc transcendentals: the product keeps
    the same correspondences with the same depths; confidences within 5e-6 (contract vs glibc expf / acosf)."""
    import ctypes as C
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_spliced", "libkernels_spliced.so")
    if not os.path.exists(path) or not hasattr(C.CDLL(path), "l3dref_pairwise_matches"):
        pytest.skip("oracle/_spliced/libkernels_spliced.so with the reference's kernels is not built")
    _l, lists = cfg2_chain
    vid, lo, hi = 33, 900, 1028
    exp, mv, existing = oracle_view_slice(cfg2_scene, lists, vid, lo, hi, N2, reference=C.CDLL(path))
    got = lists[vid][0]
    got = got[(got["segID1"] >= lo) & (got["segID1"] < hi)]
    assert len(exp) > 300 and len(got) == len(exp) and len(existing) > 10000
    for k in ("segID1", "camID2", "segID2"):
        assert np.array_equal(got[k], exp[k]), k
    assert got["depths"].tobytes() == exp["depths"].tobytes()
    assert np.max(np.abs(got["confidence"] - exp["confidence"])) <= 5e-6


def test_config3_512_views_stage1_filters_decide_nothing_against_the_exact_test():
    """1.22e10 segment pairs of configs[2]'s scene with all levels of k_pair_mask (sector test, interval bounds that reject AND accept) against the exact
    sequence alone: the same NUMBER OF CANDIDATES (kept lists can agree while candidates differ) and the same kept lists.  This is the size at which round 4
    found pairs the bounds decided against the exact test (an intersection point on an end point of a segment: D_segment_overlap_2D returns 0 or thousands
    there, whatever the intervals are; tests/golden/endpoint_quirk_pairs.npz) -- nine candidates in 7.9e8, invisible at 64 views."""
    from line3d_amd.synth import make_scene
    V, S, N = 512, 2000, 12
    scene = make_scene(V, S, N, seed=20260)
    out = {}
    for name, pretest in (("all levels", None), ("exact test alone", 0), ("no accepts", 7)):
        l, lists = _run(scene, N, pretest=pretest)
        st = l.stats()
        out[name] = (digest_lists(lists), int(st["raw"]), int(st["kept"]))
        l.close()
        if pretest is None:
            # the ORACLE on the source segments around the three pairs the first accepts let through (view, source segment): mid-chain views, the
            # kept lists of the earlier views towards them taken from this run
            for vid, src in ((440, 182), (450, 466), (509, 1896)):
                lo = (src // 64) * 64
                exp, mv, _existing = oracle_view_slice(scene, lists, vid, lo, lo + 64, N)
                got = lists[vid][0]
                got = got[(got["segID1"] >= lo) & (got["segID1"] < lo + 64)]
                assert len(exp) > 10 and got.tobytes() == exp.tobytes(), (vid, lo)
        del lists
    assert out["exact test alone"][1] > 700_000_000
    assert out["all levels"] == out["exact test alone"], out
    assert out["no accepts"] == out["exact test alone"], out


# ---------------------------------------------------------------------------------------------------------------------
# configs[2]: 512 views x 2000 x 12, sharded over 8 ranks -- on one GPU: the unsharded chain against rank 0's committed lists
# of a recorded 8-virtual-rank run replayed through the native loop
# ---------------------------------------------------------------------------------------------------------------------
def test_config3_512_views_chain_equals_eight_virtual_ranks():
    import torch
    from line3d_amd.pipeline import Line3D, load_scene
    from line3d_amd.synth import make_scene
    from line3d_amd.distributed import default_slot_records
    V, S, N, W = 512, 2000, 12, 8
    scene = make_scene(V, S, N, seed=20263)
    l, lists = _run(scene, N)
    ref = digest_lists(lists)
    kept = sum(len(m) for m, _ in lists.values())
    assert kept > 12000000
    l.match_views()
    assert digest_lists({v["id"]: l.view_matches(v["id"]) for v in scene.views}) == ref, "second pass"
    l.close()
    del lists
    slot = default_slot_records(S, N, W)
    dev = torch.device("cuda", 0)
    ls = []
    for r in range(W):
        lr = Line3D("", matchingNeighbors=N, useCollinearity=False)
        lr.keep_view_matches(r in (0, 3))
        load_scene(lr, scene)
        lr.prepare()
        ls.append(lr)
    n_views, slot_bytes = [lr.shard_open(r, W, slot) for r, lr in enumerate(ls)][0]
    assert n_views == V
    gathered = torch.zeros(n_views * W * slot_bytes, dtype=torch.uint8, device=dev)
    send = [torch.zeros(n_views * slot_bytes, dtype=torch.uint8, device=dev) for _ in range(W)]
    torch.cuda.synchronize()
    for k in range(n_views):
        for r, lr in enumerate(ls):
            lr.shard_enqueue(k, send[r].data_ptr() + k * slot_bytes, gathered.data_ptr())
        torch.cuda.synchronize()
        if ls[0].shard_view_verified(k):
            for r in range(W):
                gathered[(k * W + r) * slot_bytes:(k * W + r + 1) * slot_bytes].copy_(send[r][k * slot_bytes:(k + 1) * slot_bytes])
        torch.cuda.synchronize()
        for lr in ls:
            lr.shard_mark(k)
    for lr in ls:
        lr.shard_close(False)
    del send
    # ranks replay their part through the native loop (l3d_shard_chain_run) against the recorded blocks; rank 0 commits on the host
    # (kept lists handed over view by view), rank 3 on its device (matchViews' products built from the gathered slots)
    for r in (W - 1, 3, 0):
        ls[r].shard_run(r, W, slot, "replay", gathered.data_ptr(), commit=("device" if r == 3 else r == 0))
        torch.cuda.synchronize()
    assert digest_lists({v["id"]: ls[0].view_matches(v["id"]) for v in scene.views}) == ref
    assert digest_lists({v["id"]: ls[3].view_matches(v["id"]) for v in scene.views}) == ref
    assert ls[3].resident_products() is not None and ls[0].resident_products() is None
    for lr in ls:
        lr.close()


# ---------------------------------------------------------------------------------------------------------------------
# configs[4] per-view shape: 4000 segments x 24 neighbours.  26 views: view 13 has 12 cameras to match and 12 sources of
# reverse matches.  The scene keeps ~10^3 matches per segment (a quarter of the candidates), so lists are digested view by
# view and only what the oracle slice needs is kept.
# ---------------------------------------------------------------------------------------------------------------------
V5, S5, N5, VID5 = 26, 4000, 24, 13


def _digest_stream(l, scene, keep_towards=None):
    import hashlib
    h = hashlib.sha256()
    kept, towards, own = 0, {}, None
    for v in scene.views:
        m, med = l.view_matches(v["id"])
        h.update(m.tobytes())
        h.update(np.float32(med).tobytes())
        kept += len(m)
        if keep_towards is not None and v["id"] < keep_towards:
            towards[v["id"]] = (m[m["camID2"] == keep_towards].copy(), med)
        if keep_towards is not None and v["id"] == keep_towards:
            own = m
    return h.hexdigest(), kept, towards, own


def _run5(scene, **kw):
    from line3d_amd.pipeline import Line3D, load_scene
    from line3d_amd import distributed as l3dist
    l = Line3D("", matchingNeighbors=N5, useCollinearity=False)
    l.keep_view_matches(True)
    l.set_sync_matching(bool(kw.get("sync")))
    if kw.get("pretest") is not None:
        l.context().set_pair_pretest(kw["pretest"])
    load_scene(l, scene)
    l.prepare()
    if kw.get("native"):
        l3dist.match_views_chain_native(l, 0, 1, None, commit=True, n_segments=S5, n_neighbors=N5)
    else:
        l.match_views()
    return l


@pytest.fixture(scope="module")
def cfg5_scene():
    from line3d_amd.synth import make_scene
    return make_scene(V5, S5, N5, seed=20265)


@pytest.fixture(scope="module")
def cfg5_chain(cfg5_scene):
    l = _run5(cfg5_scene)
    out = _digest_stream(l, cfg5_scene, keep_towards=VID5)
    yield l, out
    l.close()


def test_config5_shape_paths_agree(cfg5_scene, cfg5_chain):
    l, (ref, kept, _t, _o) = cfg5_chain
    assert kept > 10000000
    l.match_views()
    assert _digest_stream(l, cfg5_scene)[0] == ref, "second pass"
    for name, kw in (("per-view seam calls", dict(sync=True)), ("native sharded run, world 1", dict(native=True)),
                     ("no stage-1 filters", dict(pretest=0))):
        l2 = _run5(cfg5_scene, **kw)
        d = _digest_stream(l2, cfg5_scene)[0]
        l2.close()
        assert d == ref, name


def test_config5_shape_mid_chain_slice_against_the_oracle(cfg5_scene, cfg5_chain):
    """128 source segments of view 13 (12 cameras to match, 12 sources): several thousand candidates per segment, the
    all-pairs loop of the oracle runs ~2e9 iterations (about half a minute on 16 host threads)."""
    _l, (_ref, _kept, towards, own) = cfg5_chain
    lo, hi = 2000, 2128
    exp, mv, existing = oracle_view_slice(cfg5_scene, towards, VID5, lo, hi, N5)
    assert len(mv["tbm"]) == 12 and len(mv["l2g"]) == 24 and len(existing) > 100000
    got = own[(own["segID1"] >= lo) & (own["segID1"] < hi)]
    assert len(exp) > 500 and got.tobytes() == exp.tobytes()


def test_config3_512_views_sharded_by_blocks_of_views_equal_the_one_chain():
    """configs[2]'s 512 views x 2000 segments x 12 neighbours with the VIEWS sharded over 8 ranks in blocks of 64 (l3d_match_chain_blocks: every rank
    the full-width single-GPU chain on its block + 36 warm-up views started cold, the speculation verified with digests of the kept lists, the blocks
    all-gathered): eight virtual ranks as threads on the one GPU, an all-gather through the host.  Every rank must report an exact speculation and hold
    the ONE chain's kept lists (rank 0 and rank 5 are compared view by view with the unsharded run) -- four collectives for the whole pass."""
    import threading
    from line3d_amd.pipeline import Line3D, load_scene
    from line3d_amd.synth import make_scene
    from helpers import thread_exchange as _thread_exchange
    V, S, N, W = 512, 2000, 12, 8
    scene = make_scene(V, S, N, seed=20260)
    ref, ref_lists = _run(scene, N)
    want = digest_lists(ref_lists)
    ref.close()
    make, calls = _thread_exchange(W)
    ls, verdicts, errors = [], [None] * W, []
    for r in range(W):
        l = Line3D("", matchingNeighbors=N)
        l.keep_view_matches(r in (0, 5))
        load_scene(l, scene)
        l.prepare()
        ls.append(l)

    def run(r):
        try:
            verdicts[r] = ls[r].block_run(r, W, make(r), None, -1)
        except Exception as e:      # noqa: BLE001
            errors.append((r, e))
    th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    assert verdicts == [True] * W, verdicts
    tags = [c[0] for c in calls]          # (a block whose speculation failed is re-run warm: a hand-over and another round of digests each)
    n_rep = ls[0].partition_info()["recovery_rounds"]
    assert tags == [-1, -3] + [-3, -5, -1, -3] * n_rep + [-3, -2, -3, -3, -4], tags
    for r in (0, 5):
        assert digest_lists({v["id"]: ls[r].view_matches(v["id"]) for v in scene.views}) == want, "rank %d" % r
    for l in ls:
        l.close()


def test_window_kernel_variants_agree_at_20_neighbours():
    """The instantiations of k_verify_window that only exist above 16 neighbours or on a rank's small launches against
    the plain one: bucket starts in global memory (k_verify_window_gb), the split verification in units (k_verify_window_build + k_vw_walk, units
    of 256 and 512 hypotheses, with the starts in LDS and in global memory), and the all-pairs loop of K_verify_matches (cudawrapper.cu:614-714) --
    every kept list and median of a 36-view chain with 20 neighbours."""
    from line3d_amd.pipeline import Line3D, load_scene
    from line3d_amd.synth import make_scene
    V, S, N = 36, 700, 20
    scene = make_scene(V, S, N, seed=314)

    def run(options=None, verify_mode=None):
        l = Line3D("", matchingNeighbors=N)
        l.keep_view_matches(True)
        load_scene(l, scene)
        l.prepare()
        for k, v in (options or {}).items():
            l.context().set_option(k, v)
        if verify_mode is not None:
            l.context().set_verify_mode(verify_mode)
        l.match_views()
        assert l.match_path() == 0
        d = digest_lists({v["id"]: l.view_matches(v["id"]) for v in scene.views})
        kept = int(l.stats()["kept"])
        l.close()
        return d, kept
    ref, kept = run({"L3D_VW_GB": 0, "L3D_VW_SPLIT": 0})
    assert kept > 300000
    for name, opts in (("bucket starts in global memory", {"L3D_VW_GB": 1, "L3D_VW_SPLIT": 0}),
                       ("split, units of 512, starts in LDS during the build", {"L3D_VW_GB": 0, "L3D_VW_SPLIT": 1}),
                       ("split, units of 256", {"L3D_VW_GB": 1, "L3D_VW_SPLIT": 1, "L3D_VW_UNIT": 256}),
                       ("defaults", {})):
        assert run(opts) == (ref, kept), name
    assert run(verify_mode=1) == (ref, kept), "all-pairs verification"


@pytest.mark.parametrize("chunk", [0, 7])
def test_affinity_fill_short_list_path_equals_the_general_path(chunk):
    """clusterSegments2D's fill (line3D.cc:968-1221) on a symmetric collinearity table (segments.h:94-95): "does an expanded earlier target list c"
    asked from c's own list and k_aff_groups resolving by collinear predecessors (L3D_AFF_SYM=1, the default) against the general path that walks
    every earlier target (L3D_AFF_SYM=0) -- the affinity list, node table and lines; also with passes of 7 targets (groups spanning passes)."""
    from helpers import assert_lines_equal
    from line3d_amd.pipeline import Line3D, load_scene
    from line3d_amd.synth import make_scene
    V, S, N = 24, 500, 8
    scene = make_scene(V, S, N, seed=99)
    out = []
    for sym in (0, 1):
        l = Line3D("", matchingNeighbors=N)
        load_scene(l, scene)
        l.prepare()
        l.context().set_option("L3D_AFF_SYM", sym)
        if chunk:
            l.context().set_option("L3D_AFF_CHUNK", chunk)
        l.match_views()
        l.finish(False)
        A, n_nodes = l.affinity()
        out.append((A.copy(), n_nodes, l.getResult()))
        l.close()
    assert len(out[0][0]) > 20000 and out[0][1] == out[1][1]
    assert out[0][0].tobytes() == out[1][0].tobytes()
    assert_lines_equal(out[1][2], out[0][2], 0.0)
