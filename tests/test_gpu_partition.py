"""matchViews sharded by blocks of views with NOTHING replicated (l3d_match_chain_partition, l3d_affinity_fill_sharded: BASELINE configs[4] as a
job that fits -- every rank holds its block's share of the kept records and of matchViews' products, the affinity fill is sharded by source key,
SURVEY.md 8e) against the ONE single-GPU chain: virtual ranks as threads of this process on the one GPU of the test box, the all-gather through
the host (helpers.thread_exchange).  Reference behaviour: line3D.cc:620-648 (matchViews), :834-884 (what performMatching leaves behind),
:968-1221 (clusterSegments2D), view.cc:150-224 (the reference's own way of not holding everything: a file per view)."""
import os
import threading

import numpy as np
import pytest

from helpers import assert_lines_equal, thread_exchange

pytestmark = pytest.mark.gpu


def _reference(scene, N, diffusion, loader=None):
    from line3d_amd.pipeline import Line3D, load_scene
    ref = Line3D("", matchingNeighbors=N)
    ref.keep_view_matches(True)
    (loader or load_scene)(ref, scene)
    ref.prepare()
    ref.match_views()
    lists = {v["id"]: ref.view_matches(v["id"]) for v in scene.views}
    prod = ref.resident_products()
    ref.finish(diffusion)
    A, n_nodes = ref.affinity()
    out = dict(lists=lists, pot_start=prod["pot_start"].copy(), pot_tgt=prod["pot_tgt"].copy(), best=prod["best"].copy(), seg_base=prod["seg_base"].copy(),
               A=A.copy(), n_nodes=n_nodes, lines=ref.getResult(), hyp=ref.resident_products()["hyp"].copy())
    ref.close()
    return out


def _run_partitioned(scene, N, W, warmup, diffusion, options=None, segments=False, arena_of=None, loader=None):
    from line3d_amd.pipeline import Line3D, load_scene
    make, calls = thread_exchange(W)
    ls, verdicts, errors = [], [None] * W, []
    for r in range(W):
        l = Line3D("", matchingNeighbors=N)
        l.keep_view_matches(True)
        (loader or load_scene)(l, scene)
        l.prepare()
        for k, v in (options or {}).items():
            l.context().set_option(k, v)
        if arena_of and r in arena_of:          # (the first guess of this rank's kept arena, in records)
            l.context().set_chain_capacities(0, arena_of[r])
        ls.append(l)
    shares = [None] * W

    def run(r):
        try:
            if segments:        # the segment-sharded run, partitioned: exact by construction, no verdict
                ls[r].shard_run(r, W, 4096, make(r), None, commit="partition")
                verdicts[r] = True
            else:
                verdicts[r] = ls[r].partition_run(r, W, make(r), None, warmup)
            if verdicts[r]:
                shares[r] = (ls[r].partition_info(), ls[r].resident_products())
                ls[r].finish_sharded(diffusion)
        except Exception as e:      # noqa: BLE001
            errors.append((r, e))
    th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    return ls, verdicts, errors, shares, calls


def _check_against(ref, scene, ls, shares):
    ids = sorted(v["id"] for v in scene.views)               # (the dense map: views in id order)
    for r, l in enumerate(ls):
        info, prod = shares[r]
        assert info["world"] == len(ls) and info["rank"] == r
        o0, o1 = info["own"]
        # the kept lists of the rank's block, byte for byte (and of every other view it holds)
        for vi in range(info["held"][0], info["held"][1]):
            m, med = l.view_matches(ids[vi])
            assert m.tobytes() == ref["lists"][ids[vi]][0].tobytes(), "rank %d view %d: kept list" % (r, ids[vi])
        # its rows of potential_correspondences_ are the one table's rows; rows it does not hold are empty
        sb = ref["seg_base"]
        ps, pt = prod["pot_start"], prod["pot_tgt"]
        for vi in range(len(ids)):
            for d in (int(sb[vi]), int(sb[vi + 1]) - 1):
                mine = pt[ps[d]:ps[d + 1]]
                if info["rows"][0] <= vi < info["rows"][1]:
                    assert np.array_equal(mine, ref["pot_tgt"][ref["pot_start"][d]:ref["pot_start"][d + 1]]), "rank %d: row of dense segment %d" % (r, d)
                else:
                    assert len(mine) == 0
        lo, hi = int(sb[info["rows"][0]]), int(sb[info["rows"][1]])
        for d in range(lo, hi, 7):
            assert np.array_equal(pt[ps[d]:ps[d + 1]], ref["pot_tgt"][ref["pot_start"][d]:ref["pot_start"][d + 1]])
        assert info["n_pot_all"] >= len(ref["pot_tgt"])           # (rows within `reach` of a block boundary are built by both neighbours)
        # best matches of the views it holds
        lo, hi = int(sb[info["held"][0]]), int(sb[info["held"][1]])
        assert prod["best"][lo:hi].tobytes() == ref["best"][lo:hi].tobytes(), "rank %d: best matches" % r
        # after the collective finish: the ONE affinity list, the one hypothesis table, the one result -- on every rank
        A, n_nodes = l.affinity()
        assert n_nodes == ref["n_nodes"] and A.tobytes() == ref["A"].tobytes(), "rank %d: affinity list" % r
        assert l.resident_products()["hyp"].tobytes() == ref["hyp"].tobytes(), "rank %d: hypothesis table" % r
        assert_lines_equal(l.getResult(), ref["lines"], 0.0)


@pytest.mark.parametrize("W,diffusion", [(3, False), (4, True), (1, False)])
def test_partitioned_run_and_sharded_fill_equal_the_one_chain(W, diffusion):
    from line3d_amd.synth import make_scene
    V, S, N = 60, 160, 6
    scene = make_scene(V, S, N, seed=11)
    ref = _reference(scene, N, diffusion)
    assert len(ref["lines"]) > 20 and len(ref["A"]) > 1000
    ls, verdicts, errors, shares, calls = _run_partitioned(scene, N, W, -1, diffusion)
    try:
        assert not errors, errors
        assert verdicts == [True] * W
        _check_against(ref, scene, ls, shares)
        # nothing big travels: no block gather (-2), no table pieces (-4); the early-return records (-6) and the fill's five small gathers (-7 .. -10)
        tags = [c[0] for c in calls]
        assert -2 not in tags and -4 not in tags
        if W > 1:
            assert tags.count(-7) == 1 and tags.count(-8) == 1 and tags.count(-9) == 1 and tags.count(-10) == 1
    finally:
        for l in ls:
            l.close()


@pytest.mark.parametrize("W,diffusion,options", [(3, True, None), (4, False, None), (3, False, dict(L3D_SLOT_CAMS_MIN=0, L3D_CHECK_POT=1)),
                                                 (4, True, dict(L3D_SLOT_CAMS_MIN=0, L3D_RETIRE_TABLES=0, L3D_CHECK_POT=1))],
                         ids=["3 ranks", "4 ranks", "3 ranks, slots with side words and run tables, retired with the records", "4 ranks, the same slots, tables rebuilt from the records"])
def test_segment_sharded_run_partitioned_equals_the_one_chain(W, diffusion, options):
    """l3d_shard_chain_partition: the source segments of every view sharded over the ranks (no speculation: l3d_shard_chain_run), every rank retiring
    only what its block of views needs; then the very same collective finish.  Slots of a dense scene carry a (local camera, target) word per record and
    a run table (forced here on a small scene: L3D_SLOT_CAMS_MIN=0); the retire kernel files both with the records, so the products transpose the lists
    without rebuilding either (round 6; L3D_RETIRE_TABLES=0: rebuilt) -- each rank's rows are checked against the plain host construction too (L3D_CHECK_POT)."""
    from line3d_amd.synth import make_scene
    V, S, N = 60, 160, 6
    scene = make_scene(V, S, N, seed=11)
    ref = _reference(scene, N, diffusion)
    ls, verdicts, errors, shares, calls = _run_partitioned(scene, N, W, -1, diffusion, options=options, segments=True)
    try:
        assert not errors, errors
        _check_against(ref, scene, ls, shares)
        tags = [c[0] for c in calls]
        assert -1 not in tags and -2 not in tags and -4 not in tags and -5 not in tags and -6 not in tags       # no digests, no blocks, no hand-over: nothing speculated
        assert tags.count(-7) == 1 and tags.count(-9) == 1
    finally:
        for l in ls:
            l.close()


@pytest.mark.parametrize("segments", [False, True], ids=["blocks of views", "segments of every view"])
def test_partitioned_jobs_on_scattered_non_mutual_neighbourhoods(segments):
    """Cameras in no order, ragged views, neighbours picked by the library from shared world points (line3D.cc:476-549): not mutual, up to the whole
    scene apart in processing order -- reach is then most of the chain and every rank keeps most views; both partitioned jobs must still be the one chain."""
    from line3d_amd.pipeline import load_scene_worldpoints
    from line3d_amd.synth import make_scene_scattered
    V, S, N, W = 36, 260, 8, 3
    scene = make_scene_scattered(V, S, seed=77)
    ref = _reference(scene, N, False, loader=load_scene_worldpoints)
    assert len(ref["lines"]) > 5 and len(ref["A"]) > 200
    ls, verdicts, errors, shares, calls = _run_partitioned(scene, N, W, -1, False, segments=segments, loader=load_scene_worldpoints)
    try:
        assert not errors, errors
        assert verdicts == [True] * W
        _check_against(ref, scene, ls, shares)
    finally:
        for l in ls:
            l.close()


# (L3D_FUZZ_SEEDS=n: n more seeds, outside the suite's time budget)
@pytest.mark.parametrize("seed", [11, 12, 13] + [7000 + i for i in range(int(os.environ.get("L3D_FUZZ_SEEDS", "0")))])
def test_partitioned_jobs_on_randomly_drawn_scenes(seed):
    """Both partitioned jobs on scenes drawn per seed -- 24-72 views, 60-200 segments per view (ragged prefixes), 6-10 neighbours, helix or scattered cameras,
    2-4 virtual ranks, with or without diffusion; the segment-sharded one also with slots that carry side words and run tables (what a dense scene's do: retired with
    the records or rebuilt by the products) -- against the ONE chain: kept lists, rows of the table, best matches, affinity list, hypotheses, lines."""
    from line3d_amd.pipeline import load_scene_worldpoints
    from line3d_amd.synth import make_scene, make_scene_scattered
    rng = np.random.default_rng(seed)
    N, W, diffusion = int(2 * rng.integers(3, 6)), int(rng.integers(2, 5)), bool(rng.integers(0, 2))      # (4 neighbours keep nothing: two witnesses' cameras are needed)
    scattered = bool(rng.integers(0, 4) == 0)
    V, S = int(rng.integers(max(24, 4 * N), 73)), int(rng.integers(60, 201))
    loader = None
    if scattered:
        scene, loader = make_scene_scattered(min(V, 40), S + 60, seed=seed), load_scene_worldpoints
    else:
        scene = make_scene(V, S, N, seed=seed, noise_px=float(rng.choice([0.3, 0.5, 1.5])))
        for v in scene.views:
            keep = int(rng.integers(S // 2, S + 1))
            v["segments"] = np.ascontiguousarray(v["segments"][:keep])
            v["gt"] = v["gt"][:keep]
    segments = bool(rng.integers(0, 3) != 0) or scattered
    options = None
    if segments and rng.integers(0, 3) != 0:
        options = dict(L3D_SLOT_CAMS_MIN=0, L3D_RETIRE_TABLES=int(rng.integers(0, 2)), L3D_RETIRE_APART=int(rng.integers(0, 2)), L3D_CHECK_POT=1)
    ref = _reference(scene, N, diffusion, loader=loader)
    ls, verdicts, errors, shares, calls = _run_partitioned(scene, N, W, -1, diffusion, options=options, segments=segments, loader=loader)
    try:
        assert not errors, (seed, errors)
        if verdicts != [True] * W:          # (blocks of views shorter than the neighbour window: refused on every rank alike -- the segment-sharded job takes such scenes)
            assert not segments and verdicts == [False] * W, (seed, verdicts)
            print("seed %d: %d views on %d ranks, blocks of views refused" % (seed, len(scene.views), W))
            return
        _check_against(ref, scene, ls, shares)
        print("seed %d: %d views x <= %d segments, N %d, %d ranks, %s, %s, options %r: %d kept, %d affinity entries, %d lines" % (
            seed, len(scene.views), max(len(v["segments"]) for v in scene.views), N, W, "scattered" if scattered else "helix", "segments sharded" if segments else "blocks of views", options,
            sum(len(m[0]) for m in ref["lists"].values()), len(ref["A"]), len(ref["lines"])))
    finally:
        for l in ls:
            l.close()


def test_segment_sharded_partitioned_run_with_a_broken_exchange_fails_on_every_rank():
    """the all-gather of one rank breaks in the middle of the run (what a lost peer is to RCCL): every rank comes back with an error, nobody waits in
    the two exchanges that follow the run (arena verdicts, status words of the products)"""
    from line3d_amd.capi import L3DError
    from line3d_amd.pipeline import Line3D, load_scene
    from line3d_amd.synth import make_scene
    V, S, N, W = 60, 160, 6, 3
    scene = make_scene(V, S, N, seed=11)
    make, calls = thread_exchange(W, timeout=60.0)
    ls, outcome = [], [None] * W
    for r in range(W):
        l = Line3D("", matchingNeighbors=N)
        load_scene(l, scene)
        l.prepare()
        ls.append(l)

    def breaking(r):
        inner = make(r)

        def exchange(user, view, send, recv, slot_bytes, w, stream):
            if r == 1 and view == 20:
                make.abort()
                return 1
            return inner(user, view, send, recv, slot_bytes, w, stream)
        return exchange

    def run(r):
        try:
            ls[r].shard_run(r, W, 4096, breaking(r), None, commit="partition")
            outcome[r] = "ok"
        except L3DError as e:
            outcome[r] = "error: %s" % e
    th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=120)
    try:
        assert not any(x.is_alive() for x in th), "a rank is still waiting"
        assert all(o and o.startswith("error") for o in outcome), outcome
        assert "exchange" in outcome[1]
    finally:
        for l in ls:
            l.close()


def test_segment_sharded_partitioned_run_grows_one_ranks_arena_on_every_rank():
    """one rank's arena far too small: its verdict is no shared one by itself (every rank keeps other views) -- the ranks exchange it, all run again,
    the rank with more room"""
    from line3d_amd.synth import make_scene
    V, S, N, W = 60, 160, 6, 3
    scene = make_scene(V, S, N, seed=11)
    ref = _reference(scene, N, False)
    ls, verdicts, errors, shares, calls = _run_partitioned(scene, N, W, -1, False, segments=True, arena_of={1: 1000})
    try:
        assert not errors, errors
        _check_against(ref, scene, ls, shares)
        views_exchanged = [c[0] for c in calls if c[0] >= 0]
        assert len(views_exchanged) == 2 * len(set(views_exchanged))          # every view twice: two attempts, on every rank
    finally:
        for l in ls:
            l.close()


def test_partitioned_run_repairs_failed_speculations_block_by_block():
    """A warm-up far too short for the chain's memory (one view; then none at all): every rank's speculation fails, and every block is re-run warm from
    its predecessor's true lists, one after the other -- the result is still the one chain's, and nothing falls through to another mode."""
    from line3d_amd.synth import make_scene
    V, S, N, W = 60, 160, 6, 4
    scene = make_scene(V, S, N, seed=11)
    ref = _reference(scene, N, False)
    for warmup, chunk_kb in ((1, 0), (0, 16)):
        # (chunk_kb: the hand-over in chunks of 16 KB per slot -- an all-gather hands every rank every slot, so a tail travels in bounded pieces)
        ls, verdicts, errors, shares, calls = _run_partitioned(scene, N, W, warmup, False, options={"L3D_HANDOVER_CHUNK_KB": chunk_kb} if chunk_kb else None)
        try:
            assert not errors, errors
            assert verdicts == [True] * W
            info = shares[0][0]
            assert 1 <= info["recovery_rounds"] <= W - 1 and info["blocks_rerun"] >= W - 1    # (every rank but the first was re-run, all missed blocks of a round at once)
            n5 = [c[0] for c in calls].count(-5)
            assert n5 == info["recovery_rounds"] if not chunk_kb else n5 > 4 * info["recovery_rounds"]   # one hand-over per round (in many chunks when they are small)
            _check_against(ref, scene, ls, shares)
        finally:
            for l in ls:
                l.close()


def test_partitioned_fill_in_small_blocks_of_sources():
    """the sharded fill with every rank's candidates cut into many small blocks of sources (L3D_AFF_BLOCK / L3D_AFF_WORD_BLOCK): same list"""
    from line3d_amd.synth import make_scene
    V, S, N, W = 48, 150, 6, 3
    scene = make_scene(V, S, N, seed=5)
    ref = _reference(scene, N, False)
    ls, verdicts, errors, shares, calls = _run_partitioned(scene, N, W, -1, False, options={"L3D_AFF_BLOCK": 500, "L3D_AFF_WORD_BLOCK": 300})
    try:
        assert not errors, errors
        assert verdicts == [True] * W
        _check_against(ref, scene, ls, shares)
    finally:
        for l in ls:
            l.close()


def test_segment_sharded_partition_at_256_views_of_2000x24_with_8_virtual_ranks(tmp_path):
    """The 256-view validation of rounds 5-6 under pytest: scripts/validate_partition_big.py at 256 views x 2000 segments x 24 neighbours -- the one
    chain on one GPU as reference (per-view digest of every kept list, affinity list, lines), then the PARTITIONED segment-sharded job with 8 virtual ranks
    (l3d_shard_chain_partition + finish_sharded, all-gather through the host): every rank's kept lists of its block, its affinity list and its lines
    byte-equal to the reference.  (The same script at 256 x 3000 x 24 -- profiles/r5_validate_seg_partition_256x3000x24_w8.json, re-run in round 6: 218 s as a test --
    and at 256 x 4000 x 24 with 4 ranks stays a script: eight ranks share ONE GPU here and most of the time is host-side hand-over of 20 GB of lists.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "scripts", "validate_partition_big.py")
    ref = str(tmp_path / "ref.json")
    r = subprocess.run([sys.executable, script, "ref", "256", "2000", "24", ref], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, script, "seg", "256", "2000", "24", ref, "8"], capture_output=True, text=True, timeout=1200)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and out["n_mismatches"] == 0 and not out["errors"], (out.get("mismatches"), out.get("errors"), r.stderr[-1500:])
    assert out["result"] == out["ref_result"] and out["result"]["lines"] > 1000
    assert len(out["infos"]) == 8 and all(i["world"] == 8 for i in out["infos"])
