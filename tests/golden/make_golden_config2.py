"""Generates tests/golden/config2_full.npz: the whole BASELINE configs[1] / configs[3] scene (64 views x 2000 segments x 12
neighbours, seed 20260 -- the scene bench.py times) through the ORACLE ALONE, no GPU input anywhere:

    python tests/golden/make_golden_config2.py [--views 64] [--threads 8] [--out tests/golden/config2_full.npz]

matchViews (line3D.cc:620-648) view by view: the C oracle's compute_pairwise_matches on ranges of source segments, one range
per host thread (a source segment's verification reads that segment's candidates only, cudawrapper.cu:656-706, so ranges
concatenate; the median of cudawrapper.cu:1058-1076 is formed over the concatenated best-depth list), then the oracle's
matching_commit (reverse matches, potential correspondences, only-best overwrite: line3D.cc:834-884); greedySelection and
clusterSegments2D with the literal `used` rule (line3D.cc:899-1221), clustering, line fit -- once without and once with
the diffusion (line3D.cc:1255-1303).  About 45 core-minutes.

Fixture = data only: per view the sha256 of the kept list (36-byte records) and its median, the number of kept matches;
sha256 + size of the affinity list and of the first-touch node table; the final lines of both settings (2-D segment ids and
3-D end points).  The GPU test is tests/test_gpu_config2_golden.py.
"""
import argparse
import hashlib
import os
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import l3d_oracle_pipeline as op  # noqa: E402
from line3d_amd.synth import make_scene, make_scene_scattered  # noqa: E402

V, S, N, SEED = 64, 2000, 12, 20260


def match_view_threaded(o, v, threads):
    """OracleLine3D.perform_matching with the seam call cut into source-segment ranges."""
    mv = o.marshal_view(v)
    in_arr = o.existing_localized(v, mv)
    S_src = len(mv["src_segs"])
    if len(mv["tbm"]) == 0:                                   # cudawrapper.cu:877-878: returns before touching anything
        matches, median = op.compute_pairwise_matches(
            o.lib, mv["src_segs"], mv["RtKinv_src"], mv["C_src"], mv["tgt_segs"], mv["offsets"], mv["F"], mv["RtKinv"],
            mv["centers"], mv["P"], mv["tbm"], in_arr, mv["l2g"], mv["k_upper"], mv["k_lower"], float(o.sigma_p),
            float(o.sigma_a), mv["spatial_k"], median_depth=1.0)
        return mv, in_arr, matches, median
    T = max(1, min(threads, S_src))
    cuts = [S_src * i // T for i in range(T + 1)]
    parts = [None] * T

    def work(i):
        parts[i] = op.compute_pairwise_matches(
            o.lib, mv["src_segs"], mv["RtKinv_src"], mv["C_src"], mv["tgt_segs"], mv["offsets"], mv["F"], mv["RtKinv"],
            mv["centers"], mv["P"], mv["tbm"], in_arr, mv["l2g"], mv["k_upper"], mv["k_lower"], float(o.sigma_p),
            float(o.sigma_a), mv["spatial_k"], median_depth=1.0, seg_range=(cuts[i], cuts[i + 1]), want_best=True)
    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    matches = np.concatenate([p[0] for p in parts])
    best = np.concatenate([p[2] for p in parts])              # depth pairs entering the median, in segment order
    median = np.float32(-1.0)                                 # cudawrapper.cu:1064-1076
    if len(best):
        median = np.sort(best, kind="stable")[len(best) // 2]
    return mv, in_arr, matches, float(median)


def pack_lines(result):
    """[(seg2 list of (cam, seg), seg3 list of (P, Q))] -> flat arrays."""
    ids, id_off, pts, pt_off = [], [0], [], [0]
    for seg2, seg3 in result:
        for c, s in seg2:
            ids.append((int(c), int(s)))
        id_off.append(len(ids))
        for P, Q in seg3:
            pts.append(np.concatenate([np.asarray(P, np.float64), np.asarray(Q, np.float64)]))
        pt_off.append(len(pts))
    return (np.array(ids, np.int32).reshape(-1, 2), np.array(id_off, np.int64),
            np.array(pts, np.float64).reshape(-1, 6), np.array(pt_off, np.int64))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=V)
    ap.add_argument("--segments", type=int, default=S)
    ap.add_argument("--neighbors", type=int, default=N)
    ap.add_argument("--seed", type=int, default=SEED)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=os.path.join(HERE, "config2_full.npz"))
    ap.add_argument("--cache", default="", help="scratch .npz of the oracle's own kept lists: written after matchViews, read instead of recomputing the "
                                                "seam calls when it exists (the commits are always replayed); not a fixture, not committed")
    ap.add_argument("--matching-only", action="store_true", help="matchViews only (kept lists and medians per view): what tests/golden/config3_matching.npz holds for the "
                                                                  "512-view scene -- python tests/golden/make_golden_config2.py --views 512 --matching-only --out tests/golden/config3_matching.npz "
                                                                  "(about 6 core-hours)")
    ap.add_argument("--scattered", action="store_true", help="synth.make_scene_scattered instead of the helix: cameras in no order, ragged views, neighbours chosen by the library from "
                                                             "shared world points through Line3D::addImage (line3D.cc:95-217, 476-549: not mutual, twins under min_baseline) -- "
                                                             "python tests/golden/make_golden_config2.py --scattered --views 48 --segments 1500 --neighbors 10 --seed 4242 --out tests/golden/scattered_48x1500x10.npz")
    a = ap.parse_args()
    t0 = time.time()
    o = op.OracleLine3D(matching_neighbors=a.neighbors)
    if a.scattered:
        scene = make_scene_scattered(a.views, a.segments, seed=a.seed)
        for v in scene.views:
            assert o.add_image(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["worldpoints"])
    else:
        scene = make_scene(a.views, a.segments, a.neighbors, seed=a.seed)
        for v in scene.views:
            o.add_image_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
    print("scene + collinearity %.1f s" % (time.time() - t0), flush=True)
    o.computation = True
    o.track_potential = not a.matching_only
    o.matched, o.potential, o.result = {}, {}, []
    o.find_visual_neighbors()
    o.transform_geometry()
    g = {"shape": np.array([a.views, a.segments, a.neighbors, a.seed], np.int64), "scattered": np.int64(1 if a.scattered else 0)}
    if a.scattered:         # what the library chose: the neighbourhoods (ragged rows: view id, then its neighbours), the views' segment counts
        g["neighbors_flat"] = np.array([x for v in sorted(o.visual_neighbors) for x in [v, len(o.visual_neighbors[v])] + list(o.visual_neighbors[v])], np.int64)
        g["n_segments"] = np.array([len(v["segments"]) for v in scene.views], np.int64)
    kept_sha, kept_n, medians = [], [], []
    cache = dict(np.load(a.cache)) if a.cache and os.path.exists(a.cache) else None
    if cache is not None:
        assert [int(x) for x in cache["shape"]] == [a.views, a.segments, a.neighbors, a.seed]
    store = {"shape": g["shape"]}
    for v in sorted(o.visual_neighbors):                      # match_views, line3D.cc:620-648
        if len(o.visual_neighbors[v]) == 0:
            continue
        t1 = time.time()
        for n in o.visual_neighbors[v]:
            o._fundamental(v, n)
        if cache is not None:
            in_arr, matches, median = cache["in_%d" % v], cache["kept_%d" % v], float(cache["median_%d" % v])
        else:
            mv, in_arr, matches, median = match_view_threaded(o, v, a.threads)
        store["in_%d" % v], store["kept_%d" % v], store["median_%d" % v] = in_arr, matches, np.float32(median)
        t2 = time.time()
        o.matching_commit(v, matches, median)
        kept_sha.append(sha(matches))
        kept_n.append(len(matches))
        medians.append(np.float32(median))
        print("view %d: %d existing, %d kept, median %.6f  (%.1f s seam, %.1f s commit)" % (v, len(in_arr), len(matches), median, t2 - t1, time.time() - t2), flush=True)
    if a.cache and cache is None:
        np.savez(a.cache, **store)
    g["kept_sha256"] = np.array(kept_sha)
    g["kept_n"] = np.array(kept_n, np.int64)
    g["median"] = np.array(medians, np.float32)
    if a.matching_only:
        np.savez_compressed(a.out, **g)
        print("written %s (%d bytes) in %.0f s" % (a.out, os.path.getsize(a.out), time.time() - t0))
        return
    t1 = time.time()
    o.greedy_selection()
    print("greedy selection %.1f s: %d hypotheses" % (time.time() - t1, len(o.best_match)), flush=True)
    g["n_hypotheses"] = np.int64(len(o.best_match))
    for diffusion in (False, True):
        t1 = time.time()
        o.cluster_segments_2D(diffusion)
        tag = "rdd" if diffusion else "plain"
        ids, id_off, pts, pt_off = pack_lines(o.result)
        g["%s_ids" % tag], g["%s_id_off" % tag], g["%s_pts" % tag], g["%s_pt_off" % tag] = ids, id_off, pts, pt_off
        g["%s_labels_sha256" % tag] = np.array(sha(o.labels))
        g["%s_edges_sha256" % tag] = np.array(sha(o.affinity_final))
        if not diffusion:
            l2g = np.array([o.local2global[i] for i in range(len(o.local2global))], np.int32).reshape(-1, 2)
            g["affinity_sha256"] = np.array(sha(o.affinity))
            g["affinity_n"] = np.int64(len(o.affinity))
            g["affinity_wsum"] = np.float64(o.affinity["w"].astype(np.float64).sum())
            g["local2global_sha256"] = np.array(sha(l2g))
            g["n_nodes"] = np.int64(len(l2g))
        print("clusterSegments2D(diffusion=%s) %.1f s: %d affinity entries, %d nodes, %d lines" % (diffusion, time.time() - t1, len(o.affinity), len(o.local2global), len(o.result)), flush=True)
    np.savez_compressed(a.out, **g)
    print("written %s (%d bytes) in %.0f s" % (a.out, os.path.getsize(a.out), time.time() - t0))


if __name__ == "__main__":
    main()
