#!/usr/bin/env python3
"""Every shortcut of the matching path against the plain formulation, on one scene of any shape: kept lists (sha256 over all views) and candidate /
kept counts with (a) everything on, (b) k_pair_mask's exact test alone (no sector test, no interval bounds), (c) the bounds without accepts,
(d) the all-pairs verification loop instead of the depth-window search.

    python scripts/ab_exactness.py VIEWS SEGMENTS NEIGHBOURS [seed] [--skip-all-pairs]"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
from line3d_amd.pipeline import Line3D, load_scene      # noqa: E402
from line3d_amd.synth import make_scene                 # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
V, S, N = int(args[0]), int(args[1]), int(args[2])
seed = int(args[3]) if len(args) > 3 else 20260
sc = make_scene(V, S, N, seed=seed)
variants = [("all shortcuts", {}), ("exact pair test alone", dict(pretest=0)), ("interval bounds without accepts", dict(pretest=7))]
if "--skip-all-pairs" not in sys.argv:
    variants.append(("all-pairs verification", dict(verify_mode=1)))
out = {}
for name, kw in variants:
    l = Line3D("", matchingNeighbors=N)
    l.keep_view_matches(True)
    if "pretest" in kw:
        l.context().set_pair_pretest(kw["pretest"])
    if "verify_mode" in kw:
        l.context().set_verify_mode(kw["verify_mode"])
    load_scene(l, sc)
    l.prepare()
    t0 = time.perf_counter()
    l.match_views()
    dt = time.perf_counter() - t0
    h = hashlib.sha256()
    for v in sc.views:
        m, med = l.view_matches(v["id"])
        h.update(m.tobytes())
    st = l.stats()
    out[name] = dict(kept_lists_sha256=h.hexdigest()[:16], candidates=int(st["raw"]), kept=int(st["kept"]), match_views_s=round(dt, 3))
    l.close()
ref = out["all shortcuts"]
ok = all((o["kept_lists_sha256"], o["candidates"], o["kept"]) == (ref["kept_lists_sha256"], ref["candidates"], ref["kept"]) for o in out.values())
print(json.dumps(dict(shape=[V, S, N], seed=seed, identical=ok, variants=out)))
