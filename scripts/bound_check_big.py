#!/usr/bin/env python3
"""Level 2 of k_pair_mask against the exact pair test on MANY more pairs than the test suite affords (tests/test_gpu_bound_check.py: 1e10):

    python3 scripts/bound_check_big.py out.json [--pairs 1e11] [--seconds 1200]

drives the diagnostic build (line3d_amd/libline3d_amd_diag.so, -DL3D_BOUND_CHECK: the exact overlap test beside EVERY decision of the interval bounds) over
freshly seeded scenes of the geometries the second test uses -- 640 x 480 and 7680 x 4320 images with off-centre principal points, 1-3-pixel and image-spanning
segments, narrow and wide baselines, noise from 0 to 2 pixels -- and over sets of adversarial pairs (tests/adversarial_pairs.py), seed after seed, until the
pair target or the time limit is reached.  Every scene runs in a child process (the library prints its counters when a context closes); the JSON holds the
pairs checked per family and the number of decisions against the exact test (must be 0)."""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "line3d_amd", "libline3d_amd_diag.so")

CHILD = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene
from adversarial_pairs import adversarial_view_pairs
GEO = dict(vga=dict(width=640, height=480, f=500.0, pp=(37.0, -21.0)), uhd=dict(width=7680, height=4320, f=6000.0, pp=(-400.0, 250.0)), hd=dict())
family, seed = %(family)r, %(seed)d
rng = np.random.default_rng(seed)
if family == "adversarial":
    for geo in ("hd", "vga", "uhd"):
        for s in range(seed * 16, seed * 16 + 16):
            vs, F, kinds = adversarial_view_pairs(s, n_sources=600, per_source=16, **GEO[geo])
            l = Line3D("", matchingNeighbors=2)
            for v in vs:
                l.addImage_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
            l.prepare(); l.match_views(); l.close()
else:
    kw = dict(GEO[family.split("+")[0]])
    if "+short" in family:
        kw["seg_len"] = (0.003, 0.01)
    if "+long" in family:
        kw["seg_len"] = (0.5, 1.6); kw["pool_factor"] = 10.0
    kw["noise_px"] = float(rng.choice([0.0, 0.1, 0.5, 1.0, 2.0]))
    kw["step"] = float(rng.choice([0.02, 0.08, 0.12, 0.2, 0.45]))
    V, S, N = 96, 3000, 12
    l = Line3D("", matchingNeighbors=N)
    load_scene(l, make_scene(V, S, N, seed=seed, **kw))
    l.prepare(); l.match_views(); l.close()
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--pairs", type=float, default=1e11)
    ap.add_argument("--seconds", type=float, default=1200.0)
    ap.add_argument("--families", default="", help="comma-separated subset of the families (default: all eight in turn)")
    ap.add_argument("--first-seed", type=int, default=1000)
    a = ap.parse_args()
    if not os.path.exists(DIAG):
        sys.exit("line3d_amd/libline3d_amd_diag.so is not built (make -C line3d_amd/csrc diag)")
    env = dict(os.environ, L3D_LIBRARY=DIAG, L3D_PAIR_STATS="1")
    families = ["hd", "vga", "uhd", "hd+short", "hd+long", "uhd+short", "vga+long", "adversarial"]
    if a.families:
        families = [f for f in a.families.split(",") if f in families]
    per = {f: dict(pairs=0, runs=0, against=0, to_exact_test_pct_max=0.0) for f in families}
    t0 = time.time()
    seed, total, against = a.first_seed, 0, 0
    first_offenders = []
    while total < a.pairs and time.time() - t0 < a.seconds:
        fam = families[seed % len(families)]
        p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, tests=os.path.join(ROOT, "tests"), family=fam, seed=seed)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        err = p.stderr.decode()
        if p.returncode != 0:
            sys.exit("child failed (family %s, seed %d):\n%s" % (fam, seed, err[-3000:]))
        stats = re.findall(r"\[l3d pair_mask\] pairs (\d+)\s+after wedge test ([0-9.]+)%\s+after overlap-bound test ([0-9.]+)%", err)
        if not stats:
            sys.exit("no counters (is the diagnostic library loaded?):\n" + err[-2000:])
        n = sum(int(s[0]) for s in stats)
        bad = [int(x) for x in re.findall(r"(\d+) pairs were decided by level 2 AGAINST the exact test", err)]
        per[fam]["pairs"] += n; per[fam]["runs"] += 1; per[fam]["against"] += sum(bad)
        per[fam]["to_exact_test_pct_max"] = max(per[fam]["to_exact_test_pct_max"], max(float(s[2]) for s in stats))
        total += n; against += sum(bad)
        if bad and len(first_offenders) < 4:
            first_offenders.append(dict(family=fam, seed=seed, text=err[err.index("AGAINST") - 200:][:1500]))
        print("seed %d %-12s %.3e pairs (total %.3e, %.0f s), against the exact test: %d" % (seed, fam, n, total, time.time() - t0, against), flush=True)
        seed += 1
    out = dict(pairs_checked=total, decisions_against_the_exact_test=against, seconds=round(time.time() - t0, 1), families=per, first_offenders=first_offenders,
               note="diagnostic build (-DL3D_BOUND_CHECK): the exact overlap test of cudawrapper.cu:569-588 runs beside every decision of level 2's interval bounds; "
                    "scenes of 96 views x 3000 segments x 12 neighbours with the geometry of the family, noise and baseline drawn per seed; adversarial: 48 two-view sets of 9600 crafted pairs per seed")
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("pairs_checked", "decisions_against_the_exact_test", "seconds")}))
    sys.exit(1 if against else 0)


if __name__ == "__main__":
    main()
