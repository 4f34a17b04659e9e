#!/usr/bin/env python3
"""A/B of library builds on one box: python scripts/ab_variants.py [--passes 20] [--rounds 3] name=path.so ...
Every round runs every build once (interleaved, so drift of the box hits all alike); prints the best and the median ms per matchViews pass of
config 2 per build (scripts/bench_shape.py in a fresh process with L3D_LIBRARY)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
passes, rounds = 20, 3
while args and args[0].startswith("--"):
    k, v = args[0], args[1]
    args = args[2:]
    if k == "--passes":
        passes = int(v)
    elif k == "--rounds":
        rounds = int(v)
builds = [a.split("=", 1) for a in args]
res = {n: [] for n, _ in builds}
for r in range(rounds):
    for n, p in builds:
        env = dict(os.environ)
        p, _, envs = p.partition(":")                       # name=path.so[:ENV=VALUE,ENV=VALUE]
        for kv in filter(None, envs.split(",")):
            env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
        if p != "default":
            env["L3D_LIBRARY"] = os.path.abspath(p)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_shape.py"), "64", "2000", "12", str(passes)], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[n].append((d["ms_per_pass"], d["median_ms"], d["kernels_ms"]))
        except Exception:       # noqa: BLE001
            res[n].append((float("nan"), float("nan"), out.stderr[-300:]))
for n, _ in builds:
    best = min(x[0] for x in res[n])
    med = sorted(x[1] for x in res[n])[len(res[n]) // 2]
    print("%-28s best %.3f  median-of-medians %.3f  rounds %s  kernels %s" % (n, best, med, " ".join("%.2f" % x[0] for x in res[n]), res[n][-1][2]))
