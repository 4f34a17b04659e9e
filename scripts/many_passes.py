"""400 consecutive matchViews of the bench scene on one object: candidate and kept counts of every pass, lines and affinity entries of every 20th,
against the first (a stale read of a result record, a race between the streams would show up here): python scripts/many_passes.py"""
import sys, os, hashlib
sys.path.insert(0, os.getcwd())
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene
sc = make_scene(64, 2000, 12, seed=20260)
l = Line3D("", matchingNeighbors=12)
load_scene(l, sc); l.prepare()
ref = None; bad = 0
for p in range(400):
    l.match_views()
    s1 = l.stats()
    key = (s1["raw"], s1["kept"])
    if p % 20 == 0:
        l.finish(False)
        key = key + (len(l.getResult()), l.stats()["edges"])
    else:
        key = key + (None, None)
    if ref is None: ref = key
    if key[:2] != ref[:2] or (key[2] is not None and key[2:] != ref[2:]): bad += 1; print("pass", p, key, ref)
print("passes 400, deviations", bad, ref)
