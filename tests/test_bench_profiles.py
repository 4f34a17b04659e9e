"""bench.py prices a run's dominant kernel against PMC summaries under profiles/ (VALU wave instructions and HBM bytes PER LAUNCH).  Those are
properties of the per-view shape (segments, neighbours): a run of another shape must get no fraction rather than one computed from the
default shape's counters (a 40 x 4000 x 24 run printed 0.0043 from config 2's instruction count)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_profiles_are_looked_up_by_shape():
    import bench
    d, src = bench.profile_for_shape("valu", 2000, 12)
    assert src and "verify_window" in d and d.get("_shape", [64, 2000, 12])[1:] == [2000, 12]
    d2, src2 = bench.profile_for_shape("valu", 1234, 7)            # no such profile: nothing, never another shape's
    assert d2 == {} and src2 is None
    for kind in ("valu", "traffic"):
        d5, src5 = bench.profile_for_shape(kind, 4000, 24)
        if src5 is not None:                                        # (present once scripts/measure_round.sh ran at that shape)
            assert d5["_shape"][1:] == [4000, 24] and "4000x24" in src5


def test_every_committed_summary_names_a_commit_and_unstamped_ones_are_the_default_shape():
    import glob
    for q in glob.glob(os.path.join(ROOT, "profiles", "r*_valu.json")) + glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")):
        d = json.load(open(q))
        shp = d.get("_shape", [64, 2000, 12])
        assert len(shp) == 3 and all(int(x) > 0 for x in shp), q
