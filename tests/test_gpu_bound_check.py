"""Level 2 of k_pair_mask decides pairs -- rejects AND accepts -- from float interval bounds of the two overlap ratios, without the reference's exact
overlap test (cudawrapper.cu:569-588); their error model is a set of measured margins (l3d_kernels.hip: kLineCond, the 4 e end-point guard).  A
wrong accept would change candidate and kept lists silently.  The guard: the DIAGNOSTIC build of the library (make -C line3d_amd/csrc diag:
-DL3D_BOUND_CHECK) runs the exact test beside EVERY decision of level 2 and counts disagreements; this test drives scenes of other geometry than
the goldens' through it in a subprocess (L3D_LIBRARY selects the build) and fails on the first disagreement."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "line3d_amd", "libline3d_amd_diag.so")

DRIVER = r'''
import sys
sys.path.insert(0, %r)
from line3d_amd.pipeline import Line3D, load_scene, load_scene_worldpoints
from line3d_amd.synth import make_scene, make_scene_scattered
# (views, segments, neighbours, seed, angular step between neighbouring cameras, pixel noise): narrow and wide baselines, noise-free and noisy, dense and sparse
for (V, S, N, seed, step, noise) in ((40, 1200, 10, 11, 0.02, 0.5), (40, 1200, 10, 12, 0.45, 0.5), (40, 1200, 10, 13, 0.12, 0.0), (40, 1200, 10, 14, 0.12, 2.0), (24, 3000, 16, 15, 0.08, 0.3), (120, 500, 6, 16, 0.2, 1.0)):
    l = Line3D("", matchingNeighbors=N)
    load_scene(l, make_scene(V, S, N, seed=seed, step=step, noise_px=noise))
    l.prepare(); l.match_views()
    print("scene", (V, S, N, seed, step, noise), "kept", int(l.stats()["kept"]), flush=True)
    l.close()
l = Line3D("", matchingNeighbors=10)            # cameras in no order, epipoles inside the images, neighbours not mutual
load_scene_worldpoints(l, make_scene_scattered(32, 1500, seed=4242))
l.prepare(); l.match_views()
print("scene scattered kept", int(l.stats()["kept"]), flush=True)
l.close()
'''


def test_interval_bounds_never_decide_against_the_exact_pair_test():
    if not os.path.exists(DIAG):
        pytest.fail("line3d_amd/libline3d_amd_diag.so is not built (make -C line3d_amd/csrc diag; __graft_entry__.build() does it)")
    env = dict(os.environ, L3D_LIBRARY=DIAG, L3D_PAIR_STATS="1")
    p = subprocess.run([sys.executable, "-c", DRIVER % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, err[-3000:]
    assert out.count("scene") == 7, out
    stats = re.findall(r"\[l3d pair_mask\] pairs (\d+)\s+after wedge test ([0-9.]+)%\s+after overlap-bound test ([0-9.]+)%", err)
    assert len(stats) == 7, err[-3000:]                          # one line per context: the diagnostic counters were live
    assert sum(int(s[0]) for s in stats) > 2.5e9
    assert all(float(s[2]) < 2.0 for s in stats), stats         # the bounds decide: only a sliver reaches the exact test in the shipped build
    assert "AGAINST the exact test" not in err, err[err.index("AGAINST") - 200:][:3000]
