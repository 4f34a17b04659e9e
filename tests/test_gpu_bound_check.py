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


DRIVER2 = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene
from adversarial_pairs import adversarial_view_pairs
# image sizes, principal points and segment lengths the goldens never see (round 6): VGA and 8K images with the principal point off the centre,
# segments of 1-3 pixels and segments across half the image -- (views, segments, neighbours, seed, keyword arguments of make_scene)
GEO = dict(vga=dict(width=640, height=480, f=500.0, pp=(37.0, -21.0)), uhd=dict(width=7680, height=4320, f=6000.0, pp=(-400.0, 250.0)), hd=dict())
for (V, S, N, seed, kw) in ((64, 2500, 12, 21, GEO["vga"]), (64, 2500, 12, 22, GEO["uhd"]), (48, 2500, 12, 23, dict(seg_len=(0.003, 0.008))),
                            (48, 2000, 12, 24, dict(seg_len=(0.8, 1.6), pool_factor=10.0)), (48, 2500, 12, 25, dict(GEO["uhd"], seg_len=(0.003, 0.01), noise_px=0.1)),
                            (48, 2500, 12, 26, dict(GEO["vga"], seg_len=(0.5, 1.2), pool_factor=10.0, noise_px=1.0))):
    l = Line3D("", matchingNeighbors=N)
    load_scene(l, make_scene(V, S, N, seed=seed, **kw))
    l.prepare(); l.match_views()
    print("scene", (V, S, N, seed), sorted(kw), "kept", int(l.stats()["kept"]), flush=True)
    l.close()
# adversarial pairs (tests/adversarial_pairs.py): intersection parameters k e off the end points, overlap ratios at the thresholds, tiny and spanning targets
n = 0
for geo in ("hd", "vga", "uhd"):
    for seed in range(100, 100 + %d):
        vs, F, kinds = adversarial_view_pairs(seed, n_sources=600, per_source=16, **GEO[geo])
        l = Line3D("", matchingNeighbors=2)
        for v in vs:
            l.addImage_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
        l.prepare(); l.match_views()
        n += len(kinds)
        l.close()
print("scene adversarial", n, flush=True)
"""


def test_interval_bounds_on_other_image_sizes_short_and_long_segments_and_adversarial_pairs():
    """Round 6: the error model of the level-2 bounds scales with (coordinate extent / segment length); the seven scenes above are all
    1920 x 1080 at f = 1500 with segments of 40-150 pixels.  Here: 640 x 480 and 7680 x 4320 images with the principal point off the centre, segments of
    1-3 pixels and of half the image, and pairs BUILT to sit on the decision points (intersection parameters at t in {0, 1} +- k e, overlap ratios at 0.1 /
    0.3 +- ulps) -- every decision of the bounds checked against the exact pair test by the diagnostic build; more than 1e10 pairs, none decided against it."""
    if not os.path.exists(DIAG):
        pytest.fail("line3d_amd/libline3d_amd_diag.so is not built (make -C line3d_amd/csrc diag; __graft_entry__.build() does it)")
    n_seeds = 12
    env = dict(os.environ, L3D_LIBRARY=DIAG, L3D_PAIR_STATS="1")
    p = subprocess.run([sys.executable, "-c", DRIVER2 % (ROOT, os.path.join(ROOT, "tests"), n_seeds)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, err[-3000:]
    assert out.count("scene") == 7, out
    stats = re.findall(r"\[l3d pair_mask\] pairs (\d+)\s+after wedge test ([0-9.]+)%\s+after overlap-bound test ([0-9.]+)%", err)
    assert len(stats) == 6 + 3 * n_seeds, err[-3000:]
    assert sum(int(s[0]) for s in stats) > 1.0e10
    assert "AGAINST the exact test" not in err, err[err.index("AGAINST") - 200:][:3000]
