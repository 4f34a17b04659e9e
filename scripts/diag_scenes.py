"""Six scenes of other geometry (baselines of 0.02 and 0.45 rad between neighbouring cameras, noise-free and 2-pixel-noise observations, a dense and a
sparse shape) through the resident chain: run it under the diagnostic build of k_pair_mask (scripts/build_variant.sh x.so -DL3D_BOUND_CHECK;
L3D_LIBRARY=x.so L3D_PAIR_STATS=1 python scripts/diag_scenes.py) -- every decision of the interval bounds is then checked against the exact pair test
and offenders are printed when a context closes.  Round 4, final code: none in 1.1e10 pairs."""
import sys, os
sys.path.insert(0, os.getcwd())
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene
for (V,S,N,seed,step,noise) in ((100,1500,10,11,0.02,0.5),(100,1500,10,12,0.45,0.5),(100,1500,10,13,0.12,0.0),(100,1500,10,14,0.12,2.0),(60,3500,20,15,0.08,0.3),(300,600,6,16,0.2,1.0)):
    sc = make_scene(V,S,N,seed=seed,step=step,noise_px=noise)
    l = Line3D("", matchingNeighbors=N)
    load_scene(l, sc); l.prepare(); l.match_views()
    st = l.stats()
    print("scene", (V,S,N,seed,step,noise), "raw", int(st["raw"]), "kept", int(st["kept"]), flush=True)
    l.close()
