"""Adversarial segment pairs for stage 1 (k_pair_mask, cudawrapper.cu:538-611): two views, the targets of view B built FROM the epipolar geometry of the
sources of view A so that the quantities the interval bounds of level 2 decide on sit at their decision points:
  * intersection parameters at an END POINT of a segment, t in {0, 1} +- k e for k = 1..8 and e from 2^-22 to 2^-8 (the point-on-segment tests of
    cudawrapper.cu:135-141 flip there; the bounds must say "cannot tell" inside their guard and be right outside it);
  * overlap ratios AT the thresholds 0.1 and 0.3 (cudawrapper.cu:586-588), +- a few ulp and +- 1e-6 .. 1e-3;
  * targets of 1-3 pixels and targets that span the image, against sources of ordinary length.
The coordinates are computed in double from the double fundamental matrix (the one the library casts to float, line3D.cc:745) and rounded to float32 --
that rounding IS the perturbation at the scale the bounds' error model works on.  Test infrastructure; no reference code involved."""
import numpy as np


def _line_through(p, q):
    return np.cross(np.append(p, 1.0), np.append(q, 1.0))


def _meet(l, m):
    x = np.cross(l, m)
    return x[:2] / x[2] if abs(x[2]) > 1e-12 else None


def craft_targets(F, src_segs, width, height, rng, per_source=16):
    """F: 3x3 double with l_B = F p_A.  Returns (targets float32 (n, 4), kind labels)."""
    out, kinds = [], []
    ks = np.arange(1, 9)
    for s in np.asarray(src_segs, dtype=np.float64):
        p1, p2 = np.array([s[0], s[1], 1.0]), np.array([s[2], s[3], 1.0])
        l1, l2 = F @ p1, F @ p2
        made = 0
        tries = 0
        while made < per_source and tries < per_source * 8:
            tries += 1
            # a random line m through the image of B: it meets the two epipolar lines in X1, X2
            a = np.array([rng.uniform(0.05, 0.95) * width, rng.uniform(0.05, 0.95) * height])
            ang = rng.uniform(0.0, np.pi)
            m = _line_through(a, a + np.array([np.cos(ang), np.sin(ang)]))
            X1, X2 = _meet(m, l1), _meet(m, l2)
            if X1 is None or X2 is None:
                continue
            d = X2 - X1
            L = np.hypot(*d)
            if not (2.0 < L < 4.0 * width):
                continue
            fam = made % 4
            if fam == 0:        # an intersection point k e off an end point of the target
                e = 2.0 ** -rng.integers(8, 23)
                k = int(rng.choice(ks)) * (1 if rng.uniform() < 0.5 else -1)
                d1, d2 = k * e, rng.uniform(-0.4, 0.4)
                q1, q2 = X1 - d1 * d, X2 + d2 * d
                kind = "end+%de" % k
            elif fam == 1:      # overlap ratio of the target against [X1, X2] at a threshold
                thr = 0.1 if rng.uniform() < 0.5 else 0.3
                eps = float(rng.choice([0.0, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 1e-3, -1e-3]))
                r = thr + eps
                if rng.uniform() < 0.5:     # the target inside the intersection pair: IoU = |target| / |pair|
                    q1, q2 = X1 + rng.uniform(0.0, 1.0 - r) * d, None
                    q2 = q1 + r * d
                else:                       # the pair inside the target: IoU = |pair| / |target|
                    ext = (1.0 / r - 1.0)
                    u = rng.uniform(0.0, 1.0)
                    q1, q2 = X1 - u * ext * d, X2 + (1.0 - u) * ext * d
                kind = "iou%.1f%+g" % (thr, eps)
            elif fam == 2:      # a target of 1-3 pixels at an end of the pair
                n = d / L
                q1 = X1 + rng.uniform(-1.0, 1.0) * n
                q2 = q1 + rng.uniform(1.0, 3.0) * n
                kind = "tiny"
            else:               # a target that spans the image along m
                n = d / L
                q1, q2 = X1 - rng.uniform(0.2, 1.0) * width * n, X2 + rng.uniform(0.2, 1.0) * width * n
                kind = "span"
            q = np.array([q1[0], q1[1], q2[0], q2[1]])
            if fam != 3 and not (np.all(q[[0, 2]] > -0.5 * width) and np.all(q[[0, 2]] < 1.5 * width) and np.all(q[[1, 3]] > -0.5 * height) and np.all(q[[1, 3]] < 1.5 * height)):
                continue
            if rng.uniform() < 0.5:
                q = q[[2, 3, 0, 1]]
            out.append(q)
            kinds.append(kind)
            made += 1
    return np.ascontiguousarray(np.array(out, dtype=np.float32)), kinds


def adversarial_view_pairs(seed, width=1920, height=1080, f=1500.0, pp=(0.0, 0.0), n_sources=600, per_source=16, step=0.12, seg_len=(0.1, 0.4)):
    """Four views (ids 0..3) as dicts for Line3D.addImage_fixed_sim: A = a helix view's own segments, B = crafted targets (+ its own segments, so that
    ordinary pairs surround the adversarial ones), C, D = ordinary views.  Returns ([A, B, C, D], F_AB, kinds)."""
    from line3d_amd.synth import make_scene
    sc = make_scene(4, n_sources, 2, seed=seed, width=width, height=height, f=f, pp=pp, step=step, seg_len=seg_len, noise_px=0.3)
    A, B, C, D = sc.views                                       # (C, D: ordinary views -- the pipeline refuses fewer than four, line3D.cc:347-351)
    K = A["K"]
    Ki = np.linalg.inv(K)
    # F with l_B = F p_A: K^-T [t]x R K^-1 for the relative pose A -> B (line3D.cc:1949-1993 computes the same matrix in double)
    R = B["R"] @ A["R"].T
    t = B["t"] - R @ A["t"]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = Ki.T @ tx @ R @ Ki
    rng = np.random.default_rng(seed)
    tg, kinds = craft_targets(F, A["segments"], width, height, rng, per_source)
    B = dict(B)
    B["segments"] = np.ascontiguousarray(np.concatenate([tg, B["segments"]])[:16000], dtype=np.float32)
    A = dict(A, sims={1: 1.0, 2: 0.5})
    B["sims"] = {0: 1.0, 2: 0.5}
    C = dict(C, sims={1: 0.5, 3: 0.5})
    D = dict(D, sims={2: 0.5, 1: 0.3})
    return [A, B, C, D], F, kinds
