"""Seeded affinity lists for the replicator-dynamics pins (tests/test_oracle_pins.py, tests/golden/make_golden_rdd.py): symmetric
PATTERNS as clusterSegments2D builds them (every edge in both directions, no self loops -- so no row or column of the sparse matrix is
empty, which the reference's kernels require: cudawrapper.cu:725, :784-785 index with the start offset unchecked), in insertion order."""
import numpy as np

EDGE_DTYPE = np.dtype([("i", np.int32), ("j", np.int32), ("w", np.float32)])


def make_list(seed, n, n_pairs, symmetric_values=True, tie_levels=True, tiny=False):
    rng = np.random.default_rng(seed)
    pairs = set()
    for v in range(n):                                               # every node gets at least one edge
        u = int(rng.integers(0, n - 1))
        u += u >= v
        pairs.add((min(u, v), max(u, v)))
    while len(pairs) < n_pairs:
        a, b = (int(x) for x in rng.integers(0, n, 2))
        if a != b:
            pairs.add((min(a, b), max(a, b)))
    levels = np.array([0.011, 0.25, 0.2500001, 0.5, 0.75, 1.0], np.float32)
    out = []
    for a, b in sorted(pairs, key=lambda p: (p[0] * 7919 + p[1] * 104729) % 1000003):
        w = levels[rng.integers(0, len(levels))] if tie_levels and rng.random() < 0.4 else np.float32(rng.random())
        if tiny and rng.random() < 0.2:
            w = np.float32(rng.random() * 1e-12)                     # (the L3D_EPS_G clamps, cudawrapper.cu:741-742, :811-812)
        w2 = w if symmetric_values else np.float32(rng.random())
        out.append((a, b, w)); out.append((b, a, w2))
    return np.array(out, dtype=EDGE_DTYPE)


CASES = [dict(seed=1, n=12, n_pairs=40), dict(seed=2, n=30, n_pairs=300), dict(seed=3, n=400, n_pairs=2500),
         dict(seed=4, n=400, n_pairs=2500, symmetric_values=False), dict(seed=5, n=150, n_pairs=900, tiny=True),
         dict(seed=6, n=1200, n_pairs=5000, tie_levels=False), dict(seed=7, n=64, n_pairs=64)]
