import numpy as np


def canon_lines(result):
    """Order-free form of a Line3D result: {sorted 2D-segment id tuple: sorted list of unordered endpoint pairs}."""
    out = {}
    for seg2, seg3 in result:
        key = tuple(sorted((int(c), int(s)) for c, s in seg2))
        pairs = []
        for P1, P2 in seg3:
            a, b = tuple(np.asarray(P1, float)), tuple(np.asarray(P2, float))
            pairs.append((a, b) if a <= b else (b, a))
        out[key] = sorted(pairs)
    return out


def assert_lines_equal(got, exp, tol=1e-4):
    g, e = canon_lines(got), canon_lines(exp)
    assert set(g) == set(e), "2D-segment id sets differ: %d vs %d lines, %d common" % (len(g), len(e), len(set(g) & set(e)))
    worst = 0.0
    for k in e:
        assert len(g[k]) == len(e[k]), "line %r: %d vs %d 3D segments" % (k[:2], len(g[k]), len(e[k]))
        for (ga, gb), (ea, eb) in zip(g[k], e[k]):
            worst = max(worst, float(np.max(np.abs(np.array(ga) - np.array(ea)))), float(np.max(np.abs(np.array(gb) - np.array(eb)))))
    assert worst <= tol, "endpoint mismatch %g > %g" % (worst, tol)
    return worst
