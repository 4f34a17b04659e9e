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


# ---- synthetic SfM files (test data writers; the product only reads these formats) ----------------------------------
def synth_worldpoints(scene, n_points=400, seed=5):
    """Random 3-D points in the scene box and, per point, the cameras that see it: list of (xyz, [(cam, key, x, y)...])."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform([-1, -0.6, -1], [1, 0.6, 1], (n_points, 3))
    out = []
    for X in pts:
        obs = []
        for ci, v in enumerate(scene.views):
            xc = v["R"] @ X + v["t"]
            if xc[2] <= 0.1:
                continue
            p = v["K"] @ xc
            x, y = p[0] / p[2], p[1] / p[2]
            if 0 <= x < v["width"] and 0 <= y < v["height"]:
                obs.append((ci, len(obs), x - v["width"] / 2.0, y - v["height"] / 2.0))
        out.append((X, obs))
    return out


def _quat_from_R(R):
    w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
    x = (R[2, 1] - R[1, 2]) / (4.0 * w)
    y = (R[0, 2] - R[2, 0]) / (4.0 * w)
    z = (R[1, 0] - R[0, 1]) / (4.0 * w)
    return w, x, y, z


def write_nvm(path, scene, points):
    with open(path, "w") as f:
        f.write("NVM_V3\n\n%d\n" % len(scene.views))
        for i, v in enumerate(scene.views):
            w, x, y, z = _quat_from_R(v["R"])
            Cc = -v["R"].T @ v["t"]
            f.write("img_%04d.jpg %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g 0 0\n" % (i, v["K"][0, 0], w, x, y, z, Cc[0], Cc[1], Cc[2]))
        f.write("\n%d\n" % len(points))
        for X, obs in points:
            f.write("%.17g %.17g %.17g 128 128 128 %d " % (X[0], X[1], X[2], len(obs)) + " ".join("%d %d %.3f %.3f" % o for o in obs) + "\n")
        f.write("\n0\n")


def write_bundler(path, scene, points):
    with open(path, "w") as f:
        f.write("# Bundle file v0.3\n%d %d\n" % (len(scene.views), len(points)))
        for v in scene.views:
            R, t = v["R"].copy(), v["t"].copy()
            R[1:] *= -1.0
            t[1:] *= -1.0
            f.write("%.17g 0 0\n" % v["K"][0, 0])
            for r in range(3):
                f.write("%.17g %.17g %.17g\n" % tuple(R[r]))
            f.write("%.17g %.17g %.17g\n" % tuple(t))
        for X, obs in points:
            f.write("%.17g %.17g %.17g\n128 128 128\n%d " % (X[0], X[1], X[2], len(obs)) + " ".join("%d %d %.3f %.3f" % o for o in obs) + "\n")


# ---- oracle on a slice of one full-size view (mid-chain: with the reverse matches its earlier neighbours handed it) -------
def digest_lists(lists):
    import hashlib
    h = hashlib.sha256()
    for vid in sorted(lists):
        h.update(lists[vid][0].tobytes())
        h.update(np.float32(lists[vid][1]).tobytes())
    return h.hexdigest()


def oracle_view_slice(scene, lists, vid, seg_lo, seg_hi, N, threads=None, reference=None):
    """compute_pairwise_matches of the oracle for source segments [seg_lo, seg_hi) of view `vid` in the state matchViews has
    when it reaches that view (line3D.cc:620-648): matched_ after views 0..vid-1 (:875-881) and, as the existing matches
    (view.cc:200-224), the kept matches of the earlier views towards `vid` taken from `lists` (view id -> (matches, median);
    view ids are 0..V-1 in processing order).  The range is cut into one piece per host thread (the C oracle releases the
    GIL; a source segment's verification only reads that segment's candidates, so the pieces concatenate).
    reference = oracle/_spliced/libkernels_spliced.so: the libm build of the oracle with the REFERENCE's own kernels plugged in (one thread: their launch
    variables are process globals)."""
    import os
    import threading
    import l3d_oracle_pipeline as op
    o = op.OracleLine3D(matching_neighbors=N, use_collinearity=False, libm=reference is not None)
    if reference is not None:
        threads = 1
    for v in scene.views:
        o.add_image_fixed_sim(v["id"], v["width"], v["height"], v["segments"], v["K"], v["R"], v["t"], v["sims"])
    o.computation = True
    o.matched, o.potential = {}, {}
    o.find_visual_neighbors()
    o.transform_geometry()
    for n in o.visual_neighbors[vid]:
        o._fundamental(vid, n)
    for a in range(vid):
        for nb in o.visual_neighbors[a]:
            o.matched.setdefault(a, {})[nb] = True
            if a in o.visual_neighbors.get(nb, []):
                o.matched.setdefault(nb, {})[a] = True
    mv = o.marshal_view(vid)
    ex = []
    for a in range(vid):
        m, _ = lists[a]
        sel = m[m["camID2"] == vid]
        if len(sel) and a in mv["g2l"]:
            r = np.zeros(len(sel), dtype=op.MATCH_DTYPE)
            r["segID1"], r["segID2"], r["camID2"] = sel["segID2"], sel["segID1"], mv["g2l"][a]
            r["depths"] = sel["depths"][:, [2, 3, 0, 1]]
            ex.append(r)
    existing = np.concatenate(ex) if ex else np.zeros(0, dtype=op.MATCH_DTYPE)
    if threads is None:
        try:
            threads = len(os.sched_getaffinity(0))
        except AttributeError:
            threads = os.cpu_count() or 1
        threads = max(1, min(threads, 64, seg_hi - seg_lo))
    cuts = [seg_lo + (seg_hi - seg_lo) * i // threads for i in range(threads + 1)]
    parts = [None] * threads

    def work(i):
        parts[i] = op.compute_pairwise_matches(
            o.lib, mv["src_segs"], mv["RtKinv_src"], mv["C_src"], mv["tgt_segs"], mv["offsets"], mv["F"], mv["RtKinv"],
            mv["centers"], mv["P"], mv["tbm"], existing, mv["l2g"], mv["k_upper"], mv["k_lower"], 3.5, 10.0, mv["spatial_k"],
            seg_range=(cuts[i], cuts[i + 1]))[0]
    if reference is not None:
        try:
            op.set_reference_kernels(o.lib, reference)
            work(0)
        finally:
            op.set_reference_kernels(o.lib, None)
        return np.concatenate(parts), mv, existing
    th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return np.concatenate(parts), mv, existing


# ---- an all-gather for virtual ranks that run as threads of one process on one GPU (block mode tests, scripts/soak_blocks.py) --------------
def thread_exchange(world, on_device=False, timeout=600.0):
    """An all-gather for `world` virtual ranks that run as threads of this process on one GPU (the l3d_exchange_fn contract: rank r's
    `slot_bytes` land at recv_block + r * slot_bytes on every rank): through host buffers, two barriers per call.  on_device: the ranks share
    the process and the device, so every rank copies the others' send slots device to device (big slots: scripts/validate_partition_big.py).
    A rank that waits longer than `timeout` seconds for the others breaks the barrier: every exchange fails, nobody hangs."""
    import ctypes as C
    import threading
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    barrier = threading.Barrier(world, timeout=timeout)
    bufs = [None] * world
    calls = []

    def make(rank):
        def exchange(user, view, send, recv, slot_bytes, w, stream):
            try:
                assert w == world
                if hip.hipStreamSynchronize(stream):
                    return 1
                if on_device:
                    bufs[rank] = send
                    barrier.wait()
                    for q in range(world):
                        if hip.hipMemcpy(recv + q * slot_bytes, bufs[q], slot_bytes, 3):   # device -> device
                            return 3
                    if hip.hipStreamSynchronize(None):        # (a device-to-device hipMemcpy may return before it is done: the null stream's work is, after this)
                        return 4
                    barrier.wait()
                    if rank == 0:
                        calls.append((view, slot_bytes))
                    return 0
                b = C.create_string_buffer(slot_bytes)
                if hip.hipMemcpy(b, send, slot_bytes, 2):                    # device -> host
                    return 2
                bufs[rank] = b
                barrier.wait()
                for q in range(world):
                    if hip.hipMemcpy(recv + q * slot_bytes, bufs[q], slot_bytes, 1):   # host -> device
                        return 3
                barrier.wait()
                if rank == 0:
                    calls.append((view, slot_bytes))
                return 0
            except Exception:      # noqa: BLE001  (a broken barrier etc.: fail the exchange, never hang the other ranks)
                barrier.abort()
                return 9
        return exchange
    make.abort = barrier.abort
    return make, calls


