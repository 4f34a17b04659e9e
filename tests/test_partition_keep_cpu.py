"""l3d_partition_keep_views (host logic of l3d_shard_chain_partition; no GPU): which chain views a rank of the partitioned segment-sharded run
retires, as a function of the schedule alone.  Checked against a plain restatement and through the properties the partition rests on
(DESIGN.md section 6 iv): every view within 2 x reach of the rank's block is kept; the views the early-return quirk couples across the scene
(cudawrapper.cu:877-878: a view with nothing left to match returns its existing matches under LOCAL camera numbers, line3D.cc:861-865 files
them read as view ids) are kept by EVERY rank -- the early-return view, its sources, the views those numbers name; the blocks cover the chain."""
import ctypes as C

import numpy as np


class ChainView(C.Structure):          # include/line3d_amd.h: l3d_chain_view
    _fields_ = [("view_id", C.c_uint32), ("src_segs", C.c_void_p), ("S_src", C.c_int32), ("RtKinv_src", C.c_void_p), ("C_src", C.c_void_p),
                ("tgt_segs", C.c_void_p), ("n_tgt", C.c_int32), ("offsets", C.c_void_p), ("N", C.c_int32),
                ("F", C.c_void_p), ("RtKinv", C.c_void_p), ("centers", C.c_void_p), ("P", C.c_void_p),
                ("to_be_matched", C.c_void_p), ("n_tbm", C.c_int32), ("local2global", C.c_void_p),
                ("source_cam", C.c_void_p), ("source_index", C.c_void_p), ("n_sources", C.c_int32),
                ("sigma_p", C.c_float), ("sigma_a", C.c_float), ("spatial_k", C.c_float)]


def _struct_size_matches_header(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "cv.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "line3d_amd.h"\nint main(void) { printf("%zu %zu %zu %zu\\n", sizeof(l3d_chain_view), '
                   'offsetof(l3d_chain_view, n_tbm), offsetof(l3d_chain_view, source_index), offsetof(l3d_chain_view, spatial_k)); return 0; }\n')
    exe = str(tmp_path / "cv")
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(root, "include"), str(src), "-o", exe])
    return [int(x) for x in subprocess.run([exe], capture_output=True, text=True).stdout.split()]


def test_chain_view_mirror_matches_the_header(tmp_path):
    assert _struct_size_matches_header(tmp_path) == [C.sizeof(ChainView), ChainView.n_tbm.offset, ChainView.source_index.offset, ChainView.spatial_k.offset]


def _schedule(ids, neighbours):
    """The static schedule of matchViews (line3D.cc:620-648, 698-730) in chain order = ascending ids: a view still has to match the neighbours
    that were not processed before it; an already processed neighbour that matched it is a source."""
    order = sorted(ids)
    pos = {v: k for k, v in enumerate(order)}
    views = []
    for k, v in enumerate(order):
        nb = neighbours[v]
        tbm = [q for q, n in enumerate(nb) if pos[n] > k]
        src = [(q, pos[n]) for q, n in enumerate(nb) if pos[n] < k and v in neighbours[n]]
        views.append(dict(id=v, l2g=np.array(nb, np.uint32), n_tbm=len(tbm), src_cam=np.array([q for q, _ in src], np.int32), src_idx=np.array([p for _, p in src], np.int32)))
    return views


def _keep(lib, views, b0, b1):
    arr = (ChainView * len(views))()
    for k, v in enumerate(views):
        arr[k].view_id = v["id"]; arr[k].N = len(v["l2g"]); arr[k].n_tbm = v["n_tbm"]
        arr[k].local2global = v["l2g"].ctypes.data; arr[k].n_sources = len(v["src_cam"])
        arr[k].source_cam = v["src_cam"].ctypes.data; arr[k].source_index = v["src_idx"].ctypes.data
    keep = np.zeros(len(views), np.uint8)
    reach = C.c_int(0)
    rc = lib.l3d_partition_keep_views(arr, C.c_int(len(views)), C.c_int(b0), C.c_int(b1), keep.ctypes.data_as(C.c_void_p), C.byref(reach))
    assert rc == 0
    return keep.astype(bool), reach.value


def _restated(views, b0, b1):
    n = len(views)
    pos = {v["id"]: k for k, v in enumerate(views)}
    reach = max([1] + [abs(pos[int(g)] - k) for k, v in enumerate(views) for g in v["l2g"] if int(g) in pos])
    keep = np.zeros(n, bool)
    keep[max(0, b0 - 2 * reach):min(n, b1 + 2 * reach)] = True
    for k, v in enumerate(views):
        if v["n_tbm"] == 0 and len(v["src_cam"]):
            keep[k] = True
            keep[v["src_idx"]] = True
            for q in v["src_cam"]:
                if int(q) in pos:
                    keep[pos[int(q)]] = True
    return keep, reach


def test_keep_set_of_a_helix_schedule():
    from line3d_amd import capi
    lib = capi.load_library()
    V, N, W = 96, 8, 4
    ids = list(range(V))
    nb = {v: [u for u in range(v - N // 2, v + N // 2 + 1) if u != v and 0 <= u < V] for v in ids}
    views = _schedule(ids, nb)
    assert views[-1]["n_tbm"] == 0 and len(views[-1]["src_cam"]) == N // 2            # the last view: the early return
    covered = np.zeros(V, int)
    for r in range(W):
        b0, b1 = V * r // W, V * (r + 1) // W
        keep, reach = _keep(lib, views, b0, b1)
        exp, reach_exp = _restated(views, b0, b1)
        assert reach == reach_exp == N // 2
        assert np.array_equal(keep, exp)
        assert keep[max(0, b0 - 2 * reach):min(V, b1 + 2 * reach)].all()
        # the quirk's views on every rank: the last view, its sources (the N/2 views in front of it), and the views its LOCAL numbers 0..N/2-1 name
        assert keep[V - 1] and keep[V - 1 - N // 2:V - 1].all() and keep[:N // 2].all()
        # ... and nothing else outside the block's surroundings
        outside = np.ones(V, bool)
        outside[max(0, b0 - 2 * reach):min(V, b1 + 2 * reach)] = False
        outside[V - 1 - N // 2:] = False
        outside[:N // 2] = False
        assert not keep[outside].any()
        covered[b0:b1] += 1
    assert (covered == 1).all()


def test_keep_set_of_scattered_neighbourhoods():
    """ids in no order, ragged, non-mutual neighbourhoods (what findVisualNeighbors produces from shared world points, line3D.cc:476-549)"""
    from line3d_amd import capi
    lib = capi.load_library()
    rng = np.random.default_rng(7)
    V = 60
    ids = [int(x) for x in rng.choice(5000, V, replace=False)]
    nb = {v: [int(u) for u in rng.choice([u for u in ids if u != v], int(rng.integers(2, 9)), replace=False)] for v in ids}
    views = _schedule(ids, nb)
    for W in (2, 3, 5):
        for r in range(W):
            b0, b1 = V * r // W, V * (r + 1) // W
            keep, reach = _keep(lib, views, b0, b1)
            exp, reach_exp = _restated(views, b0, b1)
            assert reach == reach_exp and np.array_equal(keep, exp)


def test_keep_set_rejects_bad_ranges():
    from line3d_amd import capi
    lib = capi.load_library()
    views = _schedule([0, 1, 2], {0: [1], 1: [0, 2], 2: [1]})
    arr = (ChainView * 3)()
    keep = np.zeros(3, np.uint8)
    assert lib.l3d_partition_keep_views(arr, C.c_int(3), C.c_int(2), C.c_int(1), keep.ctypes.data_as(C.c_void_p), None) != 0
    assert lib.l3d_partition_keep_views(arr, C.c_int(3), C.c_int(0), C.c_int(4), keep.ctypes.data_as(C.c_void_p), None) != 0
    assert lib.l3d_partition_keep_views(None, C.c_int(3), C.c_int(0), C.c_int(3), keep.ctypes.data_as(C.c_void_p), None) != 0
