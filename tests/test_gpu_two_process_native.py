"""The NATIVE sharded chain (l3d_shard_chain_run: the path `bench.py --gpus N` takes first) in a real world of two PROCESSES.
There is one GPU on the test box, so both ranks share it and the per-view exchange -- RCCL's all-gather over xGMI on a real
node -- is done by a Python callback: slot to the host, torch.distributed all_gather (gloo), gathered block back to the
device.  Everything else is the production path: two independent processes, the enqueue loop, the stage-1 thread, the
committing rank's bookkeeping thread, the verdict every rank reads out of the gathered slot headers.  The committing rank
must reproduce the unsharded run bit for bit; with slots that are too small both ranks must agree on growing them."""
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes as C, hashlib, os, pickle, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from line3d_amd.pipeline import Line3D, load_scene
from line3d_amd.synth import make_scene

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
V, S, N, slot_records = (int(x) for x in sys.argv[3:7])
device_commit = len(sys.argv) > 7 and sys.argv[7] == "device"       # every rank builds the products on its device, nobody commits on the host
sc = make_scene(V, S, N, seed=20271)
l = Line3D("", matchingNeighbors=N)
l.keep_view_matches(True)
load_scene(l, sc)
l.prepare()
hip = C.CDLL("libamdhip64.so")
calls = [0]

def exchange(user, view, send_slot, recv_block, slot_bytes, w, stream):
    """the all-gather of one view's slots: stream-ordered on the library's stream like the RCCL call it stands in for"""
    try:
        if hip.hipStreamSynchronize(C.c_void_p(stream)) != 0:
            return 1
        mine = np.empty(slot_bytes, np.uint8)
        if hip.hipMemcpy(mine.ctypes.data_as(C.c_void_p), C.c_void_p(send_slot), C.c_size_t(slot_bytes), C.c_int(2)) != 0:
            return 1
        outs = [torch.empty(slot_bytes, dtype=torch.uint8) for _ in range(w)]
        dist.all_gather(outs, torch.from_numpy(mine))
        block = torch.cat(outs).numpy()
        if hip.hipMemcpy(C.c_void_p(recv_block), block.ctypes.data_as(C.c_void_p), C.c_size_t(slot_bytes * w), C.c_int(1)) != 0:
            return 1
        calls[0] += 1
        return 0
    except Exception as e:      # noqa: BLE001
        print("exchange failed:", e, file=sys.stderr)
        return 1

err = None
try:
    l.shard_run(rank, world, slot_records, exchange, None, commit=("device" if device_commit else rank == 0))
except Exception as e:      # noqa: BLE001
    err = str(e)
h = hashlib.sha256()
kept = 0
mine = (rank == 0 or device_commit) and err is None
if mine:
    for v in sc.views:
        m, med = l.view_matches(v["id"])
        h.update(m.tobytes()); h.update(np.float32(med).tobytes())
        kept += len(m)
    l.finish(False)
with open(sys.argv[2] + ".%d" % rank, "wb") as f:
    pickle.dump(dict(err=err, digest=h.hexdigest(), kept=kept, calls=calls[0], lines=len(l.getResult()) if mine else 0), f)
l.close()
dist.destroy_process_group()
'''


def _run_world2(V, S, N, slot_records, port, mode="host"):
    with tempfile.TemporaryDirectory() as td:
        script = os.path.join(td, "worker.py")
        open(script, "w").write(WORKER)
        out = os.path.join(td, "out")
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), script, ROOT, out, str(V), str(S), str(N), str(slot_records), mode]
        p = subprocess.run(cmd, env=env, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert p.returncode == 0, p.stdout.decode()[-3000:]
        return [pickle.load(open(out + ".%d" % r, "rb")) for r in range(2)]


def _unsharded(V, S, N):
    import hashlib
    from line3d_amd.pipeline import Line3D, load_scene
    from line3d_amd.synth import make_scene
    sc = make_scene(V, S, N, seed=20271)
    l = Line3D("", matchingNeighbors=N)
    l.keep_view_matches(True)
    load_scene(l, sc)
    l.compute3Dmodel(False)
    h = hashlib.sha256()
    kept = 0
    for v in sc.views:
        m, med = l.view_matches(v["id"])
        h.update(m.tobytes()); h.update(np.float32(med).tobytes())
        kept += len(m)
    lines = len(l.getResult())
    l.close()
    return h.hexdigest(), kept, lines


def test_native_sharded_run_in_two_processes_equals_the_unsharded_run():
    V, S, N = 14, 600, 8
    ref, kept, lines = _unsharded(V, S, N)
    assert kept > 20000 and lines > 50
    res = _run_world2(V, S, N, 6000, 29641)
    assert res[0]["err"] is None and res[1]["err"] is None, (res[0]["err"], res[1]["err"])
    assert res[0]["calls"] == res[1]["calls"] > 0                    # one exchange per verified view on every rank
    assert res[0]["digest"] == ref and res[0]["kept"] == kept and res[0]["lines"] == lines


def test_native_sharded_run_in_two_processes_grows_slots_on_every_rank():
    """slots far too small: every rank reads the same verdict out of the gathered headers and reopens with more room"""
    V, S, N = 14, 600, 8
    ref, kept, _lines = _unsharded(V, S, N)
    res = _run_world2(V, S, N, 64, 29643)
    assert res[0]["err"] is None and res[1]["err"] is None, (res[0]["err"], res[1]["err"])
    assert res[0]["calls"] == res[1]["calls"] and res[0]["calls"] > 2 * (V - 1) - 2      # (at least two attempts)
    assert res[0]["digest"] == ref and res[0]["kept"] == kept


def test_native_sharded_run_in_two_processes_commits_on_every_device():
    """commit="device" in both processes: no rank hands kept lists to the host, each builds matchViews' products on its device from the
    gathered slots and finishes compute3Dmodel there -- both reproduce the unsharded run (kept lists, medians, number of lines)."""
    V, S, N = 14, 600, 8
    ref, kept, lines = _unsharded(V, S, N)
    res = _run_world2(V, S, N, 6000, 29645, mode="device")
    for r in range(2):
        assert res[r]["err"] is None, res[r]["err"]
        assert res[r]["digest"] == ref and res[r]["kept"] == kept and res[r]["lines"] == lines, r
    assert res[0]["calls"] == res[1]["calls"] > 0
