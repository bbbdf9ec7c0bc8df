"""The sharded entry points with the HIP path underneath.

 * RCCL through the C ABI (xmhw_comm_*, xmhw_gather_blocks) with one rank -- what a 1-GPU box can
   run: the device-resident (2D, block) gather, the mask all-gather and the table gathers.
 * the same with TWO RCCL ranks on two GPUs (skipped with a reason on a 1-GPU box; the driver's
   multi-GPU node runs it).
 * two ranks SHARING one GPU, host transport (tests/gloo_transport.py), real HIP stages: every rank
   masks / compacts / computes its own column block and, in detect, addresses the climatologies at
   a non-zero column offset (the k0 > 0 path of detect_grid); detect(intermediate=True) too, its per-step
   planes gathered block by block.
Results must be bit-identical to the single-process calls."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


COMMON = r'''
import os, sys
import numpy as np
root, mode, rank, world, port = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import xmhw_amd
from xmhw_amd import GridSeries, climatology_series
from xmhw_amd.sharded import threshold_sharded, detect_sharded, init_rccl

g = np.load(os.path.join(root, "tests", "golden", "oisst_2003_2004.npz"))
time = np.datetime64("2003-01-01") + g["time"].astype("timedelta64[D]")
# a wider grid than the fixture's 8 x 4 so that every rank owns ocean cells: tile it 6 times along lon
sst = np.tile(g["sst"], (1, 1, 6)) + np.linspace(0, 0.5, 24, dtype=np.float32)[None, None, :]
temp = GridSeries(sst, ("time", "lat", "lon"), {"time": time, "lat": g["lat"], "lon": np.arange(24.0)},
                  time_encoding={"calendar": "proleptic_gregorian"})
if mode == "rccl":
    tr = init_rccl(rank=rank, size=world, local_rank=rank, addr="127.0.0.1", port=port)
else:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from gloo_transport import GlooTransport
    from xmhw_amd._lib import hip
    hip().set_device(0)                                   # every rank on GPU 0
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = GlooTransport()
ds = threshold_sharded(temp, tr)
ref = xmhw_amd.threshold(temp)
th, se = climatology_series(ref, "thresh"), climatology_series(ref, "seas")
m1 = detect_sharded(temp, th, se, tr)
mi = detect_sharded(temp, th, se, tr, intermediate=True)      # the per-step planes, gathered block by block
if rank == 0:
    np.testing.assert_array_equal(ds["thresh"], ref["thresh"])
    np.testing.assert_array_equal(ds["seas"], ref["seas"])
    m0 = xmhw_amd.detect(temp, th, se)
    np.testing.assert_array_equal(m1.table, m0.table)
    np.testing.assert_array_equal(m1.offsets, m0.offsets)
    np.testing.assert_array_equal(m1.keep, m0.keep)
    assert m0.n_events > 0
    m0i, i0 = xmhw_amd.detect(temp, th, se, intermediate=True)
    m1i, i1 = mi
    np.testing.assert_array_equal(m1i.table, m0i.table)
    assert set(i1.data_vars) == set(i0.data_vars)
    for k, v in i0.data_vars.items():
        assert i1.data_vars[k].dtype == v.dtype, k
        np.testing.assert_array_equal(i1.data_vars[k], v, err_msg=k)
    print("sharded ok", mode, world, m0.n_events)
else:
    assert ds is None and m1 is None and mi is None
if mode == "rccl":
    tr.close()
else:
    dist.barrier()
    dist.destroy_process_group()
'''


def _run(tmp_path, mode, world):
    script = tmp_path / "worker.py"
    script.write_text(COMMON)
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, mode, str(r), str(world), str(port)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600) + (p.returncode,))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for so, se, rc in outs:
        assert rc == 0, so[-2000:] + se[-4000:]
    assert any("sharded ok" in so for so, _, _ in outs)


@pytest.mark.gpu
def test_sharded_entry_points_over_rccl_single_rank(tmp_path):
    _run(tmp_path, "rccl", 1)


@pytest.mark.gpu
def test_sharded_entry_points_over_rccl_two_ranks(tmp_path):
    from xmhw_amd._lib import hip
    n = hip().device_count()
    if n < 2:
        pytest.skip(f"needs 2 GPUs for two RCCL ranks, this box has {n} (RCCL refuses two ranks on one device)")
    _run(tmp_path, "rccl", 2)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_two_ranks_share_one_gpu_real_hip_stages(tmp_path, world):
    _run(tmp_path, "host", world)
