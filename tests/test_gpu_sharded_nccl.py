"""The sharded entry points over RCCL ("nccl" backend) with the HIP path underneath.  One MI355X is
what the test box has, so the group has a single rank: this exercises the device-tensor plumbing of
every collective (all_gather of masks and counts, padded gathers of results and event tables) that
the multi-rank gloo tests cover on the CPU with stand-ins."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = r'''
import os, sys
import numpy as np
root = sys.argv[1]
for p in (root, os.path.join(root, "oracle")):
    sys.path.insert(0, p)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
import xmhw_amd
from xmhw_amd import GridSeries, climatology_series
from xmhw_amd.sharded import threshold_sharded, detect_sharded
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
g = np.load(os.path.join(root, "tests", "golden", "oisst_2003_2004.npz"))
time = np.datetime64("2003-01-01") + g["time"].astype("timedelta64[D]")
temp = GridSeries(g["sst"], ("time", "lat", "lon"), {"time": time, "lat": g["lat"], "lon": g["lon"]},
                  time_encoding={"calendar": "proleptic_gregorian"})
ref = xmhw_amd.threshold(temp)
ds = threshold_sharded(temp)
np.testing.assert_array_equal(ds["thresh"], ref["thresh"])
np.testing.assert_array_equal(ds["seas"], ref["seas"])
th, se = climatology_series(ref, "thresh"), climatology_series(ref, "seas")
m0 = xmhw_amd.detect(temp, th, se)
m1 = detect_sharded(temp, th, se)
np.testing.assert_array_equal(m1.table, m0.table)
np.testing.assert_array_equal(m1.offsets, m0.offsets)
np.testing.assert_array_equal(m1.keep, m0.keep)
assert m0.n_events > 0
dist.barrier()
dist.destroy_process_group()
print("sharded nccl ok", m0.n_events)
'''


@pytest.mark.gpu
def test_sharded_entry_points_over_rccl_single_rank(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "sharded nccl ok" in out.stdout
