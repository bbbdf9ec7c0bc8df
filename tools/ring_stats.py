"""Print the ring kernel's pass counters on a small synthetic run (GPU box)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from xmhw_amd._lib import hip
from xmhw_amd.device import Plan, DeviceBuffer, clim_raw
from xmhw_amd.calendar import add_doy
h = hip()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nan_frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t = np.arange("1982-01-01", "2022-01-01", dtype="datetime64[D]")
doy = add_doy(t); T = len(doy)
plan = Plan(doy, 5, nchunks=1)
ts = DeviceBuffer(4 * T * C); th = DeviceBuffer(8 * plan.D * C); se = DeviceBuffer(8 * plan.D * C)
h.synth_sst(ts.ptr, 4, T, C, C, 0, 20260103, nan_frac, 0)
h.plan_debug_stats(plan.handle, 1, False)
clim_raw(plan, ts, 4, C, 0.9, False, th, se)
st = h.plan_debug_stats(plan.handle, 1, True)
rows, cnt, ext, cold = [int(v) & 0xFFFFFFFF for v in st[:4]]      # (slots 4..15: see tools/bench_ring2.py)
print(f"wave-rows {rows}  count passes/row {cnt/rows:.3f}  extractions/row {ext/rows:.3f}  cold/row {cold/rows:.4f}")
