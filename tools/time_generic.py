"""Time the generic kernel (float32 / float64) on a synthetic grid (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from xmhw_amd._lib import hip
from xmhw_amd.device import Plan, DeviceBuffer, clim_raw
from xmhw_amd.calendar import add_doy
h = hip()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64800
t = np.arange("1991-01-01", "2021-01-01", dtype="datetime64[D]")
doy = add_doy(t); T = len(doy)
for itemsize, kernel in ((4, "generic"), (8, "generic"), (4, "ring")):
    plan = Plan(doy, 5, kernel=kernel)
    ts = DeviceBuffer(itemsize * T * C); th = DeviceBuffer(8 * plan.D * C); se = DeviceBuffer(8 * plan.D * C)
    h.synth_sst(ts.ptr, itemsize, T, C, C, 0, 20260103, 0.0, 0)
    clim_raw(plan, ts, itemsize, C, 0.9, False, th, se); h.stream_sync(0)
    t0 = time.perf_counter()
    clim_raw(plan, ts, itemsize, C, 0.9, False, th, se); h.stream_sync(0)
    dt = time.perf_counter() - t0
    print(f"{kernel} f{itemsize*8}: {C} cells T={T}: {dt*1e3:.1f} ms -> {C/dt:.3g} cells/s")
    for b in (ts, th, se): b.free()
