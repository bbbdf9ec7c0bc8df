import sys, numpy as np
sys.path.insert(0, '.')
import xmhw_amd.device as dev
from xmhw_amd.calendar import add_doy
h = dev.hip()
for y0, y1, C in ((1982, 2024, 518400), (1980, 2027, 518400)):
    doy = add_doy(np.arange(f"{y0}-01-01", f"{y1 + 1}-01-01", dtype="datetime64[D]"))
    T = doy.shape[0]
    ts = dev.DeviceBuffer(4 * T * C)
    h.synth_sst(ts.ptr, 4, T, C, C, 0, 5, 0.0, 0)
    plan = dev.Plan(doy, 5)
    th, se = dev.DeviceBuffer(8 * plan.D * C), dev.DeviceBuffer(8 * plan.D * C)
    e0, e1 = h.event_create(), h.event_create()
    dev.clim_raw(plan, ts, 4, C, 0.9, False, th, se); h.stream_sync(0)
    ms = []
    for _ in range(5):
        h.event_record(e0, 0); dev.clim_raw(plan, ts, 4, C, 0.9, False, th, se); h.event_record(e1, 0); h.stream_sync(0)
        ms.append(h.event_elapsed_ms(e0, e1))
    print(f"{y0}-{y1} ({plan.ntracks} tracks) layout {plan.layout_in_use()}: {sorted(ms)[2]:.2f} ms for {C} cells", flush=True)
    for b in (th, se, ts): b.free()
    plan.destroy()
