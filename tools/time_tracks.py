"""Ring kernel time for record lengths outside the 17-40 tracks of the BASELINE configs (43 and 12 years):
round-1 kernel against the auto-selected ring2 layout, bit-identity of a cell sample.  python tools/time_tracks.py"""
import sys, numpy as np
sys.path.insert(0, '.')
import xmhw_amd.device as dev
from xmhw_amd.calendar import add_doy
h = dev.hip()
for y0, y1, C in ((1982, 2024, 518400), (2010, 2021, 518400), (1940, 2024, 259200)):
    doy = add_doy(np.arange(f"{y0}-01-01", f"{y1 + 1}-01-01", dtype="datetime64[D]"))
    T = doy.shape[0]
    ts = dev.DeviceBuffer(4 * T * C)
    h.synth_sst(ts.ptr, 4, T, C, C, 0, 5, 0.0, 0)
    res = {}
    for v in (-1, None):
        plan = dev.Plan(doy, 5, ring2=v)
        th, se = dev.DeviceBuffer(8 * plan.D * C), dev.DeviceBuffer(8 * plan.D * C)
        e0, e1 = h.event_create(), h.event_create()
        dev.clim_raw(plan, ts, 4, C, 0.9, False, th, se); h.stream_sync(0)
        ms = []
        for _ in range(3):
            h.event_record(e0, 0); dev.clim_raw(plan, ts, 4, C, 0.9, False, th, se); h.event_record(e1, 0); h.stream_sync(0)
            ms.append(h.event_elapsed_ms(e0, e1))
        idx = dev.DeviceBuffer.from_array(np.arange(0, C, 997, dtype=np.int64)); n = (C + 996) // 997
        sub = dev.DeviceBuffer(8 * plan.D * n)
        h.gather_cells(th.ptr, 8, plan.D, C, idx.ptr, n, sub.ptr, n); h.stream_sync(0)
        res[v] = sub.to_array((plan.D, n), np.float64)
        print(f"{y0}-{y1} ({plan.ntracks} tracks) ring2={plan.ring2_in_use()}: {min(ms):.2f} ms for {C} cells", flush=True)
        for b in (th, se, idx, sub): b.free()
        plan.destroy()
    print("bit-identical:", np.array_equal(res[-1], res[None], equal_nan=True))
    ts.free()

# float64 input holding float32-representable samples (the narrowing path): 64,800 cells, 40 years
if "--narrow" in sys.argv:
    C = 64800
    doy = add_doy(np.arange("1982-01-01", "2022-01-01", dtype="datetime64[D]"))
    T = doy.shape[0]
    ts = dev.DeviceBuffer(4 * T * C)
    h.synth_sst(ts.ptr, 4, T, C, C, 0, 5, 0.0, 0)
    h.stream_sync(0)
    x32 = ts.to_array((T, C), np.float32)
    d64 = dev.DeviceBuffer.from_array(x32.astype(np.float64))
    out = {}
    for label, buf, isz, kw in (("float32", ts, 4, {}), ("float64 narrowed (ring2)", d64, 8, {}),
                                ("float64 narrowed (round-1 ring)", d64, 8, {"ring2": -1}),
                                ("float64 kernel", d64, 8, {"narrowing": False})):
        plan = dev.Plan(doy, 5, **kw)
        th, se = dev.DeviceBuffer(8 * plan.D * C), dev.DeviceBuffer(8 * plan.D * C)
        e0, e1 = h.event_create(), h.event_create()
        dev.clim_raw(plan, buf, isz, C, 0.9, False, th, se); h.stream_sync(0)
        ms = []
        for _ in range(3):
            h.event_record(e0, 0); dev.clim_raw(plan, buf, isz, C, 0.9, False, th, se); h.event_record(e1, 0); h.stream_sync(0)
            ms.append(h.event_elapsed_ms(e0, e1))
        out[label] = th.to_array((plan.D, C), np.float64)
        print(f"{label}: {min(ms):.2f} ms for {C} cells (narrowed: {plan.narrowed() if isz == 8 else '-'})", flush=True)
        th.free(); se.free(); plan.destroy()
    ref = out["float32"]
    print("all bit-identical:", all(np.array_equal(v, ref, equal_nan=True) for v in out.values()))
