#!/usr/bin/env python3
"""float64 input that holds float32 values (a float32 archive promoted by a reader) against the same values as float32:
the narrowing instantiation of the ring kernel reads twice the bytes and converts on load.
   python tools/bench_narrow.py [--cells N]     (40 years daily, int16 'centi-degree' samples decoded to either type
                                                  on the device: every value is float32-representable)"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=259200)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import xmhw_amd.device as dev
    from xmhw_amd.calendar import add_doy
    h = dev.hip()
    doy = add_doy(np.arange("1982-01-01", "2022-01-01", dtype="datetime64[D]"))
    T, C = doy.shape[0], args.cells
    # seasonal cycle + noise in hundredths of a degree (what a packed SST archive holds): one block of 4,096 cells as
    # int16 on the device, decoded into every column block of the series (the timing does not care that blocks repeat)
    rng = np.random.default_rng(5)
    blk = min(4096, C)
    t = np.arange(T)[:, None]
    x = 1500 + rng.uniform(200, 1000, blk) * np.sin(2 * np.pi * (t - rng.uniform(0, 365, blk)) / 365.25) \
        + 100 * rng.normal(size=(T, blk))
    raw = dev.DeviceBuffer.from_array(np.ascontiguousarray(x.astype(np.int16)))
    out = {}
    res = {}
    for name, isz in (("f32", 4), ("f64 holding f32 values", 8)):
        ts = dev.DeviceBuffer(isz * T * C)
        for c0 in range(0, C, blk):
            n = min(blk, C - c0)
            h.decode(raw.ptr, 2, 0, T, n, blk, ts.ptr + isz * c0, isz, C, 1, 1.0, 0.0, 0, 0.0, 0)
        plan = dev.Plan(doy, 5, narrowing=True)
        D = plan.D
        th, se = dev.DeviceBuffer(8 * D * C), dev.DeviceBuffer(8 * D * C)
        dev.clim_raw(plan, ts, isz, C, 0.9, False, th, se)
        h.stream_sync(0)
        stats = None
        if isz == 4:            # what the band path makes of quantised samples (ties): counters of one launch
            h.plan_debug_stats(plan.handle, 1, False)
            dev.clim_raw(plan, ts, isz, C, 0.9, False, th, se)
            st = h.plan_debug_stats(plan.handle, 1, True)
            rows = max(float(st[0]), 1.0)
            b = int(st[6])
            stats = {"wave_rows_band_only": (int(st[5]) & 0xFFFFFFFF) / rows, "rebuilds_per_wave_row": (int(st[5]) >> 32) / rows,
                     "cell_fail_rate": (b >> 32) / max(b & 0xFFFFFFFF, 1), "failed_off_block": int(st[7]) & 0xFFFFFFFF,
                     "failed_band_too_big": int(st[7]) >> 32, "cell_rows_tried": b & 0xFFFFFFFF,
                     "count_passes_per_row": (int(st[1]) & 0xFFFFFFFF) / rows, "extractions_per_row": (int(st[2]) & 0xFFFFFFFF) / rows}
        e0, e1 = h.event_create(), h.event_create()
        ms = []
        for _ in range(args.reps):
            h.event_record(e0, 0)
            dev.clim_raw(plan, ts, isz, C, 0.9, False, th, se)
            h.event_record(e1, 0)
            ms.append(h.event_elapsed_ms(e0, e1))
        med = float(np.median(ms))
        bytes_per_cell = T * isz + 2 * D * 8
        out[name] = {"ms": med, "narrowed": bool(plan.narrowed()) if isz == 8 else None, "layout": plan.ring2_in_use(), "band_path": stats,
                     "algorithmic_GBs": C * bytes_per_cell / med / 1e6, "frac_of_8TBs": C * bytes_per_cell / med / 8e9}
        idx = np.unique(np.linspace(0, C - 1, 512).astype(np.int64))
        d_idx = dev.DeviceBuffer.from_array(idx)
        sub = dev.DeviceBuffer(8 * D * idx.size)
        h.gather_cells(th.ptr, 8, D, C, d_idx.ptr, idx.size, sub.ptr, idx.size)
        h.stream_sync(0)
        res[name] = sub.to_array((D, idx.size), np.float64)
        for b in (ts, th, se, sub, d_idx):
            b.free()
        plan.destroy()
    out["thresh_bit_identical"] = bool(np.array_equal(res["f32"], res["f64 holding f32 values"], equal_nan=True))
    out["cells"] = C
    print(json.dumps(out))


if __name__ == "__main__":
    main()
