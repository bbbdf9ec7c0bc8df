#!/usr/bin/env python3
"""Measurement of the detect() front-end kernel on one MI355X:
synthetic SST resident in HBM -> threshold() climatology (ring + finish) -> detect_events.
Prints one JSON line: cells/s, achieved HBM GB/s (algorithmic bytes = T*(4 in + 13 out) + D*8 per
cell) and the loop-oracle CPU baseline on a small sample.   python tools/bench_detect.py [cells]"""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
from xmhw_amd._lib import hip
from xmhw_amd.device import Plan, DeviceBuffer, clim_raw, clim_finish
from xmhw_amd.calendar import add_doy

h = hip()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 259200
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t = np.arange("1982-01-01", "2022-01-01", dtype="datetime64[D]")
doy = add_doy(t); T = len(doy)
plan = Plan(doy, 5)
D = plan.D
rows = np.searchsorted(plan.doys, doy).astype(np.int32)
ts = DeviceBuffer(4 * T * C)
h.synth_sst(ts.ptr, 4, T, C, C, 0, 20260105, 0.0, 0)
raw_t, raw_s, th, se = (DeviceBuffer(8 * D * C) for _ in range(4))
clim_raw(plan, ts, 4, C, 0.9, False, raw_t, raw_s)
clim_finish(plan, raw_t, raw_s, C, True, True, 31, th, se)
ev, st, en = (DeviceBuffer(4 * T * C) for _ in range(3))
b = DeviceBuffer(T * C)
nev = DeviceBuffer(4 * C)
h.stream_sync(0)
e0, e1 = h.event_create(), h.event_create()
ms = []
for i in range(steps + 1):
    h.event_record(e0, 0)
    h.detect_events(ts.ptr, 4, T, C, C, th.ptr, C, rows, 5, 1, 2, 0, ev.ptr, st.ptr, en.ptr, b.ptr, C, nev.ptr)
    h.event_record(e1, 0)
    if i: ms.append(h.event_elapsed_ms(e0, e1))
ms = float(np.mean(ms))
# ---- event statistics: per-cell counts (from detect) -> prefix offsets -> stats table ----
h.stream_sync(0)
counts = nev.to_array((C,), np.int32)
offsets = np.zeros(C + 1, np.int64); np.cumsum(counts, out=offsets[1:])
ntot = int(offsets[-1])
d_off = DeviceBuffer.from_array(offsets)
table = DeviceBuffer(8 * max(ntot, 1) * h.EVENT_COLUMNS)
ms_stats, ms_count = [], []
for i in range(steps + 1):
    h.event_record(e0, 0)
    h.count_events(st.ptr, T, C, C, nev.ptr)
    h.event_record(e1, 0)
    t_c = h.event_elapsed_ms(e0, e1)
    h.event_record(e0, 0)
    h.event_stats(ts.ptr, 4, T, C, C, se.ptr, th.ptr, C, rows, 0, ev.ptr, C, d_off.ptr, table.ptr)
    h.event_record(e1, 0)
    if i:
        ms_count.append(t_c); ms_stats.append(h.event_elapsed_ms(e0, e1))
ms_stats, ms_count = float(np.mean(ms_stats)), float(np.mean(ms_count))
# ---- table-only pipeline: exceedance bits -> run walk (count, fill) -> one thread per event -----
W = (T + 63) // 64
bits = DeviceBuffer(8 * W * C)
nev2 = DeviceBuffer(4 * C)
table2 = DeviceBuffer(8 * max(ntot, 1) * h.EVENT_COLUMNS)
evs = [h.event_create() for _ in range(5)]
t_bits, t_count, t_fill, t_sparse = [], [], [], []
for i in range(steps + 1):
    h.event_record(evs[0], 0)
    h.exceed_bits(ts.ptr, 4, T, C, C, th.ptr, C, D, rows, 0, bits.ptr, C)
    h.event_record(evs[1], 0)
    h.events_from_bits(bits.ptr, T, C, C, 5, 1, 2, 0, nev2.ptr, 0)
    h.event_record(evs[2], 0)
    h.events_from_bits(bits.ptr, T, C, C, 5, 1, 2, d_off.ptr, 0, table2.ptr)
    h.event_record(evs[3], 0)
    h.event_stats_sparse(ts.ptr, 4, T, C, C, se.ptr, th.ptr, C, rows, 0, ntot, table2.ptr)
    h.event_record(evs[4], 0)
    h.stream_sync(0)
    if i:
        t_bits.append(h.event_elapsed_ms(evs[0], evs[1])); t_count.append(h.event_elapsed_ms(evs[1], evs[2]))
        t_fill.append(h.event_elapsed_ms(evs[2], evs[3])); t_sparse.append(h.event_elapsed_ms(evs[3], evs[4]))
t_bits, t_count, t_fill, t_sparse = (float(np.mean(v)) for v in (t_bits, t_count, t_fill, t_sparse))
same_counts = bool(np.array_equal(nev2.to_array((C,), np.int32), counts))
same_table = bool(np.array_equal(table2.to_array((ntot, h.EVENT_COLUMNS), np.float64),
                                 table.to_array((ntot, h.EVENT_COLUMNS), np.float64), equal_nan=True))
stats_bytes_cell = T * (4 + 4) + 2 * D * 8           # ts + labels read once, seas/thresh rows once
bytes_cell = T * (4 + 13) + D * 8
# parity + CPU baseline on a sample
import detect_oracle as det
n = 64
idx = DeviceBuffer.from_array(np.arange(n, dtype=np.int64))
def sample(buf, itemsize, rows_, dtype):
    """first n columns of a resident (rows_, C) array via the gather kernel"""
    out = DeviceBuffer(itemsize * rows_ * n)
    if itemsize == 8 and dtype == np.float64:
        h.gather_cells(buf.ptr, 8, rows_, C, idx.ptr, n, out.ptr, n)
    else:
        h.gather_cells(buf.ptr, 4, rows_, C, idx.ptr, n, out.ptr, n)
    h.stream_sync(0)
    return out.to_array((rows_, n), dtype)
x = sample(ts, 4, T, np.float32)
res = {"stage": "detect front end (exceedance + mhw_filter + join_gaps)", "cells": C, "T": T,
       "ms_per_launch": ms, "cells_per_s": C / (ms * 1e-3),
       "roofline": {"bound": "hbm", "achieved": C * bytes_cell / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": C * bytes_cell / (ms * 1e-3) / 8e12, "algorithmic_bytes_per_cell": bytes_cell}}
res["event_stats"] = {"events": ntot, "events_per_cell": ntot / C, "ms_per_launch": ms_stats,
                      "count_events_ms": ms_count, "cells_per_s": C / (ms_stats * 1e-3),
                      "roofline": {"bound": "hbm", "achieved": C * stats_bytes_cell / (ms_stats * 1e-3) / 1e9,
                                   "peak": 8000.0, "unit": "GB/s", "frac": C * stats_bytes_cell / (ms_stats * 1e-3) / 8e12,
                                   "algorithmic_bytes_per_cell": stats_bytes_cell}}
tot = t_bits + t_count + t_fill + t_sparse
tb_bytes = T * 4 + 2 * D * 8 + (ntot / C) * h.EVENT_COLUMNS * 8      # series + climatologies once, table out
res["table_only_pipeline"] = {
    "ms": {"exceed_bits": t_bits, "events_from_bits_count": t_count, "events_from_bits_fill": t_fill,
           "event_stats_sparse": t_sparse, "total": tot},
    "per_step_kernels_total_ms": ms + ms_stats, "cells_per_s": C / (tot * 1e-3),
    "identical_to_per_step_kernels": same_counts and same_table,
    "roofline": {"bound": "hbm", "achieved": C * tb_bytes / (tot * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                 "frac": C * tb_bytes / (tot * 1e-3) / 8e12, "algorithmic_bytes_per_cell": tb_bytes}}
if x is not None:
    thh = sample(th, 8, D, np.float64)
    evh = sample(ev, 4, T, np.float32).view(np.int32)
    t0 = time.perf_counter()
    ok = True
    for c in range(n):
        _, s_, e_, v_ = det.detect_front(x[:, c], thh[:, c], rows, 5, True, 2)
        v_ = np.where(np.isnan(v_), -1, v_).astype(np.int32)
        ok &= bool(np.array_equal(v_, evh[:, c]))
    dt = time.perf_counter() - t0
    res["parity_events_bit_exact"] = ok
    import features_oracle as fo
    tab = table.to_array((ntot, h.EVENT_COLUMNS), np.float64)
    seh = sample(se, 8, D, np.float64)
    worst = 0.0
    for c in range(8):
        _, s_, e_, v_ = det.detect_front(x[:, c], thh[:, c], rows, 5, True, 2)
        want = fo.event_table(x[:, c].astype(np.float64), seh[rows, c], thh[rows, c], s_, e_, v_)
        got = tab[offsets[c]:offsets[c + 1]]
        with np.errstate(invalid="ignore", divide="ignore"):
            d = np.abs(got - want) / np.maximum(np.abs(want), 1e-12)
        worst = max(worst, float(np.nanmax(d))) if want.size else worst
    res["event_stats"]["max_rel_err_vs_oracle_8_cells"] = worst
    res["cpu_baseline"] = {"value": n / dt, "unit": "cells/s", "cores": 1, "kind": "port",
                           "sample": f"{n} cells, loop oracle (oracle/detect_oracle.py), one core"}
print(json.dumps(res))
