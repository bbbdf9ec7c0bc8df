#!/usr/bin/env python3
"""Time the ring-kernel variants on one BASELINE shape (device-resident synthetic input).
   python tools/bench_ring2.py [--config 0.25deg|1deg|0.25deg_nan|0.05deg_tstep] [--cells N] [--variants -1 0 1 2 3]
Prints one JSON line per variant: kernel ms (HIP events, median of --reps), pass statistics per
wave-row, bit-identity of thresh against variant -1 on a column sample."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="0.25deg")
    ap.add_argument("--cells", type=int, default=0)
    ap.add_argument("--variants", type=int, nargs="*", default=[-1, 0, 1, 2, 3, 4])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--years", type=int, nargs=2, default=None, help="first and last year instead of the preset's")
    args = ap.parse_args()
    import xmhw_amd.device as dev
    from xmhw_amd.calendar import add_doy
    h = dev.hip()
    presets = {"0.25deg": (1440 * 720, (1982, 2021), 0.0, False), "1deg": (360 * 180, (1991, 2020), 0.0, False),
               "0.25deg_nan": (1440 * 720, (1982, 2021), 0.05, False), "0.05deg_tstep": (810000, (2001, 2020), 0.0, True)}
    C, years, nan, tstep = presets[args.config]
    C = args.cells or C
    if args.years:
        years = tuple(args.years)
    if tstep:
        doy = np.tile(np.arange(1, 1461, dtype=np.int64), years[1] - years[0] + 1)
    else:
        doy = add_doy(np.arange(f"{years[0]}-01-01", f"{years[1] + 1}-01-01", dtype="datetime64[D]"))
    T = doy.shape[0]
    ts = dev.DeviceBuffer(4 * T * C)
    h.synth_sst(ts.ptr, 4, T, C, C, 0, 20260103, nan, 0)
    ref = None
    idx = np.unique(np.linspace(0, C - 1, 2048).astype(np.int64))
    d_idx = dev.DeviceBuffer.from_array(idx)
    for v in args.variants:
        plan = dev.Plan(doy, 5, nchunks=args.chunks, ring2=v)
        if v >= 0 and plan.ring2_in_use() != v:      # the layout is not instantiated for this record length
            print(json.dumps({"variant": v, "config": args.config, "skipped": f"not instantiated for {plan.ntracks} tracks"}), flush=True)
            plan.destroy()
            continue
        D = plan.D
        th, se = dev.DeviceBuffer(8 * D * C), dev.DeviceBuffer(8 * D * C)
        e0, e1 = h.event_create(), h.event_create()
        dev.clim_raw(plan, ts, 4, C, 0.9, False, th, se)      # warm-up (plan upload)
        h.stream_sync(0)
        h.plan_debug_stats(plan.handle, 1, False)
        ms = []
        for _ in range(args.reps):
            h.event_record(e0, 0)
            dev.clim_raw(plan, ts, 4, C, 0.9, False, th, se)
            h.event_record(e1, 0)
            ms.append(h.event_elapsed_ms(e0, e1))
        st_raw = h.plan_debug_stats(plan.handle, 1, True)
        st = st_raw.astype(np.float64)
        hist = None
        ring3 = None
        if v >= 20:        # third-generation kernel: slots 5 and 6 carry the band-path counters
            a, b = int(st_raw[5]), int(st_raw[6])
            ring3 = {"wave_rows_band_only": (a & 0xFFFFFFFF) / max(float(st_raw[0]), 1.0),
                     "rebuilds_per_wave_row": (a >> 32) / max(float(st_raw[0]), 1.0),
                     "cell_rows_tried": b & 0xFFFFFFFF, "cell_rows_failed": b >> 32,
                     "cell_fail_rate": (b >> 32) / max(b & 0xFFFFFFFF, 1),
                     "failed_off_block": int(st_raw[7]) & 0xFFFFFFFF, "failed_band_too_big": int(st_raw[7]) >> 32,
                     "hist_mismatch": int(st_raw[3]) >> 32,
                     # shader-clock ticks per wave-row: push+histogram, totals, walk, compaction, sort+pick, slow path,
                     # epilogue, rebuild
                     "ticks_per_wave_row": [round(float(x) / max(float(st_raw[0]), 1.0), 1) for x in st_raw[8:16]]}
            ring3["rebuild_asked_no_window_edge_population"] = [int(st_raw[1]) >> 32, int(st_raw[2]) >> 32, int(st_raw[4]) >> 32]
            for i in (1, 2, 3, 4):
                st[i] = float(int(st_raw[i]) & 0xFFFFFFFF)
            st[5] = st[6] = st[7] = 0.0
        if v in (0, 5, 6, 7, 8, 9, 10, 11):   # no code ring: slots 5 and 6 hold the per-cell histogram of count passes
            a, b = int(st_raw[5]), int(st_raw[6])
            hist = [a & 0xFFFFFFFF, a >> 32, b & 0xFFFFFFFF, b >> 32]
            st[5] = st[6] = 0.0
        sub = dev.DeviceBuffer(8 * D * idx.size)
        h.gather_cells(th.ptr, 8, D, C, d_idx.ptr, idx.size, sub.ptr, idx.size)
        h.stream_sync(0)
        got = sub.to_array((D, idx.size), np.float64)
        h.gather_cells(se.ptr, 8, D, C, d_idx.ptr, idx.size, sub.ptr, idx.size)
        h.stream_sync(0)
        got_se = sub.to_array((D, idx.size), np.float64)
        if ref is None:
            ref = (got, got_se)
        rows = max(st[0], 1.0)
        bytes_per_cell = T * 4 + 2 * D * 8
        med = float(np.median(ms))
        print(json.dumps({"variant": v, "config": args.config, "cells": C, "ms": med, "ms_all": [round(m, 3) for m in ms],
                          "algorithmic_GBs": C * bytes_per_cell / med / 1e6, "frac_of_8TBs": C * bytes_per_cell / med / 1e6 / 8000,
                          "count_passes_per_row": st[1] / rows, "extractions_per_row": st[2] / rows,
                          "cold_per_row": st[3] / rows, "fast_steps_per_row": st[4] / rows,
                          "probes8_per_row": st[5] / rows, "rebases_per_row": st[6] / rows,
                          # what a cell needs on its own (the wave runs the maximum over its 8 or 16 cells)
                          "count_passes_per_cell_row": st[7] / (float(C) * D * args.reps),
                          "cell_rows_needing_0_1_2_3plus_passes": None if hist is None else [round(x / max(sum(hist), 1), 4) for x in hist],
                          "ring3": ring3,
                          "thresh_bit_identical_to_first": bool(np.array_equal(got, ref[0], equal_nan=True)),
                          "seas_max_rel_diff": float(np.nanmax(np.abs(got_se - ref[1]) / np.abs(ref[1])))}), flush=True)
        for b in (th, se, sub):
            b.free()
        plan.destroy()
    ts.free()
    dev.release_device_cache()


if __name__ == "__main__":
    main()
