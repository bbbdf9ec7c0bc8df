#!/usr/bin/env python3
"""Time the sorted-list kernel (layout 40) beside a ring layout on one BASELINE shape, device-resident synthetic input.
   python tools/bench_sorted.py [--config 0.25deg|1deg|0.25deg_nan] [--cells N] [--layouts 40 21] [--reps 5]
One JSON line per layout: ms of the whole clim_raw call (HIP events, median), bit-identity of thresh against the first
layout on a column sample, and -- in `make STATS=1` builds -- the sorted kernel's counters: walk iterations per wave-row,
steps per cell-row, flagged cell-rows, shader-clock ticks per section and wave-row."""
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
    ap.add_argument("--layouts", type=int, nargs="*", default=[40, 21])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--q", type=float, default=0.9)
    ap.add_argument("--years", type=int, nargs=2, default=None, help="first and last year instead of the preset's")
    ap.add_argument("--packed", default=None, choices=["float32", "float64"],
                    help="store the series as int16 codes (scale 0.01) and read them in place (xmhw_clim_raw_i16), decoded as ...")
    ap.add_argument("--ice-patch", type=int, default=0, help="ice cells in patches of that many cells (0: scattered)")
    ap.add_argument("--gen", type=float, nargs=3, default=None, metavar=("QUANT", "ICE_FRAC", "RHO"),
                    help="the extended generator (xmhw_synth_sst_ex_f32) instead of the SURVEY 8(d) one")
    args = ap.parse_args()
    import xmhw_amd.device as dev
    from xmhw_amd.calendar import add_doy
    h = dev.hip()
    presets = {"0.25deg": (1440 * 720, (1982, 2021), 0.0), "1deg": (360 * 180, (1991, 2020), 0.0),
               "0.25deg_nan": (1440 * 720, (1982, 2021), 0.05)}
    C, years, nan = presets[args.config]
    C = args.cells or C
    if args.years:
        years = tuple(args.years)
    doy = add_doy(np.arange(f"{years[0]}-01-01", f"{years[1] + 1}-01-01", dtype="datetime64[D]"))
    T = doy.shape[0]
    ts = dev.DeviceBuffer(4 * T * C)
    if args.gen:
        h.synth_sst_ex(ts.ptr, T, C, C, 0, 20260103, nan, args.gen[0], args.gen[1], args.gen[2], args.ice_patch, 0)
    else:
        h.synth_sst(ts.ptr, 4, T, C, C, 0, 20260103, nan, 0)
    codes = None
    if args.packed:
        codes = dev.DeviceBuffer(2 * T * C)
        h.encode_i16(ts.ptr, T, C, C, codes.ptr, C, 0.01, 0.0, -999, 0)
        h.stream_sync(0)
    sc = 0.01 if args.packed == "float64" else float(np.float32(0.01))

    def call(plan, th, se):
        if codes is not None:
            dev.clim_raw_packed(plan, codes, C, args.q, False, th, se, scale_factor=sc, add_offset=0.0, fill=-999, decoded=args.packed)
        else:
            dev.clim_raw(plan, ts, 4, C, args.q, False, th, se)
    ref = None
    idx = np.unique(np.linspace(0, C - 1, 2048).astype(np.int64))
    d_idx = dev.DeviceBuffer.from_array(idx)
    for v in args.layouts:
        plan = dev.Plan(doy, 5, nchunks=args.chunks, layout=v)
        if plan.layout_in_use() != v:
            print(json.dumps({"layout": v, "skipped": f"not instantiated for {plan.ntracks} tracks"}), flush=True)
            plan.destroy()
            continue
        D = plan.D
        th, se = dev.DeviceBuffer(8 * D * C), dev.DeviceBuffer(8 * D * C)
        e0, e1 = h.event_create(), h.event_create()
        call(plan, th, se)      # warm-up (plan upload)
        h.stream_sync(0)
        h.plan_debug_stats(plan.handle, 1, False)
        ms = []
        for _ in range(args.reps):
            h.event_record(e0, 0)
            call(plan, th, se)
            h.event_record(e1, 0)
            ms.append(h.event_elapsed_ms(e0, e1))
        st = h.plan_debug_stats(plan.handle, 1, True)
        sub = dev.DeviceBuffer(8 * D * idx.size)
        h.gather_cells(th.ptr, 8, D, C, d_idx.ptr, idx.size, sub.ptr, idx.size)
        h.stream_sync(0)
        got = sub.to_array((D, idx.size), np.float64)
        if ref is None:
            ref = got
        med = float(np.median(ms))
        out = {"layout": v, "config": args.config, "cells": C, "ms": round(med, 3), "ms_all": [round(m, 3) for m in ms],
               "frac_of_8TBs": round(C * (T * 4 + 2 * D * 8) / med / 1e6 / 8000, 4),
               "thresh_bit_identical_to_first": bool(np.array_equal(got, ref, equal_nan=True))}
        if v == 40 and h.debug_stats_available() and int(st[0]) > 0:
            rows = float(st[0])
            cellrows = float(C) * D * args.reps
            out["sorted"] = {"wave_rows": int(st[0]), "walk_iterations_per_wave_row": round(float(st[1]) / rows, 2),
                             "walk_steps_per_cell_row": round(float(st[3]) / cellrows, 3),
                             "flagged_cell_rows_frac": float(st[2]) / cellrows,
                             "register_rank_corrections_per_wave_row": round(float(st[4]) / rows, 4),
                             "extra_rounds_per_wave_row_by_keys_left_le4_le8_more": [round(float(x) / rows, 4) for x in st[5:8]],
                             "ticks_per_wave_row_push_sort_book_walk_epilogue": [round(float(x) / rows, 1) for x in st[8:13]],
                             "ticks_per_wave_row_wait_samples_convert_redo": [round(float(x) / rows, 1) for x in st[13:16]]}
        print(json.dumps(out), flush=True)
        for b in (th, se, sub):
            b.free()
        plan.destroy()
    ts.free()
    dev.release_device_cache()


if __name__ == "__main__":
    main()
