#!/usr/bin/env python3
"""End-to-end time of threshold() on a large netCDF classic file that is memory-mapped from /dev/shm
(page cache speed: the disk is not what is being measured).
   python tools/time_ingest.py [--kind f32|i16] [--lat 720 --lon 1440 --years 1982 2021] [--keep]
The file is synthesised on the DEVICE (counter-based SST of SURVEY 8d) and written once; then
open_series() + threshold() is timed twice (the second call reuses cached device buffers).
Prints one JSON line: file GB, seconds, GB/s of file bytes and of float32-equivalent samples."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="f32", choices=["f32", "i16"])
    ap.add_argument("--lat", type=int, default=720)
    ap.add_argument("--lon", type=int, default=1440)
    ap.add_argument("--years", type=int, nargs=2, default=[1982, 2021])
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--profile", action="store_true", help="a third call under cProfile (host-side hot spots to stderr)")
    args = ap.parse_args()
    import xmhw_amd
    from xmhw_amd import ingest, netcdf3
    from xmhw_amd.device import DeviceBuffer, hip
    h = hip()
    t = np.arange(f"{args.years[0]}-01-01", f"{args.years[1] + 1}-01-01", dtype="datetime64[D]")
    T, N = t.shape[0], args.lat * args.lon
    path = os.path.join(args.dir, f"xmhw_ingest_{args.kind}_{args.lat}x{args.lon}_{T}.nc")
    isz = 4 if args.kind == "f32" else 2
    t0 = time.perf_counter()
    # header through the writer (a 1-step file), then the data written in place through a memmap
    vat = {"units": "degC"} if args.kind == "f32" else {"scale_factor": np.float32(0.01), "add_offset": np.float32(15.0),
                                                       "_FillValue": np.int16(-32768)}
    dt = ">f4" if args.kind == "f32" else ">i2"
    netcdf3.write_classic(path, {"time": T, "lat": args.lat, "lon": args.lon}, {
        "time": (("time",), (t - t[0]).astype(np.float64), {"units": f"days since {args.years[0]}-01-01", "calendar": "standard"}),
        "lat": (("lat",), np.linspace(-89.875, 89.875, args.lat).astype(np.float32), {}),
        "lon": (("lon",), np.linspace(0.125, 359.875, args.lon).astype(np.float32), {}),
        "sst": (("time", "lat", "lon"), np.zeros((1, args.lat, args.lon), dtype=dt), vat)})
    # sst is the last variable the writer laid out and only one of its T steps was written
    begin = os.path.getsize(path) - ((isz * N + 3) & ~3)
    with open(path, "r+b") as fh:
        fh.truncate(begin + isz * T * N)
    mm = np.memmap(path, dtype=dt, mode="r+", offset=begin, shape=(T, N))
    cb = max(1, (2 << 30) // (4 * T))                 # column blocks of ~2 GB: the generator is a function of (cell, t)
    buf = DeviceBuffer(4 * T * cb)
    for c0 in range(0, N, cb):
        c1 = min(N, c0 + cb)
        h.synth_sst(buf.ptr, 4, T, c1 - c0, c1 - c0, c0, 20260102, 0.0, 0)
        h.stream_sync(0)
        a = buf.to_array((T, c1 - c0), np.float32)
        land = (np.arange(c0, c1) % 6) == 0            # a sixth of the cells is land
        if args.kind == "f32":
            a[:, land] = np.nan
            mm[:, c0:c1] = a
        else:
            q = np.round((a - 15.0) / 0.01).astype(np.int16)
            q[:, land] = -32768
            mm[:, c0:c1] = q
    mm.flush()
    del mm
    buf.free()
    t_write = time.perf_counter() - t0
    fbytes = os.path.getsize(path)
    out = {"file": path, "kind": args.kind, "file_GB": fbytes / 1e9, "T": T, "cells": N, "write_s": t_write}
    for k in (1, 2, 3) if args.profile else (1, 2):
        pr = None
        if args.profile and k == 3:
            import cProfile
            pr = cProfile.Profile()
            pr.enable()
        t0 = time.perf_counter()
        temp = ingest.open_series(path, "sst")
        t_open = time.perf_counter() - t0
        ds = xmhw_amd.threshold(temp)
        dt_s = time.perf_counter() - t0
        if pr is not None:
            import io
            import pstats
            pr.disable()
            s_ = io.StringIO()
            pstats.Stats(pr, stream=s_).sort_stats("tottime").print_stats(18)
            print(s_.getvalue()[:6000], file=sys.stderr)
        out[f"open_s_{k}"] = t_open
        out[f"threshold_s_{k}"] = dt_s
        out[f"file_GBps_{k}"] = fbytes / 1e9 / dt_s
        out[f"float32_equiv_GBps_{k}"] = 4 * T * N / 1e9 / dt_s
        out["ocean_cells"] = int((~np.isnan(ds["thresh"][0])).sum())
        del ds, temp
    print(json.dumps(out))
    if not args.keep:
        os.remove(path)


if __name__ == "__main__":
    main()
