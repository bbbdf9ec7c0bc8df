"""Timing of the HBM-bound kernels around the ring kernel on one MI355X, each against its ALGORITHMIC
bytes (what the operation has to read and write once): land_mask, gather_cells, decode (float32 swap,
int16 unpack), pad_gaps, clim_finish_stream<31, 31> (what width 31 runs on: D = 366 and D = 1460), block_time.  One JSON line per kernel:

    python tools/bench_aux.py [--cells 259200] > profiles/r2_aux_kernels.jsonl
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/bench_aux.py

Inputs are generated on the device (synth_sst); T = 14,610 (40-year daily axis) unless stated.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

PEAK = 8000.0        # GB/s, MI355X HBM3E (guide)


def timed(h, fn, reps=5):
    e0, e1 = h.event_create(), h.event_create()
    fn()
    h.stream_sync(0)
    ms = []
    for _ in range(reps):
        h.event_record(e0, 0)
        fn()
        h.event_record(e1, 0)
        h.stream_sync(0)
        ms.append(h.event_elapsed_ms(e0, e1))
    h.event_destroy(e0)
    h.event_destroy(e1)
    return float(np.median(ms))


def line(name, nbytes, ms, note):
    gbs = nbytes / ms / 1e6
    print(json.dumps({"kernel": name, "algorithmic_GB": round(nbytes / 1e9, 3), "ms": round(ms, 4),
                      "GBs": round(gbs, 1), "frac_of_8TBs": round(gbs / PEAK, 4), "what": note}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=259200)
    args = ap.parse_args()
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.device as dev
    from xmhw_amd.calendar import add_doy
    h = dev.hip()
    C = args.cells
    t64 = np.arange("1982-01-01", "2022-01-01", dtype="datetime64[D]")
    T = t64.shape[0]
    doy = add_doy(t64)
    B = dev.DeviceBuffer

    ts = B(4 * T * C)
    h.synth_sst(ts.ptr, 4, T, C, C, 0, 7, 0.02, 0)
    h.stream_sync(0)

    # land_mask: reads the series once, writes C bytes
    keep = B(C)
    ms = timed(h, lambda: h.land_mask(ts.ptr, 4, T, C, C, 0, keep.ptr))
    line("land_mask<float>", 4 * T * C + C, ms, f"all-NaN test of {C} cells x {T} steps")

    # gather_cells: 5/6 of the columns survive (a land band): read + write of the survivors
    idx = np.nonzero(np.arange(C) % 6 != 0)[0].astype(np.int64)
    d_idx = B.from_array(idx)
    out = B(4 * T * idx.size)
    ms = timed(h, lambda: h.gather_cells(ts.ptr, 4, T, C, d_idx.ptr, idx.size, out.ptr, idx.size))
    line("gather_cells<float>", 2 * 4 * T * idx.size, ms, f"compaction of {idx.size} of {C} columns")
    out.free()
    d_idx.free()

    # decode: big-endian float32 -> float32; int16 packed -> float32
    out = B(4 * T * C)
    ms = timed(h, lambda: h.decode(ts.ptr, 4, 1, T, C, C, out.ptr, 4, C, False, 1.0, 0.0, False, 0.0))
    line("decode_slab<float,float,swap>", 2 * 4 * T * C, ms, "netCDF classic big-endian float32 -> native")
    raw = B(2 * T * C)
    h.memset(raw.ptr, 1, 2 * T * C)
    ms = timed(h, lambda: h.decode(raw.ptr, 2, 1, T, C, C, out.ptr, 4, C, True, 0.01, 15.0, True, -32768.0))
    line("decode_slab<int16,float,swap>", (2 + 4) * T * C, ms, "CF-packed int16 (scale, offset, fill) -> float32")
    raw.free()
    out.free()

    # pad_gaps: reads the series once (2 % NaN as short gaps), writes the filled samples
    x = ((t64.astype("datetime64[ns]") - np.datetime64("1970-01-01", "ns")) / np.timedelta64(1, "ns")).astype(np.float64)
    d_x = B.from_array(x)
    ms = timed(h, lambda: h.pad_gaps(ts.ptr, 4, T, C, C, d_x.ptr, 5 * 86400e9), reps=3)
    line("pad_gaps<float>", 4 * T * C, ms, "interpolate_na(max_gap = 5 days), series read once (already filled after the first pass)")
    d_x.free()

    # clim_finish_stream<31, 31> (the default smoothing width; D = 366 and D = 1460): two (D, C) float64 arrays in, two out
    for label, dd in (("clim_finish_stream<31, 31> (D = 366)", doy), ("clim_finish_stream<31, 31> (D = 1460)", np.tile(np.arange(1, 1461), 4))):
        plan = dev.Plan(dd, 5)
        D = plan.D
        a, b, c_, d_ = (B(8 * D * C) for _ in range(4))
        h.memset(a.ptr, 0, 8 * D * C)
        h.memset(b.ptr, 0, 8 * D * C)
        ms = timed(h, lambda: dev.clim_finish(plan, a, b, C, D == 366, True, 31, c_, d_))
        line(label, 4 * 8 * D * C, ms, "Feb-29 fix + circular 31-step running mean of thresh and seas")
        for q in (a, b, c_, d_):
            q.free()
        plan.destroy()

    # block_time: annual ts_mean / ts_max / ts_min of the series (block_average's time statistics)
    years = t64.astype("datetime64[Y]").astype(np.int64) + 1970
    bins = (years - years[0]).astype(np.int32)
    nb = int(bins[-1]) + 1
    d_bins = B.from_array(bins)
    o = B(8 * 7 * nb * C)
    try:
        ms = timed(h, lambda: h.block_time(ts.ptr, 4, T, C, C, 0, C, d_bins.ptr, nb, o.ptr, C))
        line("block_time<float>", 4 * T * C + 8 * 3 * nb * C, ms, f"{nb} annual blocks of ts_mean / ts_max / ts_min")
    finally:
        d_bins.free()
        o.free()
    keep.free()
    ts.free()
    dev.release_device_cache()


if __name__ == "__main__":
    main()
