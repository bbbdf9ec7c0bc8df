"""The device half of tests/test_gpu_configs.py (BASELINE.json configs[1..4] at their own shapes,
single-GPU share for the 8-GPU ones), without the oracle comparison: the thing to put under
``rocprofv3 --kernel-trace --stats`` to see which kernels each config runs and for how long.
(The parity test itself fans the oracle over a process pool, which does not belong under the
profiler.)

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/trace_configs.py
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

SEED0 = 20260101


def daily_doy(y0, y1):
    t = np.arange(f"{y0}-01-01", f"{y1 + 1}-01-01", dtype="datetime64[D]")
    from xmhw_amd.calendar import add_doy
    return add_doy(t)


def run(dev, index, C, doy, nan_frac, tstep, width, reps):
    h = dev.hip()
    T = int(doy.shape[0])
    plan = dev.Plan(doy, 5)
    D = plan.D
    ts = dev.DeviceBuffer(4 * T * C)
    raw = [dev.DeviceBuffer(8 * D * C) for _ in range(2)]
    out = [dev.DeviceBuffer(8 * D * C) for _ in range(2)]
    try:
        h.synth_sst(ts.ptr, 4, T, C, C, 0, SEED0 + index, nan_frac, 0)
        h.stream_sync(0)
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            dev.clim_raw(plan, ts, 4, C, 0.9, False, raw[0], raw[1])
            dev.clim_finish(plan, raw[0], raw[1], C, not tstep, True, width, out[0], out[1])
            h.stream_sync(0)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return dict(config=index, cells=C, T=T, D=D, kernel=plan.kernel,
                    ring2=plan.ring2_in_use(), seconds=best, cells_per_s=C / best)
    finally:
        for b in [ts] + raw + out:
            b.free()
        plan.destroy()
        dev.release_device_cache()


def main():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.device as dev
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    d30, d40 = daily_doy(1991, 2020), daily_doy(1982, 2021)
    six = np.tile(np.arange(1, 1461, dtype=np.int64), 20)
    for args in [(1, 360 * 180, d30, 0.0, False, 31), (2, 1440 * 720, d40, 0.0, False, 31),
                 (3, 1440 * 720 // 8, d40, 0.05, False, 31), (4, 810000, six, 0.0, True, 31)]:
        print(json.dumps(run(dev, *args, reps)), flush=True)


if __name__ == "__main__":
    main()
