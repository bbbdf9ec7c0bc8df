"""Throughput of the pitched slab upload (hipMemcpy2D from a row-major host array) against the contiguous
one: calc_clim_grid_device on a 7.6 GB grid in 1, 2 and 8 slabs.   python tools/time_slabs.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from xmhw_amd.device import calc_clim_grid_device
from xmhw_amd.calendar import add_doy
t = np.arange("1982-01-01", "2022-01-01", dtype="datetime64[D]")
doy = add_doy(t)
T, N = t.shape[0], 129600
rng = np.random.default_rng(0)
blk = rng.standard_normal((T, 64), dtype=np.float32) + 15
x = np.tile(blk, (1, N // 64))
print(f"{x.nbytes / 1e9:.2f} GB", flush=True)
ref = None
for budget in (None, 9 << 30, 2 << 30):
    for rep in range(2):
        t0 = time.perf_counter()
        k, d, th, se = calc_clim_grid_device(x, doy, False, 90, 5, True, 31, False, max_batch_bytes=budget)
        dt = time.perf_counter() - t0
    nslab = 1 if budget is None else int(np.ceil(N / max(1, budget // (2 * T * 4 + 4 * 366 * 8))))
    print(f"budget {budget}: ~{nslab} slabs, {dt:.2f} s ({x.nbytes / dt / 1e9:.1f} GB/s end to end)", flush=True)
    if ref is None: ref = th
    assert np.array_equal(ref, th, equal_nan=True)
