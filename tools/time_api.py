#!/usr/bin/env python3
"""End-to-end timing of the public threshold() / detect() on HOST arrays (PCIe and host-side numpy
included), 1-degree grid by default:  python tools/time_api.py [nlat nlon nyears]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import xmhw_amd
from xmhw_amd import GridSeries, climatology_series

args = [a for a in sys.argv[1:] if not a.startswith("--")]
profile = "--profile" in sys.argv
nlat, nlon, ny = (int(v) for v in args[:3]) if len(args) >= 3 else (180, 360, 30)
t = np.arange("1991-01-01", f"{1991 + ny}-01-01", dtype="datetime64[D]")
T = t.shape[0]
rng = np.random.default_rng(1)
t0 = time.perf_counter()
# random block of 48 longitudes, tiled along lon (numpy's generator makes 0.5 GB/s; tiling copies at memory speed)
nb = min(48, nlon)
blk = (15 + 5 * np.sin(2 * np.pi * np.arange(T)[:, None, None] / 365.25)
       + rng.standard_normal((T, nlat, nb), dtype=np.float32)).astype(np.float32)
x = np.empty((T, nlat, nlon), dtype=np.float32)
for j in range(0, nlon, nb):
    w = min(nb, nlon - j)
    x[:, :, j:j + w] = blk[:, :, :w] + np.float32(0.001 * j)
del blk
x[:, : nlat // 6, :] = np.nan                         # a band of land
print(f"input {x.shape} {x.nbytes / 1e9:.2f} GB built in {time.perf_counter() - t0:.1f} s", flush=True)
g = GridSeries(x, ("time", "lat", "lon"), {"time": t, "lat": np.arange(nlat), "lon": np.arange(nlon)},
               time_encoding={"calendar": "proleptic_gregorian"})
if profile:
    import cProfile, pstats, io
    xmhw_amd.threshold(g)
    pr = cProfile.Profile(); pr.enable()
    clim = xmhw_amd.threshold(g)
    mhw = xmhw_amd.detect(g, climatology_series(clim, "thresh"), climatology_series(clim, "seas"))
    pr.disable()
    s_ = io.StringIO(); pstats.Stats(pr, stream=s_).sort_stats("tottime").print_stats(16); print(s_.getvalue()[:5000])
    sys.exit(0)
for rep in range(2):
    t0 = time.perf_counter()
    clim = xmhw_amd.threshold(g)
    t1 = time.perf_counter()
    mhw = xmhw_amd.detect(g, climatology_series(clim, "thresh"), climatology_series(clim, "seas"))
    t2 = time.perf_counter()
    ncell = int(mhw.n_cells)
    print(f"rep {rep}: threshold() {t1 - t0:.2f} s ({ncell / (t1 - t0):.3g} cells/s), detect() {t2 - t1:.2f} s "
          f"({ncell / (t2 - t1):.3g} cells/s), {mhw.n_events} events in {ncell} cells", flush=True)
if "--fused" in sys.argv:
    for rep in range(2):
        t0 = time.perf_counter()
        clim2, mhw2 = xmhw_amd.threshold_detect(g)
        t1 = time.perf_counter()
        same = np.array_equal(mhw2.table, mhw.table) and np.array_equal(clim2["thresh"], clim["thresh"], equal_nan=True)
        print(f"rep {rep}: threshold_detect() {t1 - t0:.2f} s, {mhw2.n_events} events, identical to the two calls: {same}",
              flush=True)
