import cProfile, pstats, sys, os, io
sys.argv = ["time_api.py"]
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, time
import xmhw_amd
from xmhw_amd import GridSeries, climatology_series
nlat, nlon, ny = 180, 360, 30
t = np.arange("1991-01-01", f"{1991 + ny}-01-01", dtype="datetime64[D]")
T = t.shape[0]
rng = np.random.default_rng(1)
x = (15 + 5 * np.sin(2 * np.pi * np.arange(T)[:, None, None] / 365.25) + rng.standard_normal((T, nlat, nlon), dtype=np.float32)).astype(np.float32)
x[:, : nlat // 6, :] = np.nan
g = GridSeries(x, ("time", "lat", "lon"), {"time": t, "lat": np.arange(nlat), "lon": np.arange(nlon)}, time_encoding={"calendar": "proleptic_gregorian"})
clim = xmhw_amd.threshold(g)
pr = cProfile.Profile(); pr.enable()
clim = xmhw_amd.threshold(g)
mhw = xmhw_amd.detect(g, climatology_series(clim, "thresh"), climatology_series(clim, "seas"))
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
