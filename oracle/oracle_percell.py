"""Per-cell numpy restatement used as the timed CPU baseline.  TEST/BENCH
INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests import it).

It works one cell at a time exactly as the reference does (xmhw/xmhw.py:184-197:
one calc_clim per cell): window_roll (identify.py:204-208) by index arithmetic,
then per doy group ``numpy.quantile`` and ``numpy.mean`` (what xarray's
groupby().quantile/mean call, identify.py:233-235, :263), the Feb-29 step and
runavg.  It leaves out everything xarray/dask add on top (object construction,
graph building, scheduling), so it FLATTERS the reference.  Checked against the
dumb oracle in tests/test_oracle_fast.py.
"""
import numpy as np

from xmhw_oracle import XmhwException, feb29, runavg


def _pool_index(doy, w):
    T = doy.shape[0]
    order = np.argsort(doy, kind="stable")
    doys, starts = np.unique(doy[order], return_index=True)
    ends = np.append(starts[1:], T)
    off = np.arange(-w, w + 1)
    pools = []
    for s, e in zip(starts, ends):
        idx = (order[s:e][None, :] + off[:, None]).ravel()     # window-major, time-minor
        pools.append(idx[(idx >= 0) & (idx < T)])
    return doys, pools


def calc_clim_cell(x, doys, pools, q, tstep, smooth, width, skipna=False):
    x = np.asarray(x, dtype=np.float64)
    quant = np.nanquantile if skipna else np.quantile
    mean = np.nanmean if skipna else np.mean
    d_c, th, se = [], [], []
    for d, idx in zip(doys, pools):
        g = x[idx]
        g = g[~np.isnan(g)]                # dropna("z")
        if g.size == 0:
            continue                       # no group for this doy
        d_c.append(d)
        th.append(quant(g, q))
        se.append(mean(g))
    d_c, th, se = np.array(d_c), np.array(th), np.array(se)
    if th.size and tstep is False:
        th = np.where(d_c != 60, th, feb29(th, d_c))
        se = np.where(d_c != 60, se, feb29(se, d_c))
    if smooth and th.size:
        th, se = runavg(th, width), runavg(se, width)
    return d_c, th, se


def threshold_cells_percell(ts, doy, pctile=90, windowHalfWidth=5, smoothPercentile=True,
                            smoothPercentileWidth=31, tstep=False, skipna=False, coldSpells=False,
                            pools=None):
    """``pools``: a precomputed ``_pool_index(doy, windowHalfWidth)`` (the timed baseline builds it
    once per worker, outside the timed region)."""
    if smoothPercentileWidth % 2 == 0:
        raise XmhwException("smoothPercentileWidth should be odd")
    ts = np.asarray(ts)
    if ts.ndim == 1:
        ts = ts[:, None]
    doy = np.asarray(doy, dtype=np.int64)
    doys, pools = pools if pools is not None else _pool_index(doy, windowHalfWidth)
    D, C = doys.shape[0], ts.shape[1]
    thresh = np.full((D, C), np.nan)
    seas = np.full((D, C), np.nan)
    for c in range(C):
        x = ts[:, c].astype(np.float64)
        if coldSpells:
            x = -1.0 * x
        d_c, th, se = calc_clim_cell(x, doys, pools, pctile / 100.0, tstep, smoothPercentile,
                                     smoothPercentileWidth, skipna)
        idx = np.searchsorted(doys, d_c)
        thresh[idx, c] = th
        seas[idx, c] = se
    return doys, thresh, seas
