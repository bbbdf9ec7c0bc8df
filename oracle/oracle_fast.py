"""Vectorised CPU oracle for larger parity runs.  TEST INFRASTRUCTURE ONLY.

Same semantics as ``xmhw_oracle.threshold_cells`` (the dumb per-cell
restatement of xmhw/xmhw.py:184-197, 250-307 and identify.py:137-270) but
batched over cells so that a few thousand cells finish in seconds.  It is
checked against the dumb oracle in ``tests/test_oracle_fast.py``; the dumb
oracle in turn is pinned to the reference's fixtures.

Arithmetic: the pooled quantile follows numpy's ``method="linear"`` exactly
(``numpy/lib/_function_base_impl.py`` ``_quantile``/``_lerp``, numpy 2.2.6):
``vi=(n-1)q; lo=floor(vi); g=vi-lo; r=a+(b-a)g; if g>=0.5: r=b-(b-a)(1-g)``.
The pooled mean is sum/n in float64 (numpy's pairwise ``mean`` differs in the
last bits only).
"""
import numpy as np

from xmhw_oracle import XmhwException, runavg, feb29


def pool_index(doy, w):
    """For each distinct doy (ascending) the time indices of its pool
    {t+k : doy[t]==d, |k|<=w, 0<=t+k<T} (identify.py:204-208), with repeats."""
    doy = np.asarray(doy, dtype=np.int64)
    T = doy.shape[0]
    doys = np.unique(doy)
    pools = []
    for d in doys:
        centres = np.nonzero(doy == d)[0]
        idx = (centres[:, None] + np.arange(-w, w + 1)[None, :]).ravel()
        pools.append(idx[(idx >= 0) & (idx < T)])
    return doys, pools


def raw_clim(ts, doy, q, w):
    """Unsmoothed pooled quantile + mean for all cells: (doys, th[D,C], se[D,C])."""
    ts = np.asarray(ts, dtype=np.float64)
    doys, pools = pool_index(doy, w)
    D, C = doys.shape[0], ts.shape[1]
    th = np.full((D, C), np.nan)
    se = np.full((D, C), np.nan)
    cols = np.arange(C)
    for i, idx in enumerate(pools):
        p = np.sort(ts[idx, :], axis=0)             # NaN sorts last
        n = np.sum(~np.isnan(p), axis=0)
        ok = n > 0
        nn = np.where(ok, n, 1)
        vi = (nn - 1) * q
        lo = np.floor(vi).astype(np.int64)
        g = vi - lo
        hi = np.minimum(lo + 1, nn - 1)
        a = p[lo, cols]
        b = p[hi, cols]
        d = b - a
        r = a + d * g
        r2 = b - d * (1 - g)
        r = np.where(g >= 0.5, r2, r)
        th[i] = np.where(ok, r, np.nan)
        s = np.nansum(p, axis=0)
        se[i] = np.where(ok, s / nn, np.nan)
    return doys, th, se


def finish_cell(doys, col, tstep, smooth, width):
    """Feb-29 fix + runavg on the groups PRESENT for one cell (positional)."""
    present = ~np.isnan(col)
    d_c = doys[present]
    v = col[present]
    if v.size and tstep is False:
        v = np.where(d_c != 60, v, feb29(v, d_c))
    if smooth and v.size:
        v = runavg(v, width)
    out = np.full(col.shape, np.nan)
    out[present] = v
    return out


def threshold_cells_fast(ts, doy, pctile=90, windowHalfWidth=5, smoothPercentile=True,
                         smoothPercentileWidth=31, tstep=False, skipna=False,
                         coldSpells=False):
    if smoothPercentileWidth % 2 == 0:
        raise XmhwException("smoothPercentileWidth should be odd")
    ts = np.asarray(ts, dtype=np.float64)
    if ts.ndim == 1:
        ts = ts[:, None]
    if coldSpells:
        ts = -1.0 * ts
    doys, th, se = raw_clim(ts, doy, pctile / 100.0, windowHalfWidth)
    D, C = th.shape
    full = ~np.isnan(th).any(axis=0)
    # cells with every group present: vectorised finish
    if full.any():
        for arr in (th, se):
            v = arr[:, full]
            if tstep is False and (doys == 60).any():
                sel = np.isin(doys, [59, 60, 61])
                v[doys == 60] = np.mean(v[sel], axis=0)
            if smoothPercentile:
                h = (smoothPercentileWidth - 1) // 2
                padded = np.pad(v, ((h, h), (0, 0)), mode="wrap")
                out = np.empty_like(v)
                for i in range(D):
                    out[i] = np.mean(padded[i:i + smoothPercentileWidth], axis=0)
                v = out
            arr[:, full] = v
    for c in np.nonzero(~full)[0]:
        th[:, c] = finish_cell(doys, th[:, c], tstep, smoothPercentile, smoothPercentileWidth)
        se[:, c] = finish_cell(doys, se[:, c], tstep, smoothPercentile, smoothPercentileWidth)
    return doys, th, se
