"""CPU oracle for the xmhw ``threshold()`` hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm (coecms/xmhw
v0.9.3, pure Python on xarray/numpy).  It is the checker the HIP path is
compared with; it is never imported by the product package ``xmhw_amd``.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.

Where the arithmetic lives
--------------------------
The reference delegates the arithmetic to un-vendored, un-pinned third-party
code (``requirements.txt:4-7``): xarray (rolling/construct/stack/groupby/
quantile/mean/pad) and, underneath, ``numpy.quantile(method="linear")`` and
``numpy.mean``.  xarray and dask are not installed in the build image, so the
reference cannot be imported; numpy (2.2.6) IS installed and is called
directly here rather than re-derived.  The xarray semantics restated here are
anchored on the reference's call sites (cited per function) and PINNED against
the reference's own fixtures (``tests/test_oracle_golden.py``):

* ``test/testdata/test_clim_oisst_nosmooth.nc`` thresh 365/366 doys <=1e-13
  (index 59 = Feb 29 differs by construction: the fixture was produced by
  Oliver's marineHeatWaves code which uses a 2-point Feb-29 mean),
* ``test/testdata/test_clim_oisst.nc`` smoothed thresh from index 82,
* ``oisst_doy`` (bit-exact), ``tstack``, ``test_feb29`` scalar,
  ``test_runavg`` vectors, ``test_land_check`` cell counts.

Parity UNPINNED by any reference test (this restatement is the de-facto spec):
NaN inside an ocean cell, ``skipna=False``, ``coldSpells``,
``climatologyPeriod``, tstep climatology *values*, the point path, output
attrs, doy 60 / smoothed indices 44-74 against an xmhw-produced truth.

Everything is computed in float64 (inputs are converted first).  The reference
would return float32 quantiles/means for float32 input; the difference is
<=1e-7 relative, inside the 1e-6 contract.

Keep this file dumb, slow and obviously faithful.  Never optimise it; the
vectorised variant for larger parity runs lives in ``oracle_fast.py`` and is
itself checked against this file.
"""
import numpy as np


class XmhwException(Exception):
    """Mirror of xmhw/exception.py:18-19 (bare Exception subclass)."""


# --------------------------------------------------------------------------
# calendar / doy                                    xmhw/identify.py:28-134
# --------------------------------------------------------------------------

_NDAYS = {  # identify.py:104-113
    "standard": 365.25, "gregorian": 365.25, "proleptic_gregorian": 365.25,
    "all_leap": 366, "noleap": 365, "365_day": 365, "360_day": 360,
    "julian": 365.25,
}


def get_calendar(calendar):
    """identify.py:82-134.  ``calendar`` is the string the reference would find
    in ``time.encoding``/``time.attrs`` ('' if none)."""
    if calendar in ["360", "365", "366"]:           # :125-126
        calendar = f"{calendar}_day"
    elif calendar == "leap":                        # :127-128
        calendar = "standard"
    if calendar not in _NDAYS:                      # :129-131
        return 365.25
    return _NDAYS[calendar]


def _ymd(time):
    """year, month, dayofyear, is_leap for a datetime64 array (what xarray's
    ``.dt`` accessor returns for a proleptic-gregorian axis)."""
    t = np.asarray(time).astype("datetime64[D]")
    years = t.astype("datetime64[Y]")
    months = t.astype("datetime64[M]")
    year = years.astype(np.int64) + 1970
    month = (months.astype(np.int64) % 12) + 1
    dayofyear = (t - years.astype("datetime64[D]")).astype(np.int64) + 1
    leap = ((year % 4 == 0) & (year % 100 != 0)) | (year % 400 == 0)
    return year, month, dayofyear, leap


def add_doy(time, keep_tstep=False):
    """identify.py:28-79 -> int64 doy[T]."""
    year, month, dayofyear, leap = _ymd(time)
    if keep_tstep is True:
        years = np.unique(year)                     # :59
        noneyear = int(np.sum(year == years[1]))    # :60 (second year!)
        if len(year) % noneyear != 0:               # :61-66
            raise XmhwException(
                "To use original timestep as climatology base unit, "
                "timeseries has to have complete years")
        nyears = len(year) // noneyear              # :67
        return np.tile(np.arange(1, noneyear + 1), nyears).astype(np.int64)
    # :73-76
    return (dayofyear + ((~leap) & (month >= 3))).astype(np.int64)


# --------------------------------------------------------------------------
# land_check                                        xmhw/identify.py:482-529
# --------------------------------------------------------------------------

def land_check(values, dims, tdim="time", anynans=False):
    """Stack non-time dims (sorted by NAME, identify.py:520) into 'cell' and drop
    all-NaN (any-NaN if anynans) cells.

    values: ndarray with axes named by ``dims``.
    Returns (ts[T, C_ocean], keep_mask over stacked cells, stacked dim names,
    stacked shape).
    """
    dims = list(dims)
    rest = [d for d in dims if d != tdim]
    if len(rest) == 0:                              # :509-510
        raise XmhwException("Series has only time dimension use point=True option, exiting")
    for d in rest:                                  # :514-516
        if values.shape[dims.index(d)] == 0:
            raise XmhwException(f"Dimension {d} has 0 lenght, exiting")
    order = sorted(rest)                            # :520
    perm = [dims.index(tdim)] + [dims.index(d) for d in order]
    v = np.transpose(values, perm)
    sshape = v.shape[1:]
    v = v.reshape(v.shape[0], -1)
    nan = np.isnan(v)
    drop = nan.any(axis=0) if anynans else nan.all(axis=0)   # :522-525
    keep = ~drop
    if not keep.any():                              # :527-528
        raise XmhwException("All points of grid are either land or NaN")
    return v[:, keep], keep, order, sshape


# --------------------------------------------------------------------------
# window_roll / calculate_thresh / calculate_seas / feb29 / runavg
# --------------------------------------------------------------------------

def window_roll(x, doy, w):
    """identify.py:184-209 for ONE cell.

    rolling(time=2w+1, center=True).construct("wdim") gives
    win[t, j] = x[t - w + j] (NaN outside [0, T)); stack(z=("wdim", time)) is
    window-major / time-minor; the doy label of element (j, t) is doy[t] (the
    centre day's); dropna("z") removes the edge padding AND every NaN sample.
    Returns (values[z], doy_labels[z]).
    """
    x = np.asarray(x, dtype=np.float64)
    T = x.shape[0]
    vals, labs = [], []
    for j in range(2 * w + 1):           # wdim (outer)
        for t in range(T):               # time (inner)
            tt = t - w + j
            v = x[tt] if 0 <= tt < T else np.nan
            if not np.isnan(v):          # dropna("z"), :208
                vals.append(v)
                labs.append(doy[t])
    return np.array(vals, dtype=np.float64), np.array(labs, dtype=np.int64)


def _groupby_doy(vals, labs):
    """xarray groupby("doy"): groups in ascending label order, members in
    original z order."""
    uniq = np.unique(labs)
    return uniq, [vals[labs == d] for d in uniq]


def feb29(clim, doys):
    """identify.py:137-151: mean over doy in {59,60,61} of the already computed
    climatology, skipna=True."""
    sel = clim[np.isin(doys, [59, 60, 61])]
    if sel.size == 0 or np.all(np.isnan(sel)):
        return np.nan
    return np.nanmean(sel)


def calculate_thresh(vals, labs, pctile, skipna, tstep):
    """identify.py:212-242."""
    doys, groups = _groupby_doy(vals, labs)
    q = pctile / 100.0                               # :234
    f = np.nanquantile if skipna else np.quantile    # xarray's skipna switch
    th = np.array([f(g, q) for g in groups], dtype=np.float64)
    if tstep is False:                               # :237-240
        th = np.where(doys != 60, th, feb29(th, doys))
    return doys, th


def calculate_seas(vals, labs, skipna, tstep):
    """identify.py:245-270."""
    doys, groups = _groupby_doy(vals, labs)
    f = np.nanmean if skipna else np.mean
    se = np.array([f(g) for g in groups], dtype=np.float64)
    if tstep is False:                               # :265-268
        se = np.where(doys != 60, se, feb29(se, doys))
    return doys, se


def runavg(ts, w):
    """identify.py:154-181: pad(wrap) -> rolling(w, center).mean() -> dropna.
    Returns a dense array with NaN where the reference would have dropped the
    doy (quirk Q8)."""
    if w % 2 == 0:
        raise XmhwException("Running average window should be odd")
    ts = np.asarray(ts, dtype=np.float64)
    h = (w - 1) // 2
    padded = np.pad(ts, h, mode="wrap")
    out = np.empty_like(ts)
    for i in range(ts.shape[0]):
        out[i] = np.mean(padded[i:i + w])            # NaN in window -> NaN
    return out


def calc_clim(x, doy, pctile, windowHalfWidth, smoothPercentile,
              smoothPercentileWidth, tstep, skipna):
    """xmhw/xmhw.py:250-307 for ONE cell.  Returns (doys, thresh, seas)."""
    vals, labs = window_roll(x, doy, windowHalfWidth)
    doys, th = calculate_thresh(vals, labs, pctile, skipna, tstep)
    _, se = calculate_seas(vals, labs, skipna, tstep)
    if smoothPercentile:
        th = runavg(th, smoothPercentileWidth)
        se = runavg(se, smoothPercentileWidth)
    return doys, th, se


def threshold_cells(ts, doy, pctile=90, windowHalfWidth=5, smoothPercentile=True,
                    smoothPercentileWidth=31, tstep=False, skipna=False,
                    coldSpells=False):
    """The per-cell loop of xmhw/xmhw.py:184-197 on a dense (T, C) array.

    Returns (doys[D], thresh[D, C], seas[D, C]) float64; a (cell, doy) with an
    empty pool is NaN (the reference has no group there and xr.concat
    outer-joins on doy -- quirk Q8); smoothing runs on the present groups only.
    """
    if smoothPercentileWidth % 2 == 0:               # xmhw.py:103-104
        raise XmhwException("smoothPercentileWidth should be odd")
    ts = np.asarray(ts, dtype=np.float64)
    if ts.ndim == 1:
        ts = ts[:, None]
    if coldSpells:                                   # xmhw.py:153-154
        ts = -1.0 * ts
    doy = np.asarray(doy, dtype=np.int64)
    all_doys = np.unique(doy)
    D, C = all_doys.shape[0], ts.shape[1]
    thresh = np.full((D, C), np.nan)
    seas = np.full((D, C), np.nan)
    for c in range(C):
        if np.all(np.isnan(ts[:, c])):
            continue                                 # land: dropped by land_check
        # calc_clim works on the groups PRESENT for this cell: a doy whose pool
        # is empty has no group, so the per-cell series is shorter and runavg
        # rolls over it positionally (neighbours across the gap).  xr.concat
        # (xmhw.py:210-211) then outer-joins on doy -> NaN at the absent doys.
        d_c, th, se = calc_clim(ts[:, c], doy, pctile, windowHalfWidth,
                                smoothPercentile, smoothPercentileWidth, tstep, skipna)
        idx = np.searchsorted(all_doys, d_c)
        thresh[idx, c] = th
        seas[idx, c] = se
    return all_doys, thresh, seas


def threshold_grid(values, time, dims=("time", "lat", "lon"), tdim="time",
                   calendar="proleptic_gregorian", climatologyPeriod=(None, None),
                   pctile=90, windowHalfWidth=5, smoothPercentile=True,
                   smoothPercentileWidth=31, maxPadLength=None, coldSpells=False,
                   tstep=False, anynans=False, skipna=False):
    """xmhw/xmhw.py:38-247 on plain arrays.

    Returns dict(doy, thresh, seas, keep, stacked_dims, stacked_shape) with
    thresh/seas of shape (D, *stacked_shape) in sorted-dim-name order and NaN
    at land.  (The reference additionally drops a lat/lon line that is all
    land when it unstacks; callers compare on ocean cells.)
    """
    if smoothPercentileWidth % 2 == 0:
        raise XmhwException("smoothPercentileWidth should be odd")
    dims = list(dims)
    if tdim not in dims:
        raise XmhwException(f"{tdim} dimension not present")
    if maxPadLength:
        raise XmhwException("maxPadLength is not restated (quirk Q10)")
    values = np.asarray(values)
    time = np.asarray(time).astype("datetime64[D]")
    tax = dims.index(tdim)
    if all(climatologyPeriod):                       # xmhw.py:112-119 (Q7)
        y = _ymd(time)[0]
        sel = (y >= int(climatologyPeriod[0])) & (y <= int(climatologyPeriod[1]))
        values = np.compress(sel, values, axis=tax)
        time = time[sel]
    point = values.ndim == 1
    if point:
        ts, keep, sdims, sshape = values[:, None], np.array([True]), [], ()
    else:
        ts, keep, sdims, sshape = land_check(values, dims, tdim, anynans)
    if get_calendar(calendar) == 360.0:              # xmhw.py:142-144
        tstep = True
    doy = add_doy(time, keep_tstep=tstep)
    doys, th, se = threshold_cells(ts, doy, pctile, windowHalfWidth, smoothPercentile,
                                   smoothPercentileWidth, tstep, skipna, coldSpells)
    D = doys.shape[0]
    if point:
        return dict(doy=doys, thresh=th[:, 0], seas=se[:, 0], keep=keep,
                    stacked_dims=sdims, stacked_shape=sshape)
    full_th = np.full((D, keep.shape[0]), np.nan)
    full_se = np.full((D, keep.shape[0]), np.nan)
    full_th[:, keep] = th
    full_se[:, keep] = se
    return dict(doy=doys, thresh=full_th.reshape((D,) + tuple(sshape)),
                seas=full_se.reshape((D,) + tuple(sshape)), keep=keep,
                stacked_dims=sdims, stacked_shape=sshape)
