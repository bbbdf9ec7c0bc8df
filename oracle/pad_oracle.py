"""CPU restatement of ``ts.interpolate_na(dim=tdim, max_gap=maxPadLength)`` -- TEST INFRASTRUCTURE ONLY
(imported by tests/ only; the product path is the pad_gaps HIP kernel).

The reference calls xarray for this step (xmhw/xmhw.py:159-160 in threshold(), :409-410 in detect()).
xarray is a third-party dependency that is absent from /root/reference and from this image
(requirements.txt / conda/meta.yaml name it without a version), so its published algorithm is restated
here from xarray/core/missing.py, function by function:

  interp_na()                 method="linear", use_coordinate=True, limit=None
  get_clean_interp_index()    datetime index -> float64 nanoseconds since 1970-01-01
  _get_nan_block_lengths()    the distance, in coordinate units, between the valid samples either side
                              of each run of NaN (index[0] / index[-1] stand in at the two ends)
  func_interpolate_na()       out[nans] = NumpyInterpolator(x[~nans], y[~nans])(x[nans]), i.e. the REAL
                              numpy.interp (left = right = NaN), stored in y's dtype
  ... .where(nan_block_lengths <= max_gap)

PARITY UNPINNED for this one step: the reference's tests never set maxPadLength and xarray cannot be run
here to produce vectors; tests/test_real_xarray.py compares with the real thing wherever xarray is
importable.  numpy.interp itself (the arithmetic) is the genuine article.
"""
import numpy as np


def interp_index(time):
    """get_clean_interp_index(arr, dim, use_coordinate=True)"""
    t = np.asarray(time)
    if t.dtype.kind == "M":
        offset = np.datetime64("1970-01-01", "ns")
        return ((t.astype("datetime64[ns]") - offset) / np.timedelta64(1, "ns")).astype(np.float64)
    return t.astype(np.float64)


def nan_block_lengths(y, x):
    """_get_nan_block_lengths() for one column: ffill / diff / bfill spelled out"""
    n = y.shape[0]
    valid = ~np.isnan(y)
    pos = np.arange(n)
    last = np.maximum.accumulate(np.where(valid, pos, -1))             # ffill of the valid coordinates
    cumulative = np.where(last >= 0, x[np.maximum(last, 0)], x[0])     # .fillna(index[0])
    diff = np.full(n, np.nan)                                           # .diff(label="upper").reindex(...)
    diff[1:] = cumulative[1:] - cumulative[:-1]
    d = np.where(valid, diff, np.nan)                                   # .where(valid)
    have = ~np.isnan(d)                                                 # .bfill()
    nxt = np.where(have, pos, n)
    nxt = np.minimum.accumulate(nxt[::-1])[::-1]
    d = np.where(nxt < n, d[np.minimum(nxt, n - 1)], np.nan)
    d = np.where(~valid, d, 0.0)                                        # .where(~valid, 0)
    return np.where(np.isnan(d), x[-1] - cumulative, d)                 # .fillna(index[-1] - cumulative_nans)


def interpolate_na(y, x, max_gap=None):
    """(T, C) array, float64 abscissa x[T], max_gap in the units of x -> the filled array (y's dtype)"""
    y = np.asarray(y)
    out = y.copy()
    for c in range(y.shape[1]):
        col = y[:, c]
        nans = np.isnan(col)
        n_nans = int(nans.sum())
        if n_nans == 0 or n_nans == col.shape[0]:                       # func_interpolate_na's early return
            continue
        filled = col.copy()
        filled[nans] = np.interp(x[nans], x[~nans], col[~nans].astype(np.float64), left=np.nan, right=np.nan)
        if max_gap is not None:
            filled = np.where(nan_block_lengths(col, x) <= max_gap, filled, np.nan).astype(y.dtype)
        out[:, c] = filled
    return out
