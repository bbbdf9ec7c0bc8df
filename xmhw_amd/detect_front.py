"""detect() front end on the GPU: threshold re-expansion by doy, exceedance, event filter and
gap joining for all cells at once (define_events() front part, mhw_filter(), join_gaps():
xmhw/identify.py:366-372, :415-479, :273-325).  The reference does this per cell in pandas
(one dask task per cell, xmhw/xmhw.py:440-454).

This is the first consumer of threshold()'s output; event
statistics (mhw_df / mhw_features) are not part of it.
"""
import numpy as np

from ._lib import hip
from .device import DeviceBuffer
from .exception import XmhwException


def _nan_where_negative(a):
    out = a.astype(np.float64)
    out[a < 0] = np.nan
    return out


def mhw_filter_cells(ts, thresh, doy, doys, minDuration=5, joinGaps=True, maxGap=2, coldSpells=False):
    """Event filter for a dense (T, C) series.

    ts      (T, C) float32/float64 temperature (cells = land_check()'s stacked ocean cells)
    thresh  (D, C) float64 climatological threshold from threshold() on the same cells
    doy     (T,)   doy label of every step (add_doy)
    doys    (D,)   the labels of thresh's rows (clim["doy"])
    Returns dict(bthresh bool (T, C); start, end, events float64 (T, C) with NaN as in the
    reference's per-cell DataFrame of mhw_filter()).
    """
    ts = np.asarray(ts)
    if ts.dtype not in (np.float32, np.float64):
        ts = ts.astype(np.float64)
    ts = np.ascontiguousarray(ts)
    thresh = np.ascontiguousarray(thresh, dtype=np.float64)
    if ts.ndim != 2 or thresh.ndim != 2 or ts.shape[1] != thresh.shape[1]:
        raise XmhwException("ts must be (T, C) and thresh (D, C) on the same cells")
    doy = np.asarray(doy)
    doys = np.asarray(doys)
    T, C = ts.shape
    if doy.shape[0] != T or doys.shape[0] != thresh.shape[0]:
        raise XmhwException("doy must have length T and doys length D")
    rows = np.searchsorted(doys, doy)
    if np.any(rows >= doys.shape[0]) or np.any(doys[np.minimum(rows, doys.shape[0] - 1)] != doy):
        # th.sel(doy=ts.doy) raises KeyError in the reference for a label without climatology
        raise XmhwException("a time step's doy label has no row in the climatology")
    h = hip()
    bufs = []
    try:
        d_ts = DeviceBuffer.from_array(ts); bufs.append(d_ts)
        d_th = DeviceBuffer.from_array(thresh); bufs.append(d_th)
        d_ev, d_st, d_en = (DeviceBuffer(4 * T * C) for _ in range(3))
        d_b = DeviceBuffer(T * C)
        bufs += [d_ev, d_st, d_en, d_b]
        try:
            h.detect_events(d_ts.ptr, ts.dtype.itemsize, T, C, C, d_th.ptr, C, rows.astype(np.int32),
                            int(minDuration), int(bool(joinGaps)), int(maxGap), int(bool(coldSpells)),
                            d_ev.ptr, d_st.ptr, d_en.ptr, d_b.ptr, C)
        except h.InvalidArgument as e:
            raise XmhwException(str(e)) from e
        ev = d_ev.to_array((T, C), np.int32)
        st = d_st.to_array((T, C), np.int32)
        en = d_en.to_array((T, C), np.int32)
        b = d_b.to_array((T, C), np.uint8).astype(bool)
    finally:
        for x in bufs:
            x.free()
    return dict(bthresh=b, start=_nan_where_negative(st), end=_nan_where_negative(en),
                events=_nan_where_negative(ev))
