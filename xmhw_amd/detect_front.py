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


EVENT_COLUMNS = ["event", "index_start", "index_end", "time_start", "time_end", "time_peak", "intensity_max",
                 "intensity_mean", "intensity_cumulative", "severity_max", "severity_mean", "severity_cumulative",
                 "severity_var", "intensity_mean_relThresh", "intensity_cumulative_relThresh",
                 "intensity_mean_abs", "intensity_cumulative_abs", "duration_moderate", "duration_strong",
                 "duration_severe", "duration_extreme", "index_peak", "intensity_var", "intensity_max_relThresh",
                 "intensity_max_abs", "intensity_var_relThresh", "intensity_var_abs", "category", "duration",
                 "rate_onset", "rate_decline"]


def mhw_features_cells(ts, seas, thresh, doy, doys, minDuration=5, joinGaps=True, maxGap=2, coldSpells=False):
    """Event detection + per-event statistics for a dense (T, C) series on the GPU:
    define_events() (xmhw/identify.py:329-412) without the xarray/pandas packaging, i.e.
    mhw_filter() + mhw_df() + mhw_features() (xmhw/features.py:22-315) for all cells.

    seas, thresh: (D, C) climatologies on the same cells; doy (T,), doys (D,) as in
    mhw_filter_cells().  Returns (table, offsets): table (n_events, 31) float64 with the
    columns EVENT_COLUMNS (time stamps as positions along the time axis), events of cell c in
    rows offsets[c]:offsets[c+1] in time order.
    """
    ts = np.asarray(ts)
    if ts.dtype not in (np.float32, np.float64):
        ts = ts.astype(np.float64)
    ts = np.ascontiguousarray(ts)
    thresh = np.ascontiguousarray(thresh, dtype=np.float64)
    seas = np.ascontiguousarray(seas, dtype=np.float64)
    if ts.ndim != 2 or thresh.shape != seas.shape or thresh.ndim != 2 or ts.shape[1] != thresh.shape[1]:
        raise XmhwException("ts must be (T, C) and seas/thresh (D, C) on the same cells")
    doy, doys = np.asarray(doy), np.asarray(doys)
    T, C = ts.shape
    if doy.shape[0] != T or doys.shape[0] != thresh.shape[0]:
        raise XmhwException("doy must have length T and doys length D")
    rows = np.searchsorted(doys, doy)
    if np.any(rows >= doys.shape[0]) or np.any(doys[np.minimum(rows, doys.shape[0] - 1)] != doy):
        raise XmhwException("a time step's doy label has no row in the climatology")
    rows = rows.astype(np.int32)
    h = hip()
    bufs = []
    try:
        d_ts = DeviceBuffer.from_array(ts)
        d_th = DeviceBuffer.from_array(thresh)
        d_se = DeviceBuffer.from_array(seas)
        d_ev, d_st, d_en = (DeviceBuffer(4 * T * C) for _ in range(3))
        d_n = DeviceBuffer(4 * C)
        bufs += [d_ts, d_th, d_se, d_ev, d_st, d_en, d_n]
        isz = ts.dtype.itemsize
        neg = int(bool(coldSpells))
        try:
            h.detect_events(d_ts.ptr, isz, T, C, C, d_th.ptr, C, rows, int(minDuration), int(bool(joinGaps)),
                            int(maxGap), neg, d_ev.ptr, d_st.ptr, d_en.ptr, 0, C, d_n.ptr)
        except h.InvalidArgument as e:
            raise XmhwException(str(e)) from e
        h.stream_sync(0)
        counts = d_n.to_array((C,), np.int32)
        offsets = np.zeros(C + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        ntot = int(offsets[-1])
        d_off = DeviceBuffer.from_array(offsets)
        d_tab = DeviceBuffer(8 * max(ntot, 1) * h.EVENT_COLUMNS)
        bufs += [d_off, d_tab]
        h.event_stats(d_ts.ptr, isz, T, C, C, d_se.ptr, d_th.ptr, C, rows, neg, d_ev.ptr, C, d_off.ptr, d_tab.ptr)
        table = d_tab.to_array((ntot, h.EVENT_COLUMNS), np.float64) if ntot else np.zeros((0, h.EVENT_COLUMNS))
    finally:
        for b in bufs:
            b.free()
    return table, offsets
