"""detect() front end on the GPU: threshold re-expansion by doy, exceedance, event filter and
gap joining for all cells at once (define_events() front part, mhw_filter(), join_gaps():
xmhw/identify.py:366-372, :415-479, :273-325).  The reference does this per cell in pandas
(one dask task per cell, xmhw/xmhw.py:440-454).

These are the consumers of threshold()'s output: the event
filter (mhw_filter_cells) and the whole per-cell define_events() (detect_cells: filter + per-event
statistics + the optional per-step `intermediate` columns).  xmhw_amd.detect.detect() is the
reference-shaped entry point on top of detect_cells().
"""
import numpy as np

from ._lib import hip
from .device import DeviceBuffer, native_float
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
    ts, _, thresh, rows = _check_inputs(ts, thresh, thresh, doy, doys)
    T, C = ts.shape
    h = hip()
    bufs = []
    try:
        d_ts = DeviceBuffer.from_array(ts); bufs.append(d_ts)
        d_th = DeviceBuffer.from_array(thresh); bufs.append(d_th)
        d_ev, d_st, d_en = (DeviceBuffer(4 * T * C) for _ in range(3))
        d_b = DeviceBuffer(T * C)
        bufs += [d_ev, d_st, d_en, d_b]
        try:
            h.detect_events(d_ts.ptr, ts.dtype.itemsize, T, C, C, d_th.ptr, C, rows,
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


INTERMEDIATE_F64 = ["seas", "thresh", "relSeas", "relThresh", "relThreshNorm", "severity", "cats", "mabs"]
INTERMEDIATE_U8 = ["duration_moderate", "duration_strong", "duration_severe", "duration_extreme"]


def _check_inputs(ts, seas, thresh, doy, doys):
    ts = np.ascontiguousarray(native_float(ts))
    thresh = np.ascontiguousarray(thresh, dtype=np.float64)
    seas = np.ascontiguousarray(seas, dtype=np.float64)
    if ts.ndim != 2 or thresh.shape != seas.shape or thresh.ndim != 2 or ts.shape[1] != thresh.shape[1]:
        raise XmhwException("ts must be (T, C) and seas/thresh (D, C) on the same cells")
    doy, doys = np.asarray(doy), np.asarray(doys)
    if doy.shape[0] != ts.shape[0] or doys.shape[0] != thresh.shape[0]:
        raise XmhwException("doy must have length T and doys length D")
    order = np.argsort(doys, kind="stable")
    rows = np.searchsorted(doys, doy, sorter=order)
    ok = rows < doys.shape[0]
    rows = order[np.minimum(rows, doys.shape[0] - 1)]
    if not ok.all() or np.any(doys[rows] != doy):
        # th.sel(doy=ts.doy) raises KeyError in the reference for a label without climatology
        raise XmhwException("a time step's doy label has no row in the climatology")
    return ts, seas, thresh, rows.astype(np.int32)


def _table_only_device(h, d_ts, isz, se_ptr, th_ptr, ldc, D, rows, T, n, neg, minDuration, joinGaps, maxGap):
    """Event table of n cells whose dense (T, n) series and climatologies (device pointers to the
    first of the n columns, leading dimension ldc) are already on the device, without per-step
    outputs: exceedance bits -> run walk (count, device prefix sum, fill) -> one thread per event
    (csrc/kernels_events.hip)."""
    bufs = []
    try:
        W = (T + 63) // 64
        d_bits = DeviceBuffer(8 * W * n); bufs.append(d_bits)
        d_n = DeviceBuffer(4 * n); bufs.append(d_n)
        try:
            h.exceed_bits(d_ts.ptr, isz, T, n, n, th_ptr, ldc, D, rows, neg, d_bits.ptr, n)
            h.events_from_bits(d_bits.ptr, T, n, n, int(minDuration), int(bool(joinGaps)), int(maxGap), 0, d_n.ptr, 0)
        except h.InvalidArgument as e:
            raise XmhwException(str(e)) from e
        # table offsets = exclusive prefix sum of the counts, on the device; the host only learns the
        # total (8 bytes) to size the table, and takes the counts along for its own bookkeeping
        d_off = DeviceBuffer(8 * (n + 1)); bufs.append(d_off)
        h.offsets_from_counts(d_n.ptr, n, d_off.ptr)
        total = np.empty(1, dtype=np.int64)
        h.memcpy_d2h(total, d_off.ptr + 8 * n)          # synchronises the stream
        ntot = int(total[0])
        counts = d_n.to_array((n,), np.int32)
        if ntot == 0:
            return np.zeros((0, h.EVENT_COLUMNS)), counts
        d_tab = DeviceBuffer(8 * ntot * h.EVENT_COLUMNS); bufs.append(d_tab)
        h.events_from_bits(d_bits.ptr, T, n, n, int(minDuration), int(bool(joinGaps)), int(maxGap), d_off.ptr, 0,
                           d_tab.ptr)
        h.event_stats_sparse(d_ts.ptr, isz, T, n, n, se_ptr, th_ptr, ldc, rows, neg, ntot, d_tab.ptr)
        return d_tab.to_array((ntot, h.EVENT_COLUMNS), np.float64), counts
    finally:
        for b in bufs:
            b.free()


def _table_only_batch(h, ts, seas, thresh, rows, T, n, isz, neg, minDuration, joinGaps, maxGap, pad=None):
    bufs = []
    try:
        d_ts = DeviceBuffer.from_array(ts); bufs.append(d_ts)
        if pad is not None:
            pad.apply(d_ts.ptr, isz, T, n)
        d_se = DeviceBuffer.from_array(seas); bufs.append(d_se)
        d_th = DeviceBuffer.from_array(thresh); bufs.append(d_th)
        return _table_only_device(h, d_ts, isz, d_se.ptr, d_th.ptr, n, thresh.shape[0], rows, T, n, neg, minDuration,
                                  joinGaps, maxGap)
    finally:
        for b in bufs:
            b.free()


def detect_cells(ts, seas, thresh, doy, doys, minDuration=5, joinGaps=True, maxGap=2, coldSpells=False,
                 intermediate=False, max_batch_bytes=64 << 30, per_step_kernels=False, pad=None):
    """define_events() (xmhw/identify.py:329-412) for all cells of a dense (T, C) series on the
    GPU: th.sel(doy=ts.doy) + exceedance + mhw_filter() + mhw_df() + mhw_features()
    (xmhw/features.py:22-315), without the xarray/pandas packaging.

    ts (T, C) float32/float64; seas, thresh (D, C) climatologies on the same cells; doy (T,)
    label of every step, doys (D,) labels of the climatology rows.  With coldSpells the series
    is negated first (xmhw/xmhw.py:411-412; the climatologies are expected to come from
    threshold(coldSpells=True), as in the reference).
    Returns dict(table (n_events, 31) float64 with columns EVENT_COLUMNS, time stamps as
    positions; offsets (C+1,): events of cell c are table[offsets[c]:offsets[c+1]] in time
    order; inter: None or dict of (T, C) arrays with the per-step columns of mhw_df()).
    Cells are processed in batches so that the device working set stays below max_batch_bytes.
    Without `intermediate` no per-step array is produced on the device either (bit-packed
    exceedances, one thread per event); per_step_kernels=True forces the per-step kernels
    (detect_events + event_stats) that `intermediate` needs — both give the same table.
    ``pad`` (padding.PadSpec): maxPadLength's interpolate_na on the device copy of every batch
    (xmhw/xmhw.py:409-410); the `ts` column of `intermediate` is the interpolated series.
    """
    ts, seas, thresh, rows = _check_inputs(ts, seas, thresh, doy, doys)
    T, C = ts.shape
    D = thresh.shape[0]
    h = hip()
    isz = ts.dtype.itemsize
    neg = int(bool(coldSpells))
    per_cell = T * (isz + 12) + 2 * D * 8 + (T * (8 * len(INTERMEDIATE_F64) + len(INTERMEDIATE_U8) + 1) if intermediate else 0)
    batch = int(max(1, min(C, max_batch_bytes // max(per_cell, 1))))
    tables, counts_all = [], []
    inter = None
    if intermediate:
        inter = {k: np.empty((T, C), dtype=np.float64) for k in INTERMEDIATE_F64 + ["events"]}
        inter.update({k: np.empty((T, C), dtype=bool) for k in INTERMEDIATE_U8 + ["bthresh"]})
        inter["ts"] = -ts if coldSpells else ts.copy()
    for c0 in range(0, C, batch):
        c1 = min(C, c0 + batch)
        n = c1 - c0
        if not intermediate and not per_step_kernels:
            tab, counts = _table_only_batch(h, np.ascontiguousarray(ts[:, c0:c1]), np.ascontiguousarray(seas[:, c0:c1]),
                                            np.ascontiguousarray(thresh[:, c0:c1]), rows, T, n, isz, neg,
                                            minDuration, joinGaps, maxGap, pad=pad)
            tables.append(tab)
            counts_all.append(counts)
            continue
        bufs = []
        try:
            d_ts = DeviceBuffer.from_array(np.ascontiguousarray(ts[:, c0:c1])); bufs.append(d_ts)
            if pad is not None:
                pad.apply(d_ts.ptr, isz, T, n)
                if intermediate:
                    filled = d_ts.to_array((T, n), ts.dtype)
                    inter["ts"][:, c0:c1] = -filled if coldSpells else filled
            d_th = DeviceBuffer.from_array(np.ascontiguousarray(thresh[:, c0:c1])); bufs.append(d_th)
            d_se = DeviceBuffer.from_array(np.ascontiguousarray(seas[:, c0:c1])); bufs.append(d_se)
            d_ev, d_st, d_en = (DeviceBuffer(4 * T * n) for _ in range(3))
            d_n = DeviceBuffer(4 * n)
            bufs += [d_ev, d_st, d_en, d_n]
            d_b = None
            if intermediate:
                d_b = DeviceBuffer(T * n); bufs.append(d_b)
            try:
                h.detect_events(d_ts.ptr, isz, T, n, n, d_th.ptr, n, rows, int(minDuration), int(bool(joinGaps)),
                                int(maxGap), neg, d_ev.ptr, d_st.ptr, d_en.ptr, d_b.ptr if d_b else 0, n, d_n.ptr)
            except h.InvalidArgument as e:
                raise XmhwException(str(e)) from e
            h.stream_sync(0)
            counts = d_n.to_array((n,), np.int32)
            offs = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(counts, out=offs[1:])
            ntot = int(offs[-1])
            d_off = DeviceBuffer.from_array(offs); bufs.append(d_off)
            d_tab = DeviceBuffer(8 * max(ntot, 1) * h.EVENT_COLUMNS); bufs.append(d_tab)
            h.event_stats(d_ts.ptr, isz, T, n, n, d_se.ptr, d_th.ptr, n, rows, neg, d_ev.ptr, n, d_off.ptr, d_tab.ptr)
            tables.append(d_tab.to_array((ntot, h.EVENT_COLUMNS), np.float64) if ntot
                          else np.zeros((0, h.EVENT_COLUMNS)))
            counts_all.append(counts)
            if intermediate:
                d_out = DeviceBuffer(8 * len(INTERMEDIATE_F64) * T * n); bufs.append(d_out)
                d_dur = DeviceBuffer(len(INTERMEDIATE_U8) * T * n); bufs.append(d_dur)
                h.event_intermediate(d_ts.ptr, isz, T, n, n, d_se.ptr, d_th.ptr, n, rows, neg, d_ev.ptr, n,
                                     d_out.ptr, n, d_dur.ptr)
                out = d_out.to_array((len(INTERMEDIATE_F64), T, n), np.float64)
                dur = d_dur.to_array((len(INTERMEDIATE_U8), T, n), np.uint8)
                for k, name in enumerate(INTERMEDIATE_F64):
                    inter[name][:, c0:c1] = out[k]
                for k, name in enumerate(INTERMEDIATE_U8):
                    inter[name][:, c0:c1] = dur[k] != 0
                inter["events"][:, c0:c1] = _nan_where_negative(d_ev.to_array((T, n), np.int32))
                inter["bthresh"][:, c0:c1] = d_b.to_array((T, n), np.uint8) != 0
        finally:
            for b in bufs:
                b.free()
    counts = np.concatenate(counts_all) if counts_all else np.zeros(0, np.int32)
    offsets = np.zeros(C + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    table = np.concatenate(tables, axis=0) if tables else np.zeros((0, len(EVENT_COLUMNS)))
    return dict(table=table, offsets=offsets, inter=inter)


def detect_grid(stacked, anynans, seas, thresh, doy, doys, minDuration=5, joinGaps=True, maxGap=2, coldSpells=False,
                intermediate=False, max_batch_bytes=None, clim_stacked=False, columns=None, exchange=None,
                resident=None, pad=None):
    """detect_cells() for an UNCOMPACTED stacked host series (T, N): land_check()'s mask and
    compaction run on the device (device.compact_columns), slab by slab.  The climatologies are
    either already compacted (D, C) arrays, or - clim_stacked=True - uncompacted (D, N) arrays whose
    own land masks and compaction then happen on the device as well.  Cells of the series and of
    the climatologies pair up by POSITION among the survivors (xmhw/xmhw.py:398-402, 437-443).
    Returns detect_cells()'s dict plus keep[N].
    Sharded runs: ``columns=(c0, c1)`` restricts the series to a rank's block of columns; the block is
    masked and compacted first, then ``exchange(n_kept)`` must return (offset of this block's first
    survivor among all survivors, total number of survivors) - the climatology columns are taken at
    that offset - and the result covers the block only (keep has c1 - c0 entries, no error for an
    all-land block).  ``resident`` (device.ResidentSeries filled by calc_clim_grid_device for the same host
    array and mask rule): its compacted device slabs are used instead of a second upload."""
    import time as _time
    from .device import ResidentSeries, _grid_batch, _trace, compact_columns, decode_through_device, device_itemsize, is_packed
    _t_all = _time.perf_counter()
    rkey = ResidentSeries.key_of(stacked, anynans) if resident is not None else None
    T, N = stacked.shape
    block = None
    if intermediate and columns is not None:
        # sharded per-step path: only this rank's block of columns is read (and decoded)
        b0, b1 = int(columns[0]), int(columns[1])
        block = decode_through_device(stacked, b0, b1) if is_packed(stacked) else \
            np.ascontiguousarray(native_float(np.asarray(stacked)[:, b0:b1]))
    elif is_packed(stacked):
        if intermediate:
            stacked = decode_through_device(stacked)       # the per-step path compacts on the host anyway
    else:
        stacked = np.ascontiguousarray(native_float(stacked))

    def host_compact(a):
        nan = np.isnan(a)
        k = ~(nan.any(axis=0) if anynans else nan.all(axis=0))
        if not k.any():
            raise XmhwException("All points of grid are either land or NaN")
        return np.ascontiguousarray(a[:, k]), k

    def host_path(seas_c, thresh_c):
        # positional pairing needs the global compact order: compact the series on the host
        ts_c, keep = host_compact(stacked)
        if ts_c.shape[1] != thresh_c.shape[1] or seas_c.shape[1] != thresh_c.shape[1]:
            raise XmhwException(f"temp, th and se do not have the same ocean cells: {ts_c.shape[1]}, "
                                f"{thresh_c.shape[1]}, {seas_c.shape[1]}")
        r = detect_cells(ts_c, seas_c, thresh_c, doy, doys, minDuration, joinGaps, maxGap, coldSpells, intermediate,
                         max_batch_bytes if max_batch_bytes is not None else 64 << 30, pad=pad)
        r["keep"] = keep
        return r

    def host_block(seas_c, thresh_c):
        # sharded: this rank's block of columns only; its survivors pair up with the climatology columns at the
        # offset the exchange returns (an all-land block is not an error here, the grid as a whole decides)
        sub = block
        nan = np.isnan(sub)
        keep = ~(nan.any(axis=0) if anynans else nan.all(axis=0))
        n = int(keep.sum())
        k0, total = exchange(n)
        if total != thresh_c.shape[1] or seas_c.shape[1] != thresh_c.shape[1]:
            raise XmhwException(f"temp, th and se do not have the same ocean cells: {total}, "
                                f"{thresh_c.shape[1]}, {seas_c.shape[1]}")
        if n == 0:
            return dict(table=np.zeros((0, len(EVENT_COLUMNS))), offsets=np.zeros(1, dtype=np.int64), inter=None,
                        keep=keep)
        r = detect_cells(np.ascontiguousarray(sub[:, keep]), np.ascontiguousarray(seas_c[:, k0:k0 + n]),
                         np.ascontiguousarray(thresh_c[:, k0:k0 + n]), doy, doys, minDuration, joinGaps, maxGap,
                         coldSpells, True, max_batch_bytes if max_batch_bytes is not None else 64 << 30, pad=pad)
        r["keep"] = keep
        return r

    if intermediate:
        # the per-step columns come back to the host anyway: compact there and take the per-step kernels
        path = host_path if columns is None else host_block
        if clim_stacked:
            return path(host_compact(np.asarray(seas, dtype=np.float64))[0],
                        host_compact(np.asarray(thresh, dtype=np.float64))[0])
        return path(np.asarray(seas, dtype=np.float64), np.asarray(thresh, dtype=np.float64))
    def rows_as_they_are(a):
        # threshold() hands its climatologies over as row-pitched views when an all-land band was cut
        # from the grid (landmask.compress_axis): the pitched upload takes them as they are; only other
        # layouts / dtypes are copied (2 x 2.5 GB on the host for a global grid otherwise)
        a = np.asarray(a)
        if a.ndim == 2 and a.dtype == np.float64 and a.dtype.isnative and (a.shape[1] <= 1 or a.strides[1] == 8) \
                and a.strides[0] >= 8 * a.shape[1]:
            return a
        return np.ascontiguousarray(a, dtype=np.float64)

    seas = rows_as_they_are(seas)
    thresh = rows_as_they_are(thresh)
    if seas.ndim != 2 or thresh.ndim != 2 or seas.shape[0] != thresh.shape[0]:
        raise XmhwException("seas and thresh must be (D, cells) arrays")
    D = thresh.shape[0]
    sample_dtype = stacked.decoded_dtype if is_packed(stacked) else stacked.dtype
    _, _, _, rows = _check_inputs(np.zeros((T, 1), dtype=sample_dtype), seas[:, :1], thresh[:, :1], doy, doys)
    h = hip()
    isz = device_itemsize(stacked)
    neg = int(bool(coldSpells))
    clim_bufs = []
    keeps, tables, counts_all = [], [], []
    k0 = 0
    try:
        # the climatologies are small next to the series: whole on the device, compacted there if needed
        if clim_stacked:
            d_th, keep_th = compact_columns(thresh, 0, thresh.shape[1], anynans)
            clim_bufs += [d_th] if d_th is not None else []
            d_se, keep_se = compact_columns(seas, 0, seas.shape[1], anynans)
            clim_bufs += [d_se] if d_se is not None else []
            C, Cse = int(keep_th.sum()), int(keep_se.sum())
            if C == 0 or Cse == 0:
                raise XmhwException("All points of grid are either land or NaN")
        else:
            d_th = DeviceBuffer.from_array(thresh); clim_bufs.append(d_th)
            d_se = DeviceBuffer.from_array(seas); clim_bufs.append(d_se)
            C, Cse = thresh.shape[1], seas.shape[1]
        if C != Cse:
            raise XmhwException(f"th and se do not have the same ocean cells: {C}, {Cse}")
        cb = _grid_batch(stacked, max_batch_bytes, per_cell_extra=6 * D * 8 + T // 8 + 64)
        c0, c1 = (0, N) if columns is None else (int(columns[0]), int(columns[1]))
        slabs = [(lo, min(c1, lo + cb)) for lo in range(c0, c1, cb)]
        reuse = resident is not None and columns is None and resident.matches(rkey, c0, c1)
        if reuse:
            slabs = [b for b, _, _, _ in resident.slabs]
        held = []                                   # sharded: compacted slabs wait for the offset exchange
        try:
            if columns is not None:
                for lo, hi in slabs:
                    d_ts, keep = compact_columns(stacked, lo, hi, anynans)
                    held.append((d_ts, keep))
                k0, total = exchange(int(sum(int(k.sum()) for _, k in held)))
                if total != C:
                    raise XmhwException(f"temp has {total} ocean cells, th and se have {C}")
            first = k0
            for i, (lo, hi) in enumerate(slabs):
                if columns is not None:
                    d_ts, keep = held[i]
                    held[i] = (None, keep)
                elif reuse:
                    _, d_ts, keep, _ = resident.slabs[i]
                else:
                    _tl = _time.perf_counter()
                    d_ts, keep = compact_columns(stacked, lo, hi, anynans)
                    _trace(f"detect: upload + mask + compact [{lo},{hi})", _tl)
                keeps.append(keep)
                n = int(keep.sum())
                if d_ts is None:
                    continue
                try:
                    if pad is not None:
                        pad.apply(d_ts.ptr, isz, T, n)      # (a retained slab is already interpolated: a no-op then)
                    if k0 + n > C:
                        raise XmhwException(f"temp has more ocean cells than th and se ({C})")
                    _tl = _time.perf_counter()
                    tab, counts = _table_only_device(h, d_ts, isz, d_se.ptr + 8 * k0, d_th.ptr + 8 * k0, C, D, rows, T,
                                                     n, neg, minDuration, joinGaps, maxGap)
                    _trace(f"detect: bits + runs + event table ({tab.shape[0]} events)", _tl)
                finally:
                    if not reuse:
                        d_ts.free()
                tables.append(tab)
                counts_all.append(counts)
                k0 += n
        finally:
            for d_ts, _ in held:
                if d_ts is not None:
                    d_ts.free()
    finally:
        for b in clim_bufs:
            b.free()
    _trace("detect_grid: all slabs", _t_all)
    keep = np.concatenate(keeps) if keeps else np.zeros(0, dtype=bool)
    if columns is None:
        if not keep.any():
            raise XmhwException("All points of grid are either land or NaN")
        if k0 != C:
            raise XmhwException(f"temp has {k0} ocean cells, th and se have {C}")
    k0 -= first
    offsets = np.zeros(k0 + 1, dtype=np.int64)
    if counts_all:
        np.cumsum(np.concatenate(counts_all), out=offsets[1:])
    table = np.concatenate(tables, axis=0) if tables else np.zeros((0, len(EVENT_COLUMNS)))
    return dict(table=table, offsets=offsets, inter=None, keep=keep)


def mhw_features_cells(ts, seas, thresh, doy, doys, minDuration=5, joinGaps=True, maxGap=2, coldSpells=False):
    """Event table only: (table, offsets) of detect_cells()."""
    r = detect_cells(ts, seas, thresh, doy, doys, minDuration, joinGaps, maxGap, coldSpells)
    return r["table"], r["offsets"]
