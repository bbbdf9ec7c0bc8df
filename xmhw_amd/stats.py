"""block_average(): drop-in for xmhw.stats.block_average (xmhw/stats.py:27-183) on the compact event
table of detect() -- statistics of the MHW properties per block of years and grid cell.

The reference loops over the cells, converts each one's events to a DataFrame and runs
``groupby(pd.cut(years, bins, right=False)).agg(...)`` under dask (call_groupby :285-319, agg_mhw
:322-364, agg_ts / agg_cats :372-428).  Here one kernel reduces the event table (one thread per cell,
a running accumulator flushed when the year bin changes: a segmented reduction keyed by (cell,
year bin)) and one streams the series for the time-axis statistics; see csrc/kernels_stats.hip.

Differences, all deliberate and documented in DESIGN.md: the ``years`` coordinate holds the first
year of each block (the reference leaves pandas Interval objects there); ``split=True`` is accepted
and, as upstream (split_event() is a stub that returns its input, stats.py:439-443), changes nothing;
``removeMissing`` is validated as upstream and otherwise unused there too.
"""
import numpy as np

from . import calendar as cal
from .detect import EventDataset, InterDataset, _alive_axes, _compress_grid
from .device import DeviceBuffer, native_float
from .exception import XmhwException
from ._lib import hip

MHW_STATS = ["ecount", "duration", "intensity_max", "intensity_max_max", "intensity_mean", "intensity_cumulative",
             "total_icum", "intensity_mean_relThresh", "intensity_cumulative_relThresh", "severity_mean",
             "severity_cumulative", "intensity_mean_abs", "intensity_cumulative_abs", "rate_onset", "rate_decline"]
TS_STATS = ["ts_mean", "ts_max", "ts_min"]
CAT_STATS = ["moderate_days", "strong_days", "severe_days", "extreme_days"]


class BlockDataset:
    """What block_average() returns: ``data[name]`` has dims ``dims`` = ("years", *spatial dims); land
    cells are NaN; ``coords["years"]`` = first year of every block, ``year_bins`` = the bin edges."""

    def __init__(self, data, dims, coords, year_bins):
        self.data_vars, self.dims, self.coords, self.year_bins = data, tuple(dims), coords, year_bins

    def __getitem__(self, k):
        return self.data_vars[k]

    def to_xarray(self):
        import xarray as xr
        return xr.Dataset({k: (self.dims, v) for k, v in self.data_vars.items()},
                          coords={k: (k, v) for k, v in self.coords.items()})


def block_bins(period, blockLength):
    """stats.py:130: range(period[0], period[1] + blockLength + 1, blockLength)"""
    return np.arange(int(period[0]), int(period[1]) + int(blockLength) + 1, int(blockLength), dtype=np.int64)


def _bin_of_t(years, edges):
    b = np.searchsorted(edges, years, side="right") - 1          # pd.cut(..., right=False)
    b[(years < edges[0]) | (years >= edges[-1])] = -1
    return b.astype(np.int32)


def block_stats_device(table, offsets, years_of_t, edges, mtime="time_start", ts=None, cats=None):
    """The device stage on compact arrays: table (n_events, 31), offsets (C+1,), the calendar year of
    every time step, bin edges; optional ts (T, C) [+ cats (T, C)] for the time-axis statistics.
    Returns {name: (nbins, C)}."""
    h = hip()
    table = np.ascontiguousarray(table, dtype=np.float64)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    C = offsets.shape[0] - 1
    nb = edges.shape[0] - 1
    T = years_of_t.shape[0]
    if mtime not in ("time_start", "time_peak", "time_end"):
        raise XmhwException(f"mtime should be one of time_start, time_peak, time_end, got {mtime}")
    col = EventDataset.columns.index(mtime)
    out = {}
    if C == 0 or nb <= 0:
        return {k: np.zeros((max(nb, 0), C)) for k in MHW_STATS}
    bufs = []
    try:
        d_bin = DeviceBuffer.from_array(_bin_of_t(np.asarray(years_of_t, dtype=np.int64), edges)); bufs.append(d_bin)
        d_tab = DeviceBuffer.from_array(table if table.size else np.zeros((1, len(EventDataset.columns)))); bufs.append(d_tab)
        d_off = DeviceBuffer.from_array(offsets); bufs.append(d_off)
        d_out = DeviceBuffer(8 * len(MHW_STATS) * nb * C); bufs.append(d_out)
        h.block_events(d_tab.ptr, d_off.ptr, C, d_bin.ptr, T, nb, col, d_out.ptr, C)
        h.stream_sync(0)
        ev = d_out.to_array((len(MHW_STATS), nb, C), np.float64)
        out.update({k: ev[i] for i, k in enumerate(MHW_STATS)})
        if ts is not None:
            ts = np.ascontiguousarray(native_float(ts))
            names = TS_STATS + (CAT_STATS if cats is not None else [])
            d_ts = DeviceBuffer.from_array(ts); bufs.append(d_ts)
            d_cat = None
            if cats is not None:
                d_cat = DeviceBuffer.from_array(np.ascontiguousarray(cats, dtype=np.float64)); bufs.append(d_cat)
            d_o2 = DeviceBuffer(8 * len(names) * nb * C); bufs.append(d_o2)
            h.block_time(d_ts.ptr, ts.dtype.itemsize, T, C, C, d_cat.ptr if d_cat else 0, C, d_bin.ptr, nb, d_o2.ptr, C)
            h.stream_sync(0)
            tt = d_o2.to_array((len(names), nb, C), np.float64)
            out.update({k: tt[i] for i, k in enumerate(names)})
            if cats is not None:
                out["total_days"] = sum(out[k] for k in CAT_STATS)                # stats.py:306-312
        return out
    finally:
        for b in bufs:
            b.free()


def block_average(mhw, dstime=None, period=None, blockLength=1, mtime="time_start", removeMissing=False, split=False,
                  _compute=None):
    """Calculate statistics like averages, mean and maximum on blocks of years.

    Same arguments and exceptions as ``xmhw.stats.block_average`` (xmhw/stats.py:27-35, :112-121).
    ``mhw``: the EventDataset returned by detect().  ``dstime``: None, or the series the events were
    detected on -- a GridSeries (ts statistics are added) or the InterDataset of
    ``detect(..., intermediate=True)`` (ts statistics and the moderate / strong / severe / extreme day
    counts, from its ``ts`` and ``cats``).  ``period`` = [first year, last year], required without
    ``dstime`` (with it the period is the series').  Returns a BlockDataset.
    """
    if not isinstance(mhw, EventDataset):
        raise XmhwException("block_average expects the EventDataset returned by xmhw_amd.detect()")
    compute = _compute or block_stats_device
    sw_temp = dstime is not None
    ts = cats = None
    years = cal.years_of(mhw.time)
    if sw_temp:
        if isinstance(dstime, InterDataset):
            tsg, catg = np.asarray(dstime["ts"]), np.asarray(dstime["cats"]) if "cats" in dstime.data_vars else None
        else:
            tsg, catg = np.asarray(dstime.values), None
            if tuple(dstime.dims)[0] != "time" and len(dstime.dims) > 1:
                order = [list(dstime.dims).index("time")] + [list(dstime.dims).index(d) for d in mhw.sdims]
                tsg = np.transpose(tsg, order)
        if tsg.shape[0] != years.shape[0]:
            raise XmhwException("dstime and mhw do not share the time axis")
        if mhw.point:
            ts = tsg.reshape(tsg.shape[0], 1)
            cats = None if catg is None else catg.reshape(catg.shape[0], 1)
        else:
            # the series either covers the full grid the events were detected on, or (InterDataset, a
            # ClimDataset-shaped input) the grid without its all-land lines
            alive = _alive_axes(mhw.keep, mhw.sshape)
            cshape = tuple(int(m.sum()) for m in alive)
            if tuple(tsg.shape[1:]) == tuple(mhw.sshape):
                flat = mhw.cell_index
            elif tuple(tsg.shape[1:]) == cshape:
                sub = np.unravel_index(mhw.cell_index, mhw.sshape)
                ranks = [np.cumsum(m) - 1 for m in alive]
                flat = np.ravel_multi_index(tuple(r[i] for r, i in zip(ranks, sub)), cshape)
            else:
                raise XmhwException(f"dstime has grid {tuple(tsg.shape[1:])}, the events were detected on {tuple(mhw.sshape)}")
            ts = tsg.reshape(tsg.shape[0], -1)[:, flat]
            cats = None if catg is None else catg.reshape(catg.shape[0], -1)[:, flat]
        period = [int(years[0]), int(years[-1])]                                 # stats.py:106-109
    if removeMissing and not sw_temp:                                             # stats.py:112-116
        raise XmhwException("To remove missing values you need to pass "
                            "the original temperature timeseries")
    if not period and not sw_temp:                                                # stats.py:117-121
        raise XmhwException("As the original timeseries is not available, the"
                            " timeseries period as [start_year, end_year] has to be passed")
    edges = block_bins(period, blockLength)
    res = compute(mhw.table, mhw.offsets, years, edges, mtime, ts, cats)
    nb = edges.shape[0] - 1
    coords = {"years": edges[:-1].copy()}
    if mhw.point:
        return BlockDataset({k: v[:, 0] for k, v in res.items()}, ("years",), coords, edges)
    alive = _alive_axes(mhw.keep, mhw.sshape)
    for d, m in zip(mhw.sdims, alive):
        coords[d] = np.asarray(mhw.coords[d])[m]
    ncol = int(np.prod(mhw.sshape))
    data = {}
    for k, v in res.items():
        full = np.full((nb, ncol), np.nan)
        full[:, mhw.cell_index] = v
        data[k] = _compress_grid(full.reshape((nb,) + mhw.sshape), alive, 1)
    return BlockDataset(data, ("years",) + mhw.sdims, coords, edges)
