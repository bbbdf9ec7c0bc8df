"""detect(): drop-in for xmhw.xmhw.detect (xmhw/xmhw.py:310-518) with the per-cell dask loop
over define_events() (xmhw/identify.py:326-412) replaced by batched GPU calls.

Host side (this file, numpy/xarray): argument validation, land masks + compaction of the series
and of both climatologies, doy labels, cold-spell flip, attrs, assembly of the `events` dimension.
Device side (HIP, via the C ABI; xmhw_amd/detect_front.py): threshold re-expansion by doy,
exceedance, event filter + gap joining, per-event statistics, optional per-step columns.

Input may be ``xarray.DataArray``s (return: ``xarray.Dataset``s laid out like the reference's) or
``GridSeries`` (return: ``EventDataset`` / ``InterDataset``, plain-array containers of the same
content).  The reference materialises every variable densely as (events, *grid) where `events` is
the union of all cells' event labels; EventDataset keeps the compact per-event table and builds
that layout on request (``to_dense``), because it does not fit in memory for global grids.
"""
from datetime import date

import numpy as np

from . import calendar as cal
from . import landmask
from .api import GITHUB, GridSeries, _from_xarray, _is_xarray
from .detect_front import EVENT_COLUMNS, INTERMEDIATE_U8, detect_cells, detect_grid
from .exception import XmhwException
from .padding import make_pad

TIME_COLUMNS = ("time_start", "time_end", "time_peak")
INTER_VARIABLES = ["ts", "seas", "thresh", "bthresh", "events", "relSeas", "relThresh", "relThreshNorm",
                   "severity", "cats"] + INTERMEDIATE_U8 + ["mabs"]     # column order of mhw_df()


def _mhw_var_attrs(uts="degree_C"):
    """annotate_ds(kind='mhw') (xmhw/identify.py:594-696)."""
    a = {
        "event": ("MHW event identifier: starting index", "1"),
        "duration": ("MHW duration in number of days", "1"),
        "intensity_max": ("MHW maximum (peak) intensity relative to seasonal climatology", uts),
        "intensity_mean": ("MHW mean intensity relative to seasonal climatology", uts),
        "intensity_var": ("MHW intensity variability relative to seasonal climatology", uts),
        "intensity_cumulative": ("MHW cumulative intensity relative to seasonal climatology", f"{uts} day"),
        "severity_max": ("MHW maximum (peak) severity relative to seasonal climatology", uts),
        "severity_mean": ("MHW mean severity relative to seasonal climatology", uts),
        "severity_var": ("MHW severity variability relative to seasonal climatology", uts),
        "severity_cumulative": ("MHW cumulative severity relative to seasonal climatology", f"{uts} day"),
        "rate_onset": ("MHW onset rate", f"{uts} day-1"),
        "rate_decline": ("MHW decline rate", f"{uts} day-1"),
        "intensity_max_relThresh": ("MHW maximum (peak) intensity relative to threshold", uts),
        "intensity_mean_relThresh": ("MHW mean intensity relative to threshold", uts),
        "intensity_var_relThresh": ("MHW intensity variability relative to threshold", uts),
        "intensity_cumulative_relThresh": ("MHW cumulative intensity relative to threshold", f"{uts} day"),
        "intensity_max_abs": ("MHW maximum (peak) intensity absolute magnitude", uts),
        "intensity_mean_abs": ("MHW mean intensity absolute magnitude", uts),
        "intensity_var_abs": ("MHW intensity variability abosulute magnitude", uts),
        "intensity_cumulative_abs": ("MHW cumulative intensity absolute magnitude", f"{uts} day"),
        "category": ("MHW category based on peak intensity: 1: Moderate, 2: Strong, 3: Severe or 4: Extreme", None),
        "duration_moderate": ("Number of days falling in category Moderate", "1"),
        "duration_strong": ("Number of days falling in category Strong", "1"),
        "duration_severe": ("Number of days falling in category Severe", "1"),
        "duration_extreme": ("Number of days falling in category Extreme", "1"),
    }
    out = {}
    for k, (ln, units) in a.items():
        out[k] = {"long_name": ln}
        if units is not None:
            out[k]["units"] = units
    return out


def _params_text(minDuration, joinGaps, maxGap, coldSpells, maxPadLength, anynans):
    # xmhw.py:486-514
    params = f"MHW detected using: {minDuration} days of minimum duration"
    if joinGaps:
        params = (params + f""";
            events separated by {maxGap} or less days were joined""")
    if coldSpells:
        params = (params + """;
                cold events were detected instead of heat events""")
    if maxPadLength:
        params = (params + f""";
            where original timeseries had missing values interpolation
            was used to fill them. Gaps > {maxPadLength} days long were
            left as NaNs;""")
    if anynans:
        params = (params + """;
            any grid point with even only 1 NaN along time
            axis has been removed from calculation""")
    return params


def _alive_axes(keep, sshape):
    """unstack('cell') only knows the coordinate values of surviving cells: a grid line that is
    all land disappears (as in threshold(), docs/threshold.rst:104-108)."""
    keepg = keep.reshape(sshape)
    alive = []
    for ax in range(len(sshape)):
        other = tuple(i for i in range(len(sshape)) if i != ax)
        alive.append(keepg.any(axis=other) if other else keepg.copy())
    return alive


def _compress_grid(a, alive, first_axis):
    from .landmask import compress_axis
    for ax, m in enumerate(alive):
        a = compress_axis(a, m, first_axis + ax)
    return a


class EventDataset:
    """What detect() returns for GridSeries input: the content of the reference's `mhw` Dataset.

    table   (n_events, 31) float64, columns ``columns`` (= the reference's variable names); the three
            time_* columns hold positions along the time axis (``time[pos]`` gives the stamp)
    offsets (n_cells+1,): the events of ocean cell i are ``table[offsets[i]:offsets[i+1]]``, in time order
    cell_index  flat index of every ocean cell into the stacked grid ``sshape`` (dims ``sdims``, the
            non-time dims in sorted-name order, as land_check() stacks them)
    """

    columns = EVENT_COLUMNS

    def __init__(self, table, offsets, time, cell_index, keep, sdims, sshape, coords, attrs, var_attrs,
                 coord_attrs, point):
        self.table, self.offsets, self.time = table, offsets, time
        self.cell_index, self.keep = cell_index, keep
        self.sdims, self.sshape, self.coords = tuple(sdims), tuple(sshape), coords
        self.attrs, self.var_attrs, self.coord_attrs = attrs, var_attrs, coord_attrs
        self.point = point

    @property
    def n_events(self):
        return int(self.table.shape[0])

    @property
    def n_cells(self):
        return int(self.offsets.shape[0] - 1)

    @property
    def events(self):
        """The `events` coordinate of the reference's Dataset: sorted union of all cells' labels."""
        return np.unique(self.table[:, 0])

    def cell(self, i):
        """Per-event variables of ocean cell i (what define_events() returns for that cell)."""
        sl = slice(int(self.offsets[i]), int(self.offsets[i + 1]))
        out = {}
        for k, name in enumerate(self.columns):
            out[name] = self.time_stamps(self.table[sl, k]) if name in TIME_COLUMNS else self.table[sl, k]
        return out

    def time_stamps(self, pos):
        pos = np.asarray(pos, dtype=np.float64)
        ok = ~np.isnan(pos)
        t = np.asarray(self.time)
        if t.dtype.kind == "M":
            out = np.full(pos.shape, np.datetime64("NaT"), dtype=t.dtype)
        else:
            out = np.full(pos.shape, None, dtype=object)
        out[ok] = t[pos[ok].astype(np.int64)]
        return out

    def to_dense(self, variables=None):
        """The reference's layout: (dims, coords, {name: array}) with dims ("events", *sdims); NaN / NaT
        where a cell has no event with that label or is land.  Size: n_labels x grid x 8 B per variable."""
        variables = list(variables) if variables is not None else list(self.columns)
        ev = self.events
        cell_of_row = np.repeat(np.arange(self.n_cells), np.diff(self.offsets))
        erow = np.searchsorted(ev, self.table[:, 0])
        if self.point:
            dims, coords, shape = ("events",), {"events": ev}, (ev.shape[0],)
            flat = np.zeros(self.n_events, dtype=np.int64)
            ncol, alive = 1, None
        else:
            alive = _alive_axes(self.keep, self.sshape)
            dims = ("events",) + self.sdims
            coords = {"events": ev}
            for d, m in zip(self.sdims, alive):
                coords[d] = np.asarray(self.coords[d])[m]
            ncol = int(np.prod(self.sshape))
            flat = self.cell_index[cell_of_row]
        data = {}
        for name in variables:
            k = self.columns.index(name)
            full = np.full((ev.shape[0], ncol), np.nan)
            full[erow, flat] = self.table[:, k]
            if not self.point:
                full = _compress_grid(full.reshape((ev.shape[0],) + self.sshape), alive, 1)
            else:
                full = full[:, 0]
            data[name] = self.time_stamps(full) if name in TIME_COLUMNS else full
        return dims, coords, data

    def to_xarray(self):
        import xarray as xr
        dims, coords, data = self.to_dense()
        ds = xr.Dataset({k: (dims, v) for k, v in data.items()}, coords={k: (k, v) for k, v in coords.items()})
        for c, a in self.coord_attrs.items():
            ds[c].attrs.update(a)
        for v, a in self.var_attrs.items():
            ds[v].attrs.update(a)
        ds.attrs.update(self.attrs)
        return ds


class InterDataset:
    """The `intermediate` Dataset of detect() as plain arrays: ``data[name]`` has dims ``dims`` =
    (time, *sdims) (("index",) for a single point, as in the reference); land cells are NaN."""

    def __init__(self, data, dims, coords):
        self.data_vars, self.dims, self.coords = data, tuple(dims), coords

    def __getitem__(self, k):
        return self.data_vars[k]

    def to_xarray(self):
        import xarray as xr
        return xr.Dataset({k: (self.dims, v) for k, v in self.data_vars.items()},
                          coords={k: (k, v) for k, v in self.coords.items()})


def _unpack(arr, tdim):
    if _is_xarray(arr):
        coords, coord_attrs = _from_xarray(arr)
        return arr.values, list(arr.dims), coords, coord_attrs, dict(arr.attrs)
    return arr.values, list(arr.dims), dict(arr.coords), arr.coord_attrs, arr.attrs


def detect(
    temp,
    th,
    se,
    tdim="time",
    minDuration=5,
    joinGaps=True,
    maxGap=2,
    maxPadLength=None,
    coldSpells=False,
    intermediate=False,
    anynans=False,
    tstep=False,
):
    """Applies the Hobday et al. (2016) marine heat wave definition to a temperature timeseries.

    Same signature, defaults, exceptions and return values as ``xmhw.xmhw.detect``
    (xmhw/xmhw.py:310-372).  Differences: for GridSeries input the return is an EventDataset (compact table, dense on
    request) and, with ``intermediate``, an InterDataset.  The device stage is always the HIP path.
    """
    return _detect(temp, th, se, detect_cells, tdim, minDuration, joinGaps, maxGap, maxPadLength, coldSpells,
                   intermediate, anynans, tstep, grid_compute=detect_grid)



def threshold_detect(
    temp,
    tdim="time",
    climatologyPeriod=[None, None],
    pctile=90,
    windowHalfWidth=5,
    smoothPercentile=True,
    smoothPercentileWidth=31,
    maxPadLength=None,
    coldSpells=False,
    tstep=False,
    anynans=False,
    skipna=False,
    minDuration=5,
    joinGaps=True,
    maxGap=2,
    intermediate=False,
):
    """``clim = threshold(temp, ...)`` followed by ``detect(temp, clim['thresh'], clim['seas'], ...)``
    (the reference's usual pair, docs/gettingstarted.rst:37-45) with the series uploaded ONCE: the
    compacted device copy of every column slab made by threshold() stays in HBM (60.6 GB for the
    0.25 degree 40-year grid, device.ResidentSeries) and detect() runs on it.  Returns
    ``(clim, mhw)`` or ``(clim, mhw, mhw_inter)`` with ``intermediate``; every value is identical to
    the two separate calls with the same arguments.  When the climatology period selects a part of
    the time axis, or the series does not leave room in HBM, or ``intermediate`` is asked for (the
    per-step columns take the host-compacted path), detect() uploads the series again as the separate
    call would."""
    from .api import _threshold
    from .device import ResidentSeries, calc_clim_device, calc_clim_grid_device
    if maxGap >= minDuration:                                  # detect()'s check, before any work is done
        raise XmhwException("Maximum gap between mhw events should"
                            + " be smaller than event minimum duration")
    store = ResidentSeries()

    def clim_grid(*a, **k):
        return calc_clim_grid_device(*a, resident=store, **k)

    def events_grid(*a, **k):
        return detect_grid(*a, resident=store, **k)

    try:
        clim = _threshold(temp, calc_clim_device, tdim, climatologyPeriod, pctile, windowHalfWidth,
                          smoothPercentile, smoothPercentileWidth, maxPadLength, coldSpells, tstep,
                          anynans, skipna, grid_compute=clim_grid)
        if _is_xarray(clim):
            th, se = clim["thresh"], clim["seas"]
        else:
            th, se = climatology_series(clim, "thresh"), climatology_series(clim, "seas")
        out = _detect(temp, th, se, detect_cells, tdim, minDuration, joinGaps, maxGap,
                      maxPadLength, coldSpells, intermediate, anynans, tstep, grid_compute=events_grid)
    finally:
        store.free()
    if intermediate:
        return (clim,) + tuple(out)
    return clim, out

def _detect(temp, th, se, compute, tdim="time", minDuration=5, joinGaps=True, maxGap=2, maxPadLength=None,
            coldSpells=False, intermediate=False, anynans=False, tstep=False, grid_compute=None):
    """Host side of detect() around a device stage ``compute`` with the signature of
    ``detect_front.detect_cells``.  The public detect() passes the HIP path; the CPU tests of the
    host logic pass an oracle-based stand-in.  ``grid_compute`` (signature of
    ``detect_front.detect_grid``) also takes the series' land mask and compaction off the host."""
    if maxGap >= minDuration:                                  # xmhw.py:373-378
        raise XmhwException("Maximum gap between mhw events should"
                            + " be smaller than event minimum duration")
    is_xr = _is_xarray(temp)
    values, dims, coords, coord_attrs, attrs = _unpack(temp, tdim)
    if tdim not in dims:
        raise XmhwException(f"{tdim} dimension not present, default"
                            + "is 'time' or pass as tdim='time_dimension_name'")
    thv, thdims, thcoords, _, _ = _unpack(th, "doy")
    sev, sedims, secoords, _, _ = _unpack(se, "doy")
    if "doy" not in thdims or "doy" not in sedims:
        raise XmhwException("th and se must have a 'doy' dimension")
    time = np.asarray(coords[tdim])
    point = len(dims) == 1                                    # xmhw.py:381-385
    on_device = grid_compute is not None and not point
    if point:
        if len(thdims) != 1 or len(sedims) != 1:
            raise XmhwException("a single-point series needs single-point climatologies")
        ts = np.ascontiguousarray(np.asarray(values).reshape(-1, 1))
        thc, sec = np.asarray(thv, dtype=np.float64).reshape(-1, 1), np.asarray(sev, dtype=np.float64).reshape(-1, 1)
        keep, sdims, sshape = np.array([True]), [], ()
    else:
        # land_check on all three (xmhw.py:398-402); cells pair up by POSITION after each dropna
        # (stack(create_index=False), then ts.sel(cell=c) / th.sel(cell=c): xmhw.py:437-443)
        if on_device:
            # the device masks and compacts all three; here they are only stacked (views / reshapes)
            stacked, sdims, sshape = landmask.stack_cells(values, dims, tdim)
            thc, thsd, thshape = landmask.stack_cells(thv, thdims, "doy")
            sec, sesd, seshape = landmask.stack_cells(sev, sedims, "doy")
            if thsd != sdims or sesd != sdims:
                raise XmhwException(f"temp, th and se are not on the same dimensions: {sdims}, {thsd}, {sesd}")
            # (threshold() drops all-land grid lines, so th / se may live on a smaller grid than temp: the
            # survivors pair up by position, whatever the grids)
            clim_stacked = True
            ts = None
        else:
            ts, keep, sdims, sshape = landmask.land_check(values, dims, tdim, anynans)
            thc, _, thsd, _ = landmask.land_check(thv, thdims, "doy", anynans)
            sec, _, sesd, _ = landmask.land_check(sev, sedims, "doy", anynans)
            if thc.shape[1] != ts.shape[1] or sec.shape[1] != ts.shape[1] or thsd != sdims or sesd != sdims:
                raise XmhwException("temp, th and se do not have the same ocean cells: "
                                    + f"{ts.shape[1]}, {thc.shape[1]}, {sec.shape[1]} cells over dims {sdims}, {thsd}, {sesd}")
    doys = np.asarray(thcoords["doy"])
    if not np.array_equal(doys, np.asarray(secoords["doy"])):
        raise XmhwException("th and se have different doy coordinates")
    doy = cal.add_doy(time, keep_tstep=tstep)                  # xmhw.py:404 (no calendar sniffing here)

    import time as _time
    from .device import _trace
    _t0 = _time.perf_counter()
    pad = make_pad(maxPadLength, time)                         # xmhw.py:407-410, after land_check
    extra = {} if pad is None else {"pad": pad}
    try:
        if on_device:
            res = grid_compute(stacked, anynans, sec, thc, doy, doys, minDuration, joinGaps, maxGap, coldSpells,
                               intermediate, clim_stacked=clim_stacked, **extra)
            keep = res["keep"]
        else:
            res = compute(ts, sec, thc, doy, doys, minDuration, joinGaps, maxGap, coldSpells, intermediate, **extra)
    finally:
        if pad is not None:
            pad.free()
    _trace("detect: device stage", _t0)
    table, offsets = res["table"], res["offsets"]
    if coldSpells:                                             # flip_cold(), xmhw/features.py:298-315
        table = table.copy()
        for k, name in enumerate(EVENT_COLUMNS):
            if "intensity" in name and "_var" not in name:
                table[:, k] = -1 * table[:, k]

    out_attrs = {
        "source": f"xmhw code: {GITHUB}",
        "title": ("Marine heatwave events identified "
                  + "applying the Hobday et al. (2016) marine heat wave definition"),
        "history": f"{date.today()}: calculated using xmhw code {GITHUB}",
        "xmhw_parameters": _params_text(minDuration, joinGaps, maxGap, coldSpells, maxPadLength, anynans),
    }
    out_coord_attrs = {"events": {"units": "1", "long_name": "MHW event identifier: starting index"}}
    for d in sdims:
        out_coord_attrs[d] = dict(coord_attrs.get(d, {}))
    mhw = EventDataset(table, offsets, time, np.nonzero(keep)[0], keep, sdims, sshape,
                       {d: np.asarray(coords[d]) for d in sdims}, out_attrs, _mhw_var_attrs(), out_coord_attrs, point)
    mhw_inter = None
    if intermediate:
        mhw_inter = _assemble_inter(res["inter"], time, tdim, keep, sdims, sshape, coords, point)
    if is_xr:
        mhw = mhw.to_xarray()
        if intermediate:
            mhw_inter = mhw_inter.to_xarray()
    if intermediate:
        return mhw, mhw_inter
    return mhw


def _assemble_inter(inter, time, tdim, keep, sdims, sshape, coords, point):
    if point:                                                  # define_events' own frame: dim 'index'
        return InterDataset({k: inter[k][:, 0] for k in INTER_VARIABLES}, ("index",), {"index": time})
    T = time.shape[0]
    alive = _alive_axes(keep, sshape)
    whole = bool(keep.all())
    data = {}
    for k in INTER_VARIABLES:
        v = inter[k]
        if whole:
            full = v
        else:
            # xarray's unstack fills missing cells with NaN (bool columns are promoted)
            full = np.full((T, keep.shape[0]), np.nan, dtype=np.float64 if v.dtype == bool else v.dtype)
            full[:, keep] = v
        data[k] = _compress_grid(full.reshape((T,) + tuple(sshape)), alive, 1)
    ocoords = {tdim: time}
    for d, m in zip(sdims, alive):
        ocoords[d] = np.asarray(coords[d])[m]
    return InterDataset(data, (tdim,) + tuple(sdims), ocoords)


def climatology_series(clim, name):
    """ClimDataset field -> GridSeries that detect() accepts as `th` / `se`."""
    return GridSeries(clim[name], clim.dims, dict(clim.coords), coord_attrs=clim.coord_attrs)
