"""threshold(): drop-in for xmhw.xmhw.threshold (xmhw/xmhw.py:38-247) with the
per-cell dask loop replaced by one batched GPU call.

Host side (this file, numpy/xarray): argument validation, climatology-period
slice, land mask + compaction, calendar -> doy labels, attrs, unstack.
Device side (HIP, via the C ABI): pooled quantile + mean per doy, Feb-29
step, circular running mean.

Input may be an ``xarray.DataArray`` (if xarray is installed; the return is
then an ``xarray.Dataset`` laid out like the reference's) or a ``GridSeries``
(plain numpy container, same fields); the return is then a ``ClimDataset``.
"""
from datetime import date

import numpy as np

from . import calendar as cal
from . import landmask
from .device import calc_clim_device, calc_clim_grid_device
from .exception import XmhwException
from .padding import make_pad

GITHUB = "https://github.com/coecms/xmhw"


class GridSeries:
    """Minimal stand-in for an xarray.DataArray: values + named dims + coords.

    coords: {dim name: 1-D array}; the time coordinate is datetime64 (or
    cftime-like objects).  attrs / coord_attrs / time_encoding carry metadata
    the reference reads (calendar) or copies to the output.
    """

    def __init__(self, values, dims, coords, attrs=None, coord_attrs=None, time_encoding=None):
        self.values = np.asarray(values)
        self.dims = tuple(dims)
        if self.values.ndim != len(self.dims):
            raise ValueError("dims do not match values.ndim")
        self.coords = {k: np.asarray(v) for k, v in coords.items()}
        self.attrs = dict(attrs or {})
        self.coord_attrs = {k: dict(v) for k, v in (coord_attrs or {}).items()}
        self.time_encoding = dict(time_encoding or {})


class ClimDataset:
    """What threshold() returns for GridSeries input: the reference Dataset's
    content as plain arrays.  ``ds["thresh"]``/``ds["seas"]`` have dims
    ``dims`` = ("doy", *spatial dims in sorted-name order)."""

    def __init__(self, thresh, seas, dims, coords, attrs, var_attrs, coord_attrs, quantile):
        self.data_vars = {"thresh": thresh, "seas": seas}
        self.dims = tuple(dims)
        self.coords = coords
        self.attrs = attrs
        self.var_attrs = var_attrs
        self.coord_attrs = coord_attrs
        self.quantile = quantile

    def __getitem__(self, k):
        return self.data_vars[k]

    @property
    def thresh(self):
        return self.data_vars["thresh"]

    @property
    def seas(self):
        return self.data_vars["seas"]


def _is_xarray(obj):
    return type(obj).__module__.split(".")[0] == "xarray"


def _from_xarray(temp):
    coords, coord_attrs = {}, {}
    for d in temp.dims:
        if d in temp.coords:
            coords[d] = temp[d].values
            coord_attrs[d] = dict(temp[d].attrs)
        else:
            coords[d] = np.arange(temp.sizes[d])
            coord_attrs[d] = {}
    return coords, coord_attrs


def _params_text(pctile, years, windowHalfWidth, skipna, smoothPercentile, smoothPercentileWidth, anynans):
    # xmhw.py:221-246
    params = f"""Threshold calculated using:
    {pctile} percentile;
    climatology period is {years[0]}-{years[1]}';
    window half width used for percentile is {windowHalfWidth}"""
    if skipna:
        params = params + """;
            NaNs where skipped in percentile and mean calculations"""
    if smoothPercentile:
        params = params + f""";
         width of moving average window to smooth percentile is
         {smoothPercentileWidth}"""
    if anynans:
        params = params + """;
            any grid point with even only 1 NaN along time
            axis has been removed from calculation"""
    return params


def threshold(
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
):
    """Calculate threshold and mean climatology (day-of-year).

    Same signature, defaults, exceptions and return layout as
    ``xmhw.xmhw.threshold`` (xmhw/xmhw.py:38-99).  Differences, all documented
    in DESIGN.md: results are float64 whatever the input dtype; ``skipna`` only
    changes the provenance text (it never changes the reference's numbers
    either, quirk Q1); ``maxPadLength`` follows xarray's interpolate_na(max_gap=...) rules: a timedelta
    on a datetime axis, a number on a numeric one (xmhw_amd/padding.py).
    The device stage is always the HIP path (no CPU fallback).
    """
    return _threshold(temp, calc_clim_device, tdim, climatologyPeriod, pctile, windowHalfWidth,
                      smoothPercentile, smoothPercentileWidth, maxPadLength, coldSpells, tstep,
                      anynans, skipna, grid_compute=calc_clim_grid_device)


def _threshold(temp, compute, tdim="time", climatologyPeriod=[None, None], pctile=90,
               windowHalfWidth=5, smoothPercentile=True, smoothPercentileWidth=31, maxPadLength=None,
               coldSpells=False, tstep=False, anynans=False, skipna=False, grid_compute=None):
    """Host side of threshold() around a device stage ``compute`` with the signature of
    ``device.calc_clim_device``.  The public threshold() passes the HIP path; the CPU tests of
    the host logic and of the multi-rank sharding pass a stand-in here.  ``grid_compute``
    (signature of ``device.calc_clim_grid_device``, returning full-width (D, N) arrays) additionally
    takes land_check()'s mask, the compaction and the final placement on the grid off the host;
    without it they run in numpy (landmask.land_check, boolean-mask assignment)."""
    if smoothPercentileWidth % 2 == 0:                       # xmhw.py:103-104
        raise XmhwException("smoothPercentileWidth should be odd")
    is_xr = _is_xarray(temp)
    dims = list(temp.dims)
    if tdim not in dims:                                      # xmhw.py:105-109
        raise XmhwException(f"{tdim} dimension not present, default"
                            + "is 'time' or pass as tdim='time_dimension_name'")
    if is_xr:
        values = temp.values
        coords, coord_attrs = _from_xarray(temp)
        attrs = dict(temp.attrs)
        enc = dict(getattr(temp[tdim], "encoding", {}) or {})
    else:
        values, coords = temp.values, dict(temp.coords)
        coord_attrs, attrs, enc = temp.coord_attrs, temp.attrs, temp.time_encoding
    time = np.asarray(coords[tdim])
    tax = dims.index(tdim)
    if all(climatologyPeriod):                                # xmhw.py:112-119 (both truthy)
        yrs = cal.years_of(time)
        sel = (yrs >= int(climatologyPeriod[0])) & (yrs <= int(climatologyPeriod[1]))
        idx = np.nonzero(sel)[0]
        if idx.size and idx[-1] - idx[0] + 1 == idx.size:
            # a contiguous run of steps (any sorted time axis): a view, not a copy of the series
            cut = [slice(None)] * values.ndim
            cut[tax] = slice(int(idx[0]), int(idx[-1]) + 1)
            values = values[tuple(cut)]
        else:
            values = np.compress(sel, values, axis=tax)
        time = time[sel]
    if time.shape[0] == 0:
        raise XmhwException("time axis is empty")
    point = len(dims) == 1                                    # xmhw.py:122-126
    on_device = grid_compute is not None and not point
    if point:
        ts = np.ascontiguousarray(values.reshape(-1, 1))
        keep, sdims, sshape = np.array([True]), [], ()
    elif on_device:
        stacked, sdims, sshape = landmask.stack_cells(values, dims, tdim)   # raises like land_check
    else:
        ts, keep, sdims, sshape = landmask.land_check(values, dims, tdim, anynans)
    calname = cal.calendar_of(time, enc, coord_attrs.get(tdim, {}))
    if cal.get_calendar(calname) == 360.0:                    # xmhw.py:142-144
        tstep = True
    doy = cal.add_doy(time, keep_tstep=tstep)                 # xmhw.py:145
    # ts.interpolate_na(dim=tdim, max_gap=maxPadLength) after land_check (xmhw.py:157-160): on the device
    # copy of the compacted series, handed to the device stage as a recipe (None: no interpolation)
    pad = make_pad(maxPadLength, time)
    extra = {} if pad is None else {"pad": pad}
    try:
        if on_device:
            keep, doys, th, se = grid_compute(stacked, doy, anynans, pctile, windowHalfWidth, smoothPercentile,
                                              smoothPercentileWidth, tstep, coldSpells, **extra)
        else:
            doys, th, se = compute(ts, doy, pctile, windowHalfWidth, smoothPercentile,
                                   smoothPercentileWidth, tstep, coldSpells, **extra)
    finally:
        if pad is not None:
            pad.free()
    if th is None:        # a rank of a sharded run that is not the root: nothing to assemble (xmhw_amd/sharded.py)
        return None

    D = doys.shape[0]
    yrs = cal.years_of(time)
    out_attrs = {
        "source": f"xmhw code: {GITHUB}",
        "title": ("Seasonal climatology and threshold "
                  + "calculated to detect marine heatwaves following the "
                  + " Hobday et al. (2016) definition"),
        "history": f"{date.today()}: calculated using xmhw code {GITHUB}",
        "xmhw_parameters": _params_text(pctile, (int(yrs[0]), int(yrs[-1])), windowHalfWidth, skipna,
                                        smoothPercentile, smoothPercentileWidth, anynans),
    }
    var_attrs = {"thresh": {"units": "degree_C"}, "seas": {"units": "degree_C"}}   # quirk Q9
    doy_attrs = {"units": "1", "long_name": "Day of the year"}
    if point:
        thg, seg = th[:, 0], se[:, 0]
        odims = ("doy",)
        ocoords = {"doy": doys}
    else:
        if on_device:
            full_th, full_se = th, se            # already on the grid, NaN at the dropped cells
        else:
            full_th = np.full((D, keep.shape[0]), np.nan)
            full_se = np.full((D, keep.shape[0]), np.nan)
            full_th[:, keep] = th
            full_se[:, keep] = se
        thg = full_th.reshape((D,) + sshape)
        seg = full_se.reshape((D,) + sshape)
        odims = ("doy",) + tuple(sdims)
        ocoords = {"doy": doys}
        keepg = keep.reshape(sshape)
        # unstack('cell') only has the coordinate values of surviving cells: a
        # line that is all land disappears from the grid (docs/threshold.rst:104-108)
        for ax, d in enumerate(sdims):
            other = tuple(i for i in range(len(sdims)) if i != ax)
            alive = keepg.any(axis=other) if other else keepg
            if not alive.all():
                thg = landmask.compress_axis(thg, alive, ax + 1)
                seg = landmask.compress_axis(seg, alive, ax + 1)
            ocoords[d] = np.asarray(coords[d])[alive]
    q = pctile / 100.0
    out_coord_attrs = {"doy": doy_attrs}
    for d in sdims:
        out_coord_attrs[d] = dict(coord_attrs.get(d, {}))
    if is_xr:
        import xarray as xr
        ds = xr.Dataset(
            {"thresh": (odims, thg), "seas": (odims, seg)},
            coords={**{k: (k, v) for k, v in ocoords.items()}, "quantile": q},
        )
        for c, a in out_coord_attrs.items():
            ds[c].attrs.update(a)
        ds.attrs.update(out_attrs)
        ds["thresh"].attrs.update(var_attrs["thresh"])
        ds["seas"].attrs.update(var_attrs["seas"])
        return ds
    return ClimDataset(thg, seg, odims, ocoords, out_attrs, var_attrs, out_coord_attrs, q)


def threshold_array(values, time, dims=("time", "lat", "lon"), coords=None, calendar="", **kwargs):
    """Convenience: threshold() on a bare ndarray + datetime64 time axis."""
    dims = tuple(dims)
    tdim = kwargs.get("tdim", "time")
    cds = {tdim: np.asarray(time)}
    for i, d in enumerate(dims):
        if d != tdim:
            cds[d] = np.asarray(coords[d]) if coords and d in coords else np.arange(np.shape(values)[i])
    enc = {"calendar": calendar} if calendar else {}
    return threshold(GridSeries(values, dims, cds, time_encoding=enc), **kwargs)
