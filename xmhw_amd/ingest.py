"""File -> device ingest for threshold() / detect(): netCDF classic (xmhw_amd/netcdf3.py),
netCDF-4 / HDF5 (xmhw_amd/hdf5min.py) and zarr v2 directory stores (xmhw_amd/zarr2.py).

The reference leaves reading to xarray (``xr.open_dataset(...)['sst']``, docs/gettingstarted.rst:
30-33) and masks land by ``dropna`` on the stacked array (xmhw/identify.py:520-528).  Here a netCDF
classic file is memory-mapped (xmhw_amd/netcdf3.py), the variable's RAW bytes -- big-endian, possibly
CF-packed int16 -- are handed to the grid entry points as a ``PackedArray``, and everything after
that happens on the device, slab by slab with the next slab's upload overlapping the current slab's
kernels (device.SlabPrefetcher): byte swap + ``raw * scale_factor + add_offset`` + ``_FillValue`` ->
NaN (xmhw_decode), land mask, compaction, climatology, placement back on the grid.  An int16-packed
archive therefore crosses PCIe at 2 bytes per sample instead of 4 (or 8 once xarray has decoded it).

    temp = open_series("sst.day.mean.nc", "sst")      # GridSeries over the mapped file
    clim = xmhw_amd.threshold(temp)
    mhw  = xmhw_amd.detect(temp, climatology_series(clim, "thresh"), climatology_series(clim, "seas"))
"""
import re

import numpy as np

from . import netcdf3
from .api import GridSeries
from .device import PackedArray
from .exception import XmhwException

_UNITS = {"days": 86400.0, "day": 86400.0, "d": 86400.0, "hours": 3600.0, "hour": 3600.0, "hrs": 3600.0, "h": 3600.0,
          "minutes": 60.0, "minute": 60.0, "min": 60.0, "seconds": 1.0, "second": 1.0, "secs": 1.0, "s": 1.0}
_STANDARD = ("", "standard", "gregorian", "proleptic_gregorian")


class CFTime:
    """A date on a non-standard CF calendar (noleap / 365_day, all_leap / 366_day, 360_day): just what
    calendar.add_doy() reads from cftime objects -- year, month, dayofyr, calendar."""
    __slots__ = ("year", "month", "day", "dayofyr", "calendar")

    def __init__(self, year, month, day, dayofyr, calendar):
        self.year, self.month, self.day, self.dayofyr, self.calendar = year, month, day, dayofyr, calendar

    def __repr__(self):
        return f"CFTime({self.year:04d}-{self.month:02d}-{self.day:02d}, {self.calendar})"


def _month_table(calendar):
    if calendar in ("360_day",):
        return [30] * 12
    feb = 29 if calendar in ("all_leap", "366_day") else 28
    return [31, feb, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]


def decode_time(values, units, calendar=""):
    """CF time coordinate -> datetime64[s] (standard calendars) or an object array of CFTime."""
    m = re.match(r"\s*(\w+)\s+since\s+(\d{1,4})-(\d{1,2})-(\d{1,2})(?:[ T](\d{1,2}):(\d{1,2})(?::(\d{1,2}(?:\.\d*)?))?)?", units or "")
    if not m or m.group(1).lower() not in _UNITS:
        raise XmhwException(f"cannot decode time units {units!r}")
    step = _UNITS[m.group(1).lower()]
    y0, mo0, d0 = int(m.group(2)), int(m.group(3)), int(m.group(4))
    sec0 = int(m.group(5) or 0) * 3600 + int(m.group(6) or 0) * 60 + float(m.group(7) or 0)
    secs = np.asarray(values, dtype=np.float64) * step + sec0
    calendar = (calendar or "").lower()
    if calendar in _STANDARD:
        origin = np.datetime64(f"{y0:04d}-{mo0:02d}-{d0:02d}", "s")
        return origin + np.round(secs).astype("timedelta64[s]")
    months = _month_table(calendar)
    ylen = sum(months)
    day0 = sum(months[:mo0 - 1]) + (d0 - 1)
    days = np.floor(secs / 86400.0).astype(np.int64) + day0
    out = np.empty(days.shape, dtype=object)
    cum = np.cumsum([0] + months)
    for i, dd in enumerate(days):
        yr, doy = y0 + dd // ylen, dd % ylen
        mon = int(np.searchsorted(cum, doy, side="right"))
        out[i] = CFTime(int(yr), mon, int(doy - cum[mon - 1]) + 1, int(doy) + 1, calendar)
    return out


def cf_recipe(what, dtype, at):
    """The device decoder's recipe for a stored variable: xarray's CF decoding rules (float32 for int16 data with
    float32 packing attributes, float64 otherwise; ``_FillValue`` / ``missing_value`` -> NaN)."""
    dtype = np.dtype(dtype)
    kind, isz = dtype.kind, dtype.itemsize
    scale, offset, fill = at.get("scale_factor"), at.get("add_offset"), at.get("_FillValue", at.get("missing_value"))
    if kind == "i" and isz == 2:
        f64 = any(isinstance(x, (float, np.float64)) and not isinstance(x, np.float32) for x in (scale, offset) if x is not None)
        out = np.float64 if f64 else np.float32
        if scale is None and offset is not None:
            scale = 1.0
    elif kind == "f" and isz in (4, 8):
        out = np.float32 if isz == 4 else np.float64
        if scale is not None or offset is not None:
            scale = 1.0 if scale is None else scale
    else:
        raise XmhwException(f"{what} is stored as {dtype}; the device decoder takes int16, float32, float64")
    return dict(scale=None if scale is None else float(scale), offset=None if offset is None else float(offset),
                fill=None if fill is None else float(fill), out=np.dtype(out).name)


def open_series(path, varname=None, tdim=None):
    """A (time, y, x) variable of a netCDF classic file, or of a zarr v2 directory store (a directory holding
    ``.zgroup`` / ``.zarray``: xmhw_amd/zarr2.py), as a GridSeries whose values are a zero-copy
    PackedArray over the mapped file: nothing is read or decoded on the host."""
    import os
    if os.path.isdir(path):
        from . import zarr2
        return zarr2.open_series(path, varname, tdim)
    with open(path, "rb") as fh:
        magic = fh.read(8)
    if magic == b"\x89HDF\r\n\x1a\n":
        return _open_netcdf4(path, varname, tdim)
    f = netcdf3.File(path)
    cands = [v for v in f.variables.values() if len(v.dims) >= 2 and v.name not in f.dimensions]
    if varname is None:
        if len(cands) != 1:
            raise XmhwException(f"{path}: name the variable, candidates {[v.name for v in cands]}")
        var = cands[0]
    else:
        if varname not in f.variables:
            raise XmhwException(f"{path}: no variable {varname!r}")
        var = f.variables[varname]
    tdim = tdim or var.dims[0]
    if var.dims[0] != tdim:
        raise XmhwException(f"{path}: {var.name} must have {tdim!r} as its first (slowest) dimension, has {var.dims}")
    inner = var.data[0] if var.shape[0] else var.data
    if var.shape[0] and not np.asarray(inner).flags.c_contiguous:
        raise XmhwException(f"{path}: {var.name}: the non-time dimensions must be contiguous in the file")
    at = var.attrs
    decode = cf_recipe(f"{path}: {var.name}", var.dtype, at)
    # where the bytes live, for uploads that pread() instead of faulting the mapping in
    decode["file"] = dict(fd=f.fileno(), address=f.map_address, length=f.map_length)
    coords, coord_attrs = {}, {}
    for d in var.dims:
        if d in f.variables and f.variables[d].dims == (d,):
            cv = f.variables[d]
            coords[d] = np.asarray(cv.data).astype(cv.dtype.newbyteorder("="))
            coord_attrs[d] = {k: v for k, v in cv.attrs.items()}
        else:
            coords[d] = np.arange(var.shape[var.dims.index(d)])
            coord_attrs[d] = {}
    tat = coord_attrs.get(tdim, {})
    enc = {}
    if "units" in tat and " since " in str(tat["units"]):
        cal = str(tat.get("calendar", ""))
        coords[tdim] = decode_time(coords[tdim], tat["units"], cal)
        if cal:
            enc["calendar"] = cal
    attrs = {k: v for k, v in at.items() if k not in ("scale_factor", "add_offset", "_FillValue", "missing_value")}
    gs = GridSeries.__new__(GridSeries)
    gs.values = PackedArray(var.data, decode)
    gs.dims = tuple(var.dims)
    gs.coords = coords
    gs.attrs = attrs
    gs.coord_attrs = coord_attrs
    gs.time_encoding = enc
    gs._file = f                     # keeps the mapping alive
    return gs


def _open_netcdf4(path, varname=None, tdim=None):
    """netCDF-4 (HDF5) through xmhw_amd/hdf5min.py: a contiguous, unfiltered variable is a zero-copy window of
    the mapped file (uploaded with pread() like a classic file); a chunked / deflated one -- what OISST and the
    reference's own fixtures are -- is inflated once into its stored dtype on the host.  Either way byte order,
    CF packing and the fill value are left to the device decoder."""
    from . import hdf5min
    f = hdf5min.File(path)
    sets = {k: f[k] for k in f.keys()}
    sets = {k: v for k, v in sets.items() if not v.is_group}
    cands = [k for k, v in sets.items() if v.shape is not None and len(v.shape) >= 2]
    if varname is None:
        if len(cands) != 1:
            raise XmhwException(f"{path}: name the variable, candidates {cands}")
        varname = cands[0]
    if varname not in sets:
        raise XmhwException(f"{path}: no variable {varname!r}")
    var = sets[varname]
    # dimension names: netCDF-4 numbers its dimensions (_Netcdf4Dimid on the dimension scales) and lists a
    # variable's dimension ids in _Netcdf4Coordinates; without that attribute the scales are matched by length
    scales = {k: v for k, v in sets.items() if v.attrs.get("CLASS") == "DIMENSION_SCALE" and v.shape is not None and len(v.shape) == 1}
    by_id = {int(v.attrs["_Netcdf4Dimid"]): k for k, v in scales.items() if "_Netcdf4Dimid" in v.attrs}
    ids = var.attrs.get("_Netcdf4Coordinates")
    dims = None
    if ids is not None:
        ids = [int(i) for i in np.atleast_1d(ids)]
        if len(ids) == len(var.shape) and all(i in by_id for i in ids):
            dims = tuple(by_id[i] for i in ids)
    if dims is None:
        dims = []
        for n in var.shape:
            m = [k for k, v in scales.items() if v.shape[0] == n and k not in dims]
            dims.append(m[0] if len(m) == 1 else f"dim_{len(dims)}")
        dims = tuple(dims)
    tdim = tdim or dims[0]
    if dims[0] != tdim:
        raise XmhwException(f"{path}: {varname} must have {tdim!r} as its first (slowest) dimension, has {dims}")
    at = {}
    for k, v in var.attrs.items():
        if k in ("DIMENSION_LIST", "_Netcdf4Coordinates", "_Netcdf4Dimid", "CLASS", "NAME", "REFERENCE_LIST") or v is None:
            continue
        at[k] = v[0] if isinstance(v, np.ndarray) and v.size == 1 else v
    for k in ("_FillValue", "missing_value"):
        if k in at and isinstance(at[k], (float, np.floating)) and np.isnan(at[k]):
            del at[k]                                    # NaN as the fill value of float data: nothing to replace
    decode = cf_recipe(f"{path}: {varname}", var.dtype, at)
    off = var.contiguous_offset()
    if off is not None:
        data = f.window(off, var.dtype, var.shape)
        decode["file"] = dict(fd=f.fileno(), address=f.map_address, length=f.map_length)
    else:
        data = var.read()
    coords, coord_attrs = {}, {}
    for i, d in enumerate(dims):
        cv = sets.get(d)
        if cv is not None and cv.shape == (var.shape[i],) and "netCDF dimension but not" not in str(cv.attrs.get("NAME", "")):
            coords[d] = cv.read().astype(cv.dtype.newbyteorder("="))
            coord_attrs[d] = {k: v for k, v in cv.attrs.items()
                              if v is not None and k not in ("CLASS", "NAME", "REFERENCE_LIST", "_Netcdf4Dimid", "_FillValue")}
        else:
            coords[d] = np.arange(var.shape[i])
            coord_attrs[d] = {}
    tat = coord_attrs.get(tdim, {})
    enc = {}
    if "units" in tat and " since " in str(tat["units"]):
        cal = str(tat.get("calendar", ""))
        coords[tdim] = decode_time(coords[tdim], tat["units"], cal)
        if cal:
            enc["calendar"] = cal
    gs = GridSeries.__new__(GridSeries)
    gs.values = PackedArray(data, decode)
    gs.dims = dims
    gs.coords = coords
    gs.attrs = {k: v for k, v in at.items() if k not in ("scale_factor", "add_offset", "_FillValue", "missing_value")}
    gs.coord_attrs = coord_attrs
    gs.time_encoding = enc
    gs._file = f
    return gs


def threshold_file(path, varname=None, **kwargs):
    """threshold() straight from a netCDF classic file (see the module docstring)."""
    from .api import threshold
    return threshold(open_series(path, varname, kwargs.get("tdim")), **kwargs)
