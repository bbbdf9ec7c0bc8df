"""zarr v2 directory stores -> device ingest ("netCDF/zarr").

The reference leaves reading to xarray (``xr.open_zarr(store)["sst"]``; docs/gettingstarted.rst:30-33 shows the
netCDF twin).  This is a reader of the zarr v2 layout itself -- no zarr / numcodecs package is needed, none is in
the image: ``<store>/<array>/.zarray`` (shape, chunks, dtype, compressor, fill_value, order, dimension_separator),
``.zattrs`` (xarray's ``_ARRAY_DIMENSIONS`` and the CF packing attributes) and one file per chunk.  Supported:
C order, no filters, compressor null / zlib / gzip (the standard library inflates those); anything else (blosc,
zstd, lz4: codecs that are not in this image) is refused by name.  The chunks are assembled ONCE into the stored
dtype and byte order on the host; byte swap, CF unpacking and the fill value are left to the device decoder
exactly as for a netCDF file, so an int16-packed store crosses PCIe at 2 bytes per sample.
(The assembled raw array lives in host memory: 30 GB for 40 years of 0.25 degree int16 data.)
"""
import json
import os
import zlib

import numpy as np

from .api import GridSeries
from .device import PackedArray
from .exception import XmhwException


class ZarrArray:
    def __init__(self, path):
        self.path = path
        try:
            with open(os.path.join(path, ".zarray")) as fh:
                meta = json.load(fh)
        except OSError as e:
            raise XmhwException(f"{path}: not a zarr v2 array ({e})")
        if meta.get("zarr_format") != 2:
            raise XmhwException(f"{path}: zarr_format {meta.get('zarr_format')!r}; only version 2 stores are read")
        if meta.get("order", "C") != "C":
            raise XmhwException(f"{path}: order {meta.get('order')!r}; only C order is read")
        if meta.get("filters"):
            raise XmhwException(f"{path}: filters {[f.get('id') for f in meta['filters']]} are not supported")
        comp = meta.get("compressor")
        self.codec = None if comp is None else comp.get("id")
        if self.codec not in (None, "zlib", "gzip"):
            raise XmhwException(f"{path}: compressor {self.codec!r} is not available here (null, zlib and gzip are)")
        self.shape = tuple(int(x) for x in meta["shape"])
        self.chunks = tuple(int(x) for x in meta["chunks"])
        self.dtype = np.dtype(meta["dtype"])
        self.fill_value = meta.get("fill_value")
        self.sep = meta.get("dimension_separator", ".")
        self.attrs = {}
        za = os.path.join(path, ".zattrs")
        if os.path.exists(za):
            with open(za) as fh:
                self.attrs = json.load(fh)

    def _fill(self):
        fv = self.fill_value
        if fv is None:
            return 0
        if isinstance(fv, str):
            return {"NaN": np.nan, "Infinity": np.inf, "-Infinity": -np.inf}.get(fv, 0)
        return fv

    def read(self):
        """the whole array in its stored dtype / byte order"""
        out = np.empty(self.shape, dtype=self.dtype)
        if out.size == 0:
            return out
        if not self.shape:                      # a 0-d array: one chunk named "0"
            grid = [()]
        else:
            counts = [-(-s // c) for s, c in zip(self.shape, self.chunks)]
            grid = np.ndindex(*counts)
        for idx in grid:
            name = self.sep.join(str(i) for i in idx) if idx else "0"
            sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, self.chunks, self.shape))
            fn = os.path.join(self.path, *name.split("/")) if self.sep == "/" else os.path.join(self.path, name)
            if not os.path.exists(fn):
                out[sel] = self._fill()         # an absent chunk is all fill_value
                continue
            with open(fn, "rb") as fh:
                raw = fh.read()
            if self.codec == "zlib":
                raw = zlib.decompress(raw)
            elif self.codec == "gzip":
                raw = zlib.decompress(raw, 16 + zlib.MAX_WBITS)
            chunk = np.frombuffer(raw, dtype=self.dtype)
            want = int(np.prod(self.chunks)) if self.chunks else 1
            if chunk.size != want:
                raise XmhwException(f"{fn}: {chunk.size} items, the chunk shape {self.chunks} holds {want}")
            chunk = chunk.reshape(self.chunks)
            out[sel] = chunk[tuple(slice(0, s.stop - s.start) for s in sel)]
        return out


def open_series(store, varname=None, tdim=None):
    """A (time, y, x) array of a zarr v2 directory store as a GridSeries over a PackedArray of the stored samples
    (xmhw_amd.ingest.open_series dispatches here for directories)."""
    from .ingest import cf_recipe, decode_time
    if os.path.exists(os.path.join(store, ".zarray")):           # the path names the array itself
        store, varname = os.path.dirname(os.path.abspath(store)), os.path.basename(os.path.abspath(store))
    names = sorted(d for d in os.listdir(store) if os.path.exists(os.path.join(store, d, ".zarray")))
    arrays = {n: ZarrArray(os.path.join(store, n)) for n in names}
    cands = [n for n, a in arrays.items() if len(a.shape) >= 2]
    if varname is None:
        if len(cands) != 1:
            raise XmhwException(f"{store}: name the variable, candidates {cands}")
        varname = cands[0]
    if varname not in arrays:
        raise XmhwException(f"{store}: no array {varname!r}")
    var = arrays[varname]
    dims = tuple(var.attrs.get("_ARRAY_DIMENSIONS") or [f"dim_{i}" for i in range(len(var.shape))])
    tdim = tdim or dims[0]
    if dims[0] != tdim:
        raise XmhwException(f"{store}: {varname} must have {tdim!r} as its first (slowest) dimension, has {dims}")
    at = dict(var.attrs)
    at.pop("_ARRAY_DIMENSIONS", None)
    if "_FillValue" not in at and var.fill_value is not None:
        # xarray's zarr v2 backend keeps the CF fill value of EVERY dtype in .zarray's fill_value (a float store
        # written with _FillValue=-999 has it nowhere else): land cells must reach the device as NaN, not as -999.
        # A NaN fill value needs no decoding.
        fv = var.fill_value
        if isinstance(fv, str):                         # zarr's JSON spelling of the non-finite floats
            fv = {"NaN": float("nan"), "Infinity": float("inf"), "-Infinity": float("-inf")}.get(fv)
        if fv is not None and not (isinstance(fv, float) and fv != fv):
            at["_FillValue"] = fv
    for k in ("scale_factor", "add_offset"):
        # JSON has one float type: the packing attributes of a zarr store are float64 unless the store says otherwise
        if k in at and not isinstance(at[k], float):
            at[k] = float(at[k])
    decode = cf_recipe(f"{store}: {varname}", var.dtype, at)
    coords, coord_attrs = {}, {}
    for i, d in enumerate(dims):
        if d in arrays and len(arrays[d].shape) == 1:
            cv = arrays[d]
            coords[d] = cv.read().astype(cv.dtype.newbyteorder("="))
            coord_attrs[d] = {k: v for k, v in cv.attrs.items() if k != "_ARRAY_DIMENSIONS"}
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
    gs.values = PackedArray(var.read(), decode)
    gs.dims = dims
    gs.coords = coords
    gs.attrs = {k: v for k, v in at.items() if k not in ("scale_factor", "add_offset", "_FillValue", "missing_value")}
    gs.coord_attrs = coord_attrs
    gs.time_encoding = enc
    return gs


def write_store(store, arrays, compressor=None, dimension_separator="."):
    """Minimal zarr v2 writer (tests, examples): arrays = {name: (dims, ndarray, attrs, chunks or None)}."""
    os.makedirs(store, exist_ok=True)
    with open(os.path.join(store, ".zgroup"), "w") as fh:
        json.dump({"zarr_format": 2}, fh)
    for name, (dims, a, attrs, chunks) in arrays.items():
        a = np.asarray(a)
        chunks = tuple(chunks or a.shape)
        d = os.path.join(store, name)
        os.makedirs(d, exist_ok=True)
        fill = attrs.get("_FillValue")
        meta = {"zarr_format": 2, "shape": list(a.shape), "chunks": list(chunks), "dtype": a.dtype.str, "order": "C",
                "compressor": None if compressor is None else {"id": compressor, "level": 1}, "filters": None,
                "fill_value": None if fill is None else (float(fill) if a.dtype.kind == "f" else int(fill)),
                "dimension_separator": dimension_separator}
        with open(os.path.join(d, ".zarray"), "w") as fh:
            json.dump(meta, fh)
        with open(os.path.join(d, ".zattrs"), "w") as fh:
            json.dump({"_ARRAY_DIMENSIONS": list(dims), **{k: (v.item() if hasattr(v, "item") else v) for k, v in attrs.items()
                                                          if k != "_FillValue" or a.dtype.kind != "i"}}, fh)
        counts = [-(-s // c) for s, c in zip(a.shape, chunks)]
        for idx in np.ndindex(*counts):
            sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, a.shape))
            block = np.zeros(chunks, dtype=a.dtype)
            part = a[sel]
            block[tuple(slice(0, n) for n in part.shape)] = part
            raw = block.tobytes()
            if compressor == "zlib":
                raw = zlib.compress(raw, 1)
            elif compressor == "gzip":
                co = zlib.compressobj(1, zlib.DEFLATED, 16 + zlib.MAX_WBITS)
                raw = co.compress(raw) + co.flush()
            name_ = dimension_separator.join(str(i) for i in idx)
            fn = os.path.join(d, *name_.split("/")) if dimension_separator == "/" else os.path.join(d, name_)
            os.makedirs(os.path.dirname(fn), exist_ok=True)
            with open(fn, "wb") as fh:
                fh.write(raw)
