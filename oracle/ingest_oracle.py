"""CF decoding of a stored netCDF variable, restated in numpy.  TEST INFRASTRUCTURE ONLY (see oracle/__init__ and
DESIGN.md section 4): the product decodes on the device (xmhw_amd/csrc/kernels_ingest.hip: xmhw_decode).

What is restated: xarray's CF decoding as the reference's users get it from ``xr.open_dataset(...)["sst"]``
(docs/gettingstarted.rst:30-33) -- ``raw * scale_factor + add_offset`` in float32 when the packing attributes are
float32 and in float64 otherwise, ``_FillValue`` / ``missing_value`` -> NaN -- which is what decides which cells
land_check() drops (xmhw/identify.py:520-528).  tests/test_gpu_ingest.py additionally pins the device decoder to
scipy.io.netcdf_file(maskandscale=True), an independent reader and decoder of the same files.
"""
import numpy as np


def decode_cf(raw, decode):
    """raw: the stored array (any byte order); decode: dict(scale, offset, fill, out) as built by
    xmhw_amd.ingest.open_series()"""
    raw = np.asarray(raw)
    out_t = np.dtype(decode["out"]).type
    out = raw.astype(np.dtype(decode["out"]))
    if decode.get("scale") is not None:
        out = out * out_t(decode["scale"]) + out_t(decode.get("offset") or 0.0)
    if decode.get("fill") is not None:
        out[raw == raw.dtype.type(decode["fill"])] = np.nan
    return out


def decode_packed(a):
    """a: xmhw_amd.device.PackedArray"""
    return decode_cf(np.asarray(a), a.decode)
