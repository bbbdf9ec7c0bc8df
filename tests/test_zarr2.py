"""zarr v2 directory stores (xmhw_amd/zarr2.py): the reader against stores written chunk by chunk with numpy and the
standard library (uncompressed, zlib, gzip; "." and "/" chunk keys; ragged edge chunks; an absent chunk), and the
refusal of codecs that are not available.  CPU only; the device half is in tests/test_gpu_ingest.py."""
import json
import os

import numpy as np
import numpy.testing as npt
import pytest

from xmhw_amd import XmhwException, ingest, zarr2
from xmhw_amd.device import is_packed
from ingest_oracle import decode_packed


def _store(tmp_path, compressor, sep, dtype="<i2"):
    rng = np.random.default_rng(3)
    T, ny, nx = 40, 5, 7
    raw = rng.integers(-3000, 3000, size=(T, ny, nx)).astype(dtype)
    raw[:, 0, 0] = -32768
    time = np.arange(T, dtype="<i8")
    arrays = {
        "sst": (("time", "lat", "lon"), raw, {"scale_factor": 0.01, "add_offset": 5.0, "_FillValue": -32768, "units": "degC"},
                (16, 3, 4)),
        "time": (("time",), time, {"units": "days since 2003-01-01 12:00:00", "calendar": "standard"}, None),
        "lat": (("lat",), np.linspace(-40, -36, ny).astype("<f4"), {}, None),
        "lon": (("lon",), np.linspace(140, 146, nx).astype("<f4"), {}, None),
    }
    p = str(tmp_path / "store.zarr")
    zarr2.write_store(p, arrays, compressor=compressor, dimension_separator=sep)
    return p, raw


@pytest.mark.parametrize("compressor,sep", [(None, "."), ("zlib", "."), ("gzip", "/")])
def test_reader_equals_what_was_written(tmp_path, compressor, sep):
    p, raw = _store(tmp_path, compressor, sep)
    a = zarr2.ZarrArray(os.path.join(p, "sst"))
    assert a.shape == raw.shape and a.chunks == (16, 3, 4) and a.dtype == np.dtype("<i2")
    npt.assert_array_equal(a.read(), raw)
    gs = ingest.open_series(p)                      # a directory: dispatched to the zarr reader
    assert is_packed(gs.values) and gs.dims == ("time", "lat", "lon")
    assert gs.values.decode["out"] == "float64" and gs.values.decode["fill"] == -32768.0      # JSON floats are float64
    assert gs.coords["time"][0] == np.datetime64("2003-01-01T12:00:00") and gs.time_encoding == {"calendar": "standard"}
    want = raw.astype(np.float64) * 0.01 + 5.0
    want[raw == -32768] = np.nan
    npt.assert_array_equal(decode_packed(gs.values), want)
    assert "scale_factor" not in gs.attrs and gs.attrs["units"] == "degC"


def test_big_endian_and_absent_chunk(tmp_path):
    p, raw = _store(tmp_path, None, ".", dtype=">i2")
    os.remove(os.path.join(p, "sst", "1.0.1"))          # an absent chunk reads as fill_value
    got = zarr2.ZarrArray(os.path.join(p, "sst")).read()
    want = raw.copy()
    want[16:32, 0:3, 4:7] = -32768
    npt.assert_array_equal(got, want)
    assert got.dtype == np.dtype(">i2")


def test_unavailable_codecs_and_layouts_are_refused_by_name(tmp_path):
    p, _ = _store(tmp_path, None, ".")
    meta_p = os.path.join(p, "sst", ".zarray")
    meta = json.load(open(meta_p))
    for change, word in (({"compressor": {"id": "blosc", "cname": "lz4"}}, "blosc"), ({"order": "F"}, "order"),
                         ({"filters": [{"id": "delta"}]}, "delta"), ({"zarr_format": 3}, "version 2")):
        json.dump({**meta, **change}, open(meta_p, "w"))
        with pytest.raises(XmhwException, match=word):
            ingest.open_series(p, "sst")


@pytest.mark.parametrize("dtype", ["<f4", "<f8"])
def test_float_store_with_a_fill_value_only_in_zarray(tmp_path, dtype):
    """what xarray's zarr v2 backend writes for a FLOAT variable with _FillValue=-999: the value is in .zarray's
    fill_value and nowhere else; xr.open_zarr() turns it into NaN, so must the reader (land cells would
    otherwise reach land_check as -999 data)"""
    rng = np.random.default_rng(5)
    T, ny, nx = 24, 4, 6
    x = rng.normal(15, 3, size=(T, ny, nx)).astype(dtype)
    x[:, 1, 2] = -999.0                                  # a land cell
    x[3, 0, 0] = -999.0
    arrays = {"sst": (("time", "lat", "lon"), x, {"_FillValue": -999.0, "units": "degC"}, (8, 4, 6)),
              "time": (("time",), np.arange(T, dtype="<i8"), {"units": "days since 2003-01-01", "calendar": "standard"}, None)}
    p = str(tmp_path / "f.zarr")
    zarr2.write_store(p, arrays)
    za = os.path.join(p, "sst", ".zattrs")
    at = json.load(open(za))
    del at["_FillValue"]                                 # (write_store also puts it there; xarray does not)
    json.dump(at, open(za, "w"))
    assert json.load(open(os.path.join(p, "sst", ".zarray")))["fill_value"] == -999.0
    gs = ingest.open_series(p)
    assert gs.values.decode["fill"] == -999.0 and gs.values.decode["out"] == np.dtype(dtype).name
    got = decode_packed(gs.values)
    want = x.copy()
    want[x == -999.0] = np.nan
    npt.assert_array_equal(got, want)
    # a NaN fill value (zarr's default for floats) needs no decoding
    meta = json.load(open(os.path.join(p, "sst", ".zarray")))
    meta["fill_value"] = "NaN"
    json.dump(meta, open(os.path.join(p, "sst", ".zarray"), "w"))
    vals = ingest.open_series(p).values
    if is_packed(vals):                                  # (a plain array otherwise: nothing to decode at all)
        assert vals.decode["fill"] is None
