"""The minimal netCDF classic reader / writer (xmhw_amd/netcdf3.py) against scipy.io.netcdf_file,
and the host side of the ingest path (xmhw_amd/ingest.py): CF time decoding, packing recipe."""
import numpy as np
import numpy.testing as npt
import pytest

from xmhw_amd import ingest, netcdf3
from xmhw_amd.device import is_packed
from ingest_oracle import decode_packed as decode_on_host          # numpy restatement kept under oracle/
from xmhw_amd.exception import XmhwException

scipy_io = pytest.importorskip("scipy.io")


def _packed_file(path, T=40, record=True, interleave=True, dtype=np.int16):
    rng = np.random.default_rng(3)
    if dtype == np.int16:
        data = rng.integers(-200, 3000, size=(T, 4, 5)).astype(np.int16)
        data[:, 0, 0] = -32768
        data[3, 1, 1] = -32768
        vat = {"scale_factor": np.float32(0.01), "add_offset": np.float32(5.0), "_FillValue": np.int16(-32768), "units": "degC"}
    else:
        data = rng.normal(15, 3, size=(T, 4, 5)).astype(dtype)
        data[:, 0, 0] = np.nan
        vat = {"units": "degC"}
    variables = {"lat": (("lat",), np.arange(4, dtype=np.float32), {"units": "degrees_north"}),
                 "lon": (("lon",), np.arange(5, dtype=np.float32), {}),
                 "sst": (("time", "lat", "lon"), data, vat)}
    tv = (("time",), np.arange(T, dtype=np.float64) + 0.5, {"units": "days since 2003-01-01 00:00:00", "calendar": "standard"})
    if interleave:
        variables = {"time": tv, **variables}
    else:
        variables["time"] = (("time",), tv[1], tv[2])
    netcdf3.write_classic(str(path), {"time": T, "lat": 4, "lon": 5}, variables, attrs={"title": "t"},
                          record_dim="time" if record else None)
    return data


@pytest.mark.parametrize("record,interleave", [(True, True), (False, False)])
def test_writer_and_reader_agree_with_scipy(tmp_path, record, interleave):
    p = tmp_path / "a.nc"
    data = _packed_file(p, record=record, interleave=interleave)
    g = scipy_io.netcdf_file(str(p), "r", mmap=False, maskandscale=False)
    npt.assert_array_equal(g.variables["sst"][:], data)
    npt.assert_array_equal(g.variables["time"][:], np.arange(40) + 0.5)
    assert g.variables["sst"].scale_factor == np.float32(0.01) and g.title == b"t"
    f = netcdf3.File(str(p))
    v = f.variables["sst"]
    assert v.dims == ("time", "lat", "lon") and v.shape == (40, 4, 5) and v.dtype == np.dtype(">i2")
    npt.assert_array_equal(v.data, data)
    npt.assert_array_equal(f.variables["time"].data, np.arange(40) + 0.5)
    assert v.attrs["_FillValue"] == -32768 and f.attrs["title"] == "t"
    assert (f.dimensions["time"] is None) == record


def test_reader_on_a_scipy_written_file(tmp_path):
    p = str(tmp_path / "s.nc")
    f = scipy_io.netcdf_file(p, "w", version=2)
    f.createDimension("time", None); f.createDimension("y", 3); f.createDimension("x", 4)
    v = f.createVariable("sst", "f", ("time", "y", "x")); v.units = "K"
    t = f.createVariable("time", "d", ("time",)); t.units = "hours since 1990-01-01"
    t[:] = np.arange(6.0) * 24
    v[:] = np.arange(72, dtype=np.float32).reshape(6, 3, 4)
    f.close()
    g = netcdf3.File(p)
    npt.assert_array_equal(g.variables["sst"].data, np.arange(72).reshape(6, 3, 4))
    assert g.variables["sst"].dtype == np.dtype(">f4") and g.numrecs == 6
    gs = ingest.open_series(p, "sst")
    assert gs.coords["time"][1] == np.datetime64("1990-01-02T00:00:00")


def test_hdf5_is_refused(tmp_path):
    p = tmp_path / "h.nc"
    p.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\x00" * 64)
    with pytest.raises(netcdf3.NetCDF3Error, match="netCDF-4"):
        netcdf3.File(str(p))


def test_open_series_builds_the_decoding_recipe(tmp_path):
    p = tmp_path / "p.nc"
    data = _packed_file(p)
    gs = ingest.open_series(str(p))
    assert is_packed(gs.values) and gs.values.dtype == np.dtype(">i2")
    assert gs.values.decode["out"] == "float32" and gs.values.decode["fill"] == -32768.0
    assert gs.dims == ("time", "lat", "lon") and gs.time_encoding == {"calendar": "standard"}
    assert gs.coords["time"][0] == np.datetime64("2003-01-01T12:00:00")
    assert "scale_factor" not in gs.attrs and gs.attrs["units"] == "degC"
    want = data.astype(np.float32) * np.float32(0.01) + np.float32(5.0)
    want[data == -32768] = np.nan
    got = decode_on_host(gs.values)
    assert got.dtype == np.float32
    npt.assert_array_equal(got, want)
    # the recipe survives the reshapes threshold() applies before the device sees the array
    from xmhw_amd import landmask
    st, order, shape = landmask.stack_cells(gs.values, gs.dims, "time")
    assert is_packed(st) and st.shape == (40, 20) and st.strides[1] == 2


def test_cf_time_decoding():
    t = ingest.decode_time([0, 1.5], "days since 2000-02-28 12:00:00", "proleptic_gregorian")
    assert t[0] == np.datetime64("2000-02-28T12:00:00") and t[1] == np.datetime64("2000-03-01T00:00:00")
    n = ingest.decode_time(np.arange(0, 365 * 3, 365), "days since 2001-01-01", "noleap")
    assert [x.year for x in n] == [2001, 2002, 2003] and all(x.dayofyr == 1 for x in n)
    d = ingest.decode_time([59, 60, 359, 360], "days since 2001-01-01", "360_day")
    assert (d[0].month, d[0].day) == (2, 30) and (d[1].month, d[1].day) == (3, 1)
    assert d[2].dayofyr == 360 and d[3].year == 2002 and d[3].calendar == "360_day"
    with pytest.raises(XmhwException):
        ingest.decode_time([0], "fortnights since 2000-01-01")


def test_read_rows_and_file_window(tmp_path):
    """xmhw_read_rows (pread into a dense buffer; host-only entry of the C ABI) and the window arithmetic
    that maps a view of the mapped variable back to file offsets (device._file_window)"""
    import os
    from xmhw_amd import _xmhw_hip as h
    from xmhw_amd import netcdf3
    from xmhw_amd.device import _file_window
    T, ny, nx = 9, 5, 7
    data = np.arange(T * ny * nx, dtype=np.float32).reshape(T, ny, nx)
    other = np.arange(T, dtype=np.float64)            # a second record variable: rows of `sst` become pitched
    path = str(tmp_path / "two_records.nc")
    netcdf3.write_classic(path, {"time": T, "y": ny, "x": nx},
                          {"sst": (("time", "y", "x"), data, {}), "aux": (("time",), other, {})}, record_dim="time")
    f = netcdf3.File(path)
    v = np.asarray(f.variables["sst"].data).reshape(T, ny * nx)
    assert v.strides[0] > ny * nx * 4                 # interleaved with `aux`
    file = dict(fd=f.fileno(), address=f.map_address, length=f.map_length)
    view = v[:, 3:20]
    fd, off, pitch = _file_window(view, file)
    dst = np.zeros(view.shape, dtype=">f4")
    h.read_rows(fd, off, pitch, view.shape[1] * 4, T, dst.ctypes.data)
    npt.assert_array_equal(dst, view)
    npt.assert_array_equal(dst.astype(np.float32), data.reshape(T, -1)[:, 3:20])
    # a part of the rows, as the staging threads ask for them
    part = np.zeros((4, view.shape[1]), dtype=">f4")
    h.read_rows(fd, off + 2 * pitch, pitch, view.shape[1] * 4, 4, part.ctypes.data)
    npt.assert_array_equal(part, view[2:6])
    # views that are not windows of the file are refused (the copy path takes them)
    assert _file_window(np.ascontiguousarray(view), file) is None
    assert _file_window(view[:, ::2], file) is None
    big = np.zeros((T + 50, view.shape[1]), ">f4")     # (kept in a name: the callee writes through the raw address)
    with pytest.raises(h.InvalidArgument):
        h.read_rows(fd, off, pitch, view.shape[1] * 4, T + 50, big.ctypes.data)
    f.close()
