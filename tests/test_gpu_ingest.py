"""File -> device ingest: the raw bytes of a netCDF classic variable are decoded on
the device (byte order, CF packing, fill value); threshold() / detect() on the mapped file must equal
the same calls on the host-decoded array bit for bit (land mask semantics xmhw/identify.py:520-528)."""
import os

import numpy as np
import numpy.testing as npt
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def fixture():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    g = np.load(os.path.join(ROOT, "tests", "golden", "oisst_2003_2004.npz"))
    return g["sst"], g["lat"], g["lon"], g["time"]


def _write(path, sst, lat, lon, time, kind, interleave=True):
    from xmhw_amd import netcdf3
    if kind == "i16":
        packed = np.where(np.isnan(sst), -32768, np.round((sst - 10.0) / 0.01)).astype(np.int16)
        var = (("time", "lat", "lon"), packed, {"scale_factor": np.float32(0.01), "add_offset": np.float32(10.0),
                                                 "_FillValue": np.int16(-32768), "units": "degC"})
    elif kind == "i16_f64attrs":
        packed = np.where(np.isnan(sst), -32768, np.round((sst - 10.0) / 0.005)).astype(np.int16)
        var = (("time", "lat", "lon"), packed, {"scale_factor": 0.005, "add_offset": 10.0, "_FillValue": np.int16(-32768)})
    elif kind == "f32":
        var = (("time", "lat", "lon"), sst.astype(np.float32), {"units": "degC"})
    else:
        var = (("time", "lat", "lon"), sst.astype(np.float64) + 1e-9, {})
    tv = (("time",), time.astype(np.float64), {"units": "days since 2003-01-01 12:00:00", "calendar": "proleptic_gregorian"})
    variables = {"lat": (("lat",), lat.astype(np.float32), {}), "lon": (("lon",), lon.astype(np.float32), {}), "sst": var}
    variables = {"time": tv, **variables} if interleave else {**variables, "time": tv}
    netcdf3.write_classic(str(path), {"time": sst.shape[0], "lat": sst.shape[1], "lon": sst.shape[2]}, variables,
                          record_dim="time" if interleave else None)


@pytest.mark.parametrize("kind,interleave", [("i16", True), ("i16", False), ("i16_f64attrs", True), ("f32", True), ("f64", False)])
def test_threshold_and_detect_from_file_equal_host_decoded(tmp_path, fixture, kind, interleave):
    import xmhw_amd
    from xmhw_amd import GridSeries, climatology_series, ingest
    from ingest_oracle import decode_packed          # numpy restatement (oracle/), not product code
    sst, lat, lon, time = fixture
    p = tmp_path / "f.nc"
    _write(p, sst, lat, lon, time, kind, interleave)
    temp = ingest.open_series(str(p), "sst")
    host = decode_packed(temp.values)
    assert host.dtype == (np.float64 if kind in ("f64", "i16_f64attrs") else np.float32)
    ref_in = GridSeries(host, temp.dims, temp.coords, time_encoding=temp.time_encoding)
    ref = xmhw_amd.threshold(ref_in)
    got = xmhw_amd.threshold(temp)
    npt.assert_array_equal(got["thresh"], ref["thresh"])
    npt.assert_array_equal(got["seas"], ref["seas"])
    npt.assert_array_equal(got.coords["lat"], ref.coords["lat"])
    got2 = ingest.threshold_file(str(p), "sst")
    npt.assert_array_equal(got2["thresh"], ref["thresh"])
    th, se = climatology_series(ref, "thresh"), climatology_series(ref, "seas")
    m0 = xmhw_amd.detect(ref_in, th, se)
    m1 = xmhw_amd.detect(temp, th, se)
    assert m0.n_events > 0
    npt.assert_array_equal(m1.table, m0.table)
    npt.assert_array_equal(m1.offsets, m0.offsets)
    npt.assert_array_equal(m1.keep, m0.keep)


def test_pipelined_slabs_equal_one_slab(tmp_path, fixture):
    """the slab pipeline (next upload + decode in a second thread, pitched device-to-host placement of
    the results) against a single-slab run, on a wider grid with land"""
    from xmhw_amd import ingest, landmask
    from xmhw_amd.device import calc_clim_grid_device
    sst, lat, lon, time = fixture
    wide = np.tile(sst, (1, 1, 9)) + np.linspace(0, 1, 36, dtype=np.float32)[None, None, :]
    p = tmp_path / "w.nc"
    _write(p, wide, lat, np.arange(36.0), time, "i16")
    temp = ingest.open_series(str(p), "sst")
    st, _, _ = landmask.stack_cells(temp.values, temp.dims, "time")
    import xmhw_oracle as ora
    doy = ora.add_doy(temp.coords["time"])
    one = calc_clim_grid_device(st, doy, False, 90, 5, True, 31, False)
    per_cell = st.shape[0] * (2 + 2 * 4 + 2) + 4 * 366 * 8
    many = calc_clim_grid_device(st, doy, False, 90, 5, True, 31, False, max_batch_bytes=per_cell * 50)
    for a, b in zip(one, many):
        npt.assert_array_equal(a, b)
    compact = calc_clim_grid_device(st, doy, False, 90, 5, True, 31, False, max_batch_bytes=per_cell * 50, scatter=False)
    npt.assert_array_equal(compact[2], one[2][:, one[0]])


def test_decode_kernel_against_numpy(fixture):
    """xmhw_decode on its own: every stored/decoded pair, both byte orders, fill values"""
    from xmhw_amd.device import DeviceBuffer, hip
    h = hip()
    rng = np.random.default_rng(5)
    rows, cols = 37, 301
    for raw_dt, out_dt, scale, offset, fill in [(">i2", np.float32, 0.01, 3.5, -32768), ("<i2", np.float32, 0.25, -1.0, 7),
                                                (">i2", np.float64, 0.001, 20.0, -1), (">f4", np.float32, None, None, None),
                                                ("<f4", np.float32, 2.0, 1.0, -999.0), (">f8", np.float64, None, None, -999.0)]:
        dt = np.dtype(raw_dt)
        if dt.kind == "i":
            raw = rng.integers(-32768, 32767, size=(rows, cols)).astype(dt)
        else:
            raw = rng.normal(0, 50, size=(rows, cols)).astype(dt)
        if fill is not None:
            raw[::5, ::7] = fill
        want = raw.astype(out_dt)
        if scale is not None:
            want = want * out_dt(scale) + out_dt(offset)
        if fill is not None:
            want[raw == dt.type(fill)] = np.nan
        d_in, d_out = DeviceBuffer.from_array(raw), DeviceBuffer(np.dtype(out_dt).itemsize * rows * cols)
        h.decode(d_in.ptr, dt.itemsize, int(dt.byteorder == ">"), rows, cols, cols, d_out.ptr, np.dtype(out_dt).itemsize, cols,
                 scale is not None, float(scale or 1.0), float(offset or 0.0), fill is not None, float(fill or 0.0))
        h.stream_sync(0)
        got = d_out.to_array((rows, cols), out_dt)
        npt.assert_array_equal(got, want, err_msg=str((raw_dt, out_dt)))
        d_in.free(); d_out.free()


@pytest.mark.parametrize("kind", ["i16", "i16_f64attrs", "f32", "f64"])
def test_device_decode_against_scipy_netcdf(tmp_path, fixture, kind):
    """An INDEPENDENT pin of the decode semantics: scipy.io.netcdf_file(maskandscale=True) parses the same file
    with its own header reader and applies scale / offset / fill on its own; the device decoder (reached through
    the product's own upload path) must agree -- bit for bit where scipy computes in the same precision
    (float64 packing attributes, unpacked floats), to one float32 rounding where scipy decodes in float64 and
    xarray / the device in float32 (float32 attributes); the fill mask must be identical."""
    from scipy.io import netcdf_file
    from xmhw_amd import ingest
    from xmhw_amd.device import decode_through_device
    from xmhw_amd import landmask
    sst, lat, lon, time = fixture
    p = tmp_path / "s.nc"
    _write(p, sst, lat, lon, time, kind, True)
    with netcdf_file(str(p), "r", mmap=False, maskandscale=True) as f:
        want = f.variables["sst"][:]
        want_mask = np.ma.getmaskarray(want).copy()
        want = np.ma.filled(want.astype(np.float64), np.nan)
    temp = ingest.open_series(str(p), "sst")
    stacked, _, _ = landmask.stack_cells(temp.values, temp.dims, "time")
    got = decode_through_device(stacked).reshape(want.shape)
    if kind in ("i16", "i16_f64attrs"):
        npt.assert_array_equal(np.isnan(got), want_mask)
    ok = ~np.isnan(got)
    npt.assert_array_equal(np.isnan(got), np.isnan(want))
    if kind == "i16":
        assert got.dtype == np.float32
        npt.assert_allclose(got[ok], want[ok], rtol=1.2e-7, atol=0)       # one float32 rounding
    else:
        npt.assert_array_equal(got[ok].astype(np.float64), want[ok])


@pytest.mark.parametrize("compressor,dtype", [(None, "<i2"), ("zlib", ">i2"), ("gzip", "<f4")])
def test_threshold_from_a_zarr_store_equals_host_decoded(tmp_path, fixture, compressor, dtype):
    """zarr v2 directory store (xmhw_amd/zarr2.py): chunked along every axis, packed int16 of either byte order or
    plain float32; threshold() on the store equals threshold() on the array the oracle's numpy decoder yields"""
    import xmhw_amd
    from xmhw_amd import GridSeries, ingest, zarr2
    from ingest_oracle import decode_packed
    sst, lat, lon, time = fixture
    if dtype.endswith("i2"):
        raw = np.where(np.isnan(sst), -32768, np.round((sst - 10.0) / 0.01)).astype(dtype)
        attrs = {"scale_factor": 0.01, "add_offset": 10.0, "_FillValue": -32768}
    else:
        raw, attrs = sst.astype(dtype), {}
    store = str(tmp_path / "sst.zarr")
    zarr2.write_store(store, {
        "sst": (("time", "lat", "lon"), raw, attrs, (100, 3, 3)),
        "time": (("time",), time.astype("<f8"), {"units": "days since 2003-01-01 12:00:00", "calendar": "proleptic_gregorian"}, None),
        "lat": (("lat",), lat.astype("<f4"), {}, None), "lon": (("lon",), lon.astype("<f4"), {}, None)},
        compressor=compressor)
    temp = ingest.open_series(store, "sst")
    host = decode_packed(temp.values)
    ref = xmhw_amd.threshold(GridSeries(host, temp.dims, temp.coords, time_encoding=temp.time_encoding))
    got = xmhw_amd.threshold(temp)
    npt.assert_array_equal(got["thresh"], ref["thresh"])
    npt.assert_array_equal(got["seas"], ref["seas"])
    npt.assert_array_equal(got.coords["lat"], ref.coords["lat"])


def test_threshold_from_the_reference_netcdf4_fixture(fixture):
    """netCDF-4 / HDF5 (xmhw_amd/hdf5min.py): threshold() straight from the reference's own fixture file -- chunked,
    shuffled, deflated float32 with dense attribute storage -- equals threshold() on the array extracted from it"""
    import xmhw_amd
    from xmhw_amd import ingest
    sst, lat, lon, time = fixture
    p = os.path.join(ROOT, "tests", "golden", "ref_testdata", "oisst_2003_2004.nc")
    got = ingest.threshold_file(p, "sst")
    t = np.datetime64("2003-01-01T12:00:00") + time.astype("timedelta64[D]")
    ref = xmhw_amd.threshold_array(sst, t, dims=("time", "lat", "lon"), coords={"lat": lat, "lon": lon},
                                   calendar="proleptic_gregorian")
    npt.assert_array_equal(got["thresh"], ref["thresh"])
    npt.assert_array_equal(got["seas"], ref["seas"])
    npt.assert_array_equal(got.coords["lat"], ref.coords["lat"])
    # the all-land fixture raises like land_check() (identify.py:527-528)
    with pytest.raises(xmhw_amd.XmhwException):
        ingest.threshold_file(os.path.join(ROOT, "tests", "golden", "ref_testdata", "land.nc"), "sst")


@pytest.mark.parametrize("var", ["sst_be_contig", "sst_le_chunked", "f32_chunked"])
def test_threshold_from_h5py_written_netcdf4_layouts(var):
    """packed int16 of both byte orders (contiguous: zero-copy window + pread upload; chunked + shuffle + deflate +
    fletcher32: inflated once on the host), device decode, against the oracle's numpy decoder"""
    import xmhw_amd
    from xmhw_amd import GridSeries, ingest
    from ingest_oracle import decode_packed
    p = os.path.join(ROOT, "tests", "golden", "hdf5", "packed_earliest.h5")
    temp = ingest.open_series(p, var)
    host = decode_packed(temp.values)
    ref = xmhw_amd.threshold(GridSeries(host, temp.dims, temp.coords, time_encoding=temp.time_encoding), smoothPercentileWidth=11)
    got = xmhw_amd.threshold(temp, smoothPercentileWidth=11)
    npt.assert_array_equal(got["thresh"], ref["thresh"])
    npt.assert_array_equal(got["seas"], ref["seas"])
