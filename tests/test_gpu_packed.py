"""int16-packed input read in place (xmhw_clim_raw_i16): the sorted-list kernel, its recomputation and
the leftover kernel take the CODES and the CF recipe (scale_factor, add_offset, _FillValue; what xr.open_dataset() applies
before the reference's threshold() sees the data, docs/gettingstarted.rst:30-33) instead of a decoded copy of the series.

float32 decode (float32 attributes, the packing of tests/golden/hdf5/packed_earliest.h5: 0.01 / 10.0 / -32768, both byte
orders): bit-identical, thresh AND seas, to xmhw_decode + the float32 path.  float64 decode (float64 attributes): the
kernel selects on the codes and decodes the two selected codes in float64 -- thresh bit-identical to the float64 path on
the decoded series and to the oracle on numpy's decode; seas = decode(exact mean of the codes), within rounding.
"""
import numpy as np
import numpy.testing as npt
import pytest

import xmhw_oracle as ora
import oracle_fast as fast

pytestmark = pytest.mark.gpu

FILL = -32768


@pytest.fixture(scope="module")
def dev():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.device as d
    return d


def _daily(y0, y1, start=None, stop=None):
    time = np.arange(start or f"{y0}-01-01", stop or f"{y1 + 1}-01-01", dtype="datetime64[D]")
    return ora.add_doy(time)


def _codes(T, C, seed, scale=0.01, offset=10.0, fillfrac=0.0, amp=(2, 10)):
    """an SST-like series stored as int16 codes: round((x - offset) / scale)"""
    rng = np.random.default_rng(seed)
    t = np.arange(T)[:, None]
    x = 15.0 + rng.uniform(*amp, C) * np.sin(2 * np.pi * (t - rng.uniform(0, 365, C)) / 365.25) \
        + 0.0005 * t * rng.uniform(-1, 1, C) + rng.normal(size=(T, C))
    codes = np.clip(np.rint((x - offset) / scale), -32767, 32767).astype(np.int16)
    if fillfrac:
        codes[rng.random((T, C)) < fillfrac] = FILL
    return codes


def _run(dev, doy, C, fn, kernel="auto", layout=None):
    plan = dev.Plan(doy, 5, kernel=kernel, layout=layout)
    th, se = dev.DeviceBuffer(8 * plan.D * C), dev.DeviceBuffer(8 * plan.D * C)
    try:
        fn(plan, th, se)
        dev.hip().stream_sync(0)
        return th.to_array((plan.D, C), np.float64), se.to_array((plan.D, C), np.float64)
    finally:
        th.free(); se.free(); plan.destroy()


def _packed(dev, codes, doy, q, negate, scale, offset, fill, decoded, big_endian=False):
    stored = codes.astype(">i2") if big_endian else codes
    d_codes = dev.DeviceBuffer.from_array(np.ascontiguousarray(stored).view(np.int16))
    try:
        return _run(dev, doy, codes.shape[1],
                    lambda plan, th, se: dev.clim_raw_packed(plan, d_codes, codes.shape[1], q, negate, th, se, scale_factor=scale,
                                                             add_offset=offset, fill=fill, decoded=decoded, big_endian=big_endian))
    finally:
        d_codes.free()


def _decoded_on_device(dev, codes, scale, offset, fill, out):
    """xmhw_decode: the series the float paths would be handed"""
    h = dev.hip()
    T, C = codes.shape
    d_raw = dev.DeviceBuffer.from_array(codes)
    isz = np.dtype(out).itemsize
    d_out = dev.DeviceBuffer(isz * T * C)
    h.decode(d_raw.ptr, 2, 0, T, C, C, d_out.ptr, isz, C, scale is not None or offset is not None,
             1.0 if scale is None else scale, 0.0 if offset is None else offset, fill is not None, 0.0 if fill is None else fill, 0)
    h.stream_sync(0)
    d_raw.free()
    return d_out


def _float_path(dev, codes, doy, q, negate, scale, offset, fill, out, kernel="auto", layout=None):
    d_ts = _decoded_on_device(dev, codes, scale, offset, fill, out)
    try:
        return _run(dev, doy, codes.shape[1],
                    lambda plan, th, se: dev.clim_raw(plan, d_ts, np.dtype(out).itemsize, codes.shape[1], q, negate, th, se),
                    kernel=kernel, layout=layout)
    finally:
        d_ts.free()


@pytest.mark.parametrize("big_endian", [False, True])
@pytest.mark.parametrize("negate", [False, True])
def test_float32_decode_is_bit_identical_to_decode_plus_float32_path(dev, big_endian, negate):
    doy = _daily(1982, 2021)
    codes = _codes(doy.shape[0], 77, 3, fillfrac=0.01)
    codes[:, 5] = FILL                                  # land
    codes[2000:9000, 9] = FILL
    scale, offset = float(np.float32(0.01)), float(np.float32(10.0))
    tp, sp = _packed(dev, codes, doy, 0.9, negate, scale, offset, FILL, "float32", big_endian)
    tf, sf = _float_path(dev, codes, doy, 0.9, negate, scale, offset, FILL, np.float32)
    npt.assert_array_equal(tp, tf)
    npt.assert_array_equal(sp, sf)
    tg, sg = _float_path(dev, codes, doy, 0.9, negate, scale, offset, FILL, np.float32, kernel="generic")
    npt.assert_array_equal(tp, tg)
    npt.assert_allclose(sp, sg, rtol=1e-12, atol=1e-300, equal_nan=True)
    assert np.isnan(tp[:, 5]).all() and np.isfinite(tp[:, 6]).all()


@pytest.mark.parametrize("negate", [False, True])
@pytest.mark.parametrize("scale,offset", [(0.01, 10.0), (0.0021973, 273.15), (-0.01, 40.0)])
def test_float64_decode_selects_on_codes_and_decodes_two_values(dev, negate, scale, offset):
    """float64 packing attributes: xarray hands the reference float64 samples code * scale + offset.  The packed path is
    bit-identical to that series on the generic float64 kernel (thresh) and to the oracle on numpy's decode; a negative
    scale_factor reverses the order of the codes"""
    doy = _daily(1991, 2020)
    codes = _codes(doy.shape[0], 50, 7, scale=abs(scale), offset=10.0, fillfrac=0.02)
    tp, sp = _packed(dev, codes, doy, 0.9, negate, scale, offset, FILL, "float64")
    tg, sg = _float_path(dev, codes, doy, 0.9, negate, scale, offset, FILL, np.float64, kernel="generic")
    npt.assert_array_equal(tp, tg)
    npt.assert_allclose(sp, sg, rtol=1e-12, atol=1e-12, equal_nan=True)
    x = codes.astype(np.float64) * scale + offset
    x[codes == FILL] = np.nan
    _, th, se = fast.raw_clim(-x if negate else x, doy, 0.9, 5)
    npt.assert_array_equal(tp, th)
    npt.assert_allclose(sp, se, rtol=1e-12, atol=1e-12)


def test_no_packing_attributes_and_no_fill(dev):
    doy = _daily(2001, 2020)
    codes = _codes(doy.shape[0], 33, 11, scale=0.01, offset=0.0)
    tp, sp = _packed(dev, codes, doy, 0.95, False, None, None, None, "float32")
    tf, sf = _float_path(dev, codes, doy, 0.95, False, None, None, None, np.float32)
    npt.assert_array_equal(tp, tf)
    npt.assert_array_equal(sp, sf)


def test_flagged_rows_are_recomputed_from_the_codes(dev):
    """a steep seasonal cycle overflows the row-lists: the recomputation (runs of rows, one wave each) and, past the work
    list's capacity, the leftover kernel read the codes through the same recipe"""
    doy = _daily(1982, 2021)
    codes = _codes(doy.shape[0], 48, 13, scale=0.01, offset=10.0, amp=(20, 30), fillfrac=0.005)
    for decoded, out in (("float32", np.float32), ("float64", np.float64)):
        tp, sp = _packed(dev, codes, doy, 0.9, False, 0.01, 10.0, FILL, decoded)
        tg, sg = _float_path(dev, codes, doy, 0.9, False, 0.01, 10.0, FILL, out, kernel="generic")
        npt.assert_array_equal(tp, tg)
        npt.assert_allclose(sp, sg, rtol=1e-12, atol=1e-12, equal_nan=True)


@pytest.mark.parametrize("years", [9, 17, 24, 33, 48])
def test_record_lengths(dev, years):
    doy = _daily(1975, 1975 + years - 1)
    codes = _codes(doy.shape[0], 70, 40 + years, fillfrac=0.01 if years % 2 else 0.0)
    tp, sp = _packed(dev, codes, doy, 0.9, False, float(np.float32(0.01)), float(np.float32(10.0)), FILL, "float32")
    tf, sf = _float_path(dev, codes, doy, 0.9, False, float(np.float32(0.01)), float(np.float32(10.0)), FILL, np.float32)
    npt.assert_array_equal(tp, tf)
    npt.assert_array_equal(sp, sf)


def test_partial_years_and_ties(dev):
    """coarse codes (0.5 K: a pool of 440 samples holds ~20 distinct values), a record from September to March"""
    doy = _daily(0, 0, start="1982-09-01", stop="2021-03-15")
    codes = _codes(doy.shape[0], 40, 23, scale=0.5, offset=0.0)
    tp, sp = _packed(dev, codes, doy, 0.9, False, 0.5, 0.0, FILL, "float64")
    tg, sg = _float_path(dev, codes, doy, 0.9, False, 0.5, 0.0, FILL, np.float64, kernel="generic")
    npt.assert_array_equal(tp, tg)
    npt.assert_allclose(sp, sg, rtol=1e-12, atol=1e-12, equal_nan=True)


def test_plans_the_sorted_kernel_does_not_serve_are_refused(dev):
    from xmhw_amd.exception import XmhwException
    doy = _daily(1982, 2021)
    codes = _codes(doy.shape[0], 8, 1)
    with pytest.raises(XmhwException, match="sorted-list kernel"):
        _packed(dev, codes, doy, 0.5, False, 0.01, 10.0, FILL, "float32")            # a median: the ring kernel's
    long = _daily(1950, 2019)
    with pytest.raises(XmhwException, match="sorted-list kernel"):
        _packed(dev, _codes(long.shape[0], 8, 2), long, 0.9, False, 0.01, 10.0, FILL, "float32")     # 70 tracks
    with pytest.raises(XmhwException):
        _packed(dev, codes, doy, 0.9, False, 0.0, 10.0, FILL, "float64")             # scale_factor 0


@pytest.mark.parametrize("kind", ["i16_f32attrs", "i16_f64attrs"])
@pytest.mark.parametrize("cold", [False, True])
def test_threshold_from_a_packed_file_reads_the_codes_in_place(tmp_path, dev, monkeypatch, kind, cold):
    """threshold() on the mapped netCDF file of a 12-year packed archive: the codes are what is uploaded, masked (land =
    every code the fill code), compacted and read by the kernels; the result equals threshold() on the host-decoded
    array (float32 attributes: bit for bit; float64 attributes: thresh bit for bit, seas within rounding)"""
    import xmhw_amd
    import xmhw_amd.device as device
    from xmhw_amd import GridSeries, ingest, netcdf3
    from ingest_oracle import decode_packed          # numpy restatement (oracle/), not product code
    time = np.arange("2001-01-01", "2013-01-01", dtype="datetime64[D]")
    T, ny, nx = time.shape[0], 6, 8
    rng = np.random.default_rng(5)
    t = np.arange(T)[:, None, None]
    sst = 15.0 + rng.uniform(2, 8, (ny, nx)) * np.sin(2 * np.pi * (t - rng.uniform(0, 365, (ny, nx))) / 365.25) + rng.normal(size=(T, ny, nx))
    sst[:, 2, 3] = np.nan                                # land
    sst[:, 5, :2] = np.nan
    sst[rng.random(sst.shape) < 0.002] = np.nan          # holes (the default anynans=False keeps those cells)
    if kind == "i16_f32attrs":
        at = {"scale_factor": np.float32(0.01), "add_offset": np.float32(10.0), "_FillValue": np.int16(-32768)}
        packed = np.where(np.isnan(sst), -32768, np.round((sst - 10.0) / 0.01)).astype(np.int16)
    else:
        at = {"scale_factor": 0.005, "add_offset": 10.0, "_FillValue": np.int16(-32768)}
        packed = np.where(np.isnan(sst), -32768, np.round((sst - 10.0) / 0.005)).astype(np.int16)
    p = tmp_path / "packed.nc"
    days = (time - np.datetime64("2001-01-01")).astype(np.float64)
    netcdf3.write_classic(str(p), {"time": T, "lat": ny, "lon": nx},
                          {"time": (("time",), days, {"units": "days since 2001-01-01 00:00:00", "calendar": "proleptic_gregorian"}),
                           "lat": (("lat",), np.arange(ny, dtype=np.float32), {}), "lon": (("lon",), np.arange(nx, dtype=np.float32), {}),
                           "sst": (("time", "lat", "lon"), packed, at)}, record_dim="time")
    temp = ingest.open_series(str(p), "sst")
    host = decode_packed(temp.values)
    ref = xmhw_amd.threshold(GridSeries(host, temp.dims, temp.coords, time_encoding=temp.time_encoding), coldSpells=cold)
    calls = []
    real = device.clim_raw_packed
    monkeypatch.setattr(device, "clim_raw_packed", lambda *a, **k: (calls.append(k.get("decoded")), real(*a, **k))[1])
    got = xmhw_amd.threshold(temp, coldSpells=cold)
    assert calls == ["float32" if kind == "i16_f32attrs" else "float64"], "the packed file did not take the in-place path"
    npt.assert_array_equal(got["thresh"], ref["thresh"])
    if kind == "i16_f32attrs":
        npt.assert_array_equal(got["seas"], ref["seas"])
    else:
        npt.assert_allclose(got["seas"], ref["seas"], rtol=1e-12, atol=1e-12)
    assert np.isnan(got["thresh"][:, 2, 3]).all() and np.isfinite(got["thresh"][:, 0, 0]).all()
    # ... and with the in-place path switched off the decoded path gives the same
    monkeypatch.setenv("XMHW_PACKED_DIRECT", "0")
    calls.clear()
    off = xmhw_amd.threshold(temp, coldSpells=cold)
    assert calls == []
    npt.assert_array_equal(off["thresh"], ref["thresh"])
    npt.assert_array_equal(off["seas"], ref["seas"])


def test_encode_land_mask_and_gather_on_codes(dev):
    """xmhw_encode_i16 against numpy's rint / clip, xmhw_land_mask_i16 (all codes the fill code: land; anynans: any) for both
    byte orders, xmhw_gather_cells_i16"""
    h = dev.hip()
    rng = np.random.default_rng(3)
    T, C = 500, 70
    x = (15 + 8 * rng.normal(size=(T, C))).astype(np.float32)
    x[:, 3] = np.nan
    x[10:20, 7] = np.nan
    x[0, 0], x[1, 0] = 1e6, -1e6                       # clamped to the ends of the int16 range (without the lowest code)
    d_x = dev.DeviceBuffer.from_array(x)
    d_c = dev.DeviceBuffer(2 * T * C)
    h.encode_i16(d_x.ptr, T, C, C, d_c.ptr, C, 0.01, 10.0, -32768, 0)
    h.stream_sync(0)
    codes = d_c.to_array((T, C), np.int16)
    with np.errstate(invalid="ignore"):
        want = np.clip(np.rint((x.astype(np.float64) - 10.0) / 0.01), -32767, 32767)
    want = np.where(np.isnan(x), -32768, want).astype(np.int16)
    npt.assert_array_equal(codes, want)
    for big in (False, True):
        stored = np.ascontiguousarray(codes.astype(">i2") if big else codes).view(np.int16)
        d_s = dev.DeviceBuffer.from_array(stored)
        d_m = dev.DeviceBuffer(C)
        for anynans in (0, 1):
            h.land_mask_i16(d_s.ptr, T, C, C, int(big), 1, -32768, anynans, d_m.ptr)
            h.stream_sync(0)
            keep = d_m.to_array((C,), np.uint8) != 0
            miss = codes == -32768
            npt.assert_array_equal(keep, ~miss.any(axis=0) if anynans else ~miss.all(axis=0))
        h.land_mask_i16(d_s.ptr, T, C, C, int(big), 0, 0, 0, d_m.ptr)          # no fill code: every cell stays
        h.stream_sync(0)
        assert d_m.to_array((C,), np.uint8).all()
        idx = np.array([1, 5, 6, 40, 69], dtype=np.int64)
        d_i = dev.DeviceBuffer.from_array(idx)
        d_g = dev.DeviceBuffer(2 * T * idx.size)
        h.gather_cells(d_s.ptr, 2, T, C, d_i.ptr, idx.size, d_g.ptr, idx.size)
        h.stream_sync(0)
        npt.assert_array_equal(d_g.to_array((T, idx.size), np.int16), stored[:, idx])
        for b in (d_s, d_m, d_i, d_g):
            b.free()
    d_x.free(); d_c.free()
