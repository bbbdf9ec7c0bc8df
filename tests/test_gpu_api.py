"""threshold() end to end on the GPU (host logic + HIP path) against the oracle,
and the remaining C-ABI entry points: one-shot host call, land mask, synthetic
generator, error codes."""
import numpy as np
import numpy.testing as npt
import pytest

import xmhw_oracle as ora
import oracle_fast as fast

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def h():
    from xmhw_amd._lib import require_gpu, hip
    require_gpu()
    return hip()


def _grid(oisst, sst=None):
    from xmhw_amd import GridSeries
    return GridSeries(oisst["sst"] if sst is None else sst, ("time", "lat", "lon"),
                      {"time": oisst["time64"], "lat": oisst["lat"], "lon": oisst["lon"]},
                      time_encoding={"calendar": "proleptic_gregorian"})


@pytest.mark.parametrize("kw", [dict(), dict(smoothPercentile=False, skipna=True),
                                dict(coldSpells=True, pctile=10), dict(windowHalfWidth=2, smoothPercentileWidth=7),
                                dict(climatologyPeriod=[2004, 2004], windowHalfWidth=1, smoothPercentileWidth=5)])
def test_threshold_grid_vs_oracle(h, oisst, kw):
    from xmhw_amd import threshold
    ds = threshold(_grid(oisst), **kw)
    ref = ora.threshold_grid(oisst["sst"], oisst["time64"], **kw)
    keep = ref["keep"].reshape(8, 4)
    rows, cols = keep.any(axis=1), keep.any(axis=0)
    npt.assert_array_equal(ds.coords["doy"], ref["doy"])
    npt.assert_allclose(ds["thresh"], ref["thresh"][:, rows][:, :, cols], rtol=1e-12, equal_nan=True)
    npt.assert_allclose(ds["seas"], ref["seas"][:, rows][:, :, cols], rtol=1e-12, equal_nan=True)


def test_threshold_point_and_nan_holes(h, oisst):
    from xmhw_amd import threshold, GridSeries
    x = oisst["sst"][:, 1, 2].copy()
    x[100:130] = np.nan
    ds = threshold(GridSeries(x, ("time",), {"time": oisst["time64"]}), skipna=True)
    ref = ora.threshold_grid(x, oisst["time64"], dims=("time",), skipna=True)
    npt.assert_allclose(ds["thresh"], ref["thresh"], rtol=1e-12, equal_nan=True)
    npt.assert_allclose(ds["seas"], ref["seas"], rtol=1e-12, equal_nan=True)


def test_one_shot_host_entry_and_errors(h, oisst):
    ts, keep, _, _ = ora.land_check(oisst["sst"], ("time", "lat", "lon"))
    doy = ora.add_doy(oisst["time64"]).astype(np.int32)
    th, se = h.clim_host(np.ascontiguousarray(ts), doy, 366, 5, 0.9, 1, 31, 1, 0)
    _, t0, s0 = fast.threshold_cells_fast(ts, doy)
    npt.assert_allclose(th, t0, rtol=1e-12)
    npt.assert_allclose(se, s0, rtol=1e-12)
    th64, _ = h.clim_host(np.ascontiguousarray(ts.astype(np.float64)), doy, 366, 5, 0.9, 1, 31, 1, 0)
    npt.assert_allclose(th64, t0, rtol=1e-12)
    with pytest.raises(h.InvalidArgument):          # even smoothing width (xmhw.py:103-104)
        h.clim_host(np.ascontiguousarray(ts), doy, 366, 5, 0.9, 1, 30, 1, 0)
    with pytest.raises(h.InvalidArgument):          # wrong D
        h.clim_host(np.ascontiguousarray(ts), doy, 365, 5, 0.9, 0, 31, 1, 0)
    with pytest.raises(h.InvalidArgument):          # quantile outside [0, 1]
        h.clim_host(np.ascontiguousarray(ts), doy, 366, 5, 1.5, 0, 31, 1, 0)


def test_land_mask_kernel(h, oisst):
    from xmhw_amd.device import DeviceBuffer
    sst = oisst["sst"].reshape(731, 32).copy()
    sst[245, 6] = np.nan
    d = DeviceBuffer.from_array(sst)
    for anynans in (0, 1):
        k = DeviceBuffer(32)
        h.land_mask(d.ptr, 4, 731, 32, 32, anynans, k.ptr)
        h.stream_sync(0)
        keep = k.to_array((32,), np.uint8).astype(bool)
        nan = np.isnan(sst)
        want = ~(nan.any(axis=0) if anynans else nan.all(axis=0))
        npt.assert_array_equal(keep, want)


def test_synth_generator_is_deterministic_and_plausible(h):
    from xmhw_amd.device import DeviceBuffer
    T, C = 2000, 300
    a, b = DeviceBuffer(4 * T * C), DeviceBuffer(4 * T * 100)
    h.synth_sst(a.ptr, 4, T, C, C, 0, 42, 0.05)
    h.synth_sst(b.ptr, 4, T, 100, 100, 200, 42, 0.05)      # cells 200..299 generated alone
    h.stream_sync(0)
    xa = a.to_array((T, C), np.float32)
    xb = b.to_array((T, 100), np.float32)
    npt.assert_array_equal(xa[:, 200:], xb)                # counter-based: any subset reproduces
    frac = np.isnan(xa).mean()
    assert 0.04 < frac < 0.06
    assert 5 < np.nanmean(xa) < 25 and 1.5 < np.nanstd(xa) < 9


def test_many_cells_not_multiple_of_tile(h):
    """ragged cell counts (not a multiple of the 8-cell wave / 32-cell block tile)."""
    from xmhw_amd.device import calc_clim_device
    time = np.arange("2001-01-01", "2007-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    rng = np.random.default_rng(4)
    for C in (1, 7, 9, 33, 65):
        x = (15 + rng.normal(size=(time.shape[0], C))).astype(np.float32)
        _, t1, s1 = calc_clim_device(x, doy, 90, 5, True, 31, False)
        _, t0, s0 = fast.threshold_cells_fast(x, doy)
        npt.assert_allclose(t1, t0, rtol=1e-12)
        npt.assert_allclose(s1, s0, rtol=1e-12)


def test_resident_pipeline_mask_compact_clim_scatter(h, oisst):
    """land_check + calc_clim + unstack entirely on the device: mask -> gather ->
    raw + finish -> scatter, against the oracle's threshold_grid()."""
    from xmhw_amd.device import DeviceBuffer, Plan, clim_raw, clim_finish
    sst = oisst["sst"].reshape(731, 32)
    T, C = sst.shape
    d_in = DeviceBuffer.from_array(sst)
    d_keep = DeviceBuffer(C)
    h.land_mask(d_in.ptr, 4, T, C, C, 0, d_keep.ptr)
    h.stream_sync(0)
    keep = d_keep.to_array((C,), np.uint8).astype(bool)
    idx = np.nonzero(keep)[0].astype(np.int64)
    n = idx.size
    d_idx = DeviceBuffer.from_array(idx)
    d_ts = DeviceBuffer(4 * T * n)
    h.gather_cells(d_in.ptr, 4, T, C, d_idx.ptr, n, d_ts.ptr, n)
    doy = ora.add_doy(oisst["time64"])
    plan = Plan(doy, 5)
    D = plan.D
    raw_t, raw_s, fin_t, fin_s = (DeviceBuffer(8 * D * n) for _ in range(4))
    clim_raw(plan, d_ts, 4, n, 0.9, False, raw_t, raw_s)
    clim_finish(plan, raw_t, raw_s, n, True, True, 31, fin_t, fin_s)
    out_t, out_s = DeviceBuffer(8 * D * C), DeviceBuffer(8 * D * C)
    h.scatter_cells(fin_t.ptr, D, n, d_idx.ptr, n, out_t.ptr, C)
    h.scatter_cells(fin_s.ptr, D, n, d_idx.ptr, n, out_s.ptr, C)
    h.stream_sync(0)
    ref = ora.threshold_grid(oisst["sst"], oisst["time64"])
    npt.assert_array_equal(keep, ref["keep"])
    npt.assert_allclose(out_t.to_array((D, C), np.float64).reshape(D, 8, 4), ref["thresh"], rtol=1e-12, equal_nan=True)
    npt.assert_allclose(out_s.to_array((D, C), np.float64).reshape(D, 8, 4), ref["seas"], rtol=1e-12, equal_nan=True)


def test_cell_batches_give_identical_results(h):
    """calc_clim_device in several cell batches == one batch (cells are independent)."""
    from xmhw_amd.device import calc_clim_device
    time = np.arange("2001-01-01", "2006-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    rng = np.random.default_rng(8)
    x = (15 + rng.normal(size=(time.shape[0], 45))).astype(np.float32)
    x[rng.random(x.shape) < 0.02] = np.nan
    one = calc_clim_device(x, doy, 90, 5, True, 31, False)
    many = calc_clim_device(x, doy, 90, 5, True, 31, False, max_batch_bytes=7 * x.shape[0] * 4)
    for a, b in zip(one, many):
        npt.assert_array_equal(a, b)
    empty = calc_clim_device(x[:, :0], doy, 90, 5, True, 31, False)
    assert empty[1].shape == (366, 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_strided_slabs_ld_and_ldo(h, dtype):
    """A slab of a wider resident array: ld > C on input, ldo > C on output (what
    bench.py does per slab and per rank)."""
    from xmhw_amd.device import DeviceBuffer, Plan, clim_raw, clim_finish
    time = np.arange("2001-01-01", "2005-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    T, Cfull, a, n = time.shape[0], 50, 13, 21
    rng = np.random.default_rng(5)
    x = (15 + rng.normal(size=(T, Cfull))).astype(dtype)
    plan = Plan(doy, 5)
    D, isz = plan.D, x.dtype.itemsize
    d_x = DeviceBuffer.from_array(x)
    ldo = 32
    raw_t, raw_s, out_t, out_s = (DeviceBuffer(8 * D * ldo) for _ in range(4))
    for b in (raw_t, raw_s, out_t, out_s):
        h.memset(b.ptr, 0xFF, 8 * D * ldo)
    clim_raw(plan, d_x.ptr + isz * a, isz, n, 0.9, False, raw_t, raw_s, ld=Cfull, ldo=ldo)
    clim_finish(plan, raw_t, raw_s, n, True, True, 31, out_t, out_s, ldo=ldo)
    h.stream_sync(0)
    got_t = out_t.to_array((D, ldo), np.float64)
    got_s = out_s.to_array((D, ldo), np.float64)
    _, t0, s0 = fast.threshold_cells_fast(x[:, a:a + n], doy)
    npt.assert_allclose(got_t[:, :n], t0, rtol=1e-12)
    npt.assert_allclose(got_s[:, :n], s0, rtol=1e-12)
    # the padding columns of the output rows were not touched
    assert np.isnan(got_t[:, n:]).all() and (got_t[:, n:].view(np.uint64) == 0xFFFFFFFFFFFFFFFF).all()


def test_device_side_land_mask_and_compaction_equal_host_path(oisst):
    """threshold()/detect() hand the uncompacted grid to the device (pitched slab uploads, land_mask,
    gather); same results as numpy's land_check() followed by the compact-array entry points, also
    with several slabs, with anynans, and with a float64 grid."""
    from xmhw_amd import landmask
    from xmhw_amd.device import calc_clim_device, calc_clim_grid_device
    from xmhw_amd.detect_front import detect_cells, detect_grid
    import xmhw_oracle as ora
    rng = np.random.default_rng(3)
    sst = oisst["sst"].copy()
    sst[rng.random(sst.shape) < 0.01] = np.nan               # scattered NaNs: anynans drops more cells
    doy = ora.add_doy(oisst["time64"])
    for dtype in (np.float32, np.float64):
        for anynans in (False, True):
            x = sst.astype(dtype)
            x[:, 1, 1] = oisst["sst"][:, 1, 1] if not np.isnan(oisst["sst"][:, 1, 1]).all() else 1.0
            ts, keep, sdims, sshape = landmask.land_check(x, ("time", "lat", "lon"), "time", anynans)
            stacked, _, _ = landmask.stack_cells(x, ("time", "lat", "lon"), "time")
            d0, th0, se0 = calc_clim_device(ts, doy, 90, 5, True, 31, False)
            for budget in (64 << 30, 60_000):                # one slab / a handful of columns per slab
                k1, d1, th1, se1 = calc_clim_grid_device(stacked, doy, anynans, 90, 5, True, 31, False,
                                                         max_batch_bytes=budget, scatter=False)
                npt.assert_array_equal(k1, keep)
                npt.assert_array_equal(th1, th0)
                npt.assert_array_equal(se1, se0)
                # default: results placed back on the grid by the device, NaN at the dropped cells
                k2, d2, th2, se2 = calc_clim_grid_device(stacked, doy, anynans, 90, 5, True, 31, False,
                                                         max_batch_bytes=budget)
                assert th2.shape == (366, stacked.shape[1]) and np.isnan(th2[:, ~keep]).all()
                npt.assert_array_equal(th2[:, keep], th0)
                npt.assert_array_equal(se2[:, keep], se0)
                r0 = detect_cells(ts, se0, th0, doy, d0, 5, True, 2)
                r1 = detect_grid(stacked, anynans, se0, th0, doy, d0, 5, True, 2, max_batch_bytes=budget)
                npt.assert_array_equal(r1["keep"], keep)
                npt.assert_array_equal(r1["offsets"], r0["offsets"])
                npt.assert_array_equal(r1["table"], r0["table"])
                # climatologies handed over on the grid too (NaN at land): masked / compacted on the device
                r2 = detect_grid(stacked, anynans, se2, th2, doy, d0, 5, True, 2, max_batch_bytes=budget,
                                 clim_stacked=True)
                npt.assert_array_equal(r2["offsets"], r0["offsets"])
                npt.assert_array_equal(r2["table"], r0["table"])
    # a rank's block of columns (sharded runs): same cells as the corresponding part of the whole
    kf, df, thf, sef = calc_clim_grid_device(stacked, doy, False, 90, 5, True, 31, False)
    for c0, c1 in ((0, 11), (11, 32), (5, 6), (30, 32)):
        kp, dp, thp, sep = calc_clim_grid_device(stacked, doy, False, 90, 5, True, 31, False, columns=(c0, c1))
        npt.assert_array_equal(kp, kf[c0:c1])
        npt.assert_array_equal(thp, thf[:, c0:c1])
        npt.assert_array_equal(sep, sef[:, c0:c1])
    # detect on blocks of columns with the survivor offsets a sharded run exchanges
    full = detect_grid(stacked, False, sef, thf, doy, df, 5, True, 2, clim_stacked=True)
    total = int(kf.sum())
    tabs, cnts = [], []
    for c0, c1 in ((0, 11), (11, 12), (12, 32)):
        off = int(kf[:c0].sum())
        r = detect_grid(stacked, False, sef, thf, doy, df, 5, True, 2, clim_stacked=True, columns=(c0, c1),
                        exchange=lambda n, off=off, c0=c0, c1=c1: (off, total) if n == int(kf[c0:c1].sum()) else (-1, -1))
        npt.assert_array_equal(r["keep"], kf[c0:c1])
        tabs.append(r["table"]); cnts.append(np.diff(r["offsets"]))
    npt.assert_array_equal(np.concatenate(tabs, axis=0), full["table"])
    npt.assert_array_equal(np.concatenate(cnts), np.diff(full["offsets"]))
    # all land -> the reference's exception, raised after the device mask
    from xmhw_amd import XmhwException
    with pytest.raises(XmhwException):
        calc_clim_grid_device(np.full((731, 6), np.nan, np.float32), doy, False, 90, 5, True, 31, False)
    with pytest.raises(XmhwException):                       # climatologies on fewer cells than the series
        detect_grid(stacked, False, se0[:, :-1], th0[:, :-1], doy, d0)


def test_large_buffers_are_recycled_and_released():
    from xmhw_amd import release_device_cache
    from xmhw_amd import device as dv
    release_device_cache()
    a = dv.DeviceBuffer(3 << 29)                 # 1.5 GB
    p = a.ptr
    a.free()
    assert len(dv._POOL) == 1
    b = dv.DeviceBuffer(1 << 30)                 # fits the parked buffer (capacity <= 2x the request)
    assert b.ptr == p and b.capacity == 3 << 29
    c = dv.DeviceBuffer(1 << 20)                 # small buffers never touch the pool
    c.free(); b.free()
    assert len(dv._POOL) == 1
    release_device_cache()
    assert dv._POOL == []


def test_big_endian_and_integer_input(oisst):
    """netCDF-3 style big-endian float32 keeps its width (and the float32 kernels); integers go to float64."""
    from xmhw_amd import threshold
    from xmhw_amd.device import native_float
    ref = threshold(_grid(oisst))
    be = threshold(_grid(oisst, oisst["sst"].astype(">f4")))
    npt.assert_array_equal(be["thresh"], ref["thresh"])
    npt.assert_array_equal(be["seas"], ref["seas"])
    assert native_float(oisst["sst"].astype(">f4")).dtype == np.float32
    ints = np.where(np.isnan(oisst["sst"]), -99, np.round(oisst["sst"] * 10)).astype(np.int16)
    a = threshold(_grid(oisst, ints), smoothPercentile=False)
    b = threshold(_grid(oisst, ints.astype(np.float64)), smoothPercentile=False)
    npt.assert_array_equal(a["thresh"], b["thresh"])
