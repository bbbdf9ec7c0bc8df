"""Every BASELINE.json config as a driver-run GPU parity test, at the config's own shape.

For configs[1..4] the synthetic input of SURVEY 8(d) is generated in HBM at the config's exact
T / D / window / percentile / NaN fraction / tstep axis / smoothing width (single-GPU share for the
8-GPU configs), the hot path runs through the C ABI (plan -> clim_raw -> clim_finish), and

  * >= 4,096 cells spread over the grid are compared with ``oracle_fast`` (the vectorised numpy
    restatement of xmhw/identify.py:184-270, xmhw/xmhw.py:250-307; fanned over the host cores):
    contract 1e-6 relative (BASELINE.json north_star), asserted at 1e-11; doy rows bit-exact;
  * full-size properties: no NaN in any ocean cell's climatology, chunked == unchunked bit for
    bit on the sampled cells.

configs[0] (single point, 30-yr daily) goes through the public point path.

Kernels this file is the evidence for: every float32 config runs on the sorted-list layout (XMHW_LAYOUT_SORTED = 40,
csrc/kernels_sorted.hip) -- asserted per config together with the instantiation it picks: 30 tracks -> 12 keys per list
(configs[1]), 40 tracks -> 16 keys per list, 14 of them in LDS = 20,480 bytes per wave (configs[2], [3]), 20 tracks on
the 6-hourly axis -> 10 keys per list (configs[4]); and on the untiled ``clim_finish`` for D = 1460 > 511 (configs[4]).
A plan silently falling back to a ring layout would fail here, not just run slower.
"""
import numpy as np
import numpy.testing as npt
import pytest

import xmhw_oracle as ora
import oracle_fast as fast
import parallel as opar

pytestmark = pytest.mark.gpu

NPAR = 4096          # cells compared with the oracle per config
SEED0 = 20260101     # + config index


@pytest.fixture(scope="module")
def dev():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.device as d
    yield d
    d.release_device_cache()


def _daily(y0, y1):
    time = np.arange(f"{y0}-01-01", f"{y1 + 1}-01-01", dtype="datetime64[D]")
    return ora.add_doy(time)


def _gather(dev, src, itemsize, rows, ld, idx, dtype):
    """columns idx of a device (rows, ld) array as a host (rows, len(idx)) array"""
    h = dev.hip()
    d_idx = dev.DeviceBuffer.from_array(idx.astype(np.int64))
    d_out = dev.DeviceBuffer(itemsize * rows * idx.size)
    try:
        h.gather_cells(src.ptr, itemsize, rows, ld, d_idx.ptr, idx.size, d_out.ptr, idx.size)
        h.stream_sync(0)
        return d_out.to_array((rows, idx.size), dtype)
    finally:
        d_idx.free()
        d_out.free()


def _no_nan_columns(dev, buf, rows, C):
    """device-side check: number of columns of a (rows, C) float64 array that hold a NaN"""
    h = dev.hip()
    d_keep = dev.DeviceBuffer(C)
    try:
        h.land_mask(buf.ptr, 8, rows, C, C, 1, d_keep.ptr)      # anynans: keep = no NaN at all
        h.stream_sync(0)
        return int(C - np.count_nonzero(d_keep.to_array((C,), np.uint8)))
    finally:
        d_keep.free()


def _run_config(dev, index, C, doy, nan_frac, tstep, width, skipna, expect_yps=None, full_compare=False):
    h = dev.hip()
    T = int(doy.shape[0])
    w, pctile = 5, 90
    plan = dev.Plan(doy, w)
    plan1 = dev.Plan(doy, w, nchunks=1)
    D = plan.D
    assert plan.kernel == "ring"
    # the layout and the instantiation this config is measured on
    assert plan.layout_in_use() == 40 and plan1.layout_in_use() == 40, (plan.layout_in_use(), plan1.layout_in_use())
    keys, lds, _ = h.plan_sorted_info(plan.handle, C)
    assert (plan.ntracks, int(keys), int(lds)) == {30: (30, 12, 17920), 40: (40, 16, 20480), 20: (20, 10, 14080)}[plan.ntracks], \
        (plan.ntracks, keys, lds)
    bufs = []
    try:
        ts = dev.DeviceBuffer(4 * T * C); bufs.append(ts)
        h.synth_sst(ts.ptr, 4, T, C, C, 0, SEED0 + index, nan_frac, 0)
        raw = [dev.DeviceBuffer(8 * D * C) for _ in range(2)]; bufs += raw
        out = [dev.DeviceBuffer(8 * D * C) for _ in range(2)]; bufs += out
        dev.clim_raw(plan, ts, 4, C, pctile / 100.0, False, raw[0], raw[1])
        dev.clim_finish(plan, raw[0], raw[1], C, not tstep, True, width, out[0], out[1])
        h.stream_sync(0)

        # ---- doy rows bit-exact (xmhw/identify.py:28-79) ----------------------------------
        npt.assert_array_equal(plan.doys, np.unique(doy))

        # ---- >= 4,096 spread cells against the oracle ---------------------------------------
        idx = np.unique(np.linspace(0, C - 1, min(NPAR, C)).astype(np.int64))
        sample = _gather(dev, ts, 4, T, C, idx, np.float32)
        got_th = _gather(dev, out[0], 8, D, C, idx, np.float64)
        got_se = _gather(dev, out[1], 8, D, C, idx, np.float64)
        if nan_frac:
            frac = float(np.isnan(sample).mean())
            assert abs(frac - nan_frac) < 0.2 * nan_frac, frac
            assert not np.isnan(sample).all(axis=0).any()          # no all-NaN cell
        with opar.OraclePool(doy, w) as pool:
            th0, se0 = pool.threshold_fast(sample, pctile=pctile, windowHalfWidth=w, smoothPercentileWidth=width,
                                           tstep=tstep, skipna=skipna)
        assert not np.isnan(th0).any() and not np.isnan(se0).any()
        # contract (BASELINE.json): 1e-6 relative
        npt.assert_allclose(got_th, th0, rtol=1e-6, atol=0)
        npt.assert_allclose(got_se, se0, rtol=1e-6, atol=1e-12)
        # what the path actually delivers: float64 round-off (the seasonal mean may cross zero on a
        # tstep axis, hence the absolute term)
        npt.assert_allclose(got_th, th0, rtol=1e-11, atol=0)
        npt.assert_allclose(got_se, se0, rtol=1e-11, atol=1e-12)

        # ---- full-size properties -----------------------------------------------------------
        assert _no_nan_columns(dev, out[0], D, C) == 0
        assert _no_nan_columns(dev, out[1], D, C) == 0
        assert _no_nan_columns(dev, raw[0], D, C) == 0
        # chunked (auto) == unchunked, bit for bit
        if full_compare:
            a_th, a_se = out[0].to_array((D, C), np.float64), out[1].to_array((D, C), np.float64)
        dev.clim_raw(plan1, ts, 4, C, pctile / 100.0, False, raw[0], raw[1])
        dev.clim_finish(plan1, raw[0], raw[1], C, not tstep, True, width, out[0], out[1])
        h.stream_sync(0)
        if full_compare:
            npt.assert_array_equal(out[0].to_array((D, C), np.float64), a_th)
            npt.assert_array_equal(out[1].to_array((D, C), np.float64), a_se)
        npt.assert_array_equal(_gather(dev, out[0], 8, D, C, idx, np.float64), got_th)
        npt.assert_array_equal(_gather(dev, out[1], 8, D, C, idx, np.float64), got_se)
        return idx.size
    finally:
        for b in bufs:
            b.free()
        plan.destroy()
        plan1.destroy()
        dev.release_device_cache()


def test_config0_single_point_30yr(dev):
    """configs[0]: single-point 30-yr daily series through the public point path
    (xmhw/xmhw.py:122-126, :167-179)."""
    import xmhw_amd
    time = np.arange("1991-01-01", "2021-01-01", dtype="datetime64[D]")
    assert time.shape[0] == 10958
    rng = np.random.default_rng(SEED0)
    t = np.arange(time.shape[0])
    x = (15 + 6 * np.sin(2 * np.pi * (t - 40) / 365.25) + 0.0005 * t * 0.3 + rng.normal(size=t.shape)).astype(np.float32)
    ds = xmhw_amd.threshold_array(x, time, dims=("time",))
    doy = ora.add_doy(time)
    _, th0, se0 = fast.threshold_cells_fast(x[:, None], doy)
    assert ds["thresh"].shape == (366,) and ds["seas"].shape == (366,)
    npt.assert_array_equal(ds.coords["doy"], np.arange(1, 367))
    npt.assert_allclose(ds["thresh"], th0[:, 0], rtol=1e-12)
    npt.assert_allclose(ds["seas"], se0[:, 0], rtol=1e-12)
    # and against the dumb per-cell oracle, which follows the reference line by line
    _, th1, se1 = ora.threshold_cells(x[:, None].astype(np.float64), doy)
    npt.assert_allclose(ds["thresh"], th1[:, 0], rtol=1e-12)
    npt.assert_allclose(ds["seas"], se1[:, 0], rtol=1e-12)


def test_config1_1deg_30yr(dev):
    """configs[1]: 1 deg global (360 x 180), 30-yr daily, w=5, p=90, skipna=False."""
    doy = _daily(1991, 2020)
    assert doy.shape[0] == 10958
    n = _run_config(dev, 1, 360 * 180, doy, 0.0, False, 31, False, full_compare=True)
    assert n >= 4096


def test_config2_quarter_degree_40yr(dev):
    """configs[2]: 0.25 deg global (1440 x 720), 40-yr daily, defaults -- the metric's config, full size."""
    doy = _daily(1982, 2021)
    assert doy.shape[0] == 14610
    n = _run_config(dev, 2, 1440 * 720, doy, 0.0, False, 31, False)
    assert n >= 4096


def test_config3_quarter_degree_5pct_nan_share(dev):
    """configs[3]: as configs[2] with skipna=True and ~5 % NaN; one GPU's share of the 8-way shard
    (129,600 cells)."""
    doy = _daily(1982, 2021)
    n = _run_config(dev, 3, 1440 * 720 // 8, doy, 0.05, False, 31, True)
    assert n >= 4096


def test_config4_005deg_6hourly_tstep_share(dev):
    """configs[4]: 0.05 deg tile, 20-yr 6-hourly on a 1460-steps-per-year axis (tstep path,
    docs/frequency.rst:42-50), smoothPercentileWidth=31; one GPU's share of the 8-way shard
    (810,000 cells).  D = 1460 takes the untiled clim_finish; runavg at D = 1460 follows
    xmhw/identify.py:154-181."""
    doy = np.tile(np.arange(1, 1461, dtype=np.int64), 20)
    assert doy.shape[0] == 29200
    n = _run_config(dev, 4, 810000, doy, 0.0, True, 31, False)
    assert n >= 4096
