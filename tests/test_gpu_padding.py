"""maxPadLength on the GPU: the pad_gaps kernel through the C ABI against the oracle restatement of
xarray's interpolate_na (oracle/pad_oracle.py: numpy.interp + the max_gap block rule), bit for bit, and
threshold() / detect() end to end with the recipe applied on the device."""
import numpy as np
import numpy.testing as npt
import pytest

import oracle_fast as fast
import pad_oracle as po
from detect_standin import oracle_detect_cells
from test_padding import _gappy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.device as d
    yield d
    d.release_device_cache()


def _pad_on_device(dev, y, x, max_gap):
    h = dev.hip()
    d_y = dev.DeviceBuffer.from_array(y)
    d_x = dev.DeviceBuffer.from_array(np.ascontiguousarray(x, dtype=np.float64))
    try:
        h.pad_gaps(d_y.ptr, y.dtype.itemsize, y.shape[0], y.shape[1], y.shape[1], d_x.ptr, float(max_gap))
        h.stream_sync(0)
        return d_y.to_array(y.shape, y.dtype)
    finally:
        d_y.free()
        d_x.free()


def _random_gappy(T, C, dtype, seed, irregular=False):
    rng = np.random.default_rng(seed)
    y = (15 + 5 * np.sin(np.arange(T)[:, None] / 58.0) + rng.normal(size=(T, C))).astype(dtype)
    for c in range(C):
        for _ in range(int(rng.integers(0, 12))):
            a = int(rng.integers(0, T))
            y[a:a + int(rng.integers(1, 12)), c] = np.nan
    y[:, 0] = np.nan                                  # an all-NaN cell
    y[:, 1] = (np.arange(T) % 7).astype(dtype)         # no NaN at all
    y[:-1, 2] = np.nan                                # a single valid sample
    y[:5, 3] = np.nan                                 # leading run
    y[-5:, 4] = np.nan                                # trailing run
    y[10, 5], y[11, 5], y[12, 5] = np.inf, np.nan, 1.0        # numpy.interp's NaN fallbacks (inf - inf)
    y[20, 5], y[21, 5], y[22, 5] = np.inf, np.nan, np.inf
    y[30, 5], y[31, 5], y[32, 5] = 4.0, np.nan, 4.0
    if irregular:
        steps = rng.integers(1, 4, size=T).astype(np.int64)
        t = np.datetime64("1999-12-31T12", "h") + np.cumsum(steps * 24).astype("timedelta64[h]")
    else:
        t = np.datetime64("2000-01-01", "D") + np.arange(T).astype("timedelta64[D]")
    return y, t


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("T,C,irregular", [(731, 300, False), (1003, 257, True), (8, 64, False), (1, 5, False)])
def test_pad_gaps_kernel_bit_exact(dev, dtype, T, C, irregular):
    if T < 40:
        rng = np.random.default_rng(T)
        y = rng.normal(size=(T, C)).astype(dtype)
        y[rng.random((T, C)) < 0.4] = np.nan
        t = np.datetime64("2000-01-01", "D") + np.arange(T).astype("timedelta64[D]")
    else:
        y, t = _random_gappy(T, C, dtype, seed=T + C, irregular=irregular)
    x = po.interp_index(t)
    for days in (1, 2, 3.5, 6, 1000):
        g = days * 86400e9
        with np.errstate(invalid="ignore"):
            ref = po.interpolate_na(y, x, g)
        got = _pad_on_device(dev, y, x, g)
        npt.assert_array_equal(got, ref, err_msg=f"max_gap {days} days")
        assert got.dtype == y.dtype
    if T > 40:
        assert np.isnan(y).sum() > np.isnan(ref).sum()


def test_numeric_axis_and_exact_boundary(dev):
    """a tstep-style integer axis; a gap whose coordinate distance equals max_gap is filled (<=)"""
    y = np.array([[1, np.nan, np.nan, 4, np.nan, np.nan, np.nan, 8]], dtype=np.float32).T.copy()
    y = np.repeat(y, 70, axis=1)
    x = np.arange(8.0)
    npt.assert_array_equal(_pad_on_device(dev, y, x, 3.0)[:, 0], [1, 2, 3, 4, np.nan, np.nan, np.nan, 8])
    npt.assert_array_equal(_pad_on_device(dev, y, x, 4.0)[:, 69], [1, 2, 3, 4, 5, 6, 7, 8])
    npt.assert_array_equal(_pad_on_device(dev, y, x, 2.999)[:, 5], y[:, 5])


@pytest.mark.parametrize("cold", [False, True])
def test_threshold_and_detect_with_maxPadLength(oisst, cold):
    from xmhw_amd import climatology_series, detect, threshold, threshold_detect
    from xmhw_amd.calendar import add_doy
    g = _gappy(oisst)
    gap = np.timedelta64(4, "D")
    time = oisst["time64"]
    flat = g.values.reshape(g.values.shape[0], -1)
    keep = ~np.isnan(flat).all(axis=0)
    filled = po.interpolate_na(flat[:, keep], po.interp_index(time), 4 * 86400e9)
    doy = add_doy(time)
    doys, th0, se0 = fast.threshold_cells_fast(filled, doy, coldSpells=cold)
    clim = threshold(g, maxPadLength=gap, coldSpells=cold)
    got_th = clim["thresh"].reshape(len(doys), -1)
    got_se = clim["seas"].reshape(len(doys), -1)
    alive = ~np.isnan(got_th).all(axis=0)
    npt.assert_allclose(got_th[:, alive], th0, rtol=1e-12)
    npt.assert_allclose(got_se[:, alive], se0, rtol=1e-12, atol=1e-12)
    # without the interpolation the result differs (the gaps matter)
    plain = threshold(g, coldSpells=cold)
    assert not np.array_equal(plain["thresh"], clim["thresh"], equal_nan=True)
    th, se = climatology_series(clim, "thresh"), climatology_series(clim, "seas")
    mhw, inter = detect(g, th, se, maxPadLength=gap, coldSpells=cold, intermediate=True)
    # reference: the same host logic on a series interpolated by the oracle, oracle detect as device stage
    from xmhw_amd import GridSeries
    from xmhw_amd.detect import _detect
    g_filled = GridSeries(po.interpolate_na(flat, po.interp_index(time), 4 * 86400e9).reshape(g.values.shape),
                          g.dims, g.coords, time_encoding=g.time_encoding)
    ref = _detect(g_filled, th, se, oracle_detect_cells, coldSpells=cold)
    npt.assert_array_equal(mhw.offsets, ref.offsets)
    assert mhw.n_events > 0
    ts_col = np.asarray(inter["ts"]).reshape(filled.shape[0], -1)
    # the `ts` column of mhw_df() is the interpolated (and, for cold spells, negated) series; land cells are NaN
    npt.assert_array_equal(ts_col[:, ~np.isnan(ts_col).all(axis=0)], -filled if cold else filled)
    # the table-only path and the fused call see the same interpolated series
    mhw2 = detect(g, th, se, maxPadLength=gap, coldSpells=cold)
    npt.assert_array_equal(mhw2.table, mhw.table)
    clim3, mhw3 = threshold_detect(g, maxPadLength=gap, coldSpells=cold)
    npt.assert_array_equal(clim3["thresh"], clim["thresh"])
    npt.assert_array_equal(mhw3.table, mhw.table)
    assert ref.table.shape == mhw.table.shape
    npt.assert_allclose(mhw.table, ref.table, rtol=1e-9, atol=1e-11, equal_nan=True)
