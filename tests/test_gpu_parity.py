"""Parity of the HIP path (through the C ABI) against the CPU oracle.

Tolerances: thresh/seas 1e-12 relative here (contract: 1e-6, BASELINE.json);
the raw pooled quantile is expected bit-exact for float32 input because the
selection is exact and the interpolation repeats numpy's _lerp in float64.
"""
import numpy as np
import numpy.testing as npt
import pytest

import xmhw_oracle as ora
import oracle_fast as fast

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.device as d
    return d


def _series(T, C, seed, nanfrac=0.0, dtype=np.float32, quant=None):
    rng = np.random.default_rng(seed)
    t = np.arange(T)[:, None]
    x = 15 + rng.uniform(2, 10, C) * np.sin(2 * np.pi * (t - rng.uniform(0, 365, C)) / 365.25) \
        + 0.0005 * t * rng.uniform(-1, 1, C) + rng.normal(size=(T, C))
    if quant:
        x = np.round(x / quant) * quant       # many exact ties, like 0.01 K OISST
    x = x.astype(dtype)
    if nanfrac:
        x[rng.random((T, C)) < nanfrac] = np.nan
    return x


def _daily(y0, y1):
    time = np.arange(f"{y0}-01-01", f"{y1 + 1}-01-01", dtype="datetime64[D]")
    return time, ora.add_doy(time)


def _check(dev, x, doy, kernel, nchunks=0, rtol=1e-12, **kw):
    d1, t1, s1 = dev.calc_clim_device(x, doy, kw.get("pctile", 90), kw.get("windowHalfWidth", 5),
                                      kw.get("smoothPercentile", True), kw.get("smoothPercentileWidth", 31),
                                      kw.get("tstep", False), kw.get("coldSpells", False),
                                      kernel=kernel, nchunks=nchunks)
    d0, t0, s0 = fast.threshold_cells_fast(x, doy, **kw)
    npt.assert_array_equal(d1, d0)
    npt.assert_array_equal(np.isnan(t1), np.isnan(t0))
    npt.assert_allclose(t1, t0, rtol=rtol, atol=0, equal_nan=True)
    npt.assert_allclose(s1, s0, rtol=rtol, atol=0, equal_nan=True)
    return t1, t0


@pytest.mark.parametrize("kernel", ["generic", "ring"])
def test_oisst_fixture_vs_reference_clim(dev, oisst, clim_golden, kernel):
    """The reference's own test_threshold (test/test_xmhw.py:24-66) on the GPU."""
    sst = oisst["sst"]
    ts, keep, _, _ = ora.land_check(sst, ("time", "lat", "lon"))
    doy = ora.add_doy(oisst["time64"])
    cells = np.nonzero(keep)[0]
    for smooth, tag, start in ((False, "nosmooth", 60), (True, "smooth", 82)):
        d, th, se = dev.calc_clim_device(ts, doy, 90, 5, smooth, 31, False, kernel=kernel)
        for k, latlon in ((1, clim_golden["point1_latlon"]), (2, clim_golden["point2_latlon"])):
            i = int(np.argmin(np.abs(oisst["lat"] - latlon[0])))
            j = int(np.argmin(np.abs(oisst["lon"] - latlon[1])))
            c = int(np.nonzero(cells == i * 4 + j)[0][0])
            npt.assert_array_almost_equal(clim_golden[f"{tag}_thresh{k}"][start:], th[start:, c])
            npt.assert_array_almost_equal(clim_golden[f"{tag}_seas{k}"][start:], se[start:, c], decimal=4)
            assert np.max(np.abs(clim_golden[f"{tag}_thresh{k}"][start:] - th[start:, c])) < 1e-12


@pytest.mark.parametrize("kernel", ["generic", "ring"])
@pytest.mark.parametrize("nanfrac", [0.0, 0.05])
def test_daily_random(dev, kernel, nanfrac):
    time, doy = _daily(2001, 2012)
    x = _series(time.shape[0], 203, 11, nanfrac)
    t1, t0 = _check(dev, x, doy, kernel, smoothPercentile=False)
    if nanfrac == 0.0:
        # exact selection + numpy's lerp: every doy but 60 (3-point mean) bit-exact
        m = np.ones(366, bool); m[59] = False
        npt.assert_array_equal(t1[m], t0[m])
    _check(dev, x, doy, kernel)


@pytest.mark.parametrize("kernel", ["generic", "ring"])
def test_ties_quantised_input(dev, kernel):
    time, doy = _daily(1995, 2004)
    x = _series(time.shape[0], 64, 5, 0.02, quant=0.25)
    _check(dev, x, doy, kernel, smoothPercentile=False)
    _check(dev, x, doy, kernel, pctile=50, windowHalfWidth=2, smoothPercentileWidth=5)


@pytest.mark.parametrize("kernel", ["generic", "ring"])
def test_absent_groups_and_all_nan_cells(dev, kernel):
    time, doy = _daily(2001, 2008)
    x = _series(time.shape[0], 40, 7)
    x[(doy >= 150) & (doy <= 230), 3] = np.nan      # seasonal gap: no group for those doys
    x[:, 9] = np.nan                                # land
    x[: 365 * 4, 11] = np.nan                       # half the record missing
    _check(dev, x, doy, kernel)
    _check(dev, x, doy, kernel, smoothPercentile=False)


@pytest.mark.parametrize("kernel", ["generic", "ring"])
def test_tstep_cold_pctile10(dev, kernel):
    n, ny = 73, 9
    doy = np.tile(np.arange(1, n + 1), ny)
    x = _series(n * ny, 77, 3, 0.03)
    _check(dev, x, doy, kernel, tstep=True, windowHalfWidth=2, smoothPercentileWidth=5,
           coldSpells=True, pctile=10)


@pytest.mark.parametrize("nchunks", [1, 3, 7])
def test_ring_chunks_identical(dev, nchunks):
    time, doy = _daily(2003, 2014)           # starts in a non-leap year, has 3 leap years
    x = _series(time.shape[0], 50, 21, 0.04)
    d, t1, s1 = dev.calc_clim_device(x, doy, 90, 5, False, 31, False, kernel="ring", nchunks=nchunks)
    d, t0, s0 = dev.calc_clim_device(x, doy, 90, 5, False, 31, False, kernel="generic")
    npt.assert_array_equal(t1, t0)
    npt.assert_allclose(s1, s0, rtol=1e-13, equal_nan=True)


def test_ring_partial_years_and_period(dev):
    """Series starting mid-year and ending mid-year: first/last tracks are partial."""
    time = np.arange("2001-07-15", "2009-03-10", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    x = _series(time.shape[0], 33, 9, 0.01)
    _check(dev, x, doy, "ring", smoothPercentile=False)
    _check(dev, x, doy, "generic", smoothPercentile=False)


def test_float64_input_generic(dev):
    time, doy = _daily(2001, 2006)
    x = _series(time.shape[0], 31, 13, 0.02, dtype=np.float64)
    _check(dev, x, doy, "generic")


def test_extreme_values(dev):
    time, doy = _daily(2001, 2004)
    x = _series(time.shape[0], 16, 17)
    x[5, 0] = np.inf; x[100, 0] = -np.inf; x[7, 1] = 0.0; x[8, 1] = -0.0
    x[:, 2] = 1.5                                    # constant cell: every key tied
    x[:, 3] = np.float32(1e-42)                      # subnormal
    for kernel in ("ring", "generic"):
        d1, t1, s1 = dev.calc_clim_device(x, doy, 90, 5, False, 31, True, kernel=kernel)
        d0, t0, s0 = fast.threshold_cells_fast(x, doy, smoothPercentile=False, tstep=True)
        fin = np.isfinite(t0)
        npt.assert_allclose(t1[fin], t0[fin], rtol=1e-12)
        npt.assert_array_equal(np.isfinite(t1), fin)
        # an infinite sample makes the mean of the pools that hold it +-inf (NaN with both signs),
        # as numpy's mean does, and leaves every other row untouched
        npt.assert_allclose(s1, s0, rtol=1e-12, equal_nan=True)
        # ... and through the circular running mean (sliding sums fall back to direct sums there)
        d1, t1, s1 = dev.calc_clim_device(x, doy, 90, 5, True, 31, False, kernel=kernel)
        d0, t0, s0 = fast.threshold_cells_fast(x, doy, smoothPercentile=True)
        npt.assert_allclose(t1, t0, rtol=1e-12, equal_nan=True)
        npt.assert_allclose(s1, s0, rtol=1e-12, equal_nan=True)


@pytest.mark.parametrize("years", [(1960, 2020), (1931, 2020), (1925, 2020), (1920, 2020), (1850, 2014)])
def test_long_records_use_the_16_and_32_lane_ring(dev, years):
    """49..96 tracks: the float32 ring kernel with 16 lanes per cell (61, 90 and 96 years), 97..192 tracks
    with 32 lanes per cell (101 and 165 years); bit-identical raw thresholds to the generic kernel, oracle
    parity, also as float64 input holding float32 values."""
    time, doy = _daily(*years)
    nyears = years[1] - years[0] + 1
    x = _series(time.shape[0], 19, 23, 0.01)
    x[:, 3] = np.nan
    plan = dev.Plan(doy, 5)
    assert plan.kernel == "ring" and plan.ntracks == nyears > 48
    plan.destroy()
    _check(dev, x, doy, "ring", smoothPercentile=False)
    t_ring = dev.calc_clim_device(x, doy, 90, 5, False, 31, True, kernel="ring")
    t_gen = dev.calc_clim_device(x, doy, 90, 5, False, 31, True, kernel="generic")
    npt.assert_array_equal(t_ring[1], t_gen[1])
    npt.assert_allclose(t_ring[2], t_gen[2], rtol=1e-13, equal_nan=True)
    t_chunks = dev.calc_clim_device(x, doy, 90, 5, False, 31, True, kernel="ring", nchunks=5)
    npt.assert_array_equal(t_chunks[1], t_ring[1])
    t64 = dev.calc_clim_device(x.astype(np.float64), doy, 90, 5, False, 31, True)
    npt.assert_array_equal(t64[1], t_ring[1])


def test_more_than_192_tracks_fall_back_to_the_generic_kernel(dev):
    time, doy = _daily(1820, 2020)
    plan = dev.Plan(doy, 5)
    assert plan.kernel == "generic" and plan.ntracks == 201
    plan.destroy()


@pytest.mark.parametrize("w,years", [(7, (1982, 2021)), (10, (1982, 2021)), (15, (1991, 2020)), (4, (1982, 2021)),
                                      (7, (1991, 2020)), (10, (2001, 2020))])
def test_other_window_widths_stay_on_the_ring_kernel(dev, w, years):
    """Window half widths 4, 7, 10, 15 on 20-40 year records (8 or 16 lanes per cell): raw thresholds
    bit-identical to the generic kernel, sums to rounding."""
    time, doy = _daily(*years)
    x = _series(time.shape[0], 13, 31 + w, 0.02)
    plan = dev.Plan(doy, w)
    assert plan.kernel == "ring"
    plan.destroy()
    a = dev.calc_clim_device(x, doy, 90, w, False, 31, True, kernel="ring")
    b = dev.calc_clim_device(x, doy, 90, w, False, 31, True, kernel="generic")
    npt.assert_array_equal(a[1], b[1])
    npt.assert_allclose(a[2], b[2], rtol=1e-13, equal_nan=True)
    _check(dev, x[:, :5], doy, "ring", windowHalfWidth=w)
