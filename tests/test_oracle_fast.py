"""The vectorised oracle must reproduce the dumb oracle (which is pinned to the
reference fixtures) including NaN samples, empty pools and the tstep path."""
import numpy as np
import numpy.testing as npt
import pytest

import xmhw_oracle as ora
import oracle_fast as fast


def _series(T, C, seed, nanfrac=0.0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    t = np.arange(T)[:, None]
    x = 15 + rng.uniform(2, 10, C) * np.sin(2 * np.pi * (t - rng.uniform(0, 365, C)) / 365.25) \
        + rng.normal(size=(T, C))
    x = x.astype(dtype)
    if nanfrac:
        x[rng.random((T, C)) < nanfrac] = np.nan
    return x


@pytest.mark.parametrize("smooth", [False, True])
@pytest.mark.parametrize("nanfrac", [0.0, 0.07])
def test_fast_matches_dumb_daily(smooth, nanfrac):
    time = np.arange("2001-01-01", "2005-01-01", dtype="datetime64[D]")   # 2004 is leap
    doy = ora.add_doy(time)
    x = _series(time.shape[0], 6, 3, nanfrac)
    d0, t0, s0 = ora.threshold_cells(x, doy, smoothPercentile=smooth, windowHalfWidth=5)
    d1, t1, s1 = fast.threshold_cells_fast(x, doy, smoothPercentile=smooth, windowHalfWidth=5)
    npt.assert_array_equal(d0, d1)
    npt.assert_allclose(t1, t0, rtol=1e-13, atol=0, equal_nan=True)
    npt.assert_allclose(s1, s0, rtol=1e-13, atol=0, equal_nan=True)


def test_fast_matches_dumb_absent_groups():
    """A cell that is NaN for a whole season has no group for those doys: the
    per-cell series is shorter and the smoothing rolls across the gap."""
    time = np.arange("2001-01-01", "2004-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    x = _series(time.shape[0], 3, 5).astype(np.float64)
    x[(doy >= 150) & (doy <= 230), 1] = np.nan
    d0, t0, s0 = ora.threshold_cells(x, doy, windowHalfWidth=3, smoothPercentileWidth=11, pctile=75)
    d1, t1, s1 = fast.threshold_cells_fast(x, doy, windowHalfWidth=3, smoothPercentileWidth=11, pctile=75)
    assert np.isnan(t0[:, 1]).sum() > 50 and np.isfinite(t0[:, 0]).all()
    npt.assert_allclose(t1, t0, rtol=1e-13, equal_nan=True)
    npt.assert_allclose(s1, s0, rtol=1e-13, equal_nan=True)


def test_fast_matches_dumb_tstep_and_cold():
    T, n = 5 * 73, 73
    doy = np.tile(np.arange(1, n + 1), 5)
    x = _series(T, 4, 7, 0.02)
    d0, t0, s0 = ora.threshold_cells(x, doy, tstep=True, windowHalfWidth=2,
                                     smoothPercentileWidth=5, coldSpells=True, pctile=10)
    d1, t1, s1 = fast.threshold_cells_fast(x, doy, tstep=True, windowHalfWidth=2,
                                           smoothPercentileWidth=5, coldSpells=True, pctile=10)
    npt.assert_array_equal(d0, np.arange(1, n + 1))
    npt.assert_allclose(t1, t0, rtol=1e-13, equal_nan=True)
    npt.assert_allclose(s1, s0, rtol=1e-13, equal_nan=True)


def test_percell_baseline_matches_dumb():
    import oracle_percell as opc
    time = np.arange("2001-01-01", "2005-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    x = _series(time.shape[0], 5, 31, 0.05)
    x[(doy >= 100) & (doy <= 140), 2] = np.nan
    for kw in (dict(), dict(smoothPercentile=False, skipna=True), dict(coldSpells=True, pctile=10, windowHalfWidth=2)):
        d0, t0, s0 = ora.threshold_cells(x, doy, **kw)
        d1, t1, s1 = opc.threshold_cells_percell(x, doy, **kw)
        npt.assert_array_equal(d0, d1)
        npt.assert_array_equal(t1, t0)
        npt.assert_array_equal(s1, s0)
