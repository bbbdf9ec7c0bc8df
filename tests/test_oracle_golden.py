"""Pin the CPU oracle to the reference's own fixtures.

Each test mirrors a reference test: test/test_xmhw.py:24-66 (test_threshold),
test/test_identify.py:38-49 (add_doy), :52-59 (feb29), :62-77 (runavg),
:80-87 (window_roll), :132-155 (land_check), :197-217 (get_calendar table).
"""
import numpy as np
import numpy.testing as npt
import pytest

import xmhw_oracle as ora


def _point(oisst, latlon):
    i = int(np.argmin(np.abs(oisst["lat"] - latlon[0])))
    j = int(np.argmin(np.abs(oisst["lon"] - latlon[1])))
    return i, j


def test_add_doy_bit_exact(oisst, literals):
    doy = ora.add_doy(oisst["time64"])
    assert doy.dtype == np.int64
    npt.assert_array_equal(doy, literals["oisst_doy"])


def test_add_doy_tstep(oisst, literals):
    # 5-day means (coarsen trim) -> 146 steps, 73 per year
    t5 = oisst["time64"][: (731 // 5) * 5].reshape(-1, 5)[:, 2]
    npt.assert_array_equal(ora.add_doy(t5, keep_tstep=True), literals["days5_doy"])
    # monthly means -> 24 steps
    tm = np.arange("2003-01", "2005-01", dtype="datetime64[M]").astype("datetime64[D]") + 14
    npt.assert_array_equal(ora.add_doy(tm, keep_tstep=True), literals["mon_doy"])
    with pytest.raises(ora.XmhwException):
        ora.add_doy(oisst["time64"][:700], keep_tstep=True)


def test_feb29(oisst, literals):
    # test_identify.py:52-59 -- mean over DAYS with doy in {59,60,61} at [1,2]
    doy = ora.add_doy(oisst["time64"])
    x = oisst["sst"][:, 1, 2].astype(np.float64)
    b = ora.feb29(x, doy)
    npt.assert_almost_equal(literals["feb29_expected"][0], b, decimal=5)


def test_runavg(literals):
    a = literals["runavg_in"]
    npt.assert_almost_equal(ora.runavg(a, 3), literals["runavg_w3"], decimal=5)
    npt.assert_almost_equal(ora.runavg(a, 5), literals["runavg_w5"], decimal=5)
    with pytest.raises(ora.XmhwException):
        ora.runavg(a, 2)


def test_window_roll(oisst, literals):
    i, j = _point(oisst, (-42.625, 148.125))
    x = oisst["sst"][:3, i, j]
    doy = ora.add_doy(oisst["time64"][:3])
    vals, labs = ora.window_roll(x, doy, 1)
    npt.assert_almost_equal(vals, literals["tstack"], decimal=5)
    npt.assert_array_equal(labs, [2, 3, 1, 2, 3, 1, 2])


def test_land_check(oisst, literals):
    sst = oisst["sst"]
    ts, keep, order, sshape = ora.land_check(sst, ("time", "lat", "lon"))
    assert ts.shape == (731, int(literals["land_check_ocean_cells"]))
    assert order == ["lat", "lon"] and tuple(sshape) == (8, 4)
    few = sst.copy()
    few[tuple(literals["land_check_nan_index"])] = np.nan
    assert ora.land_check(few, ("time", "lat", "lon"), anynans=True)[0].shape == (731, 11)
    assert ora.land_check(few, ("time", "lat", "lon"))[0].shape == (731, 12)
    # renamed dims, time called "c"
    assert ora.land_check(sst, ("c", "a", "b"), tdim="c")[0].shape == (731, 12)
    land = np.full(tuple(literals["land_shape"]), np.nan, dtype=np.float32)
    assert bool(literals["land_all_nan"])
    with pytest.raises(ora.XmhwException):
        ora.land_check(land, ("time", "lat", "lon"))
    with pytest.raises(ora.XmhwException):
        ora.land_check(sst[:, :0, :], ("time", "lat", "lon"))
    with pytest.raises(ora.XmhwException):
        ora.land_check(sst[:, 0, 0], ("time",))


def test_get_calendar_table():
    # xmhw_fixtures.py:349-359
    for cal, n in {"noleap": 365, "all_leap": 366, "365_day": 365, "366_day": 365.25,
                   "360_day": 360, "gregorian": 365.25, "standard": 365.25,
                   "julian": 365.25, "proleptic_gregorian": 365.25,
                   "360": 360, "365": 365, "leap": 365.25, "": 365.25}.items():
        assert ora.get_calendar(cal) == n, cal


def test_threshold_vs_reference_clim_fixtures(oisst, clim_golden):
    """test/test_xmhw.py:24-66 on the oracle: thresh decimal=6, seas decimal=4,
    unsmoothed from index 60, smoothed from index 82."""
    with pytest.raises(ora.XmhwException):
        ora.threshold_grid(oisst["sst"], oisst["time64"], smoothPercentileWidth=6)
    pts = [_point(oisst, clim_golden["point1_latlon"]), _point(oisst, clim_golden["point2_latlon"])]
    res = ora.threshold_grid(oisst["sst"], oisst["time64"], smoothPercentile=False, skipna=True)
    npt.assert_array_equal(res["doy"], np.arange(1, 367))
    for k, (i, j) in enumerate(pts, start=1):
        npt.assert_array_almost_equal(clim_golden[f"nosmooth_thresh{k}"][60:], res["thresh"][60:, i, j])
        npt.assert_array_almost_equal(clim_golden[f"nosmooth_seas{k}"][60:], res["seas"][60:, i, j], decimal=4)
        # stronger than the reference asks: every doy but Feb 29 (quirk Q3)
        m = np.ones(366, bool)
        m[59] = False
        assert np.max(np.abs(clim_golden[f"nosmooth_thresh{k}"][m] - res["thresh"][m, i, j])) < 1e-12
    res = ora.threshold_grid(oisst["sst"], oisst["time64"], skipna=True)
    for k, (i, j) in enumerate(pts, start=1):
        npt.assert_array_almost_equal(clim_golden[f"smooth_thresh{k}"][82:], res["thresh"][82:, i, j])
        npt.assert_array_almost_equal(clim_golden[f"smooth_seas{k}"][82:], res["seas"][82:, i, j], decimal=4)
        assert np.max(np.abs(clim_golden[f"smooth_thresh{k}"][82:] - res["thresh"][82:, i, j])) < 1e-12
    # land stays NaN, ocean is finite
    assert np.isnan(res["thresh"][:, ~res["keep"].reshape(8, 4)]).all()
    assert np.isfinite(res["thresh"][:, res["keep"].reshape(8, 4)]).all()


def test_skipna_is_a_no_op_on_values(oisst):
    """Quirk Q1: dropna('z') already removed NaNs, so skipna only changes the
    numpy routine, not the numbers."""
    sst = oisst["sst"].astype(np.float64).copy()
    rng = np.random.default_rng(1)
    keep = ~np.isnan(sst).all(axis=0)
    holes = rng.random(sst.shape) < 0.05
    sst[holes & keep[None]] = np.nan
    a = ora.threshold_grid(sst, oisst["time64"], skipna=True)
    b = ora.threshold_grid(sst, oisst["time64"], skipna=False)
    npt.assert_array_equal(a["thresh"], b["thresh"])
    npt.assert_array_equal(a["seas"], b["seas"])
