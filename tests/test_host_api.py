"""Host side of threshold() (xmhw/xmhw.py:38-247) without a GPU: the device
stage is replaced by the CPU oracle through the private `api._threshold(temp, compute, ...)`, so that
validation, period slicing, land masking, doy labels, unstacking and attrs are
checked against the oracle's threshold_grid()."""
from datetime import date

import numpy as np
import numpy.testing as npt
import pytest

import xmhw_oracle as ora
import oracle_fast as fast
import xmhw_amd
from xmhw_amd import GridSeries, XmhwException, threshold
from xmhw_amd.api import _threshold


def oracle_compute(ts, doy, pctile, windowHalfWidth, smoothPercentile, smoothPercentileWidth,
                   tstep, coldSpells=False):
    return fast.threshold_cells_fast(ts, doy, pctile=pctile, windowHalfWidth=windowHalfWidth,
                                     smoothPercentile=smoothPercentile,
                                     smoothPercentileWidth=smoothPercentileWidth, tstep=tstep,
                                     coldSpells=coldSpells)


def grid(oisst):
    return GridSeries(oisst["sst"], ("time", "lat", "lon"),
                      {"time": oisst["time64"], "lat": oisst["lat"], "lon": oisst["lon"]},
                      attrs={"units": "Celsius"},
                      coord_attrs={"lat": {"units": "degrees_north"}, "lon": {"units": "degrees_east"}},
                      time_encoding={"calendar": "proleptic_gregorian"})


def test_exceptions_like_the_reference(oisst):
    g = grid(oisst)
    with pytest.raises(XmhwException):                       # xmhw.py:103-104
        _threshold(g, oracle_compute, smoothPercentileWidth=6)
    with pytest.raises(XmhwException):                       # xmhw.py:105-109
        _threshold(g, oracle_compute, tdim="t")
    land = GridSeries(np.full((731, 3, 2), np.nan, np.float32), ("time", "lat", "lon"),
                      {"time": oisst["time64"], "lat": np.arange(3), "lon": np.arange(2)})
    with pytest.raises(XmhwException):                       # identify.py:527-528
        _threshold(land, oracle_compute)
    empty = GridSeries(oisst["sst"][:, :0], ("time", "lat", "lon"),
                       {"time": oisst["time64"], "lat": np.arange(0), "lon": oisst["lon"]})
    with pytest.raises(XmhwException):                       # identify.py:514-516
        _threshold(empty, oracle_compute)
    with pytest.raises(TypeError):                           # xarray's interpolate_na: a bare number on a datetime axis
        _threshold(g, oracle_compute, maxPadLength=3)


def test_grid_layout_attrs_and_values(oisst):
    ds = _threshold(grid(oisst), oracle_compute, skipna=True)
    ref = ora.threshold_grid(oisst["sst"], oisst["time64"])
    keep = ref["keep"].reshape(8, 4)
    rows, cols = keep.any(axis=1), keep.any(axis=0)
    assert ds.dims == ("doy", "lat", "lon")
    # all-land lon columns vanish from the output grid (docs/threshold.rst:104-108)
    npt.assert_array_equal(ds.coords["lat"], oisst["lat"][rows])
    npt.assert_array_equal(ds.coords["lon"], oisst["lon"][cols])
    npt.assert_array_equal(ds.coords["doy"], np.arange(1, 367))
    assert ds.coords["doy"].dtype == np.int64
    npt.assert_allclose(ds["thresh"], ref["thresh"][:, rows][:, :, cols], rtol=1e-13, equal_nan=True)
    npt.assert_allclose(ds["seas"], ref["seas"][:, rows][:, :, cols], rtol=1e-13, equal_nan=True)
    assert ds.quantile == 0.9
    assert ds.var_attrs["thresh"]["units"] == "degree_C"        # quirk Q9
    assert ds.coord_attrs["doy"] == {"units": "1", "long_name": "Day of the year"}
    assert ds.coord_attrs["lat"] == {"units": "degrees_north"}
    assert ds.attrs["source"] == "xmhw code: https://github.com/coecms/xmhw"
    assert ds.attrs["history"].startswith(str(date.today()))
    p = ds.attrs["xmhw_parameters"]
    assert "90 percentile" in p and "2003-2004" in p and "window half width used for percentile is 5" in p
    assert "NaNs where skipped" in p and "moving average window" in p and "31" in p


def test_point_path(oisst):
    x = oisst["sst"][:, 1, 2]
    g = GridSeries(x, ("time",), {"time": oisst["time64"]})
    ds = _threshold(g, oracle_compute, smoothPercentile=False)
    ref = ora.threshold_grid(x, oisst["time64"], dims=("time",), smoothPercentile=False)
    assert ds.dims == ("doy",) and ds["thresh"].shape == (366,)
    npt.assert_allclose(ds["thresh"], ref["thresh"], rtol=1e-13)
    npt.assert_allclose(ds["seas"], ref["seas"], rtol=1e-13)


def test_climatology_period_both_truthy_only(oisst):
    g = grid(oisst)
    a = _threshold(g, oracle_compute, climatologyPeriod=[2004, 2004], smoothPercentile=False)
    ref = ora.threshold_grid(oisst["sst"], oisst["time64"], climatologyPeriod=(2004, 2004),
                             smoothPercentile=False)
    keep = ref["keep"].reshape(8, 4)
    npt.assert_allclose(a["thresh"], ref["thresh"][:, keep.any(axis=1)][:, :, keep.any(axis=0)],
                        rtol=1e-13, equal_nan=True)
    assert "2004-2004" in a.attrs["xmhw_parameters"]
    b = _threshold(g, oracle_compute, climatologyPeriod=[2004, None], smoothPercentile=False)
    assert "2003-2004" in b.attrs["xmhw_parameters"]          # quirk Q7: ignored unless both set


def test_anynans_drops_cells_and_sorted_dim_order(oisst):
    sst = oisst["sst"].copy()
    sst[245, 1, 2] = np.nan
    # dims given as (lon, time, lat): the stacked order is still sorted names (lat, lon)
    v = np.transpose(sst, (2, 0, 1))
    g = GridSeries(v, ("lon", "time", "lat"), {"time": oisst["time64"], "lat": oisst["lat"], "lon": oisst["lon"]})
    ds = _threshold(g, oracle_compute, anynans=True, smoothPercentile=False)
    assert ds.dims == ("doy", "lat", "lon")
    i = list(ds.coords["lat"]).index(oisst["lat"][1])
    j = list(ds.coords["lon"]).index(oisst["lon"][2])
    assert np.isnan(ds["thresh"][:, i, j]).all()
    assert "any grid point with even only 1 NaN" in ds.attrs["xmhw_parameters"]
    ds2 = _threshold(g, oracle_compute, smoothPercentile=False)
    assert np.isfinite(ds2["thresh"][:, i, j]).all()


def test_360_day_calendar_forces_tstep():
    # 3 years sampled every 5 days (73 steps per year), declared as a 360-day calendar
    time = np.concatenate([np.datetime64(f"{y}-01-01") + 5 * np.arange(73).astype("timedelta64[D]")
                           for y in (2001, 2002, 2003)])
    rng = np.random.default_rng(0)
    x = rng.normal(size=(time.shape[0], 2, 2)).astype(np.float32)
    g = GridSeries(x, ("time", "y", "x"), {"time": time, "y": np.arange(2), "x": np.arange(2)},
                   time_encoding={"calendar": "360_day"})
    seen = {}

    def spy(ts, doy, *a):
        seen["doy"] = doy
        seen["tstep"] = a[4]
        return oracle_compute(ts, doy, *a)
    _threshold(g, spy, windowHalfWidth=1, smoothPercentileWidth=3)
    assert seen["tstep"] is True                               # xmhw.py:143-144
    npt.assert_array_equal(seen["doy"], ora.add_doy(time, keep_tstep=True))


def test_add_doy_and_calendar_match_oracle(oisst, literals):
    npt.assert_array_equal(xmhw_amd.add_doy(oisst["time64"]), literals["oisst_doy"])
    assert xmhw_amd.add_doy(oisst["time64"]).dtype == np.int64
    with pytest.raises(XmhwException):
        xmhw_amd.add_doy(oisst["time64"][:700], keep_tstep=True)
    for cal in ("noleap", "all_leap", "365_day", "366_day", "360_day", "gregorian", "standard",
                "julian", "proleptic_gregorian", "360", "leap", "", "bogus"):
        assert xmhw_amd.get_calendar(cal) == ora.get_calendar(cal)


def test_land_check_matches_reference_counts(oisst, literals):
    ts, keep, order, sshape = xmhw_amd.land_check(oisst["sst"], ("time", "lat", "lon"))
    assert ts.shape == (731, 12) and order == ["lat", "lon"] and sshape == (8, 4)
    assert ts.flags["C_CONTIGUOUS"]
    few = oisst["sst"].copy()
    few[tuple(literals["land_check_nan_index"])] = np.nan
    assert xmhw_amd.land_check(few, ("time", "lat", "lon"), anynans=True)[0].shape == (731, 11)
    with pytest.raises(XmhwException):
        xmhw_amd.land_check(oisst["sst"][:, 0, 0], ("time",))


def test_product_path_fails_loudly_without_gpu(oisst):
    """No CPU fallback: without a usable HIP device threshold() must raise."""
    from xmhw_amd._lib import hip
    h = hip()
    try:
        h.device_count()
    except h.HipError:
        with pytest.raises(Exception) as e:
            threshold(grid(oisst))
        assert "hip" in type(e.value).__name__.lower() or "hip" in str(e.value).lower()
    else:
        pytest.skip("a GPU is present")


def test_packed_recipe_ignores_a_fill_value_no_int16_code_can_equal():
    """a NaN / infinite / fractional / out-of-range _FillValue on an int16 variable means "nothing is missing";
    int() of a NaN or an infinity used to raise before the range check was reached"""
    import xmhw_amd.device as dev

    class View:
        dtype = np.dtype("<i2")

        def __init__(self, fill):
            self.decode = {"scale": 0.01, "offset": 0.0, "fill": fill, "out": "float32"}

    for fill, want in ((float("nan"), None), (float("inf"), None), (-float("inf"), None), (1.5, None), (40000, None),
                       (-999, -999), (-32768.0, -32768), (None, None)):
        assert dev.packed_recipe(View(fill))["fill"] == want, fill
