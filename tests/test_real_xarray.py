"""threshold() / detect() with a REAL xarray.DataArray, when xarray is importable (it is not in the
build image nor, so far, on the GPU box: the tests then SKIP with that reason, so that a verdict can
see whether they ran; tests/test_xarray_branch.py exercises the same branch with a stand-in).
What is asserted is the reference's return contract: xmhw/xmhw.py:204-219 (dims (doy, lat, lon),
scalar `quantile` coordinate, Dataset keys), docs/gettingstarted.rst:37-45, identify.py:561-563 (doy
attributes), docs/threshold.rst:104-108 (an all-land grid line vanishes)."""
import os

import numpy as np
import numpy.testing as npt
import pytest

xr = pytest.importorskip("xarray", reason="xarray is not installed on this machine")
import oracle_fast as fast

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _dataarray():
    g = np.load(os.path.join(GOLD, "oisst_2003_2004.npz"))
    time = np.datetime64("2003-01-01") + g["time"].astype("timedelta64[D]")
    sst = g["sst"].copy()
    sst[:, 0, :] = np.nan                                   # the first latitude line is all land
    da = xr.DataArray(sst, dims=("time", "lat", "lon"), coords={"time": time, "lat": g["lat"], "lon": g["lon"]},
                      attrs={"units": "degC"}, name="sst")
    da["lat"].attrs["units"] = "degrees_north"
    return da


def _oracle_compute(ts, doy, pctile, w, smooth, width, tstep, cold=False):
    return fast.threshold_cells_fast(ts, doy, pctile=pctile, windowHalfWidth=w, smoothPercentile=smooth,
                                     smoothPercentileWidth=width, tstep=tstep, coldSpells=cold)


def _check_contract(ds, da):
    assert isinstance(ds, xr.Dataset) and set(ds.data_vars) == {"thresh", "seas"}
    assert ds["thresh"].dims == ("doy", "lat", "lon")
    assert ds["doy"].dtype == np.int64 and ds["doy"].values[0] == 1 and ds["doy"].values[-1] == 366
    assert ds["doy"].attrs == {"units": "1", "long_name": "Day of the year"}
    assert ds["quantile"].ndim == 0 and float(ds["quantile"]) == 0.9
    assert ds["thresh"].attrs["units"] == "degree_C" and ds["seas"].attrs["units"] == "degree_C"     # quirk Q9
    assert "xmhw_parameters" in ds.attrs and "history" in ds.attrs
    assert ds["lat"].attrs.get("units") == "degrees_north"
    # the all-land latitude line is gone, land cells inside the remaining grid are NaN
    assert ds.sizes["lat"] == da.sizes["lat"] - 1
    npt.assert_array_equal(ds["lat"].values, da["lat"].values[1:])
    assert ds["thresh"].dtype == np.float64


def test_threshold_with_real_xarray_host_logic():
    from xmhw_amd.api import _threshold
    da = _dataarray()
    _check_contract(_threshold(da, _oracle_compute), da)


@pytest.mark.gpu
def test_threshold_and_detect_with_real_xarray_on_the_gpu():
    import xmhw_amd
    from xmhw_amd._lib import require_gpu
    require_gpu()
    da = _dataarray()
    ds = xmhw_amd.threshold(da)
    _check_contract(ds, da)
    ref = xmhw_amd.api._threshold(da, _oracle_compute)
    npt.assert_allclose(ds["thresh"].values, ref["thresh"].values, rtol=1e-12, equal_nan=True)
    mhw = xmhw_amd.detect(da, ds["thresh"], ds["seas"])
    assert isinstance(mhw, xr.Dataset) and mhw["duration"].dims == ("events", "lat", "lon")
    assert str(mhw["time_start"].dtype).startswith("datetime64")


def test_pad_oracle_equals_real_interpolate_na():
    """oracle/pad_oracle.py (the checker of the pad_gaps kernel) against the real
    DataArray.interpolate_na(dim, max_gap=...), including its type rules"""
    import pad_oracle as po
    from xmhw_amd import padding
    rng = np.random.default_rng(5)
    T, C = 400, 12
    time = np.datetime64("2001-01-01") + np.arange(T).astype("timedelta64[D]")
    for dtype in (np.float32, np.float64):
        y = rng.normal(size=(T, C)).astype(dtype)
        for c in range(C):
            for _ in range(8):
                a = int(rng.integers(0, T))
                y[a:a + int(rng.integers(1, 9)), c] = np.nan
        y[:, 0] = np.nan
        da = xr.DataArray(y, dims=("time", "cell"), coords={"time": time})
        for gap in (np.timedelta64(2, "D"), np.timedelta64(5, "D"), "3D"):
            ref = da.interpolate_na(dim="time", max_gap=gap).values
            got = po.interpolate_na(y, po.interp_index(time), padding.max_gap_value(gap, time))
            npt.assert_array_equal(got, ref)
    with pytest.raises(TypeError):
        da.interpolate_na(dim="time", max_gap=5)            # what the reference's documented call runs into
    with pytest.raises(TypeError):
        padding.max_gap_value(5, time)
