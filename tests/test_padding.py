"""maxPadLength on the CPU: the oracle restatement of xarray's interpolate_na against the published
examples of xarray's documentation and test suite, the host rules of xmhw_amd/padding.py, and the
wiring of the recipe through threshold() / detect() with oracle stand-ins as device stage."""
import datetime

import numpy as np
import numpy.testing as npt
import pytest

import oracle_fast as fast
import pad_oracle as po
from detect_standin import oracle_detect_cells
from xmhw_amd import GridSeries, climatology_series
from xmhw_amd import padding
from xmhw_amd.api import _threshold
from xmhw_amd.detect import _detect

NAN = np.nan


# ---- the oracle against xarray's own published answers ------------------------------------------
def test_oracle_docstring_example_of_interpolate_na():
    """DataArray.interpolate_na docstring: [nan, 2, 3, nan, 0] over x = 0..4 -> [nan, 2, 3, 1.5, 0]"""
    y = np.array([[NAN, 2, 3, NAN, 0]]).T
    out = po.interpolate_na(y, np.arange(5.0))
    npt.assert_array_equal(out[:, 0], [NAN, 2.0, 3.0, 1.5, 0.0])


def test_oracle_max_gap_examples_of_xarray_test_missing():
    """xarray/tests/test_missing.py: test_interpolate_na_2d (max_gap=3 on an integer coordinate) and
    test_interpolate_na_max_gap_time_specifier (the da_time fixture, hourly axis, max_gap = 3 hours)"""
    row = [1, 2, 3, 4, NAN, 6, 7, NAN, NAN, NAN, 11]
    y = np.array([row, row], dtype=np.float64).T
    out = po.interpolate_na(y, np.arange(11.0), max_gap=3)
    npt.assert_array_equal(out[:, 0], [1, 2, 3, 4, 5, 6, 7, NAN, NAN, NAN, 11])
    npt.assert_array_equal(out[:, 1], out[:, 0])
    t = np.datetime64("2001-01-01T00", "h") + np.arange(11).astype("timedelta64[h]")
    y = np.array([[NAN, 1, 2, NAN, NAN, 5, NAN, NAN, NAN, NAN, 10]]).T
    x = po.interp_index(t)
    for g in (np.timedelta64(3, "h"), datetime.timedelta(hours=3), "3h"):
        out = po.interpolate_na(y, x, max_gap=padding.max_gap_value(g, t))
        npt.assert_array_equal(out[:, 0], [NAN, 1, 2, 3, 4, 5, NAN, NAN, NAN, NAN, 10])


def test_oracle_dtype_all_nan_and_ends():
    t = np.arange("2000-01-01", "2000-01-11", dtype="datetime64[D]")
    x = po.interp_index(t)
    y = np.array([[NAN, NAN, 1, NAN, 3, NAN, NAN, NAN, 7, NAN],
                  [NAN] * 10,
                  [1, 2, 3, 4, 5, 6, 7, 8, 9, 10]], dtype=np.float32).T
    out = po.interpolate_na(y, x, max_gap=2 * 86400e9)
    assert out.dtype == np.float32
    npt.assert_array_equal(out[:, 0], [NAN, NAN, 1, 2, 3, NAN, NAN, NAN, 7, NAN])     # 2-day gap yes, 4-day gap no
    assert np.isnan(out[:, 1]).all()
    npt.assert_array_equal(out[:, 2], y[:, 2])
    # block lengths: xarray's _get_nan_block_lengths (index[0] / index[-1] stand in at the ends)
    bl = po.nan_block_lengths(y[:, 0].astype(np.float64), x) / 86400e9
    npt.assert_array_equal(bl, [2, 2, 0, 2, 0, 4, 4, 4, 0, 1])


# ---- host rules ------------------------------------------------------------------------------------
def test_max_gap_type_rules_follow_xarray():
    t = np.arange("2000-01-01", "2000-02-01", dtype="datetime64[D]")
    day = 86400e9
    assert padding.max_gap_value(np.timedelta64(5, "D"), t) == 5 * day
    assert padding.max_gap_value(datetime.timedelta(days=2, hours=12), t) == 2.5 * day
    assert padding.max_gap_value("3D", t) == 3 * day
    with pytest.raises(TypeError, match="but received int"):
        padding.max_gap_value(5, t)                       # the reference's documented call fails in xarray too
    with pytest.raises(TypeError):
        padding.max_gap_value(5.0, t)
    with pytest.raises(ValueError):
        padding.max_gap_value([1, 2], t)
    steps = np.arange(100)
    assert padding.max_gap_value(4, steps) == 4.0 and padding.max_gap_value(2.5, steps) == 2.5
    with pytest.raises(TypeError):
        padding.max_gap_value("3D", steps)
    assert padding.make_pad(None, t) is None and padding.make_pad(0, t) is None


def test_interp_index():
    t = np.array(["1970-01-02", "1970-01-03", "1970-01-05"], dtype="datetime64[D]")
    npt.assert_array_equal(padding.interp_index(t), np.array([1, 2, 4]) * 86400e9)
    npt.assert_array_equal(padding.interp_index(t), po.interp_index(t))
    npt.assert_array_equal(padding.interp_index(np.array([0, 1, 5], dtype=np.int32)), [0.0, 1.0, 5.0])
    with pytest.raises(ValueError, match="monotonically"):
        padding.interp_index(t[::-1])
    with pytest.raises(ValueError, match="duplicate"):
        padding.interp_index(np.array([0, 1, 1]))


# ---- wiring through the public host logic ---------------------------------------------------------
def _gappy(oisst, seed=3):
    rng = np.random.default_rng(seed)
    x = oisst["sst"].astype(np.float32).copy()
    T = x.shape[0]
    flat = x.reshape(T, -1)
    ocean = np.nonzero(~np.isnan(flat).all(axis=0))[0]
    for c in ocean[::2]:
        for _ in range(6):
            a = int(rng.integers(1, T - 12))
            flat[a:a + int(rng.integers(1, 9)), c] = np.nan
    flat[:3, ocean[0]] = np.nan                      # a leading run
    flat[-2:, ocean[1]] = np.nan                     # a trailing run
    return GridSeries(x, ("time", "lat", "lon"), {"time": oisst["time64"], "lat": oisst["lat"], "lon": oisst["lon"]},
                      time_encoding={"calendar": "proleptic_gregorian"})


def clim_with_pad(ts, doy, pctile, w, smooth, width, tstep, coldSpells=False, pad=None):
    if pad is not None:
        ts = po.interpolate_na(ts, pad.x, pad.max_gap)
    return fast.threshold_cells_fast(ts, doy, pctile=pctile, windowHalfWidth=w, smoothPercentile=smooth,
                                     smoothPercentileWidth=width, tstep=tstep, coldSpells=coldSpells)


def detect_with_pad(ts, seas, thresh, doy, doys, minDuration=5, joinGaps=True, maxGap=2, coldSpells=False,
                    intermediate=False, pad=None):
    if pad is not None:
        ts = po.interpolate_na(ts, pad.x, pad.max_gap)
    return oracle_detect_cells(ts, seas, thresh, doy, doys, minDuration, joinGaps, maxGap, coldSpells, intermediate)


@pytest.mark.parametrize("period", [[None, None], [2004, 2004]])
def test_threshold_and_detect_hand_the_recipe_to_the_device_stage(oisst, period):
    g = _gappy(oisst)
    gap = np.timedelta64(4, "D")
    clim = _threshold(g, clim_with_pad, maxPadLength=gap, climatologyPeriod=period)
    # by hand: slice, mask, interpolate (ns since 1970 along the SLICED axis), climatology
    time = oisst["time64"]
    sel = np.ones(time.shape[0], bool) if period[0] is None else (time.astype("datetime64[Y]").astype(int) + 1970 == 2004)
    flat = g.values[sel].reshape(int(sel.sum()), -1)
    keep = ~np.isnan(flat).all(axis=0)
    filled = po.interpolate_na(flat[:, keep], po.interp_index(time[sel]), 4 * 86400e9)
    assert np.isnan(flat[:, keep]).sum() > np.isnan(filled).sum() > 0          # some gaps closed, the long ones kept
    from xmhw_amd.calendar import add_doy
    _, th0, se0 = fast.threshold_cells_fast(filled, add_doy(time[sel]))
    got = clim["thresh"].reshape(clim["thresh"].shape[0], -1)
    npt.assert_array_equal(got[:, ~np.isnan(got).all(axis=0)], th0)
    if period[0] is None:
        th, se = climatology_series(clim, "thresh"), climatology_series(clim, "seas")
        mhw, inter = _detect(g, th, se, detect_with_pad, maxPadLength=gap, intermediate=True)
        ref, ref_inter = _detect(GridSeries(po.interpolate_na(g.values.reshape(g.values.shape[0], -1),
                                                              po.interp_index(time), 4 * 86400e9).reshape(g.values.shape),
                                            g.dims, g.coords, time_encoding=g.time_encoding),
                                 th, se, oracle_detect_cells, intermediate=True)
        npt.assert_array_equal(mhw.table, ref.table)
        npt.assert_array_equal(np.asarray(inter["ts"]), np.asarray(ref_inter["ts"]))
