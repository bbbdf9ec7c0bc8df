"""Host side of detect() (xmhw/xmhw.py:310-518) without a GPU: the device stage is replaced by the
CPU oracles through the private `detect._detect(temp, th, se, compute, ...)`."""
import numpy as np
import numpy.testing as npt
import pytest

import detect_oracle as det
import features_oracle as fo
import oracle_fast as fast
from detect_standin import oracle_detect_cells
from xmhw_amd import GridSeries, XmhwException
from xmhw_amd.detect import EVENT_COLUMNS, INTER_VARIABLES, _detect, climatology_series
from xmhw_amd.api import _threshold
from test_reference_known_answers import DEF_EVENT, DEF_SE, DEF_TH, DEF_TS


def oracle_clim(ts, doy, pctile, windowHalfWidth, smoothPercentile, smoothPercentileWidth, tstep, coldSpells=False):
    return fast.threshold_cells_fast(ts, doy, pctile=pctile, windowHalfWidth=windowHalfWidth,
                                     smoothPercentile=smoothPercentile, smoothPercentileWidth=smoothPercentileWidth,
                                     tstep=tstep, coldSpells=coldSpells)


def grid(oisst):
    return GridSeries(oisst["sst"], ("time", "lat", "lon"),
                      {"time": oisst["time64"], "lat": oisst["lat"], "lon": oisst["lon"]},
                      coord_attrs={"lat": {"units": "degrees_north"}, "lon": {"units": "degrees_east"}},
                      time_encoding={"calendar": "proleptic_gregorian"})


def clims(oisst, **kw):
    clim = _threshold(grid(oisst), oracle_clim, **kw)
    return climatology_series(clim, "thresh"), climatology_series(clim, "seas")


def test_exceptions(oisst):
    g = grid(oisst)
    th, se = clims(oisst)
    with pytest.raises(XmhwException):                       # xmhw.py:373-378
        _detect(g, th, se, oracle_detect_cells, minDuration=3, maxGap=3)
    with pytest.raises(XmhwException):
        _detect(g, th, se, oracle_detect_cells, tdim="t")
    with pytest.raises(TypeError):                           # xarray's interpolate_na: a bare number on a datetime axis
        _detect(g, th, se, oracle_detect_cells, maxPadLength=2)
    # climatologies over other cells than the series
    th2 = GridSeries(th.values[:, :2], th.dims, {**th.coords, "lat": th.coords["lat"][:2]})
    with pytest.raises(XmhwException):
        _detect(g, th2, se, oracle_detect_cells)


def test_grid_events_match_per_cell_oracle(oisst):
    g = grid(oisst)
    th, se = clims(oisst)
    mhw, inter = _detect(g, th, se, oracle_detect_cells, intermediate=True)
    sst = oisst["sst"]
    T = sst.shape[0]
    stacked = sst.reshape(T, -1)            # dims already (time, lat, lon): sorted-name stacking is row-major
    keep = ~np.isnan(stacked).all(axis=0)
    assert mhw.n_cells == int(keep.sum())
    npt.assert_array_equal(mhw.cell_index, np.nonzero(keep)[0])
    from xmhw_amd import calendar as cal
    doy = cal.add_doy(oisst["time64"])
    rows = np.searchsorted(th.coords["doy"], doy)
    thc = th.values.reshape(th.values.shape[0], -1)
    sec = se.values.reshape(se.values.shape[0], -1)
    thk = thc[:, ~np.isnan(thc).all(axis=0)]
    sek = sec[:, ~np.isnan(sec).all(axis=0)]
    nev = 0
    for i, cidx in enumerate(np.nonzero(keep)[0]):
        x = stacked[:, cidx].astype(np.float64)
        _, s, e, ev = det.detect_front(x, thk[:, i], rows, 5, True, 2)
        tab = fo.event_table(x, sek[rows, i], thk[rows, i], s, e, ev)
        got = mhw.table[mhw.offsets[i]:mhw.offsets[i + 1]]
        npt.assert_allclose(got, tab, rtol=1e-12, equal_nan=True)
        nev += tab.shape[0]
        c = mhw.cell(i)
        assert c["time_start"].dtype.kind == "M"
        npt.assert_array_equal(c["time_start"], oisst["time64"][tab[:, 3].astype(int)])
    assert nev == mhw.n_events > 0
    # dense layout of the reference: (events, lat, lon), union of labels, NaN elsewhere
    dims, coords, data = mhw.to_dense(["intensity_max", "time_peak", "duration"])
    assert dims == ("events", "lat", "lon")
    labels = np.unique(mhw.table[:, 0])
    npt.assert_array_equal(coords["events"], labels)
    keepg = keep.reshape(sst.shape[1:])
    npt.assert_array_equal(coords["lat"], oisst["lat"][keepg.any(axis=1)])
    npt.assert_array_equal(coords["lon"], oisst["lon"][keepg.any(axis=0)])
    dense = data["intensity_max"]
    assert dense.shape == (labels.shape[0], int(keepg.any(axis=1).sum()), int(keepg.any(axis=0).sum()))
    assert int((~np.isnan(dense)).sum()) == mhw.n_events
    assert data["time_peak"].dtype.kind == "M" and int((~np.isnat(data["time_peak"])).sum()) == mhw.n_events
    # first ocean cell, its first event
    i0 = int(np.nonzero(keep)[0][0])
    la, lo = np.unravel_index(i0, sst.shape[1:])
    la2 = int(np.nonzero(keepg.any(axis=1))[0].tolist().index(la))
    lo2 = int(np.nonzero(keepg.any(axis=0))[0].tolist().index(lo))
    row0 = mhw.table[0]
    k = int(np.searchsorted(labels, row0[0]))
    assert dense[k, la2, lo2] == row0[EVENT_COLUMNS.index("intensity_max")]
    # attrs as annotate_ds / detect() write them
    assert mhw.attrs["xmhw_parameters"].startswith("MHW detected using: 5 days of minimum duration")
    assert "events separated by 2 or less days were joined" in mhw.attrs["xmhw_parameters"]
    assert mhw.var_attrs["intensity_cumulative"]["units"] == "degree_C day"
    assert mhw.coord_attrs["lat"] == {"units": "degrees_north"}
    # intermediate: (time, lat, lon); land NaN; event steps only
    assert inter.dims == ("time", "lat", "lon")
    assert list(inter.data_vars) == INTER_VARIABLES
    rel = inter["relSeas"]
    assert rel.shape == (T,) + dense.shape[1:]
    ev = inter["events"]
    assert np.array_equal(np.isnan(rel), np.isnan(ev))
    npt.assert_array_equal(inter["ts"][:, la2, lo2], stacked[:, i0])


def test_point_known_answer_and_intermediate():
    time = np.datetime64("2001-01-01") + np.arange(9)
    doy = np.arange(1, 10)
    ts = GridSeries(DEF_TS, ("time",), {"time": time})
    th = GridSeries(DEF_TH, ("doy",), {"doy": doy})
    se = GridSeries(DEF_SE, ("doy",), {"doy": doy})
    mhw, inter = _detect(ts, th, se, oracle_detect_cells, intermediate=True)
    assert mhw.point and mhw.n_events == 1
    dims, coords, data = mhw.to_dense()
    assert dims == ("events",) and list(coords["events"]) == [1.0]
    for k, v in DEF_EVENT.items():
        if k.startswith("time_"):
            assert data[k][0] == time[v]
        else:
            npt.assert_allclose(data[k][0], v, rtol=1e-5, atol=1e-8)
    # inter_data fixture (xmhw_fixtures.py:266-332)
    nan = np.nan
    assert inter.dims == ("index",)
    npt.assert_allclose(inter["relSeas"], [nan, 1.3, 2.0, 3.0, 2.79999, 3.2, 1.5, nan, nan], rtol=1e-5)
    npt.assert_allclose(inter["relThresh"], [nan, 0.6, 0.6, 1.6, 1.3, 1.4, 0.8, nan, nan], rtol=1e-5)
    npt.assert_allclose(inter["relThreshNorm"], [nan, 0.85714, 0.4285714, 1.142857, 0.866667, 0.77778, 1.142857, nan, nan],
                        rtol=1e-5)
    npt.assert_allclose(inter["severity"], [nan, -1.857143, -1.42857, -2.142857, -1.8666667, -1.77778, -2.142857, nan, nan],
                        rtol=1e-5)
    npt.assert_array_equal(inter["cats"], [nan, 1, 1, 2, 1, 1, 2, nan, nan])
    npt.assert_array_equal(inter["duration_moderate"], [0, 1, 1, 0, 1, 1, 0, 0, 0])
    npt.assert_array_equal(inter["duration_strong"], [0, 0, 0, 1, 0, 0, 1, 0, 0])
    npt.assert_array_equal(inter["bthresh"], [0, 1, 1, 1, 1, 1, 1, 0, 0])
    npt.assert_array_equal(inter["mabs"], [nan, 17.3, 18.2, 19.5, 19.4, 19.6, 18.1, nan, nan])
    npt.assert_array_equal(inter["seas"], [nan, 16.0, 16.2, 16.5, 16.6, 16.4, 16.6, nan, nan])


def test_cold_spells_flip(oisst):
    g = grid(oisst)
    th, se = clims(oisst, coldSpells=True, pctile=90)
    mhw = _detect(g, th, se, oracle_detect_cells, coldSpells=True)
    assert mhw.n_events > 0
    t = dict(zip(EVENT_COLUMNS, mhw.table.T))
    # the series was negated for detection, the intensities are flipped back (features.py:298-315)
    assert (t["intensity_max"] < 0).all() and (t["intensity_mean_abs"] > 0).all()
    assert (t["intensity_var"] >= 0).all() or np.isnan(t["intensity_var"]).any()
    assert "cold events were detected" in mhw.attrs["xmhw_parameters"]
