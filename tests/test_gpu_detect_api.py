"""detect() end to end on the GPU (public entry point, HIP path through the C ABI) against the
same host logic driven by the CPU oracles, plus the per-step `intermediate` kernel against columns
produced by RUNNING the reference's mhw_df() (tests/golden/mhw_df_cases.npz)."""
import os

import numpy as np
import numpy.testing as npt
import pytest

from detect_standin import oracle_detect_cells
from test_reference_known_answers import DEF_EVENT, DEF_SE, DEF_TH, DEF_TS

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _grid(oisst, dtype=np.float32):
    from xmhw_amd import GridSeries
    return GridSeries(oisst["sst"].astype(dtype), ("time", "lat", "lon"),
                      {"time": oisst["time64"], "lat": oisst["lat"], "lon": oisst["lon"]},
                      time_encoding={"calendar": "proleptic_gregorian"})


def _compare(a, b, inter_a=None, inter_b=None, min_events=1):
    npt.assert_array_equal(a.offsets, b.offsets)
    assert a.table.shape == b.table.shape and a.n_events >= min_events
    npt.assert_allclose(a.table, b.table, rtol=1e-9, atol=1e-11, equal_nan=True)
    for k in (0, 1, 2, 3, 4, 5, 17, 18, 19, 20, 21, 27, 28):       # integer-valued columns: exact
        npt.assert_array_equal(a.table[:, k], b.table[:, k])
    if inter_a is not None:
        assert inter_a.dims == inter_b.dims and list(inter_a.data_vars) == list(inter_b.data_vars)
        for k in inter_a.data_vars:
            # elementwise float64 arithmetic in the same order: bit-identical
            npt.assert_array_equal(np.asarray(inter_a[k], dtype=np.float64), np.asarray(inter_b[k], dtype=np.float64),
                                   err_msg=k)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cold", [False, True])
def test_threshold_then_detect_matches_oracle_pipeline(oisst, dtype, cold):
    from xmhw_amd import climatology_series, detect, threshold
    from xmhw_amd.detect import _detect
    g = _grid(oisst, dtype)
    clim = threshold(g, coldSpells=cold)
    th, se = climatology_series(clim, "thresh"), climatology_series(clim, "seas")
    mhw, inter = detect(g, th, se, coldSpells=cold, intermediate=True)
    ref, ref_inter = _detect(g, th, se, oracle_detect_cells, coldSpells=cold, intermediate=True)
    _compare(mhw, ref, inter, ref_inter)
    dims, coords, data = mhw.to_dense(["duration", "time_start"])
    assert dims == ("events", "lat", "lon") and data["duration"].shape[0] == coords["events"].shape[0]


@pytest.mark.parametrize("params", [(3, True, 1), (5, False, 2), (8, True, 4)])
def test_detect_parameters_and_batching(oisst, params):
    from xmhw_amd import climatology_series, threshold
    from xmhw_amd.detect import _detect
    from xmhw_amd.detect_front import detect_cells
    m, jg, gap = params
    g = _grid(oisst)
    clim = threshold(g, smoothPercentile=False, pctile=80)
    th, se = climatology_series(clim, "thresh"), climatology_series(clim, "seas")

    def small_batches(*a, **k):
        return detect_cells(*a, **k, max_batch_bytes=200_000)      # a few cells per batch

    mhw, inter = _detect(g, th, se, small_batches, minDuration=m, joinGaps=jg, maxGap=gap, intermediate=True)
    ref, ref_inter = _detect(g, th, se, oracle_detect_cells, minDuration=m, joinGaps=jg, maxGap=gap, intermediate=True)
    _compare(mhw, ref, inter, ref_inter, min_events=0 if m == 8 else 1)


def test_point_known_answer():
    """define_events() known answer of the reference (test/test_identify.py:158-190) through detect()."""
    from xmhw_amd import GridSeries, detect
    time = np.datetime64("2001-01-01") + np.arange(9)
    doy = np.arange(1, 10)
    mhw, inter = detect(GridSeries(DEF_TS, ("time",), {"time": time}), GridSeries(DEF_TH, ("doy",), {"doy": doy}),
                        GridSeries(DEF_SE, ("doy",), {"doy": doy}), intermediate=True)
    dims, coords, data = mhw.to_dense()
    assert dims == ("events",) and list(coords["events"]) == [1.0]
    for k, v in DEF_EVENT.items():
        if k.startswith("time_"):
            assert data[k][0] == time[v]
        else:
            npt.assert_allclose(data[k][0], v, rtol=1e-5, atol=1e-8)
    nan = np.nan
    npt.assert_allclose(inter["relThreshNorm"], [nan, 0.85714, 0.4285714, 1.142857, 0.866667, 0.77778, 1.142857, nan, nan],
                        rtol=1e-5)
    npt.assert_array_equal(inter["cats"], [nan, 1, 1, 2, 1, 1, 2, nan, nan])
    npt.assert_array_equal(inter["duration_strong"], [0, 0, 0, 1, 0, 0, 1, 0, 0])


def test_intermediate_kernel_matches_reference_mhw_df():
    from xmhw_amd.detect_front import detect_cells
    g = np.load(os.path.join(GOLD, "mhw_features_cases.npz"))
    d = np.load(os.path.join(GOLD, "mhw_df_cases.npz"))
    offs = g["offsets"]
    cols = list(d["columns"])
    for j, case in enumerate(d["cases"]):
        sl = slice(offs[case], offs[case + 1])
        ts, se, th = g["ts"][sl], g["seas"][sl], g["thresh"][sl]
        T = ts.shape[0]
        m, jg, gap = (int(v) for v in g["params"][case])
        doy = np.arange(1, T + 1)                  # one climatology row per step
        r = detect_cells(ts[:, None], se[:, None], th[:, None], doy, doy, m, bool(jg), gap, intermediate=True)
        want = d["values"][:, d["offsets"][j]:d["offsets"][j + 1]]
        for k, name in enumerate(cols):
            npt.assert_array_equal(np.asarray(r["inter"][name][:, 0], dtype=np.float64), want[k],
                                   err_msg=f"case {case} {name}")


def test_tstep_axis_end_to_end():
    """tstep=True (labels = step of the year; no Feb-29 handling): threshold() -> detect() on a
    365-steps-per-year axis against the oracle-driven host pipeline."""
    from xmhw_amd import GridSeries, climatology_series, detect, threshold
    from xmhw_amd.detect import _detect
    t = np.arange("2001-01-01", "2010-01-01", dtype="datetime64[D]")
    t = t[~((t.astype("datetime64[M]").astype(int) % 12 == 1) & ((t - t.astype("datetime64[M]")).astype(int) == 28))]
    assert t.shape[0] == 9 * 365
    rng = np.random.default_rng(12)
    T = t.shape[0]
    anom = np.zeros((T, 3, 4))
    e = rng.normal(size=(T, 3, 4))
    for k in range(1, T):
        anom[k] = 0.9 * anom[k - 1] + e[k]
    x = (15 + 3 * np.sin(2 * np.pi * np.arange(T)[:, None, None] / 365.0) + anom).astype(np.float32)
    x[:, 0, 0] = np.nan
    g = GridSeries(x, ("time", "lat", "lon"), {"time": t, "lat": np.arange(3), "lon": np.arange(4)})
    clim = threshold(g, tstep=True, smoothPercentileWidth=11)
    assert clim["thresh"].shape[0] == 365
    th, se = climatology_series(clim, "thresh"), climatology_series(clim, "seas")
    mhw = detect(g, th, se, tstep=True)
    ref = _detect(g, th, se, oracle_detect_cells, tstep=True)
    _compare(mhw, ref)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cold", [False, True])
def test_threshold_detect_equals_the_two_calls(oisst, dtype, cold, monkeypatch):
    """threshold_detect() (series uploaded once, kept in HBM between the two stages) returns what
    threshold() followed by detect() returns; the second stage must not upload the series again."""
    import xmhw_amd.device as dev
    import xmhw_amd.detect_front as front
    from xmhw_amd import climatology_series, detect, threshold, threshold_detect
    g = _grid(oisst, dtype)
    clim0 = threshold(g, coldSpells=cold)
    mhw0 = detect(g, climatology_series(clim0, "thresh"), climatology_series(clim0, "seas"), coldSpells=cold)

    uploads = []
    real = dev.compact_columns

    def counting(stacked, lo, hi, anynans):
        uploads.append(stacked.shape)
        return real(stacked, lo, hi, anynans)

    monkeypatch.setattr(dev, "compact_columns", counting)
    clim1, mhw1 = threshold_detect(g, coldSpells=cold)
    T = g.values.shape[0]
    assert not [s for s in uploads if s[0] == T], uploads       # only the (D, N) climatologies were uploaded
    npt.assert_array_equal(clim1["thresh"], clim0["thresh"])
    npt.assert_array_equal(clim1["seas"], clim0["seas"])
    npt.assert_array_equal(mhw1.offsets, mhw0.offsets)
    npt.assert_array_equal(mhw1.table, mhw0.table)
    assert mhw1.n_events > 0
    clim2, mhw2, inter2 = threshold_detect(g, coldSpells=cold, intermediate=True)
    npt.assert_array_equal(mhw2.table, mhw0.table)
    assert "ts" in inter2.data_vars or len(inter2.data_vars) > 0


def test_threshold_detect_with_a_climatology_period_falls_back(oisst):
    """a climatology period cuts the series for threshold(): nothing is reused, results equal the two calls"""
    from xmhw_amd import climatology_series, detect, threshold, threshold_detect
    g = _grid(oisst)
    period = [2004, 2004]          # the leap year of the 2003-2004 fixture: every doy has a row
    clim0 = threshold(g, climatologyPeriod=period)
    mhw0 = detect(g, climatology_series(clim0, "thresh"), climatology_series(clim0, "seas"))
    clim1, mhw1 = threshold_detect(g, climatologyPeriod=period)
    npt.assert_array_equal(clim1["thresh"], clim0["thresh"])
    npt.assert_array_equal(mhw1.table, mhw0.table)
    with pytest.raises(Exception, match="Maximum gap"):
        threshold_detect(g, maxGap=5, minDuration=5)
