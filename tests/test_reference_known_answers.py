"""Known answers held by the reference's own unit tests for the detect() stages
(test/test_features.py:46-87, test/test_identify.py:158-190 with the fixtures
test/xmhw_fixtures.py:169-332), restated as data.  The CPU tests pin the oracle to them, the
GPU tests the kernels (through the C ABI)."""
import numpy as np
import numpy.testing as npt
import pytest

import detect_oracle as det
import features_oracle as fo

# define_events() known answer: 9 daily steps, one event (xmhw_fixtures.py:185-263)
DEF_TS = np.array([15.6, 17.3, 18.2, 19.5, 19.4, 19.6, 18.1, 17.0, 15.2])
DEF_SE = np.array([15.8, 16.0, 16.2, 16.5, 16.6, 16.4, 16.6, 16.7, 16.4])
DEF_TH = np.array([16.0, 16.7, 17.6, 17.9, 18.1, 18.2, 17.3, 17.2, 17.0])
DEF_EVENT = dict(
    event=1.0, index_start=1.0, index_end=6.0, time_start=1, time_end=6, time_peak=5,
    intensity_max=3.2, intensity_mean=2.3, intensity_cumulative=13.8, severity_max=-1.42857,
    severity_mean=-1.86931, severity_cumulative=-11.215873, severity_var=0.265495,
    intensity_mean_relThresh=1.05, intensity_cumulative_relThresh=6.30, intensity_mean_abs=18.6834,
    intensity_cumulative_abs=112.1, duration_moderate=4, duration_strong=2, duration_severe=0,
    duration_extreme=0, index_peak=5.0, intensity_var=0.809938, intensity_max_relThresh=1.40,
    intensity_max_abs=19.6, intensity_var_relThresh=0.437035, intensity_var_abs=0.9495613, category=2.0,
    duration=6.0, rate_onset=0.5888889, rate_decline=1.5333333)
# the `intermediate` dataset of the same call (xmhw_fixtures.py:266-332)
DEF_BTHRESH = np.array([0, 1, 1, 1, 1, 1, 1, 0, 0], dtype=bool)
DEF_EVENTS = np.array([np.nan, 1, 1, 1, 1, 1, 1, np.nan, np.nan])


def _check_event_row(row):
    assert list(DEF_EVENT) == fo.COLUMNS
    want = np.array([DEF_EVENT[k] for k in fo.COLUMNS], dtype=np.float64)
    # xarray.testing.assert_allclose defaults, as in the reference's test
    npt.assert_allclose(row, want, rtol=1e-5, atol=1e-8)


def test_define_events_known_answer_oracle():
    b, s, e, ev = det.detect_front(DEF_TS, DEF_TH, np.arange(9), 5, True, 2)
    npt.assert_array_equal(b, DEF_BTHRESH)
    npt.assert_array_equal(ev, DEF_EVENTS)
    tab = fo.event_table(DEF_TS, DEF_SE, DEF_TH, s, e, ev)
    assert tab.shape == (1, 31)
    _check_event_row(tab[0])


def test_onset_decline_known_answer_oracle():
    # test_onset_decline (test/test_features.py:46-51, fixture xmhw_fixtures.py:169-182), tsend = 321
    onset_p, decline_p = fo.get_period(3.0, 10.0, 8.0 - 3.0, 321)
    onset = fo.get_rate(3.1, fo.get_edge(2.3, 0.3, 3.0, 0), onset_p)
    decline = fo.get_rate(3.1, fo.get_edge(1.8, 0.2, 10.0, 321), decline_p)
    npt.assert_almost_equal(onset, 0.32727273)
    npt.assert_almost_equal(decline, 0.84)


def _rates_series():
    """A series that realises the onset/decline fixture: event on steps 3..10 of 322, peak at 8."""
    ts = np.zeros(322)
    ts[2], ts[11] = 0.3, 0.2
    ts[3:11] = [2.3, 2.5, 2.6, 2.7, 2.9, 3.1, 2.0, 1.8]
    return ts, np.zeros(322), np.ones(322)


def test_onset_decline_known_answer_event_table():
    ts, se, th = _rates_series()
    _, s, e, ev = det.detect_front(ts, th, np.arange(322), 5, True, 2)
    tab = fo.event_table(ts, se, th, s, e, ev)
    assert tab.shape[0] == 1
    r = dict(zip(fo.COLUMNS, tab[0]))
    assert (r["index_start"], r["index_end"], r["index_peak"]) == (3.0, 10.0, 8.0)
    npt.assert_almost_equal(r["rate_onset"], 0.32727273)
    npt.assert_almost_equal(r["rate_decline"], 0.84)


def test_get_edge_known_answer():
    # test_get_edge (test/test_features.py:54-60)
    assert fo.get_edge(2.3, 1.7, 2, 0) == 2.0
    assert fo.get_edge(2.3, 1.7, 0, 0) == 2.3


def test_get_period_known_answer():
    # test_get_period (test/test_features.py:63-79), tsend = 25
    start, end = [0, 8, 18], [4, 15, 25]
    for peak, ons, dec in (([0, 10, 19], [1, 10.5, 19.5], [4.5, -2.5, -12]),
                           ([3, 15, 25], [3.0, 15.5, 25.5], [1.5, -7.5, 1.0])):
        got = [fo.get_period(s, e, p, 25) for s, e, p in zip(start, end, peak)]
        assert [g[0] for g in got] == ons
        assert [g[1] for g in got] == dec


def test_get_rate_known_answer():
    # test_get_rate (test/test_features.py:82-87)
    got = fo.get_rate(np.array([1.4, 2.4, 1.8]), np.array([1.0, 1.5, 2.5]), np.array([1, 10.5, 19.5]))
    npt.assert_allclose(got, [0.4, 0.08571429, -0.03589744], rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_define_events_known_answer_gpu(dtype):
    from xmhw_amd.detect_front import mhw_features_cells, mhw_filter_cells
    doy = np.arange(1, 10)
    ts = DEF_TS.astype(dtype)[:, None]
    out = mhw_filter_cells(ts, DEF_TH[:, None], doy, doy, 5, True, 2)
    npt.assert_array_equal(out["bthresh"][:, 0], DEF_BTHRESH)
    npt.assert_array_equal(out["events"][:, 0], DEF_EVENTS)
    table, offsets = mhw_features_cells(ts, DEF_SE[:, None], DEF_TH[:, None], doy, doy, 5, True, 2)
    assert list(offsets) == [0, 1]
    _check_event_row(table[0])


@pytest.mark.gpu
def test_onset_decline_known_answer_gpu():
    from xmhw_amd.detect_front import mhw_features_cells
    ts, se, th = _rates_series()
    doy = np.arange(1, 323)
    table, offsets = mhw_features_cells(ts[:, None], se[:, None], th[:, None], doy, doy, 5, True, 2)
    assert list(offsets) == [0, 1]
    r = dict(zip(fo.COLUMNS, table[0]))
    assert (r["index_start"], r["index_end"], r["index_peak"]) == (3.0, 10.0, 8.0)
    npt.assert_almost_equal(r["rate_onset"], 0.32727273)
    npt.assert_almost_equal(r["rate_decline"], 0.84)
