"""The block_average() oracle against outputs of the reference's own agg_mhw / agg_ts / agg_cats
(tests/golden/block_stats_cases.npz, made by tools/make_golden_stats.py by RUNNING xmhw/stats.py)."""
import os

import numpy as np
import numpy.testing as npt
import pandas as pd

import stats_oracle as so

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cases():
    g = np.load(os.path.join(GOLD, "mhw_features_cases.npz"))
    b = np.load(os.path.join(GOLD, "block_stats_cases.npz"))
    return g, b


def test_columns_are_the_reference_aggregation_dictionary():
    _, b = _cases()
    assert [str(c) for c in b["mhw_columns"]] == so.MHW_STATS == [a[0] for a in so.MHW_AGG]
    assert [str(c) for c in b["time_columns"]] == so.TIME_STATS


def test_event_aggregation_matches_the_reference():
    g, b = _cases()
    cols = [str(c) for c in g["columns"]]
    n = 0
    for i, (case, blockLength, mt, nb, y0) in enumerate(b["event_meta"]):
        T = int(g["offsets"][case + 1] - g["offsets"][case])
        tab = g["table"][g["table_offsets"][case]:g["table_offsets"][case + 1]]
        years = pd.date_range("2001-01-01", periods=T).year.to_numpy()
        edges = so.block_bins(int(years[0]), int(years[-1]), int(blockLength))
        assert len(edges) - 1 == nb and y0 == years[0]
        pos = tab[:, cols.index("time_start" if mt == 0 else "time_peak")].astype(int)
        got = so.agg_mhw(tab, cols, years[pos], edges)
        want = b["event_stats"][b["event_offsets"][i]:b["event_offsets"][i + 1]]
        npt.assert_allclose(got, want, rtol=1e-13, atol=0, equal_nan=True)
        n += nb
    assert n == b["event_stats"].shape[0] > 900


def test_time_axis_aggregation_matches_the_reference():
    g, b = _cases()
    k = 0
    for case in range(len(g["offsets"]) - 1):
        sl = slice(g["offsets"][case], g["offsets"][case + 1])
        ts, se, th = g["ts"][sl], g["seas"][sl], g["thresh"][sl]
        years = pd.date_range("2001-01-01", periods=ts.shape[0]).year.to_numpy()
        cats = np.floor(1 + (ts - th) / (th - se))
        for blockLength in (1, 2):
            edges = so.block_bins(int(years[0]), int(years[-1]), blockLength)
            want = b["time_stats"][b["time_offsets"][k]:b["time_offsets"][k + 1]]
            npt.assert_allclose(so.agg_time(ts, cats, years, edges), want, rtol=1e-13, atol=0, equal_nan=True)
            npt.assert_allclose(so.agg_time(ts, None, years, edges), want[:, :3], rtol=1e-13, atol=0, equal_nan=True)
            k += 1
    assert k == len(b["time_offsets"]) - 1


def test_cat_days_known_answer():
    """test/test_stats.py:38-43 of the reference"""
    cats = np.array([1, 2, 1, 1, 2, 3, 1, 4, 3, 2, 1, 1, 2], dtype=float)
    out = so.agg_time(np.zeros(13), cats, np.full(13, 2001), np.array([2001, 2002]))
    assert list(out[0, 3:]) == [6, 4, 2, 1]
