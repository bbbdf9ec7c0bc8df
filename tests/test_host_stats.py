"""Host side of block_average() (xmhw_amd/stats.py) with the device stage replaced by the oracle:
argument checks and messages of xmhw/stats.py:112-121, bins of :130, grid placement."""
import os

import numpy as np
import numpy.testing as npt
import pytest

import stats_oracle as so
from xmhw_amd import XmhwException
from xmhw_amd.detect import EventDataset
from xmhw_amd.stats import MHW_STATS, block_average, block_bins

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def oracle_compute(table, offsets, years, edges, mtime, ts, cats):
    cols = EventDataset.columns
    C = offsets.shape[0] - 1
    out = {k: np.empty((len(edges) - 1, C)) for k in MHW_STATS}
    for c in range(C):
        tab = table[offsets[c]:offsets[c + 1]]
        pos = tab[:, cols.index(mtime)].astype(int)
        a = so.agg_mhw(tab, cols, years[pos], edges)
        for j, k in enumerate(MHW_STATS):
            out[k][:, c] = a[:, j]
    if ts is not None:
        names = so.TIME_STATS[:3] + (so.TIME_STATS[3:] if cats is not None else [])
        for k in names:
            out[k] = np.empty((len(edges) - 1, C))
        for c in range(C):
            a = so.agg_time(ts[:, c].astype(np.float64), None if cats is None else cats[:, c], years, edges)
            for j, k in enumerate(names):
                out[k][:, c] = a[:, j]
        if cats is not None:
            out["total_days"] = sum(out[k] for k in so.TIME_STATS[3:])
    return out


def _mhw(point=False):
    g = np.load(os.path.join(GOLD, "mhw_features_cases.npz"))
    cases = [0, 3, 6, 9] if not point else [0]
    T = min(int(g["offsets"][c + 1] - g["offsets"][c]) for c in cases)
    tabs = []
    for c in cases:
        t = g["table"][g["table_offsets"][c]:g["table_offsets"][c + 1]]
        tabs.append(t[t[:, 2] < T])                      # events inside the common axis
    table = np.concatenate(tabs)
    offsets = np.cumsum([0] + [t.shape[0] for t in tabs])
    time = np.arange("2001-01-01", T, dtype="datetime64[D]") if False else np.datetime64("2001-01-01") + np.arange(T)
    if point:
        return EventDataset(table, offsets, time, np.array([0]), np.array([True]), (), (), {}, {}, {}, {}, True), T
    keep = np.array([True, False, True, True, True, False])       # 2 x 3 grid, two land cells, no all-land line
    return EventDataset(table, offsets, time, np.nonzero(keep)[0], keep, ("lat", "lon"), (2, 3),
                        {"lat": np.array([10.0, 20.0]), "lon": np.array([1.0, 2.0, 3.0])}, {}, {}, {}, False), T


def test_argument_checks_follow_the_reference():
    mhw, _ = _mhw()
    with pytest.raises(XmhwException, match="has to be passed"):
        block_average(mhw, _compute=oracle_compute)
    with pytest.raises(XmhwException, match="To remove missing values"):
        block_average(mhw, period=[2001, 2003], removeMissing=True, _compute=oracle_compute)
    with pytest.raises(XmhwException):
        block_average("not an EventDataset")


@pytest.mark.parametrize("blockLength", [1, 2])
def test_grid_layout_and_bins(blockLength):
    mhw, T = _mhw()
    years = (mhw.time.astype("datetime64[Y]").astype(int) + 1970)
    blk = block_average(mhw, period=[int(years[0]), int(years[-1])], blockLength=blockLength, _compute=oracle_compute,
                        split=True)
    edges = block_bins([years[0], years[-1]], blockLength)
    npt.assert_array_equal(blk.year_bins, edges)
    npt.assert_array_equal(blk.coords["years"], edges[:-1])
    assert blk.dims == ("years", "lat", "lon") and blk["ecount"].shape == (len(edges) - 1, 2, 3)
    assert np.isnan(blk["ecount"][:, 0, 1]).all() and np.isnan(blk["duration"][:, 1, 2]).all()     # land
    assert blk["ecount"][:, 0, 0].sum() == mhw.offsets[1] - mhw.offsets[0]
    total = sum(np.nansum(blk["ecount"][:, i, j]) for i in range(2) for j in range(3))
    assert total == mhw.n_events


def test_point_and_time_statistics():
    mhw, T = _mhw(point=True)
    from xmhw_amd import GridSeries
    from xmhw_amd.detect import InterDataset
    rng = np.random.default_rng(0)
    ts = (15 + rng.normal(size=T)).astype(np.float32)
    cats = np.floor(rng.uniform(-2, 5, size=T))
    blk = block_average(mhw, dstime=GridSeries(ts, ("time",), {"time": mhw.time}), _compute=oracle_compute)
    assert blk.dims == ("years",) and set(["ts_mean", "ts_max", "ts_min"]) <= set(blk.data_vars)
    assert "moderate_days" not in blk.data_vars
    inter = InterDataset({"ts": ts, "cats": cats}, ("index",), {})
    blk2 = block_average(mhw, dstime=inter, blockLength=2, _compute=oracle_compute)
    npt.assert_array_equal(blk2["total_days"], blk2["moderate_days"] + blk2["strong_days"] + blk2["severe_days"]
                           + blk2["extreme_days"])
    assert blk2["total_days"].sum() == np.isin(cats, [1, 2, 3, 4]).sum()
