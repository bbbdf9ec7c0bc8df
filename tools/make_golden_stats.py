#!/usr/bin/env python3
"""Golden vectors for block_average(), produced by RUNNING the reference's own
pandas aggregation -- agg_mhw(), agg_ts(), agg_cats(), cat_days() (xmhw/stats.py:322-428) -- in the
build container:

    python tools/make_golden_stats.py      # writes tests/golden/block_stats_cases.npz

xmhw/stats.py imports xarray and dask at module level (neither is installed here); the aggregation
functions themselves are pure pandas.  As in make_golden_detect.py the two modules (and the
xarray-based land_check import) are replaced by inert placeholders for the import only.
Inputs: the per-event tables of tests/golden/mhw_features_cases.npz (themselves outputs of the
reference's mhw_features()) re-dated on a daily axis that starts on 2001-01-01, plus the per-step
ts / cats columns of the same cases.  Bins follow block_average() (stats.py:130):
range(period[0], period[1] + blockLength + 1, blockLength), pd.cut(..., right=False).
Only DATA is stored -- no reference source text.
"""
import os
import sys
import types
import warnings

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden_detect import REF

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")
OUT = os.path.join(GOLD, "block_stats_cases.npz")
MHW_STATS = ["ecount", "duration", "intensity_max", "intensity_max_max", "intensity_mean", "intensity_cumulative",
             "total_icum", "intensity_mean_relThresh", "intensity_cumulative_relThresh", "severity_mean",
             "severity_cumulative", "intensity_mean_abs", "intensity_cumulative_abs", "rate_onset", "rate_decline"]
CAT_STATS = ["ts_mean", "ts_max", "ts_min", "moderate_days", "strong_days", "severe_days", "extreme_days"]


def load_reference_stats():
    fake_xr, fake_dask = types.ModuleType("xarray"), types.ModuleType("dask")

    def delayed(*a, **k):
        if a and callable(a[0]):
            return a[0]
        return lambda f: f
    fake_dask.delayed = delayed
    saved = {k: sys.modules.get(k) for k in ("xarray", "dask")}
    sys.modules["xarray"], sys.modules["dask"] = fake_xr, fake_dask
    sys.path.insert(0, REF)
    try:
        import xmhw.stats as stats
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return stats


def main():
    stats = load_reference_stats()
    warnings.simplefilter("ignore")
    g = np.load(os.path.join(GOLD, "mhw_features_cases.npz"))
    cols = [str(c) for c in g["columns"]]
    ci = {c: i for i, c in enumerate(cols)}
    offs, toffs = g["offsets"], g["table_offsets"]
    ev_out, ts_out, cat_out, meta = [], [], [], []
    for case in range(len(offs) - 1):
        T = int(offs[case + 1] - offs[case])
        tab = g["table"][toffs[case]:toffs[case + 1]]
        ts = g["ts"][offs[case]:offs[case + 1]]
        se = g["seas"][offs[case]:offs[case + 1]]
        th = g["thresh"][offs[case]:offs[case + 1]]
        time = pd.date_range("2001-01-01", periods=T)
        years = time.year.to_numpy()
        for blockLength in (1, 2):
            bins = range(int(years[0]), int(years[-1]) + blockLength + 1, blockLength)
            nb = len(bins) - 1
            for mtime in ("time_start", "time_peak"):
                df = pd.DataFrame({c: tab[:, ci[c]] for c in cols})
                tg = pd.Series(time[tab[:, ci[mtime]].astype(int)].year if tab.shape[0] else np.zeros(0, int))
                blk = stats.agg_mhw(df, tg, bins)
                assert len(blk) == nb
                ev_out.append(blk[MHW_STATS].to_numpy(dtype=np.float64))
                meta.append((case, blockLength, 0 if mtime == "time_start" else 1, nb, int(years[0])))
            # the time-axis statistics: ts alone, and ts + categories
            cats = np.floor(1 + (ts - th) / (th - se))
            dft = pd.DataFrame({"ts": ts, "cats": cats}, index=time)
            tgt = pd.Series(years, index=time)
            b_ts = stats.agg_ts(dft[["ts"]], tgt, bins)
            b_ct = stats.agg_cats(dft, tgt, bins)
            ts_out.append(b_ts[["ts_mean", "ts_max", "ts_min"]].to_numpy(dtype=np.float64))
            cat_out.append(b_ct[CAT_STATS].to_numpy(dtype=np.float64))
            np.testing.assert_array_equal(ts_out[-1], cat_out[-1][:, :3])
    ev_off = np.cumsum([0] + [a.shape[0] for a in ev_out])
    t_off = np.cumsum([0] + [a.shape[0] for a in cat_out])
    np.savez_compressed(OUT, event_stats=np.concatenate(ev_out), event_offsets=ev_off, event_meta=np.array(meta),
                        time_stats=np.concatenate(cat_out), time_offsets=t_off,
                        mhw_columns=np.array(MHW_STATS), time_columns=np.array(CAT_STATS))
    # cat_days known answer of the reference's own test (test/test_stats.py:38-43) -- re-run here
    s = pd.Series(data=[1, 2, 1, 1, 2, 3, 1, 4, 3, 2, 1, 1, 2])
    assert [stats.cat_days(s, c) for c in (1, 2, 3, 4)] == [6, 4, 2, 1]
    print("event blocks", ev_off[-1], "time blocks", t_off[-1], "->", OUT)


if __name__ == "__main__":
    main()
