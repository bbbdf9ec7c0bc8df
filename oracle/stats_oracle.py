"""CPU restatement of block_average()'s aggregation (xmhw/stats.py:285-428).  TEST INFRASTRUCTURE ONLY.

The reference converts one cell's Dataset to a DataFrame and calls pandas
``groupby(pd.cut(years, bins, right=False)).agg(...)`` (call_groupby :285-319; agg_mhw :322-364,
agg_cats :372-401, agg_ts :404-428, cat_days :367-369).  Plain loops over the bins here; pinned to
outputs of those very functions (tests/golden/block_stats_cases.npz, tools/make_golden_stats.py).

pandas semantics restated: every bin is present (observed=False); ``count`` = number of non-NaN
values; ``mean`` / ``max`` / ``min`` / ``sum`` skip NaN; an empty (or all-NaN) bin gives NaN for mean /
max / min, 0 for sum and count; cat_days counts equality with the category number.
"""
import numpy as np

MHW_STATS = ["ecount", "duration", "intensity_max", "intensity_max_max", "intensity_mean", "intensity_cumulative",
             "total_icum", "intensity_mean_relThresh", "intensity_cumulative_relThresh", "severity_mean",
             "severity_cumulative", "intensity_mean_abs", "intensity_cumulative_abs", "rate_onset", "rate_decline"]
# (output name, source column, aggregation) -- stats.py:344-362.  NB the reference takes
# intensity_mean_abs / intensity_cumulative_abs from intensity_mean / intensity_cumulative (:358-359)
MHW_AGG = [("ecount", "event", "count"), ("duration", "duration", "mean"), ("intensity_max", "intensity_max", "mean"),
           ("intensity_max_max", "intensity_max", "max"), ("intensity_mean", "intensity_mean", "mean"),
           ("intensity_cumulative", "intensity_cumulative", "mean"), ("total_icum", "intensity_cumulative", "sum"),
           ("intensity_mean_relThresh", "intensity_mean_relThresh", "mean"),
           ("intensity_cumulative_relThresh", "intensity_cumulative_relThresh", "mean"),
           ("severity_mean", "severity_mean", "mean"), ("severity_cumulative", "severity_cumulative", "mean"),
           ("intensity_mean_abs", "intensity_mean", "mean"), ("intensity_cumulative_abs", "intensity_cumulative", "mean"),
           ("rate_onset", "rate_onset", "mean"), ("rate_decline", "rate_decline", "mean")]
TIME_STATS = ["ts_mean", "ts_max", "ts_min", "moderate_days", "strong_days", "severe_days", "extreme_days"]


def block_bins(first_year, last_year, blockLength):
    """stats.py:130: bin edges; bin b covers years [edges[b], edges[b+1])"""
    return np.arange(first_year, last_year + blockLength + 1, blockLength)


def _agg(v, how):
    v = v[~np.isnan(v)]
    if how == "count":
        return float(v.size)
    if how == "sum":
        return float(np.sum(v)) if v.size else 0.0
    if v.size == 0:
        return np.nan
    return {"mean": np.mean, "max": np.max, "min": np.min}[how](v)


def agg_mhw(table, columns, event_years, edges):
    """one cell: table (n_events, ncol) with ``columns``, the year of each event's mtime -> (nbins, 15)"""
    ci = {c: i for i, c in enumerate(columns)}
    nb = len(edges) - 1
    out = np.full((nb, len(MHW_AGG)), np.nan)
    b = np.searchsorted(edges, event_years, side="right") - 1 if len(event_years) else np.zeros(0, int)
    for k in range(nb):
        sel = b == k
        for j, (_, src, how) in enumerate(MHW_AGG):
            out[k, j] = _agg(table[sel, ci[src]], how)
    return out


def agg_time(ts, cats, years, edges):
    """one cell: ts (T,), cats (T,) or None, calendar year of every step -> (nbins, 3 or 7)"""
    nb = len(edges) - 1
    out = np.full((nb, 3 if cats is None else 7), np.nan)
    b = np.searchsorted(edges, years, side="right") - 1
    for k in range(nb):
        sel = b == k
        out[k, 0], out[k, 1], out[k, 2] = _agg(ts[sel], "mean"), _agg(ts[sel], "max"), _agg(ts[sel], "min")
        if cats is not None:
            for c in (1, 2, 3, 4):
                out[k, 2 + c] = float(np.sum(cats[sel] == c))
    return out
