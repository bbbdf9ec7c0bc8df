"""CPU oracle for the per-event statistics.  TEST INFRASTRUCTURE ONLY.

Loop restatement, for ONE cell, of mhw_df() (xmhw/features.py:22-70) and mhw_features()
(features.py:72-97: agg_df :100-160, properties :163-196, onset_decline :266-295 with get_period
:224-263, get_edge :204-221, get_rate :199-201).  PINNED against outputs of the reference's own
functions (tests/golden/mhw_features_cases.npz, tools/make_golden_features.py).

Conventions of the reference kept on purpose:
* statistics run over every step LABELLED with the event, gap steps of joined events included;
  NaN samples are skipped (pandas skipna);
* severity = relSeas / -(thresh - seas)  (features.py:58-60: the sign is negative);
* variances are sample variances (ddof=1) and are returned as standard deviations
  (np.sqrt in properties());
* get_period() compares the RELATIVE peak index with the last index of the series
  (features.py:259) and adds 0.5 unless the event touches the series boundary.
Time stamps are returned as positions along the time axis.
"""
import numpy as np

COLUMNS = ["event", "index_start", "index_end", "time_start", "time_end", "time_peak", "intensity_max",
           "intensity_mean", "intensity_cumulative", "severity_max", "severity_mean", "severity_cumulative",
           "severity_var", "intensity_mean_relThresh", "intensity_cumulative_relThresh", "intensity_mean_abs",
           "intensity_cumulative_abs", "duration_moderate", "duration_strong", "duration_severe",
           "duration_extreme", "index_peak", "intensity_var", "intensity_max_relThresh", "intensity_max_abs",
           "intensity_var_relThresh", "intensity_var_abs", "category", "duration", "rate_onset", "rate_decline"]


def _first(v):
    ok = ~np.isnan(v)
    return v[ok][0] if ok.any() else np.nan


def _last(v):
    ok = ~np.isnan(v)
    return v[ok][-1] if ok.any() else np.nan


def _mean(v):
    ok = ~np.isnan(v)
    return v[ok].sum() / ok.sum() if ok.any() else np.nan


def _sum(v):
    return v[~np.isnan(v)].sum()


def _max(v):
    ok = ~np.isnan(v)
    return v[ok].max() if ok.any() else np.nan


def _var(v):
    x = v[~np.isnan(v)]
    if x.size < 2:
        return np.nan
    m = x.sum() / x.size
    return ((x - m) ** 2).sum() / (x.size - 1)


def get_rate(relSeas_peak, relSeas_edge, period):
    """features.py:199-201"""
    with np.errstate(divide="ignore", invalid="ignore"):
        return (relSeas_peak - relSeas_edge) / np.float64(period)


def get_edge(relS_edge, anom_edge, idx, edge):
    """features.py:204-221: the event's own edge value if it touches the series boundary `edge`,
    else the mean of it and the anomaly one step outside the event."""
    return relS_edge if idx == edge else 0.5 * (relS_edge + anom_edge)


def get_period(start, end, peak, tsend):
    """features.py:224-263.  `peak` is index_peak - index_start; note that it (not the absolute
    peak index) is what the reference compares with tsend."""
    esp = end - start - peak
    x = peak if peak != 0 else 1
    onset_period = x if start == 0 else x + 0.5
    y = esp if peak != tsend else 1
    decline_period = y if end == tsend else y + 0.5
    return onset_period, decline_period


def intermediate_columns(ts, seas, thresh, events):
    """mhw_df() (features.py:36-69): the per-step columns, for one cell.  ts/seas/thresh (T,) with
    the climatologies already re-expanded along time, events: mhw_filter() labels (NaN = none)."""
    ts, seas, thresh = (np.asarray(a, dtype=np.float64) for a in (ts, seas, thresh))
    ismhw = ~np.isnan(events)
    mt = np.where(ismhw, ts, np.nan)
    ms = np.where(ismhw, seas, np.nan)
    mth = np.where(ismhw, thresh, np.nan)
    relS = mt - ms
    relT = mt - mth
    th_se = mth - ms
    with np.errstate(divide="ignore", invalid="ignore"):
        relTN = relT / th_se
        sev = relS / -(th_se)
    cats = np.floor(1.0 + relTN)
    return dict(seas=ms, thresh=mth, relSeas=relS, relThresh=relT, relThreshNorm=relTN, severity=sev, cats=cats,
                duration_moderate=cats == 1.0, duration_strong=cats == 2.0, duration_severe=cats == 3.0,
                duration_extreme=cats >= 4.0, mabs=mt)


def event_table(ts, seas, thresh, start, end, events):
    """Per-event statistics of one cell.  ts/seas/thresh: (T,) with seas/thresh already
    re-expanded along time; start/end/events: mhw_filter() output.  Returns (n_events, 31)."""
    ts, seas, thresh = (np.asarray(a, dtype=np.float64) for a in (ts, seas, thresh))
    T = ts.shape[0]
    last = T - 1
    ismhw = ~np.isnan(events)
    anom = ts - seas
    anom_plus = np.concatenate(([np.nan], anom[:-1]))
    anom_minus = np.concatenate((anom[1:], [np.nan]))
    ic = intermediate_columns(ts, seas, thresh, events)
    mt, relS, relT, sev, cats = ic["mabs"], ic["relSeas"], ic["relThresh"], ic["severity"], ic["cats"]
    rows = []
    for L in np.unique(events[ismhw]):
        idx = np.nonzero(events == L)[0]
        r = dict(event=L, index_start=_first(start[idx]), index_end=_first(end[idx]),
                 time_start=idx[0], time_end=idx[-1])
        rs = relS[idx]
        ok = ~np.isnan(rs)
        if ok.any():
            imax = int(np.nonzero(ok)[0][np.argmax(rs[ok])])       # first maximum, NaN skipped
        else:
            imax = -1
        r["time_peak"] = idx[imax] if imax >= 0 else np.nan
        r["intensity_max"], r["intensity_mean"], r["intensity_cumulative"] = _max(rs), _mean(rs), _sum(rs)
        sv = sev[idx]
        r["severity_max"], r["severity_mean"], r["severity_cumulative"] = _max(sv), _mean(sv), _sum(sv)
        r["severity_var"] = np.sqrt(_var(sv))
        rt = relT[idx]
        r["intensity_mean_relThresh"], r["intensity_cumulative_relThresh"] = _mean(rt), _sum(rt)
        ma = mt[idx]
        r["intensity_mean_abs"], r["intensity_cumulative_abs"] = _mean(ma), _sum(ma)
        c = cats[idx]
        r["duration_moderate"] = float(np.sum(c == 1.0))
        r["duration_strong"] = float(np.sum(c == 2.0))
        r["duration_severe"] = float(np.sum(c == 3.0))
        r["duration_extreme"] = float(np.sum(c >= 4.0))
        r["index_peak"] = L + imax
        r["intensity_var"] = np.sqrt(_var(rs))
        r["intensity_max_relThresh"] = relT[idx[imax]] if imax >= 0 else np.nan
        r["intensity_max_abs"] = mt[idx[imax]] if imax >= 0 else np.nan
        r["intensity_var_relThresh"] = np.sqrt(_var(rt))
        r["intensity_var_abs"] = np.sqrt(_var(ma))
        r["category"] = np.minimum(_max(c), 4)
        r["duration"] = r["index_end"] - r["index_start"] + 1
        # onset / decline rates (features.py:224-295)
        onset_period, decline_period = get_period(r["index_start"], r["index_end"],
                                                  r["index_peak"] - r["index_start"], last)
        edge0 = get_edge(_first(rs), _first(anom_plus[idx]), r["index_start"], 0)
        edge1 = get_edge(_last(rs), _last(anom_minus[idx]), r["index_end"], last)
        r["rate_onset"] = get_rate(r["intensity_max"], edge0, onset_period)
        r["rate_decline"] = get_rate(r["intensity_max"], edge1, decline_period)
        rows.append([r[k] for k in COLUMNS])
    return np.array(rows, dtype=np.float64).reshape(-1, len(COLUMNS))
