#!/usr/bin/env python3
"""Golden vectors for the per-event statistics, produced by RUNNING the
reference's own mhw_filter() (xmhw/identify.py:415-479), mhw_df() and mhw_features()
(xmhw/features.py:22-70, :72-315) in the build container on seeded synthetic cells:

    python tools/make_golden_features.py      # writes tests/golden/mhw_features_cases.npz

identify.py's unused xarray/dask imports are replaced by inert placeholders for the import only
(as in make_golden_detect.py); features.py needs nothing but numpy/pandas.  The per-step inputs
(ts, seas, thresh re-expanded along time) and the per-event table are stored; time stamps are
stored as positions along the time axis.  Only DATA is stored, no reference source text.
"""
import os
import sys
import warnings

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden_detect import load_reference_identify, REF

DF_COLS = ["seas", "thresh", "relSeas", "relThresh", "relThreshNorm", "severity", "cats", "duration_moderate",
           "duration_strong", "duration_severe", "duration_extreme", "mabs"]
OUT_DF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "mhw_df_cases.npz")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "mhw_features_cases.npz")
COLS = ["event", "index_start", "index_end", "time_start", "time_end", "time_peak", "intensity_max",
        "intensity_mean", "intensity_cumulative", "severity_max", "severity_mean", "severity_cumulative",
        "severity_var", "intensity_mean_relThresh", "intensity_cumulative_relThresh", "intensity_mean_abs",
        "intensity_cumulative_abs", "duration_moderate", "duration_strong", "duration_severe",
        "duration_extreme", "index_peak", "intensity_var", "intensity_max_relThresh", "intensity_max_abs",
        "intensity_var_relThresh", "intensity_var_abs", "category", "duration", "rate_onset", "rate_decline"]


def main():
    ident = load_reference_identify()
    sys.path.insert(0, REF)
    import xmhw.features as feat
    sys.path.remove(REF)
    warnings.simplefilter("ignore")
    rng = np.random.default_rng(20260103)
    ts_all, se_all, th_all, offs, params, tables, toffs = [], [], [], [0], [], [], [0]
    df_cases, df_cols = [], []      # per-step mhw_df() columns of a few cases -> mhw_df_cases.npz
    for i in range(36):
        T = int(rng.integers(200, 1500))
        t = np.arange(T)
        seas = 15 + 4 * np.sin(2 * np.pi * (t - rng.uniform(0, 365)) / 365.25)
        thresh = seas + rng.uniform(0.8, 1.6) + 0.2 * np.sin(2 * np.pi * t / 90.0)
        anom = np.zeros(T)
        e = rng.normal(size=T) * rng.uniform(0.4, 1.0)
        for k in range(1, T):
            anom[k] = 0.92 * anom[k - 1] + e[k]
        ts = seas + anom + rng.choice([0.0, 0.6])
        if i % 3 == 0:
            ts[rng.random(T) < 0.02] = np.nan                  # NaN holes (also inside joined gaps)
        if i % 5 == 0:
            ts[: int(rng.integers(3, 12))] = thresh[:12].max() + 2   # event at the series start
        if i % 7 == 0:
            ts[-int(rng.integers(3, 12)):] = thresh[-12:].max() + 2  # event reaching the series end
        if i % 11 == 0:
            ts = ts.astype(np.float32).astype(np.float64)
        for (m, jg, g) in ((5, True, 2), (3, True, 4), (5, False, 2)):
            time = pd.date_range("2001-01-01", periods=T)
            idxarr = pd.Series(data=np.arange(T), index=time)
            df = pd.DataFrame({"ts": ts, "seas": seas, "thresh": thresh}, index=time)
            df["bthresh"] = df.ts > df.thresh
            dfev = ident.mhw_filter(df.bthresh, idxarr, m, jg, g)
            df = feat.mhw_df(pd.concat([df, dfev], axis=1))
            if i % 6 == 0 and (m, jg, g) == (5, True, 2):
                df_cases.append(len(params))
                df_cols.append(np.stack([df[c].to_numpy(dtype=np.float64) for c in DF_COLS]))
            if df.events.notna().sum() == 0:
                tab = np.zeros((0, len(COLS)))
            else:
                out = feat.mhw_features(df, T - 1, "time", [])
                for c in ("time_start", "time_end", "time_peak"):
                    out[c] = time.get_indexer(pd.DatetimeIndex(out[c]))
                tab = out[COLS].to_numpy(dtype=np.float64)
            ts_all.append(ts); se_all.append(seas); th_all.append(thresh)
            offs.append(offs[-1] + T)
            params.append((m, int(jg), g))
            tables.append(tab)
            toffs.append(toffs[-1] + tab.shape[0])
    np.savez_compressed(OUT, ts=np.concatenate(ts_all), seas=np.concatenate(se_all), thresh=np.concatenate(th_all),
                        offsets=np.array(offs), params=np.array(params), table=np.concatenate(tables, axis=0),
                        table_offsets=np.array(toffs), columns=np.array(COLS))
    np.savez_compressed(OUT_DF, cases=np.array(df_cases), columns=np.array(DF_COLS),
                        values=np.concatenate(df_cols, axis=1),
                        offsets=np.cumsum([0] + [c.shape[1] for c in df_cols]))
    print("mhw_df cases", df_cases, "->", OUT_DF)
    print("cases", len(params), "events", toffs[-1], "samples", offs[-1], "->", OUT)


if __name__ == "__main__":
    main()
