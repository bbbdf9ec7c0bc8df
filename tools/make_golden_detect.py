#!/usr/bin/env python3
"""Golden vectors for the detect() front end, produced by RUNNING the reference's own
functions in the build container:

    python tools/make_golden_detect.py        # reads /root/reference, writes tests/golden/mhw_filter_cases.npz

xmhw/identify.py imports xarray and dask at module level (neither is installed here), but
mhw_filter() (identify.py:415-479), join_gaps() (:273-325) and join_events() (:532-536) are pure
pandas/numpy.  The two missing modules are replaced by inert placeholders for the duration of the
import only (dask.delayed becomes the identity decorator); none of the three functions touches
them.  Only DATA (inputs, outputs) is stored -- no reference source text.
Cases: the reference's own fixture (test/xmhw_fixtures.py:100-162) plus seeded random boolean
series chosen to hit the edge cases (run at the series start/end, chains of joins, no events).
"""
import os
import sys
import types

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "mhw_filter_cases.npz")


def load_reference_identify():
    fake_xr = types.ModuleType("xarray")
    fake_dask = types.ModuleType("dask")

    def delayed(*a, **k):
        if a and callable(a[0]):
            return a[0]
        return lambda f: f
    fake_dask.delayed = delayed
    saved = {k: sys.modules.get(k) for k in ("xarray", "dask")}
    sys.modules["xarray"], sys.modules["dask"] = fake_xr, fake_dask
    sys.path.insert(0, REF)
    try:
        import xmhw.identify as ident
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ident


def main():
    ident = load_reference_identify()
    rng = np.random.default_rng(20260102)
    fixture = [0, 1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 0, 0, 0, 0]
    series = [np.array(fixture, dtype=bool)]
    for i in range(70):
        T = int(rng.integers(1, 120))
        p = rng.choice([0.2, 0.5, 0.7, 0.9])
        # persistent (Markov) series so that long runs and short gaps are common
        b = np.zeros(T, bool)
        state = rng.random() < p
        for t in range(T):
            if rng.random() < 0.25:
                state = rng.random() < p
            b[t] = state
        if i % 7 == 0:
            b[: int(rng.integers(1, 8))] = True          # run at the series start
        if i % 5 == 0:
            b[-int(rng.integers(1, 8)):] = True          # run reaching the series end
        if i % 11 == 0:
            b[:] = False
        if i % 13 == 0:
            b[:] = True
        series.append(b)
    params = [(5, False, 2), (5, True, 2), (5, True, 3), (3, True, 1), (1, True, 0), (2, True, 5), (4, False, 0)]
    bs, offs, par, starts, ends, events = [], [0], [], [], [], []
    for b in series:
        T = b.shape[0]
        for (m, jg, g) in params:
            time = pd.date_range("2001-01-01", periods=T)
            bthresh = pd.Series(b, index=time)
            idxarr = pd.Series(data=np.arange(T), index=time)
            df = ident.mhw_filter(bthresh, idxarr, m, jg, g)
            bs.append(b)
            offs.append(offs[-1] + T)
            par.append((m, int(jg), g))
            starts.append(df["start"].to_numpy(dtype=np.float64))
            ends.append(df["end"].to_numpy(dtype=np.float64))
            events.append(df["events"].to_numpy(dtype=np.float64))
    np.savez_compressed(
        OUT, bthresh=np.concatenate(bs), offsets=np.array(offs, dtype=np.int64),
        params=np.array(par, dtype=np.int64), start=np.concatenate(starts), end=np.concatenate(ends),
        events=np.concatenate(events), fixture_len=np.array(len(fixture)))
    print("cases", len(par), "samples", offs[-1], "->", OUT, "pandas", pd.__version__)


if __name__ == "__main__":
    main()
