"""Pin the event-statistics oracle to outputs of the reference's own mhw_df()/mhw_features()
(tests/golden/mhw_features_cases.npz, produced by tools/make_golden_features.py)."""
import os

import numpy as np
import numpy.testing as npt

import detect_oracle as det
import features_oracle as fo

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "mhw_features_cases.npz")


def test_event_tables_match_the_reference():
    g = np.load(GOLDEN)
    assert list(g["columns"]) == fo.COLUMNS
    offs, toffs = g["offsets"], g["table_offsets"]
    nev = 0
    for i, (m, jg, gap) in enumerate(g["params"]):
        sl = slice(offs[i], offs[i + 1])
        ts, se, th = g["ts"][sl], g["seas"][sl], g["thresh"][sl]
        with np.errstate(invalid="ignore"):
            b = ts > th
        s, e, ev = det.mhw_filter(b, int(m), bool(jg), int(gap))
        tab = fo.event_table(ts, se, th, s, e, ev)
        want = g["table"][toffs[i]:toffs[i + 1]]
        assert tab.shape == want.shape, f"case {i}"
        for k, col in enumerate(fo.COLUMNS):
            npt.assert_allclose(tab[:, k], want[:, k], rtol=1e-10, atol=1e-12, equal_nan=True,
                                err_msg=f"case {i} column {col}")
        nev += tab.shape[0]
    assert nev == toffs[-1] > 1500


def test_per_step_columns_match_the_reference_mhw_df():
    """mhw_df()'s per-step columns (tests/golden/mhw_df_cases.npz, made by RUNNING the reference)."""
    g = np.load(GOLDEN)
    d = np.load(os.path.join(os.path.dirname(GOLDEN), "mhw_df_cases.npz"))
    offs = g["offsets"]
    cols = list(d["columns"])
    for j, case in enumerate(d["cases"]):
        sl = slice(offs[case], offs[case + 1])
        ts, se, th = g["ts"][sl], g["seas"][sl], g["thresh"][sl]
        m, jg, gap = g["params"][case]
        with np.errstate(invalid="ignore"):
            _, _, ev = det.mhw_filter(ts > th, int(m), bool(jg), int(gap))
        ic = fo.intermediate_columns(ts, se, th, ev)
        want = d["values"][:, d["offsets"][j]:d["offsets"][j + 1]]
        for k, name in enumerate(cols):
            npt.assert_array_equal(np.asarray(ic[name], dtype=np.float64), want[k], err_msg=f"case {case} {name}")
