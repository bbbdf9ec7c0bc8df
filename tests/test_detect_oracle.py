"""Pin the detect-front-end oracle to the reference: outputs of the reference's own
mhw_filter()/join_gaps() (tests/golden/mhw_filter_cases.npz) and the literal expectations
of the reference tests test_mhw_filter / test_join_gaps (test/test_identify.py:88-118,
fixture test/xmhw_fixtures.py:100-162)."""
import os

import numpy as np
import numpy.testing as npt

import detect_oracle as det

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "mhw_filter_cases.npz")


def test_reference_fixture_expectations():
    a = [0, 1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 0, 0, 0, 0]
    b = np.array(a) == 1
    T = len(a)
    st = np.full(T, np.nan); en = np.full(T, np.nan); ev = np.full(T, np.nan)
    st[5], st[16], st[24] = 1, 11, 20
    en[5], en[16], en[24] = 5, 16, 24
    ev[1:6], ev[11:17], ev[20:25] = 1, 11, 20
    s, e, v = det.mhw_filter(b, 5, False)
    npt.assert_array_equal(s, st); npt.assert_array_equal(e, en); npt.assert_array_equal(v, ev)
    s, e, v = det.mhw_filter(b, 5, True, 2)          # gap of 3 > 2: unchanged
    npt.assert_array_equal(s, st); npt.assert_array_equal(e, en); npt.assert_array_equal(v, ev)
    st2, en2, ev2 = st.copy(), en.copy(), ev.copy()
    st2[24] = np.nan; en2[16] = np.nan; ev2[17:25] = 11
    s, e, v = det.mhw_filter(b, 5, True, 3)          # joins the 2nd and 3rd events
    npt.assert_array_equal(s, st2); npt.assert_array_equal(e, en2); npt.assert_array_equal(v, ev2)


def test_against_reference_outputs():
    g = np.load(GOLDEN)
    offs = g["offsets"]
    assert len(offs) - 1 == g["params"].shape[0] >= 400
    for i, (m, jg, gap) in enumerate(g["params"]):
        sl = slice(offs[i], offs[i + 1])
        s, e, v = det.mhw_filter(g["bthresh"][sl], int(m), bool(jg), int(gap))
        msg = f"case {i} m={m} joinGaps={jg} maxGap={gap}"
        npt.assert_array_equal(s, g["start"][sl], err_msg=msg)
        npt.assert_array_equal(e, g["end"][sl], err_msg=msg)
        npt.assert_array_equal(v, g["events"][sl], err_msg=msg)


def test_exceedance_nan_is_false():
    ts = np.array([1.0, np.nan, 3.0, 2.0])
    th = np.array([0.5, np.nan])
    b = det.exceedance(ts, th, [0, 0, 1, 0])
    npt.assert_array_equal(b, [True, False, False, True])
