"""Per-event statistics on the GPU vs the reference (event tables produced by its own
mhw_filter + mhw_df + mhw_features, tests/golden/mhw_features_cases.npz) and vs the oracle."""
import os

import numpy as np
import numpy.testing as npt
import pytest

import detect_oracle as det
import features_oracle as fo

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "mhw_features_cases.npz")


@pytest.fixture(scope="module")
def front():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.detect_front as f
    return f


def test_event_tables_against_reference(front):
    """Each golden case is one cell whose climatology varies per step: a climatology with one row
    per time step (doy label = step index) re-expands to exactly the stored seas/thresh series."""
    g = np.load(GOLDEN)
    assert list(g["columns"]) == front.EVENT_COLUMNS
    offs, toffs = g["offsets"], g["table_offsets"]
    nev = 0
    for i, (m, jg, gap) in enumerate(g["params"]):
        sl = slice(offs[i], offs[i + 1])
        ts, se, th = g["ts"][sl], g["seas"][sl], g["thresh"][sl]
        T = ts.shape[0]
        lab = np.arange(1, T + 1)
        table, offsets = front.mhw_features_cells(ts[:, None], se[:, None], th[:, None], lab, lab,
                                                  int(m), bool(jg), int(gap))
        want = g["table"][toffs[i]:toffs[i + 1]]
        assert table.shape == want.shape and offsets[-1] == want.shape[0], f"case {i}"
        for k, col in enumerate(front.EVENT_COLUMNS):
            npt.assert_allclose(table[:, k], want[:, k], rtol=1e-9, atol=1e-11, equal_nan=True,
                                err_msg=f"case {i} column {col}")
        nev += table.shape[0]
    assert nev == toffs[-1]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gridded_against_oracle(front, dtype):
    import xmhw_oracle as ora
    import oracle_fast as fast
    time = np.arange("2001-01-01", "2006-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    rng = np.random.default_rng(9)
    T, C = time.shape[0], 21
    t = np.arange(T)[:, None]
    anom = np.zeros((T, C))
    e = rng.normal(size=(T, C))
    for k in range(1, T):
        anom[k] = 0.9 * anom[k - 1] + e[k]
    x = (15 + 4 * np.sin(2 * np.pi * (t - rng.uniform(0, 365, C)) / 365.25) + anom).astype(dtype)
    x[rng.random((T, C)) < 0.01] = np.nan
    doys, th, se = fast.threshold_cells_fast(x, doy)
    table, offsets = front.mhw_features_cells(x, se, th, doy, doys, 5, True, 2)
    rows = np.searchsorted(doys, doy)
    assert offsets[-1] > 40
    for c in range(C):
        xc = x[:, c].astype(np.float64)
        b, s, en, ev = det.detect_front(xc, th[:, c], rows, 5, True, 2)
        want = fo.event_table(xc, se[rows, c], th[rows, c], s, en, ev)
        got = table[offsets[c]:offsets[c + 1]]
        assert got.shape == want.shape
        npt.assert_allclose(got, want, rtol=1e-9, atol=1e-11, equal_nan=True, err_msg=f"cell {c}")
