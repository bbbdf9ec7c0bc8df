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


@pytest.mark.parametrize("per_step", [False, True])
def test_event_tables_against_reference(front, per_step):
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
        r = front.detect_cells(ts[:, None], se[:, None], th[:, None], lab, lab, int(m), bool(jg), int(gap),
                               per_step_kernels=per_step)
        table, offsets = r["table"], r["offsets"]
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


@pytest.mark.parametrize("T", [5, 63, 64, 65, 128, 1000])
@pytest.mark.parametrize("params", [(5, True, 2), (2, True, 1), (3, False, 0), (1, True, 0), (70, True, 3), (64, True, 5)])
def test_table_only_path_equals_per_step_path(front, T, params):
    """The bit-packed / thread-per-event pipeline and the per-step kernels run the same per-event
    arithmetic in the same order: identical tables, for series whose runs touch both ends, span
    word boundaries, or cover everything."""
    m, jg, gap = params
    rng = np.random.default_rng(T * 31 + m)
    C = 70                                       # more than one wave, not a multiple of 64
    t = np.arange(T)[:, None]
    x = rng.normal(size=(T, C)).cumsum(axis=0) * 0.3 + rng.normal(size=(T, C))
    x[:, 0] = 10.0                               # always above
    x[:, 1] = -10.0                              # never above
    x[: min(T, 7), 2] = 10.0                     # run at the series start
    x[-min(T, 7):, 3] = 10.0                     # run reaching the series end
    x[rng.random((T, C)) < 0.03] = np.nan
    x[:, 4] = np.nan
    if T >= 200:                                 # runs longer than the opening filter's 64-step reach
        x[20:20 + 75, 5] = 10.0; x[100:100 + 64, 5] = 10.0; x[100 + 64 + 2:100 + 64 + 2 + 90, 5] = 10.0
        x[T - 130:, 6] = 10.0
    doy = (np.arange(T) % 366) + 1
    doys = np.arange(1, min(T, 366) + 1)
    se = rng.normal(size=(doys.shape[0], C)) * 0.1
    th = se + 0.8 + 0.1 * rng.random((doys.shape[0], C))
    from xmhw_amd._lib import hip
    h = hip()
    try:
        for dtype in (np.float32, np.float64):
            for cold in (False, True):
                b = front.detect_cells(x.astype(dtype), se, th, doy, doys, m, jg, gap, coldSpells=cold,
                                       per_step_kernels=True)
                for mode in (1, 2):                  # exceedance bits: per-step kernel, tiled kernel
                    h.set_exceed_kernel(mode)
                    a = front.detect_cells(x.astype(dtype), se, th, doy, doys, m, jg, gap, coldSpells=cold)
                    npt.assert_array_equal(a["offsets"], b["offsets"])
                    npt.assert_array_equal(a["table"], b["table"])
        # and against the loop oracles, for the cells with the long runs
        if T >= 200:
            rows = np.searchsorted(doys, doy)
            for c in (5, 6, 9):
                xc = x[:, c]
                _, s0, e0, ev = det.detect_front(xc, th[:, c], rows, m, jg, gap)
                want = fo.event_table(xc, se[rows, c], th[rows, c], s0, e0, ev)
                r = front.detect_cells(x, se, th, doy, doys, m, jg, gap)
                got = r["table"][r["offsets"][c]:r["offsets"][c + 1]]
                assert got.shape == want.shape, (c, m)
                npt.assert_allclose(got, want, rtol=1e-9, atol=1e-11, equal_nan=True)
    finally:
        h.set_exceed_kernel(0)


def test_exceed_bits_float32_floor_is_exact(front):
    """xmhw_exceed_bits_f32 compares against the float32 floor of each threshold: same bits as
    (double)x > th for thresholds equal to, one float64 ulp around, and far from float32 values,
    and for inf / NaN / huge magnitudes on either side."""
    from xmhw_amd._lib import hip
    from xmhw_amd.device import DeviceBuffer
    h = hip()
    rng = np.random.default_rng(77)
    T, C = 200, 96
    x = (rng.normal(size=(T, C)) * 10.0 ** rng.integers(-3, 4, size=(T, C))).astype(np.float32)
    x[5, :] = np.inf; x[6, :] = -np.inf; x[7, :] = np.nan; x[8, :] = np.float32(3.4e38); x[9, :] = 0.0; x[10, :] = -0.0
    xd = x.astype(np.float64)
    th = np.empty((T, C))
    kind = rng.integers(0, 8, size=(T, C))
    th[:] = rng.normal(size=(T, C))
    th = np.where(kind == 0, xd, th)                                   # equal: not an exceedance
    th = np.where(kind == 1, np.nextafter(xd, -np.inf), th)            # one f64 ulp below x
    th = np.where(kind == 2, np.nextafter(xd, np.inf), th)             # one f64 ulp above x
    th = np.where(kind == 3, np.nextafter(x, np.float32(-np.inf)).astype(np.float64), th)   # previous f32
    th = np.where(kind == 4, xd * (1 + 1e-9), th)
    th[11, :] = np.inf; th[12, :] = -np.inf; th[13, :] = np.nan; th[14, :] = 1e300; th[15, :] = -1e300
    th[16, :] = 5e-324; th[17, :] = -5e-324
    rows = np.arange(T, dtype=np.int32)                                # one threshold row per step
    with np.errstate(invalid="ignore"):
        want = xd > th
    for neg, mode in ((0, 1), (1, 1), (0, 2), (1, 2)):
        with np.errstate(invalid="ignore"):
            w = (-xd > th) if neg else (xd > th)
        d_x, d_th = DeviceBuffer.from_array(x), DeviceBuffer.from_array(th)
        W = (T + 63) // 64
        d_b = DeviceBuffer(8 * W * C)
        try:
            h.set_exceed_kernel(mode)
            h.exceed_bits(d_x.ptr, 4, T, C, C, d_th.ptr, C, T, rows, neg, d_b.ptr, C)
            words = d_b.to_array((W, C), np.uint64)
        finally:
            h.set_exceed_kernel(0)
            for b in (d_x, d_th, d_b):
                b.free()
        got = np.zeros((T, C), dtype=bool)
        for t in range(T):
            got[t] = (words[t // 64] >> np.uint64(t % 64)) & np.uint64(1)
        npt.assert_array_equal(got, w)
        # bits beyond T are zero
        assert not np.any(words[-1] >> np.uint64(T % 64)) if T % 64 else True


def test_tiled_exceedance_on_calendar_and_irregular_labels(front):
    """exceed_bits_tiled with real calendar labels (leap days split the chunks), with a label
    sequence that jumps around (every step its own chunk), and with a leading dimension != C."""
    import xmhw_oracle as ora
    from xmhw_amd._lib import hip
    from xmhw_amd.device import DeviceBuffer
    h = hip()
    rng = np.random.default_rng(5)
    time = np.arange("1999-11-20", "2005-03-07", dtype="datetime64[D]")
    cal = ora.add_doy(time) - 1
    T, C, ld = time.shape[0], 130, 137
    for rows, D in ((cal, 366), (rng.integers(0, 50, size=T), 50), ((np.arange(T) * 7) % 366, 366)):
        for dtype in (np.float32, np.float64):
            x = np.full((T, ld), np.nan, dtype=dtype)
            x[:, :C] = rng.normal(size=(T, C))
            th = rng.normal(size=(D, ld)) * 0.5 + 0.5
            want = x[:, :C].astype(np.float64) > th[rows][:, :C]
            d_x, d_th = DeviceBuffer.from_array(x), DeviceBuffer.from_array(th)
            W = (T + 63) // 64
            got = {}
            try:
                for mode in (1, 2):
                    d_b = DeviceBuffer.from_array(np.full((W, ld), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))
                    h.set_exceed_kernel(mode)
                    h.exceed_bits(d_x.ptr, x.dtype.itemsize, T, C, ld, d_th.ptr, ld, D, rows.astype(np.int32), 0,
                                  d_b.ptr, ld)
                    got[mode] = d_b.to_array((W, ld), np.uint64)
                    d_b.free()
            finally:
                h.set_exceed_kernel(0)
                d_x.free(); d_th.free()
            for mode, words in got.items():
                bits = np.zeros((T, C), dtype=bool)
                for t in range(T):
                    bits[t] = (words[t // 64, :C] >> np.uint64(t % 64)) & np.uint64(1)
                npt.assert_array_equal(bits, want, err_msg=f"mode {mode}")
                # columns beyond C are left alone
                assert (words[:, C:] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
