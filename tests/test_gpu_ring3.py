"""Third-generation ring kernel (kernels_ring3.hip, ring2 variants 20 / 21 / 22: 8 / 4 / 2 lanes per cell): per-cell histogram in LDS, band
compaction, bitonic sort across the lanes of a cell.  Raw thresh must be bit-identical to the generic kernel (an
independent algorithm) and to the oracle; seas a float64 sum of the same float32 samples in another order.
The debug counters show that the band path -- not the round-2 slow path kept inside the kernel -- settles the
rows, and that the histogram never disagreed with the ring (slot 3, high word).
"""
import numpy as np
import numpy.testing as npt
import pytest

import xmhw_oracle as ora
import oracle_fast as fast

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.device as d
    return d


def _series(T, C, seed, nanfrac=0.0, quant=None, base=15.0):
    rng = np.random.default_rng(seed)
    t = np.arange(T)[:, None]
    x = base + rng.uniform(2, 10, C) * np.sin(2 * np.pi * (t - rng.uniform(0, 365, C)) / 365.25) \
        + 0.0005 * t * rng.uniform(-1, 1, C) + rng.normal(size=(T, C))
    if quant:
        x = np.round(x / quant) * quant
    x = x.astype(np.float32)
    if nanfrac:
        x[rng.random((T, C)) < nanfrac] = np.nan
    return x


def _daily(y0, y1, start=None, stop=None):
    time = np.arange(start or f"{y0}-01-01", stop or f"{y1 + 1}-01-01", dtype="datetime64[D]")
    return ora.add_doy(time)


def _raw(dev, x, doy, q=0.9, negate=False, nchunks=0, kernel="ring", ring2=None):
    h = dev.hip()
    T, C = x.shape
    plan = dev.Plan(doy, 5, kernel=kernel, nchunks=nchunks, ring2=ring2)
    bufs = []
    try:
        use = plan.ring2_in_use()
        d_ts = dev.DeviceBuffer.from_array(x); bufs.append(d_ts)
        th, se = dev.DeviceBuffer(8 * plan.D * C), dev.DeviceBuffer(8 * plan.D * C)
        bufs += [th, se]
        h.plan_debug_stats(plan.handle, 1, False)
        dev.clim_raw(plan, d_ts, 4, C, q, negate, th, se)
        h.stream_sync(0)
        st = h.plan_debug_stats(plan.handle, 1, True)
        return th.to_array((plan.D, C), np.float64), se.to_array((plan.D, C), np.float64), st, use
    finally:
        for b in bufs:
            b.free()
        plan.destroy()


def _band_share(st):
    """wave-rows settled by the band path alone / wave-rows"""
    return (int(st[5]) & 0xFFFFFFFF) / max(int(st[0]), 1)


def _check(dev, x, doy, q=0.9, negate=False, nchunks=1, min_band=None):
    tg, sg, _, _ = _raw(dev, x, doy, q, negate, kernel="generic")
    out = {}
    for v in (20, 21, 22):
        t1, s1, st, use = _raw(dev, x, doy, q, negate, nchunks, ring2=v)
        if use != v:            # the layout is not instantiated for this track count
            continue
        npt.assert_array_equal(t1, tg, err_msg=f"variant {v}")
        npt.assert_allclose(s1, sg, rtol=1e-12, atol=1e-300, equal_nan=True, err_msg=f"variant {v}")
        if dev.hip().debug_stats_available():       # counter twins: make STATS=1 (the product build has none)
            assert st[0] > 0, "the ring3 kernel did not run"
            assert int(st[3]) >> 32 == 0, f"variant {v}: the histogram disagreed with the ring {st}"
            if min_band is not None:
                assert _band_share(st) >= min_band, (v, _band_share(st), st)
        out[v] = (t1, s1, st)
    assert out, "no ring3 layout ran"
    return tg, sg, out


def test_default_is_ring3_on_a_60_year_axis(dev):
    """60-year daily axis: the library picks the 8-lane third-generation kernel on its own (records of 9..48 tracks run
    on the sorted-list kernel since round 5: test_gpu_sorted.py)"""
    doy = _daily(1960, 2019)
    plan = dev.Plan(doy, 5)
    try:
        assert plan.ring2_in_use() == 20
    finally:
        plan.destroy()


@pytest.mark.parametrize("years,C", [((1982, 2021), 77), ((1991, 2020), 64), ((2001, 2020), 33), ((1982, 2024), 40),
                                     ((2010, 2020), 19)])
def test_daily_clean_equals_generic_and_oracle(dev, years, C):
    """40 / 30 / 20 / 43 / 11 tracks; the band path settles most rows of clean SST-like data"""
    doy = _daily(*years)
    x = _series(doy.shape[0], C, 5 + years[0])
    tg, sg, _ = _check(dev, x, doy, min_band=0.6)
    _, th, se = fast.raw_clim(x.astype(np.float64), doy, 0.9, 5)
    npt.assert_array_equal(tg, th)
    npt.assert_allclose(sg, se, rtol=1e-13)


@pytest.mark.parametrize("q", [0.1, 0.5, 0.9, 0.99, 0.0, 1.0])
def test_percentiles(dev, q):
    doy = _daily(1982, 2021)
    x = _series(doy.shape[0], 40, 3)
    _check(dev, x, doy, q=q)


def test_nan_holes_and_all_nan_cell(dev):
    doy = _daily(1982, 2021)
    x = _series(doy.shape[0], 50, 7, nanfrac=0.05)
    x[:, 3] = np.nan
    x[100:4000, 5] = np.nan
    _check(dev, x, doy, min_band=0.3)


def test_ties_quantised_and_constant_cells(dev):
    """0.01 degree data (what OISST stores), a constant cell (every key in one bucket: the band never fits a list,
    the slow path settles its wave), a cell with very few distinct values"""
    doy = _daily(1982, 2021)
    x = _series(doy.shape[0], 48, 9, quant=0.01)
    x[:, 0] = 7.0
    x[:, 1] = np.round(x[:, 1])
    x[:, 17] = -1.8                      # sea ice: constant at the freezing point
    _check(dev, x, doy)
    # quantised data alone keeps to the band path
    xq = _series(doy.shape[0], 32, 10, quant=0.01)
    _check(dev, xq, doy, min_band=0.5)


def test_values_straddling_zero_and_wide_ranges(dev):
    """polar cells (-1.8 .. 4 degrees: keys on both sides of zero, where the key density per degree explodes),
    a cell that spans many binades, infinities, cold spells"""
    doy = _daily(1982, 2021)
    x = _series(doy.shape[0], 40, 21, base=1.0)
    x[:, 5] = np.float32(1e-3) * x[:, 5]
    x[:, 6] = np.float32(1e6) * x[:, 6]
    x[5000, 2] = np.inf
    x[6000, 3] = -np.inf
    x[::7, 8] = 0.0
    x[1::7, 8] = -0.0
    x[::3, 9] = 1e-42
    _check(dev, x, doy)
    _check(dev, x, doy, negate=True)


def test_partial_years_chunks_and_ragged_cells(dev):
    """record starting in spring and ending in autumn (hold steps, ring rotation), chunked vs unchunked"""
    doy = _daily(0, 0, "1982-04-17", "2021-10-03")
    x = _series(doy.shape[0], 45, 13)
    a = _check(dev, x, doy, nchunks=1)[2]
    b = _check(dev, x, doy, nchunks=5)[2]
    for v in a:
        npt.assert_array_equal(a[v][0], b[v][0])


def test_tstep_axis(dev):
    """1460 steps per year x 20 years (config 5's axis): 20 tracks (5 per lane at 4 lanes), no Feb-29 row"""
    doy = np.tile(np.arange(1, 1461, dtype=np.int64), 20)
    x = _series(doy.shape[0], 24, 17)
    _check(dev, x, doy, min_band=0.6)


def test_every_track_count_9_to_48(dev):
    """one short run per instantiated layout: 3..12 tracks per lane at 4 lanes, 2..6 at 8 lanes"""
    for ny in range(9, 49, 3):
        doy = _daily(1975, 1975 + ny - 1)
        x = _series(doy.shape[0], 20, 100 + ny)
        _check(dev, x, doy)


def test_float64_input_holding_float32_values_narrows_onto_the_ring3_kernel(dev):
    """float64 input whose samples are float32-representable (a float32 archive promoted by a reader) runs on the
    narrowing instantiation of the third-generation kernel: bit-identical to the float32 call on the same values, for
    heat waves and cold spells, with NaN and infinite samples; one lossy sample hidden from the sparse probe makes
    the kernel give up and the float64 kernel behind it does the work (same result as with narrowing off)."""
    from xmhw_amd.device import DeviceBuffer, Plan, clim_raw
    h = dev.hip()
    doy = _daily(1982, 2021)
    T, C = doy.shape[0], 83
    x32 = _series(T, C, 31, 0.01)
    x32[:, 7] = np.nan
    x32[200:260, 9] = np.inf
    x64 = x32.astype(np.float64)
    hidden = x64.copy()
    hidden[9001, 40] = 17.123456789012345            # float32 cannot hold it; off the probe's rows
    assert 9001 % max(T // 32, 1) != 0
    D = 366

    def run(arr, narrowing, neg=False, nchunks=0):
        plan = Plan(doy, 5, kernel="ring", narrowing=narrowing, nchunks=nchunks, layout=21)     # (the ring3 kernel: the default
        assert plan.ring2_in_use() == 21                                                        # float32 layout of this shape is 40)
        d_ts = DeviceBuffer.from_array(np.ascontiguousarray(arr))
        th, se = DeviceBuffer(8 * D * C), DeviceBuffer(8 * D * C)
        try:
            clim_raw(plan, d_ts, arr.dtype.itemsize, C, 0.9, neg, th, se)
            h.stream_sync(0)
            narrowed = plan.narrowed() if arr.dtype == np.float64 else None
            return th.to_array((D, C), np.float64), se.to_array((D, C), np.float64), narrowed
        finally:
            for b in (d_ts, th, se):
                b.free()
            plan.destroy()

    for neg in (False, True):
        for nchunks in (0, 3):
            t32, s32, _ = run(x32, True, neg, nchunks)
            tn, sn, narrowed = run(x64, True, neg, nchunks)
            assert narrowed
            npt.assert_array_equal(tn, t32)
            npt.assert_array_equal(sn, s32)
    t64, s64, narrowed = run(x64, False)               # the float64 kernel on the same values
    assert not narrowed
    tn, sn, _ = run(x64, True)
    npt.assert_array_equal(tn, t64)
    npt.assert_allclose(sn, s64, rtol=1e-13, equal_nan=True)
    ta, sa, narrowed = run(hidden, True)
    assert not narrowed
    tb, sb, _ = run(hidden, False)
    npt.assert_array_equal(ta, tb)
    npt.assert_array_equal(sa, sb)
