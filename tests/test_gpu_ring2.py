"""Second-generation ring kernel (kernels_ring2.hip): every variant against the round-1 ring kernel,
the generic kernel and the oracle.  Raw thresh must be bit-identical (the selection is exact and the
interpolation is the same float64 code); seas is a float64 sum of the same float32 samples in a
different order (one running sum per lane instead of one per track): equal to 1e-13 relative.
"""
import numpy as np
import numpy.testing as npt
import pytest

import xmhw_oracle as ora
import oracle_fast as fast

pytestmark = pytest.mark.gpu
VARIANTS = [8, 10, 12]        # (the layouts the library is built with; the others are refused since round 5)


@pytest.fixture(scope="module")
def dev():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.device as d
    return d


def _series(T, C, seed, nanfrac=0.0, quant=None):
    rng = np.random.default_rng(seed)
    t = np.arange(T)[:, None]
    x = 15 + rng.uniform(2, 10, C) * np.sin(2 * np.pi * (t - rng.uniform(0, 365, C)) / 365.25) \
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


def _raw(dev, x, doy, q=0.9, negate=False, nchunks=0, kernel="ring", ring2=-1):
    h = dev.hip()
    T, C = x.shape
    plan = dev.Plan(doy, 5, kernel=kernel, nchunks=nchunks, ring2=ring2)
    bufs = []
    try:
        d_ts = dev.DeviceBuffer.from_array(x); bufs.append(d_ts)
        th, se = dev.DeviceBuffer(8 * plan.D * C), dev.DeviceBuffer(8 * plan.D * C)
        bufs += [th, se]
        h.plan_debug_stats(plan.handle, 1, False)
        dev.clim_raw(plan, d_ts, 4, C, q, negate, th, se)
        h.stream_sync(0)
        st = h.plan_debug_stats(plan.handle, 1, True)
        return th.to_array((plan.D, C), np.float64), se.to_array((plan.D, C), np.float64), st
    finally:
        for b in bufs:
            b.free()
        plan.destroy()


def _compare(dev, x, doy, q=0.9, negate=False, nchunks=1, expect_fast=None):
    t0, s0, _ = _raw(dev, x, doy, q, negate, nchunks, ring2=-1)
    tg, sg, _ = _raw(dev, x, doy, q, negate, kernel="generic")
    npt.assert_array_equal(t0, tg)
    for v in VARIANTS:
        probe = dev.Plan(doy, 5, ring2=v)
        built = probe.ring2_in_use() == v
        probe.destroy()
        if not built:       # a round-2 experiment (code rings, other extraction widths): -DXMHW_RING2_EXPERIMENTS builds only
            continue
        t1, s1, st = _raw(dev, x, doy, q, negate, nchunks, ring2=v)
        npt.assert_array_equal(np.isnan(t1), np.isnan(t0), err_msg=f"variant {v}")
        npt.assert_array_equal(t1, t0, err_msg=f"variant {v}")
        npt.assert_allclose(s1, s0, rtol=1e-13, atol=0, equal_nan=True, err_msg=f"variant {v}")
        if not dev.hip().debug_stats_available():      # the product build has no counter twins (make STATS=1 builds them)
            continue
        assert st[0] > 0, "the ring2 kernel did not run"
        if expect_fast is not None:
            frac = st[4] / max(1, st[0])
            assert (frac > 0.4) == expect_fast, (v, st)
        if v in (1, 2, 3, 4):         # the variants that carry a code ring
            assert st[5] > 0 or np.isnan(x).any() or expect_fast is False, (v, st)
    return t0, s0


@pytest.mark.parametrize("years,C", [((1982, 2021), 77), ((1991, 2020), 64), ((2001, 2020), 33)])
def test_daily_clean(dev, years, C):
    """40 / 30 / 20 tracks: 5, 4 (2 padded) and 3 (4 padded) tracks per lane; fast steps dominate."""
    doy = _daily(*years)
    x = _series(doy.shape[0], C, 5 + years[0])
    t0, s0 = _compare(dev, x, doy, expect_fast=True)
    _, th, se = fast.raw_clim(x.astype(np.float64), doy, 0.9, 5)
    npt.assert_array_equal(t0, th)
    npt.assert_allclose(s0, se, rtol=1e-13)


@pytest.mark.parametrize("q", [0.1, 0.5, 0.9, 0.0, 1.0])
def test_percentiles(dev, q):
    doy = _daily(1982, 2021)
    x = _series(doy.shape[0], 40, 3)
    _compare(dev, x, doy, q=q)


def test_nan_holes_and_all_nan_cell(dev):
    doy = _daily(1982, 2021)
    x = _series(doy.shape[0], 50, 7, nanfrac=0.05)
    x[:, 3] = np.nan
    x[100:4000, 5] = np.nan            # a long gap: empty pools for nobody, short pools for many
    _compare(dev, x, doy, expect_fast=False)


def test_ties_quantised(dev):
    doy = _daily(1982, 2021)
    x = _series(doy.shape[0], 48, 9, quant=0.01)
    x[:, 0] = 7.0                       # a constant cell: every key equal
    x[:, 1] = np.round(x[:, 1])         # very few distinct values
    _compare(dev, x, doy)


def test_cold_spells_and_extremes(dev):
    doy = _daily(1982, 2021)
    x = _series(doy.shape[0], 32, 11)
    x[5000, 2] = np.inf
    x[5100, 2] = -np.inf
    x[200, 4] = np.inf
    x[::7, 6] = 0.0
    x[1::7, 6] = -0.0
    x[::3, 8] = 1e-42                   # subnormals
    x[:, 9] *= 1e30
    _compare(dev, x, doy, negate=True)
    _compare(dev, x, doy, negate=False)


def test_partial_years_and_chunks(dev):
    """record starting in spring and ending in autumn (first and last track are partial), chunked
    vs unchunked, ragged cell count"""
    doy = _daily(0, 0, "1982-04-17", "2021-10-03")
    x = _series(doy.shape[0], 45, 13)
    a, sa = _compare(dev, x, doy, nchunks=1)
    b, sb = _compare(dev, x, doy, nchunks=5)
    c, sc = _compare(dev, x, doy, nchunks=0)
    npt.assert_array_equal(a, c)
    npt.assert_array_equal(a, b)
    npt.assert_allclose(sa, sb, rtol=1e-13)


def test_tstep_axis(dev):
    """1460 steps per year x 20 years (config 5's axis): 20 tracks, no Feb-29 row"""
    doy = np.tile(np.arange(1, 1461, dtype=np.int64), 20)
    x = _series(doy.shape[0], 24, 17)
    _compare(dev, x, doy, expect_fast=True)


def test_no_leap_year_in_period(dev):
    """3 tracks would not use ring2 (padding beyond the last slot) -> falls back to the round-1 kernel"""
    doy = _daily(2001, 2003)
    x = _series(doy.shape[0], 16, 19)
    t0, s0, st0 = _raw(dev, x, doy, ring2=-1)
    t1, s1, st1 = _raw(dev, x, doy, ring2=8)
    npt.assert_array_equal(t0, t1)
    npt.assert_array_equal(s0, s1)


from tools.fuzz_ring2 import check_ring2_case, random_ring2_case      # generator + checker shared with the long fuzz


@pytest.mark.parametrize("seed", [11, 12])
def test_ring2_random_cases_equal_generic_kernel(dev, seed):
    """random calendars (17..40 tracks), NaN / inf / tie hazards, percentiles incl. 0 and 100, cold spells,
    chunking: both lane layouts bit-identical to the generic kernel on the raw percentile"""
    rng = np.random.default_rng(seed)
    layouts = set()
    for i in range(20):
        x, doy, pct, tstep, cold, nchunks = random_ring2_case(rng)
        layouts |= check_ring2_case(dev, x, doy, pct, tstep, cold, nchunks,
                                    msg=f"seed {seed} case {i}: T={x.shape[0]} C={x.shape[1]} pct={pct} tstep={tstep} cold={cold}")
    assert layouts >= {8, 10, 20, 21}       # 20 .. 22: the third-generation kernel (tests/test_gpu_ring3.py)
