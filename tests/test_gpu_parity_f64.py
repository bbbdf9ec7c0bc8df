"""Parity of the float64 paths (the second-generation ring kernel's 64-bit mode where it is instantiated: w = 5;
the generic kernel elsewhere) against the oracle and against each other, on the same families of inputs as the
float32 tests.  The round-1 float64 ring kernel is gone (round 3): an explicit ring request on a plan the 64-bit
mode does not cover is refused, see test_explicit_ring_request_on_an_uncovered_float64_plan_is_refused."""
import numpy as np
import numpy.testing as npt
import pytest

import xmhw_oracle as ora
import oracle_fast as fast
from test_gpu_parity import _series, _daily

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.device as d
    return d


def _check(dev, x, doy, nchunks=0, rtol=1e-12, **kw):
    assert x.dtype == np.float64
    args = (kw.get("pctile", 90), kw.get("windowHalfWidth", 5), kw.get("smoothPercentile", True),
            kw.get("smoothPercentileWidth", 31), kw.get("tstep", False), kw.get("coldSpells", False))
    # narrowing off: the float64 ring kernel itself (quantised test data is float32-representable
    # and would otherwise take the float32 kernel, which test_float64_narrowing covers)
    kernel = "ring" if kw.get("windowHalfWidth", 5) == 5 else "auto"      # (other windows: the generic kernel)
    d1, t1, s1 = dev.calc_clim_device(x, doy, *args, kernel=kernel, nchunks=nchunks, narrowing=False)
    d0, t0, s0 = fast.threshold_cells_fast(x, doy, **kw)
    npt.assert_array_equal(d1, d0)
    npt.assert_array_equal(np.isnan(t1), np.isnan(t0))
    npt.assert_allclose(t1, t0, rtol=rtol, atol=0, equal_nan=True)
    npt.assert_allclose(s1, s0, rtol=rtol, atol=0, equal_nan=True)
    return t1, t0


@pytest.mark.parametrize("nanfrac", [0.0, 0.05])
def test_daily_random_f64(dev, nanfrac):
    time, doy = _daily(2001, 2012)
    x = _series(time.shape[0], 101, 11, nanfrac, dtype=np.float64)
    t1, t0 = _check(dev, x, doy, smoothPercentile=False)
    if nanfrac == 0.0:
        m = np.ones(366, bool); m[59] = False
        npt.assert_array_equal(t1[m], t0[m])          # exact selection + numpy's lerp
    _check(dev, x, doy)


def test_forty_tracks_f64(dev):
    time, doy = _daily(1982, 2021)                     # 40 tracks -> 3 tracks per lane
    x = _series(time.shape[0], 37, 2, 0.01, dtype=np.float64)
    _check(dev, x, doy, smoothPercentile=False)


def test_ties_and_absent_groups_f64(dev):
    time, doy = _daily(1995, 2004)
    x = _series(time.shape[0], 40, 5, 0.02, dtype=np.float64, quant=0.25)
    x[(doy >= 150) & (doy <= 230), 3] = np.nan
    x[:, 9] = np.nan
    x[:, 2] = 1.5
    _check(dev, x, doy, smoothPercentile=False)
    _check(dev, x, doy, pctile=50, windowHalfWidth=2, smoothPercentileWidth=5)


def test_tstep_cold_f64(dev):
    n, ny = 73, 9
    doy = np.tile(np.arange(1, n + 1), ny)
    x = _series(n * ny, 33, 3, 0.03, dtype=np.float64)
    _check(dev, x, doy, tstep=True, windowHalfWidth=2, smoothPercentileWidth=5, coldSpells=True, pctile=10)


@pytest.mark.parametrize("nchunks", [1, 3, 7])
def test_chunks_and_generic_identical_f64(dev, nchunks):
    time, doy = _daily(2003, 2014)
    x = _series(time.shape[0], 30, 21, 0.04, dtype=np.float64)
    d, t1, s1 = dev.calc_clim_device(x, doy, 90, 5, False, 31, False, kernel="ring", nchunks=nchunks)
    d, t0, s0 = dev.calc_clim_device(x, doy, 90, 5, False, 31, False, kernel="generic")
    npt.assert_array_equal(t1, t0)
    npt.assert_allclose(s1, s0, rtol=1e-13, equal_nan=True)


def test_partial_years_and_extremes_f64(dev):
    time = np.arange("2001-07-15", "2009-03-10", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    x = _series(time.shape[0], 21, 9, 0.01, dtype=np.float64)
    _check(dev, x, doy, smoothPercentile=False)
    time, doy = _daily(2001, 2004)
    x = _series(time.shape[0], 8, 17, dtype=np.float64)
    x[5, 0] = np.inf; x[100, 0] = -np.inf; x[7, 1] = 0.0; x[8, 1] = -0.0
    x[:, 3] = 5e-320                                   # subnormal
    x[:, 4] = -np.inf                                  # a cell that is -inf throughout
    d1, t1, s1 = dev.calc_clim_device(x, doy, 90, 5, False, 31, True, kernel="ring")
    d0, t0, s0 = fast.threshold_cells_fast(x, doy, smoothPercentile=False, tstep=True)
    fin = np.isfinite(t0)
    npt.assert_allclose(t1[fin], t0[fin], rtol=1e-12)
    npt.assert_array_equal(t1[:, 4], t0[:, 4])
    npt.assert_array_equal(np.isfinite(t1), fin)
    npt.assert_allclose(s1, s0, rtol=1e-12, equal_nan=True)     # means of pools holding +-inf, and the rows after
    # low percentile with -inf present: the inclusive extraction bound must admit -inf
    d1, t1, s1 = dev.calc_clim_device(x, doy, 0, 5, False, 31, True, kernel="ring")
    d0, t0, s0 = fast.threshold_cells_fast(x, doy, pctile=0, smoothPercentile=False, tstep=True)
    npt.assert_array_equal(t1[:, [0, 4]], t0[:, [0, 4]])


def test_float64_narrowing(dev):
    """float64 input holding float32-representable samples runs on the float32 ring kernel:
    bit-identical to the float32 call; genuinely float64 data, and data with a single lossy sample
    hidden from the probe, end on the float64 kernel with its exact result."""
    from xmhw_amd.device import DeviceBuffer, Plan, clim_raw
    from xmhw_amd._lib import hip
    h = hip()
    time, doy = _daily(1990, 2012)
    T, C = time.shape[0], 77
    x32 = _series(T, C, 21, 0.02, dtype=np.float32)
    x32[:, 5] = np.nan
    x32[100:140, 6] = np.inf
    x64 = x32.astype(np.float64)
    hidden = x64.copy()
    hidden[1237, 41] = 12.345678901234567          # one sample that float32 cannot hold, off the probe rows
    assert 1237 % max(T // 32, 1) != 0
    rand = _series(T, C, 22, 0.02, dtype=np.float64)
    D = 366

    def run(arr, narrowing, q=0.9, neg=False):
        plan = Plan(doy, 5, kernel="ring", narrowing=narrowing)
        d_ts = DeviceBuffer.from_array(np.ascontiguousarray(arr))
        th, se = DeviceBuffer(8 * D * C), DeviceBuffer(8 * D * C)
        try:
            clim_raw(plan, d_ts, arr.dtype.itemsize, C, q, neg, th, se)
            h.stream_sync(0)
            narrowed = plan.narrowed() if arr.dtype == np.float64 else None
            return th.to_array((D, C), np.float64), se.to_array((D, C), np.float64), narrowed
        finally:
            for b in (d_ts, th, se):
                b.free()
            plan.destroy()

    for neg in (False, True):
        t32, s32, _ = run(x32, True, neg=neg)
        tn, sn, narrowed = run(x64, True, neg=neg)
        assert narrowed
        npt.assert_array_equal(tn, t32)
        npt.assert_array_equal(sn, s32)
        t64, s64, narrowed = run(x64, False, neg=neg)      # the float64 kernel on the same values
        assert not narrowed
        npt.assert_array_equal(tn, t64)                     # exact selection + the same lerp
        npt.assert_allclose(sn, s64, rtol=1e-13, equal_nan=True)
    for arr in (hidden, rand):
        ta, sa, narrowed = run(arr, True)
        assert not narrowed
        tb, sb, _ = run(arr, False)
        npt.assert_array_equal(ta, tb)
        npt.assert_array_equal(sa, sb)
    # and through the host API, against the oracle
    d1, t1, s1 = dev.calc_clim_device(x64, doy, 90, 5, True, 31, False)
    d0, t0, s0 = fast.threshold_cells_fast(x64, doy)
    npt.assert_allclose(t1, t0, rtol=1e-12, equal_nan=True)
    npt.assert_allclose(s1, s0, rtol=1e-12, equal_nan=True)


def test_long_float64_record_narrows_onto_the_16_lane_ring(dev):
    """61 years as float64: float32-representable samples take the float32 kernel (16 lanes per cell),
    genuine float64 samples the 64-bit mode on the same layout; both against the generic kernel."""
    from xmhw_amd.device import DeviceBuffer, Plan, clim_raw
    from xmhw_amd._lib import hip
    h = hip()
    time, doy = _daily(1960, 2020)
    T, C, D = time.shape[0], 9, 366
    x32 = _series(T, C, 41, 0.01, dtype=np.float32)
    rand = _series(T, C, 42, 0.01, dtype=np.float64)

    def run(arr, kernel="auto"):
        plan = Plan(doy, 5, kernel=kernel)
        d_ts = DeviceBuffer.from_array(np.ascontiguousarray(arr))
        th, se = DeviceBuffer(8 * D * C), DeviceBuffer(8 * D * C)
        try:
            clim_raw(plan, d_ts, arr.dtype.itemsize, C, 0.9, False, th, se)
            h.stream_sync(0)
            return th.to_array((D, C), np.float64), se.to_array((D, C), np.float64), plan.narrowed()
        finally:
            for b in (d_ts, th, se):
                b.free()
            plan.destroy()

    t32, s32, _ = run(x32)
    tn, sn, narrowed = run(x32.astype(np.float64))
    assert narrowed
    npt.assert_array_equal(tn, t32)
    npt.assert_array_equal(sn, s32)
    tr, sr, narrowed = run(rand)
    assert not narrowed
    tg, sg, _ = run(rand, kernel="generic")
    npt.assert_array_equal(tr, tg)
    # (since round 2 such a record runs on the second-generation kernel's 64-bit mode, 16 lanes per cell: the
    # percentile is exact, the mean a running float64 sum -- equal to the generic kernel's up to rounding)
    npt.assert_allclose(sr, sg, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("seed", [61, 62])
def test_float64_random_cases_equal_generic_kernel(dev, seed):
    """the float64 path as the library takes it (probe -> ring2 narrowing -> ring2 64-bit mode / round-1 float64
    ring) on random plans with full-precision doubles, exact repeats and distinct doubles sharing the high word
    of their key: raw percentile bit-identical to the generic float64 kernel"""
    from tools.fuzz_ring2 import check_f64_case, random_f64_case
    rng = np.random.default_rng(seed)
    for i in range(16):
        x, doy, pct, tstep, cold, nchunks = random_f64_case(rng)
        check_f64_case(dev, x, doy, pct, tstep, cold, nchunks, msg=f"seed {seed} case {i}")


def _clustered(T, C, seed):
    """doubles that defeat anything working on a prefix of the key: a few levels per cell, each sample a distinct
    double within 1e-9 of its level, exact repeats mixed in, and both zeros"""
    rng = np.random.default_rng(seed)
    lev = np.round(rng.normal(size=(T, C)) * 2.0) / 2.0 + 10.0
    x = lev * (1.0 + rng.integers(0, 9, size=(T, C)) * 1e-9)
    x[rng.random((T, C)) < 0.2] = 10.5                    # exact repeats
    x[:, 1] = np.where(rng.random(T) < 0.5, 0.0, -0.0)    # a cell of zeros of both signs
    x[:, 2] = rng.integers(-2, 3, size=T) * 1e-9          # clustered around zero, both signs, with repeats
    x[rng.random((T, C)) < 0.01] = np.nan
    return x


@pytest.mark.parametrize("years", [(1982, 2021), (2001, 2012), (1960, 2020)])
@pytest.mark.parametrize("w", [2, 5])
def test_clustered_doubles_through_every_dispatchable_float64_kernel(dev, years, w):
    """every float64 kernel the C ABI can dispatch for the plan -- auto and explicit ring (w = 5: the 64-bit mode on
    8 lanes with the low words in registers or LDS, or on 16 lanes), generic, each with and without the narrowing
    launches in front -- against the oracle, raw percentile bit for bit (verdict round 2: the withdrawn
    float64 ring failed exactly this kind of input)"""
    from xmhw_amd.device import DeviceBuffer, Plan, clim_raw
    from xmhw_amd._lib import hip
    h = hip()
    time, doy = _daily(*years)
    T, C = time.shape[0], 24
    x = _clustered(T, C, 7 + w + years[0])
    _, t0, s0 = fast.raw_clim(x, doy, 0.9, w)
    D = t0.shape[0]
    ran = []
    for kernel in ("auto", "ring", "generic"):
        for narrowing in (True, False):
            plan = Plan(doy, w, kernel=kernel, narrowing=narrowing)
            d_ts = DeviceBuffer.from_array(np.ascontiguousarray(x))
            th, se = DeviceBuffer(8 * D * C), DeviceBuffer(8 * D * C)
            try:
                if kernel == "ring" and plan.f64_mode() < 0:
                    with pytest.raises(Exception):
                        clim_raw(plan, d_ts, 8, C, 0.9, False, th, se)
                    continue
                clim_raw(plan, d_ts, 8, C, 0.9, False, th, se)
                h.stream_sync(0)
                t1, s1 = th.to_array((D, C), np.float64), se.to_array((D, C), np.float64)
                ran.append((kernel, narrowing, plan.f64_mode()))
            finally:
                for b in (d_ts, th, se):
                    b.free()
                plan.destroy()
            npt.assert_array_equal(t1, t0, err_msg=f"{kernel} narrowing={narrowing} w={w}")
            npt.assert_allclose(s1, s0, rtol=1e-12, atol=1e-12, equal_nan=True)
    assert any(k == "generic" for k, _, _ in ran)
    if w == 5:
        assert any(k == "ring" and m >= 0 for k, _, m in ran), ran


def test_explicit_ring_request_on_an_uncovered_float64_plan_is_refused(dev):
    """w = 2 has no float64 ring kernel any more: XMHW_KERNEL_RING must fail loudly (XMHW_ERR_UNSUPPORTED), auto
    takes the generic kernel"""
    from xmhw_amd.device import DeviceBuffer, Plan, clim_raw
    time, doy = _daily(2001, 2012)
    x = _series(time.shape[0], 8, 3, dtype=np.float64) + 1e-9
    plan = Plan(doy, 2, kernel="ring", narrowing=False)
    d_ts = DeviceBuffer.from_array(x)
    th, se = DeviceBuffer(8 * 366 * 8), DeviceBuffer(8 * 366 * 8)
    try:
        assert plan.f64_mode() < 0
        with pytest.raises(Exception, match="(?i)unsupported|not available"):
            clim_raw(plan, d_ts, 8, 8, 0.9, False, th, se)
    finally:
        for b in (d_ts, th, se):
            b.free()
        plan.destroy()
