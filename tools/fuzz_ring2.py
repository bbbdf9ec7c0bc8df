"""Long randomised cross-check of the ring2 kernel (both lane layouts) against the GENERIC KERNEL (an
independent algorithm on the same device), on random plans inside ring2's instantiations.  The generator
and the checker are shared with tests/test_gpu_ring2.py::test_ring2_random_cases_equal_generic_kernel.

    python tools/fuzz_ring2.py [--cases 400] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np
import numpy.testing as npt

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from xmhw_amd.calendar import add_doy      # noqa: E402


def _raw(dev, x, doy, q, negate, nchunks=0, kernel="ring", ring2=-1, narrowing=True):
    """raw (unfinished) thresh / seas of one kernel + its debug pass counters"""
    h = dev.hip()
    T, C = x.shape
    plan = dev.Plan(doy, 5, kernel=kernel, nchunks=nchunks, ring2=ring2, narrowing=narrowing)
    bufs = []
    try:
        d_ts = dev.DeviceBuffer.from_array(x); bufs.append(d_ts)
        th, se = dev.DeviceBuffer(8 * plan.D * C), dev.DeviceBuffer(8 * plan.D * C)
        bufs += [th, se]
        h.plan_debug_stats(plan.handle, 1, False)
        dev.clim_raw(plan, d_ts, x.dtype.itemsize, C, q, negate, th, se)
        h.stream_sync(0)
        st = h.plan_debug_stats(plan.handle, 1, True)
        return th.to_array((plan.D, C), np.float64), se.to_array((plan.D, C), np.float64), st
    finally:
        for b in bufs:
            b.free()
        plan.destroy()


def random_ring2_case(rng, years=(9, 49)):
    """a random plan inside ring2's instantiations (w = 5, 9..48 tracks) with random data hazards"""
    kind = rng.choice(["daily", "daily_partial", "tstep", "tstep_short"])
    ny = int(rng.integers(years[0], years[1]))
    if kind in ("tstep", "tstep_short"):
        n = int(rng.integers(12, 90)) if kind == "tstep" else int(rng.integers(12, 20))
        doy = np.tile(np.arange(1, n + 1), ny)
        tstep = True
    else:
        y0 = int(rng.integers(1950, 1985))
        if kind == "daily":
            time = np.arange(f"{y0}-01-01", f"{y0 + ny}-01-01", dtype="datetime64[D]")
        else:
            a = np.datetime64(f"{y0}-01-01") + int(rng.integers(1, 360))
            time = np.arange(a, a + int(365.25 * (ny - 1)) - int(rng.integers(0, 300)), dtype="datetime64[D]")
        doy = add_doy(time)
        tstep = False
    T = doy.shape[0]
    C = int(rng.choice([1, 7, 8, 9, 15, 16, 17, 31, 33, 64, 65, 130]))
    t = np.arange(T)[:, None]
    x = rng.uniform(-2, 25) + rng.uniform(0, 10, C) * np.sin(2 * np.pi * (t - rng.uniform(0, 365, C)) / 365.25) \
        + rng.normal(size=(T, C)) * rng.uniform(0.01, 3)
    quant = rng.choice([0, 0, 0.01, 0.5, 2.0])
    if quant:
        x = np.round(x / quant) * quant
    x = x.astype(np.float32)
    nanfrac = rng.choice([0.0, 0.0, 0.0, 0.02, 0.3, 0.95])
    if nanfrac:
        x[rng.random((T, C)) < nanfrac] = np.nan
    hazard = rng.integers(0, 6)
    if hazard == 0:
        x[:, rng.integers(0, C)] = np.nan                          # an all-NaN cell
    elif hazard == 1:
        x[rng.integers(0, T, 5), rng.integers(0, C, 5)] = np.inf
    elif hazard == 2:
        x[rng.integers(0, T, 5), rng.integers(0, C, 5)] = -np.inf
    elif hazard == 3:
        x[:, rng.integers(0, C)] = 4.25                            # a constant cell: every key ties
    elif hazard == 4:
        # clusters: a few levels, each sample a few float32 ulps off its level (distinct but adjacent keys)
        lev = np.round(x.astype(np.float64) * 2.0) / 2.0
        x = (lev.astype(np.float32).view(np.int32) + rng.integers(-3, 4, size=x.shape).astype(np.int32)).view(np.float32)
        x = np.where(np.isfinite(x), x, np.float32(1.0)).astype(np.float32)
    pct = float(rng.choice([0, 1, 10, 50, 75, 90, 90, 90, 95, 99, 100]))
    return x, doy, pct, tstep, bool(rng.integers(0, 2)), int(rng.integers(0, 4))


def check_ring2_case(dev, x, doy, pct, tstep, cold, nchunks, msg="", sorted_only=False):
    t0, s0, _ = _raw(dev, x, doy, pct / 100.0, cold, kernel="generic")
    if sorted_only:
        fin = np.abs(x[np.isfinite(x)])
        t1, s1, st = _raw(dev, x, doy, pct / 100.0, cold, nchunks, ring2=40)
        with np.errstate(invalid="ignore"):
            npt.assert_array_equal(t1, t0, err_msg=f"{msg} variant 40")
            npt.assert_allclose(s1, s0, rtol=1e-12, atol=1e-13 * float(fin.max()) if fin.size else 0.0, equal_nan=True,
                                err_msg=f"{msg} variant 40")
        return {40}
    # the round-1 float32 ring kernel (what other windows and longer records still run on) rides along
    tr, sr, _ = _raw(dev, x, doy, pct / 100.0, cold, nchunks, ring2=-1)
    with np.errstate(invalid="ignore"):
        npt.assert_array_equal(tr, t0, err_msg=f"{msg} round-1 ring")
        # (float64 sums in another order than the generic kernel's: see below)
        fin = np.abs(x[np.isfinite(x)])
        npt.assert_allclose(sr, s0, rtol=1e-12, atol=1e-13 * float(fin.max()) if fin.size else 0.0, equal_nan=True,
                            err_msg=f"{msg} round-1 ring")
    seen = set()
    for v in (None, 8, 10, 12, 20, 21, 22, 40):
        try:
            plan = dev.Plan(doy, 5, ring2=v)
        except Exception:          # (the sorted-list layout is refused where it is not instantiated: < 9 or > 48 tracks)
            continue
        use = plan.ring2_in_use()
        plan.destroy()
        if use < 0 or use in seen:
            continue
        seen.add(use)
        t1, s1, st = _raw(dev, x, doy, pct / 100.0, cold, nchunks, ring2=use)
        assert st[0] > 0 or not dev.hip().debug_stats_available(), "the ring2 kernel did not run"
        with np.errstate(invalid="ignore"):
            npt.assert_array_equal(t1, t0, err_msg=f"{msg} variant {use}")
            # (the absolute term: float64 sums in another order -- a pool of +0.5, -0.5 and a few denormals has the mean
            # 5e-47 or 0 depending on which addition came first: seed 71 case 9600)
            npt.assert_allclose(s1, s0, rtol=1e-12, atol=1e-13 * float(fin.max()) if fin.size else 0.0, equal_nan=True,
                                err_msg=f"{msg} variant {use}")
    return seen


def random_f64_case(rng, years=(9, 49)):
    """as random_ring2_case with genuinely float64 samples: full-precision doubles, exact repeats (quantised),
    and clusters of DISTINCT doubles that share the high word of their 64-bit key (within 2^-20 relative), which
    is what the low-word pass of the 64-bit mode has to sort out"""
    x32, doy, pct, tstep, cold, nchunks = random_ring2_case(rng, years)
    x = x32.astype(np.float64)
    T, C = x.shape
    mode = rng.integers(0, 4)
    if mode == 0:                                           # full precision
        x = x + rng.normal(size=x.shape) * 1e-9
    elif mode == 1:                                         # near ties: a few levels, jittered in the low word only
        lev = np.round(x * 2.0) / 2.0
        x = lev * (1.0 + rng.integers(0, 7, size=x.shape) * 1e-9)
    elif mode == 2:                                         # mixture: repeats, near ties and free values
        lev = np.round(x)
        pick = rng.integers(0, 3, size=x.shape)
        x = np.where(pick == 0, lev, np.where(pick == 1, lev * (1.0 + rng.integers(0, 3, size=x.shape) * 3e-8), x + 1e-7))
    else:                                                   # float32-representable: the narrowing path (kept for contrast)
        pass
    x[np.isnan(x32)] = np.nan
    x[np.isinf(x32)] = x32[np.isinf(x32)]
    return x, doy, pct, tstep, cold, nchunks


def check_f64_case(dev, x, doy, pct, tstep, cold, nchunks, msg="", kernel="auto"):
    """the float64 path as the library takes it (narrowing probe -> ring2 narrowing -> ring2 64-bit mode or the
    round-1 float64 ring) against the generic float64 kernel: raw percentile bit for bit"""
    t0, s0, _ = _raw(dev, x, doy, pct / 100.0, cold, kernel="generic")
    for narrowing in (True, False):
        t1, s1, _ = _raw(dev, x, doy, pct / 100.0, cold, nchunks, kernel=kernel, ring2=None, narrowing=narrowing)
        with np.errstate(invalid="ignore"):
            npt.assert_array_equal(t1, t0, err_msg=f"{msg} narrowing={narrowing}")
            # (float64 sums are not exact: a running sum and a direct one differ by rounding, which shows
            # relative to a mean that crosses zero -- hence the absolute term)
            npt.assert_allclose(s1, s0, rtol=1e-12, atol=1e-12, equal_nan=True, err_msg=f"{msg} narrowing={narrowing}")


def check_packed_case(dev, x, doy, pct, cold, rng, msg=""):
    """int16 codes read in place (xmhw_clim_raw_i16) against the generic kernel on the series xmhw_decode() makes of the
    same codes: thresh bit for bit, seas within rounding; float32 and float64 decode, either byte order, a random recipe"""
    h = dev.hip()
    T, C = x.shape
    scale = float(rng.choice([0.01, 0.5, 0.001, -0.01, 0.0021973]))
    offset = float(rng.choice([0.0, 10.0, 273.15]))
    fill = int(rng.choice([-32768, -999, 32767]))
    decoded = str(rng.choice(["float32", "float64"]))
    big = bool(rng.integers(0, 2))
    finite = np.isfinite(x)
    with np.errstate(invalid="ignore", over="ignore"):
        c = np.rint((np.where(finite, x, 0.0).astype(np.float64) - offset) / scale)
    codes = np.clip(c, -32767, 32766).astype(np.int16)
    codes[codes == fill] += 1 if fill < 32766 else -1
    codes[~finite] = fill
    if decoded == "float32":
        scale, offset = float(np.float32(scale)), float(np.float32(offset))
    isz = 4 if decoded == "float32" else 8
    stored = np.ascontiguousarray(codes.astype(">i2") if big else codes).view(np.int16)
    d_codes = dev.DeviceBuffer.from_array(stored)
    d_dec = dev.DeviceBuffer(isz * T * C)
    bufs = [d_codes, d_dec]
    try:
        h.decode(d_codes.ptr, 2, int(big), T, C, C, d_dec.ptr, isz, C, True, scale, offset, True, float(fill), 0)
        h.stream_sync(0)
        out = {}
        for which in ("packed", "generic"):
            plan = dev.Plan(doy, 5, kernel="generic" if which == "generic" else "auto")
            th, se = dev.DeviceBuffer(8 * plan.D * C), dev.DeviceBuffer(8 * plan.D * C)
            try:
                if which == "packed":
                    dev.clim_raw_packed(plan, d_codes, C, pct / 100.0, cold, th, se, scale_factor=scale, add_offset=offset,
                                        fill=fill, decoded=decoded, big_endian=big)
                else:
                    dev.clim_raw(plan, d_dec, isz, C, pct / 100.0, cold, th, se)
                h.stream_sync(0)
                out[which] = (th.to_array((plan.D, C), np.float64), se.to_array((plan.D, C), np.float64))
            finally:
                th.free(); se.free(); plan.destroy()
        m = f"{msg} scale={scale} offset={offset} fill={fill} decoded={decoded} big_endian={big}"
        with np.errstate(invalid="ignore"):
            npt.assert_array_equal(out["packed"][0], out["generic"][0], err_msg=m)
            amax = float(np.abs(codes.astype(np.float64) * scale + offset).max())
            npt.assert_allclose(out["packed"][1], out["generic"][1], rtol=1e-12, atol=1e-13 * amax, equal_nan=True, err_msg=m)
    finally:
        for b in bufs:
            b.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64", "i16"])
    ap.add_argument("--kernel", default="auto", help="f64 mode: 'ring' with XMHW_RING2_F64=0 exercises the round-1 float64 ring")
    ap.add_argument("--long", action="store_true", help="records of 49..120 years: the 16- and 32-lane round-1 float32 rings")
    ap.add_argument("--years", type=int, nargs=2, default=None, metavar=("FIRST", "PAST_LAST"),
                    help="record lengths drawn (default 9 49); 37 41 = the sorted-list kernel's two-tier lists (round 6)")
    ap.add_argument("--sorted-only", action="store_true",
                    help="f32: percentiles >= 85 or <= 15 and the sorted-list layout (40) against the generic kernel only -- five times the cases per minute")
    args = ap.parse_args()
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.device as dev
    rng = np.random.default_rng(args.seed)
    t0 = time.perf_counter()
    if args.dtype == "i16":
        from xmhw_amd.exception import XmhwException
        done = refused = 0
        while done < args.cases:
            x, doy, pct, tstep, cold, nchunks = random_ring2_case(rng, (9, 49))
            if 15 < pct < 85:
                continue
            try:
                check_packed_case(dev, x, doy, pct, cold, rng, msg=f"seed {args.seed} case {done}: T={x.shape[0]} C={x.shape[1]} pct={pct} cold={cold}")
            except XmhwException as e:         # (a plan the sorted-list kernel does not serve -- a year of 12 steps, say -- is refused)
                if "sorted-list kernel" not in str(e):
                    raise
                refused += 1
                continue
            done += 1
        print(f"{args.cases} random int16-packed cases (codes read in place; float32 / float64 decode, either byte order): 0 mismatches "
              f"against the generic kernel on the decoded series; {refused} draws refused as plans of another kernel ({time.perf_counter() - t0:.0f} s)")
        return
    if args.dtype == "f64":
        for i in range(args.cases):
            x, doy, pct, tstep, cold, nchunks = random_f64_case(rng, (49, 121) if args.long else (9, 49))
            check_f64_case(dev, x, doy, pct, tstep, cold, nchunks, kernel=args.kernel,
                           msg=f"seed {args.seed} case {i}: T={x.shape[0]} C={x.shape[1]} pct={pct} tstep={tstep} cold={cold}")
        print(f"{args.cases} random float64 cases: 0 mismatches against the generic kernel ({time.perf_counter() - t0:.0f} s)")
        return
    layouts = {0: 0, 7: 0, 8: 0, 10: 0, 20: 0, 21: 0, 31: 0}
    yrs = tuple(args.years) if args.years else ((49, 121) if args.long else (9, 49))
    i = 0
    while i < args.cases:
        x, doy, pct, tstep, cold, nchunks = random_ring2_case(rng, yrs)
        if args.sorted_only:
            if 15 < pct < 85:
                continue
            plan = dev.Plan(doy, 5)
            ok = plan.ring2_in_use() == 40
            plan.destroy()
            if not ok:
                continue
        seen = check_ring2_case(dev, x, doy, pct, tstep, cold, nchunks, sorted_only=args.sorted_only,
                                msg=f"seed {args.seed} case {i}: T={x.shape[0]} C={x.shape[1]} pct={pct} tstep={tstep} cold={cold}")
        i += 1
        for v in seen:
            layouts[v] = layouts.get(v, 0) + 1
    print(f"{args.cases} random cases, runs per layout {layouts}: 0 mismatches against the generic kernel "
          f"({time.perf_counter() - t0:.0f} s)")


if __name__ == "__main__":
    main()
