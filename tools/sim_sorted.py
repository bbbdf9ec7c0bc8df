"""Offline model of the sorted-list selection (round 5, kernels_sorted.hip) -- numpy only, no GPU.

A row's pool is the union of the last R = 11 ROW-LISTS (the samples that all tracks push at one step); a
row-list is evicted whole.  The kernel keeps, per list, its K largest keys sorted in LDS and a staircase
c_i = number of keys of list i that belong to the top set (the `rt` largest keys of the pool, rt = n - 1 - lo,
so that order statistic lo + 1 is the smallest key of the top set and lo the largest key outside it).  Per
row one list is replaced and the staircase is walked |delta| single steps, delta = c_new - c_evicted (+ the
change of rt).  This script measures, on the SURVEY 8(d) generator and on variants (quantised values,
sea-ice plateaus, AR(1) anomalies):
  * walk steps per cell-row (mean / p90 / p99) and per wave-row (max over the 16 / 32 / 64 cells of a wave),
  * the list capacity needed: max over lists of c_i, and how often c_i >= K for K = 12..20 per cell-row and per
    wave-row (those rows refill the list from global memory).
Usage: python tools/sim_sorted.py [cells] [kind]      kind = gauss | quant | ice | ar1 | nan
"""
import sys
import numpy as np


def synth(T, C, seed, kind):
    rng = np.random.default_rng(seed)
    t = np.arange(T)[:, None]
    A = rng.uniform(2, 10, C)
    phi = rng.uniform(0, 365, C)
    beta = rng.uniform(-1, 1, C)
    if kind == "ar1":
        eps = np.empty((T, C))
        e = rng.normal(size=C)
        rho = 0.9
        s = np.sqrt(1 - rho * rho)
        for i in range(T):
            e = rho * e + s * rng.normal(size=C)
            eps[i] = e
    else:
        eps = rng.normal(size=(T, C))
    x = 15 + A * np.sin(2 * np.pi * (t - phi) / 365.25) + 0.0005 * t * beta + eps
    if kind in ("quant", "ice"):
        x = np.round(x * 100.0) / 100.0
    if kind == "ice":
        # 10 % of the cells sit at -1.8 for 120 days of every year
        ice = rng.uniform(size=C) < 0.10
        doy = np.arange(T) % 365
        hold = (doy >= 200) & (doy < 320)
        x[np.ix_(hold, np.where(ice)[0])] = -1.8
    x = x.astype(np.float32)
    if kind == "nan":
        x[rng.uniform(size=x.shape) < 0.05] = np.nan
    return x


def f32_key(x):
    b = x.view(np.uint32).astype(np.int64)
    k = np.where(b & 0x80000000, (~b) & 0xFFFFFFFF, b | 0x80000000)
    return np.where(np.isnan(x), 0, k)


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    kind = sys.argv[2] if len(sys.argv) > 2 else "gauss"
    Y, DY, w, q = 40, 365, 5, 0.9
    R = 2 * w + 1
    T = Y * DY
    keys = f32_key(synth(T, C, 1, kind)).reshape(Y, DY, C)          # no-leap calendar: list(d) = keys[:, d, :]
    # lists in push order; row r pools lists r .. r+R-1 (wrapping at the end of the year is good enough here)
    idx = np.arange(DY + R) % DY
    lists = np.sort(keys[:, idx, :], axis=0)[::-1]                  # [Y][DY+R][C] descending
    steps = np.zeros((DY, C), dtype=np.int64)
    cmax = np.zeros((DY, C), dtype=np.int64)
    cnew0 = np.zeros((DY, C), dtype=np.int64)
    prev_c = None
    prev_vlo = None
    for r in range(DY):
        pool = lists[:, r:r + R, :]                                 # [Y][R][C]
        flat = np.sort(pool.reshape(Y * R, C), axis=0)              # ascending; invalid keys (0) first
        n = (flat > 0).sum(axis=0)
        nn = np.maximum(n, 1)
        lo = np.floor((nn - 1) * q).astype(np.int64)
        # order statistic lo of the valid keys
        vlo = flat[(Y * R - n) + lo, np.arange(C)]
        c = (pool > vlo[None, None, :]).sum(axis=0)                 # [R][C] staircase (ties: keys equal to vlo are outside)
        cmax[r] = c.max(axis=0)
        if prev_c is not None:
            new = pool[:, R - 1, :]
            c_new0 = (new > prev_vlo[None, :]).sum(axis=0)
            cnew0[r] = c_new0
            before = prev_c[1:].sum(axis=0) + c_new0
            steps[r] = np.abs(before - c.sum(axis=0))
        prev_c = c
        prev_vlo = vlo
    st = steps[1:]
    print(f"kind {kind}: {C} cells x {DY - 1} rows, {Y} tracks, pool {Y * R}, pctile {q}")
    print(f"walk steps per cell-row: mean {st.mean():.2f}  p50 {np.percentile(st, 50):.0f}  p90 {np.percentile(st, 90):.0f}"
          f"  p99 {np.percentile(st, 99):.0f}  max {st.max()}")
    for cw in (16, 32, 64):
        m = st[:, :C // cw * cw].reshape(DY - 1, -1, cw).max(axis=2)
        print(f"  per wave-row of {cw} cells (max over the wave): mean {m.mean():.2f}  p90 {np.percentile(m, 90):.0f}  max {m.max()}")
    need = np.maximum(cmax, cnew0)
    print(f"list share of the top set: mean of max-over-lists {cmax.mean():.2f}, p99 {np.percentile(cmax, 99):.0f}, max {cmax.max()};"
          f" new list against the old pivot: max {cnew0.max()}")
    for K in (10, 12, 13, 14, 16, 18, 20):
        over = need >= K
        line = f"  K = {K:2d}: c_i >= K on {100.0 * over.mean():.4f} % of cell-rows;"
        for cw in (16, 32, 64):
            ow = over[:, :C // cw * cw].reshape(DY, -1, cw).any(axis=2)
            line += f"  {100.0 * ow.mean():.3f} % of wave-rows ({cw})"
        print(line)


if __name__ == "__main__":
    main()
