"""Model of the round-3 bucket design (clim_ring3).  Per cell a histogram window of NB
buckets of 2^shift keys placed around the target (clamped, not circular); per row the BAND = NBAND
buckets starting at the bucket B0 that holds order statistic lo.  The window is rebuilt (re-centred,
shift re-derived from the keys-per-rank estimate) when the target comes within EDGE buckets of an end
or when the running mean of the band population leaves [MLO, MHI].  Printed: band population m, walk
distance per row (buckets, from the previous anchor), how the band keys spread over the lanes of a cell
(list capacity), how often lo+1 falls outside the band, rebuilds per cell-year.  Pure numpy."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sim_select import synth, f32_key
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import xmhw_oracle as ora
from oracle_fast import pool_index


def run(factor, NB=256, NBAND=3, C=64, seed=0, q=0.9, subs=(4, 8), EDGE=40, MLO=None, MHI=None, x=None, years=(1982, 2021)):
    time = np.arange(f"{years[0]}-01-01", f"{years[1] + 1}-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    yr = time.astype("datetime64[Y]").astype(int)
    yr = yr - yr.min()
    if x is None:
        x = synth(time.shape[0], C, seed)
    keys = f32_key(x)
    doys, pools = pool_index(doy, 5)
    D = len(doys)
    MLO = MLO if MLO is not None else 0.45 * factor * NBAND
    MHI = MHI if MHI is not None else 1.8 * factor * NBAND
    m_all = np.zeros((D, C), int); walk = np.zeros((D, C), int); miss = np.zeros((D, C), bool)
    lanemax = {s: np.zeros((D, C), int) for s in subs}
    shift = np.zeros(C, int); base = np.zeros(C, np.int64); prevB = np.zeros(C, np.int64); mavg = np.zeros(C)
    kpr = np.ones(C)
    rebuilds = 0
    for i, idx in enumerate(pools):
        k = keys[idx, :]
        tr = yr[idx]
        order = np.argsort(k, axis=0, kind="stable")
        n = k.shape[0]; lo = int(np.floor((n - 1) * q))
        for c in range(C):
            col = k[order[:, c], c]; trc = tr[order[:, c]]
            need = i == 0
            if not need:
                b = (int(col[lo]) - base[c]) >> shift[c]
                need = b < EDGE or b >= NB - EDGE or mavg[c] > MHI or mavg[c] < MLO
            if need:
                # the kernel's estimate: span of the previous band / its population
                if i == 0:
                    kpr[c] = max(1.0, (col[min(lo + 4, n - 1)] - col[lo - 4]) / 8.0)
                shift[c] = max(0, int(np.floor(np.log2(max(factor * kpr[c], 1.0)))))
                base[c] = max(0, int(col[lo]) - ((NB // 2) << shift[c]))
                prevB[c] = (int(col[lo]) - base[c]) >> shift[c]
                mavg[c] = factor * NBAND
                rebuilds += i > 0
            sh = int(shift[c])
            tags = np.clip((col - base[c]) >> sh, 0, NB - 1)
            B0 = int(tags[lo])
            sel = (tags >= B0) & (tags < B0 + NBAND)
            m = int(sel.sum())
            jj = lo - int(np.searchsorted(tags, B0, side="left"))
            miss[i, c] = jj + 1 >= m and lo + 1 < n
            m_all[i, c] = m
            mavg[c] = 0.75 * mavg[c] + 0.25 * m
            kpr[c] = 0.75 * kpr[c] + 0.25 * max(1.0, (NBAND << sh) / max(m, 1))
            walk[i, c] = B0 - (int(prevB[c]) & ~3)          # from the previous anchor (aligned down to 4)
            prevB[c] = B0
            for s in subs:
                lanemax[s][i, c] = np.bincount(trc[sel] % s, minlength=s).max()
    return m_all[1:], walk[1:], miss[1:], {s: v[1:] for s, v in lanemax.items()}, rebuilds


def report(tag, m, walk, miss, lm, reb, C):
    up = walk[walk >= 0]; dn = -walk[walk < 0]
    print(f"{tag}: m mean {m.mean():.2f} p99 {np.percentile(m, 99):.0f} max {m.max()}; miss(lo+1 outside) {miss.mean():.5f}; "
          f"walk up P(>=16) {np.mean(walk >= 16):.4f} P(>=32) {np.mean(walk >= 32):.5f}; down P(>16) {np.mean(walk < -16):.4f} "
          f"P(>32) {np.mean(walk < -32):.5f}; rebuilds/cell-yr {reb / C:.2f}")
    for s, v in lm.items():
        w = v.reshape(v.shape[0], -1, 64 // s).max(axis=2)
        print(f"   {s} lanes/cell: per cell P(>4) {np.mean(v > 4):.4f} P(>6) {np.mean(v > 6):.4f} P(>8) {np.mean(v > 8):.5f}; "
              f"per wave P(>4) {np.mean(w > 4):.4f} P(>6) {np.mean(w > 6):.4f} P(>8) {np.mean(w > 8):.5f}")


if __name__ == "__main__":
    C = 64
    for factor, nband in ((2.0, 3), (3.0, 3), (4.0, 2), (3.0, 2), (2.0, 4)):
        m, walk, miss, lm, reb = run(factor, NBAND=nband, C=C)
        report(f"factor {factor} band {nband} NB 256", m, walk, miss, lm, reb, C)
