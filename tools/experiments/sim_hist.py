"""Model of the histogram variant: per cell NB equal-width buckets over [base, base + NB<<sh) in key
space (range = the first pool's range widened by MARGIN on both sides); per row the pivot is the
lower edge of the bucket that contains order statistic lo; need = lo - F(pivot).  How often is
need <= J-2 (no count pass), per cell and per wave of 8 cells; how often does the target leave
the covered range (re-base)?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sim_select import synth, f32_key
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import xmhw_oracle as ora
from oracle_fast import pool_index


def run(NB, margin, C=64, seed=0, q=0.9):
    time = np.arange("1982-01-01", "2022-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    x = synth(time.shape[0], C, seed)
    keys = f32_key(x)
    doys, pools = pool_index(doy, 5)
    D = len(doys)
    need = np.zeros((D, C), int); rebase = np.zeros(C, int)
    base = np.zeros(C, np.int64); sh = np.zeros(C, np.int64)
    for i, idx in enumerate(pools):
        pk = np.sort(keys[idx, :], axis=0); n = pk.shape[0]; lo = int(np.floor((n - 1) * q))
        for c in range(C):
            col = pk[:, c]
            tgt = int(col[lo])
            def setbase():
                w = int(col[-1] - col[0]) + 1
                b = int(col[0]) - int(margin * w)
                span = int((1 + 2 * margin) * w)
                s = max(0, int(np.ceil(np.log2(span / NB))))
                base[c], sh[c] = b, s
            if i == 0: setbase()
            b = (tgt - base[c]) >> sh[c]
            if b <= 0 or b >= NB - 1:
                rebase[c] += 1; setbase(); b = (tgt - base[c]) >> sh[c]
            edge = base[c] + (b << sh[c]) - 1              # pivot: everything in lower buckets is <= edge
            Fl = int(np.searchsorted(col, edge, side="right"))
            need[i, c] = lo - Fl
    need = need[1:]
    return need, rebase


if __name__ == "__main__":
    for NB in (512, 1024, 2048, 4096):
        for margin in (1.0, 1.5):
            need, rebase = run(NB, margin)
            D1, C = need.shape
            wave = need.reshape(D1, C // 8, 8).max(axis=2)
            print(f"NB={NB} margin={margin}: mean need {need.mean():.2f}; per cell P(need<=3) {np.mean(need<=3):.3f} "
                  f"P(<=5) {np.mean(need<=5):.3f}; per wave P(max<=3) {np.mean(wave<=3):.3f} P(max<=5) {np.mean(wave<=5):.3f} "
                  f"P(max<=7) {np.mean(wave<=7):.3f}; rebases per cell-year {rebase.mean():.2f}")
