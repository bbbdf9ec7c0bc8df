#!/usr/bin/env python3
"""What a narrower select round would cost (round 5, after the serial finish): for every cell-row of a synthetic 40-year
daily grid the keys that must move (the symmetric difference between the carried top set and the new one), the lists they
come from, and what ONE round with windows of W keys per list and a cap of `cap` keys moves of them (a window is trusted
down to the largest (W + 1)-th key).  Per wave of 32 cells: rows that need nothing more, rows finished key by key (worst
cell <= 3 keys left), rows that need another round.  Usage: python tools/experiments/sim_window.py [cells]
"""
import sys
import numpy as np

R, q = 11, 0.9


def series(T, C, seed=1):
    rng = np.random.default_rng(seed)
    t = np.arange(T)[:, None]
    return (15 + rng.uniform(2, 10, C) * np.sin(2 * np.pi * (t - rng.uniform(0, 365, C)) / 365.25)
            + 0.0005 * t * rng.uniform(-1, 1, C) + rng.normal(size=(T, C))).astype(np.float32)


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    ny = 40
    T = 365 * ny                       # (no leap days: every row regular)
    x = series(T, C)
    x3 = x.reshape(ny, 365, C)         # [year][day][cell]
    rows = range(20, 345)              # away from the year ends
    out = {}
    for W, cap in ((4, 15), (3, 11), (3, 15), (2, 7)):
        left = np.zeros((len(rows), C), dtype=np.int32)
        first = np.zeros((len(rows), C), dtype=np.int32)
        for ri, r in enumerate(rows):
            # lists of the row's pool: days r-5 .. r+5, each 40 keys; the previous row's pool: r-6 .. r+4
            pool = x3[:, r - 5:r + 6, :]                    # [ny][11][C]
            prev = x3[:, r - 6:r + 5, :]
            n = ny * R
            lo = int(np.floor((n - 1) * q))
            Cs = n - 1 - lo
            for c in range(C):
                pv = np.sort(prev[:, :, c].ravel())
                B = pv[lo]                                   # largest key outside the previous top set
                lists = pool[:, :, c].T                      # [11][40]
                inside = lists > B
                Ctop = int(inside.sum())
                if Ctop == Cs:
                    continue
                flat = lists.ravel()
                lid = np.repeat(np.arange(R), ny)
                if Ctop < Cs:                                # grow: the largest keys <= B, in descending order
                    cand = np.where(flat <= B)[0]
                    order = cand[np.argsort(-flat[cand], kind="stable")]
                else:                                        # shrink: the smallest keys > B, ascending
                    cand = np.where(flat > B)[0]
                    order = cand[np.argsort(flat[cand], kind="stable")]
                rem = abs(Cs - Ctop)
                first[ri, c] = rem
                d = min(rem, cap)
                # per list: its keys in move order
                moved = d
                cnt = np.bincount(lid[order[:d]], minlength=R)
                if cnt.max() > W:
                    # the largest (W + 1)-th key over the lists, in move order: the first position at which some list shows its
                    # (W + 1)-th key; everything before that position is safe
                    seen = np.zeros(R, dtype=np.int32)
                    moved = 0
                    for p in order[:d]:
                        if seen[lid[p]] == W:
                            break
                        seen[lid[p]] += 1
                        moved += 1
                left[ri, c] = rem - moved
        lw = left.reshape(len(rows), C // 32, 32).max(axis=2)
        tot = lw.size
        out[(W, cap)] = dict(done=float((lw == 0).sum()) / tot, serial=float(((lw > 0) & (lw <= 3)).sum()) / tot,
                             serial_keys=float(lw[(lw > 0) & (lw <= 3)].sum()) / tot, again=float((lw > 3).sum()) / tot,
                             mean_steps=float(first.mean()))
        print(W, cap, out[(W, cap)], flush=True)


if __name__ == "__main__":
    main()
