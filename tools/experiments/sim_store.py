"""Offline model of the round-4 selection (kernels_ring4.hip): a per-cell WINDOW of NS rows of 2^shift keys each around the
target, every row a ring of CAPB keys (the keys of the pool inside the window are stored, the rest only counted).  Per row
of the climatology: which window row holds order statistic lo, how populous it is, whether the window has to be re-placed
(the target row nears an end), re-made (row population out of range: the width changes) or is lost (target outside).  Per
WAVE of `cpw` cells: how often a fill pass (a pass over the ring) and a slow row happen.  `--catchall`: the keys above the
window are kept in one more row, so that a window moving UP needs no pass over the ring.
Pure numpy; mirrors the placement policy of the kernel so that NS / CAPB / PLACE / EDGE / the population range can be
chosen before touching the HIP code.

    python tools/sim_store.py [--cells 64] [--ns 8] [--capb 32]
"""
import argparse
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sim_select import synth, f32_key      # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import xmhw_oracle as ora                  # noqa: E402
from oracle_fast import pool_index         # noqa: E402

_cache = {}


def pools_sorted(C, seed, years):
    key = (C, seed, years)
    if key not in _cache:
        time = np.arange(f"{years[0]}-01-01", f"{years[1] + 1}-01-01", dtype="datetime64[D]")
        doy = ora.add_doy(time)
        keys = f32_key(synth(time.shape[0], C, seed))
        _, pools = pool_index(doy, 5)
        _cache[key] = [np.sort(keys[idx, :], axis=0) for idx in pools]
    return _cache[key]


def run(C=64, NS=8, CAPB=32, mu=(5, 18), mu_t=10, place=2, edge_lo=1, edge_hi=1, cpw=16, seed=0, q=0.9, years=(1982, 2021),
        catchall=False, ccap=16, ema=0.0, replace_all=True, gain=8):
    P = pools_sorted(C, seed, years)
    D = len(P)
    mu_lo, mu_hi = mu
    wbase = np.zeros(C, np.int64); sh = np.zeros(C, np.int64); valid = np.zeros(C, bool); built = np.zeros(C, bool)
    m16 = np.full(C, 16.0 * mu_t)
    kfill = np.zeros(C, np.int64); vel = np.zeros(C); prev = np.zeros(C, np.int64)
    fill = np.zeros((D, C), bool)      # asks for a pass over the ring
    slow = np.zeros((D, C), bool)
    cheap = np.zeros((D, C), bool)     # a move up served from the catch-all row
    why = dict(inv=0, edge_up=0, edge_dn=0, pop=0, lost=0, cap=0)
    for i, pk in enumerate(P):
        n = pk.shape[0]
        lo = int(np.floor((n - 1) * q))
        ask = np.zeros(C, bool); ask_cheap = np.zeros(C, bool); fresh = np.zeros(C, bool); rl = np.zeros(C, int)
        tgt = pk[lo, :].astype(np.int64)
        if i:
            vel = ema * vel + (1 - ema) * (tgt - prev)
        prev = tgt
        for c in range(C):
            col = pk[:, c]
            t, t1 = int(tgt[c]), int(col[min(lo + 1, n - 1)])
            if not valid[c]:
                slow[i, c] = True; ask[c] = True; fresh[c] = True; why["inv"] += 1
                continue
            R, R1 = (t - wbase[c]) >> sh[c], (t1 - wbase[c]) >> sh[c]
            top = NS if not catchall else NS + 1
            if R < 0 or R >= NS or R1 >= top:
                slow[i, c] = True; ask[c] = True; fresh[c] = True; why["lost"] += 1
                continue
            w = 1 << sh[c]
            e0 = wbase[c] + R * w
            cR = int(np.searchsorted(col, e0 + w - 1, side="right") - np.searchsorted(col, e0 - 1, side="right"))
            if R1 >= NS:       # the catch-all row: everything above the window
                cR1 = int(n - np.searchsorted(col, wbase[c] + NS * w - 1, side="right"))
                capR1 = ccap
            else:
                e1 = wbase[c] + R1 * w
                cR1 = int(np.searchsorted(col, e1 + w - 1, side="right") - np.searchsorted(col, e1 - 1, side="right"))
                capR1 = CAPB
            rl[c] = R
            if cR >= CAPB or (R1 != R and cR1 >= capR1):
                slow[i, c] = True; why["cap"] += 1
                m16[c] += (16 * min(cR, 63) - m16[c]) / 4
            else:
                m16[c] += (16 * cR - m16[c]) / gain
            if (m16[c] > 16 * mu_hi and sh[c] > 0) or m16[c] < 16 * mu_lo:
                ask[c] = True; fresh[c] = True; why["pop"] += 1
            elif R < edge_lo:
                ask[c] = True; why["edge_dn"] += 1
            elif R >= NS - edge_hi:
                why["edge_up"] += 1
                if catchall:
                    above = int(n - np.searchsorted(col, wbase[c] + NS * w - 1, side="right"))
                    if above < ccap:
                        ask_cheap[c] = True
                    else:
                        ask[c] = True
                else:
                    ask[c] = True
        for w0 in range(0, C, cpw):
            cells = range(w0, min(w0 + cpw, C))
            any_fill = any(ask[c] for c in cells)
            any_cheap = any(ask_cheap[c] for c in cells)
            if not (any_fill or any_cheap):
                continue
            for c in cells:
                col = pk[:, c]
                t = int(tgt[c])
                if not any_fill and not ask_cheap[c] and not (replace_all and catchall):
                    continue
                if not any_fill:
                    # only moves UP are possible without a pass over the ring
                    if not (ask_cheap[c] or (replace_all and rl[c] > place)):
                        continue
                if fresh[c] or not valid[c]:
                    if not built[c]:
                        span = int(col[min(lo + 4, n - 1)] - col[max(lo - 4, 0)]) / 8.0
                        sh[c] = int(np.floor(np.log2(max(span * mu_t, 1.0))))
                        m16[c] = 16.0 * mu_t
                        built[c] = True
                    else:
                        for _ in range(3):
                            if m16[c] > 16 * mu_hi and sh[c] > 0:
                                sh[c] -= 1; m16[c] /= 2
                            elif m16[c] < 16 * mu_lo and sh[c] < 26:
                                sh[c] += 1; m16[c] *= 2
                if not ask[c] and not fresh[c] and not replace_all and not ask_cheap[c]:
                    continue
                up = (vel[c] >= 0) if ema > 0 else (t >= kfill[c])
                if not any_fill:
                    up = True
                behind = place if up else NS - 1 - place
                rb = max((t >> sh[c]) - behind, 1)
                if not any_fill and (rb << sh[c]) < wbase[c]:
                    continue
                wbase[c] = rb << sh[c]
                valid[c] = True
                kfill[c] = t
            if any_fill:
                for c in cells:
                    fill[i, c] = True
            else:
                for c in cells:
                    cheap[i, c] = True
    f = fill[1:].reshape(D - 1, C // cpw, cpw).any(axis=2).mean()
    s = slow[1:].reshape(D - 1, C // cpw, cpw).any(axis=2).mean()
    ch = cheap[1:].reshape(D - 1, C // cpw, cpw).any(axis=2).mean()
    cells_rows = float((D - 1) * C)
    return dict(fill_wave=f, cheap_wave=ch, slow_wave=s, slow_cell=slow[1:].mean(),
                **{k: v / cells_rows for k, v in why.items()})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=64)
    ap.add_argument("--cpw", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    base = dict(C=args.cells, cpw=args.cpw, seed=args.seed)
    trials = [
        dict(),                                                  # the first build of the kernel
        dict(place=3),
        dict(ema=0.7),
        dict(ema=0.85, place=2),
        dict(mu=(7, 24), mu_t=14),
        dict(catchall=True),
        dict(catchall=True, edge_hi=2),
        dict(catchall=True, edge_hi=2, ema=0.8),
        dict(catchall=True, edge_hi=2, ema=0.8, place=1),
        dict(catchall=True, edge_hi=2, ema=0.8, mu=(7, 24), mu_t=14),
        dict(NS=16, CAPB=16, mu=(2.5, 9), mu_t=5, place=4, edge_lo=2, edge_hi=2),
        dict(NS=16, CAPB=16, mu=(2.5, 9), mu_t=5, place=4, edge_lo=2, edge_hi=4, catchall=True, ema=0.8),
    ]
    for t in trials:
        r = run(**base, **t)
        print(f"{t}: " + " ".join(f"{k}={v:.4f}" for k, v in r.items()), flush=True)
