#!/usr/bin/env python3
"""Re-draw one case of tools/fuzz_ring2.py (same generator, same seed) and show where layout 40 differs from the generic
kernel:  python tools/repro_fuzz.py --seed 601 --case 485 --years 37 41 --sorted-only"""
import argparse
import os
import sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_ring2 as fz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--case", type=int, required=True)
    ap.add_argument("--years", type=int, nargs=2, default=[9, 49])
    ap.add_argument("--sorted-only", action="store_true")
    args = ap.parse_args()
    import xmhw_amd.device as dev
    rng = np.random.default_rng(args.seed)
    i = 0
    while True:
        x, doy, pct, tstep, cold, nchunks = fz.random_ring2_case(rng, tuple(args.years))
        if args.sorted_only:
            if 15 < pct < 85:
                continue
            plan = dev.Plan(doy, 5)
            ok = plan.ring2_in_use() == 40
            plan.destroy()
            if not ok:
                continue
        if i == args.case:
            break
        i += 1
    t0, s0, _ = fz._raw(dev, x, doy, pct / 100.0, cold, kernel="generic")
    t1, s1, _ = fz._raw(dev, x, doy, pct / 100.0, cold, nchunks, ring2=40)
    bad = np.argwhere(~((t1 == t0) | (np.isnan(t1) & np.isnan(t0))))
    print("T", x.shape, "pct", pct, "cold", cold, "nchunks", nchunks, "tracks", len(np.unique(doy)), "mismatches", len(bad))
    cells = sorted(set(int(b[1]) for b in bad))
    print("cells", cells)
    for c in cells[:4]:
        rows = [int(b[0]) for b in bad if b[1] == c]
        print("cell", c, "rows", rows[:40])
        for r in rows[:6]:
            print("   row", r, "got", t1[r, c], "want", t0[r, c])
        col = x[:, c]
        print("   distinct values", len(np.unique(col[np.isfinite(col)])), "nan", int(np.isnan(col).sum()))


if __name__ == "__main__":
    main()
