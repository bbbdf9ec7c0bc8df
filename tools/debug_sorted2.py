"""debug: the random cases of tests/test_gpu_sorted.py one by one"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import xmhw_amd.device as dev
import test_gpu_sorted as T
rng = np.random.default_rng(2026)
for it in range(12):
    years = int(rng.integers(9, 49))
    y0 = int(rng.integers(1950, 1975))
    doy = T._daily(y0, y0 + years - 1)
    C = int(rng.integers(1, 90))
    nanfrac = float(rng.choice([0.0, 0.02, 0.3])); quant = float(rng.choice([0.0, 0.01, 0.5])) or None; amp = (0.1, float(rng.choice([3, 10, 25])))
    x = T._series(doy.shape[0], C, 100 + it, nanfrac=nanfrac, quant=quant, amp=amp)
    q = float(rng.choice([0.9, 0.8, 0.97])); neg = bool(rng.integers(0, 2))
    tg, sg, _, _ = T._raw(dev, x, doy, q, neg, kernel="generic")
    t1, s1, st, use = T._raw(dev, x, doy, q, neg, layout="sorted")
    bad = np.argwhere(~((t1 == tg) | (np.isnan(t1) & np.isnan(tg))))
    print(it, "years", years, "C", C, "nan", nanfrac, "quant", quant, "amp", amp, "q", q, "neg", neg, "mismatches", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:12], "cells", sorted(set(bad[:, 1].tolist()))[:8])
