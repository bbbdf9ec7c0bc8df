"""Randomised cross-check of the two define_events() pipelines (table-only vs per-step kernels, both
exceedance kernels) on random shapes, label sequences, parameters, NaN fractions and persistences.
    python tools/fuzz_detect.py        # prints the number of mismatches (expected 0)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from xmhw_amd.detect_front import detect_cells
from xmhw_amd._lib import hip
h = hip()
rng = np.random.default_rng(2026)
bad = 0
for it in range(40):
    T = int(rng.integers(1, 900)); C = int(rng.integers(1, 3000))
    m = int(rng.integers(1, 12)); gap = int(rng.integers(0, m)); jg = bool(rng.integers(0, 2))
    D = int(rng.integers(1, min(T, 400) + 1))
    doy = (np.arange(T) % D) + 1 if rng.integers(0, 2) else rng.integers(1, D + 1, size=T)
    doys = np.arange(1, D + 1)
    rho = rng.uniform(0, 0.98)
    x = np.zeros((T, C)); e = rng.normal(size=(T, C))
    for k in range(1, T): x[k] = rho * x[k - 1] + e[k]
    x *= rng.uniform(0.3, 3.0)
    x[rng.random((T, C)) < rng.uniform(0, 0.1)] = np.nan
    se = rng.normal(size=(D, C)) * 0.2; th = se + rng.uniform(0.0, 1.5)
    dt = np.float32 if rng.integers(0, 2) else np.float64
    for mode in (1, 2):
        h.set_exceed_kernel(mode)
        a = detect_cells(x.astype(dt), se, th, doy, doys, m, jg, gap, coldSpells=bool(it % 2))
        b = detect_cells(x.astype(dt), se, th, doy, doys, m, jg, gap, coldSpells=bool(it % 2), per_step_kernels=True)
        ok = np.array_equal(a["offsets"], b["offsets"]) and np.array_equal(a["table"], b["table"], equal_nan=True)
        if not ok:
            bad += 1; print("MISMATCH", it, mode, T, C, m, gap, jg, D)
h.set_exceed_kernel(0)
print("fuzz done, mismatches:", bad)
