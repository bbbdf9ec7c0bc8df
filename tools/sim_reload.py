"""Offline model: how often would the sorted-list kernel have to RELOAD a row-list (re-read its samples, sort them, store
another window of K ranks) if a list kept a sliding window [s0, s0 + K) of its ranks instead of its K largest keys?
Counts reload events per cell-row and per wave-row (any of the 32 cells of a wave) for K = 8..16 on the SURVEY 8(d)
generator.  A new list starts at s0 = 0; a reload re-centres the window on the list's pointer."""
import sys
import numpy as np
sys.path.insert(0, "tools")
import sim_sorted as S

C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "gauss"
Y, DY, w, q = 40, 365, 5, 0.9
R = 11
keys = S.f32_key(S.synth(Y * DY, C, 1, kind)).reshape(Y, DY, C)
idx = np.arange(DY + R) % DY
lists = np.sort(keys[:, idx, :], axis=0)[::-1]
cs = np.zeros((DY, R, C), dtype=np.int64)            # staircase per row: c of the lists in the pool (oldest first)
for r in range(DY):
    pool = lists[:, r:r + R, :]
    flat = np.sort(pool.reshape(Y * R, C), axis=0)
    n = (flat > 0).sum(axis=0)
    lo = np.floor((n - 1) * q).astype(int)
    vlo = flat[(Y * R - n) + lo, np.arange(C)]
    cs[r] = (pool > vlo[None, None, :]).sum(axis=0)
for K in (8, 10, 12, 14, 16):
    ev = np.zeros((DY, C), dtype=np.int64)
    # follow every list (pushed at row p, slot R-1, ages to slot 0 at row p + R - 1)
    for p in range(DY - R):
        s0 = np.zeros(C, dtype=np.int64)
        for a in range(R):
            r = p + a
            c = cs[r, R - 1 - a]                     # absolute pointer of this list at row r
            need = (c >= s0 + K) | (c <= s0 - 1) | ((c == s0) & (s0 > 0) & False)
            # pointer at the bottom of the window with keys beyond it, or above the window
            need = (c - s0 >= K) | (c - s0 < 0)
            ev[r] += need
            s0 = np.where(need, np.clip(c - K // 2, 0, Y - K), s0)
    e = ev[R:DY - R]
    wv = (e.reshape(e.shape[0], -1, 32) > 0).any(axis=2)
    print(f"K = {K:2d}: reloads per cell-row {e.mean():.5f}; wave-rows (32 cells) with a reload {100 * wv.mean():.2f} %; "
          f"reloads per wave-row {e.reshape(e.shape[0], -1, 32).sum(axis=2).mean():.3f}")
