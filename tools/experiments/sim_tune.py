"""Parameter scan of the probe heuristics of the ring kernel (offline model, as sim_phased.py):
aim point inside the rank window, keys-per-rank blending, step damping.  Reports count passes per
wave-row (max over the 8 cells of a wave) for J = 5, budget 6."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sim_select import synth, f32_key
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import xmhw_oracle as ora
from oracle_fast import pool_index


def cell_row(st, col, lo, n, J, aim_off, beta, gamma):
    F = lambda p: int(np.searchsorted(col, p, side="right"))
    slack = J - 2
    kpr = st.get("kpr", 8192.0)
    cnt = 0
    if "pc" in st:
        p0 = st["pc"]; F0 = F(p0)
    else:
        p0 = int(col[len(col) // 2]); F0 = F(p0); cnt += 1
    pl, Fl, ph, Fh = 0, 0, 0xFFFFFFFF, n
    lreal = hreal = False
    if F0 <= lo: pl, Fl, lreal = p0, F0, True
    else: ph, Fh, hreal = p0, F0, True
    aim = lo - aim_off
    grow = 1.0; it = 0
    while cnt < 12 and not (0 <= lo - Fl <= slack or ph - pl <= 1):
        room = ph - pl
        both = lreal and hreal
        slope = room / (Fh - Fl) if both else kpr * grow
        ranks = (aim - Fl) if lreal else (Fh - aim)
        g = gamma if it == 0 else 1.0
        stf = min(max(ranks * slope * g, 1.0), 2e9)
        stf = stf if lreal else room - stf
        stf = min(max(stf, 1.0), 4e9)
        off = int(stf) if it < 5 else room >> 1
        if not both: grow *= 2
        off = max(1, min(off, room - 1))
        p = pl + off
        Fp = F(p); cnt += 1
        if Fp <= lo: pl, Fl, lreal = p, Fp, True
        else: ph, Fh, hreal = p, Fp, True
        it += 1
    alo = int(col[lo]); gap = lo - F0
    if abs(gap) > 1:
        obs = (alo - p0) / gap
        if 1 <= obs < 1e8: st["kpr"] = (1 - beta) * kpr + beta * obs
    st["pc"] = pl
    return cnt


def run(aim_off, beta, gamma, C=64, J=5, seed=0, q=0.9, w=5, years=(1982, 2021)):
    time = np.arange(f"{years[0]}-01-01", f"{years[1]+1}-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    x = synth(time.shape[0], C, seed)
    keys = f32_key(x)
    doys, pools = pool_index(doy, w)
    D = len(doys)
    P = np.zeros((D, C), int)
    state = [dict() for _ in range(C)]
    for i, idx in enumerate(pools):
        pk = np.sort(keys[idx, :], axis=0)
        n = pk.shape[0]; lo = int(np.floor((n - 1) * q))
        for c in range(C):
            P[i, c] = cell_row(state[c], pk[:, c], lo, n, J, aim_off, beta, gamma)
    Pw = P[1:].reshape(D - 1, C // 8, 8).max(axis=2)
    return P[1:].mean(), Pw.mean()


if __name__ == "__main__":
    base = run(1.0, 0.5, 1.0)
    print("current (aim_off 1.0, beta 0.5, gamma 1.0): per cell %.3f per wave %.3f" % base)
    for aim_off in (0.5, 1.0, 1.5, 2.0):
        for beta in (0.25, 0.5, 0.75):
            for gamma in (0.9, 1.0, 1.1):
                pc, pw = run(aim_off, beta, gamma)
                print(f"aim_off {aim_off} beta {beta} gamma {gamma}: per cell {pc:.3f} per wave {pw:.3f}")
