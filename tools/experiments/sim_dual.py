"""Model: each count pass evaluates F at TWO pivots (v_sad_u16 on packed 16-bit
local keys costs 0.5 op per key per S evaluation).  How many passes per wave-row?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sim_select import synth, f32_key
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import xmhw_oracle as ora
from oracle_fast import pool_index


def cell_row(st, col, lo, n, J, delta, grid):
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
    aim = lo - slack / 2.0 + 0.5
    grow = 1.0
    it = 0
    while not (0 <= lo - Fl <= slack or ph - pl <= 1):
        room = ph - pl
        both = lreal and hreal
        slope = room / (Fh - Fl) if both else kpr * grow
        probes = []
        for d in (-delta, +delta):
            ranks = (aim + d - Fl) if lreal else (Fh - (aim + d))
            stf = min(max(ranks * slope, 1.0), 2e9)
            stf = stf if lreal else room - stf
            off = int(min(max(stf, 1.0), 4e9)) if it < 5 else (room >> 1) + int(d)
            off = max(1, min(off, room - 1))
            p = pl + off
            if grid > 1:                         # probes live on the 16-bit grid
                p = max(pl + 1, min(ph - 1, (p // grid) * grid + grid - 1))
            probes.append(p)
        if not both: grow *= 2
        cnt += 1
        for p in probes:
            Fp = F(p)
            if Fp <= lo:
                if p > pl: pl, Fl, lreal = p, Fp, True
            else:
                if p < ph: ph, Fh, hreal = p, Fp, True
        it += 1
    alo = int(col[lo]); gap = lo - F0
    if abs(gap) > 1:
        obs = (alo - p0) / gap
        if 1 <= obs < 1e8: st["kpr"] = 0.5 * kpr + 0.5 * obs
    st["pc"] = pl if 0 <= lo - Fl <= slack else ph
    return cnt


def run(J, delta, grid=1, C=64, seed=0):
    time = np.arange("1982-01-01", "2022-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    x = synth(time.shape[0], C, seed)
    keys = f32_key(x)
    doys, pools = pool_index(doy, 5)
    D = len(doys)
    P = np.zeros((D, C), int)
    state = [dict() for _ in range(C)]
    for i, idx in enumerate(pools):
        pk = np.sort(keys[idx, :], axis=0); n = pk.shape[0]; lo = int(np.floor((n - 1) * 0.9))
        for c in range(C):
            P[i, c] = cell_row(state[c], pk[:, c], lo, n, J, delta, grid)
    P = P[1:]
    return P.mean(), P.reshape(P.shape[0], C // 8, 8).max(axis=2).mean()


if __name__ == "__main__":
    for J in (4, 5, 6):
        for delta in (1.0, 1.5, 2.0, 3.0):
            pc, pw = run(J, delta, grid=256)
            print(f"J={J} delta={delta}: dual passes per cell {pc:.2f}, per wave {pw:.2f}")
