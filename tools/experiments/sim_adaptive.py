"""Model: exactly N1 count passes (fewer if every cell is inside the widest window), then
one extraction whose width J is chosen per wave from a menu, covering max over the wave's
cells of (lo - Fl) + 2; cells beyond the widest J take repair rounds."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sim_select import synth, f32_key
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import xmhw_oracle as ora
from oracle_fast import pool_index

C_ITER = 170
def E(J): return 55 * (J + 1) + 3 * (2 * J + {3: 2, 4: 4, 5: 5, 6: 7, 7: 9, 8: 12}.get(J, 12)) + 25


def cell_pass(st, col, lo, n, aimslack, N1, Jmax):
    """run up to N1 probes aiming at window slack aimslack; stop early if within slack Jmax-2.
    returns (passes, need) with need = lo - Fl after the passes"""
    F = lambda p: int(np.searchsorted(col, p, side="right"))
    kpr = st.get("kpr", 8192.0)
    if "pc" in st:
        p0 = st["pc"]; F0 = F(p0); cnt = 0
    else:
        p0 = int(col[len(col) // 2]); F0 = F(p0); cnt = 1
    pl, Fl, ph, Fh = 0, 0, 0xFFFFFFFF, n
    lreal = hreal = False
    if F0 <= lo: pl, Fl, lreal = p0, F0, True
    else: ph, Fh, hreal = p0, F0, True
    aim = lo - aimslack / 2.0 + 0.5
    grow = 1.0
    while cnt < N1 and not (lo - Fl <= aimslack or ph - pl <= 1):
        room = ph - pl
        both = lreal and hreal
        slope = room / (Fh - Fl) if both else kpr * grow
        ranks = (aim - Fl) if lreal else (Fh - aim)
        stf = min(max(ranks * slope, 1.0), 2e9)
        stf = stf if lreal else room - stf
        off = max(1, min(int(stf), room - 1))
        if not both: grow *= 2
        p = pl + off
        Fp = F(p); cnt += 1
        if Fp <= lo: pl, Fl, lreal = p, Fp, True
        else: ph, Fh, hreal = p, Fp, True
    alo = int(col[lo]); gap = lo - F0
    if abs(gap) > 1:
        obs = (alo - p0) / gap
        if 1 <= obs < 1e8: st["kpr"] = 0.5 * kpr + 0.5 * obs
    need = lo - Fl
    # carried pivot: the best lower pivot we end with (its count is exact)
    st["pc"] = pl if pl > 0 else int(col[max(lo - 1, 0)])
    return cnt, need


def run(N1, menu, aimslack, C=64, seed=0):
    time = np.arange("1982-01-01", "2022-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    x = synth(time.shape[0], C, seed)
    keys = f32_key(x)
    doys, pools = pool_index(doy, 5)
    D = len(doys)
    P = np.zeros((D, C), int); N = np.zeros((D, C), int)
    state = [dict() for _ in range(C)]
    for i, idx in enumerate(pools):
        pk = np.sort(keys[idx, :], axis=0); n = pk.shape[0]; lo = int(np.floor((n - 1) * 0.9))
        for c in range(C):
            P[i, c], N[i, c] = cell_pass(state[c], pk[:, c], lo, n, aimslack, N1, max(menu))
    P, N = P[1:], N[1:]
    Pw = P.reshape(P.shape[0], C // 8, 8).max(axis=2)
    Nw = N.reshape(N.shape[0], C // 8, 8).max(axis=2) + 2        # J needed by the wave
    cost = np.zeros(Pw.shape)
    hist = {}
    for J in sorted(menu):
        pass
    Jsel = np.zeros(Nw.shape, int)
    for idx_, need in np.ndenumerate(Nw):
        js = [J for J in sorted(menu) if J >= need]
        if js:
            Jsel[idx_] = js[0]; cost[idx_] = E(js[0])
        else:   # beyond the menu: repair rounds, each advances max(menu)-1 ranks
            Jm = max(menu); extra = int(np.ceil((need - Jm) / (Jm - 1)))
            Jsel[idx_] = 99; cost[idx_] = E(Jm) + extra * (C_ITER + E(Jm))
    total = (Pw * C_ITER + cost).mean()
    return Pw.mean(), {int(j): float((Jsel == j).mean()) for j in np.unique(Jsel)}, total


if __name__ == "__main__":
    print("baseline (current kernel): J=5 window, unbounded passes ->", 2.76 * C_ITER + E(5))
    for N1 in (1, 2, 3):
        for menu in ((3, 5, 8), (4, 6, 8), (5, 8), (3, 5, 7, 10)):
            for aimslack in (1, 3):
                pw, h, tot = run(N1, menu, aimslack)
                print(f"N1={N1} menu={menu} aim-slack={aimslack}: passes/wave {pw:.2f} J mix {h} -> selection VALU/wave-row {tot:.0f}")
