"""Model of the phased selection: <=N1 count passes, one top-J extraction, then
repair rounds (count at the largest extracted key + extraction) while any cell
of the wave is unresolved.  Reports wave-level cost in VALU instructions."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sim_select import synth, f32_key
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import xmhw_oracle as ora
from oracle_fast import pool_index

C_COUNT, C_ITER = 115, 25          # VALU per count pass, per-iteration control


def extraction_cost(J): return (J + 1) * 55 + 16 * 3 + 30


def cell_row(st, col, lo, n, J, N1, aim):
    """returns (count_passes_phase1, repair_rounds).  st carries pc/kpr."""
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
    tgt = lo - aim
    grow = 1.0; it = 0
    while cnt < N1 and not (0 <= lo - Fl <= slack or ph - pl <= 1):
        room = ph - pl
        if lreal and hreal:
            frac = min(max((tgt - Fl + 0.5) / (Fh - Fl), 0.0), 1.0)
            off = int(room * frac)
        elif lreal:
            off = int((tgt - Fl + 0.5) * kpr * grow); grow *= 2
        else:
            off = room - int((Fh - tgt - 0.5) * kpr * grow); grow *= 2
        off = max(1, min(off, room - 1))
        p = pl + off
        Fp = F(p); cnt += 1
        if Fp <= lo: pl, Fl, lreal = p, Fp, True
        else: ph, Fh, hreal = p, Fp, True
        it += 1
    rounds = 0
    while True:
        if ph - pl <= 1 and not (0 <= lo - Fl <= slack):
            break
        m = col[Fl:Fl + J]
        if lo - Fl <= slack:
            break
        rounds += 1
        p = int(m[-1]); Fp = F(p)
        if Fp <= lo:
            pl, Fl = p, Fp
        else:
            ph, Fh = p, Fp
            pl, Fl = p - 1, Fl + int(np.sum(m < p))
    alo = int(col[lo]); gap = lo - F0
    if abs(gap) > 1:
        obs = (alo - p0) / gap
        if 1 <= obs < 1e8: st["kpr"] = 0.5 * kpr + 0.5 * obs
    st["pc"] = pl
    return cnt, rounds


def run(J, N1, aim=None, C=32, quant=None, seed=0, q=0.9, w=5, years=(1982, 2021)):
    aim = (J - 2) / 2.0 if aim is None else aim
    time = np.arange(f"{years[0]}-01-01", f"{years[1]+1}-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    x = synth(time.shape[0], C, seed)
    if quant: x = (np.round(x / quant) * quant).astype(np.float32)
    keys = f32_key(x)
    doys, pools = pool_index(doy, w)
    D = len(doys)
    P = np.zeros((D, C), int); Rr = np.zeros((D, C), int)
    state = [dict() for _ in range(C)]
    for i, idx in enumerate(pools):
        pk = np.sort(keys[idx, :], axis=0)
        n = pk.shape[0]; lo = int(np.floor((n - 1) * q))
        for c in range(C):
            P[i, c], Rr[i, c] = cell_row(state[c], pk[:, c], lo, n, J, N1, aim)
    Pw = P[1:].reshape(D - 1, C // 8, 8).max(axis=2)
    Rw = Rr[1:].reshape(D - 1, C // 8, 8).max(axis=2)
    cost = (Pw * (C_COUNT + C_ITER) + extraction_cost(J) + Rw * (C_COUNT + C_ITER + extraction_cost(J))).mean()
    return P[1:].mean(), Pw.mean(), Rw.mean(), cost


if __name__ == "__main__":
    for quant in (None, 0.01):
        for J in (3, 4, 5, 6):
            for N1 in (1, 2, 3, 4):
                pc, pw, rw, cost = run(J, N1, quant=quant, C=16)
                print(f"quant={quant} J={J} N1={N1}: count/cell {pc:.2f} count/wave {pw:.2f} repair/wave {rw:.2f} -> VALU/wave-row {cost:.0f}")
