"""Offline model of the ring kernel's bracket search: how many count passes per
row does a probe strategy need (mean per cell, and mean of the max over the 8
cells of a wave, which is what the wave pays)?  Pure numpy; used to tune the
heuristics before touching the HIP code."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import xmhw_oracle as ora
from oracle_fast import pool_index


def f32_key(x):
    b = x.astype(np.float32).view(np.uint32).astype(np.int64)
    return np.where(b & 0x80000000, (~b) & 0xFFFFFFFF, b | 0x80000000)


def synth(T, C, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(T)[:, None]
    A = rng.uniform(2, 10, C); phi = rng.uniform(0, 365, C); beta = rng.uniform(-1, 1, C)
    return (15 + A * np.sin(2 * np.pi * (t - phi) / 365.25) + 0.0005 * t * beta + rng.normal(size=(T, C))).astype(np.float32)


def run(strategy, C=32, years=(1982, 2021), q=0.9, w=5, seed=0, verbose=False):
    time = np.arange(f"{years[0]}-01-01", f"{years[1]+1}-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    x = synth(time.shape[0], C, seed)
    keys = f32_key(x)
    doys, pools = pool_index(doy, w)
    D = len(doys)
    passes = np.zeros((D, C), dtype=np.int64)
    state = [dict() for _ in range(C)]
    for i, idx in enumerate(pools):
        pk = np.sort(keys[idx, :], axis=0)
        n = pk.shape[0]
        lo = int(np.floor((n - 1) * q))
        for c in range(C):
            col = pk[:, c]
            F = lambda p: int(np.searchsorted(col, p, side="right"))
            passes[i, c] = strategy(state[c], col, F, lo, n)
    per_cell = passes[1:].mean()
    per_wave = passes[1:].reshape(D - 1, C // 8, 8).max(axis=2).mean()
    return per_cell, per_wave, passes


def bracket_loop(st, F, lo, n, first_probes, kpr, maxsec=6):
    """generic bracket search; first_probes: list of (p, Fknown or None)."""
    pl, Fl, ph, Fh = 0, 0, 0xFFFFFFFF, n
    lreal = hreal = False
    cnt = 0
    grow = 1.0
    it = 0
    for p, Fk in first_probes:
        if Fk is None:
            Fk = F(p); cnt += 1
        if Fk <= lo:
            if p > pl: pl, Fl, lreal = p, Fk, True
        else:
            if p < ph: ph, Fh, hreal = p, Fk, True
    while not (Fl == lo or ph - pl <= 1):
        room = ph - pl
        if it < maxsec and lreal and hreal:
            off = int(room * (lo - Fl) / (Fh - Fl))
        elif it < maxsec and lreal:
            off = int((lo - Fl + 0.25) * kpr * grow); grow *= 2
        elif it < maxsec and hreal:
            off = room - int((Fh - lo - 0.25) * kpr * grow); grow *= 2
        else:
            off = room >> 1
        off = max(1, min(off, room - 1))
        p = pl + off
        Fp = F(p); cnt += 1
        if Fp <= lo: pl, Fl, lreal = p, Fp, True
        else: ph, Fh, hreal = p, Fp, True
        it += 1
    return cnt, pl, Fl, ph, Fh


def strat_v2(st, col, F, lo, n):
    """what kernels_ring.hip (second version) does: free probe at carried pivot."""
    kpr = st.get("kpr", 8192.0)
    if "pc" in st:
        p0 = st["pc"]; F0 = F(p0)   # free (incremental)
        probes = [(p0, F0)]
    else:
        p0 = int(col[len(col)//2]); F0 = F(p0); probes = [(p0, F0)]
    cnt, pl, Fl, ph, Fh = bracket_loop(st, F, lo, n, probes, kpr)
    if "pc" not in st: cnt += 1
    alo = int(col[lo])
    gap = lo - F0
    if abs(gap) > 1:
        obs = (alo - p0) / gap
        if 1 <= obs < 1e8: st["kpr"] = 0.5 * kpr + 0.5 * obs
    st["pc"] = pl if Fl == lo else ph
    return cnt


def strat_v1(st, col, F, lo, n):
    """first version: real probe at previous answer - 1, rho from gap EMA."""
    rho = st.get("rho", 8192.0)
    p0 = st["prev"] - 1 if "prev" in st else int(col[len(col)//2])
    cnt, pl, Fl, ph, Fh = bracket_loop(st, F, lo, n, [(p0, None)], rho, maxsec=7)
    alo, ahi = int(col[lo]), int(col[min(lo+1, n-1)])
    if ahi > alo: st["rho"] = max(1.0, 0.75 * rho + 0.25 * min(ahi - alo, 1 << 24))
    st["prev"] = alo
    return cnt


if __name__ == "__main__":
    for name, s in (("v1", strat_v1), ("v2", strat_v2)):
        pc, pw, _ = run(s, C=32, years=(1982, 2021))
        print(f"{name}: count passes per cell-row {pc:.2f}, per wave-row (max of 8) {pw:.2f}")


def hist(passes):
    h = np.bincount(passes[1:].ravel())
    return {i: int(v) for i, v in enumerate(h) if v}


def make_strat(J=4, aim=None, maxsec=6, blend=False):
    """free probe at carried pivot; stop when 0 <= lo - Fl <= J-2 (top-J extraction
    above pl then yields a[lo], a[lo+1]); probes aim at rank lo - aim."""
    slack = J - 2
    aim_ = slack / 2.0 if aim is None else aim

    def strat(st, col, F, lo, n):
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
        tgt = lo - aim_          # fractional target rank for F
        grow = 1.0; it = 0
        while not (0 <= lo - Fl <= slack or ph - pl <= 1):
            room = ph - pl
            if it < maxsec and lreal and hreal:
                frac = (tgt - Fl + 0.5) / (Fh - Fl)
                frac = min(max(frac, 0.0), 1.0)
                off = int(room * frac)
                if blend and it >= 2: off = (off + (room >> 1)) >> 1
            elif it < maxsec and lreal:
                off = int((tgt - Fl + 0.5) * kpr * grow); grow *= 2
            elif it < maxsec and hreal:
                off = room - int((Fh - tgt - 0.5) * kpr * grow); grow *= 2
            else:
                off = room >> 1
            off = max(1, min(off, room - 1))
            p = pl + off
            Fp = F(p); cnt += 1
            if Fp <= lo: pl, Fl, lreal = p, Fp, True
            else: ph, Fh, hreal = p, Fp, True
            it += 1
        alo = int(col[lo])
        gap = lo - F0
        if abs(gap) > 1:
            obs = (alo - p0) / gap
            if 1 <= obs < 1e8: st["kpr"] = 0.5 * kpr + 0.5 * obs
        st["pc"] = pl if (0 <= lo - Fl <= slack) else ph
        return cnt
    return strat
