"""Generates xmhw_amd/csrc/sortnet_gen.h: straight-line comparator networks for keys held in registers (static
indices only), used by kernels_sorted.hip.  A comparator (i, j) leaves max at i and min at j ("descending").

  Desc<N>            sorts N keys descending.  Built as: two halves sorted recursively + Batcher's odd-even merge of the
                     two sorted halves (for N <= 8 Batcher's odd-even merge sort pruned to N inputs, whichever is
                     shorter).
  MergeTop<A, B, K>  v[0..A) and v[A..A+B) are sorted descending on entry; on exit v[0..K) holds the K largest of the
                     A + B keys, descending (the other entries are unspecified).  Batcher's odd-even merge of the two
                     runs padded with -inf to a power of two; the pads are tracked statically (a comparator that meets a
                     pad is a renaming, not an instruction) and comparators that cannot reach outputs 0..K-1 are dropped.
  BitonicDesc<N>     sorts a BITONIC sequence of N keys (cyclic rotation of rise-then-fall) descending: half-cleaners
                     while N is even, a full network on the odd remainders.

  Ins<N>             (round 6) sorts N keys descending by INSERTION with three-input instructions: a 3-sorter is v_max3 /
                     v_med3 / v_min3 (3 instructions where 3 comparators are 6), and inserting a key x into a sorted run
                     a[0..n) is n + 1 independent instructions: max(a0, x), med3(a[i-1], a[i], x), min(a[n-1], x).
                     N keys cost 3 + 4 + ... + N = N (N + 1) / 2 - 3 instructions: fewer than the comparator network's
                     2 x comparators up to N = 9 (5 keys: 12 against 18) -- used for the halves of the 10-key sorts
                     (two Ins<5> + MergeTop<5,5,10> = 50 instructions against 58).  min / med / max commute with every
                     monotone map, so the 0-1 principle holds for these circuits too; all 0-1 inputs AND all
                     permutations (N <= 7) are checked.

Every emitted network is verified before it is written: Desc and BitonicDesc with the 0-1 principle over all inputs of
their class, MergeTop over all pairs of sorted 0-1 runs and on random keys with ties.
Usage: python tools/gen_sortnet.py            (writes the header; prints the comparator counts)
"""
import itertools
import os
import numpy as np

NMAX = 24
MERGES = [(4, 4, 8), (8, 8, 12), (12, 8, 12), (8, 8, 16), (16, 8, 24), (10, 10, 20), (5, 5, 10), (6, 6, 12), (12, 12, 24),
          (8, 8, 8), (8, 4, 12), (4, 4, 4), (8, 4, 8), (11, 11, 22), (12, 12, 12), (10, 10, 16), (10, 10, 14), (10, 10, 18),
          (8, 8, 14), (5, 5, 8), (5, 5, 6), (6, 6, 8), (6, 6, 10), (7, 7, 14), (7, 7, 12), (9, 9, 16), (9, 9, 18), (11, 11, 16),
          (12, 12, 16), (12, 12, 18), (12, 12, 20), (16, 8, 16), (16, 8, 20), (8, 8, 10), (10, 8, 10), (16, 16, 16),
          (3, 3, 6), (6, 6, 10), (9, 9, 14), (11, 11, 18), (12, 12, 18)]
BITONIC = [3, 4, 5, 6, 7, 8, 9, 10, 11, 12]


def batcher_sort(n):
    """odd-even merge sort on p = next pow2 >= n, pruned to indices < n (missing inputs = -inf at the top indices)"""
    p = 1
    while p < n:
        p *= 2
    net = []

    def merge(lo, hi, r):
        step = r * 2
        if step < hi - lo:
            merge(lo, hi, step)
            merge(lo + r, hi, step)
            for i in range(lo + r, hi - r, step):
                net.append((i, i + r))
        else:
            net.append((lo, lo + r))

    def sort(lo, hi):
        if hi - lo >= 1:
            mid = lo + (hi - lo) // 2
            sort(lo, mid)
            sort(mid + 1, hi)
            merge(lo, hi, 1)

    sort(0, p - 1)
    return [(i, j) for (i, j) in net if j < n]


def batcher_merge_pow2(p):
    """odd-even merge of two sorted runs of p keys each at [0, p) and [p, 2p)"""
    net = []

    def merge(lo, hi, r):
        step = r * 2
        if step < hi - lo:
            merge(lo, hi, step)
            merge(lo + r, hi, step)
            for i in range(lo + r, hi - r, step):
                net.append((i, i + r))
        else:
            net.append((lo, lo + r))

    merge(0, 2 * p - 1, 1)
    return net


def merge_top(a, b, keep):
    """comparators on an array v[0..a+b): runs v[0..a), v[a..a+b) -> top `keep` at v[0..keep).  Returns (net, perm):
    net over PHYSICAL slots 0..a+b-1, perm[k] = physical slot that holds output k after the network."""
    p = 1
    while p < max(a, b):
        p *= 2
    raw = batcher_merge_pow2(p)
    # logical index -> physical slot (None = pad)
    slot = [None] * (2 * p)
    for i in range(a):
        slot[i] = i
    for i in range(b):
        slot[p + i] = a + i
    net = []
    for (i, j) in raw:
        si, sj = slot[i], slot[j]
        if si is None and sj is None:
            continue
        if si is None:            # pad above a key: the key moves up (a renaming)
            slot[i], slot[j] = sj, None
            continue
        if sj is None:
            continue
        net.append((si, sj))
    perm = [slot[k] for k in range(a + b)]
    assert all(s is not None for s in perm)
    # drop comparators that cannot reach the kept outputs
    live = set(perm[:keep])
    out = []
    for (i, j) in reversed(net):
        if i in live or j in live:
            out.append((i, j))
            live.add(i)
            live.add(j)
    out.reverse()
    return out, perm[:keep]


def apply_net(net, cols):
    for (i, j) in net:
        hi = np.maximum(cols[i], cols[j])
        lo = np.minimum(cols[i], cols[j])
        cols[i], cols[j] = hi, lo
    return cols


def verify_sort(n, net):
    total = 1 << n
    chunk = 1 << 20
    for base in range(0, total, chunk):
        x = np.arange(base, min(base + chunk, total), dtype=np.int64)
        bits = [((x >> k) & 1).astype(np.int8) for k in range(n)]
        bits = apply_net(net, bits)
        for k in range(n - 1):
            if np.any(bits[k] < bits[k + 1]):
                return False
    return True


def verify_merge(a, b, keep, net, perm):
    rng = np.random.default_rng(a * 100 + b)
    trials = 4000
    for rng_hi in (3, 1000):      # heavy ties, few ties
        x = np.sort(rng.integers(0, rng_hi, size=(trials, a)), axis=1)[:, ::-1]
        y = np.sort(rng.integers(0, rng_hi, size=(trials, b)), axis=1)[:, ::-1]
        cols = [x[:, i].copy() for i in range(a)] + [y[:, i].copy() for i in range(b)]
        cols = apply_net(net, cols)
        want = np.sort(np.concatenate([x, y], axis=1), axis=1)[:, ::-1][:, :keep]
        got = np.stack([cols[perm[k]] for k in range(keep)], axis=1)
        if not np.array_equal(got, want):
            return False
    # all pairs of sorted 0-1 runs
    for na in range(a + 1):
        for nb in range(b + 1):
            cols = [np.array([1 if i < na else 0]) for i in range(a)] + [np.array([1 if i < nb else 0]) for i in range(b)]
            cols = apply_net(net, cols)
            for k in range(keep):
                if int(cols[perm[k]][0]) != (1 if k < na + nb else 0):
                    return False
    return True


# networks from the literature that beat the constructions below (verified like every other network before use)
KNOWN = {
    10: [(0, 8), (1, 9), (2, 7), (3, 5), (4, 6), (0, 2), (1, 4), (5, 8), (7, 9), (0, 3), (2, 4), (5, 7), (6, 9), (0, 1), (3, 6),
         (8, 9), (1, 5), (2, 3), (4, 8), (6, 7), (1, 2), (3, 5), (4, 6), (7, 8), (2, 3), (4, 5), (6, 7), (3, 4), (5, 6)],      # 29, depth 8
}

_memo = {}


def compose(n):
    """comparator list + output permutation (output k is left in slot perm[k]) of the shortest Desc<n> found: Batcher's
    pruned odd-even merge sort, or two sorted halves (recursively) + the odd-even merge of the two runs"""
    if n in _memo:
        return _memo[n]
    if n == 1:
        _memo[n] = ([], [0])
        return _memo[n]
    best = (batcher_sort(n), list(range(n)))
    if n in KNOWN and len(KNOWN[n]) < len(best[0]):
        best = (list(KNOWN[n]), list(range(n)))
    if n >= 4:
        h = n // 2
        ln, lp = compose(h)
        rn, rp = compose(n - h)
        rn = [(i + h, j + h) for (i, j) in rn]
        rp = [q + h for q in rp]
        # after the halves: run A's element k sits in slot lp[k], run B's element k in slot rp[k]
        mnet, mperm = merge_top(h, n - h, n)
        phys = lp + rp                      # merge-network slot index -> physical slot
        mnet = [(phys[i], phys[j]) for (i, j) in mnet]
        mperm = [phys[q] for q in mperm]
        cand = (ln + rn + mnet, mperm)
        if len(cand[0]) < len(best[0]):
            best = cand
    _memo[n] = best
    return best


def bitonic_desc(n):
    """network + output permutation sorting a bitonic sequence of n keys descending"""
    if n <= 1:
        return [], list(range(n))
    if n % 2 == 1:
        net, perm = compose(n)
        return net, perm
    h = n // 2
    net = [(i, i + h) for i in range(h)]
    un, up = bitonic_desc(h)
    ln, lp = bitonic_desc(h)
    net += un + [(i + h, j + h) for (i, j) in ln]
    return net, up + [q + h for q in lp]


def verify_bitonic(n, net, perm):
    # all 0-1 bitonic sequences: cyclic rotations of 1^a 0^(n-a)
    for a in range(n + 1):
        base = [1] * a + [0] * (n - a)
        for r in range(n):
            seq = base[r:] + base[:r]
            cols = [np.array([v]) for v in seq]
            cols = apply_net(net, cols)
            for k in range(n):
                if int(cols[perm[k]][0]) != (1 if k < a else 0):
                    return False
    rng = np.random.default_rng(n)
    for _ in range(300):
        up = np.sort(rng.integers(0, 50, size=rng.integers(0, n + 1)))
        dn = np.sort(rng.integers(0, 50, size=n - up.size))[::-1]
        seq = np.concatenate([up, dn])
        seq = np.roll(seq, int(rng.integers(0, n)))
        cols = [np.array([v]) for v in seq]
        cols = apply_net(net, cols)
        got = [int(cols[perm[k]][0]) for k in range(n)]
        if got != sorted(seq.tolist(), reverse=True):
            return False
    return True


def depth(net):
    d = {}
    m = 0
    for (i, j) in net:
        t = max(d.get(i, 0), d.get(j, 0)) + 1
        d[i] = d[j] = t
        m = max(m, t)
    return m


def emit(out, name, nslots, net, perm, nout):
    """a function that runs the comparators on v[] and then moves output k from slot perm[k] to v[k] (a register
    renaming for the compiler: the moves cost nothing once everything is in registers)"""
    out.append(f"    static __device__ __forceinline__ void run(uint32_t (&v)[{nslots}]) {{")
    line = "       "
    for (i, j) in net:
        tok = f" XMHW_CE({i}, {j})"
        if len(line) + len(tok) > 118:
            out.append(line)
            line = "       "
        line += tok
    if line.strip():
        out.append(line)
    if perm is not None and list(perm) != list(range(nout)):
        out.append(f"        const uint32_t r_[{nout}] = {{" + ", ".join(f"v[{q}]" for q in perm) + "};")
        out.append(f"#pragma unroll")
        out.append(f"        for (int i = 0; i < {nout}; ++i) v[i] = r_[i];")
    out.append("    }")


INSERTION = [3, 4, 5, 6, 7]


def insertion_program(n):
    """[(op, dst, srcs)]: a 3-sorter, then n - 3 insertions; registers are numbered, inputs 0..n-1"""
    prog = [("max3", n, (0, 1, 2)), ("med3", n + 1, (0, 1, 2)), ("min3", n + 2, (0, 1, 2))]
    srt = [n, n + 1, n + 2]
    nxt = n + 3
    for k in range(3, n):
        new = []
        prog.append(("max", nxt, (srt[0], k))); new.append(nxt); nxt += 1
        for i in range(1, k):
            prog.append(("med3", nxt, (srt[i - 1], srt[i], k))); new.append(nxt); nxt += 1
        prog.append(("min", nxt, (srt[k - 1], k))); new.append(nxt); nxt += 1
        srt = new
    prog.append(("out", None, tuple(srt)))
    return prog[:-1] + [prog[-1]]


def verify_insertion(n, prog):
    ops = {"max3": max, "min3": min, "max": max, "min": min, "med3": lambda *a: sorted(a)[1]}
    def run(inp):
        r = dict(enumerate(inp))
        for op, dst, src in prog:
            if op == "out":
                return [r[i] for i in src]
            r[dst] = ops[op](*[r[i] for i in src])
    for bits in itertools.product((0, 1), repeat=n):
        if run(bits) != sorted(bits, reverse=True):
            return False
    for perm in itertools.permutations(range(n)):
        if run(perm) != sorted(perm, reverse=True):
            return False
    rng = np.random.default_rng(n)
    for _ in range(2000):                    # (ties)
        x = list(rng.integers(0, 3, n))
        if run(x) != sorted(x, reverse=True):
            return False
    return True


def main():
    out = []
    out.append("// sortnet_gen.h -- GENERATED by tools/gen_sortnet.py; do not edit.")
    out.append("// Straight-line comparator networks, descending, for keys held in registers (static indices only).")
    out.append("// Every network below was verified when it was generated (tools/gen_sortnet.py: 0-1 principle / all pairs of")
    out.append("// sorted 0-1 runs / random keys with ties).")
    out.append("#pragma once")
    out.append("#include <stdint.h>")
    out.append("namespace xmhw {")
    out.append("namespace sortnet {")
    out.append("#define XMHW_CE(i, j) { const uint32_t a_ = v[i], b_ = v[j]; v[i] = a_ > b_ ? a_ : b_; v[j] = a_ > b_ ? b_ : a_; }")
    out.append("template <int N> struct Desc;")
    out.append("template <> struct Desc<1> { static __device__ __forceinline__ void run(uint32_t (&)[1]) {} };")
    report = []
    for n in range(2, NMAX + 1):
        net, perm = compose(n)
        # verify through the permutation
        ok = True
        if n <= 22:
            total = 1 << n
            for base in range(0, total, 1 << 20):
                x = np.arange(base, min(base + (1 << 20), total), dtype=np.int64)
                bits = [((x >> k) & 1).astype(np.int8) for k in range(n)]
                bits = apply_net(net, bits)
                for k in range(n - 1):
                    if np.any(bits[perm[k]] < bits[perm[k + 1]]):
                        ok = False
        else:
            rng = np.random.default_rng(n)
            x = rng.integers(0, 2, size=(1 << 21, n)).astype(np.int8)
            bits = apply_net(net, [x[:, k].copy() for k in range(n)])
            for k in range(n - 1):
                if np.any(bits[perm[k]] < bits[perm[k + 1]]):
                    ok = False
        assert ok, n
        report.append(f"Desc<{n}>: {len(net)} comparators, depth {depth(net)}")
        out.append(f"// n = {n}: {len(net)} comparators, depth {depth(net)}")
        out.append(f"template <> struct Desc<{n}> {{")
        emit(out, "Desc", n, net, perm, n)
        out.append("};")
    out.append("// v[0..A) and v[A..A+B) sorted descending -> v[0..K) = the K largest, descending")
    out.append("template <int A, int B, int K> struct MergeTop;")
    for (a, b, keep) in sorted(set(MERGES)):
        net, perm = merge_top(a, b, keep)
        assert verify_merge(a, b, keep, net, perm), (a, b, keep)
        report.append(f"MergeTop<{a},{b},{keep}>: {len(net)} comparators, depth {depth(net)}")
        out.append(f"// {a} + {b} -> top {keep}: {len(net)} comparators, depth {depth(net)}")
        out.append(f"template <> struct MergeTop<{a}, {b}, {keep}> {{")
        emit(out, "MergeTop", a + b, net, perm, keep)
        out.append("};")
    out.append("// a bitonic sequence of N keys -> descending")
    out.append("template <int N> struct BitonicDesc;")
    for n in BITONIC:
        net, perm = bitonic_desc(n)
        assert verify_bitonic(n, net, perm), n
        report.append(f"BitonicDesc<{n}>: {len(net)} comparators, depth {depth(net)}")
        out.append(f"// n = {n}: {len(net)} comparators, depth {depth(net)}")
        out.append(f"template <> struct BitonicDesc<{n}> {{")
        emit(out, "BitonicDesc", n, net, perm, n)
        out.append("};")
    # ---- insertion sorters on three-input instructions --------------------------------------------------------
    out.append("// N keys -> descending, by insertion with three-input instructions (v_max3 / v_med3 / v_min3): see tools/gen_sortnet.py")
    out.append("__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {")
    out.append("    uint32_t r;")
    out.append('    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));')
    out.append("    return r;")
    out.append("}")
    out.append("__device__ __forceinline__ uint32_t max_u32(uint32_t a, uint32_t b) { return a > b ? a : b; }")
    out.append("__device__ __forceinline__ uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }")
    out.append("template <int N> struct Ins;")
    for n in INSERTION:
        prog = insertion_program(n)
        assert verify_insertion(n, prog), n
        report.append(f"Ins<{n}>: {len(prog) - 1} three-input instructions")
        out.append(f"// n = {n}: {len(prog) - 1} instructions")
        out.append(f"template <> struct Ins<{n}> {{")
        out.append(f"    static __device__ __forceinline__ void run(uint32_t (&v)[{n}]) {{")
        tmp = 0
        cur = [f"v[{i}]" for i in range(n)]
        # 3-sorter
        out.append(f"        uint32_t s0 = max_u32(max_u32({cur[0]}, {cur[1]}), {cur[2]}), s1 = med3_u32({cur[0]}, {cur[1]}, {cur[2]}), "
                   f"s2 = min_u32(min_u32({cur[0]}, {cur[1]}), {cur[2]});")
        srt = ["s0", "s1", "s2"]
        for k in range(3, n):
            x = cur[k]
            new = [f"t{k}_{i}" for i in range(k + 1)]
            parts = [f"{new[0]} = max_u32({srt[0]}, {x})"]
            for i in range(1, k):
                parts.append(f"{new[i]} = med3_u32({srt[i - 1]}, {srt[i]}, {x})")
            parts.append(f"{new[k]} = min_u32({srt[k - 1]}, {x})")
            out.append("        const uint32_t " + ", ".join(parts) + ";")
            srt = new
        out.append("        " + " ".join(f"v[{i}] = {srt[i]};" for i in range(n)))
        out.append("    }")
        out.append("};")
    out.append("#undef XMHW_CE")
    out.append("}  // namespace sortnet")
    out.append("}  // namespace xmhw")
    path = os.path.join(os.path.dirname(__file__), "..", "xmhw_amd", "csrc", "sortnet_gen.h")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()
