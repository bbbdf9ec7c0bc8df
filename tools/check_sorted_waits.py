#!/usr/bin/env python3
"""No `s_waitcnt vmcnt` may sit inside a comparator block of any clim_sorted_* kernel.

The row loop of the sorted-list kernel requests the NEXT row's 20 samples right after it has converted this row's, and needs
them a whole row later.  The only vmcnt waits the loop needs are the conversion's (and those of two cold blocks that load
themselves).  The compiler's waitcnt pass can add one in front of the sort -- round 6: a register that a cold path loads into
and the hot path keeps a key in -- and that wait sits out the requests just issued: same instructions, 12 % slower
(profiles/r6_experiments.txt).  This script compiles kernels_sorted.hip to ISA (hipcc -S, no GPU) and looks at every
instantiation: a basic block with >= 40 comparator instructions (v_min / v_max / v_med3) and no global load of its own must
hold no vmcnt wait.  Exit status 1 and a list if one does.  tests/test_kernel_isa.py runs it.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_CACHE = {}


def kernels(extra_flags=()):
    """[(mangled name, [(block label, [instructions])])] of every clim_sorted_* kernel, compiled once per flag set"""
    key = tuple(extra_flags)
    if key in _CACHE:
        return _CACHE[key]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-fno-fast-math", "--cuda-device-only", "-S", "-o", out, *extra_flags,
                               os.path.join(ROOT, "xmhw_amd", "csrc", "kernels_sorted.hip")], stderr=subprocess.DEVNULL)
        txt = open(out).read()
    funcs = re.split(r"\n(?=_ZN4xmhw15clim_sorted_[a-z0-9]+I[^\n]*:\s*;\s*@)", txt)[1:]
    res = []
    for f in funcs:
        name = f.split(":")[0]
        body = f[:f.index(".Lfunc_end")] if ".Lfunc_end" in f else f
        blocks, cur = [], ["entry", []]
        blocks.append(cur)
        for line in body.split("\n"):
            m = re.match(r"^(\.LBB\d+_\d+):", line) or re.match(r"^; %bb\.(\d+):", line)
            if m:
                cur = [m.group(0).strip(":; "), []]
                blocks.append(cur)
            elif re.match(r"^\s+(v_|s_|ds_|global_|scratch_)", line):
                cur[1].append(line.strip())
        res.append((name, blocks))
    _CACHE[key] = res
    return res


def check_branches(extra_flags=()):
    """Divergent mini-branches.  Round 6: `move ? (grow ? P + c : P - c) : P` had become 36 branches per select round, each an
    s_and_saveexec + s_cbranch_execz + s_or around ONE vector instruction -- 4.8 % of the kernel with hardly an instruction in
    them (profiles/r6_experiments.txt).  A kernel fails if a select round (a block with >= 20 LDS reads) has more than two
    exec-mask branches in the blocks right behind it, or more than 24 in all (15 today: the epilogue's, the cold paths')."""
    bad = []
    for name, blocks in kernels(extra_flags):
        total = sum(1 for _, ins in blocks for o in ins if o.startswith("s_cbranch_exec"))
        near = 0
        for i, (_, ins) in enumerate(blocks):
            if sum(1 for o in ins if o.startswith("ds_read")) >= 20:
                near = max(near, sum(1 for _, i2 in blocks[i:i + 14] for o in i2 if o.startswith("s_cbranch_exec")))
        if near > 2 or total > 24:
            bad.append((name, near, total))
    return len(kernels(extra_flags)), bad


def check(extra_flags=()):
    bad = []
    funcs = kernels(extra_flags)
    for name, blocks in funcs:
        for nm, ins in blocks:
            ncmp = sum(1 for o in ins if o.startswith(("v_max_u32", "v_min_u32", "v_med3_u32")))
            if ncmp >= 40 and not any(o.startswith("global_load") for o in ins):
                w = [o for o in ins if o.startswith("s_waitcnt") and "vmcnt" in o]
                if w:
                    bad.append((name, nm, w))
    return len(funcs), bad


if __name__ == "__main__":
    n, bad = check(sys.argv[1:])
    for name, nm, w in bad:
        print(f"{name}: block {nm}: {w}")
    print(f"{n} clim_sorted_* kernels checked, {len(bad)} comparator blocks with a vmcnt wait")
    _, badb = check_branches(sys.argv[1:])
    for name, near, total in badb:
        print(f"{name}: {near} exec-mask branches behind a select round, {total} in all")
    print(f"{len(badb)} kernels with divergent mini-branches in or behind a select round")
    sys.exit(1 if bad or badb or n == 0 else 0)
