#!/usr/bin/env python3
"""The issue floor of the sorted-list kernel: the vector instructions of a wave-row by issue class, priced with the
per-instruction costs measured on MI355X (tools/ubench_valu2.hip; profiles/r6_ubench_valu2.txt if present, else the round-1
table: cycles per wave64 instruction and SIMD):

    class A  2.75 cycles at >= 2 waves per SIMD (5.7 for a wave alone)   v_add / v_sub / v_and / v_or / v_xor / v_mov / v_lshrrev /
                                                                         v_ashrrev / v_add_f32 / v_mul_f32 / v_fmac / v_bitop3
    class B  4.4 cycles at >= 2 waves per SIMD (5.7 for a wave alone)    v_min / v_max / v_med3 / v_cmp / v_addc / v_subb / v_mad /
                                                                         v_lshl_add / v_bfe / v_xad / VOP3-only integer,
                                                                         every float64 instruction, conversions, DPP forms

How: clim_sorted_f32<20, 16, 14, false> is compiled to ISA (hipcc -S, no GPU needed); its hot basic blocks are found by their
signatures (key conversion: v_cvt_f64_f32; requests: global_load + 64-bit adds; sort: comparators + ds_write; bookkeeping +
direction set-up: the blocks between the sort and the first select round; the select round: >= 20 ds_read) and give the
CLASS SHARES of a wave-row.  The instruction COUNT is the live one (--measured-valu: SQ_INSTS_VALU per wave-row of the
product kernel on this run's box); the select rounds per wave-row follow from it -- (measured - the blocks that run once)
/ the round's size -- so nothing is read from a tracked profile of another round.
Output: JSON on stdout (bench.py embeds it), a table on stderr.
Usage: python tools/issue_mix.py [--measured-valu 1020] [--measured-quad-cycles 2380] [--waves-per-cu 8]
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASS_A = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_mov_b64",
           "v_lshrrev_b32", "v_ashrrev_i32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fmac_f32", "v_not_b32",
           "v_accvgpr", "v_readlane", "v_writelane", "v_readfirstlane", "v_nop",
           "v_bitop3_b32")     # (v_bitop3: 3.4 cycles at two waves per SIMD -- profiles/r6_ubench_valu3.txt)
COST = {"A": (2.75, 5.7), "B": (4.4, 5.7)}


def klass(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return "B"
    return "A" if any(base.startswith(a) for a in CLASS_A) else "B"


def class_costs():
    """(cost at >= 2 waves per SIMD, cost of a wave alone) per class, from this round's microbenchmark table if it is there"""
    cost = {"A": [2.75, 5.7], "B": [4.4, 5.7]}
    src = "profiles/r1_ubench_valu2.txt"
    for name in ("r6_ubench_valu2.txt",):
        f = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(f):
            continue
        rows = {}
        for line in open(f):
            t = line.split()
            if len(t) == 4 and t[0].startswith("k_"):
                try:
                    rows[t[0]] = [float(x) for x in t[1:]]
                except ValueError:
                    pass
        a_ops = [rows[k] for k in ("k_sub_u32", "k_and_b32", "k_xor_b32", "k_mov_b32", "k_ashrrev_i32") if k in rows]
        b_ops = [rows[k] for k in ("k_max_u32", "k_med3_u32", "k_cmp_only", "k_addc_only", "k_bfe_u32", "k_add_f64") if k in rows]
        if a_ops and b_ops:
            # (columns: 1, 2, 4 waves per SIMD)
            cost["A"] = [sum(r[1] for r in a_ops) / len(a_ops), sum(r[0] for r in a_ops) / len(a_ops)]
            cost["B"] = [sum(r[1] for r in b_ops) / len(b_ops), sum(r[0] for r in b_ops) / len(b_ops)]
            src = "profiles/" + name + " (2 waves per SIMD; one wave alone)"
    return cost, src


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--measured-valu", type=float, default=0.0, help="SQ_INSTS_VALU per wave-row of the product kernel")
    ap.add_argument("--waves-per-cu", type=int, default=8, help="waves of this kernel a CU holds (LDS, registers): 4..8")
    ap.add_argument("--measured-quad-cycles", type=float, default=0.0, help="SQ_WAVE_CYCLES per wave-row (quad-cycles)")
    ap.add_argument("--rounds", type=float, default=0.0, help="select rounds per wave-row (0: solved from --measured-valu)")
    args = ap.parse_args()
    COST, cost_src = class_costs()
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-fno-fast-math", "--cuda-device-only", "-DXMHW_SORTED_ONLY", "-S", "-o", out,
                               os.path.join(ROOT, "xmhw_amd", "csrc", "kernels_sorted.hip")], stderr=subprocess.DEVNULL)
        txt = open(out).read()
    sym = "_ZN4xmhw15clim_sorted_f32ILi20ELi16ELi14ELb0EEE"
    i = txt.index(sym)
    i = txt.index(sym, i + 10)
    body = txt[i:txt.index(".Lfunc_end", i)]
    blocks, cur = [], ["entry", []]
    blocks.append(cur)
    for line in body.split("\n"):
        m = re.match(r"^(\.LBB\d+_\d+):", line) or re.match(r"^; %bb\.(\d+):", line)
        if m:
            cur = [m.group(0).strip(":; "), []]
            blocks.append(cur)
        elif re.match(r"^\s+(v_|s_|ds_|global_|buffer_|scratch_)", line):
            cur[1].append(line.split()[0])

    def count(ins, pat):
        return sum(1 for o in ins if o.startswith(pat))
    tagged = {}
    order = []
    for bi, (name, ins) in enumerate(blocks):
        v = count(ins, "v_")
        if v < 15:
            continue
        tag = None
        if count(ins, "v_cvt_f64_f32") >= 15 and "conv" not in tagged:
            tag = "conv"                                   # (copies exist -- heat waves / cold spells, NaN rows; a row runs one)
        elif count(ins, "global_load_dword") >= 15 and count(ins, "v_mad_u64_u32") == 0 and count(ins, "v_cvt_f64") == 0 and "requests" not in tagged:
            tag = "requests"
        elif count(ins, "v_max_u32") + count(ins, "v_min_u32") > 100 and count(ins, "ds_read") == 0 and "sort" not in tagged:
            tag = "sort"
        elif count(ins, "ds_read") >= 20 and "select round" not in tagged:
            tag = "select round"
        elif "sort" in tagged and "select round" not in tagged and count(ins, "ds_read") < 20:
            tag = "book+direction %d" % (len([t for t in tagged if t.startswith("book")]) + 1)
        elif count(ins, "v_div_fmas_f64") >= 1 and "epilogue" not in tagged:
            tag = "epilogue"
        if tag:
            tagged[tag] = (name, ins, 0.5 if tag == "epilogue" else 1.0)
            order.append(tag)

    def classes(ins):
        c = collections.Counter()
        for o in ins:
            if o.startswith("v_"):
                c[klass(o)] += 1
        return c
    once = collections.Counter()
    for tag in order:
        if tag != "select round":
            c = classes(tagged[tag][1])
            w = tagged[tag][2]
            once["A"] += w * c["A"]
            once["B"] += w * c["B"]
    rc = classes(tagged["select round"][1])
    round_valu = rc["A"] + rc["B"]
    once_valu = once["A"] + once["B"]
    rounds = args.rounds or (max(args.measured_valu - once_valu, 0.0) / round_valu if args.measured_valu else 1.05)
    tot = {"A": once["A"] + rounds * rc["A"], "B": once["B"] + rounds * rc["B"]}
    static_valu = tot["A"] + tot["B"]
    fa, fb = tot["A"] / static_valu, tot["B"] / static_valu
    valu = args.measured_valu or static_valu
    floor2 = valu * (fa * COST["A"][0] + fb * COST["B"][0])
    floor1 = valu * (fa * COST["A"][1] + fb * COST["B"][1])
    per_block = {}
    for tag in order:
        name, ins, wgt = tagged[tag]
        c = classes(ins)
        per_block[tag] = {"block": name, "weight": rounds if tag == "select round" else wgt, "valu_A": c["A"], "valu_B": c["B"],
                          "salu": count(ins, "s_") - count(ins, "s_waitcnt") - count(ins, "s_nop"), "s_nop": count(ins, "s_nop"),
                          "lds": count(ins, "ds_"), "vmem": count(ins, "global_")}
    two = max(args.waves_per_cu - 4, 0)
    res = {"kernel": "clim_sorted_f32<20, 16, 14, false>", "blocks": per_block,
           "valu_of_blocks_that_run_once_per_row": once_valu, "valu_of_a_select_round": round_valu,
           "select_round_equivalents_per_wave_row": rounds,
           "select_round_equivalents_source": "--rounds" if args.rounds else ("(live SQ_INSTS_VALU per wave-row - the blocks that run once per row) / the select round: further rounds, the key-by-key finish, the corrections and the chunks' first rows all count here" if args.measured_valu else "assumed"),
           "measured_valu_per_wave_row": args.measured_valu or None,
           "class_A_share": fa, "class_B_share": fb,
           "cost_cycles": {"A": COST["A"], "B": COST["B"], "source": cost_src},
           "issue_floor_cycles_per_wave_row": {"two_or_more_waves_per_simd": floor2, "one_wave_per_simd": floor1,
                                               "this_kernel_occupancy": (two * floor2 + (4 - two) * floor1) / 4.0}}
    if args.measured_quad_cycles:
        cyc = 4.0 * args.measured_quad_cycles           # cycles a wave spends on one of its rows
        res["measured_wave_cycles_per_wave_row"] = cyc
        # `two` SIMDs of a CU run two waves (a wave-row costs the SIMD floor2 cycles there), the others a wave alone (floor1).
        # Wave-rows a CU could retire per cycle at the floor, against what it retires:
        at_floor = two * (1.0 / floor2) + (4 - two) * (1.0 / floor1)
        measured = args.waves_per_cu / cyc
        res["frac_of_issue_floor_at_this_occupancy"] = measured / at_floor
        res["frac_of_issue_floor_at_two_waves_per_simd"] = measured / (4.0 / floor2)
        res["note"] = ("wave-rows per CU-cycle measured (waves per CU / wave cycles per row) over wave-rows per CU-cycle if every SIMD issued "
                       "a vector instruction whenever the class costs allow, at this kernel's occupancy")
    json.dump(res, sys.stdout, indent=1)
    print(file=sys.stdout)
    for tag, b in per_block.items():
        print(f"{tag:20s} {b['block']:12s} x{b['weight']:.2f}  VALU A {b['valu_A']:4d} B {b['valu_B']:4d}  SALU {b['salu']:4d} (s_nop {b['s_nop']})  LDS {b['lds']:3d}  VMEM {b['vmem']:3d}",
              file=sys.stderr)
    print(f"VALU per wave-row {valu:.0f} (blocks that run once {once_valu:.0f} + {rounds:.2f} rounds x {round_valu}; class A {100 * fa:.1f} %, "
          f"class B {100 * fb:.1f} %); floor at >= 2 waves per SIMD {floor2:.0f} cycles, one wave alone {floor1:.0f}", file=sys.stderr)


if __name__ == "__main__":
    main()
