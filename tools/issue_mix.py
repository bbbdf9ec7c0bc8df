#!/usr/bin/env python3
"""The issue floor of the sorted-list kernel: the vector instructions of a wave-row by issue class, priced
with the per-instruction costs measured on MI355X (profiles/r1_ubench_valu2.txt, cycles per wave64 instruction and SIMD):

    class A  2.75 cycles at >= 2 waves per SIMD (5.7 for a wave alone)   v_add / v_sub / v_and / v_or / v_xor / v_mov / v_lshrrev /
                                                                         v_ashrrev / v_add_f32 / v_mul_f32 / v_fmac / v_cndmask
    class B  4.4 cycles at >= 2 waves per SIMD (5.7 for a wave alone)    v_min / v_max / v_med3 / v_cmp / v_addc / v_subb / v_mad /
                                                                         v_lshl_add / v_bfe / v_bitop3 / v_perm / VOP3-only integer,
                                                                         every float64 instruction, conversions, DPP forms

How: the kernel clim_sorted_f32<20, 16, false> is compiled to ISA (hipcc -S, no GPU needed), its hot basic blocks are found
by their signatures (key conversion: v_bitop3; sort + bookkeeping: ds_write + comparators; the select round: >= 20 ds_read
+ v_med3_i32; direction set-up; epilogue: float64 division), each block weighted by how often a wave-row runs it (the select
round `--rounds` times: the counter twin's measurement, profiles/r5_ticks.jsonl; the epilogue every second row).
Output: JSON on stdout (bench.py reads it back through profiles/r5_issue_mix.json), a table on stderr.
Usage: python tools/issue_mix.py [--rounds 1.26] [--measured-valu 1127] > profiles/r5_issue_mix.json
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
           "v_lshrrev_b32", "v_ashrrev_i32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fmac_f32", "v_cndmask_b32", "v_not_b32",
           "v_accvgpr", "v_readlane", "v_writelane", "v_readfirstlane", "v_nop")
COST = {"A": (2.75, 5.7), "B": (4.4, 5.7)}


def klass(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return "B"
    return "A" if any(base.startswith(a) for a in CLASS_A) else "B"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=float, default=1.26)
    ap.add_argument("--measured-valu", type=float, default=0.0, help="SQ_INSTS_VALU per wave-row of the product kernel")
    ap.add_argument("--waves-per-cu", type=int, default=7, help="waves of this kernel a CU holds (LDS): 4..8")
    ap.add_argument("--measured-quad-cycles", type=float, default=0.0, help="SQ_WAVE_CYCLES per wave-row (quad-cycles)")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-fno-fast-math", "--cuda-device-only", "-S", "-o", out,
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
        elif re.match(r"^\s+(v_|s_|ds_|global_|buffer_)", line):
            cur[1].append(line.split()[0])
    def count(ins, pat):
        return sum(1 for o in ins if o.startswith(pat))
    tagged = {}
    for name, ins in blocks:
        v = count(ins, "v_")
        if v < 20:
            continue
        if count(ins, "v_bitop3") >= 15 and "conv" not in tagged:
            tagged["conv"] = (name, ins, 1.0)            # (two copies exist -- heat waves / cold spells; a row runs one)
        elif count(ins, "ds_write") >= 3 and count(ins, "v_max_u32") + count(ins, "v_min_u32") > 100:
            tagged["sort+book"] = (name, ins, 1.0)
        elif count(ins, "ds_read") >= 20 and count(ins, "v_bfe_i32") + count(ins, "v_med3_i32") >= 10:
            tagged["select round"] = (name, ins, args.rounds)
        elif count(ins, "ds_read") >= 5 and count(ins, "ds_read") < 12 and count(ins, "v_min") >= 1 and "direction" not in tagged and v < 80:
            tagged["direction"] = (name, ins, 1.0)
        elif count(ins, "v_div_fmas_f64") >= 1:
            tagged["epilogue"] = (name, ins, 0.5)
        elif count(ins, "global_load_dword") >= 15 and count(ins, "v_mad_u64_u32") == 0 and "loads" not in tagged:
            tagged["loads"] = (name, ins, 1.0)
    tot = collections.Counter()
    per_block = {}
    for tag, (name, ins, wgt) in tagged.items():
        c = collections.Counter()
        for o in ins:
            if o.startswith("v_"):
                c[klass(o)] += 1
        per_block[tag] = {"block": name, "weight": wgt, "valu_A": c["A"], "valu_B": c["B"],
                          "salu": count(ins, "s_") - count(ins, "s_waitcnt") - count(ins, "s_nop"), "s_nop": count(ins, "s_nop"),
                          "lds": count(ins, "ds_"), "vmem": count(ins, "global_")}
        tot["A"] += wgt * c["A"]
        tot["B"] += wgt * c["B"]
    static_valu = tot["A"] + tot["B"]
    scale = args.measured_valu / static_valu if args.measured_valu else 1.0
    fa, fb = tot["A"] / static_valu, tot["B"] / static_valu
    valu = args.measured_valu or static_valu
    floor2 = valu * (fa * COST["A"][0] + fb * COST["B"][0])
    floor1 = valu * (fa * COST["A"][1] + fb * COST["B"][1])
    res = {"kernel": "clim_sorted_f32<20, 16, false>", "rounds_per_wave_row": args.rounds, "blocks": per_block,
           "static_valu_per_wave_row": static_valu, "measured_valu_per_wave_row": args.measured_valu or None,
           "static_to_measured": scale if args.measured_valu else None,
           "class_A_share": fa, "class_B_share": fb,
           "cost_cycles": {"A": COST["A"], "B": COST["B"], "source": "profiles/r1_ubench_valu2.txt (>= 2 waves per SIMD, one wave alone)"},
           "issue_floor_cycles_per_wave_row": {"two_or_more_waves_per_simd": floor2, "one_wave_per_simd": floor1,
                                               # 7 waves per CU: three SIMDs hold two waves, one holds one
                                               "this_kernel_mixed_occupancy": 0.75 * floor2 + 0.25 * floor1}}
    if args.measured_quad_cycles:
        cyc = 4.0 * args.measured_quad_cycles           # cycles a wave spends on one of its rows
        res["measured_wave_cycles_per_wave_row"] = cyc
        # LDS holds 7 waves per CU (--waves-per-cu): three SIMDs run two waves (a wave-row costs the SIMD floor2 cycles
        # there), one runs a wave alone (floor1).  Wave-rows a CU could retire per cycle at the floor, against what it retires:
        two = args.waves_per_cu - 4
        at_floor = two * (1.0 / floor2) + (4 - two) * (1.0 / floor1)
        measured = args.waves_per_cu / cyc
        res["frac_of_issue_floor_at_this_occupancy"] = measured / at_floor
        res["frac_of_issue_floor_at_two_waves_per_simd"] = measured / (4.0 / floor2)
        res["note"] = ("wave-rows per CU-cycle measured (waves per CU / wave cycles per row) over wave-rows per CU-cycle if every SIMD issued "
                       "a vector instruction whenever the class costs allow: at the occupancy LDS gives (2, 2, 2, 1 waves per SIMD) and "
                       "if every SIMD had two waves")
    json.dump(res, sys.stdout, indent=1)
    print(file=sys.stdout)
    for tag, b in per_block.items():
        print(f"{tag:14s} {b['block']:12s} x{b['weight']:.2f}  VALU A {b['valu_A']:4d} B {b['valu_B']:4d}  SALU {b['salu']:4d} (s_nop {b['s_nop']})  LDS {b['lds']:3d}  VMEM {b['vmem']:3d}",
              file=sys.stderr)
    print(f"static VALU per wave-row {static_valu:.0f} (class A {100 * fa:.1f} %, class B {100 * fb:.1f} %); floor at >= 2 waves per SIMD "
          f"{floor2:.0f} cycles, one wave alone {floor1:.0f}", file=sys.stderr)


if __name__ == "__main__":
    main()
