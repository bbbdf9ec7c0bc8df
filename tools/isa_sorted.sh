#!/bin/bash
# ISA of clim_sorted_f32<20, 16, false> alone (no GPU needed): /tmp/isa/k20.s + the vector instructions of its hot blocks.
#   bash tools/isa_sorted.sh [extra hipcc flags]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -DXMHW_SORTED_ONLY "$@" \
    -Rpass-analysis=kernel-resource-usage -S -o /tmp/isa/k.s $R/xmhw_amd/csrc/kernels_sorted.hip 2> /tmp/isa/res.txt || { cat /tmp/isa/res.txt; exit 1; }
python3 - <<'PY'
import re
txt = open('/tmp/isa/k.s').read()
sym = "_ZN4xmhw15clim_sorted_f32ILi20ELi16ELi14ELb0EEE"
i = txt.index(sym); i = txt.index(sym, i + 10)
body = txt[i:txt.index(".Lfunc_end", i)]
open('/tmp/isa/k20.s', 'w').write(body)
blocks = []; cur = ['entry', [], 0]; blocks.append(cur)
for ln, line in enumerate(body.split('\n')):
    m = re.match(r"^(\.LBB\d+_\d+):", line) or re.match(r"^; %bb\.(\d+):", line)
    if m:
        cur = [m.group(0).strip(':; '), [], ln]; blocks.append(cur)
    elif re.match(r"^\s+(v_|s_|ds_|global_|buffer_)", line):
        cur[1].append(line.split()[0])
tot = 0
for name, ins, ln in blocks:
    v = sum(1 for o in ins if o.startswith('v_'))
    if v >= 15:
        print(f"{name:12s} line {ln:5d} valu {v:4d} ds {sum(1 for o in ins if o.startswith('ds_')):3d} salu {sum(1 for o in ins if o.startswith('s_') and not o.startswith('s_waitcnt') and not o.startswith('s_nop')):3d} nop {sum(1 for o in ins if o.startswith('s_nop')):2d} vmem {sum(1 for o in ins if o.startswith('global')):2d}")
PY
# vmcnt waits inside the row loop: the key conversion's 20 (one per sample, the last one vmcnt(0)) are what the loop needs; the
# address rebuild of irregular steps has 20 more (a cold block).  ANY other vmcnt wait in the hot blocks (sort, bookkeeping,
# select) sits out the requests for the next row's samples -- the prefetch is gone and the kernel 12 % slower (round 6: a
# register shared between a cold path's load and a key made the compiler put one in front of the sort).
echo "s_waitcnt vmcnt(0) in the row loop at lines: $(grep -n 's_waitcnt vmcnt(0)' /tmp/isa/k20.s | awk -F: '$1 > 440 && $1 < 3300 {printf "%s ", $1}') (expected: two -- the conversion's and the cold address rebuild's)"
grep -A12 "clim_sorted_f32ILi20ELi16ELi14ELb0" /tmp/isa/res.txt | grep -E "VGPRs:|SGPRs:|Occupancy|LDS Size|Scratch" | head -6
