#!/bin/bash
# SQ / LDS counters + kernel trace of the PRODUCT sorted-list kernel on configs[2] (run on the GPU box):
#   bash tools/pmc_sorted.sh [outdir] [bench_sorted args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/pmc_sorted}; shift; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/bench_sorted.py --layouts 40 --reps 5 "$@" > $O/trace.log 2>&1
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- python3 $R/tools/bench_sorted.py --layouts 40 --reps 1 "$@" > $O/sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/lds -- python3 $R/tools/bench_sorted.py --layouts 40 --reps 1 "$@" > $O/lds.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in ("sq", "lds"):
    fs = glob.glob('$O/' + tag + '/*/*_counter_collection.csv')
    agg = collections.defaultdict(float); meta = None
    for r in csv.DictReader(open(fs[0])) if fs else []:
        if 'clim_sorted' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
            meta = (r['Kernel_Name'][:60], 'VGPR', r['VGPR_Count'], 'AGPR', r['Accum_VGPR_Count'], 'SGPR', r['SGPR_Count'], 'LDS', r.get('LDS_Block_Size'))
    w = max(agg.get('SQ_WAVES', 0.0), 1.0)
    lines = [f"{meta} waves {w:.0f} (all launches of the run); per wave"]
    for k in sorted(agg):
        lines.append(f"{k:24s} {agg[k] / w:12.1f}")
    if 'SQ_ACTIVE_INST_VALU' in agg:
        wc = agg['SQ_WAVE_CYCLES']
        lines.append('VALU busy / wave cycles %.3f  wait_any %.3f  wait_inst_any %.3f' % (agg['SQ_ACTIVE_INST_VALU'] / wc, agg['SQ_WAIT_ANY'] / wc, agg['SQ_WAIT_INST_ANY'] / wc))
    open('$O/' + tag + '.txt', 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
PY
cat $O/kernel_stats.csv | cut -c1-160 | head -8
