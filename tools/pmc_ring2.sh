#!/bin/bash
# SQ counters of one ring-kernel variant on 129,600 cells, one chunk (16,200 waves x 376 steps).
# usage (on the GPU box, via gpurun): bash tools/pmc_ring2.sh <tag> <variant> [config]
TAG=${1:-x}; VAR=${2:-0}; CFG=${3:-0.25deg}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -- python3 $R/tools/bench_ring2.py --config $CFG --cells 129600 --chunks 1 --variants $VAR --reps 1 > $O/p1.log 2>&1
python3 - <<PY | tee $O/summary.txt
import csv, glob, collections
f = glob.glob('$O/p1/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(float); meta=None; n=0
for r in csv.DictReader(open(f)):
    if 'clim_ring' in r['Kernel_Name']:
        agg[r['Counter_Name']] += float(r['Counter_Value']); meta=(r['Kernel_Name'][:60], r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r.get('LDS_Block_Size'))
        if r['Counter_Name'] == 'SQ_WAVES': n += 1
w = agg['SQ_WAVES']
import json
steps = 376.0 if '$CFG' != '0.05deg_tstep' else 1470.0
print('variant $VAR config $CFG', meta, 'launches', n, 'waves', w)
for k in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_WAVE_CYCLES','SQ_ACTIVE_INST_VALU','SQ_WAIT_ANY','SQ_WAIT_INST_ANY'):
    print(f'{k:22s} per wave-step {agg[k]/w/steps:9.1f}')
print('VALU busy quad-cycles / wave-cycles %.3f  wait_any %.3f  wait_inst_any %.3f' % (agg['SQ_ACTIVE_INST_VALU']/agg['SQ_WAVE_CYCLES'], agg['SQ_WAIT_ANY']/agg['SQ_WAVE_CYCLES'], agg['SQ_WAIT_INST_ANY']/agg['SQ_WAVE_CYCLES']))
PY
tail -2 $O/p1.log
