#!/bin/bash
# Collect SQ counters for the ring kernel on a small run (129,600 cells) and print per-wave-step figures.
# usage (on the GPU box, via gpurun): bash tools/pmc_ring.sh <tag>
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $O/p1 -- python3 $R/bench.py --cells 129600 --chunks 1 --steps 1 --warmup 0 --no-cpu --no-pmc --no-other --parity-cells 8 > $O/p1.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('$O/p1/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(float); meta=None
for r in csv.DictReader(open(f)):
    if 'clim_ring' in r['Kernel_Name']:
        agg[r['Counter_Name']] += float(r['Counter_Value']); meta=(r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'])
w = agg['SQ_WAVES']; steps = 376.0
print('regs', meta, 'waves', w)
for k in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS'):
    print(f'{k:22s} per wave-step {agg[k]/w/steps:9.1f}')
print('VALU busy quad-cycles / wave-cycles', agg['SQ_ACTIVE_INST_VALU']/agg['SQ_WAVE_CYCLES'], ' wait_any frac', agg['SQ_WAIT_ANY']/agg['SQ_WAVE_CYCLES'])
PY
tail -1 $O/p1.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ring ms', d['roofline']['avg_launch_ms'], 'cells/s', d['value'])"
