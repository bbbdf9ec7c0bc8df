#!/bin/bash
# Round profile: kernel-trace stats of the default bench command + HBM traffic PMC passes.
# usage on the GPU box (via gpurun): bash tools/profile_round.sh r1
TAG=${1:-rX}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-pmc --no-other > $O/trace.log 2>&1
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), one step, no warm-up
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-pmc --no-other --parity-cells 8 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-pmc --no-other --parity-cells 8 > $O/pmc_write.log 2>&1
$R/tools/ubench_valu > $O/ubench_valu.txt 2>&1
python3 - <<PY
import csv, glob, collections
out = open('$O/pmc_hbm.txt', 'w')
def p(*a):
    print(*a); print(*a, file=out)
T, C, D = 14610, 1036800, 366
for name, d in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
    f = glob.glob('$O/' + d + '/*/*_counter_collection.csv')[0]
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k] += float(r['Counter_Value']); n[k] += 1
    for k, v in agg.items():
        if any(s in k for s in ('clim_ring', 'clim_finish', 'synth_sst')):
            p(f'{name:11s} {k[:60]:60s} launches {n[k]:2d}  per launch {v / n[k] / 1e6:10.3f} GB (counter unit = KB)')
p('known bytes: synth_sst writes T*C*4 = %.3f GB; ring algorithmic read %.3f GB + write %.3f GB; finish reads >= %.3f GB, writes %.3f GB'
  % (T*C*4/1e9, T*C*4/1e9, 2*D*C*8/1e9, 2*D*C*8/1e9, 2*D*C*8/1e9))
PY
ls $O
