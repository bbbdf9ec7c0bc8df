#!/bin/bash
# Round-3 profile set (run on the GPU box via gpurun): bash tools/profile_round3.sh
#   r3_bench.json            the default bench line (live HBM traffic by rocprofv3 --pmc inside bench.py, other_configs)
#   r3_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command (no CPU leg, no nested profiler)
#   r3_pmc_sq_v21.txt        SQ counters of clim_ring3_f32 on 4 lanes per cell (variant 21, shipped), 129,600 cells
#   r3_pmc_sq_v20.txt        ... on 8 lanes per cell (variant 20)
#   r3_pmc_lds_v21.txt       LDS counters of variant 21: instructions, bank conflicts, busy cycles (own --pmc pass)
#   r3_ring3_variants.jsonl  tools/bench_ring2.py on the four grid configs, variants 8 / 10 (ring2) and 20 / 21 / 22 (ring3 on 8 / 4 / 2 lanes per cell)
# Every step runs under its own timeout.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r3; mkdir -p $O
timeout 900 python3 $R/bench.py > $O/r3_bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-pmc --no-other > $O/trace.log 2>&1
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/r3_kernel_stats.csv
timeout 600 bash $R/tools/pmc_ring2.sh r3v21 21 > /dev/null 2>&1
cp $R/gpurun_out/pmc_r3v21/summary.txt $O/r3_pmc_sq_v21.txt
timeout 600 bash $R/tools/pmc_ring2.sh r3v20 20 > /dev/null 2>&1
cp $R/gpurun_out/pmc_r3v20/summary.txt $O/r3_pmc_sq_v20.txt
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $O/lds -- python3 $R/tools/bench_ring2.py --config 0.25deg --cells 129600 --chunks 1 --variants 21 --reps 1 > $O/lds.log 2>&1
python3 - <<PY | tee $O/r3_pmc_lds_v21.txt
import csv, glob, collections
fs = glob.glob('$O/lds/*/*_counter_collection.csv')
agg = collections.defaultdict(float)
for r in csv.DictReader(open(fs[0])) if fs else []:
    if 'clim_ring3' in r['Kernel_Name']:
        agg[r['Counter_Name']] += float(r['Counter_Value'])
w = max(agg.get('SQ_WAVES', 0.0), 1.0)
print('clim_ring3_f32<10, 4> (variant 21), 129,600 cells, one chunk: per wave-row (376 rows per wave)')
for k in sorted(agg):
    print(f'{k:24s} {agg[k] / w / 376.0:10.1f}')
PY
for cfg in 0.25deg 1deg 0.25deg_nan 0.05deg_tstep; do
  timeout 600 python3 $R/tools/bench_ring2.py --config $cfg --variants 8 10 20 21 22 --reps 3 >> $O/r3_ring3_variants.jsonl 2>> $O/variants.err
done
head -4 $O/r3_kernel_stats.csv
cat $O/r3_pmc_sq_v21.txt
cut -c1-600 $O/r3_bench.json
