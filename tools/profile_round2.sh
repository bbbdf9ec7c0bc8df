#!/bin/bash
# Round-2 profile set (run on the GPU box via gpurun): bash tools/profile_round2.sh
#   r2_bench.json                 the default bench line (live HBM traffic by rocprofv3 --pmc inside bench.py)
#   r2_kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command (no CPU leg, no nested profiler)
#   r2_pmc_sq_ring2.txt           SQ counters of the ring kernel (129,600 cells, one chunk)
#   r2_configs_kernel_stats.csv   kernel trace of tools/trace_configs.py (the device half of tests/test_gpu_configs.py):
#                                 which kernels the BASELINE configs run; r2_configs.jsonl = its timings
# Every step runs under its own timeout.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r2; mkdir -p $O
timeout 600 python3 $R/bench.py > $O/r2_bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-pmc > $O/trace.log 2>&1
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/r2_kernel_stats.csv
timeout 900 bash $R/tools/pmc_ring2.sh r2final 8 > /dev/null 2>&1
cp $R/gpurun_out/pmc_r2final/summary.txt $O/r2_pmc_sq_ring2.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfgtrace -- python3 $R/tools/trace_configs.py > $O/cfgtrace.log 2>&1
grep '^{' $O/cfgtrace.log > $O/r2_configs.jsonl
cp $(ls $O/cfgtrace/*/*kernel_stats.csv | head -1) $O/r2_configs_kernel_stats.csv
tail -3 $O/cfgtrace.log
head -5 $O/r2_kernel_stats.csv
cat $O/r2_pmc_sq_ring2.txt
cut -c1-400 $O/r2_bench.json
