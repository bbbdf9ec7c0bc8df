#!/bin/bash
# Round-6 profile set (run on the GPU box via gpurun): bash tools/profile_round6.sh
#   r6_bench.json          the default bench line (live rocprofv3 --pmc traffic + SQ issue counters inside bench.py, 4,096-cell
#                          parity, the other_configs legs)
#   r6_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command (no CPU leg, no nested profiler)
#   r6_pmc_sq_prod.txt     SQ counters of the PRODUCT kernel clim_sorted_f32<20, 16, 14, false> (own --pmc pass)
#   r6_pmc_lds_prod.txt    its LDS counters (own --pmc pass)
#   r6_ticks.jsonl         section ticks + counters of the STATS twin (ab/stats.so through LD_PRELOAD, if present): white noise,
#                          quantised + ice (scattered cells, ice packs), 5 % NaN
#   r6_issue_mix.json/txt  the vector instructions of the product kernel by issue class and its issue floor (tools/issue_mix.py
#                          with this run's SQ counters)
# Every step runs under its own timeout; the program sits directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r6; mkdir -p $O
timeout 900 python3 $R/bench.py > $O/r6_bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-pmc --no-other --parity-cells 0 > $O/trace.log 2>&1
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/r6_kernel_stats.csv
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-pmc --no-cpu --no-other --parity-cells 0 > $O/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/lds -- python3 $R/bench.py --steps 1 --warmup 0 --no-pmc --no-cpu --no-other --parity-cells 0 > $O/lds.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag, out in (("sq", "r6_pmc_sq_prod.txt"), ("lds", "r6_pmc_lds_prod.txt")):
    fs = glob.glob('$O/' + tag + '/*/*_counter_collection.csv')
    agg = collections.defaultdict(float); meta = None; n = 0
    for r in csv.DictReader(open(fs[0])) if fs else []:
        if 'clim_sorted' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
            meta = (r['Kernel_Name'][:64], 'VGPR', r['VGPR_Count'], 'AGPR', r['Accum_VGPR_Count'], 'SGPR', r['SGPR_Count'], 'LDS', r.get('LDS_Block_Size'))
            n += r['Counter_Name'] == 'SQ_WAVES'
    w = max(agg.get('SQ_WAVES', 0.0), 1.0)
    rows = (69.0 + 11.0 + 316.0) / 3.0          # a wave runs one of the three chunks of the 40-year daily plan: 59 + 1 + 306 rows with output, 10 warm-up rows each
    lines = [f"{meta} launches {n} waves {w:.0f}; bench.py --steps 1 --no-pmc --no-cpu --no-other (configs[2], 1,036,800 cells); per wave-row ({rows:.1f} rows per wave on average, 32 cells per wave; warm-up rows build lists only)"]
    for k in sorted(agg):
        lines.append(f"{k:24s} {agg[k] / w / rows:10.1f}")
    if 'SQ_ACTIVE_INST_VALU' in agg:
        wc = agg['SQ_WAVE_CYCLES']
        lines.append('VALU busy quad-cycles / wave quad-cycles %.3f (x 2 waves per SIMD = %.3f of the SIMD)  wait_any %.3f  wait_inst_any %.3f' % (
            agg['SQ_ACTIVE_INST_VALU'] / wc, 2.0 * agg['SQ_ACTIVE_INST_VALU'] / wc, agg['SQ_WAIT_ANY'] / wc, agg['SQ_WAIT_INST_ANY'] / wc))
    open('$O/' + out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
PY
head -6 $O/r6_kernel_stats.csv | cut -c1-160
if [ -f $R/ab/stats.so ]; then
  (export LD_PRELOAD=$R/ab/stats.so; timeout 120 python3 $R/tools/bench_sorted.py --layouts 40 --reps 3; timeout 120 python3 $R/tools/bench_sorted.py --layouts 40 --reps 3 --gen 0.01 0.1 0; timeout 120 python3 $R/tools/bench_sorted.py --layouts 40 --reps 3 --gen 0.01 0.1 0 --ice-patch 4320; timeout 120 python3 $R/tools/bench_sorted.py --layouts 40 --reps 3 --config 0.25deg_nan; timeout 120 python3 $R/tools/bench_sorted.py --layouts 40 --reps 3 --q 0.1) > $O/r6_ticks.jsonl 2> $O/ticks.err
  cat $O/r6_ticks.jsonl
fi
python3 -c "
import json; d=json.load(open('$O/r6_bench.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'frac', d['roofline']['frac'], 'kernel ms', d['roofline']['avg_launch_ms'], 'raw call', d['roofline']['raw_call_avg_ms'], 'traffic/alg', (d['roofline']['traffic'] or 0)/d['roofline']['algorithmic_bytes_per_launch'])
print('binding', d['roofline']['binding'])
print('parity', d['parity']); print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
for o in d['other_configs']: print(o.get('workload','')[:50], o.get('dtype'), o.get('ms_per_step'), o.get('kernel_avg_launch_ms'), o.get('roofline_frac'), o.get('parity_cells'), o.get('parity_ok'), o.get('error'), o.get('note'))
"
python3 - <<PY
import json, subprocess, sys
d = json.load(open('$O/r6_bench.json')); b = d['roofline']['binding']
out = subprocess.run([sys.executable, '$R/tools/issue_mix.py', '--measured-valu', str(b['insts_per_wave_row']['valu']),
                      '--measured-quad-cycles', str(b['wave_quad_cycles_per_wave_row']), '--waves-per-cu', '8'], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
open('$O/r6_issue_mix.json', 'wb').write(out.stdout); open('$O/r6_issue_mix.txt', 'wb').write(out.stderr)
print(out.stderr.decode()[-1200:])
PY
tail -3 $O/bench.err
