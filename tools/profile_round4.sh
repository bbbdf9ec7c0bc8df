#!/bin/bash
# Round-4 profile set (run on the GPU box via gpurun): bash tools/profile_round4.sh
#   r4_bench.json          the default bench line (live rocprofv3 --pmc traffic + SQ issue counters inside bench.py,
#                          4,096-cell parity, six other_configs)
#   r4_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command (no CPU leg, no nested profiler)
#   r4_pmc_sq_prod.txt     SQ counters of the PRODUCT kernel clim_ring3_f32<10, 4, false, float, false> under
#                          `bench.py --steps 1 --no-pmc --no-cpu --no-other` (task 3)
#   r4_pmc_lds_prod.txt    its LDS counters (own --pmc pass)
# Every step runs under its own timeout; the program sits directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r4; mkdir -p $O
timeout 900 python3 $R/bench.py > $O/r4_bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-pmc --no-other --parity-cells 0 > $O/trace.log 2>&1
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/r4_kernel_stats.csv
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-pmc --no-cpu --no-other --parity-cells 0 > $O/sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $O/lds -- python3 $R/bench.py --steps 1 --warmup 0 --no-pmc --no-cpu --no-other --parity-cells 0 > $O/lds.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag, out in (("sq", "r4_pmc_sq_prod.txt"), ("lds", "r4_pmc_lds_prod.txt")):
    fs = glob.glob('$O/' + tag + '/*/*_counter_collection.csv')
    agg = collections.defaultdict(float); meta = None; n = 0
    for r in csv.DictReader(open(fs[0])) if fs else []:
        if 'clim_ring' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
            meta = (r['Kernel_Name'][:80], 'VGPR', r['VGPR_Count'], 'AGPR', r['Accum_VGPR_Count'], 'SGPR', r['SGPR_Count'], 'LDS', r.get('LDS_Block_Size'))
            n += r['Counter_Name'] == 'SQ_WAVES'
    w = max(agg.get('SQ_WAVES', 0.0), 1.0)
    rows = 376.0          # 366 rows with output + 10 warm-up rows, one chunk
    lines = [f"{meta} launches {n} waves {w:.0f}; bench.py --steps 1 --no-pmc --no-cpu --no-other (configs[2], 1,036,800 cells); per wave-row ({rows:.0f} rows per wave, 16 cells per wave)"]
    for k in sorted(agg):
        lines.append(f"{k:24s} {agg[k] / w / rows:10.1f}")
    if 'SQ_ACTIVE_INST_VALU' in agg:
        wc = agg['SQ_WAVE_CYCLES']
        lines.append('VALU busy quad-cycles / wave quad-cycles %.3f (x 2 waves per SIMD = %.3f of the SIMD)  wait_any %.3f  wait_inst_any %.3f' % (
            agg['SQ_ACTIVE_INST_VALU'] / wc, 2 * agg['SQ_ACTIVE_INST_VALU'] / wc, agg['SQ_WAIT_ANY'] / wc, agg['SQ_WAIT_INST_ANY'] / wc))
    open('$O/' + out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
PY
head -5 $O/r4_kernel_stats.csv
python3 -c "
import json; d=json.load(open('$O/r4_bench.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'frac', d['roofline']['frac'], 'traffic/alg', (d['roofline']['traffic'] or 0)/d['roofline']['algorithmic_bytes_per_launch'])
print('binding', d['roofline']['binding'])
print('parity', d['parity']); print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
for o in d['other_configs']: print(o.get('workload','')[:50], o.get('dtype'), o.get('kernel_avg_launch_ms'), o.get('roofline_frac'), o.get('parity_cells'), o.get('parity_ok'), o.get('error'), o.get('note'))
"
tail -3 $O/bench.err
