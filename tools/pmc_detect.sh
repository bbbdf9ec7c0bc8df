#!/bin/bash
# HBM read bytes (by request size) and written bytes of the detect() kernels:
# TCC_EA0_RDREQ_{32B,64B,128B}_sum and WRITE_SIZE in separate --pmc passes.  $1 = cells (default 259200)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_detect; mkdir -p $O
N=${1:-259200}
C="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/rd -- python3 $R/tools/bench_detect.py $N 1 > $O/rd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/wr -- python3 $R/tools/bench_detect.py $N 1 > $O/wr.log 2>&1
python3 - <<PY
import csv, glob, collections
def agg(d, key):
    f = glob.glob('$O/' + d + '/*/*_counter_collection.csv')[0]
    a = collections.defaultdict(float); n = 0
    for r in csv.DictReader(open(f)):
        if key in r['Kernel_Name']:
            a[r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] in ('TCC_EA0_RDREQ_sum', 'WRITE_SIZE'): n += 1
    return a, max(n, 1)
T, N = 14610, $N
for key, alg_r, alg_w in (('detect_events', T*4 + 366*8, T*13), ('event_stats<', T*8 + 2*366*8, 14*31*8), ('count_events', T*4, 4),
                          ('exceed_bits', T*4 + 366*4, T/8), ('events_from_bits', T/8, 14*4*8/2), ('event_stats_sparse', 14*10*24, 14*31*8)):
    a, n = agg('rd', key)
    w, nw = agg('wr', key)
    by = (32*a['TCC_EA0_RDREQ_32B_sum'] + 64*a['TCC_EA0_RDREQ_64B_sum'] + 128*a['TCC_EA0_RDREQ_128B_sum']) / n
    wr = w['WRITE_SIZE'] * 1024 / nw      # KB units
    print(f'{key:14s} launches {n}  read {by/1e9:.3f} GB/launch (algorithmic {N*alg_r/1e9:.3f})  written {wr/1e9:.3f} GB/launch (algorithmic {N*alg_w/1e9:.3f})')
PY
