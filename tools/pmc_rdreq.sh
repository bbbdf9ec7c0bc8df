#!/bin/bash
# Exact HBM read bytes by request size: TCC_EA0_RDREQ{,_32B,_64B,_128B}_sum for the ring kernel
# (full bench workload, one step) and for land_mask (known bytes) as calibration.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rdreq; mkdir -p $O
C="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/ring -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-pmc --no-other --parity-cells 8 $BENCH_ARGS > $O/ring.log 2>&1
timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/calib -- python3 $R/tools/calib_fetch.py > $O/calib.log 2>&1
python3 - <<PY
import csv, glob, collections
def agg(d, key):
    f = glob.glob('$O/' + d + '/*/*_counter_collection.csv')[0]
    a = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if key in r['Kernel_Name']: a[r['Counter_Name']] += float(r['Counter_Value'])
    return a
for d, key, true in (('calib', 'land_mask', 4*14610*262144), ('ring', 'clim_ring', None), ('ring', 'clim_finish', None)):
    a = agg(d, key)
    n, n32, n64, n128 = (a['TCC_EA0_RDREQ_sum'], a['TCC_EA0_RDREQ_32B_sum'], a['TCC_EA0_RDREQ_64B_sum'], a['TCC_EA0_RDREQ_128B_sum'])
    other = n - n32 - n64 - n128
    by = 32*n32 + 64*n64 + 128*n128
    print(f'{key:12s} RDREQ {n:.4g}  32B {n32:.4g}  64B {n64:.4g}  128B {n128:.4g}  (unsized {other:.4g})  bytes by size {by/1e9:.3f} GB'
          + (f'  input bytes {true/1e9:.3f} GB  ratio {by/true:.4f}' if true else ''))
PY
