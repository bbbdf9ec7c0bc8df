#!/bin/bash
# A/B timing of alternative builds of libxmhw_amd.so on ONE box: every ab/<name>.so is copied over the in-tree
# library in turn; the headline bench is run REPS times (kernel ms from the HIP events of bench.py), then the
# ring kernel alone on the shapes named in SHAPES ("config:cells:variant ...", tools/bench_ring2.py).
#   gpurun -- 'bash tools/ab_bench.sh 2 "1deg:1036800:21"'
REPS=${1:-2}
SHAPES=${2:-}
cp xmhw_amd/libxmhw_amd.so /tmp/lib_keep.so
for round in $(seq 1 $REPS); do
  for f in ab/*.so; do
    cp "$f" xmhw_amd/libxmhw_amd.so
    python bench.py --no-pmc --no-cpu --no-other > /tmp/ab.json 2> /tmp/ab.err || { echo "$f failed"; tail -3 /tmp/ab.err; continue; }
    python -c "
import json; d=json.load(open('/tmp/ab.json')); print('$f', round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],2), d['parity']['ok'])"
    for sh in $SHAPES; do
      IFS=: read cfg cells var <<< "$sh"
      python tools/bench_ring2.py --config $cfg --cells $cells --variants $var 2>/tmp/ab.err | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('   $f $sh', round(d['ms'],3))"
    done
  done
done
cp /tmp/lib_keep.so xmhw_amd/libxmhw_amd.so
