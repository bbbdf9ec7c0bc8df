#!/bin/bash
# finish-kernel time of alternative library builds (ab/*.so) on one box: D = 366 (configs[2]) and D = 1460 (configs[4])
cp xmhw_amd/libxmhw_amd.so /tmp/lib_keep.so
for f in ab/*.so; do
  cp "$f" xmhw_amd/libxmhw_amd.so
  for cfg in "" "--config 0.05deg_tstep --steps 3"; do
    python bench.py --no-pmc --no-cpu --no-other $cfg > /tmp/ab.json 2> /tmp/ab.err || { echo "$f failed"; tail -3 /tmp/ab.err; continue; }
    python -c "
import json; d=json.load(open('/tmp/ab.json')); print('$f', d['config']['workload'][:40], 'finish ms', round(d['finish_kernel_avg_launch_ms'],3), 'step', round(d['ms_per_step'],2), d['parity']['ok'])"
  done
done
cp /tmp/lib_keep.so xmhw_amd/libxmhw_amd.so
