#!/bin/bash
# float64 configs[2] (genuinely float64 samples) on alternative library builds (ab/*.so), one box
cp xmhw_amd/libxmhw_amd.so /tmp/lib_keep.so
for round in 1 2; do
for f in ab/*.so; do
  cp "$f" xmhw_amd/libxmhw_amd.so
  python bench.py --dtype f64 --no-pmc --no-cpu --no-other --steps 3 > /tmp/ab.json 2> /tmp/ab.err || { echo "$f failed"; tail -3 /tmp/ab.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/ab.json')); r=d['roofline']; print('$f', round(d['ms_per_step'],2), round(r['avg_launch_ms'],2), r['kernel'][:40], d['parity']['ok'])"
done
done
cp /tmp/lib_keep.so xmhw_amd/libxmhw_amd.so
