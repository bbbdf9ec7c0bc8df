#!/bin/bash
# like ab_bench.sh, with the HBM traffic of each build (bench.py's PMC leg): kernel ms, traffic / algorithmic bytes
cp xmhw_amd/libxmhw_amd.so /tmp/lib_keep.so
for f in ab/*.so; do
  cp "$f" xmhw_amd/libxmhw_amd.so
  python bench.py --no-cpu --no-other > /tmp/ab.json 2> /tmp/ab.err || { echo "$f failed"; tail -3 /tmp/ab.err; continue; }
  python -c "
import json; d=json.load(open('/tmp/ab.json')); r=d['roofline']; print('$f', round(d['ms_per_step'],2), round(r['avg_launch_ms'],2), 'traffic x', round((r['traffic'] or 0)/r['algorithmic_bytes_per_launch'],3), d['parity']['ok'])"
done
cp /tmp/lib_keep.so xmhw_amd/libxmhw_amd.so
