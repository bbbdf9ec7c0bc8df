#!/bin/bash
# tools/bench_ring2.py on one shape for every alternative build of the library (ab/*.so): bash tools/ab_variant.sh "<bench_ring2 args>"
cp xmhw_amd/libxmhw_amd.so /tmp/lib_keep.so
for f in ab/*.so; do
  cp "$f" xmhw_amd/libxmhw_amd.so
  python tools/bench_ring2.py $1 2>/tmp/ab.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('ring3') or {}; print('$f', d['variant'], round(d['ms'],2), 'band-only', round(r.get('wave_rows_band_only',0),3), 'rebuilds', round(r.get('rebuilds_per_wave_row',0),3), 'fail', round(r.get('cell_fail_rate',0),4), d['thresh_bit_identical_to_first'])"
done
cp /tmp/lib_keep.so xmhw_amd/libxmhw_amd.so
