#!/bin/bash
# per-kernel average durations of one command (run on the GPU box): bash tools/kstats.sh <args of tools/bench_sorted.py>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tr
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $R/tools/bench_sorted.py "$@" > /tmp/log 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('/tmp/tr/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:48]:48s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e6:9.4f} ms")
PY
tail -1 /tmp/log | cut -c1-100
