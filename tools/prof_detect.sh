#!/bin/bash
# rocprofv3 kernel stats of tools/bench_detect.py ($1 cells)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_detect; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/bench_detect.py ${1:-518400} 3 > $O/run.log 2>&1
f=$(ls $O/*/*kernel_stats.csv | head -1)
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
