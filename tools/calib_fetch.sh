cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/calib; mkdir -p $O
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/tools/calib_fetch.py > $O/f.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$O/f/*/*_counter_collection.csv')[0]
for r in csv.DictReader(open(f)):
    if 'land_mask' in r['Kernel_Name']:
        v = float(r['Counter_Value']) * 1024
        print('land_mask FETCH_SIZE bytes', v, 'true', 4*14610*262144, 'ratio', v/(4*14610*262144))
PY
