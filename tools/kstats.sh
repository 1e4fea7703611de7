#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats), stages serialised: tools/kstats.sh <tag> <frames.py args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift
SWR_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$TAG -- python3 $R/tools/frames.py "$@" > $R/gpurun_out/ks_$TAG.log 2>&1 || tail -3 $R/gpurun_out/ks_$TAG.log
python3 - <<PY
import csv, glob
for f in glob.glob('$R/gpurun_out/ks_$TAG/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'swr::' in r['Name']: print(f"$TAG {r['Name'].split('(')[0][:46]:46s} calls {r['Calls']:>4} avg {float(r['AverageNs'])/1e3:8.1f} us  total% {r['Percentage']}")
PY
