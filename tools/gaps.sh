#!/bin/bash
# start / end / duration / gap to the previous kernel of the same queue, from a rocprofv3 kernel trace:
#   tools/gaps.sh <tag> <SWR_PIPELINE 0|1> <frames.py args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; PL=$2; shift; shift
rm -rf $R/gpurun_out/gaps_$TAG
SWR_PIPELINE=$PL rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps_$TAG -- python3 $R/tools/frames.py "$@" > $R/gpurun_out/gaps_$TAG.log 2>&1 || tail -3 $R/gpurun_out/gaps_$TAG.log
python3 - <<PY > $R/gpurun_out/gaps_$TAG.txt
import csv, glob
rows = []
for f in glob.glob('$R/gpurun_out/gaps_$TAG/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'swr::' in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void swr::', '').split('<')[0], r['Queue_Id']))
rows.sort()
t0 = rows[len(rows) // 2][0]
last_end = {}
print("# start end dur gap_to_previous_kernel_of_same_queue(us) kernel queue")
for s, e, n, q in rows[len(rows) // 2: len(rows) // 2 + 60]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else float('nan')
    last_end[q] = e
    print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {gap:7.1f}  {n} q{q}")
PY
head -45 $R/gpurun_out/gaps_$TAG.txt
