#!/bin/bash
# SQ counters of the raster kernel for several hook settings (SWR_AB_HOOKS), pipelining off: tools/sq_ab.sh "insort=0" "" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for H in "$@"; do
  i=$((i+1))
  SWR_AB_HOOKS="$H" SWR_PIPELINE=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/sqab_$i -- python3 $R/tools/frames.py cfg4 6 > $R/gpurun_out/sqab_$i.log 2>&1 || { tail -3 $R/gpurun_out/sqab_$i.log; }
  python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$R/gpurun_out/sqab_$i/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'k_raster' in k or 'k_sort' in k: print("[$H]", k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
