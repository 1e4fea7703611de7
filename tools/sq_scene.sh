#!/bin/bash
# SQ counters + HBM traffic of every kernel of a named scene (tools/frames.py), pipelining off: tools/sq_scene.sh cfg4c [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SC=${1:-cfg4c}; TAG=${2:-sqs}
n=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  SWR_PIPELINE=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_$n -- python3 $R/tools/frames.py $SC 6 > $R/gpurun_out/${TAG}_$n.log 2>&1 || { echo "pass $n failed"; tail -3 $R/gpurun_out/${TAG}_$n.log; }
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$R/gpurun_out/${TAG}_*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'swr::' in k and ('raster' in k or 'k_bin' in k or 'sort' in k): print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
