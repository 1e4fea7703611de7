#!/bin/bash
# SQ counter passes (separate --pmc runs, kernel-trace only) over a short serialised cfg4 run; prints per-kernel averages
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-sq}
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR"; do
  n=$((n+1))
  SWR_PIPELINE=0 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_$n -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/${TAG}_$n.log 2>&1 || { echo "pass $n failed"; tail -3 $R/gpurun_out/${TAG}_$n.log; }
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$R/gpurun_out/${TAG}_*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'swr::' in k: print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
