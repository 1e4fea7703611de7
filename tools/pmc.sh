#!/bin/bash
# two SQ counter passes over a short bench run; prints per-kernel averages
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmcA -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/pmcA.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmcB -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/pmcB.log 2>&1
python3 - <<PY
import csv, collections, glob
for d in ('pmcA','pmcB'):
    for f in glob.glob('$R/gpurun_out/%s/*/*counter_collection.csv' % d):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:30]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items():
            if 'swr::' in k: print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
