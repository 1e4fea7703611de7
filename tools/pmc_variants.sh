#!/bin/bash
# SQ counters of k_raster for a list of ablation variants
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-0}; do
  export SWR_DEBUG_VARIANT=$v
  rm -rf $R/gpurun_out/pmcV
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmcV -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/pmcV.log 2>&1
  python3 - <<PY
import csv, collections, glob
for f in glob.glob('$R/gpurun_out/pmcV/*/*counter_collection.csv'):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:30]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'raster' in k: print('variant $v', {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
