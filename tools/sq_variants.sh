#!/bin/bash
# SQ_INSTS_VALU / SALU / LDS + wave cycles of the k_raster ablation variants (ablation build), one rocprofv3 pass each
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SWR_LIBRARY=$R/software-renderer_amd/lib/libswr_hip_ablation.so SWR_PIPELINE=0
for v in ${VARIANTS:-0 1 2 3 4 8 9 10 11}; do
  export SWR_DEBUG_VARIANT=$v
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/sqv_$v -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/sqv_$v.log 2>&1 || { echo "variant $v failed"; tail -3 $R/gpurun_out/sqv_$v.log; }
  python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$R/gpurun_out/sqv_$v/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'k_raster' in k: print('variant $v', {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
