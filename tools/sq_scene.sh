#!/bin/bash
# SQ counters of k_raster on a named scene of tools/frames.py for several library builds: tools/sq_scene.sh <scene> <lib.so> [<lib.so> ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; SCENE=$1; shift
for lib in "$@"; do
  export SWR_LIBRARY=$R/$lib
  tag=$(basename $lib .so)
  n=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD" \
             "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR"; do
    n=$((n+1))
    SWR_PIPELINE=0 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/sqs_${tag}_$n -- python3 $R/tools/frames.py $SCENE 6 > $R/gpurun_out/sqs_${tag}_$n.log 2>&1 || tail -3 $R/gpurun_out/sqs_${tag}_$n.log
  done
  python3 - <<PY
import csv, collections, glob
out = {}
for f in sorted(glob.glob('$R/gpurun_out/sqs_${tag}_*/*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_raster' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, x in agg.items():
        x = sorted(x)[len(x)//2:]            # drop the small (overflowed / warm-up) launches: upper half
        out[c] = round(sum(x)/len(x))
print("$tag $SCENE", out)
PY
done
