cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in ab_r02 libswr_hip; do
  export SWR_LIBRARY=$R/software-renderer_amd/lib/$lib.so
  SWR_PIPELINE=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/sqbig_$lib -- python3 $R/tools/frames.py big 6 > $R/gpurun_out/sqbig_$lib.log 2>&1 || tail -3 $R/gpurun_out/sqbig_$lib.log
  python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$R/gpurun_out/sqbig_$lib/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'k_raster' in k: print("$lib", k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
