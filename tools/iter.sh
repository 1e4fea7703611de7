#!/bin/bash
# One development iteration on the GPU box: tools/iter.sh <tag> [tests|notests] [pmc|nopmc]
#   gpu parity tests (stop at the first failure), the bench line with the driver's arguments and with the defaults,
#   per-kernel durations with the stages serialised (k_raster alone), SQ_INSTS_VALU of k_raster.
# Everything lands in gpurun_out/<tag>_*.
R=$GRAFT_REPO_ROOT; TAG=${1:-it}; TESTS=${2:-tests}; PMC=${3:-pmc}
cd $R
if [ "$TESTS" = tests ]; then
  timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1
  rc=$?; tail -3 gpurun_out/${TAG}_tests.log
  [ $rc -ne 0 ] && { echo "TESTS FAILED"; exit 1; }
fi
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/${TAG}_bench20.json 2> gpurun_out/${TAG}_bench20.err || { tail -5 gpurun_out/${TAG}_bench20.err; exit 1; }
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python3 - <<PY
import json
for f in ("bench20", "bench"):
    d = json.loads(open("gpurun_out/${TAG}_%s.json" % f).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print(f, "ms_per_step", d["ms_per_step"], "value", d["value"], "raster_ms", r.get("avg_launch_ms"), "isolated", (r.get("isolated") or {}).get("avg_launch_ms"),
          "color", (d.get("extra") or {}).get("color_plus_depth", {}).get("ms_per_step"), "render_ms", (d.get("extra") or {}).get("swr_render_ms"))
PY
timeout -k 10 300 bash tools/kstats.sh ${TAG} cfg4 30 || exit 1
if [ "$PMC" = pmc ]; then
  cd /tmp && export TMPDIR=/tmp
  SWR_PIPELINE=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/${TAG}_sq -- python3 $R/tools/frames.py cfg4 6 > $R/gpurun_out/${TAG}_sq.log 2>&1 || { tail -3 $R/gpurun_out/${TAG}_sq.log; exit 1; }
  python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$R/gpurun_out/${TAG}_sq/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'swr::' in k: print("$TAG", k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
fi
