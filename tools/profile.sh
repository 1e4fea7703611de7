#!/bin/bash
# Round profile: kernel-trace stats of the default bench + HBM traffic PMC passes (separate runs,
# as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra > $OUT/stats_bench.json 2> $OUT/stats.log &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/write.log 2>&1 &&
python3 - <<PY
import csv, collections, glob, json
res = {}
for name in ("fetch", "write"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % name):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            if "swr::" in k:
                res.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(res, open("$OUT/traffic_raw.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv
