#!/bin/bash
# quick A/B of library builds on the headline: pipelined frame time, k_raster in the pipeline, isolated: tools/ab_bench.sh lib1 lib2 ...
for rep in 1 2; do for lib in "$@"; do
  echo "$lib: $(SWR_LIBRARY=$PWD/$lib python bench.py --steps 200 --no-cpu-baseline --no-extra | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step", d["ms_per_step"], "latency", d["latency_ms"], "raster(pipelined)", d["roofline"]["avg_launch_ms"])')"
done; done
