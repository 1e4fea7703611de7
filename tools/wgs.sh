#!/bin/bash
# EXPERIMENT: raster workgroups per CU (dynamic-LDS padding) vs frame time and isolated k_raster time on cfg4
for k in 0 5 4 3; do
  echo "SWR_RASTER_WGS=$k: $(SWR_RASTER_WGS=$k python bench.py --steps 200 --no-cpu-baseline --no-extra | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["latency_ms"], d["roofline"]["avg_launch_ms"])')"
  echo "   isolated: $(SWR_RASTER_WGS=$k SWR_PIPELINE=0 python bench.py --steps 100 --no-cpu-baseline --no-extra | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["avg_launch_ms"])')"
done
