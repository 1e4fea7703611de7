#!/bin/bash
for pl in ${PLS:-1 0}; do
  echo "SWR_PIPELINE=$pl: $(SWR_PIPELINE=$pl python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c '
import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["extra"]["kernel_ms_avg"], d["extra"]["color_plus_depth"])')"
done
