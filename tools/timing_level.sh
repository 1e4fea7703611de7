#!/bin/bash
for lvl in 1 0; do for pl in 1 0; do
  echo "timing_level=$lvl SWR_PIPELINE=$pl: $(SWR_BENCH_TIMING_LEVEL=$lvl SWR_PIPELINE=$pl python bench.py --no-cpu-baseline --no-extra --steps 300 2>/dev/null | python -c '
import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done; done
