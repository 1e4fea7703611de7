#!/bin/bash
# timing-only ablations of k_raster (results invalid for variant != 0)
for v in ${VARIANTS:-0 1 2 3}; do
  echo "variant $v: $(SWR_DEBUG_VARIANT=$v python bench.py --steps 100 --no-cpu-baseline --no-extra | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["extra"]["kernel_ms_avg"])')"
done
