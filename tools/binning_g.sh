#!/bin/bash
for g in 512 384 256 192 128; do
  echo "G=$g: $(SWR_BIN_G=$g python bench.py --steps 100 --no-cpu-baseline --no-extra 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["extra"]["kernel_ms_avg"])')"
done
