#!/bin/bash
# perf exploration: rebuild with different tile shapes and run the bench (on the GPU box)
for shape in "64 32" "64 16" "32 32" "32 16"; do
  set -- $shape
  sed -i "s/constexpr int TILE_W = [0-9]*;/constexpr int TILE_W = $1;/; s/constexpr int TILE_H = [0-9]*;/constexpr int TILE_H = $2;/" software-renderer_amd/csrc/swr_internal.h
  make -C software-renderer_amd -s lib/libswr_hip.so 2>&1 | grep -E "error" 
  echo "tile $1x$2: $(python bench.py --steps 100 --no-cpu-baseline --no-extra 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["extra"]["kernel_ms_avg"], d["extra"]["tile_pairs"])')"
done
sed -i "s/constexpr int TILE_W = [0-9]*;/constexpr int TILE_W = 64;/; s/constexpr int TILE_H = [0-9]*;/constexpr int TILE_H = 32;/" software-renderer_amd/csrc/swr_internal.h
