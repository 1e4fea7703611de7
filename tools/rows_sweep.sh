#!/bin/bash
# sweep ROWS (rows pooled per dealing round) and the k_raster occupancy bound
for cfg in "1 6" "2 6" "2 5" "3 6" "3 5" "4 5" "4 6"; do
  set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -w -DSWR_ROWS=$1 -DSWR_RASTER_MIN_WAVES=$2 -shared -o software-renderer_amd/lib/libswr_hip.so software-renderer_amd/csrc/swr_kernels.hip software-renderer_amd/csrc/swr_api.hip
  echo "ROWS=$1 minwaves=$2: $(SWR_PIPELINE=0 python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["extra"]["kernel_ms_avg"]["raster_ms"])')"
done
make -C software-renderer_amd -s -B lib/libswr_hip.so
