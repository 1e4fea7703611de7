#!/bin/bash
# Binning workgroup size x count: can a binning workgroup co-reside with five raster workgroups per CU?
cd $GRAFT_REPO_ROOT
for cfg in "1024 256" "256 256" "256 512" "512 256" "128 512"; do
  set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -w -DSWR_BIN_THREADS=$1 -shared -o software-renderer_amd/lib/libswr_hip.so software-renderer_amd/csrc/swr_kernels.hip software-renderer_amd/csrc/swr_api.hip software-renderer_amd/csrc/swr_upload.hip || exit 1
  echo "BIN_THREADS=$1 G=$2: $(SWR_BIN_G=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["extra"]["kernel_ms_avg"])')" || exit 1
done
