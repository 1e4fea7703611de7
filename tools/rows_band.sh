#!/bin/bash
# ROWS / occupancy sweep of k_raster for a thin band (rank 4 of 8) and the full frame: kernel durations from events.
cd $GRAFT_REPO_ROOT
for cfg in "2 5" "4 4" "4 5" "3 5"; do
  set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -w -DSWR_ROWS=$1 -DSWR_RASTER_MIN_WAVES=$2 -shared -o software-renderer_amd/lib/libswr_hip.so software-renderer_amd/csrc/swr_kernels.hip software-renderer_amd/csrc/swr_api.hip software-renderer_amd/csrc/swr_upload.hip || exit 1
  echo "ROWS=$1 minwaves=$2"
  timeout -k 10 120 python tools/band_proxy.py 8 1 2>&1 | grep "band [04]" || exit 1
done
