#!/bin/bash
# Working-set depth of the frame pipeline (NSLOT) x sort-stream placement, untimed cfg4 frames (best of 5 x 300).
cd $GRAFT_REPO_ROOT
for ns in 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -w -DSWR_NSLOT=$ns -shared -o software-renderer_amd/lib/libswr_hip.so software-renderer_amd/csrc/swr_kernels.hip software-renderer_amd/csrc/swr_api.hip software-renderer_amd/csrc/swr_upload.hip || exit 1
  echo "NSLOT=$ns"
  timeout -k 10 300 python tools/ab_sort_stream.py 2>&1 | head -2 || exit 1
done
