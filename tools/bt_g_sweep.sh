#!/bin/bash
# binning workgroup size (SWR_BIN_BT) x count (SWR_BIN_G) x pipeline depth, untimed cfg4 frames
cd $GRAFT_REPO_ROOT
for ns in 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -w -DSWR_NSLOT=$ns -shared -o software-renderer_amd/lib/libswr_hip.so software-renderer_amd/csrc/swr_kernels.hip software-renderer_amd/csrc/swr_api.hip software-renderer_amd/csrc/swr_upload.hip || exit 1
  for cfg in "1024 256" "256 256" "256 512" "256 1024"; do
    set -- $cfg
    echo "NSLOT=$ns BT=$1 G=$2: $(SWR_BIN_BT=$1 SWR_BIN_G=$2 timeout -k 10 300 python tools/ab_sort_stream.py 2>&1 | head -2 | tr '\n' ' ')" || exit 1
  done
done
