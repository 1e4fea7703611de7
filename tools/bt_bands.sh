#!/bin/bash
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -w -DSWR_NSLOT=3 -shared -o software-renderer_amd/lib/libswr_hip.so software-renderer_amd/csrc/swr_kernels.hip software-renderer_amd/csrc/swr_api.hip software-renderer_amd/csrc/swr_upload.hip || exit 1
for m in 0 1; do for g in 256 512; do
  echo "BT=256 G=$g SORT_STREAM=$m: $(SWR_BIN_BT=256 SWR_BIN_G=$g SWR_SORT_STREAM=$m timeout -k 10 300 python tools/band_proxy.py 2>&1 | grep '^N=' | cut -c1-14 | tr '\n' ' ')" || exit 1
done; done
echo "configs BT=256 G=256:"; SWR_BIN_BT=256 timeout -k 10 200 python tools/configs.py 2>&1 | cut -c1-90
