#!/bin/bash
# Kernel timeline (rocprofv3 --kernel-trace) of pipelined frames of band 2 of 4, optionally with another binning workgroup size.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/band_trace
rm -rf $OUT && mkdir -p $OUT
cd $R
if [ -n "$BT" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -w -DSWR_BIN_THREADS=$BT -shared -o software-renderer_amd/lib/libswr_hip.so software-renderer_amd/csrc/swr_kernels.hip software-renderer_amd/csrc/swr_api.hip software-renderer_amd/csrc/swr_upload.hip || exit 1
fi
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra PAIRS <<< "${BANDS:-4 2}"      # BANDS="8 4;4 2" = band 4 of 8, then band 2 of 4
for cfg in "${PAIRS[@]}"; do
  IFS=' ' read -r a b <<< "$cfg"; set -- $a $b
  (cd $R && rocprofv3 --kernel-trace --output-format csv -d $OUT/n$1 -- python3 tools/band_frames.py $1 $2 60 1 > $OUT/n$1.log 2>&1) || exit 1
  cp $OUT/n$1/*/*kernel_trace.csv $OUT/n$1_kernel_trace.csv
done
