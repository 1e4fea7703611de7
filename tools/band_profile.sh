#!/bin/bash
# True kernel durations of a thin band (rank k of N) from rocprofv3's kernel trace, for k_raster ablation variants.
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/band_profile
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-0 11 9 10 4 8}; do
  d=$OUT/v$v
  export SWR_DEBUG_VARIANT=$v
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/band_frames.py ${BAND:-8 4} 200 0 > $d.log 2>&1) || exit 1
  echo "variant $v: $(grep k_raster $d/*/*kernel_stats.csv | cut -d, -f2-4,6-7 | tail -1)"
done
