#!/bin/bash
# alternating comparison of prebuilt libraries lib/libswr_hip.so.<TAG> on untimed cfg4 frames (whole frame + half band)
cd $GRAFT_REPO_ROOT/software-renderer_amd/lib
for rep in 1 2; do for f in libswr_hip.so.*; do
  cp $f libswr_hip.so; touch libswr_hip.so
  echo "${f##*.}: $(cd ../.. && timeout -k 10 200 python tools/band_proxy.py 1 2 2>&1 | grep '^N=' | cut -c1-14 | tr '\n' ' ')"
done; done
