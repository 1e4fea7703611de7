#!/bin/bash
# alternate the prebuilt libraries lib/libswr_hip.so.<TAG> under the command given as arguments
cd $GRAFT_REPO_ROOT/software-renderer_amd/lib
for rep in 1 2 3; do for f in libswr_hip.so.*; do
  cp $f libswr_hip.so; touch libswr_hip.so
  echo "${f##*.}: $(cd ../.. && timeout -k 10 300 "$@" 2>&1 | tail -1)"
done; done
