#!/bin/bash
cd $GRAFT_REPO_ROOT/software-renderer_amd/lib
for rep in 1 2; do for f in libswr_hip.so.*; do
  cp $f libswr_hip.so; touch libswr_hip.so
  echo "${f##*.}: $(cd ../.. && timeout -k 10 200 python tools/configs.py 2>&1 | grep -i 'phong' | cut -c1-75 | tr '\n' ' ')"
done; done
