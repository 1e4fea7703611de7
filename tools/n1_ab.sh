#!/bin/bash
# alternating A/B of two prebuilt libraries (lib/libswr_hip.so.A / .B) on untimed cfg4 frames, whole frame + half band
cd $GRAFT_REPO_ROOT/software-renderer_amd/lib
for rep in 1 2 3; do for v in A B; do
  cp libswr_hip.so.$v libswr_hip.so
  echo "$v: $(cd ../.. && timeout -k 10 200 python tools/band_proxy.py 1 2 2>&1 | grep '^N=' | cut -c1-14 | tr '\n' ' ')"
done; done
