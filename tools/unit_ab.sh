#!/bin/bash
cd $GRAFT_REPO_ROOT
cp software-renderer_amd/lib/libswr_hip.so.B software-renderer_amd/lib/libswr_hip.so
touch software-renderer_amd/lib/libswr_hip.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "random_soup or cfg or metal" 2>&1 | tail -2 || exit 1
bash tools/n1_ab.sh
