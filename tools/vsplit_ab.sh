#!/bin/bash
# A/B: 2-4 workgroups per tile on small grids (default) vs one workgroup per tile always (SWR_VSPLIT=0)
for vs in -1 0; do
  if [ $vs = -1 ]; then unset SWR_VSPLIT; else export SWR_VSPLIT=$vs; fi
  echo "== band proxy, SWR_VSPLIT=${SWR_VSPLIT:-auto}"; timeout -k 10 200 python tools/band_proxy.py
  echo "== configs, SWR_VSPLIT=${SWR_VSPLIT:-auto}"; timeout -k 10 200 python tools/configs.py | cut -c1-260
done
