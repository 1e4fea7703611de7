#!/bin/bash
# A/B: host-paced ordering (ras_worker polls hipEventQuery, no wait packets; default) vs event waits on the streams (SWR_EVENT_WAITS=1)
for rep in 1 2 3; do for ew in 0 1; do
  echo "SWR_EVENT_WAITS=$ew: $(SWR_EVENT_WAITS=$ew timeout -k 10 120 python bench.py --steps 300 --no-cpu-baseline --no-extra | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step", d["ms_per_step"], "latency", d["latency_ms"], "raster(pipelined)", d["roofline"]["avg_launch_ms"])')"
done; done
for ew in 0 1; do echo "== band proxy, SWR_EVENT_WAITS=$ew"; SWR_EVENT_WAITS=$ew timeout -k 10 200 python tools/band_proxy.py | tail -4; done
for ew in 0 1; do echo "== configs, SWR_EVENT_WAITS=$ew"; SWR_EVENT_WAITS=$ew timeout -k 10 200 python tools/configs.py | cut -c1-75; done
