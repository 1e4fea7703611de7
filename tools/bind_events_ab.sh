#!/bin/bash
# A/B: bin_done / ras_done bound to their kernels (hipExtLaunchKernelGGL stop event; default) vs recorded behind them (SWR_BIND_EVENTS=0)
for rep in 1 2 3; do for be in 1 0; do
  echo "SWR_BIND_EVENTS=$be: $(SWR_BIND_EVENTS=$be timeout -k 10 120 python bench.py --steps 300 --no-cpu-baseline --no-extra | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step", d["ms_per_step"], "latency", d["latency_ms"], "raster(pipelined)", d["roofline"]["avg_launch_ms"])')"
done; done
for be in 1 0; do echo "== band proxy, SWR_BIND_EVENTS=$be"; SWR_BIND_EVENTS=$be timeout -k 10 200 python tools/band_proxy.py | tail -4; done
for be in 1 0; do echo "== configs, SWR_BIND_EVENTS=$be"; SWR_BIND_EVENTS=$be timeout -k 10 200 python tools/configs.py | cut -c1-75; done
echo "== event waits on the streams + bound events"; SWR_EVENT_WAITS=1 timeout -k 10 120 python bench.py --steps 300 --no-cpu-baseline --no-extra | cut -c1-200
