#!/bin/bash
# Everything profiles/ and DESIGN.md quote for the final build of a round, in TWO gpurun calls (each well inside the 20-minute limit):
#   tools/round_refresh.sh <prefix> a   -> -m gpu tests, the two bench lines, kernel stats + traffic PMC, SQ counters
#   tools/round_refresh.sh <prefix> b   -> other configs, band proxy, large-triangle scenes, ablations, kernel timelines
# -> gpurun_out/refresh/<prefix>_*; then  python tools/profiles_from_refresh.py <prefix> <profiles prefix> r04
P=${1:-z}; PART=${2:-a}
O=gpurun_out/refresh; mkdir -p $O
if [ "$PART" = a ]; then
  timeout -k 10 600 python -m pytest tests -q -m gpu > $O/${P}_gpu_tests.txt 2>&1; tail -2 $O/${P}_gpu_tests.txt
  timeout -k 10 400 python bench.py > $O/${P}_bench.json 2> $O/bench.err || tail -5 $O/bench.err
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/${P}_bench_driver_args.json 2> $O/bench20.err || tail -5 $O/bench20.err
  tools/profile.sh > $O/profile.log 2>&1 || tail -5 $O/profile.log
  cp gpurun_out/profile/kernel_stats.csv $O/${P}_kernel_stats.csv; cp gpurun_out/profile/traffic_raw.json $O/${P}_traffic_raw_KB.json; cp gpurun_out/profile/stats_bench.json $O/${P}_bench_under_rocprof.json
  tools/sq.sh ${P}sq > $O/${P}_sq_counters.txt 2>&1
  cat $O/${P}_bench.json
else
  timeout -k 10 300 python tools/configs.py > $O/${P}_configs.txt 2>&1
  timeout -k 10 300 python tools/band_proxy.py > $O/${P}_band_proxy.txt 2>&1
  timeout -k 10 300 python tools/big_ab.py software-renderer_amd/lib/libswr_hip.so > $O/${P}_large_triangle_scenes.txt 2>&1
  timeout -k 10 300 python tools/ablate.py > $O/${P}_raster_ablation_timings.txt 2>&1
  VARIANTS="0 1 2 3 4 9 11" tools/sq_variants.sh x > $O/${P}_raster_ablation_valu_counts.txt 2>&1
  tools/gaps.sh pipelined 1 cfg4 60 > /dev/null 2>&1; cp gpurun_out/gaps_pipelined.txt $O/${P}_kernel_trace_pipelined.txt
  tools/gaps.sh band8 1 cfg4 60 4 8 > /dev/null 2>&1; cp gpurun_out/gaps_band8.txt $O/${P}_kernel_trace_band4of8.txt
  tail -n 4 $O/${P}_band_proxy.txt
fi
