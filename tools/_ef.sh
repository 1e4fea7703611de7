for rep in 1 2; do for v in 0 0x40000000 0x20000000; do
  echo "SWR_EVENT_FLAGS=$v: $(SWR_EVENT_FLAGS=$v timeout -k 10 120 python bench.py --steps 300 --no-cpu-baseline --no-extra 2>&1 | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms/step", d["ms_per_step"])')"
done; done
SWR_EVENT_FLAGS=0x20000000 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
