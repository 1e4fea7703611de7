#!/bin/bash
# smoke() + the two N = 2 shapes of bench.py on a ONE-GPU box (bands share the GPU: a rehearsal of the code paths, not a result)
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
SWR_BENCH_ALLOW_SHARED=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/reh_group.json 2> gpurun_out/reh_group.err; echo "in-process group rc=$?"
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/reh_ranks.json 2> gpurun_out/reh_ranks.err; echo "two ranks rc=$?"
python3 - <<'PY'
import json
for f in ("gpurun_out/reh_group.json", "gpurun_out/reh_ranks.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["n_gpus"], d["ms_per_step"], d["ms_per_step_first_pass"], d["config"]["sharding"][:100], d["value_host_visible"])
    except Exception as e:
        print(f, "ERR", e)
PY
