"""The N > 1 path with the HIP kernels under the ranks (the CPU-suite gloo test renders every band with
the oracle).  A one-GPU box cannot give every rank its own device, so the ranks SHARE cuda:0 and gloo carries the barrier /
MAX — the same code path bench.py takes under a launcher (one process per band, tile-row shards from swr_band_rows, every
rank presenting its band into ONE page-locked /dev/shm image, no data-path collective), rehearsed, not measured:
  * world 2 and 3: each rank renders its band of a depth-only (32-bit depth keys) and of a colour frame through the C-ABI
    and swr_present's it into the shared image; the assembled images must equal the oracle's bit for bit;
  * bench.py itself under `python -m torch.distributed.run --nproc-per-node 2`: exits 0 and labels the line a rehearsal."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, NTRI = 960, 544, 40000


def _worker(rank, world, port, shm_prefix):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import swr_amd
    S = swr_amd.scenes
    s = S.cfg4_soup(ntri=NTRI, width=W, height=H, r_ndc=0.02, depth_only=False)
    r0, r1 = bench.shard_rows(swr_amd, H, world, rank)
    color = np.memmap(shm_prefix + "_color.bin", dtype=np.uint8, mode="r+", shape=(H, W, 4))
    depth = np.memmap(shm_prefix + "_depth.bin", dtype=np.float32, mode="r+", shape=(H, W))
    depth2 = np.memmap(shm_prefix + "_depth2.bin", dtype=np.float32, mode="r+", shape=(H, W))
    for img in (color, depth, depth2):
        swr_amd.host_register(img)
    with swr_amd.Context(0) as ctx:                       # every rank on cuda:0: a rehearsal of the launcher path
        ctx.scene_upload(s.vertices, s.indices)
        ctx.target_set(W, H, r0, r1)
        assert [(a, b) for _, a, b in ctx.bands()] == [(r0, r1)]
        dist.barrier()
        for _ in range(3):                                # a short burst, like the timed region of bench.py
            ctx.draw(s.transform, S.FLAG_DEPTH_TEST)
            ctx.present(color, depth)
        ctx.present_wait()
        for _ in range(3):
            ctx.draw(s.transform, S.FLAG_DEPTH_TEST | S.FLAG_NO_COLOR)
            ctx.present(None, depth2)
        ctx.present_wait()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the MAX over ranks of bench.py's timing
        assert t.item() == float(world)
    for img in (color, depth, depth2):
        swr_amd.host_unregister(img)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_hip_bands_from_gloo_ranks_assemble_in_one_shared_image(swr, oracle, world):
    import torch.multiprocessing as mp
    prefix = f"/dev/shm/swr_test_{os.getpid()}_{world}"
    shapes = {"_color.bin": W * H * 4, "_depth.bin": W * H * 4, "_depth2.bin": W * H * 4}
    try:
        for suffix, nbytes in shapes.items():
            with open(prefix + suffix, "wb") as f:
                f.truncate(nbytes)
        port = 29500 + (os.getpid() % 2000) + 10 * world
        mp.spawn(_worker, args=(world, port, prefix), nprocs=world, join=True)
        s = swr.scenes.cfg4_soup(ntri=NTRI, width=W, height=H, r_ndc=0.02, depth_only=False)
        ref_c, ref_d, _, rc = oracle.render(s.vertices, s.indices, s.transform, W, H, 1 | oracle.TINV_PER_TRIANGLE)
        assert rc == 0
        color = np.fromfile(prefix + "_color.bin", dtype=np.uint8).reshape(H, W, 4)
        depth = np.fromfile(prefix + "_depth.bin", dtype=np.float32).reshape(H, W)
        depth2 = np.fromfile(prefix + "_depth2.bin", dtype=np.float32).reshape(H, W)
        assert np.array_equal(color, ref_c)
        assert np.array_equal(depth.view(np.uint32), ref_d.view(np.uint32))
        assert np.array_equal(depth2.view(np.uint32), ref_d.view(np.uint32))       # the depth-only frames (32-bit keys)
    finally:
        for suffix in shapes:
            if os.path.exists(prefix + suffix):
                os.remove(prefix + suffix)


def test_bench_py_under_the_launcher_with_two_ranks_sharing_the_gpu(swr):
    """bench.py's launcher path end to end (init_process_group, barrier + MAX over ranks, the /dev/shm image every rank presents
    its band into): two ranks on one GPU = a labelled rehearsal, not a result."""
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--triangles", "60000", "--width", "1280", "--height", "720", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert "REHEARSAL" in d["config"]["sharding"] and "no collective" in d["config"]["sharding"]
    assert d["extra"]["host_visible"]["Mpixels_per_s"] > 0 and "cpu_baseline" not in d
