"""N > 1 path on CPU: world_size-2 (and 3) gloo processes shard the framebuffer into tile-row
bands exactly as bench.py does (shard_rows -> swr_band_rows), render their band, and the bands
assemble into the full frame with no data-path collective (bands are disjoint).
There is no GPU here, so each rank's band is produced by the CPU oracle standing in for the
kernel; what is under test is the sharding arithmetic, the band-local addressing contract of
swr_read_* (rows [row_begin,row_end) of a full-size image) and the rank plumbing."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import swr_amd
    from oracle import oracle

    s = swr_amd.scenes.cfg4_soup(ntri=3000, width=320, height=200, r_ndc=0.05, depth_only=False)
    r0, r1 = bench.shard_rows(swr_amd, s.height, world, rank)
    color = np.zeros((s.height, s.width, 4), dtype=np.uint8)
    depth = np.zeros((s.height, s.width), dtype=np.float32)
    _, _, st, rc = oracle.render(s.vertices, s.indices, s.transform, s.width, s.height, s.flags,
                                 r0, r1, color, depth)
    assert rc == 0
    # timing protocol of bench.py: barrier, local time, MAX over ranks
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    # gather the band edges only (the pixels themselves go through host memory, not a collective)
    edges = [None] * world
    dist.all_gather_object(edges, (r0, r1))
    np.savez(os.path.join(out_dir, f"band{rank}.npz"), color=color[r0:r1], depth=depth[r0:r1], r0=r0, r1=r1)
    if rank == 0:
        assert edges[0][0] == 0 and edges[-1][1] == s.height
        for a, b in zip(edges, edges[1:]):
            assert a[1] == b[0]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bands_from_gloo_ranks_assemble(tmp_path, world):
    sys.path.insert(0, ROOT)
    import swr_amd
    from oracle import oracle
    swr_amd.build()
    oracle.build()
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    s = swr_amd.scenes.cfg4_soup(ntri=3000, width=320, height=200, r_ndc=0.05, depth_only=False)
    ref_c, ref_d, _, _ = oracle.render_scene(s)
    color = np.zeros_like(ref_c)
    depth = np.zeros_like(ref_d)
    for k in range(world):
        b = np.load(tmp_path / f"band{k}.npz")
        color[int(b["r0"]):int(b["r1"])] = b["color"]
        depth[int(b["r0"]):int(b["r1"])] = b["depth"]
    assert np.array_equal(color, ref_c)
    assert np.array_equal(depth.view(np.uint32), ref_d.view(np.uint32))
