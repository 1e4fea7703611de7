#!/usr/bin/env python3
"""Wall time of swr_scene_upload (H2D copies + validation + triangle-stream build) for cfg4."""
import sys, time
sys.path.insert(0, '.')
import swr_amd
sc = swr_amd.scenes.cfg4_soup()
with swr_amd.Context() as ctx:
    for _ in range(4):
        t0 = time.perf_counter(); ctx.scene_upload(sc.vertices, sc.indices); dt = time.perf_counter() - t0
        print(f"scene_upload 1M triangles: {dt*1e3:.2f} ms")
