#!/usr/bin/env python3
"""Draw `frames` cfg4 frames of band k of N on one GPU (for rocprofv3 --kernel-trace --stats):
    python3 tools/band_frames.py N k [frames] [pipeline 0/1]"""
import sys
sys.path.insert(0, '.')
import swr_amd
N, k = int(sys.argv[1]), int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 100
sc = swr_amd.scenes.cfg4_soup()
with swr_amd.Context() as ctx:
    if len(sys.argv) > 4:
        ctx.pipeline_enable(bool(int(sys.argv[4])))
    ctx.scene_upload(sc.vertices, sc.indices)
    r0, r1 = swr_amd.band_rows(sc.height, N, k)
    ctx.target_set(sc.width, sc.height, r0, r1)
    for _ in range(frames):
        ctx.draw(sc.transform, sc.flags)
    ctx.sync()
print("done", N, k, frames)
