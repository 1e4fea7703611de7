#!/usr/bin/env python3
"""The reference-shaped call, cold (scene_id 0: upload every call): where its time goes.  python tools/cold_render.py [n]
(SWR_SORT=0 in the environment: the triangle stream keeps index order — no Morton sort at upload)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, swr_amd
S = swr_amd.scenes
sc = S.cfg4_soup()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
depth = swr_amd.HostImage((sc.height, sc.width), np.float32)
with swr_amd.Context() as ctx:
    rows = []
    for _ in range(n + 2):
        t0 = time.perf_counter()
        ctx.render(sc.vertices, sc.indices, sc.transform, sc.width, sc.height, sc.flags, color=None, depth=depth.array, scene_id=0)
        rows.append(dict(ctx.render_timings(), wall_ms=(time.perf_counter() - t0) * 1e3))
    rows = rows[2:]
    print("SWR_SORT=%s" % os.environ.get("SWR_SORT", "1"), {k: round(float(np.median([r[k] for r in rows])), 3) for k in rows[0]})
depth.free()
