#!/usr/bin/env python3
"""Untimed cfg4 frames, whole frame, default scheduling (best of 5 x 300) — one number for library A/Bs."""
import sys, time
sys.path.insert(0, '.')
import swr_amd
sc = swr_amd.scenes.cfg4_soup()
with swr_amd.Context() as ctx:
    ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, sc.height)
    best = 1e9
    for rep in range(5):
        for _ in range(20): ctx.draw(sc.transform, sc.flags)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(300): ctx.draw(sc.transform, sc.flags)
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / 300)
print("cfg4 whole frame %.1f us" % (best * 1e6))
