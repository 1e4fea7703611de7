#!/usr/bin/env python3
"""Untimed colour + depth frames (best of 4 x 200): cfg4, cfg5 (8K), cfg2 — for comparing library variants."""
import sys, time
sys.path.insert(0, '.')
import swr_amd
S = swr_amd.scenes
out = []
for name, sc in (("cfg4 colour", S.cfg4_soup(depth_only=False)), ("cfg5 8K", S.cfg5_sponza_scale()), ("cfg3 phong", S.cfg3_phong())):
    with swr_amd.Context() as ctx:
        ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, sc.height)
        if sc.shading is not None: ctx.shading_set(sc.shading)
        best = 1e9
        for rep in range(4):
            for _ in range(10): ctx.draw(sc.transform, 1)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(200): ctx.draw(sc.transform, 1)
            ctx.sync()
            best = min(best, (time.perf_counter() - t0) / 200)
        out.append("%s %.1f" % (name, best * 1e6))
print("; ".join(out))
