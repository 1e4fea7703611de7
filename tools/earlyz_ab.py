#!/usr/bin/env python3
"""A/B of hierarchical early-z (SWR_EARLYZ=0 build in lib/ab_noez.so vs the product library): cfg4 (depth complexity 4.8,
no occluders: the test can only cost) and the occluded soups (a screen-filling quad at z = 0.5 / 0.02)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys
sys.path.insert(0, %r)
import swr_amd
S = swr_amd.scenes
for name, sc in (("cfg4", S.cfg4_soup()), ("occluded z=0.5", S.occluded_soup(z_occluder=0.5)), ("occluded z=0.02", S.occluded_soup(z_occluder=0.02)),
                 ("cfg5", S.cfg5_sponza_scale()), ("cfg3", S.cfg3_bunny_scale())):
    with swr_amd.Context() as ctx:
        ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, sc.height)
        import time
        for _ in range(20): ctx.draw(sc.transform, sc.flags)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(200): ctx.draw(sc.transform, sc.flags)
        ctx.sync(); dt = (time.perf_counter() - t0) / 200
        ctx.pipeline_enable(False)
        for _ in range(5): ctx.draw(sc.transform, sc.flags)
        ctx.sync(); ctx.timing_enable(1); ctx.timing_reset()
        for _ in range(40): ctx.draw(sc.transform, sc.flags)
        sums, n = ctx.timing_totals(); ctx.timing_enable(0)
        print(f"  {name:16s} frame {dt*1e6:7.1f} us   k_raster alone {sums['raster_ms']/n*1e3:7.1f} us", flush=True)
""" % ROOT
for tag, lib in (("early-z ON ", "libswr_hip.so"), ("early-z OFF", "ab_noez.so"), ("early-z ON ", "libswr_hip.so"), ("early-z OFF", "ab_noez.so")):
    print(tag, flush=True)
    r = subprocess.run([sys.executable, "-c", CODE], env={**os.environ, "SWR_LIBRARY": os.path.join(ROOT, "software-renderer_amd", "lib", lib)}, capture_output=True, text=True)
    print(r.stdout, r.stderr[-300:], flush=True)
