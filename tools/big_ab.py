#!/usr/bin/env python3
"""A/B of library builds on scenes with large triangles: python tools/big_ab.py lib1.so lib2.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys, time
sys.path.insert(0, %r)
import swr_amd
S = swr_amd.scenes
for name, sc, fl in (("big 300 1080p", S.random_soup(300, 1920, 1080, 91, r_ndc=1.5, flags=1, margin=0.5), 1), ("occluded z=0.5", S.occluded_soup(z_occluder=0.5), 3),
                 ("20k tris of ~100 px 1080p", S.random_soup(20000, 1920, 1080, 92, r_ndc=0.08, flags=1, margin=1.0), 1), ("5k tris of ~200 px 1080p", S.random_soup(5000, 1920, 1080, 93, r_ndc=0.16, flags=1, margin=1.0), 1), ("cfg5", S.cfg5_sponza_scale(), 1), ("cfg3", S.cfg3_bunny_scale(), 1), ("cfg4", S.cfg4_soup(), 3), ("mixed 30 big + 6k small", None, 1)):
    if sc is None:
        import numpy as np
        a = S.random_soup(30, 1280, 720, 31, r_ndc=1.0, flags=1, margin=0.9); b = S.random_soup(6000, 1280, 720, 32, r_ndc=0.03, flags=1, margin=1.1)
        sc = S.Scene("mixed", 1280, 720, np.concatenate([a.vertices, b.vertices]), np.concatenate([a.indices, b.indices + a.vertices.shape[0]]), S.identity(), 1)
    with swr_amd.Context() as ctx:
        ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, sc.height)
        ctx.draw(sc.transform, fl); ctx.sync()
        for _ in range(20): ctx.draw(sc.transform, fl)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(100): ctx.draw(sc.transform, fl)
        ctx.sync(); dt = (time.perf_counter() - t0) / 100
        ctx.pipeline_enable(False)
        for _ in range(5): ctx.draw(sc.transform, fl)
        ctx.sync(); ctx.timing_enable(1); ctx.timing_reset()
        for _ in range(30): ctx.draw(sc.transform, fl)
        sums, n = ctx.timing_totals(); ctx.timing_enable(0)
        print(f"  {name:24s} frame {dt*1e6:7.1f} us   k_raster alone {sums['raster_ms']/n*1e3:7.1f} us", flush=True)
""" % ROOT
for lib in sys.argv[1:]:
    print("==", lib, flush=True)
    r = subprocess.run([sys.executable, "-c", CODE], env={**os.environ, "SWR_LIBRARY": os.path.join(ROOT, lib)}, capture_output=True, text=True)
    print(r.stdout, r.stderr[-300:], flush=True)
