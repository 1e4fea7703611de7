#!/usr/bin/env python3
"""Stage times (pipelining off, events) of the large-triangle scenes of tools/big_ab.py: python tools/big_stages.py [big|occluded|mixed|mid]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, swr_amd
S = swr_amd.scenes
name = sys.argv[1] if len(sys.argv) > 1 else "big"
if name == "big": sc, fl = S.random_soup(300, 1920, 1080, 91, r_ndc=1.5, flags=1, margin=0.5), 1
elif name == "occluded": sc, fl = S.occluded_soup(z_occluder=0.5), 3
elif name == "mid": sc, fl = S.random_soup(20000, 1920, 1080, 92, r_ndc=0.08, flags=1, margin=1.0), 1
else:
    a = S.random_soup(30, 1280, 720, 31, r_ndc=1.0, flags=1, margin=0.9); b = S.random_soup(6000, 1280, 720, 32, r_ndc=0.03, flags=1, margin=1.1)
    sc, fl = S.Scene("mixed", 1280, 720, np.concatenate([a.vertices, b.vertices]), np.concatenate([a.indices, b.indices + a.vertices.shape[0]]), S.identity(), 1), 1
with swr_amd.Context() as ctx:
    ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, sc.height)
    for _ in range(20): ctx.draw(sc.transform, fl)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(100): ctx.draw(sc.transform, fl)
    ctx.sync(); dt = (time.perf_counter() - t0) / 100
    ctx.pipeline_enable(False); ctx.timing_enable(2); ctx.timing_reset()
    for _ in range(20): ctx.draw(sc.transform, fl)
    sums, n = ctx.timing_totals()
    t = ctx.timings()
    print(f"{name:9s} {os.environ.get('SWR_LIBRARY', 'default').split('/')[-1]:16s} frame {dt*1e6:6.1f} us  stages(us)", {k: round(v / n * 1e3, 1) for k, v in sums.items() if k.endswith('_ms')}, "pairs", t["tile_pairs"], flush=True)
