#!/usr/bin/env python3
"""k_raster alone + frame time on the 300-screen-filling-triangles scene: python tools/big_one.py (honours SWR_LIBRARY, SWR_BIN_MODE)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swr_amd
S = swr_amd.scenes
sc = S.random_soup(300, 1920, 1080, 91, r_ndc=1.5, flags=1, margin=0.5)
with swr_amd.Context() as ctx:
    ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, sc.height)
    ctx.draw(sc.transform, 1); ctx.sync()
    for _ in range(20): ctx.draw(sc.transform, 1)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(100): ctx.draw(sc.transform, 1)
    ctx.sync(); dt = (time.perf_counter() - t0) / 100
    ctx.pipeline_enable(False)
    for _ in range(5): ctx.draw(sc.transform, 1)
    ctx.sync(); ctx.timing_enable(2); ctx.timing_reset()
    for _ in range(30): ctx.draw(sc.transform, 1)
    sums, n = ctx.timing_totals(); ctx.timing_enable(0)
    print(os.environ.get("SWR_LIBRARY", "product")[-20:], os.environ.get("SWR_BIN_MODE", "-"), f"frame {dt*1e6:7.1f} us", {k: round(v / n * 1e3, 1) for k, v in sums.items() if k.endswith("_ms")}, ctx.timings()["tile_pairs"], flush=True)
