#!/usr/bin/env python3
"""cfg5 (262 144 triangles, 8K): k_raster alone for depth-only / colour / painter frames — what the colour store costs there."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swr_amd
S = swr_amd.scenes
sc = S.cfg5_sponza_scale()
with swr_amd.Context() as ctx:
    ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, sc.height)
    for name, fl in (("z-test, depth-only", 3), ("z-test, colour + depth", 1), ("painter, colour + depth", 0), ("painter, depth-only", 2)):
        ctx.pipeline_enable(True)
        for _ in range(10): ctx.draw(sc.transform, fl)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(60): ctx.draw(sc.transform, fl)
        ctx.sync(); dt = (time.perf_counter() - t0) / 60
        ctx.pipeline_enable(False); ctx.timing_enable(2); ctx.timing_reset()
        for _ in range(20): ctx.draw(sc.transform, fl)
        sums, n = ctx.timing_totals(); ctx.timing_enable(0)
        print(f"{name:26s} frame {dt*1e6:6.1f} us  k_raster alone {sums['raster_ms']/n*1e3:6.1f} us  binning {sums['setup_bin_ms']/n*1e3:5.1f} us", flush=True)
