#!/usr/bin/env python3
"""What the driver's bench arguments measure: a FRESH context, 5 warm-up frames, then 20 timed frames — against the same
20 frames once the context has been running for a while.  python tools/first_burst.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, swr_amd
S = swr_amd.scenes
sc = S.cfg4_soup()
def burst(ctx, n=20):
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(n): ctx.draw(sc.transform, sc.flags)
    ctx.sync(); return (time.perf_counter() - t0) / n * 1e6
for rep in range(4):
    with swr_amd.Context() as ctx:
        ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, sc.height)
        for _ in range(5): ctx.draw(sc.transform, sc.flags)
        first = burst(ctx)
        later = [burst(ctx) for _ in range(6)]
        for _ in range(300): ctx.draw(sc.transform, sc.flags)
        warm = [burst(ctx) for _ in range(4)]
        print(f"fresh context: first burst of 20 after 5 warm-up frames {first:6.1f} us/frame | next six {' '.join('%.1f' % v for v in later)} | after 300 more frames {' '.join('%.1f' % v for v in warm)}", flush=True)
