#!/usr/bin/env python3
"""Is a thin band's frame rate bound by the host's enqueue rate?  Time 200 draws up to the last enqueue, and up to
the sync, for band 4 of 8 and for the full frame."""
import sys, time
sys.path.insert(0, '.')
import swr_amd
sc = swr_amd.scenes.cfg4_soup()
with swr_amd.Context() as ctx:
    ctx.scene_upload(sc.vertices, sc.indices)
    for parts, k in ((8, 4), (1, 0)):
        r0, r1 = swr_amd.band_rows(sc.height, parts, k)
        ctx.target_set(sc.width, sc.height, r0, r1)
        for _ in range(20): ctx.draw(sc.transform, sc.flags)
        ctx.sync()
        for rep in range(3):                     # 40 draws: below the 64-frame bound on frames waiting for the helper threads
            t0 = time.perf_counter()
            for _ in range(40): ctx.draw(sc.transform, sc.flags)
            t1 = time.perf_counter()
            ctx.sync()
            print(f"band {k} of {parts}: caller-side cost of swr_draw {1e6*(t1-t0)/40:.1f} us/frame (40 un-waited draws)", flush=True)
        for rep in range(3):                     # 200 draws: the caller is paced by the GPU once it is 64 frames ahead
            t0 = time.perf_counter()
            for _ in range(200): ctx.draw(sc.transform, sc.flags)
            t1 = time.perf_counter()
            ctx.sync()
            t2 = time.perf_counter()
            print(f"band {k} of {parts}: enqueue {1e6*(t1-t0)/200:.1f} us/frame, enqueue+drain {1e6*(t2-t0)/200:.1f} us/frame", flush=True)
