#!/usr/bin/env python3
"""Single-frame latency (draw -> sync on an idle context) and short bursts of 20 frames: python tools/lat.py (honours SWR_LIBRARY)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, swr_amd
S = swr_amd.scenes
for name, sc in (("cfg4", S.cfg4_soup()), ("cfg3", S.cfg3_bunny_scale()), ("cfg2", S.cfg2_teapot_scale())):
    with swr_amd.Context() as ctx:
        ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, sc.height)
        for _ in range(10): ctx.draw(sc.transform, sc.flags)
        ctx.sync()
        lat = []
        for _ in range(30):
            t0 = time.perf_counter(); ctx.draw(sc.transform, sc.flags); ctx.sync(); lat.append(time.perf_counter() - t0)
        b = []
        for _ in range(12):
            t0 = time.perf_counter()
            for _ in range(20): ctx.draw(sc.transform, sc.flags)
            ctx.sync(); b.append((time.perf_counter() - t0) / 20)
        bi = []
        for _ in range(12):                       # the driver's shape: a burst of 20 after a short idle (sync + ~1 ms of host work)
            ctx.sync(); time.sleep(0.001)
            t0 = time.perf_counter()
            for _ in range(20): ctx.draw(sc.transform, sc.flags)
            ctx.sync(); bi.append((time.perf_counter() - t0) / 20)
        print(f"   burst of 20 after 1 ms idle: {np.median(bi)*1e6:7.1f} us/frame (min {min(bi)*1e6:.1f})")
        print(f"{os.environ.get('SWR_LIBRARY', 'product')[-16:]:16s} {name}: one frame {np.median(lat)*1e6:7.1f} us   burst of 20: {np.median(b)*1e6:7.1f} us/frame (min {min(b)*1e6:.1f})", flush=True)
