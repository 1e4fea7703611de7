#!/usr/bin/env python3
"""One frame at a time with the host idle in between (an interactive app): swr_draw -> swr_sync latency when the
context's helper threads have gone to sleep, vs a tight loop; cfg4 and the app's sphere."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import swr_amd
S = swr_amd.scenes
for name, sc in (("cfg4", S.cfg4_soup()), ("cfg2", S.cfg2_teapot_scale())):
    with swr_amd.Context() as ctx:
        ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, sc.height)
        for _ in range(20): ctx.draw(sc.transform, sc.flags)
        ctx.sync()
        for pause in (0.0, 0.0005, 0.005, 0.03):
            ts = []
            for _ in range(40):
                if pause: time.sleep(pause)
                t0 = time.perf_counter(); ctx.draw(sc.transform, sc.flags); ctx.sync(); ts.append(time.perf_counter() - t0)
            print(f"{name}: pause {pause*1e3:5.1f} ms between frames: draw->sync median {np.median(ts)*1e6:7.1f} us  (min {min(ts)*1e6:.1f})", flush=True)
