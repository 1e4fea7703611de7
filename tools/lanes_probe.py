#!/usr/bin/env python3
"""Feasibility probe for frame 'lanes' (every frame's binning + raster back to back on ONE stream, consecutive frames on
different streams): L contexts created with SWR_PIPELINE=0 (one stream each, no helper threads, no events) render the SAME band
round-robin; the aggregate frame period is what L lanes inside one context would give.
    python tools/lanes_probe.py [parts] [lanes ...]        e.g.  8 1 2 3 4"""
import os, sys, time
os.environ["SWR_PIPELINE"] = "0"
sys.path.insert(0, '.')
import swr_amd
S = swr_amd.scenes
sc = S.cfg4_soup()
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lanes_list = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 4]
k = parts // 2
r0, r1 = swr_amd.band_rows(sc.height, parts, k)
for L in lanes_list:
    ctxs = [swr_amd.Context() for _ in range(L)]
    for c in ctxs:
        c.scene_upload(sc.vertices, sc.indices); c.target_set(sc.width, sc.height, r0, r1)
    best = 1e9
    for rep in range(4):
        for i in range(30): ctxs[i % L].draw(sc.transform, sc.flags)
        for c in ctxs: c.sync()
        t0 = time.perf_counter()
        for i in range(300): ctxs[i % L].draw(sc.transform, sc.flags)
        for c in ctxs: c.sync()
        best = min(best, (time.perf_counter() - t0) / 300)
    print(f"band {k} of {parts}, {L} lane(s): {best*1e6:.1f} us/frame", flush=True)
    for c in ctxs: c.close()
