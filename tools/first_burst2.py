#!/usr/bin/env python3
"""Why are the first ~150 frames of a fresh context slower?  python tools/first_burst2.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, swr_amd
S = swr_amd.scenes
sc = S.cfg4_soup()
def burst(ctx, n=20):
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(n): ctx.draw(sc.transform, sc.flags)
    ctx.sync(); return (time.perf_counter() - t0) / n * 1e6
def series(ctx, k=8): return ' '.join('%.1f' % burst(ctx) for _ in range(k))
with swr_amd.Context() as a:
    a.scene_upload(sc.vertices, sc.indices); a.target_set(sc.width, sc.height)
    print("A fresh context          :", series(a), flush=True)
    for _ in range(400): a.draw(sc.transform, sc.flags)
    print("A after 400 more frames  :", series(a, 4), flush=True)
    a.scene_upload(sc.vertices, sc.indices)
    print("A after a new upload     :", series(a), flush=True)
    for _ in range(400): a.draw(sc.transform, sc.flags)
    time.sleep(0.05)
    print("A after 50 ms of idle    :", series(a), flush=True)
    for _ in range(400): a.draw(sc.transform, sc.flags)
    with swr_amd.Context() as b:
        b.scene_upload(sc.vertices, sc.indices); b.target_set(sc.width, sc.height)
        print("B fresh, A alive and hot :", series(b), flush=True)
    print("A again                  :", series(a, 4), flush=True)
