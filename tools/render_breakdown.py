#!/usr/bin/env python3
"""Where the time of one synchronous swr_render of cfg4 goes: upload (pageable vs page-locked source arrays), draw,
gather into page-locked images."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import swr_amd
sc = swr_amd.scenes.cfg4_soup()
W, H = sc.width, sc.height
def med(f, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
with swr_amd.Context() as ctx:
    v, i = np.ascontiguousarray(sc.vertices), np.ascontiguousarray(sc.indices)
    print(f"scene: {v.nbytes/1e6:.0f} MB vertices + {i.nbytes/1e6:.0f} MB indices")
    print(f"swr_scene_upload, pageable arrays:    {med(lambda: ctx.scene_upload(v, i)):.2f} ms")
    swr_amd.host_register(v); swr_amd.host_register(i)
    print(f"swr_scene_upload, page-locked arrays: {med(lambda: ctx.scene_upload(v, i)):.2f} ms")
    swr_amd.host_unregister(v); swr_amd.host_unregister(i)
    ctx.target_set(W, H)
    hd = swr_amd.HostImage((H, W), np.float32)
    def frame():
        ctx.draw(sc.transform, sc.flags); ctx.present(None, hd); ctx.present_wait()
    print(f"draw + present(depth) + wait:         {med(frame):.2f} ms")
    def frame2():
        ctx.draw(sc.transform, sc.flags); ctx.sync()
    print(f"draw + sync:                          {med(frame2):.2f} ms")
    hd.free()
