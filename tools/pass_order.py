#!/usr/bin/env python3
"""Does an instrumented pass leave anything behind?  One context, a sequence of passes of 200 frames each, some with the
sampled event timing bench.py's roofline pass uses, some with single synced frames in between.  python tools/pass_order.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, swr_amd
S = swr_amd.scenes
sc = S.cfg4_soup()
def plain(ctx, n=200):
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(n): ctx.draw(sc.transform, sc.flags)
    ctx.sync(); return (time.perf_counter() - t0) / n * 1e6
def sampled(ctx, n=200, every=8):
    ctx.timing_sample(every); ctx.timing_enable(1); ctx.timing_reset()
    v = plain(ctx, n)
    ctx.timing_totals(); ctx.timing_enable(0)
    return v
def singles(ctx, n=12):
    for _ in range(n):
        ctx.draw(sc.transform, sc.flags); ctx.sync()
with swr_amd.Context() as ctx:
    ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, sc.height)
    row = []
    for name in "plain plain plain singles plain sampled plain plain sampled plain singles plain plain".split():
        if name == "plain": row.append("plain %.1f" % plain(ctx))
        elif name == "sampled": row.append("sampled %.1f" % sampled(ctx))
        else: singles(ctx); row.append("12 single frames")
    print(" | ".join(row), flush=True)
