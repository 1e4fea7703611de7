#!/usr/bin/env python3
"""A/B on one box, alternating, of SWR_SORT_STREAM (read once per process -> child processes)."""
import subprocess, sys, os, json
code = r"""
import sys, time
sys.path.insert(0, '.')
import swr_amd
S = swr_amd.scenes
out = []
for name, sc, band in (("cfg4 full", S.cfg4_soup(), None), ("cfg4 band 1/2", S.cfg4_soup(), (2, 0)), ("cfg2 1080p", S.cfg2_teapot_scale(), None)):
    with swr_amd.Context() as ctx:
        ctx.scene_upload(sc.vertices, sc.indices)
        if band: r0, r1 = swr_amd.band_rows(sc.height, *band); ctx.target_set(sc.width, sc.height, r0, r1)
        else: ctx.target_set(sc.width, sc.height)
        best = 1e9
        for rep in range(5):
            for _ in range(20): ctx.draw(sc.transform, sc.flags | 1)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(300): ctx.draw(sc.transform, sc.flags | 1)
            ctx.sync()
            best = min(best, (time.perf_counter() - t0) / 300)
        out.append("%s %.1f" % (name, best * 1e6))
print("; ".join(out))
"""
for rep in range(3):
    for m in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "SWR_SORT_STREAM": m}, capture_output=True, text=True)
        print(f"SWR_SORT_STREAM={m}: {r.stdout.strip()} {r.stderr.strip()[-200:]}", flush=True)
