#!/usr/bin/env python3
"""Timing-only ablations of k_raster<ztest> on cfg4 (results of variant != 0 are INVALID): needs the ablation build
(`make -C software-renderer_amd ablation`), loaded through SWR_LIBRARY; one child process per variant because the
library reads SWR_DEBUG_VARIANT once.  Frame pipelining off: the kernel alone on the GPU.

  1 no LDS atomic | 2 no per-pixel maths (unit fetched, tables read) | 3 producer only (no unit consumed)
  4 no row walk at all | 8 no resolve | 9 no chunk at all (no record loads) | 10 = 4 + 8 | 11 = 9 + 8 (LDS init only)
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys, json
sys.path.insert(0, %r)
import swr_amd
S = swr_amd.scenes
sc = S.cfg4_soup()
with swr_amd.Context() as ctx:
    ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, sc.height)
    ctx.pipeline_enable(False)
    for fl in (sc.flags, 1):
        for _ in range(10): ctx.draw(sc.transform, fl)
        ctx.sync(); ctx.timing_enable(1); ctx.timing_reset()
        for _ in range(60): ctx.draw(sc.transform, fl)
        sums, n = ctx.timing_totals(); ctx.timing_enable(0)
        print(round(sums['raster_ms'] / n * 1e3, 1), end=' ')
print()
""" % ROOT
lib = os.path.join(ROOT, "software-renderer_amd", "lib", "libswr_hip_ablation.so")
for v in (sys.argv[1:] or ["0", "1", "2", "3", "4", "8", "9", "10", "11"]):
    r = subprocess.run([sys.executable, "-c", CODE], env={**os.environ, "SWR_LIBRARY": lib, "SWR_DEBUG_VARIANT": v},
                       capture_output=True, text=True)
    print(f"variant {v:>2}: k_raster us (depth-only, colour+depth) = {r.stdout.strip()} {r.stderr.strip()[-300:]}", flush=True)
