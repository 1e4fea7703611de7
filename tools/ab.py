#!/usr/bin/env python3
"""Alternating A/B of library builds on one box: python tools/ab.py [--scene cfg4|cfg4c|cfg5|cfg3|metal|metalc|cfg5t|cfg3p|micro|micro2] [--reps 3] lib/a.so lib/b.so ...
Per build: pipelined ms/frame (300 untimed frames), k_raster alone (pipelining off, events around the kernel)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
import numpy as np, swr_amd
S = swr_amd.scenes
name = sys.argv[1]
sc = {"cfg4": lambda: S.cfg4_soup(), "cfg4c": lambda: S.cfg4_soup(depth_only=False), "cfg5": lambda: S.cfg5_sponza_scale(),
      "cfg3": lambda: S.cfg3_bunny_scale(), "cfg2": lambda: S.cfg2_teapot_scale(), "metal": lambda: S.cfg4_soup(),
      "metalc": lambda: S.cfg4_soup(depth_only=False), "cfg5t": lambda: S.cfg5_textured(), "cfg3p": lambda: S.cfg3_phong(),
      "micro": lambda: S.cfg4_soup(ntri=1_000_000, width=1920, height=1080, r_ndc=0.004, depth_only=False),   # ~1000 two-pixel triangles per tile
      "micro2": lambda: S.cfg4_soup(ntri=500_000, width=1920, height=1080, r_ndc=0.008, depth_only=False),
      "occluded": lambda: S.occluded_soup(z_occluder=0.5), "occluded_near": lambda: S.occluded_soup(z_occluder=0.1)}[name]()
flags = S.FLAG_METAL_RULES | S.FLAG_NO_COLOR if name == "metal" else (S.FLAG_METAL_RULES if name == "metalc" else sc.flags)
with swr_amd.Context() as ctx:
    ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, sc.height)
    if sc.shading is not None: ctx.shading_set(sc.shading)
    for _ in range(30): ctx.draw(sc.transform, flags)
    ctx.sync()
    N = 300
    t0 = time.perf_counter()
    for _ in range(N): ctx.draw(sc.transform, flags)
    ctx.sync()
    ms = (time.perf_counter() - t0) / N * 1e3
    ctx.pipeline_enable(False); ctx.timing_enable(1); ctx.timing_reset()
    for _ in range(40): ctx.draw(sc.transform, flags)
    ctx.sync()
    t, n = ctx.timing_totals()
    print(json.dumps({"ms": ms, "raster_us": t["raster_ms"] / max(n, 1) * 1e3}))
''' % ROOT
args = sys.argv[1:]
scene, reps = "cfg4", 3
while args and args[0].startswith("--"):
    if args[0] == "--scene": scene = args[1]
    if args[0] == "--reps": reps = int(args[1])
    args = args[2:]
res = {l: [] for l in args}
for r in range(reps):
    for lib in args:
        env = dict(os.environ, SWR_LIBRARY=os.path.join(ROOT, lib))
        p = subprocess.run([sys.executable, "-c", CHILD, scene], env=env, capture_output=True, text=True)
        try:
            res[lib].append(json.loads(p.stdout.strip().splitlines()[-1]))
        except Exception:
            print(lib, "FAILED", p.stderr[-300:]); continue
for lib in args:
    v = res[lib]
    if not v: continue
    print(f"{scene:6s} {lib:40s} pipelined ms/frame {' '.join('%.4f' % x['ms'] for x in v)}  (min {min(x['ms'] for x in v):.4f})   "
          f"k_raster alone us {' '.join('%.1f' % x['raster_us'] for x in v)}  (min {min(x['raster_us'] for x in v):.1f})", flush=True)
