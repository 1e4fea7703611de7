#!/usr/bin/env python3
"""Draw N frames of a named config (for rocprofv3 --kernel-trace --stats): python tools/frames.py cfg5 [N] [band k of n]"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swr_amd
S = swr_amd.scenes
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sc = {"cfg4": lambda: S.cfg4_soup(), "cfg5": lambda: S.cfg5_sponza_scale(), "cfg3": lambda: S.cfg3_bunny_scale(),
      "cfg2": lambda: S.cfg2_teapot_scale(), "cfg4c": lambda: S.cfg4_soup(depth_only=False),
      "big": lambda: S.random_soup(300, 1920, 1080, 91, r_ndc=1.5, flags=1, margin=0.5),
      "mid": lambda: S.random_soup(5000, 1920, 1080, 93, r_ndc=0.16, flags=1, margin=1.0)}[name]()
with swr_amd.Context() as ctx:
    # swr_debug_set hooks for A/Bs under the profiler: SWR_AB_HOOKS="insort=0 k32=0"
    for kv in os.environ.get("SWR_AB_HOOKS", "").split():
        k, v = kv.split("=")
        ctx.debug_set({"order": 1, "cull": 2, "binmode": 3, "oneshot": 4, "k32": 5, "insort": 6}[k], int(v))
    ctx.scene_upload(sc.vertices, sc.indices)
    r0, r1 = 0, sc.height
    if len(sys.argv) > 4:
        r0, r1 = swr_amd.band_rows(sc.height, int(sys.argv[4]), int(sys.argv[3]))
    ctx.target_set(sc.width, sc.height, r0, r1)
    ctx.draw(sc.transform, sc.flags)
    ctx.sync()                       # (bins grown if the first frame overflowed them: the frames below are real)
    for _ in range(n):
        ctx.draw(sc.transform, sc.flags)
    ctx.sync()
    print(name, "rows", r0, r1, ctx.timings())
