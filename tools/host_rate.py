#!/usr/bin/env python3
"""Is a short frame bound by the host's launch rate?  Per draw: the time swr_draw_primitives takes to return (enqueue only,
the GPU far behind or far ahead) next to the frame period with the GPU in the loop.  Thin bands of cfg4 and the small configs."""
import sys, time
sys.path.insert(0, '.')
import swr_amd
S = swr_amd.scenes

def run(name, sc, r0=None, r1=None, flags=None):
    flags = sc.flags if flags is None else flags
    with swr_amd.Context() as ctx:
        ctx.scene_upload(sc.vertices, sc.indices)
        ctx.target_set(sc.width, sc.height, 0 if r0 is None else r0, r1)
        for _ in range(50): ctx.draw(sc.transform, flags)
        ctx.sync()
        best_q = best_t = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            for _ in range(400): ctx.draw(sc.transform, flags)
            t1 = time.perf_counter()
            ctx.sync()
            t2 = time.perf_counter()
            best_q = min(best_q, (t1 - t0) / 400); best_t = min(best_t, (t2 - t0) / 400)
        print(f"{name:28s} enqueue {best_q*1e6:6.2f} us/draw   period {best_t*1e6:6.2f} us/frame", flush=True)

c4 = S.cfg4_soup()
for parts, k in ((1, 0), (4, 2), (8, 4), (16, 8)):
    r0, r1 = swr_amd.band_rows(c4.height, parts, k)
    run(f"cfg4 band {k} of {parts}", c4, r0, r1)
run("cfg2", S.cfg2_teapot_scale())
run("cfg3", S.cfg3_bunny_scale())
