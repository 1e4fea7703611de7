#!/usr/bin/env python3
"""Strong-scaling proxy on ONE GPU: time cfg4 frames when this GPU owns band k of N (what rank k of an N-GPU
run does, minus host-side contention).  Speed-up bound at N = t(1 band of 1) / max_k t(band k of N)."""
import sys, time, json
sys.path.insert(0, '.')
import swr_amd
S = swr_amd.scenes
sc = S.cfg4_soup()
flags = sc.flags
res = {}
# test hooks for A/Bs: key=value arguments go to swr_debug_set (e.g. insort=0 k32=0 binmode=1)
KEYS = {"order": 1, "cull": 2, "binmode": 3, "oneshot": 4, "k32": 5, "insort": 6}
hooks = [a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("color=")]
if "color=1" in sys.argv[1:]:
    flags = S.FLAG_DEPTH_TEST            # colour + depth instead of the depth-only headline
parts_arg = [int(x) for x in sys.argv[1:] if "=" not in x]
with swr_amd.Context() as ctx:
    for k, v in hooks:
        ctx.debug_set(KEYS[k], int(v))
    ctx.scene_upload(sc.vertices, sc.indices)
    for parts in (parts_arg or (1, 2, 4, 8)):
        worst = 0.0
        for k in sorted({0, parts // 2, parts - 1}):
            r0, r1 = swr_amd.band_rows(sc.height, parts, k)
            ctx.target_set(sc.width, sc.height, r0, r1)
            dt = 1e9
            for rep in range(4):                      # best of 4 x 200 frames (the first batch also warms the clocks)
                for _ in range(20): ctx.draw(sc.transform, flags)
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(200): ctx.draw(sc.transform, flags)
                ctx.sync()
                dt = min(dt, (time.perf_counter() - t0) / 200)
            ctx.pipeline_enable(False); ctx.timing_enable(2); ctx.timing_reset()
            for _ in range(20): ctx.draw(sc.transform, flags)
            sums, n = ctx.timing_totals()
            ctx.timing_enable(0); ctx.pipeline_enable(True)
            st = {k2: round(v / n * 1e3, 1) for k2, v in sums.items() if k2.endswith('_ms')}
            print(f"N={parts} band {k} rows [{r0},{r1}): {dt*1e6:.1f} us/frame; stages(us) {json.dumps(st)}", flush=True)
            worst = max(worst, dt)
        res[parts] = worst
for parts, t in res.items():
    if 1 in res:
        print(f"N={parts}: {t*1e6:.1f} us -> speed-up {res[1]/t:.2f}x, efficiency {res[1]/t/parts:.2f}")
