#!/usr/bin/env python3
"""Timing sanity of the other BASELINE configs (cfg2/3/5) on one GPU; parity is in tests/."""
import sys, time, json
sys.path.insert(0, '.')  # run from the repo root: python tools/configs.py
import swr_amd
S = swr_amd.scenes
for name, scene, flags in (("cfg2 torus 6320 tris 1080p painter", S.cfg2_teapot_scale(), 0),
                           ("cfg2 torus 6320 tris 1080p z-test", S.cfg2_teapot_scale(), 1),
                           ("cfg3 torus 69451 tris 4K z-test", S.cfg3_bunny_scale(), 1),
                           ("cfg3 torus 69451 tris 4K per-pixel Phong + z", S.cfg3_phong(), 1),
                           ("cfg5 grid 262144 tris 8K z-test", S.cfg5_sponza_scale(), 1),
                           ("cfg5 grid 262144 tris 8K textured + Phong + z", S.cfg5_textured(), 1),
                           ("big: 300 screen-filling tris 1080p z", S.random_soup(300, 1920, 1080, 91, r_ndc=1.5, flags=1, margin=0.5), 1),
                           ("cfg4 soup 1M tris 4K METAL rules (colour+depth)", S.cfg4_soup(depth_only=False), 4),
                           ("cfg4 soup 1M tris 4K CPU rules colour+depth", S.cfg4_soup(depth_only=False), 1),
                           ("app sphere 338 tris 512^2 METAL rules", None, 4),
                           ("app sphere 338 tris 512^2 z", None, 1)):
    if scene is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("fl", "examples/frame_loop.py"); fl = importlib.util.module_from_spec(spec); spec.loader.exec_module(fl)
        v, i = fl.sphere_mesh(); W = H = 512; m = S.app_transform(0.5)
    else:
        v, i, W, H, m = scene.vertices, scene.indices, scene.width, scene.height, scene.transform
    with swr_amd.Context() as ctx:
        ctx.scene_upload(v, i); ctx.target_set(W, H)
        if scene is not None and scene.shading is not None:
            ctx.shading_set(scene.shading)
        for _ in range(5): ctx.draw(m, flags)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(50): ctx.draw(m, flags)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 50
        ctx.pipeline_enable(False); ctx.timing_enable(2); ctx.timing_reset()     # stages serialised on one stream
        for _ in range(10): ctx.draw(m, flags)
        sums, n = ctx.timing_totals()
        print(f"{name}: {dt*1e3:.4f} ms/frame = {W*H/dt/1e6:.0f} Mpix/s; stages(ms) " +
              json.dumps({k: round(x / n, 4) for k, x in sums.items() if k.endswith('_ms')}) + f" pairs={ctx.timings()['tile_pairs']}")
