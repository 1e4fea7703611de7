#!/usr/bin/env python3
"""How much of k_raster's time is wave quantisation?  Same soup density, targets whose tile count is / is not a multiple of
the 1280 resident workgroups (256 CUs x 5): us per 1000 tiles should be flat if there were no tail."""
import sys, time
sys.path.insert(0, '.')
import swr_amd
S = swr_amd.scenes
for H in (2048, 2160, 1376, 1344, 2720, 2752):
    ntri = int(1_000_000 * H / 2160)
    sc = S.cfg4_soup(ntri=ntri, height=H)
    with swr_amd.Context() as ctx:
        ctx.scene_upload(sc.vertices, sc.indices); ctx.target_set(sc.width, H)
        ctx.pipeline_enable(False)
        for _ in range(10): ctx.draw(sc.transform, sc.flags)
        ctx.sync(); ctx.timing_enable(1); ctx.timing_reset()
        for _ in range(60): ctx.draw(sc.transform, sc.flags)
        sums, n = ctx.timing_totals(); ctx.timing_enable(0)
        t = ctx.timings()
        us = sums['raster_ms'] / n * 1e3
        print(f"H={H}: {t['tiles']} tiles = {t['tiles']/1280:.2f} x 1280, pairs {t['tile_pairs']}, k_raster {us:.1f} us, {us/t['tiles']*1000:.2f} us per 1000 tiles", flush=True)
