"""A soak of the frame loop: pipelined frames of a scene with small and tile-spanning triangles, a new transform and rule set every
frame, presents in between; every n-th frame is compared with the oracle bit for bit.  900 frames in the suite; a long one:
SWR_SOAK_FRAMES=6000 SWR_SOAK_EVERY=61 python -m pytest tests/test_gpu_soak.py -m gpu -s"""
import os
import time

import numpy as np
import pytest


def soak(swr_amd, oracle, frames, every, log):
    S = swr_amd.scenes
    small = S.random_soup(30000, 1280, 720, 0x57E5, r_ndc=0.025, flags=1, margin=1.05)
    big = S.random_soup(24, 1280, 720, 0x57E6, r_ndc=1.2, flags=1, margin=0.8)
    v = np.concatenate([big.vertices, small.vertices]); i = np.concatenate([big.indices, small.indices + big.vertices.shape[0]])
    W, H = 1280, 720
    rng = np.random.default_rng(7)
    checked = 0
    dropped = 0
    t0 = time.time()
    with swr_amd.Context() as ctx:
        ctx.scene_upload(v, i); ctx.target_set(W, H)
        img = swr_amd.HostImage((H, W), np.float32)
        for f in range(frames):
            m = S.app_transform(float(rng.uniform(0, 6.28))) if f % 3 else S.identity()
            flags = int(rng.choice([1, 3, 0, 5, 7]))
            try:
                ctx.draw(m, flags)
                if f % 11 == 0:
                    ctx.present(None, img)
                if f % every == 0:
                    ctx.sync()
            except swr_amd.SwrError as e:
                # a presented frame of an un-waited burst outgrew the tile regions under this transform: reported (the regions have
                # been grown), as documented (include/swr.h SWR_ERR_FRAME_DROPPED); anything else is a failure
                if e.code != -8: raise
                dropped += 1
                ctx.draw(m, flags); ctx.sync()
            if f % every == 0:
                d = ctx.read_depth()
                if flags & 4:
                    rc, rd, _, _ = oracle.render_metal(v, i, m, W, H, flags & 2)
                else:
                    rc, rd, _, _ = oracle.render(v, i, m, W, H, (flags & 3) | oracle.TINV_PER_TRIANGLE)
                assert d.tobytes() == rd.tobytes(), f"frame {f} flags {flags}: depth differs"
                if not (flags & 2):
                    assert np.array_equal(ctx.read_color(), rc), f"frame {f} flags {flags}: colour differs"
                checked += 1
            if f % 500 == 0:
                log(f"frame {f}: {checked} frames checked, {time.time() - t0:.0f} s")
        try:
            ctx.present_wait()
        except swr_amd.SwrError as e:          # (the last presented frame of the burst may be the one that outgrew its regions)
            if e.code != -8: raise
            dropped += 1
        img.free()
    log(f"stress ok: {frames} frames, {checked} checked against the oracle, {dropped} reported as dropped and redrawn, {time.time() - t0:.0f} s")
    return checked, dropped


@pytest.mark.gpu
def test_soak_of_the_frame_loop(swr, oracle):
    frames = int(os.environ.get("SWR_SOAK_FRAMES", "900"))
    every = int(os.environ.get("SWR_SOAK_EVERY", "41"))
    checked, dropped = soak(swr, oracle, frames, every, print)
    assert checked >= frames // every
