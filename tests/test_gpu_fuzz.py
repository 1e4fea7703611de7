"""Seeded random scenes against the oracle: image sizes (tile counts on both sides of the kernels' grid thresholds), triangle
counts and sizes (sparse / dense tiles, row-split and wide chunks, triangles for the deferred list), every rule set, colour
and depth-only, one context or bands, one-shot renders and frame loops.  40 cases in the suite;  SWR_FUZZ_CASES=400 for more."""
import os

import numpy as np
import pytest

DT, NC, MR = 1, 2, 4


def one_case(swr, oracle, rng, case):
    S = swr.scenes
    w = int(rng.choice([64, 200, 255, 512, 640, 960, 1280, 1920]))
    h = int(rng.choice([32, 129, 256, 360, 540, 720, 1080]))
    kind = int(rng.integers(0, 5))
    if kind == 0:   ntri, r = int(rng.integers(1, 60)), float(rng.uniform(0.3, 1.6))         # few, large
    elif kind == 1: ntri, r = int(rng.integers(100, 700)), float(rng.uniform(0.2, 0.9))      # many large: deep tiles
    elif kind == 2: ntri, r = int(rng.integers(2000, 40000)), float(rng.uniform(0.005, 0.05))  # small
    elif kind == 3: ntri, r = int(rng.integers(300, 6000)), float(rng.uniform(0.05, 0.25))   # mid-size: 100-200 per tile
    else:           ntri, r = int(rng.integers(1, 3000)), float(rng.uniform(0.01, 1.0))
    # keep the oracle's work bounded (fragments ~ ntri * (r * w / 2)^2 / 2)
    while ntri * (r * w * 0.5) * (r * h * 0.5) * 0.5 > 1.5e8 and ntri > 1:
        ntri //= 2
    s = S.random_soup(ntri, w, h, 0xF000 + case, r_ndc=r, flags=DT, margin=float(rng.uniform(0.5, 1.2)), shared=bool(rng.integers(0, 2)))
    if kind == 4 and ntri > 40:                                    # a few tile-spanning triangles among the rest
        b = S.random_soup(int(rng.integers(1, 12)), w, h, 0xF800 + case, r_ndc=1.3, flags=DT, margin=0.7)
        s = S.Scene("mix", w, h, np.concatenate([b.vertices, s.vertices]),
                    np.concatenate([b.indices, s.indices + b.vertices.shape[0]]), s.transform, DT)
    flags = int(rng.choice([DT, DT | NC, 0, NC, MR, MR | NC]))
    m = S.app_transform(float(rng.uniform(0, 6.28))) if rng.integers(0, 3) == 0 else s.transform
    # the extended fragment stage (per-pixel Phong, textured) on a third of the colour frames
    sh = None
    if not (flags & NC) and rng.integers(0, 3) == 0:
        sh = S.random_shading(s.vertices.shape[0], 0xFA00 + case, int(rng.choice([S.SHADER_PHONG, S.SHADER_TEXTURED_PHONG])),
                              shininess_log2=int(rng.integers(0, 7)))
    if flags & MR:
        rc, rd, _, err = oracle.render_metal(s.vertices, s.indices, m, w, h, flags & NC, shading=sh)
    else:
        rc, rd, _, err = oracle.render(s.vertices, s.indices, m, w, h, flags | oracle.TINV_PER_TRIANGLE, shading=sh)
    assert err == 0
    bands = int(rng.choice([0, 0, 2, 3]))
    what = f"case {case}: {ntri} tris r={r:.3f} {w}x{h} flags={flags} bands={bands} kind={kind} shader={None if sh is None else sh.shader}"
    with swr.Context(0, device_count=bands) as ctx:
        if rng.integers(0, 2):
            c, d = ctx.render(s.vertices, s.indices, m, w, h, flags, shading=sh, scene_id=int(rng.choice([0, 17])))
            assert d.tobytes() == rd.tobytes(), what + " (render): depth"
            if not (flags & NC):
                assert np.array_equal(c, rc), what + " (render): colour"
        else:
            ctx.scene_upload(s.vertices, s.indices)
            ctx.target_set(w, h)
            if sh is not None:
                ctx.shading_set(sh)
            for frame in range(3):                                  # (frames 2 and 3: sort heuristics and the deferred list settle)
                ctx.draw(m, flags)
                ctx.sync()
                assert ctx.read_depth().tobytes() == rd.tobytes(), what + f" (frame {frame}): depth"
                if not (flags & NC):
                    assert np.array_equal(ctx.read_color(), rc), what + f" (frame {frame}): colour"


@pytest.mark.gpu
def test_seeded_random_scenes(swr, oracle):
    cases = int(os.environ.get("SWR_FUZZ_CASES", "40"))
    rng = np.random.default_rng(int(os.environ.get("SWR_FUZZ_SEED", "20261004")))
    for case in range(cases):
        one_case(swr, oracle, rng, case)
