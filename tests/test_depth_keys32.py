"""Depth-only z-tested frames (SWR_FLAG_DEPTH_TEST | SWR_FLAG_NO_COLOR, BASELINE config 4) take 32-bit depth keys in
k_raster_depth (LDS float minimum); a tile whose result holds a zero of either sign — under '<' the two zeros are equal and
the first drawn keeps its sign, Renderer.swift:257-261, which a 32-bit key cannot know — is rastered again by the same
workgroup with the 64-bit (depth, primitive) keys.  Every case below must be bit-exact against the oracle whichever way its
tiles go: zeros, negative depths (they win, and cfg4's extrapolated spans produce them), denormals, NaN / inf."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT, NC = 1, 2


def depth_only(ctx, oracle, scene, what):
    flags = DT | NC
    _, ref_d, st, rc = oracle.render(scene.vertices, scene.indices, scene.transform, scene.width, scene.height,
                                     flags | oracle.TINV_PER_TRIANGLE)
    assert rc == 0
    _, d = ctx.render(scene.vertices, scene.indices, scene.transform, scene.width, scene.height, flags)
    gb, rb = d.view(np.uint32), ref_d.view(np.uint32)
    bad = np.nonzero(gb != rb)
    assert bad[0].size == 0, (f"{what}: {bad[0].size} depth values differ; first (y,x)=({bad[0][0]},{bad[1][0]}) "
                              f"got {d[bad[0][0], bad[1][0]]!r} want {ref_d[bad[0][0], bad[1][0]]!r}")
    return ref_d, st


@pytest.mark.parametrize("ntri,w,h,r,seed", [(3000, 640, 360, 0.05, 11), (20000, 512, 512, 0.02, 12), (400, 130, 70, 0.3, 13)])
def test_all_depths_zero(gpu_ctx, oracle, swr, ntri, w, h, r, seed):
    """Every vertex at z = 0 (a 2-D scene): every covered tile takes the 64-bit path; the signs of the zeros (products of a
    zero z with negative weights give -0, sums of mixed zeros +0) must be the first-drawn winner's."""
    s = swr.scenes.random_soup(ntri, w, h, seed, r_ndc=r, flags=DT | NC, margin=1.1)
    s.vertices[:, 2] = 0.0
    ref, st = depth_only(gpu_ctx, oracle, s, "all z = 0")
    assert st.fragments > 0 and (ref == 0).any()


def test_mixed_signed_zeros_and_first_drawn_sign(gpu_ctx, oracle, swr):
    s = swr.scenes.random_soup(4000, 400, 300, 21, r_ndc=0.08, flags=DT | NC, margin=1.1)
    z = np.where(swr.scenes.splitmix64(77, s.vertices.shape[0]) & 1, np.float32(-0.0), np.float32(0.0))
    s.vertices[:, 2] = z
    ref, _ = depth_only(gpu_ctx, oracle, s, "mixed +-0")
    assert (ref == 0).any()
    # the sign of a surviving zero is the first drawn fragment's: drawing the same triangles in reverse order changes signs
    s2 = swr.scenes.Scene("rev", s.width, s.height, s.vertices, np.ascontiguousarray(s.indices.reshape(-1, 3)[::-1]).reshape(-1),
                          s.transform, s.flags)
    depth_only(gpu_ctx, oracle, s2, "mixed +-0, reversed draw order")


def test_negative_depths_depth_only(gpu_ctx, oracle, swr):
    s = swr.scenes.random_soup(6000, 640, 400, 41, r_ndc=0.06, flags=DT | NC)
    s.vertices[:, 2] = s.vertices[:, 2] * 6.0 - 3.0      # z in [-3, 3): negative and > 1
    ref, _ = depth_only(gpu_ctx, oracle, s, "negative depths")
    assert (ref < 0).any() and (ref > 1).any()


def test_negative_depths_in_a_few_tiles_only(gpu_ctx, oracle, swr):
    """One corner of the screen holds negative depths (they win every z-test there); zeros only where z was cleared."""
    s = swr.scenes.random_soup(30000, 1024, 768, 43, r_ndc=0.02, flags=DT | NC)
    v = s.vertices
    corner = (v[:, 0] < -0.6) & (v[:, 1] > 0.6)
    v[corner, 2] -= 1.5
    v[(v[:, 0] > 0.7) & (v[:, 1] < -0.7), 2] = 0.0
    ref, _ = depth_only(gpu_ctx, oracle, s, "negative corner")
    assert (ref < 0).any()


def test_denormal_and_tiny_depths(gpu_ctx, oracle, swr):
    s = swr.scenes.random_soup(5000, 512, 384, 51, r_ndc=0.07, flags=DT | NC)
    scale = np.float32(2.0) ** np.float32(-140)          # z * scale is denormal; the interpolated depths stay denormal / zero
    s.vertices[:, 2] = (s.vertices[:, 2] * scale).astype(np.float32)
    ref, _ = depth_only(gpu_ctx, oracle, s, "denormal depths")
    fin = ref[np.isfinite(ref)]
    assert fin.size and (np.abs(fin) < 1e-38).all()


def test_nan_and_infinite_vertex_depths(gpu_ctx, oracle, swr):
    """A NaN / +-inf z makes NaN / inf depths: NaN and +inf never pass '<' (:258), -inf always does."""
    s = swr.scenes.random_soup(3000, 480, 320, 61, r_ndc=0.08, flags=DT | NC)
    r = swr.scenes.splitmix64(5, s.vertices.shape[0]) % 23
    s.vertices[r == 0, 2] = np.nan
    s.vertices[r == 1, 2] = np.inf
    s.vertices[r == 2, 2] = -np.inf
    s.vertices[r == 3, 2] = -np.nan
    depth_only(gpu_ctx, oracle, s, "non-finite z")


@pytest.mark.parametrize("w,h", [(1, 1), (5, 3), (64, 32), (65, 33), (130, 70), (513, 257)])
def test_small_and_ragged_targets(gpu_ctx, oracle, swr, w, h):
    s = swr.scenes.random_soup(300, w, h, 70 + w, r_ndc=0.4, flags=DT | NC)
    depth_only(gpu_ctx, oracle, s, f"{w}x{h}")
    s.vertices[::2, 2] = 0.0
    depth_only(gpu_ctx, oracle, s, f"{w}x{h} with zeros")


def test_large_triangles_and_occluders(gpu_ctx, oracle, swr):
    """Cooperative walk, wide chunks and the dense phase on the 32-bit keys, with and without tiles that fall back."""
    for zq in (0.5, 0.0, -0.25):
        s = swr.scenes.occluded_soup(ntri=20000, width=1280, height=720, z_occluder=zq) if hasattr(swr.scenes, "occluded_soup") else None
        if s is None:
            pytest.skip("no occluded_soup scene")
        depth_only(gpu_ctx, oracle, s, f"occluder at z = {zq}")


def test_bands_on_32_bit_keys(swr, oracle):
    s = swr.scenes.random_soup(20000, 900, 700, 81, r_ndc=0.03, flags=DT | NC)
    s.vertices[::7, 2] = -0.5
    _, ref_d, _, rc = oracle.render(s.vertices, s.indices, s.transform, s.width, s.height, DT | NC | oracle.TINV_PER_TRIANGLE)
    assert rc == 0
    for bands in (2, 5):
        with swr.Context(0, device_count=bands) as ctx:
            _, d = ctx.render(s.vertices, s.indices, s.transform, s.width, s.height, DT | NC)
            assert np.array_equal(d.view(np.uint32), ref_d.view(np.uint32)), f"{bands} bands"


def test_depth_only_equals_depth_of_colour_frame(gpu_ctx, swr):
    """The 32-bit path (depth-only) and the 64-bit path (colour + depth) must write the same depth image."""
    s = swr.scenes.random_soup(50000, 1920, 1080, 91, r_ndc=0.015, flags=DT)
    s.vertices[::5, 2] *= -1.0
    _, d64 = gpu_ctx.render(s.vertices, s.indices, s.transform, s.width, s.height, DT)
    _, d32 = gpu_ctx.render(s.vertices, s.indices, s.transform, s.width, s.height, DT | NC)
    assert np.array_equal(d64.view(np.uint32), d32.view(np.uint32))
