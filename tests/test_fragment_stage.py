"""The extended fragment stage (per-pixel Phong, textures — BASELINE configs 3 and 5).

NOT IN THE REFERENCE (its fragment stage is float4(color, 1), Shaders.metal:116-121), so there is nothing of
the reference's to pin against: parity here is build-internal.  The definition lives in include/swr.h
(swr_material) and oracle/swr_oracle.h (swro_fragment); these tests pin it with hand-derived known answers,
cross-check the C oracle against the independent NumPy restatement, and (GPU part) compare the HIP kernels with
the oracle bit for bit through the C-ABI.
"""
import numpy as np
import pytest

from oracle import swr_oracle_np as onp

DT, NC, METAL = 1, 2, 4
F = np.float32


def shading(swr, shader=1, k=0, texture=None, attrs=None):
    S = swr.scenes
    if attrs is None:
        attrs = S.pack_attrs(np.tile(np.array([0, 0, -2], dtype=F), (3, 1)), np.zeros((3, 2), dtype=F))
    return S.Shading(attrs, shader, k, (0.0, 0.0, -1.0), (0.0, 0.0, -1.0), 0.1, 0.5, 0.25, texture)


# ---- known answers (derived by hand from the definition) -----------------------------------------
def test_fragment_kat_phong(oracle, swr):
    sh = shading(swr, 1, 3)
    # N = (0,0,-1): N.L = N.H = 1, s = 1 -> rgb = c * fl(0.1 + 0.5) + 0.25
    lit = F(0.1) + F(0.5) * F(1.0)
    c = np.array([1.0, 0.5, 0.25], dtype=F)
    want = np.append(c * lit + F(0.25), F(1.0))
    got = oracle.fragment(sh, c, [0, 0, -2], [0, 0])
    assert got.tobytes() == want.astype(F).tobytes()
    # normal facing away: diffuse and specular clamp to 0 -> ambient only
    got = oracle.fragment(sh, c, [0, 0, 5], [0, 0])
    assert got.tobytes() == np.append(c * F(0.1) + F(0.0), F(1.0)).astype(F).tobytes()
    # zero normal: N = 0 -> ambient only (no division by zero)
    got = oracle.fragment(sh, c, [0, 0, 0], [0, 0])
    assert got.tobytes() == np.append(c * F(0.1) + F(0.0), F(1.0)).astype(F).tobytes()
    # shininess: ndh = 0.5 (normal at 60 degrees), k = 2 -> s = 0.5^4
    sh2 = shading(swr, 1, 2)
    n = np.array([np.sqrt(3.0), 0.0, -1.0])
    got = oracle.fragment(sh2, [0, 0, 0], n, [0, 0])
    N = (n.astype(F) / np.sqrt(F(n.astype(F)[0] * n.astype(F)[0]) + F(0) + F(1.0)))
    ndh = max(F(N[0] * F(0) + N[1] * F(0)) + N[2] * F(-1), F(0))
    s = ndh * ndh
    s = s * s
    assert abs(float(got[0]) - 0.25 * 0.5 ** 4) < 1e-6 and got[0] == F(0.25) * s


def test_fragment_kat_texture(oracle, swr):
    # 2x2 texture, b,g,r,a bytes; r channel: 0, 255 / 51, 102
    tex = np.zeros((2, 2, 4), dtype=np.uint8)
    tex[0, 0] = (10, 20, 0, 255); tex[0, 1] = (10, 20, 255, 255)
    tex[1, 0] = (10, 20, 51, 255); tex[1, 1] = (10, 20, 102, 255)
    sh = shading(swr, 2, 0, tex)
    sh.ambient, sh.diffuse, sh.specular = 1.0, 0.0, 0.0           # out = color * texel
    one = np.ones(3, dtype=F)
    # texel centres: uv = ((i + .5)/2, (j + .5)/2) -> exactly that texel
    assert oracle.fragment(sh, one, [0, 0, -1], [0.25, 0.25])[0] == F(0.0)
    assert oracle.fragment(sh, one, [0, 0, -1], [0.75, 0.25])[0] == F(1.0)
    assert oracle.fragment(sh, one, [0, 0, -1], [0.25, 0.75])[0] == F(51) / F(255)
    assert oracle.fragment(sh, one, [0, 0, -1], [0.75, 0.75])[0] == F(102) / F(255)
    # half way between the two texels of row 0: 0 + (1 - 0) * 0.5
    assert oracle.fragment(sh, one, [0, 0, -1], [0.5, 0.25])[0] == F(0.5)
    # repeat addressing: uv + integer = same texel; left of texel 0 wraps to texel 1
    assert oracle.fragment(sh, one, [0, 0, -1], [3.75, -1.75])[0] == F(1.0)
    assert oracle.fragment(sh, one, [0, 0, -1], [0.0, 0.25])[0] == F(0.5)      # x = -0.5: (t[1] + t[0]) / 2
    # g and b channels come from bytes 1 and 0
    got = oracle.fragment(sh, one, [0, 0, -1], [0.25, 0.25])
    assert got[1] == F(20) / F(255) and got[2] == F(10) / F(255) and got[3] == F(1.0)


def test_passthrough_material_is_the_reference_stage(oracle, swr):
    s = swr.scenes.random_soup(80, 96, 64, 0x33, r_ndc=0.3, flags=DT)
    sh = swr.scenes.random_shading(s.vertices.shape[0], 5, shader=0)
    a = oracle.render(s.vertices, s.indices, s.transform, 96, 64, DT)
    b = oracle.render(s.vertices, s.indices, s.transform, 96, 64, DT, shading=sh)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("shader", [1, 2])
@pytest.mark.parametrize("mode", ["painter", "z", "metal"])
def test_c_oracle_equals_numpy_restatement_shaded(oracle, swr, shader, mode):
    s = swr.scenes.random_soup(70, 96, 64, 0x51 + shader, r_ndc=0.3, flags=DT)
    sh = swr.scenes.random_shading(s.vertices.shape[0], 77, shader)
    if mode == "metal":
        c, d, _, rc = oracle.render_metal(s.vertices, s.indices, s.transform, 96, 64, 0, shading=sh)
        c2, d2 = onp.render_metal(s.vertices, s.indices, s.transform, 96, 64, shading=sh)
    else:
        z = mode == "z"
        c, d, _, rc = oracle.render(s.vertices, s.indices, s.transform, 96, 64, DT if z else 0, shading=sh)
        c2, d2, _ = onp.render(s.vertices, s.indices, s.transform, 96, 64, depth_test=z, shading=sh)
    assert rc == 0
    assert np.array_equal(c, c2) and d.tobytes() == d2.tobytes()
    assert (c[..., 3] == 255).sum() > 500 and len(np.unique(c[..., :3])) > 50     # something was shaded


def test_shaded_depth_is_unchanged_and_errors(oracle, swr):
    s = swr.scenes.random_soup(50, 64, 64, 9, r_ndc=0.4, flags=DT)
    sh = swr.scenes.random_shading(s.vertices.shape[0], 3, 2)
    plain = oracle.render(s.vertices, s.indices, s.transform, 64, 64, DT)
    lit = oracle.render(s.vertices, s.indices, s.transform, 64, 64, DT, shading=sh)
    assert plain[1].tobytes() == lit[1].tobytes() and not np.array_equal(plain[0], lit[0])
    sh.texture = None                                   # textured material without a texture
    assert oracle.render(s.vertices, s.indices, s.transform, 64, 64, DT, shading=sh)[3] == -1
    sh.shader, sh.shininess_log2 = 1, 17
    assert oracle.render(s.vertices, s.indices, s.transform, 64, 64, DT, shading=sh)[3] == -1


def test_named_configs_carry_their_shading(swr):
    a = swr.scenes.cfg3_phong(width=384, height=216)
    assert a.shading.shader == 1 and a.shading.attrs.shape == a.vertices.shape
    n = a.shading.attrs[:, 0:3]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-5)
    b = swr.scenes.cfg5_textured(tex=64, width=768, height=432)
    assert b.shading.shader == 2 and b.shading.texture.shape == (64, 64, 4)
    for v in (a.shading.light_dir, a.shading.half_dir):
        assert abs(np.linalg.norm(v) - 1.0) < 1e-6


# ---- committed fixtures (tests/golden/shaded/*.npz, written by tests/golden/make_golden.py) -------------
import glob
import os

SHADED = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "shaded", "*.npz")))


def load_shaded(swr, path):
    g = np.load(path)
    a, d_, s_ = (float(x) for x in g["ads"])
    sh = swr.scenes.Shading(g["attrs"], int(g["shader"]), int(g["shininess_log2"]), tuple(g["light_dir"]),
                            tuple(g["half_dir"]), a, d_, s_, g["texture"] if "texture" in g.files else None)
    return g, sh


def test_shaded_golden_dir_has_vectors():
    assert len(SHADED) >= 4


@pytest.mark.parametrize("path", SHADED, ids=os.path.basename)
def test_oracle_matches_shaded_golden(oracle, swr, path):
    g, sh = load_shaded(swr, path)
    W, H = int(g["width"]), int(g["height"])
    if int(g["metal"]):
        c, d, _, rc = oracle.render_metal(g["vertices"], g["indices"], g["transform"], W, H, 0, shading=sh)
    else:
        c, d, _, rc = oracle.render(g["vertices"], g["indices"], g["transform"], W, H, int(g["flags"]), shading=sh)
    assert rc == 0 and np.array_equal(c, g["color"]) and d.tobytes() == g["depth"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("path", SHADED, ids=os.path.basename)
def test_gpu_matches_shaded_golden(gpu_ctx, swr, path):
    """The HIP path against the committed fixture directly (no oracle in the loop)."""
    g, sh = load_shaded(swr, path)
    flags = METAL if int(g["metal"]) else int(g["flags"])
    c, d = gpu_ctx.render(g["vertices"], g["indices"], g["transform"], int(g["width"]), int(g["height"]), flags,
                          shading=sh)
    assert np.array_equal(c, g["color"]) and d.tobytes() == g["depth"].tobytes()


# ---- GPU parity ---------------------------------------------------------------------------------
def assert_same(got_c, got_d, ref_c, ref_d, what):
    bad = np.nonzero((got_c != ref_c).any(axis=-1))
    assert bad[0].size == 0, (f"{what}: {bad[0].size} colour pixels differ; first (y,x)=({bad[0][0]},{bad[1][0]}) "
                              f"got {got_c[bad[0][0], bad[1][0]]} want {ref_c[bad[0][0], bad[1][0]]}")
    assert got_d.tobytes() == ref_d.tobytes(), f"{what}: depth differs"


def oracle_frame(oracle, s, flags, sh):
    if flags & METAL:
        return oracle.render_metal(s.vertices, s.indices, s.transform, s.width, s.height, flags & NC, shading=sh)
    return oracle.render(s.vertices, s.indices, s.transform, s.width, s.height,
                         flags | oracle.TINV_PER_TRIANGLE, shading=sh)


@pytest.mark.gpu
@pytest.mark.parametrize("shader", [1, 2])
@pytest.mark.parametrize("flags", [0, DT, METAL])
@pytest.mark.parametrize("ntri,w,h,r,seed", [(7, 64, 32, 0.6, 2), (300, 256, 256, 0.15, 3), (4000, 640, 360, 0.05, 4),
                                             (400, 255, 129, 0.3, 6), (20000, 512, 512, 0.01, 8)])
def test_gpu_shaded_soup(gpu_ctx, oracle, swr, ntri, w, h, r, seed, flags, shader):
    s = swr.scenes.random_soup(ntri, w, h, seed, r_ndc=r, flags=flags, margin=1.1)
    sh = swr.scenes.random_shading(s.vertices.shape[0], seed * 7 + shader, shader, shininess_log2=seed % 7)
    ref_c, ref_d, _, rc = oracle_frame(oracle, s, flags, sh)
    assert rc == 0
    c, d = gpu_ctx.render(s.vertices, s.indices, s.transform, w, h, flags, shading=sh)
    assert_same(c, d, ref_c, ref_d, f"shader {shader} flags {flags} {s.name}")
    # and the next plain render on the same context is the reference's stage again
    c, d = gpu_ctx.render(s.vertices, s.indices, s.transform, w, h, flags)
    ref_c, ref_d, _, _ = oracle_frame(oracle, s, flags, None)
    assert_same(c, d, ref_c, ref_d, "passthrough after a shaded pass")


@pytest.mark.gpu
def test_gpu_shared_vertices_and_large_texture(gpu_ctx, oracle, swr):
    s = swr.scenes.random_soup(3000, 800, 600, 11, r_ndc=0.4, flags=DT, margin=1.0, shared=True)
    sh = swr.scenes.random_shading(s.vertices.shape[0], 21, 2, texture=swr.scenes.checker_texture(301, 173, 5))
    ref_c, ref_d, _, _ = oracle_frame(oracle, s, DT, sh)
    c, d = gpu_ctx.render(s.vertices, s.indices, s.transform, 800, 600, DT, shading=sh)
    assert_same(c, d, ref_c, ref_d, "indexed + 301x173 texture")


@pytest.mark.gpu
def test_gpu_cfg3_phong_full_size(gpu_ctx, oracle, swr):
    """BASELINE config 3 as named: ~69k triangles, per-pixel Phong + z-buffer, 3840x2160."""
    s = swr.scenes.cfg3_phong()
    ref_c, ref_d, st, _ = oracle_frame(oracle, s, s.flags, s.shading)
    c, d = gpu_ctx.render(s.vertices, s.indices, s.transform, s.width, s.height, s.flags, shading=s.shading)
    assert_same(c, d, ref_c, ref_d, "cfg3 phong 4K")
    assert st.fragments_written > 100000


@pytest.mark.gpu
def test_gpu_cfg5_textured_full_size(gpu_ctx, oracle, swr):
    """BASELINE config 5 as named: 262 144 triangles, textured + Phong, 7680x4320 (one GPU here)."""
    s = swr.scenes.cfg5_textured()
    ref_c, ref_d, _, _ = oracle_frame(oracle, s, s.flags, s.shading)
    c, d = gpu_ctx.render(s.vertices, s.indices, s.transform, s.width, s.height, s.flags, shading=s.shading)
    assert_same(c, d, ref_c, ref_d, "cfg5 textured 8K")


@pytest.mark.gpu
def test_gpu_resident_material_changes_and_bands(swr, oracle):
    s = swr.scenes.cfg3_phong(width=640, height=360)
    sh = s.shading
    tw, th = swr.tile_shape()
    with swr.Context() as top, swr.Context() as bottom:
        for k, ctx in enumerate((top, bottom)):
            ctx.scene_upload(s.vertices, s.indices)
            r0, r1 = swr.band_rows(360, 2, k)
            ctx.target_set(640, 360, r0, r1)
            ctx.shading_set(sh)
        color = np.zeros((360, 640, 4), dtype=np.uint8)
        depth = np.zeros((360, 640), dtype=np.float32)
        for frame, (k_shin, amb) in enumerate(((5, 0.15), (0, 0.5), (9, 0.0))):
            sh.shininess_log2, sh.ambient = k_shin, amb
            m = swr.scenes.app_transform(0.3 + frame)
            for ctx in (top, bottom):
                ctx.material_set(swr.binding.Material.from_shading(sh))     # no re-upload of the scene
                ctx.draw(m, DT)
            for ctx in (top, bottom):
                ctx.read_color(color); ctx.read_depth(depth)
            ref_c, ref_d, _, _ = oracle.render(s.vertices, s.indices, m, 640, 360, DT, shading=sh)
            assert_same(color, depth, ref_c, ref_d, f"resident frame {frame}")
        # back to the reference's stage without touching the scene
        top.material_set(None); bottom.material_set(None)
        for ctx in (top, bottom):
            ctx.draw(m, DT)
        for ctx in (top, bottom):
            ctx.read_color(color); ctx.read_depth(depth)
        ref_c, ref_d, _, _ = oracle.render(s.vertices, s.indices, m, 640, 360, DT)
        assert_same(color, depth, ref_c, ref_d, "material_set(None)")


@pytest.mark.gpu
def test_gpu_fragment_stage_errors(swr):
    S = swr.scenes
    s = S.random_soup(10, 64, 64, 1, r_ndc=0.3)
    sh = S.random_shading(s.vertices.shape[0], 1, 2)
    mat = swr.binding.Material.from_shading(sh)
    with swr.Context() as ctx:
        with pytest.raises(swr.SwrError) as e:               # attributes before a scene
            ctx.scene_attributes(sh.attrs)
        assert e.value.code == -6
        ctx.scene_upload(s.vertices, s.indices)
        ctx.target_set(64, 64)
        ctx.material_set(mat)
        with pytest.raises(swr.SwrError) as e:               # Phong material, no attributes
            ctx.draw(s.transform, DT)
        assert e.value.code == -1
        with pytest.raises(swr.SwrError) as e:               # wrong attribute count
            ctx.scene_attributes(sh.attrs[:-1])
        assert e.value.code == -1
        ctx.scene_attributes(sh.attrs)
        with pytest.raises(swr.SwrError) as e:               # textured material, no texture
            ctx.draw(s.transform, DT)
        assert e.value.code == -1
        ctx.draw(s.transform, DT | NC); ctx.sync()           # depth-only never runs the fragment stage
        ctx.texture_upload(sh.texture)
        ctx.draw(s.transform, DT); ctx.sync()
        bad = swr.binding.Material.from_shading(sh); bad.shader = 7
        with pytest.raises(swr.SwrError) as e:
            ctx.material_set(bad)
        assert e.value.code == -5
        bad.shader, bad.shininess_log2 = 1, 40
        with pytest.raises(swr.SwrError) as e:
            ctx.material_set(bad)
        assert e.value.code == -1
        ctx.scene_upload(s.vertices, s.indices)              # a new scene discards the attributes
        with pytest.raises(swr.SwrError):
            ctx.draw(s.transform, DT)
        ctx.material_set(None)
        ctx.draw(s.transform, DT); ctx.sync()                # the context stays usable


# ---- the colour resolve's winner table (swr_kernels.hip, raster_tile): records by bin position (bins of up to ~300 entries), by
# a bitmap over the bin (fuller bins), and the per-thread path for a tile with more winners than records fit -----------------------
WINNER_TABLE_SCENES = [
    # (triangles, width, height, r_ndc, what the 60 tiles of 64x32 hold; winners counted with the oracle, colours = ids)
    (12000, 640, 192, 0.02, "~200 entries, ~180 winners per tile: one record per bin entry"),
    (20000, 640, 192, 0.15, "~740 entries, 130-180 winners: numbered through the bitmap (the extended stage's 168 records: some tiles fit, some do not)"),
    (60000, 640, 192, 0.004, "~1000 tiny triangles, ~240 winners per tile: bitmap, fits (not the extended stage)"),
    (36000, 640, 192, 0.05, "~1000 entries, ~380 winners: more than records fit, the per-thread path"),
    (150000, 640, 192, 0.01, "~3000 entries, ~1100 winners"),
    (300000, 640, 192, 0.004, "> 4096 entries per tile: the keys carry no bin position"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [DT, 0, METAL])
@pytest.mark.parametrize("ntri,w,h,r,what", WINNER_TABLE_SCENES)
def test_gpu_winner_table_modes(gpu_ctx, oracle, swr, ntri, w, h, r, what, flags):
    s = swr.scenes.random_soup(ntri, w, h, 4242 + ntri % 97, r_ndc=r, flags=flags, margin=1.05)
    ref_c, ref_d, st, rc = oracle_frame(oracle, s, flags, None)
    assert rc == 0
    c, d = gpu_ctx.render(s.vertices, s.indices, s.transform, w, h, flags)
    assert_same(c, d, ref_c, ref_d, f"{what}, flags {flags}")


@pytest.mark.gpu
@pytest.mark.parametrize("flags,shader", [(DT, 2), (METAL, 1), (0, 2)])
@pytest.mark.parametrize("ntri,w,h,r,what", WINNER_TABLE_SCENES)
def test_gpu_winner_table_modes_extended_stage(gpu_ctx, oracle, swr, ntri, w, h, r, what, flags, shader):
    s = swr.scenes.random_soup(ntri, w, h, 777 + ntri % 89, r_ndc=r, flags=flags, margin=1.05)
    sh = swr.scenes.random_shading(s.vertices.shape[0], 31 + shader, shader, shininess_log2=3)
    ref_c, ref_d, _, rc = oracle_frame(oracle, s, flags, sh)
    assert rc == 0
    c, d = gpu_ctx.render(s.vertices, s.indices, s.transform, w, h, flags, shading=sh)
    assert_same(c, d, ref_c, ref_d, f"{what}, flags {flags}, shader {shader}")


@pytest.mark.gpu
def test_gpu_winner_table_in_bands_and_small_windows(swr, oracle):
    """Bands (the tile grid starts at the band's first row) and a window small enough for four workgroups per tile."""
    s = swr.scenes.random_soup(30000, 640, 200, 99, r_ndc=0.05, flags=DT, margin=1.05)
    ref_c, ref_d, _, rc = oracle_frame(oracle, s, DT, None)
    assert rc == 0
    for bands in (1, 3):
        with swr.Context(0, device_count=bands) as ctx:
            c, d = ctx.render(s.vertices, s.indices, s.transform, s.width, s.height, DT)
            assert_same(c, d, ref_c, ref_d, f"{bands} band(s)")


@pytest.mark.gpu
@pytest.mark.parametrize("flags,shader", [(DT, 0), (0, 0), (METAL, 0), (DT, 2), (METAL, 1)])
def test_gpu_colour_kernels_of_scenes_beyond_2_to_20_primitives(gpu_ctx, oracle, swr, flags, shader):
    """More than 2^20 primitives: no room for the bin position beside the original index in the key — the colour kernels without
    the winner table (k_raster<.., PLAIN>, k_raster_ext<.., PLAIN>)."""
    s = swr.scenes.random_soup((1 << 20) + 4099, 1280, 720, 515 + shader, r_ndc=0.006, flags=flags, margin=1.02)
    sh = swr.scenes.random_shading(s.vertices.shape[0], 5 + shader, shader, shininess_log2=2) if shader else None
    ref_c, ref_d, _, rc = oracle_frame(oracle, s, flags, sh)
    assert rc == 0
    c, d = gpu_ctx.render(s.vertices, s.indices, s.transform, s.width, s.height, flags, shading=sh)
    assert_same(c, d, ref_c, ref_d, f"2^20 + 4099 triangles, flags {flags}, shader {shader}")
