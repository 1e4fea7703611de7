"""Generates the golden fixtures in this directory from the CPU oracle (oracle/swr_oracle.c),
cross-checked against the independent NumPy restatement (oracle/swr_oracle_np.py).

The reference ships no golden images and cannot be run here (Swift + Apple frameworks), so these
vectors pin OUR restatement of renderer/Renderer.swift (parity unpinned w.r.t. the Swift binary);
the hand-derived known answers of SURVEY.md Appendix C are asserted on the way.

    python tests/golden/make_golden.py      # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import swr_amd  # noqa: E402
from oracle import oracle, swr_oracle_np  # noqa: E402

S = swr_amd.scenes


def emit(name, scene, flags):
    c, d, st, rc = oracle.render(scene.vertices, scene.indices, scene.transform, scene.width, scene.height, flags)
    assert rc == 0
    c2, d2, _ = swr_oracle_np.render(scene.vertices, scene.indices, scene.transform, scene.width, scene.height,
                                     depth_test=bool(flags & 1), no_color=bool(flags & 2))
    assert np.array_equal(d.view(np.uint32), d2.view(np.uint32)), name
    out = dict(vertices=scene.vertices, indices=scene.indices, transform=scene.transform,
               width=np.int64(scene.width), height=np.int64(scene.height), flags=np.int64(flags), depth=d,
               fragments=np.int64(st.fragments), skipped=np.int64(st.triangles_skipped))
    if not flags & 2:
        assert np.array_equal(c, c2), name
        out["color"] = c
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: {scene.triangles} tris {scene.width}x{scene.height} flags={flags} fragments={st.fragments}")
    return c, d, st


def emit_shaded(name, scene, flags, sh, metal=False):
    """Extended fragment stage (not in the reference; build-internal parity): oracle frame + its inputs."""
    W, H = scene.width, scene.height
    if metal:
        c, d, st, rc = oracle.render_metal(scene.vertices, scene.indices, scene.transform, W, H, 0, shading=sh)
        c2, d2 = swr_oracle_np.render_metal(scene.vertices, scene.indices, scene.transform, W, H, shading=sh)
    else:
        c, d, st, rc = oracle.render(scene.vertices, scene.indices, scene.transform, W, H, flags, shading=sh)
        c2, d2, _ = swr_oracle_np.render(scene.vertices, scene.indices, scene.transform, W, H,
                                         depth_test=bool(flags & 1), shading=sh)
    assert rc == 0 and np.array_equal(c, c2) and d.tobytes() == d2.tobytes(), name
    out = dict(vertices=scene.vertices, indices=scene.indices, transform=scene.transform, width=np.int64(W),
               height=np.int64(H), flags=np.int64(flags), metal=np.int64(metal), color=c, depth=d,
               attrs=sh.attrs, shader=np.int64(sh.shader), shininess_log2=np.int64(sh.shininess_log2),
               light_dir=np.asarray(sh.light_dir, np.float32), half_dir=np.asarray(sh.half_dir, np.float32),
               ads=np.asarray([sh.ambient, sh.diffuse, sh.specular], np.float32))
    if sh.texture is not None:
        out["texture"] = sh.texture
    os.makedirs(os.path.join(HERE, "shaded"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "shaded", name + ".npz"), **out)
    print(f"shaded/{name}: {scene.triangles} tris {W}x{H} flags={flags} metal={metal} shader={sh.shader}")


def main():
    torus = S.cfg3_phong(width=320, height=180, nu=16, nv=24, ntri=None)
    emit_shaded("torus_phong_z", torus, 1, torus.shading)
    emit_shaded("torus_phong_metal", torus, 0, torus.shading, metal=True)
    grid = S.cfg5_textured(tex=32, width=256, height=144, nx=32, ny=16)
    emit_shaded("grid_textured_z", grid, 1, grid.shading)
    soup = S.random_soup(200, 192, 128, 0x5EED0200, r_ndc=0.2, margin=1.1)
    emit_shaded("soup_textured_painter", soup, 0, S.random_shading(soup.vertices.shape[0], 0x5EED0201, 2))
    c, d, st = emit("cfg1_flat", S.cfg1_triangle(), 0)
    cov = c[..., 3] == 255
    assert cov.sum() == 8193 and (c[cov] == (63, 127, 255, 255)).all() and np.isposinf(d).all()
    c, d, st = emit("cfg1_gouraud", S.cfg1_triangle(True), 0)
    assert tuple(c[100, 128]) == (35, 35, 183, 255) and tuple(c[150, 100]) == (141, 29, 83, 255)
    assert tuple(c[190, 160]) == (61, 189, 3, 255)
    emit("cfg1_gouraud_z", S.cfg1_triangle(True), 1)
    emit("soup300_painter", S.random_soup(300, 256, 256, 0x5EED0100, r_ndc=0.15, margin=1.2), 0)
    emit("soup300_ztest", S.random_soup(300, 256, 256, 0x5EED0101, r_ndc=0.15, margin=1.2), 1)
    emit("soup500_depth_only", S.random_soup(500, 200, 120, 0x5EED0102, r_ndc=0.1, margin=1.1), 3)
    emit("soup64_big", S.random_soup(24, 320, 200, 0x5EED0103, r_ndc=1.1, margin=0.7), 1)
    emit("torus_app_transform", S.cfg2_teapot_scale(320, 180, time=1.0, nu=16, nv=24), 1)
    emit("degenerate_painter", S.degenerate_mix(), 0)
    emit("degenerate_ztest", S.degenerate_mix(), 1)
    emit("cfg4_mini", S.cfg4_soup(ntri=4000, width=240, height=136, r_ndc=0.06, depth_only=False, seed=0x5EED0004), 1)


if __name__ == "__main__":
    main()
