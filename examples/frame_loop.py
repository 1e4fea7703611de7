#!/usr/bin/env python3
"""Headless version of the reference app's frame loop (renderer/App.swift:153-188).

Every frame the app builds `projection * Transform(scale 2, rotation(time), translation (0,0,1))`
(App.swift:169-183), hands the same sphere mesh to `renderer.render(renderPass:)` (App.swift:185)
and advances `time += 1/60` (App.swift:155-157).  Here the mesh stays resident on the MI355X
(swr_scene_upload once), each frame is one swr_draw, and frames are written as binary PPM.

    python examples/frame_loop.py --frames 4 --size 512 --out /tmp/frames [--obj mesh.obj] [--depth-test]

The demo mesh is a UV sphere standing in for ModelIO's `MDLMesh(sphereWithExtent: 0.4, segments: 13x13,
inwardNormals: true)` (App.swift:124) with colour = |normal| (App.swift:133).  `--obj` loads a
Wavefront OBJ instead (positions + optional normals; faces are fan-triangulated).
"""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import swr_amd  # noqa: E402

S = swr_amd.scenes


def sphere_mesh(extent: float = 0.4, segments: int = 13):
    """UV sphere, `segments` x `segments`, colour = |normal|."""
    nu, nv = segments, segments
    r = extent / 2.0
    verts, cols, idx = [], [], []
    for j in range(nv + 1):
        phi = math.pi * j / nv
        for i in range(nu + 1):
            th = 2.0 * math.pi * i / nu
            n = (math.sin(phi) * math.cos(th), math.cos(phi), math.sin(phi) * math.sin(th))
            verts.append((r * n[0], r * n[1], r * n[2]))
            cols.append((abs(n[0]), abs(n[1]), abs(n[2])))
    for j in range(nv):
        for i in range(nu):
            a = j * (nu + 1) + i
            b = a + 1
            c = a + nu + 1
            d = c + 1
            idx += [a, c, b, b, c, d]
    return (S.pack_vertices(np.array(verts, dtype=np.float32), np.array(cols, dtype=np.float32)),
            np.array(idx, dtype=np.int64))


def load_obj(path: str):
    """Minimal Wavefront OBJ reader: v, vn, f (v, v/vt, v//vn, v/vt/vn; negative indices; n-gons)."""
    pos, nrm, out_v, out_c, idx = [], [], [], [], []
    cache = {}
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                pos.append(tuple(float(x) for x in t[1:4]))
            elif t[0] == "vn":
                nrm.append(tuple(float(x) for x in t[1:4]))
            elif t[0] == "f":
                face = []
                for tok in t[1:]:
                    parts = tok.split("/")
                    vi = int(parts[0])
                    vi = vi - 1 if vi > 0 else len(pos) + vi
                    ni = None
                    if len(parts) == 3 and parts[2]:
                        ni = int(parts[2])
                        ni = ni - 1 if ni > 0 else len(nrm) + ni
                    key = (vi, ni)
                    if key not in cache:
                        cache[key] = len(out_v)
                        out_v.append(pos[vi])
                        n = nrm[ni] if ni is not None else (0.577, 0.577, 0.577)
                        out_c.append(tuple(abs(x) for x in n))
                    face.append(cache[key])
                for k in range(1, len(face) - 1):
                    idx += [face[0], face[k], face[k + 1]]
    return (S.pack_vertices(np.array(out_v, dtype=np.float32), np.array(out_c, dtype=np.float32)),
            np.array(idx, dtype=np.int64))


def write_ppm(path: str, bgra: np.ndarray):
    h, w, _ = bgra.shape
    with open(path, "wb") as f:
        f.write(f"P6\n{w} {h}\n255\n".encode())
        f.write(np.ascontiguousarray(bgra[..., [2, 1, 0]]).tobytes())


def run(frames: int, size: int, out: str | None, obj: str | None = None, depth_test: bool = False,
        time0: float = 0.0):
    """Returns the list of (colour, depth) frames; writes PPMs when `out` is given."""
    vertices, indices = load_obj(obj) if obj else sphere_mesh()
    flags = S.FLAG_DEPTH_TEST if depth_test else 0
    results = []
    with swr_amd.Context() as ctx:
        ctx.scene_upload(vertices, indices)            # RenderPass.vertices / .indices, App.swift:163
        ctx.target_set(size, size)                      # MetalView.Coordinator.width/height, App.swift:52-53
        time = time0
        for k in range(frames):
            m = S.app_transform(time)                   # App.swift:169-183
            ctx.draw(m, flags)                          # renderer.render(renderPass:), App.swift:185
            color, depth = ctx.read_color(), ctx.read_depth()
            results.append((color, depth, m))
            if out:
                os.makedirs(out, exist_ok=True)
                write_ppm(os.path.join(out, f"frame_{k:04d}.ppm"), color)
            time += 1.0 / 60.0                          # App.swift:155-157
    return vertices, indices, results


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--size", type=int, default=512)    # the app renders 512x512 (App.swift:52-53)
    ap.add_argument("--out", default=None)
    ap.add_argument("--obj", default=None)
    ap.add_argument("--depth-test", action="store_true")
    a = ap.parse_args()
    _, idx, res = run(a.frames, a.size, a.out, a.obj, a.depth_test)
    cov = [(c[..., 3] == 255).mean() for c, _, _ in res]
    print(f"{a.frames} frames, {idx.size // 3} triangles, coverage per frame: {[round(float(x), 4) for x in cov]}")
