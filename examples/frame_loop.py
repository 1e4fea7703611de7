#!/usr/bin/env python3
"""Headless version of the reference app's frame loop (renderer/App.swift:153-188).

Every frame the app builds `projection * Transform(scale 2, rotation(time), translation (0,0,1))`
(App.swift:169-183), hands the same sphere mesh to `renderer.render(renderPass:)` (App.swift:185)
and advances `time += 1/60` (App.swift:155-157).  Here the mesh stays resident on the MI355X
(swr_scene_upload once), each frame is one swr_draw, and frames are written as binary PPM.

    python examples/frame_loop.py --frames 4 --size 512 --out /tmp/frames [--obj mesh.obj] [--depth-test]

The demo mesh is a UV sphere standing in for ModelIO's `MDLMesh(sphereWithExtent: 0.4, segments: 13x13,
inwardNormals: true)` (App.swift:124) with colour = |normal| (App.swift:133).  `--obj` loads a
Wavefront OBJ instead (positions + optional normals; faces are fan-triangulated); `--ply` a Stanford PLY
(ascii or binary_little_endian; x y z, optional nx ny nz or red green blue; faces fan-triangulated) — the format
the Stanford bunny ships in.  Neither format is part of the reference (its only mesh is the ModelIO sphere).
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


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def load_ply(path: str):
    """Minimal Stanford PLY reader (ascii 1.0 / binary_little_endian 1.0): element vertex with x y z and optional
    nx ny nz (colour = |normal|, as the app does, App.swift:133) or red green blue (uchar 0..255 or float 0..1);
    element face with one list property (vertex_indices), fan-triangulated.  Other elements / properties are skipped."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:                          # EOF before end_header (readline returns b"" for ever)
                raise ValueError("PLY header without end_header")
            t = line.decode("ascii", "replace").split()
            if not t:
                continue
            if t[0] == "format":
                fmt = t[1]
            elif t[0] == "element":
                elements.append({"name": t[1], "count": int(t[2]), "props": []})
            elif t[0] == "property":
                if t[1] == "list":
                    elements[-1]["props"].append(("list", t[2], t[3], t[4]))
                else:
                    elements[-1]["props"].append(("scalar", t[1], t[2]))
            elif t[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian"):
            raise ValueError(f"unsupported PLY format {fmt}")
        verts, faces = None, []
        if fmt == "ascii":
            tokens = f.read().split()
            pos = 0
        for el in elements:
            scalars = [p for p in el["props"] if p[0] == "scalar"]
            lists = [p for p in el["props"] if p[0] == "list"]
            if el["name"] == "vertex" and not lists:
                names = [p[2] for p in scalars]
                if fmt == "ascii":
                    n = len(names) * el["count"]
                    arr = np.array(tokens[pos:pos + n], dtype=np.float64).reshape(el["count"], len(names))
                    pos += n
                    cols = {nm: arr[:, k] for k, nm in enumerate(names)}
                    isint = {p[2]: _PLY_TYPES[p[1]][0] in "iu" for p in scalars}
                else:
                    dt = np.dtype([(p[2], "<" + _PLY_TYPES[p[1]]) for p in scalars])
                    arr = np.frombuffer(f.read(dt.itemsize * el["count"]), dtype=dt)
                    cols = {nm: arr[nm].astype(np.float64) for nm in names}
                    isint = {p[2]: _PLY_TYPES[p[1]][0] in "iu" for p in scalars}
                xyz = np.stack([cols["x"], cols["y"], cols["z"]], -1).astype(np.float32)
                if all(k in cols for k in ("red", "green", "blue")):
                    rgb = np.stack([cols["red"], cols["green"], cols["blue"]], -1)
                    if isint["red"]:
                        rgb = rgb / 255.0
                elif all(k in cols for k in ("nx", "ny", "nz")):
                    rgb = np.abs(np.stack([cols["nx"], cols["ny"], cols["nz"]], -1))
                else:
                    rgb = np.full(xyz.shape, 0.577)
                verts = (xyz, rgb.astype(np.float32))
            else:                                   # faces (or anything else: parsed to stay in sync, kept only for "face")
                for _ in range(el["count"]):
                    row_lists = []
                    for p in el["props"]:
                        if p[0] == "scalar":
                            if fmt == "ascii":
                                pos += 1
                            else:
                                f.read(np.dtype(_PLY_TYPES[p[1]]).itemsize)
                        else:
                            if fmt == "ascii":
                                n = int(tokens[pos]); pos += 1
                                row_lists.append([int(x) for x in tokens[pos:pos + n]]); pos += n
                            else:
                                n = int(np.frombuffer(f.read(np.dtype(_PLY_TYPES[p[1]]).itemsize), dtype="<" + _PLY_TYPES[p[1]])[0])
                                dt = np.dtype("<" + _PLY_TYPES[p[2]])
                                row_lists.append(np.frombuffer(f.read(dt.itemsize * n), dtype=dt).astype(np.int64).tolist())
                    if el["name"] == "face" and row_lists:
                        face = row_lists[0]
                        for k in range(1, len(face) - 1):
                            faces += [face[0], face[k], face[k + 1]]
        if verts is None:
            raise ValueError("PLY file without a vertex element")
    return S.pack_vertices(verts[0], verts[1]), np.array(faces, dtype=np.int64)


def write_ply(path: str, vertices: np.ndarray, indices: np.ndarray, binary: bool = True):
    """Writes the mesh as PLY (x y z + float red green blue; triangles) — the inverse of load_ply, for round trips."""
    v = np.asarray(vertices, dtype=np.float32).reshape(-1, 8)
    tri = np.asarray(indices, dtype=np.int64).reshape(-1, 3)
    hdr = ("ply\nformat %s 1.0\ncomment written by examples/frame_loop.py\nelement vertex %d\nproperty float x\nproperty float y\n"
           "property float z\nproperty float red\nproperty float green\nproperty float blue\nelement face %d\n"
           "property list uchar int vertex_indices\nend_header\n") % ("binary_little_endian" if binary else "ascii", v.shape[0], tri.shape[0])
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        if binary:
            f.write(np.ascontiguousarray(v[:, [0, 1, 2, 4, 5, 6]], dtype="<f4").tobytes())
            rec = np.zeros(tri.shape[0], dtype=[("n", "u1"), ("i", "<i4", 3)])
            rec["n"] = 3
            rec["i"] = tri
            f.write(rec.tobytes())
        else:
            for row in v[:, [0, 1, 2, 4, 5, 6]]:
                f.write((" ".join(repr(float(x)) for x in row) + "\n").encode("ascii"))
            for t in tri:
                f.write(("3 %d %d %d\n" % tuple(int(x) for x in t)).encode("ascii"))


def load_mesh(path: str):
    return load_ply(path) if path.lower().endswith(".ply") else load_obj(path)


def write_ppm(path: str, bgra: np.ndarray):
    h, w, _ = bgra.shape
    with open(path, "wb") as f:
        f.write(f"P6\n{w} {h}\n255\n".encode())
        f.write(np.ascontiguousarray(bgra[..., [2, 1, 0]]).tobytes())


def run(frames: int, size: int, out: str | None, obj: str | None = None, depth_test: bool = False,
        time0: float = 0.0):
    """Returns the list of (colour, depth) frames; writes PPMs when `out` is given."""
    vertices, indices = load_mesh(obj) if obj else sphere_mesh()
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


def run_streamed(frames: int, size: int, obj: str | None = None, depth_test: bool = False, time0: float = 0.0,
                 device_count: int = 1, on_frame=None):
    """The same loop the way a host that wants every frame should drive it: two page-locked
    image sets alternate; frame k is being copied to the host (swr_present, asynchronous, every band of a multi-GPU
    context into its rows of the ONE image) while frame k+1 is drawn; the host touches frame k only after
    swr_present_wait.  `on_frame(k, colour, depth)` sees the host images (valid until the next but one present).
    Returns the frames (copies)."""
    vertices, indices = load_mesh(obj) if obj else sphere_mesh()
    flags = S.FLAG_DEPTH_TEST if depth_test else 0
    sets = [(swr_amd.HostImage((size, size, 4), np.uint8), swr_amd.HostImage((size, size), np.float32)) for _ in range(2)]
    results = []

    def deliver(k):
        c, d = sets[k & 1]
        if on_frame:
            on_frame(k, c.array, d.array)
        results.append((c.array.copy(), d.array.copy()))

    with swr_amd.Context(0, device_count=device_count) as ctx:
        ctx.scene_upload(vertices, indices)
        ctx.target_set(size, size)
        time = time0
        for k in range(frames):
            ctx.draw(S.app_transform(time), flags)
            ctx.present(*sets[k & 1])                   # returns at once
            if k >= 1:
                ctx.present_wait()                      # every enqueued copy has landed: frame k-1 (and k) are host-visible
                deliver(k - 1)
            time += 1.0 / 60.0
        ctx.present_wait()
        if frames:
            deliver(frames - 1)
    for c, d in sets:
        c.free(); d.free()
    return results


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--size", type=int, default=512)    # the app renders 512x512 (App.swift:52-53)
    ap.add_argument("--out", default=None)
    ap.add_argument("--obj", default=None, help="Wavefront OBJ mesh")
    ap.add_argument("--ply", default=None, help="Stanford PLY mesh (ascii or binary little endian)")
    ap.add_argument("--depth-test", action="store_true")
    ap.add_argument("--stream", action="store_true", help="asynchronous presents into two page-locked image sets")
    ap.add_argument("--gpus", type=int, default=1, help="bands / GPUs of the one context (with --stream)")
    a = ap.parse_args()
    if a.stream:
        res = run_streamed(a.frames, a.size, a.ply or a.obj, a.depth_test, device_count=a.gpus)
        print(f"{a.frames} frames streamed, coverage per frame: {[round(float((c[..., 3] == 255).mean()), 4) for c, _ in res]}")
        sys.exit(0)
    _, idx, res = run(a.frames, a.size, a.out, a.ply or a.obj, a.depth_test)
    cov = [(c[..., 3] == 255).mean() for c, _, _ in res]
    print(f"{a.frames} frames, {idx.size // 3} triangles, coverage per frame: {[round(float(x), 4) for x in cov]}")
