"""ctypes wrapper of oracle/libswr_oracle.so — TEST INFRASTRUCTURE (parity unpinned, see
swr_oracle.h).  Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package never imports this module."""
from __future__ import annotations

import ctypes
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libswr_oracle.so")

DEPTH_TEST = 1 << 0
NO_COLOR = 1 << 1
REAL_LINES = 1 << 3          # .line primitives: the DDA of Renderer.swift:405-419 instead of the empty stub
INV_RCP = 1 << 8
UNCLAMPED = 1 << 9
TINV_PER_TRIANGLE = 1 << 10
FMA_TRANSFORM = 1 << 11      # sensitivity switch, never the parity target


class Stats(ctypes.Structure):
    _fields_ = [("fragments", ctypes.c_int64), ("fragments_written", ctypes.c_int64),
                ("triangles_drawn", ctypes.c_int64), ("triangles_skipped", ctypes.c_int64)]


class Material(ctypes.Structure):
    _fields_ = [("shader", ctypes.c_int32), ("shininess_log2", ctypes.c_int32),
                ("light_dir", ctypes.c_float * 4), ("half_dir", ctypes.c_float * 4),
                ("ambient", ctypes.c_float), ("diffuse", ctypes.c_float), ("specular", ctypes.c_float),
                ("reserved", ctypes.c_float)]


class ShadingC(ctypes.Structure):
    _fields_ = [("attrs", ctypes.c_void_p), ("material", Material), ("texture", ctypes.c_void_p),
                ("tex_w", ctypes.c_int32), ("tex_h", ctypes.c_int32)]


def _shading_c(shading):
    """swro_shading from any object with attrs / shader / shininess_log2 / light_dir / half_dir / ambient /
    diffuse / specular / texture fields (scenes.Shading).  Returns (struct, keep-alive tuple)."""
    a = np.ascontiguousarray(shading.attrs, dtype=np.float32)
    t = None if shading.texture is None else np.ascontiguousarray(shading.texture, dtype=np.uint8)
    m = Material(int(shading.shader), int(shading.shininess_log2),
                 (ctypes.c_float * 4)(*[float(x) for x in shading.light_dir], 0.0),
                 (ctypes.c_float * 4)(*[float(x) for x in shading.half_dir], 0.0),
                 float(shading.ambient), float(shading.diffuse), float(shading.specular), 0.0)
    sc = ShadingC(a.ctypes.data, m, None if t is None else t.ctypes.data,
                  0 if t is None else t.shape[1], 0 if t is None else t.shape[0])
    return sc, (a, t)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "swr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libswr_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.swro_render.restype = ctypes.c_int
        L.swro_render.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                  ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                  ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int64, ctypes.c_int64,
                                  ctypes.POINTER(Stats)]
        L.swro_render_primitives.restype = ctypes.c_int
        L.swro_render_primitives.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                             ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.POINTER(Stats)]
        L.swro_render_metal.restype = ctypes.c_int
        L.swro_render_metal.argtypes = L.swro_render.argtypes
        L.swro_render_shaded.restype = ctypes.c_int
        L.swro_render_shaded.argtypes = L.swro_render.argtypes + [ctypes.POINTER(ShadingC)]
        L.swro_render_metal_shaded.restype = ctypes.c_int
        L.swro_render_metal_shaded.argtypes = L.swro_render_shaded.argtypes
        L.swro_fragment.restype = None
        L.swro_fragment.argtypes = [ctypes.POINTER(ShadingC), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p]
        L.swro_interpolate.restype = ctypes.c_int64
        L.swro_interpolate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]
        L.swro_quantise.restype = ctypes.c_uint8
        L.swro_quantise.argtypes = [ctypes.c_float]
        _lib = L
    return _lib


def render(vertices: np.ndarray, indices: np.ndarray, transform: np.ndarray, width: int, height: int,
           flags: int = 0, row_begin: int = 0, row_end: int | None = None,
           color: np.ndarray | None = None, depth: np.ndarray | None = None, primitive_type: int = 0,
           shading=None):
    """Returns (color uint8[H,W,4] BGRA or None, depth float32[H,W], Stats, rc).
    shading: the extended fragment stage (scenes.Shading; triangles only)."""
    L = lib()
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    i = np.ascontiguousarray(indices, dtype=np.int64)
    m = np.ascontiguousarray(transform, dtype=np.float32)
    if row_end is None:
        row_end = height
    if color is None and not (flags & NO_COLOR):
        color = np.full((height, width, 4), 0xCD, dtype=np.uint8)
    if depth is None:
        depth = np.full((height, width), -123.0, dtype=np.float32)
    st = Stats()
    if shading is not None:
        assert primitive_type == 0
        sc, keep = _shading_c(shading)
        rc = L.swro_render_shaded(color.ctypes.data if color is not None else None, depth.ctypes.data,
                                  width, height, v.ctypes.data, v.shape[0] if v.ndim == 2 else v.size // 8,
                                  i.ctypes.data, i.size, m.ctypes.data, flags, row_begin, row_end,
                                  ctypes.byref(st), ctypes.byref(sc))
        return color, depth, st, rc
    rc = L.swro_render_primitives(color.ctypes.data if color is not None else None, depth.ctypes.data,
                                  width, height, v.ctypes.data, v.shape[0] if v.ndim == 2 else v.size // 8,
                                  i.ctypes.data, i.size, m.ctypes.data, flags, primitive_type, row_begin, row_end,
                                  ctypes.byref(st))
    return color, depth, st, rc


def render_metal(vertices, indices, transform, width: int, height: int, flags: int = 0,
                 row_begin: int = 0, row_end: int | None = None, color=None, depth=None, shading=None):
    """The Metal path's rules (Shaders.metal / GpuRenderer.swift) in IEEE arithmetic."""
    L = lib()
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    i = np.ascontiguousarray(indices, dtype=np.int64)
    m = np.ascontiguousarray(transform, dtype=np.float32)
    if row_end is None:
        row_end = height
    if color is None and not (flags & NO_COLOR):
        color = np.full((height, width, 4), 0xCD, dtype=np.uint8)
    if depth is None:
        depth = np.full((height, width), -123.0, dtype=np.float32)
    st = Stats()
    if shading is not None:
        sc, keep = _shading_c(shading)
        rc = L.swro_render_metal_shaded(color.ctypes.data if color is not None else None, depth.ctypes.data,
                                        width, height, v.ctypes.data, v.shape[0] if v.ndim == 2 else v.size // 8,
                                        i.ctypes.data, i.size, m.ctypes.data, flags, row_begin, row_end,
                                        ctypes.byref(st), ctypes.byref(sc))
        return color, depth, st, rc
    rc = L.swro_render_metal(color.ctypes.data if color is not None else None, depth.ctypes.data, width, height,
                             v.ctypes.data, v.shape[0] if v.ndim == 2 else v.size // 8, i.ctypes.data, i.size,
                             m.ctypes.data, flags, row_begin, row_end, ctypes.byref(st))
    return color, depth, st, rc


def project(vertices, transform, width, height, flags: int = 0):
    """swro_project: screen x / y (before truncation) and NDC z of every vertex."""
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 8)
    m = np.ascontiguousarray(transform, dtype=np.float32).reshape(16)
    n = v.shape[0]
    sx, sy, sz = (np.empty(n, np.float32) for _ in range(3))
    L = lib()
    L.swro_project.restype = None
    L.swro_project.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                               ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.swro_project(v.ctypes.data, n, m.ctypes.data, width, height, flags, sx.ctypes.data, sy.ctypes.data, sz.ctypes.data)
    return sx, sy, sz


def render_scene(scene, extra_flags: int = 0, **kw):
    return render(scene.vertices, scene.indices, scene.transform, scene.width, scene.height,
                  scene.flags | extra_flags, **kw)


def render_threads(scene, threads: int, extra_flags: int = 0):
    """Row-band N-thread run of the same oracle; ctypes drops the GIL."""
    H, W = scene.height, scene.width
    flags = scene.flags | extra_flags
    color = None if (flags & NO_COLOR) else np.zeros((H, W, 4), dtype=np.uint8)
    depth = np.zeros((H, W), dtype=np.float32)
    edges = [H * k // threads for k in range(threads + 1)]

    def job(k):
        return render(scene.vertices, scene.indices, scene.transform, W, H, flags,
                      edges[k], edges[k + 1], color, depth)[3]

    with ThreadPoolExecutor(max_workers=threads) as ex:
        rcs = list(ex.map(job, range(threads)))
    return color, depth, rcs


def fragment(shading, color, normal, uv) -> np.ndarray:
    """swro_fragment on one fragment: returns float32[4] r,g,b,a."""
    sc, keep = _shading_c(shading)
    c = np.ascontiguousarray(color, dtype=np.float32)
    n = np.ascontiguousarray(normal, dtype=np.float32)
    t = np.ascontiguousarray(uv, dtype=np.float32)
    out = np.zeros(4, dtype=np.float32)
    lib().swro_fragment(ctypes.byref(sc), c.ctypes.data, n.ctypes.data, t.ctypes.data, out.ctypes.data)
    return out


def interpolate(points, t: int) -> int:
    p = np.ascontiguousarray(np.asarray(points, dtype=np.int64).reshape(-1))
    return int(lib().swro_interpolate(p.ctypes.data, p.size // 2, t))


def quantise(v: float) -> int:
    return int(lib().swro_quantise(ctypes.c_float(v)))
