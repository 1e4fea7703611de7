"""Synthetic scenes for the BASELINE.json configs (inputs only — no rendering here).

The reference ships no meshes (its demo mesh is a ModelIO-generated sphere, App.swift:124) and
there is no network, so every config is procedural:

  cfg1  one flat-shaded triangle, 256x256, identity transform            (KAT)
  cfg2  "teapot-scale": closed torus, 6 320 triangles, Gouraud, 1920x1080
  cfg3  "bunny-scale":  closed torus, 69 451 triangles, z-test, 3840x2160
  cfg4  1 000 000 independent random triangles, z-test, 3840x2160        (headline)
  cfg5  "Sponza-scale": 262 144 triangles (box interior grid), 7680x4320

Random numbers are SplitMix64 -> 24-bit mantissa floats, seed 0x5EED0000 + config id, pure
integer / IEEE arithmetic (no libm), so a scene is bit-reproducible on any host.

Vertex layout = swr_vertex (Renderer.swift:154-157): float32[N, 8] = xyz, pad, rgb, pad.
Indices = int64 (Swift Int, Renderer.swift:196).  Transform = float32[16] column-major.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

FLAG_DEPTH_TEST = 1
FLAG_NO_COLOR = 2
FLAG_METAL_RULES = 4
FLAG_REAL_LINES = 8      # .line primitives: the reference's DDA (Renderer.swift:405-419) instead of its empty stub (:289-293)


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n outputs of SplitMix64 seeded with `seed`, starting at draw number `offset`."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """float32 in [0,1) with 24 random mantissa bits."""
    z = splitmix64(seed, n, offset)
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / 16777216.0)


SHADER_PASSTHROUGH, SHADER_PHONG, SHADER_TEXTURED_PHONG = 0, 1, 2


@dataclass
class Shading:
    """The extended fragment stage of include/swr.h (swr_vertex_attr / swr_material / texture)."""
    attrs: np.ndarray                       # float32 [nv, 8]: nx,ny,nz,_, u,v,_,_
    shader: int = SHADER_PHONG
    shininess_log2: int = 5
    light_dir: tuple = (0.0, 0.0, -1.0)     # unit, towards the light, in the space of the normals
    half_dir: tuple = (0.0, 0.0, -1.0)      # unit Blinn half vector
    ambient: float = 0.15
    diffuse: float = 0.8
    specular: float = 0.4
    texture: np.ndarray | None = None       # uint8 [th, tw, 4] b,g,r,a


@dataclass
class Scene:
    name: str
    width: int
    height: int
    vertices: np.ndarray            # float32 [nv, 8]
    indices: np.ndarray             # int64   [3*ntri]
    transform: np.ndarray           # float32 [16], column-major
    flags: int = 0
    meta: dict = field(default_factory=dict)
    shading: Shading | None = None

    @property
    def triangles(self) -> int:
        return int(self.indices.size // 3)


def identity() -> np.ndarray:
    return np.eye(4, dtype=np.float32).T.reshape(16).copy()


def pack_vertices(xyz: np.ndarray, rgb: np.ndarray) -> np.ndarray:
    v = np.zeros((xyz.shape[0], 8), dtype=np.float32)
    v[:, 0:3] = xyz
    v[:, 4:7] = rgb
    return v


def _quat(angle: float, axis) -> np.ndarray:
    ax = np.asarray(axis, dtype=np.float64)
    ax = ax / np.linalg.norm(ax)
    s = math.sin(angle / 2.0)
    return np.array([ax[0] * s, ax[1] * s, ax[2] * s, math.cos(angle / 2.0)])


def _quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
        aw * bw - ax * bx - ay * by - az * bz,
    ])


def _quat_matrix(q) -> np.ndarray:
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def app_transform(time: float, scale: float = 2.0) -> np.ndarray:
    """projection * Transform(scale, rotation(time), translation (0,0,1)).matrix — the matrix
    the app builds every frame (App.swift:169-183).  Returned column-major float32[16]."""
    q = _quat_mul(_quat(time, (1.0, 1.0, 0.0)), _quat(0.5 * time, (0.0, 0.0, 1.0)))
    m = np.eye(4)
    m[:3, :3] = _quat_matrix(q) * scale          # rotation * scale
    m[:3, 3] = (0.0, 0.0, 1.0)                   # translation
    p = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 1, 1]], dtype=np.float64)
    full = (p @ m).astype(np.float32)
    return np.ascontiguousarray(full.T).reshape(16)   # column-major


def torus_mesh(nu: int, nv: int, major: float, minor: float):
    """Closed torus: nu*nv vertices, 2*nu*nv triangles; colour = |normal| (App.swift:133)."""
    u = (np.arange(nu, dtype=np.float64) / nu) * 2 * math.pi
    v = (np.arange(nv, dtype=np.float64) / nv) * 2 * math.pi
    uu, vv = np.meshgrid(u, v, indexing="ij")
    cx, sx = np.cos(uu), np.sin(uu)
    cv, sv = np.cos(vv), np.sin(vv)
    pos = np.stack([(major + minor * cv) * cx, (major + minor * cv) * sx, minor * sv], axis=-1)
    nrm = np.stack([cv * cx, cv * sx, sv], axis=-1)
    xyz = pos.reshape(-1, 3).astype(np.float32)
    rgb = np.abs(nrm).reshape(-1, 3).astype(np.float32)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    a = (i * nv + j)
    b = (((i + 1) % nu) * nv + j)
    c = (((i + 1) % nu) * nv + (j + 1) % nv)
    d = (i * nv + (j + 1) % nv)
    tris = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], axis=2)  # [nu,nv,2,3]
    idx = tris.reshape(-1).astype(np.int64)
    return xyz, rgb, idx


def pack_attrs(normal: np.ndarray, uv: np.ndarray) -> np.ndarray:
    a = np.zeros((normal.shape[0], 8), dtype=np.float32)
    a[:, 0:3] = normal
    a[:, 4:6] = uv
    return a


def torus_attrs(nu: int, nv: int, u_repeat: float = 8.0, v_repeat: float = 4.0) -> np.ndarray:
    """Normals and texture coordinates of torus_mesh(nu, nv, ...), vertex for vertex."""
    u = (np.arange(nu, dtype=np.float64) / nu)
    v = (np.arange(nv, dtype=np.float64) / nv)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    cx, sx = np.cos(2 * math.pi * uu), np.sin(2 * math.pi * uu)
    cv, sv = np.cos(2 * math.pi * vv), np.sin(2 * math.pi * vv)
    nrm = np.stack([cv * cx, cv * sx, sv], axis=-1).reshape(-1, 3)
    uv = np.stack([uu * u_repeat, vv * v_repeat], axis=-1).reshape(-1, 2)
    return pack_attrs(nrm.astype(np.float32), uv.astype(np.float32))


def _unit(v) -> tuple:
    a = np.asarray(v, dtype=np.float64)
    a = a / np.linalg.norm(a)
    return tuple(float(np.float32(x)) for x in a)


def light_rig(light=(0.4, 0.6, -0.7), view=(0.0, 0.0, -1.0)):
    """(light_dir, half_dir): unit vector towards a distant light and the Blinn half vector for a distant
    viewer, both in the space of the normals."""
    l = np.asarray(_unit(light)); v = np.asarray(_unit(view))
    return _unit(l), _unit(l + v)


def checker_texture(tw: int = 256, th: int = 256, seed: int = 0x7E57, cells: int = 8) -> np.ndarray:
    """Procedural b,g,r,a texture: two-tone checker modulated by SplitMix64 noise (no asset files exist
    in the reference or the container)."""
    y, x = np.meshgrid(np.arange(th), np.arange(tw), indexing="ij")
    chk = (((x * cells) // tw + (y * cells) // th) & 1).astype(np.float64)
    noise = (splitmix64(seed, tw * th) >> np.uint64(56)).astype(np.float64).reshape(th, tw) / 255.0
    t = np.zeros((th, tw, 4), dtype=np.uint8)
    t[..., 0] = np.clip(255 * (0.25 + 0.6 * chk) * (0.7 + 0.3 * noise), 0, 255)          # b
    t[..., 1] = np.clip(255 * (0.85 - 0.5 * chk) * (0.7 + 0.3 * noise), 0, 255)          # g
    t[..., 2] = np.clip(255 * (0.35 + 0.6 * (x / tw)) * (0.7 + 0.3 * noise), 0, 255)     # r
    t[..., 3] = 255
    return t


def random_shading(nv: int, seed: int, shader: int = SHADER_PHONG, texture: np.ndarray | None = None,
                   shininess_log2: int = 4) -> Shading:
    """Random normals (any length, a few exactly zero) and texture coordinates in [-2, 3) for parity tests."""
    n = uniform01(seed ^ 0x5AD, 3 * nv).reshape(nv, 3) * np.float32(2.0) - np.float32(1.0)
    n[:: 97] = 0.0
    uv = uniform01(seed ^ 0x7EC, 2 * nv, 3 * nv).reshape(nv, 2) * np.float32(5.0) - np.float32(2.0)
    l, h = light_rig()
    if shader == SHADER_TEXTURED_PHONG and texture is None:
        texture = checker_texture(64, 32, seed)
    return Shading(pack_attrs(n, uv), shader, shininess_log2, l, h, 0.2, 0.7, 0.5, texture)


# ---------------------------------------------------------------------------------------------
def cfg1_triangle(gouraud: bool = False) -> Scene:
    """one triangle, identity transform, 256x256."""
    xyz = np.array([[0.0, 0.5, 0.5], [0.5, -0.5, 0.5], [-0.5, -0.5, 0.5]], dtype=np.float32)
    if gouraud:
        rgb = np.eye(3, dtype=np.float32)
    else:
        rgb = np.tile(np.array([1.0, 0.5, 0.25], dtype=np.float32), (3, 1))
    return Scene("cfg1_triangle" + ("_gouraud" if gouraud else ""), 256, 256,
                 pack_vertices(xyz, rgb), np.arange(3, dtype=np.int64), identity(), 0)


def cfg2_teapot_scale(width: int = 1920, height: int = 1080, time: float = 1.0,
                      nu: int = 40, nv: int = 79) -> Scene:
    xyz, rgb, idx = torus_mesh(nu, nv, 0.25, 0.11)
    return Scene("cfg2_teapot_scale", width, height, pack_vertices(xyz, rgb), idx,
                 app_transform(time), 0, {"mesh": f"torus {nu}x{nv}"})


def cfg3_bunny_scale(width: int = 3840, height: int = 2160, time: float = 1.0,
                     nu: int = 194, nv: int = 179, ntri: int | None = 69451) -> Scene:
    xyz, rgb, idx = torus_mesh(nu, nv, 0.25, 0.11)
    if ntri is not None:
        idx = idx[: 3 * ntri].copy()
    return Scene("cfg3_bunny_scale", width, height, pack_vertices(xyz, rgb), idx,
                 app_transform(time), FLAG_DEPTH_TEST, {"mesh": f"torus {nu}x{nv}"})


def _disc_offsets(seed: int, n: int, offset: int, tries: int = 8) -> np.ndarray:
    """n points uniform in the unit disc by rejection from the square, using only IEEE
    arithmetic: candidate k of point i comes from draws (2*(i*tries+k), +1)."""
    u = uniform01(seed, 2 * n * tries, offset).reshape(n, tries, 2)
    c = u * np.float32(2.0) - np.float32(1.0)
    inside = (c[..., 0] * c[..., 0] + c[..., 1] * c[..., 1]) <= np.float32(1.0)
    first = np.argmax(inside, axis=1)
    pick = c[np.arange(n), first]
    none = ~inside.any(axis=1)
    pick[none] = 0.0
    return pick


def uniform01_at(seed: int, index: np.ndarray) -> np.ndarray:
    """uniform01 at arbitrary draw numbers (SplitMix64 is counter-based): float32 like uniform01(seed, n)[index]."""
    with np.errstate(over="ignore"):
        i = np.asarray(index, dtype=np.uint64) + np.uint64(1)
        z = np.uint64(seed) + i * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / 16777216.0)


def screen_truncated(xyz: np.ndarray, width: int, height: int):
    """Truncated screen coordinates of NDC positions under the identity transform, with the pipeline's own
    float32 operations (Renderer.swift:166-168, :251): returns int64 (sx, sy)."""
    f = np.float32
    sx = ((xyz[..., 0] * f(0.5) + f(0.5)) * f(width)).astype(np.int64)
    sy = ((xyz[..., 1] * f(-0.5) + f(0.5)) * f(height)).astype(np.int64)
    return sx, sy


def degenerate_mask(xyz: np.ndarray, width: int, height: int) -> np.ndarray:
    """[ntri] bool: the three truncated screen vertices are collinear (det of T() == 0, Renderer.swift:95-100)."""
    sx, sy = screen_truncated(xyz, width, height)
    det = (sx[:, 0] - sx[:, 2]) * (sy[:, 1] - sy[:, 2]) - (sx[:, 1] - sx[:, 2]) * (sy[:, 0] - sy[:, 2])
    return det == 0


CFG4_REDRAW_ROUNDS = 12   # a triangle is degenerate with p ~ 1.7 %: 12 independent redraws leave none at 1 M


def cfg4_soup(ntri: int = 1_000_000, width: int = 3840, height: int = 2160,
              r_ndc: float = 0.008, depth_only: bool = True, seed: int = 0x5EED0004) -> Scene:
    """Independent random triangles (3 unshared vertices each): centre uniform in
    [-0.98,0.98]^2, vertex offsets uniform in a disc of radius r_ndc, z uniform [0.05,0.95],
    colours uniform, identity transform (w = 1), z-test ON.  Triangles whose
    truncated screen coordinates are degenerate (det = 0) are regenerated, as §8(d) specifies: redraw round t
    takes all 62 random numbers of triangle i from the stream seeded seed + t * 2^24 at draws 62 i .. 62 i + 61."""
    centre = uniform01(seed, 2 * ntri, 0).reshape(ntri, 2) * np.float32(1.96) - np.float32(0.98)
    off = _disc_offsets(seed, 3 * ntri, 2 * ntri).reshape(ntri, 3, 2) * np.float32(r_ndc)
    base = 2 * ntri + 2 * 3 * ntri * 8
    z = uniform01(seed, 3 * ntri, base).reshape(ntri, 3) * np.float32(0.9) + np.float32(0.05)
    rgb = uniform01(seed, 9 * ntri, base + 3 * ntri).reshape(ntri, 3, 3)
    xyz = np.empty((ntri, 3, 3), dtype=np.float32)
    xyz[:, :, 0:2] = centre[:, None, :] + off
    xyz[:, :, 2] = z
    redrawn = 0
    bad = np.nonzero(degenerate_mask(xyz, width, height))[0]
    for t in range(1, CFG4_REDRAW_ROUNDS + 1):
        if bad.size == 0:
            break
        redrawn += int(bad.size)
        u = uniform01_at(seed + t * (1 << 24), bad[:, None].astype(np.uint64) * np.uint64(62) + np.arange(62, dtype=np.uint64))
        c = u[:, 0:2] * np.float32(1.96) - np.float32(0.98)
        cand = (u[:, 2:50].reshape(-1, 3, 8, 2)) * np.float32(2.0) - np.float32(1.0)      # 8 disc candidates per vertex
        inside = (cand[..., 0] * cand[..., 0] + cand[..., 1] * cand[..., 1]) <= np.float32(1.0)
        first = np.argmax(inside, axis=2)
        pick = np.take_along_axis(cand, first[:, :, None, None], axis=2)[:, :, 0, :]
        pick[~inside.any(axis=2)] = 0.0
        xyz[bad, :, 0:2] = c[:, None, :] + pick * np.float32(r_ndc)
        xyz[bad, :, 2] = u[:, 50:53] * np.float32(0.9) + np.float32(0.05)
        rgb[bad] = u[:, 53:62].reshape(-1, 3, 3)
        bad = bad[degenerate_mask(xyz[bad], width, height)]
    flags = FLAG_DEPTH_TEST | (FLAG_NO_COLOR if depth_only else 0)
    return Scene("cfg4_soup", width, height, pack_vertices(xyz.reshape(-1, 3), rgb.reshape(-1, 3)),
                 np.arange(3 * ntri, dtype=np.int64), identity(), flags,
                 {"r_ndc": r_ndc, "seed": seed, "degenerate_redrawn": redrawn, "degenerate_left": int(bad.size)})


def cfg3_phong(**kw) -> Scene:
    """BASELINE config 3 as named: the bunny-scale mesh with per-pixel Phong + z-buffer at 4K."""
    sc = cfg3_bunny_scale(**kw)
    nu, nv = (int(t) for t in sc.meta["mesh"].split()[1].split("x"))
    l, h = light_rig()
    sc.shading = Shading(torus_attrs(nu, nv), SHADER_PHONG, 5, l, h, 0.15, 0.8, 0.4, None)
    sc.name = "cfg3_phong"
    return sc


def cfg5_textured(tex: int = 1024, **kw) -> Scene:
    """BASELINE config 5 as named: the Sponza-scale grid, textured + Phong at 8K."""
    sc = cfg5_sponza_scale(**kw)
    xyz = sc.vertices[:, 0:3].astype(np.float64)
    # analytic normals of z = z0 + 0.1 sin(3x) cos(2y) (before the per-wall xy scale; good enough for lighting)
    nrm = np.stack([-0.3 * np.cos(3 * xyz[:, 0]) * np.cos(2 * xyz[:, 1]),
                    0.2 * np.sin(3 * xyz[:, 0]) * np.sin(2 * xyz[:, 1]), -np.ones(xyz.shape[0])], axis=-1)
    uv = xyz[:, 0:2] * 6.0
    l, h = light_rig()
    sc.shading = Shading(pack_attrs(nrm.astype(np.float32), uv.astype(np.float32)), SHADER_TEXTURED_PHONG, 4,
                         l, h, 0.2, 0.8, 0.3, checker_texture(tex, tex))
    sc.name = "cfg5_textured"
    return sc


def cfg5_sponza_scale(width: int = 7680, height: int = 4320, nx: int = 512, ny: int = 256) -> Scene:
    """Two facing grids of quads (a 'box interior' front/back wall pair): nx*ny*2 triangles
    per wall pair = 262 144 for 512x256."""
    gx = np.linspace(-0.95, 0.95, nx // 2 + 1, dtype=np.float64)
    gy = np.linspace(-0.95, 0.95, ny + 1, dtype=np.float64)
    xx, yy = np.meshgrid(gx, gy, indexing="ij")
    verts, cols, idxs = [], [], []
    base = 0
    for wall, zval in enumerate((0.8, 0.3)):
        zz = zval + 0.1 * np.sin(3.0 * xx) * np.cos(2.0 * yy)
        xyz = np.stack([xx * (1.0 if wall == 0 else 0.6), yy * (1.0 if wall == 0 else 0.6), zz], -1).reshape(-1, 3)
        rgb = np.stack([0.5 + 0.5 * np.sin(7 * xx + wall), 0.5 + 0.5 * np.cos(5 * yy), np.full_like(xx, 0.3 + 0.4 * wall)], -1).reshape(-1, 3)
        ncol = ny + 1
        i = np.arange(nx // 2)[:, None]
        j = np.arange(ny)[None, :]
        a = i * ncol + j
        b = (i + 1) * ncol + j
        c = (i + 1) * ncol + j + 1
        d = i * ncol + j + 1
        tris = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], axis=2).reshape(-1) + base
        verts.append(xyz.astype(np.float32)); cols.append(rgb.astype(np.float32)); idxs.append(tris.astype(np.int64))
        base += xyz.shape[0]
    return Scene("cfg5_sponza_scale", width, height,
                 pack_vertices(np.concatenate(verts), np.concatenate(cols)),
                 np.concatenate(idxs), identity(), FLAG_DEPTH_TEST, {})


def random_soup(ntri: int, width: int, height: int, seed: int, r_ndc: float = 0.1,
                flags: int = 0, margin: float = 1.2, shared: bool = False) -> Scene:
    """General random triangles for parity tests; with margin > 1 some triangles straddle or
    leave the screen (exercises the per-pixel scissor, Renderer.swift:246-250)."""
    centre = (uniform01(seed, 2 * ntri, 0).reshape(ntri, 2) * np.float32(2.0) - np.float32(1.0)) * np.float32(margin)
    off = (uniform01(seed, 6 * ntri, 2 * ntri).reshape(ntri, 3, 2) * np.float32(2.0) - np.float32(1.0)) * np.float32(r_ndc)
    z = uniform01(seed, 3 * ntri, 8 * ntri).reshape(ntri, 3)
    rgb = uniform01(seed, 9 * ntri, 11 * ntri).reshape(3 * ntri, 3) * np.float32(1.2) - np.float32(0.1)
    xyz = np.empty((ntri, 3, 3), dtype=np.float32)
    xyz[:, :, 0:2] = centre[:, None, :] + off
    xyz[:, :, 2] = z
    idx = np.arange(3 * ntri, dtype=np.int64)
    if shared and ntri > 1:
        # re-use vertices between triangles (exercise the index path)
        idx = (splitmix64(seed ^ 0xABCDEF, 3 * ntri) % np.uint64(3 * ntri)).astype(np.int64)
    return Scene(f"soup_{ntri}_{seed:x}", width, height, pack_vertices(xyz.reshape(-1, 3), rgb),
                 idx, identity(), flags, {})


def pixel_to_ndc(px, py, width: int, height: int):
    """NDC coordinates whose screen position is (px, py) under the identity transform (exact for the dyadic
    sizes the tests use: x = px / W * 2 - 1, y = 1 - py / H * 2)."""
    return (np.float32(px) / np.float32(width) * np.float32(2.0) - np.float32(1.0),
            np.float32(1.0) - np.float32(py) / np.float32(height) * np.float32(2.0))


def degenerate_mix(width: int = 256, height: int = 128, seed: int = 0x5EED0300, flags: int = 0,
                   only_degenerate: bool = False) -> Scene:
    """Triangles whose TRUNCATED vertices are collinear (det of T() == 0, Renderer.swift:95-100) between ordinary
    ones: horizontal, vertical and diagonal runs, single points, with zero / negative / mixed-sign z and colours.
    The reference draws their scanline span with +-inf / NaN weights (clamped at :119-122; a NaN depth fails :258)."""
    soup = random_soup(120, width, height, seed, r_ndc=0.2, margin=1.05)
    lines = [  # pixel-space vertex triples (a, b, c): all collinear after truncation
        ((10.5, 20.5), (30.5, 20.5), (20.5, 20.5)),      # horizontal, c in the middle (SURVEY-style KAT in tests)
        ((40.25, 30.75), (40.75, 60.25), (40.5, 45.5)),  # vertical
        ((60.5, 10.5), (80.5, 30.5), (70.5, 20.5)),      # diagonal
        ((90.5, 90.5), (90.6, 90.4), (90.7, 90.9)),      # a single pixel
        ((100.5, 50.5), (120.5, 50.5), (140.5, 50.5)),   # horizontal, c at the end
        ((150.5, 70.5), (130.5, 60.5), (170.5, 80.5)),   # slope 1/2
        ((200.5, 100.5), (200.5, 100.5), (220.5, 110.5)),  # two coincident vertices
        ((-5.5, 64.5), (300.5, 64.5), (100.5, 64.5)),    # leaves the screen on both sides
    ]
    zs = [(0.5, 0.5, 0.5), (-0.5, -0.25, 0.75), (0.0, 0.0, 0.0), (-1.0, -1.0, 2.0), (0.25, -0.5, 0.5)]
    cols = [(1.0, 0.5, 0.25), (0.0, 0.0, 0.0), (-0.5, 2.0, 0.5), (1.0, 0.0, 0.0)]
    xyz, rgb = [], []
    k = 0
    for rep in range(5):
        for tri in lines:
            z = zs[(k + rep) % len(zs)]
            for j, (px, py) in enumerate(tri):
                nx, ny = pixel_to_ndc(px + rep * 3, py + rep, width, height)
                xyz.append((nx, ny, z[j]))
                rgb.append(cols[(k + j) % len(cols)])
            k += 1
    dv = pack_vertices(np.asarray(xyz, np.float32), np.asarray(rgb, np.float32))
    if only_degenerate:
        return Scene("degenerate_only", width, height, dv, np.arange(dv.shape[0], dtype=np.int64), identity(), flags, {})
    nv0 = soup.vertices.shape[0]
    # interleave: ordinary triangles before, between and after the degenerate ones (painter's order matters)
    half = (soup.indices.size // 6) * 3
    idx = np.concatenate([soup.indices[:half], nv0 + np.arange(dv.shape[0], dtype=np.int64), soup.indices[half:]])
    return Scene("degenerate_mix", width, height, np.concatenate([soup.vertices, dv]), idx, identity(), flags, {})


def occluded_soup(ntri: int = 1_000_000, width: int = 3840, height: int = 2160, z_occluder: float = 0.5,
                  depth_only: bool = True, tilt: float = 0.0, **kw) -> Scene:
    """The cfg4 soup behind (and in front of) a screen-filling quad at depth z_occluder (two big triangles, optionally
    tilted in z): what hierarchical early-z is for — whole tiles whose farthest stored depth is nearer than every
    fragment of the small triangles that come later in the bin.  z_occluder = 0.5 hides about half of the soup."""
    sc = cfg4_soup(ntri=ntri, width=width, height=height, depth_only=depth_only, **kw)
    z0, z1 = z_occluder - tilt, z_occluder + tilt
    quad = np.array([[-1.2, -1.2, z0], [1.2, -1.2, z0], [1.2, 1.2, z1], [-1.2, -1.2, z0], [1.2, 1.2, z1], [-1.2, 1.2, z1]], np.float32)
    qv = pack_vertices(quad, np.tile(np.float32([0.2, 0.4, 0.9]), (6, 1)))
    nv0 = sc.vertices.shape[0]
    half = (sc.indices.size // 6) * 3
    idx = np.concatenate([sc.indices[:half], nv0 + np.arange(6, dtype=np.int64), sc.indices[half:]])
    return Scene("occluded_soup", width, height, np.concatenate([sc.vertices, qv]), idx, sc.transform, sc.flags,
                 {**sc.meta, "z_occluder": z_occluder})
