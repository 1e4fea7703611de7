"""Independent NumPy restatement of renderer/Renderer.swift's triangle path (second reading of
the same source, written without looking at swr_oracle.c's structure) — TEST INFRASTRUCTURE,
PARITY UNPINNED (see swr_oracle.h).  Small cases only: Python loops over triangles and rows,
NumPy float32 vectors over the pixels of a row.  Every float op is a separate NumPy float32
ufunc call, so each intermediate is rounded to binary32 and nothing is fused.

Line numbers refer to /root/reference/renderer/Renderer.swift.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def interpolate(values, t: int) -> int:
    """Renderer.interpolate(values:t:) :467-494 on Python ints (Swift Int '/' truncates)."""
    base = 0
    if len(values) == 3:
        if t >= values[2][1]:
            base = 2
        elif t >= values[1][1]:
            base = 1
    nxt = base + 1
    sx, sy = values[base]
    if nxt >= len(values):
        return sx
    ex, ey = values[nxt]
    diff = ex - sx
    dy = ey - sy
    if dy == 0:
        return sx
    num = diff * (t - sy)
    q = abs(num) // abs(dy)                      # truncating division
    if (num < 0) != (dy < 0):
        q = -q
    return sx + q


def quantise(v: np.ndarray) -> np.ndarray:
    """Pixel.floats :117-124 — clamp, *255, truncate."""
    v = np.asarray(v, dtype=F)
    c = np.fmin(np.fmax(v, F(0.0)), F(1.0))
    return (c * F(255.0)).astype(np.uint8)


def fragment(sh, c, n, uv):
    """The extended fragment stage (swr_oracle.h, swro_fragment) on arrays of fragments: c [k,3] interpolated
    colour, n [k,3] interpolated normal, uv [k,2]; returns r,g,b,a float32 [k,4].  Written from the definition
    in swr_oracle.h, one float32 ufunc per operator."""
    c = np.asarray(c, dtype=F); n = np.asarray(n, dtype=F); uv = np.asarray(uv, dtype=F)
    k = c.shape[0]
    out = np.empty((k, 4), dtype=F)
    out[:, 3] = F(1.0)
    if sh.shader == 0:
        out[:, 0:3] = c
        return out
    with np.errstate(all="ignore"):
        len2 = n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1] + n[:, 2] * n[:, 2]
        ln = np.sqrt(len2)
        pos = len2 > 0
        N = np.zeros((k, 3), dtype=F)
        for a in range(3):
            N[pos, a] = n[pos, a] / ln[pos]
        L = [F(x) for x in sh.light_dir]
        Hh = [F(x) for x in sh.half_dir]
        ndl = np.fmax(N[:, 0] * L[0] + N[:, 1] * L[1] + N[:, 2] * L[2], F(0))
        ndh = np.fmax(N[:, 0] * Hh[0] + N[:, 1] * Hh[1] + N[:, 2] * Hh[2], F(0))
        s = ndh
        for _ in range(int(sh.shininess_log2)):
            s = s * s
        base = c.copy()
        if sh.shader == 2:
            tex = np.asarray(sh.texture, dtype=np.uint8)
            th, tw = tex.shape[0], tex.shape[1]
            fu = uv[:, 0] - np.floor(uv[:, 0])
            fv = uv[:, 1] - np.floor(uv[:, 1])
            x = fu * F(tw) - F(0.5)
            y = fv * F(th) - F(0.5)
            x0f, y0f = np.floor(x), np.floor(y)
            ax, ay = x - x0f, y - y0f
            x0, y0 = x0f.astype(np.int64), y0f.astype(np.int64)

            def texel(xi, yi):
                t = tex[np.mod(yi, th), np.mod(xi, tw)]                  # b,g,r,a
                return t[:, [2, 1, 0]].astype(F) / F(255.0)              # r,g,b

            t00, t10, t01, t11 = texel(x0, y0), texel(x0 + 1, y0), texel(x0, y0 + 1), texel(x0 + 1, y0 + 1)
            top = t00 + (t10 - t00) * ax[:, None]
            bot = t01 + (t11 - t01) * ax[:, None]
            base = base * (top + (bot - top) * ay[:, None])
        lit = F(sh.ambient) + F(sh.diffuse) * ndl
        spec = F(sh.specular) * s
        out[:, 0:3] = base * lit[:, None] + spec[:, None]
    return out


def _shaded(shading, sel_attr, w0, w1, w2, rgb, metal):
    """interpolate normal / uv like colour and run the fragment stage; rgb: list of 3 arrays."""
    na, nb, nc = sel_attr
    if metal:
        n = [w0 * na[ch] + w1 * nb[ch] + w2 * nc[ch] for ch in range(3)]
        uv = [w0 * na[4 + ch] + w1 * nb[4 + ch] + w2 * nc[4 + ch] for ch in range(2)]
    else:
        n = [na[ch] * w0 + nb[ch] * w1 + nc[ch] * w2 for ch in range(3)]
        uv = [na[4 + ch] * w0 + nb[4 + ch] * w1 + nc[4 + ch] * w2 for ch in range(2)]
    shp = w0.shape
    o = fragment(shading, np.stack([r.reshape(-1) for r in rgb], -1), np.stack([q.reshape(-1) for q in n], -1),
                 np.stack([q.reshape(-1) for q in uv], -1))
    return [o[:, 0].reshape(shp), o[:, 1].reshape(shp), o[:, 2].reshape(shp)], o[:, 3].reshape(shp)


def render(vertices, indices, transform, width, height, depth_test=False, no_color=False,
           inv_rcp=False, shading=None):
    W, H = int(width), int(height)
    color = np.zeros((H, W, 4), dtype=np.uint8)            # clear :205
    depth = np.full((H, W), np.inf, dtype=F)               # clear :206
    M = np.asarray(transform, dtype=F).reshape(4, 4)       # M[c] = column c
    V = np.asarray(vertices, dtype=F).reshape(-1, 8)
    idx = np.asarray(indices, dtype=np.int64)
    assert idx.size % 3 == 0                               # :209
    skipped = 0
    with np.errstate(all="ignore"):
        for p in range(idx.size // 3):                     # :222
            sv = []
            ok = True
            for k in range(3):
                v = V[idx[3 * p + k]]
                # Vertex.apply :160-162
                r = M[0] * v[0]
                r = r + M[1] * v[1]
                r = r + M[2] * v[2]
                r = r + M[3] * F(1.0)
                ndc = r[:3] / r[3]
                # convertedToScreen :166-168
                u = ndc[0] * F(0.5) + F(0.5)
                w = ndc[1] * F(-0.5) + F(0.5)
                sx = u * F(W)
                sy = w * F(H)
                if not (abs(sx) < F(2.0 ** 30) and abs(sy) < F(2.0 ** 30)):
                    ok = False
                sv.append((sx, sy, ndc[2], v[4:7].copy()))
            if not ok:
                skipped += 1
                continue
            ints = [(int(s[0]), int(s[1])) for s in sv]    # trunc toward zero, :251
            a, b, c = ints
            cf = (F(c[0]) + F(0.5), F(c[1]) + F(0.5))
            col0 = ((F(a[0]) + F(0.5)) - cf[0], (F(a[1]) + F(0.5)) - cf[1])   # af - cf
            col1 = ((F(b[0]) + F(0.5)) - cf[0], (F(b[1]) + F(0.5)) - cf[1])   # bf - cf
            det = col0[0] * col1[1] - col1[0] * col0[1]
            # det == 0 is not skipped: the divisions give +-inf / NaN, the clamp of :119-122 maps them to 0 / 1
            if inv_rcp:
                rdet = F(1.0) / det
                T = ((col1[1] * rdet, -col1[0] * rdet), (-col0[1] * rdet, col0[0] * rdet))
            else:
                T = ((col1[1] / det, -col1[0] / det), (-col0[1] / det, col0[0] / det))
            # stable sort on float y (:271)
            order = sorted(range(3), key=lambda k: sv[k][1])     # Python's sort is stable
            S = [ints[k] for k in order]
            for y in range(max(S[0][1], 0), min(S[2][1], H - 1) + 1):        # :275 (+ scissor :246)
                lx = interpolate([S[0], S[1], S[2]], y)
                rx = interpolate([S[0], S[2]], y)
                if lx > rx:
                    lx, rx = rx, lx
                x0, x1 = max(lx, 0), min(rx, W - 1)
                if x0 > x1:
                    continue
                xs = np.arange(x0, x1 + 1)
                dx = (xs.astype(F) + F(0.5)) - cf[0]
                dy = (F(y) + F(0.5)) - cf[1]
                w0 = T[0][0] * dx + T[0][1] * dy
                w1 = T[1][0] * dx + T[1][1] * dy
                w2 = F(1.0) - w0 - w1
                sel = np.ones(xs.shape, dtype=bool)
                if depth_test:
                    d = sv[0][2] * w0 + sv[1][2] * w1 + sv[2][2] * w2
                    sel = d < depth[y, x0:x1 + 1]
                    depth[y, x0:x1 + 1][sel] = d[sel]
                if no_color:
                    continue
                ca, cb, cc = sv[0][3], sv[1][3], sv[2][3]
                rgb = [ca[ch] * w0 + cb[ch] * w1 + cc[ch] * w2 for ch in range(3)]
                alpha = np.full(xs.shape, F(1.0), dtype=F)
                if shading is not None and shading.shader != 0:
                    A = np.asarray(shading.attrs, dtype=F).reshape(-1, 8)
                    rgb, alpha = _shaded(shading, [A[idx[3 * p + k]] for k in range(3)], w0, w1, w2, rgb, False)
                px = np.stack([quantise(rgb[2]), quantise(rgb[1]), quantise(rgb[0]), quantise(alpha)], axis=-1)
                row = color[y, x0:x1 + 1]
                row[sel] = px[sel]
    return color, depth, skipped


def render_metal(vertices, indices, transform, width, height, no_color=False, shading=None):
    """Independent NumPy restatement of the Metal path's rules (renderer/Shaders.metal:57-167,
    renderer/GpuRenderer.swift:109-139) in IEEE float32 — see swr_oracle.h for the documented choices.
    Vectorised over the ROI of each primitive."""
    W, H = int(width), int(height)
    color = np.zeros((H, W, 4), dtype=np.uint8)
    depth = np.full((H, W), np.inf, dtype=F)
    M = np.asarray(transform, dtype=F).reshape(4, 4)
    V = np.asarray(vertices, dtype=F).reshape(-1, 8)
    idx = np.asarray(indices, dtype=np.int64)
    with np.errstate(all="ignore"):
        for p in range(idx.size // 3):
            P, Z, C = [], [], []
            ok = True
            for k in range(3):
                v = V[idx[3 * p + k]]
                r = M[0] * v[0]
                r = r + M[1] * v[1]
                r = r + M[2] * v[2]
                r = r + M[3] * F(1.0)
                n = r[:3] / r[3]                                         # Shaders.metal:68
                uv = (n[0] * F(0.5) + F(0.5), n[1] * F(-0.5) + F(0.5))   # :70
                px = np.trunc(uv[0] * F(W) + np.copysign(F(0.5), uv[0] * F(W)))   # round half away (:71)
                py = np.trunc(uv[1] * F(H) + np.copysign(F(0.5), uv[1] * F(H)))
                # (the +-0.5 trick is exact here: |value| < 2^23 in every test scene)
                if not (0 <= px < 2.0 ** 30 and 0 <= py < 2.0 ** 30):
                    ok = False
                P.append((F(px), F(py)))
                Z.append(n[2])
                C.append(v[4:7].copy())
            if not ok:
                continue
            xs = [int(q[0]) for q in P]
            ys = [int(q[1]) for q in P]
            if min(xs) == 0 or min(ys) == 0:                            # GpuRenderer.swift:122-124
                continue
            (p1x, p1y), (p2x, p2y), (p3x, p3y) = P
            divider = (p1x - p3x) * (p2y - p3y) - (p2x - p3x) * (p1y - p3y)
            x0, x1 = min(xs), min(max(xs), W - 1)
            y0, y1 = min(ys), min(max(ys), H - 1)
            if x0 > x1 or y0 > y1:
                continue
            gx = (np.arange(x0, x1 + 1).astype(F) + F(0.5))[None, :]
            gy = (np.arange(y0, y1 + 1).astype(F) + F(0.5))[:, None]
            w0 = ((p2y - p3y) * (gx - p3x) + (p3x - p2x) * (gy - p3y)) / divider
            w1 = ((p3y - p1y) * (gx - p3x) + (p1x - p3x) * (gy - p3y)) / divider
            w2 = F(1.0) - w0 - w1
            inside = (w0 >= 0) & (w0 <= 1) & (w1 >= 0) & (w1 <= 1) & (w2 >= 0) & (w2 <= 1)
            z = w0 * Z[0] + w1 * Z[1] + w2 * Z[2]
            sub = depth[y0:y1 + 1, x0:x1 + 1]
            win = inside & (z < sub)
            sub[win] = z[win]
            if not no_color:
                rgb = [w0 * C[0][ch] + w1 * C[1][ch] + w2 * C[2][ch] for ch in range(3)]
                alpha = np.full(z.shape, F(1.0), dtype=F)
                if shading is not None and shading.shader != 0:
                    A = np.asarray(shading.attrs, dtype=F).reshape(-1, 8)
                    rgb, alpha = _shaded(shading, [A[idx[3 * p + k]] for k in range(3)], w0, w1, w2, rgb, True)
                un = lambda a: np.rint(np.fmin(np.fmax(a, F(0)), F(1)) * F(255)).astype(np.uint8)
                px4 = np.stack([un(rgb[2]), un(rgb[1]), un(rgb[0]), un(alpha)], axis=-1)
                csub = color[y0:y1 + 1, x0:x1 + 1]
                csub[win] = px4[win]
    return color, depth
