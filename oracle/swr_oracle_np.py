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


def render(vertices, indices, transform, width, height, depth_test=False, no_color=False,
           inv_rcp=False):
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
            if not (det != 0 and np.isfinite(det)):
                skipped += 1
                continue
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
                px = np.stack([quantise(rgb[2]), quantise(rgb[1]), quantise(rgb[0]),
                               np.full(xs.shape, 255, dtype=np.uint8)], axis=-1)
                row = color[y, x0:x1 + 1]
                row[sel] = px[sel]
    return color, depth, skipped
