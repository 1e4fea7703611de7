#!/usr/bin/env python3
"""NumPy model of k_raster's producer on a scene (default cfg4): how many row steps the chunks of 64 triangles take and
how full they are.  Statistics only (degenerate edges are approximated); python tools/producer_model.py [cfg4|cfg5|cfg3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import swr_amd
S = swr_amd.scenes
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
sc = {"cfg4": S.cfg4_soup, "cfg5": S.cfg5_sponza_scale, "cfg3": S.cfg3_bunny_scale}[name]()
TW, TH, UNIT, QMAXU = 64, 32, 4, 3
W, H = sc.width, sc.height
xyz = sc.vertices.view(np.float32).reshape(-1, 8)[:, :4].astype(np.float64)
m = np.asarray(sc.transform, np.float64).reshape(4, 4)   # column-major
clip = xyz @ m if False else (m.T @ xyz.T).T if False else xyz @ m.reshape(4, 4)
# use the package's own helper for the exact integer coordinates when it exists
P = S.screen_truncated(sc.vertices.view(np.float32).reshape(-1, 8)[:, :3] if False else None, W, H) if False else None
def project():
    v = sc.vertices.view(np.float32).reshape(-1, 8)[:, :3].astype(np.float32)
    M = np.asarray(sc.transform, np.float32).reshape(4, 4)
    h = np.concatenate([v, np.ones((len(v), 1), np.float32)], 1)
    c = h @ M            # column-major storage: row-vector times the stored matrix
    ndc = c[:, :3] / c[:, 3:4]
    sx = (ndc[:, 0] + 1) * 0.5 * W
    sy = (1 - (ndc[:, 1] + 1) * 0.5) * H
    return np.trunc(sx).astype(np.int64), np.trunc(sy).astype(np.int64)
X, Y = project()
idx = np.asarray(sc.indices, np.int64).reshape(-1, 3)
tx, ty = X[idx], Y[idx]
o = np.argsort(ty, axis=1, kind="stable")
r = np.arange(len(idx))[:, None]
tx, ty = tx[r, o], ty[r, o]
ok = (tx.max(1) >= 0) & (tx.min(1) < W) & (ty[:, 2] >= 0) & (ty[:, 0] < H)
tx, ty = tx[ok], ty[ok]
n = len(tx)
y0 = np.clip(ty[:, 0], 0, H - 1); y1 = np.clip(ty[:, 2], 0, H - 1)
rows = (y1 - y0 + 1)
tri = np.repeat(np.arange(n), rows)
y = np.concatenate([np.arange(a, b + 1) for a, b in zip(y0, y1)]) if n < 200000 else None
if y is None:
    start = np.cumsum(rows) - rows
    y = np.arange(rows.sum()) - np.repeat(start, rows) + np.repeat(y0, rows)
def tdiv(a, b):
    b = np.where(b == 0, 1, b)
    q = np.abs(a) // np.abs(b)
    return np.where((a < 0) ^ (b < 0), -q, q)
def edge(xa, ya, xb, yb, yy):
    return xa + tdiv((xb - xa) * (yy - ya), yb - ya)
s0x, s1x, s2x = tx[tri, 0], tx[tri, 1], tx[tri, 2]
s0y, s1y, s2y = ty[tri, 0], ty[tri, 1], ty[tri, 2]
L = np.where(y >= s2y, s2x, np.where(y >= s1y, edge(s1x, s1y, s2x, s2y, y), edge(s0x, s0y, s1x, s1y, y)))
R = np.where(s2y == s0y, s2x, edge(s0x, s0y, s2x, s2y, y))
lo = np.maximum(np.minimum(L, R), 0); hi = np.minimum(np.maximum(L, R), W - 1)
bx0 = np.clip(tx.min(1), 0, W - 1)[tri]; bx1 = np.clip(tx.max(1), 0, W - 1)[tri]
# one record per (triangle, row, tile column of the BBOX): the producer walks the row in every tile the bbox touches
ntx = bx1 // TW - bx0 // TW + 1
rep = np.repeat(np.arange(len(y)), ntx)
st = np.cumsum(ntx) - ntx
col = np.arange(ntx.sum()) - np.repeat(st, ntx) + np.repeat(bx0 // TW, ntx)
l2 = np.maximum(lo[rep], col * TW); h2 = np.minimum(hi[rep], col * TW + TW - 1)
px = np.maximum(h2 - l2 + 1, 0)
units = (px + UNIT - 1) // UNIT
steps_row = np.maximum(1, (units + QMAXU - 1) // QMAXU)
tile = (y[rep] // TH) * ((W + TW - 1) // TW) + col
key = tile * n + tri[rep]
# per (tile, triangle) pair: rows in tile, steps
uk, inv = np.unique(key, return_inverse=True)
pair_rows = np.bincount(inv)
pair_steps = np.bincount(inv, weights=steps_row).astype(np.int64)
pair_units = np.bincount(inv, weights=units).astype(np.int64)
pair_tile = uk // n
print(f"{name}: pairs {len(uk)}, rows walked {len(rep)}, non-empty spans {(px > 0).sum()}, units {units.sum()}, "
      f"rows needing > 1 step {(steps_row > 1).sum()}, fragments {px.sum()}")
# chunks: per tile sort by rows descending, 64 per chunk; steps of a chunk = max pair_steps
order = np.lexsort((-pair_rows, pair_tile))
pt, ps, pr = pair_tile[order], pair_steps[order], pair_rows[order]
first = np.r_[0, np.flatnonzero(np.diff(pt)) + 1]
pos = np.arange(len(pt)) - np.repeat(first, np.diff(np.r_[first, len(pt)]))
cnt = np.diff(np.r_[first, len(pt)])
mtile = np.repeat(cnt, cnt)
dense = mtile > 128
chunk_id = np.cumsum(np.r_[1, (np.diff(pt) != 0) | (pos[1:] % 64 == 0)]) - 1
csteps = np.zeros(chunk_id[-1] + 1, np.int64); np.maximum.at(csteps, chunk_id, ps)
cwork = np.bincount(chunk_id, weights=ps)
cdense = np.zeros(chunk_id[-1] + 1, bool); cdense[chunk_id] = dense
for nm, sel in (("dense tiles (chunk per wave)", cdense), ("row-split tiles (x4 waves, rows/4 each)", ~cdense)):
    print(f"  {nm}: chunks {sel.sum()}, wave steps {csteps[sel].sum()}, lane-steps {int(cwork[sel].sum())}, "
          f"utilisation {cwork[sel].sum() / max(1, 64 * csteps[sel].sum()):.3f}")
print(f"  empty-span row steps: {(px == 0).sum()} of {len(px)} ({(px == 0).mean():.3f})")
hist = np.bincount(np.minimum(pair_rows, 33))
print("  rows per pair histogram:", hist.tolist())
