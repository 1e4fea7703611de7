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
# how many wave steps run with few active lanes (the tail of a chunk: its longest triangles)
srt_steps = ps  # per pair, ordered by (tile, -rows)
import collections
tail = collections.Counter()
cstart = np.flatnonzero(np.r_[1, np.diff(chunk_id)])
cend = np.r_[cstart[1:], len(ps)]
tot_steps = 0
lane_steps_tail = collections.Counter()
for T in (4, 8, 16, 24, 32):
    ws = 0; ls = 0
    for a0, b0 in zip(cstart[:20000], cend[:20000]):
        st = np.sort(ps[a0:b0])[::-1]
        smax = st[0]
        sT = st[T - 1] if len(st) >= T else 0
        ws += smax - sT                      # steps during which fewer than T lanes are active
        ls += np.maximum(st[:T - 1] - sT, 0).sum()   # lane-steps done in them
    print(f"  first 20000 chunks: steps with < {T} active lanes: {ws} of {csteps[:20000].sum()}  ({ws / csteps[:20000].sum():.3f}), lane-steps in them {ls}")
t0 = pt[len(pt) // 2]
sel = pt == t0
print("  one tile:", t0, "pairs", sel.sum())
rr, ss = pr[sel], ps[sel]
for c0 in range(0, len(rr), 64):
    print("   chunk rows", rr[c0:c0 + 64].tolist()[:64:4], " steps max", ss[c0:c0 + 64].max(), "sum", ss[c0:c0 + 64].sum(), "sorted steps top", np.sort(ss[c0:c0+64])[::-1][:10].tolist())
# --- alternative sort keys for the bins: estimated row steps instead of rows ---
pair_lo = np.full(len(uk), 1 << 30); pair_hi = np.full(len(uk), -1)
np.minimum.at(pair_lo, inv, np.where(px > 0, l2, 1 << 30)); np.maximum.at(pair_hi, inv, np.where(px > 0, h2, -1))
# clipped bbox columns as the kernel sees them (bbox of the triangle ∩ tile), not of the spans
bx0p = np.maximum(bx0[rep], col * TW); bx1p = np.minimum(bx1[rep], col * TW + TW - 1)
pair_cols = np.zeros(len(uk), np.int64); np.maximum.at(pair_cols, inv, bx1p - bx0p + 1)
def total_steps(key, label):
    order = np.lexsort((-key, pair_tile))
    pt2, ps2 = pair_tile[order], pair_steps[order]
    first = np.r_[0, np.flatnonzero(np.diff(pt2)) + 1]
    cnt = np.diff(np.r_[first, len(pt2)])
    pos = np.arange(len(pt2)) - np.repeat(first, cnt)
    cid = np.cumsum(np.r_[1, (np.diff(pt2) != 0) | (pos[1:] % 64 == 0)]) - 1
    cs = np.zeros(cid[-1] + 1, np.int64); np.maximum.at(cs, cid, ps2)
    print(f"  sort by {label}: wave steps {cs.sum()}  utilisation {pair_steps.sum() / (64 * cs.sum()):.3f}")
total_steps(pair_rows, "rows (today)")
total_steps(pair_steps, "exact steps (bound)")
total_steps(pair_rows * np.maximum(1, (pair_cols + 12) // 24), "rows * max(1,(cols+12)/24)")
total_steps(pair_rows + pair_rows * np.maximum(0, pair_cols - 12) // np.maximum(pair_cols, 1), "rows + rows*(cols-12)/cols")
total_steps(pair_rows + pair_rows * np.maximum(0, pair_cols - 12) // (2 * np.maximum(pair_cols, 1)), "rows + rows*(cols-12)/(2 cols)")
total_steps(pair_rows * 64 + pair_cols, "rows, then cols")
total_steps(pair_rows + (pair_cols > 12) * pair_rows // 3, "rows + (cols>12) rows/3")
def est(rows, cols, a, lim=12):
    e = rows.copy()
    for k in range(1, 6):
        e = e + rows * np.maximum(0, cols - a - lim * (k - 1)) // np.maximum(cols, 1)
    return e
for a in (8, 10, 12, 14):
    total_steps(est(pair_rows, pair_cols, a), f"rows + sum_k rows*(cols-{a}-12(k-1))/cols")
total_steps(2 * pair_rows + 2 * pair_rows * np.maximum(0, pair_cols - 12) // np.maximum(pair_cols, 1), "2x resolution of rows + rows*(cols-12)/cols")
# per-triangle widest span = 2*Area / height (exact for a triangle), clipped by the pair's bbox columns
ax, ay = tx[:, 0], ty[:, 0]; bx, by = tx[:, 1], ty[:, 1]; cxx, cyy = tx[:, 2], ty[:, 2]
area2 = np.abs((bx - ax) * (cyy - ay) - (cxx - ax) * (by - ay))
hh = np.maximum(ty[:, 2] - ty[:, 0], 1)
maxspan = area2 // hh + 1
pair_tri = uk % n
ms = np.minimum(maxspan[pair_tri], pair_cols)
for a in (10, 12):
    total_steps(est(pair_rows, ms, a), f"rows + sum_k rows*(ms-{a}-12(k-1))/ms, ms = min(2A/h+1, cols)")
    total_steps(est(2 * pair_rows, ms, a), f"2x: rows + sum_k rows*(ms-{a}-12(k-1))/ms")
tt = sum(np.maximum(0, pair_cols - 10 - 12 * k) for k in range(5))
e1 = pair_rows + (pair_rows * tt / np.maximum(pair_cols, 1)).astype(np.int64)
total_steps(np.minimum(e1 - 1, 62), "KERNEL FORMULA: min(rows + rows*T(cols)/cols - 1, 62)")
# --- intra-tile balance: four waves, chunks heaviest first, first four static then stolen; time of a chunk in
# "step units" = setup (5) + its producer steps + its consumer steps (units / 64 * 1.2) ---
units_pair = pair_units[order]          # same order as pt / ps (sorted by tile, rows desc)
cu = np.bincount(chunk_id, weights=units_pair)
ctime = 5.0 + csteps + cu / 64.0 * 1.2
ctile = np.zeros(chunk_id[-1] + 1, np.int64); ctile[chunk_id] = pt
def tile_times(split_first):
    tot_max = 0.0; tot_sum = 0.0
    starts = np.flatnonzero(np.r_[1, np.diff(ctile)]); ends = np.r_[starts[1:], len(ctile)]
    for a0, b0 in zip(starts, ends):
        t = list(ctime[a0:b0])
        if len(t) * 64 <= 128:   # row-split tiles: all waves walk every chunk
            w = sum(5.0 + (x - 5.0) / 4 for x in t); tot_max += w; tot_sum += 4 * w; continue
        if split_first and len(t) >= 2:
            h = 5.0 + (t[0] - 5.0) / 2
            t = [h, h] + t[1:]
        waves = [0.0] * 4
        for x in t:
            i = int(np.argmin(waves)); waves[i] += x
        tot_max += max(waves); tot_sum += sum(waves)
    return tot_max, tot_sum
for sp in (False, True):
    mx, sm = tile_times(sp)
    print(f"  tile schedule, heaviest chunk split over two waves = {sp}: sum of tile critical paths {mx:.0f}, busy fraction of the 4 waves {sm / (4 * mx):.3f}")
