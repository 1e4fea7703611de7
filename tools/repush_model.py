#!/usr/bin/env python3
"""NumPy / Python model of k_raster's dense phase for one scene (default cfg4): producer row steps and consumer
steps of a sample of chunks under different work-queue policies.  Statistics only.

  today   : lane = triangle emits up to 3 four-pixel units per row step (wider spans take another step),
            the consumer pops 64 units
  repush  : lane = triangle emits ONE entry per row (the whole span); the consumer pops 64 entries, handles
            UPX pixels of each and pushes the remainder of a longer span back (fifo: at the tail, lifo: at the head)
  +tail   : when fewer than 64 entries are left and the rows have run out, an entry is shared by 64 / pow2(n) lanes

python tools/repush_model.py [cfg4|cfg5|cfg3] [chunks]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import swr_amd
S = swr_amd.scenes
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
NCH = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
sc = {"cfg4": S.cfg4_soup, "cfg5": S.cfg5_sponza_scale, "cfg3": S.cfg3_bunny_scale}[name]()
TW, TH = 64, 32
W, H = sc.width, sc.height

def project():
    v = sc.vertices.view(np.float32).reshape(-1, 8)[:, :3].astype(np.float32)
    M = np.asarray(sc.transform, np.float32).reshape(4, 4)
    h = np.concatenate([v, np.ones((len(v), 1), np.float32)], 1)
    c = h @ M
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
ntx = bx1 // TW - bx0 // TW + 1
rep = np.repeat(np.arange(len(y)), ntx)
st = np.cumsum(ntx) - ntx
col = np.arange(ntx.sum()) - np.repeat(st, ntx) + np.repeat(bx0 // TW, ntx)
l2 = np.maximum(lo[rep], col * TW); h2 = np.minimum(hi[rep], col * TW + TW - 1)
px = np.maximum(h2 - l2 + 1, 0)
tile = (y[rep] // TH) * ((W + TW - 1) // TW) + col
key = tile * n + tri[rep]
order = np.argsort(key, kind="stable")            # records of one pair together, rows ascending
key_s, px_s = key[order], px[order]
pstart = np.flatnonzero(np.r_[1, np.diff(key_s)])
pend = np.r_[pstart[1:], len(key_s)]
pair_rows = pend - pstart
pair_tile = key_s[pstart] // n
print(f"{name}: pairs {len(pstart)}, row records {len(px_s)}, fragments {px_s.sum()}")
# chunks: per tile by rows descending, 64 per chunk
o2 = np.lexsort((-pair_rows, pair_tile))
pt = pair_tile[o2]
first = np.r_[0, np.flatnonzero(np.diff(pt)) + 1]
cnt = np.diff(np.r_[first, len(pt)])
pos = np.arange(len(pt)) - np.repeat(first, cnt)
dense = np.repeat(cnt, cnt) > 128
chunk_id = np.cumsum(np.r_[1, (np.diff(pt) != 0) | (pos[1:] % 64 == 0)]) - 1
cstart = np.flatnonzero(np.r_[1, np.diff(chunk_id)])
cend = np.r_[cstart[1:], len(pt)]
sel = [i for i in range(len(cstart)) if dense[cstart[i]]]
rng = np.random.default_rng(1)
sample = rng.choice(len(sel), size=min(NCH, len(sel)), replace=False)
chunks = []
for ci in sample:
    a, b = cstart[sel[ci]], cend[sel[ci]]
    lanes = [px_s[pstart[p]:pend[p]] for p in o2[a:b]]
    chunks.append(lanes)
total_chunks = len(sel)
print(f"dense chunks {total_chunks}, sampled {len(chunks)}")

def pow2ge(v):
    p = 1
    while p < v: p *= 2
    return p

def sim_today(lanes, UNIT=4, QMAXU=3):
    nl = len(lanes); ptr = [0] * nl; prog = [0] * nl
    q = 0; psteps = 0; csteps = 0; slots = 0
    while True:
        while q < 64 and any(ptr[i] < len(lanes[i]) for i in range(nl)):
            psteps += 1
            for i in range(nl):
                if ptr[i] < len(lanes[i]):
                    left = lanes[i][ptr[i]] - prog[i]
                    nall = (left + UNIT - 1) // UNIT
                    nu = min(nall, QMAXU)
                    q += nu
                    if nall <= QMAXU: ptr[i] += 1; prog[i] = 0
                    else: prog[i] += UNIT * QMAXU
        if q == 0: break
        take = min(q, 64); q -= take; csteps += 1; slots += 64 * UNIT
    return psteps, csteps, slots

def sim_repush(lanes, UPX=4, lifo=False, tail=False):
    nl = len(lanes); ptr = [0] * nl
    ring = []              # remaining lengths, head at index 0
    psteps = csteps = tsteps = slots = 0
    while True:
        while len(ring) < 64 and any(ptr[i] < len(lanes[i]) for i in range(nl)):
            psteps += 1
            for i in range(nl):
                if ptr[i] < len(lanes[i]):
                    if lanes[i][ptr[i]] > 0: ring.append(int(lanes[i][ptr[i]]))
                    ptr[i] += 1
        if not ring: break
        nq = min(len(ring), 64)
        J = 1
        if tail and len(ring) < 64: J = 64 // pow2ge(nq)
        if J > 1: tsteps += 1
        popped, ring = ring[:nq], ring[nq:]
        rem = [v - UPX * J for v in popped if v > UPX * J]
        ring = rem + ring if lifo else ring + rem
        csteps += 1; slots += 64 * UPX
    return psteps, csteps, tsteps, slots

frag = sum(int(l.sum()) for lanes in chunks for l in lanes)
def report(label, res):
    ps = sum(r[0] for r in res); cs = sum(r[1] for r in res); sl = sum(r[-1] for r in res)
    sc_ = total_chunks / len(chunks)
    print(f"  {label:34s}: producer steps {ps * sc_ / 1e3:7.1f} K  consumer steps {cs * sc_ / 1e3:7.1f} K  "
          f"pixel-slot use {frag / sl:.3f}  (per chunk {ps / len(chunks):.2f} / {cs / len(chunks):.2f})")
    return ps * sc_, cs * sc_
report("today (3 units / row step)", [sim_today(l) for l in chunks])
for lifo in (False, True):
    for tail in (False, True):
        report(f"repush 4px {'lifo' if lifo else 'fifo'}{' +tail' if tail else ''}", [sim_repush(l, 4, lifo, tail) for l in chunks])
report("repush 8px fifo +tail", [sim_repush(l, 8, False, True) for l in chunks])

# ---- two rings: spans (or remainders) of at most SMAX pixels wait in a ring of their own and are visited for SMAX pixels only
def sim_two_rings(lanes, UPX=4, SMAX=2):
    nl = len(lanes); ptr = [0] * nl
    L, Sr = [], []
    psteps = lsteps = ssteps = 0
    def push(v):
        (Sr if v <= SMAX else L).append(v)
    while True:
        while len(L) < 64 and len(Sr) < 64 and any(ptr[i] < len(lanes[i]) for i in range(nl)):
            psteps += 1
            for i in range(nl):
                if ptr[i] < len(lanes[i]):
                    if lanes[i][ptr[i]] > 0: push(int(lanes[i][ptr[i]]))
                    ptr[i] += 1
        rows_left = any(ptr[i] < len(lanes[i]) for i in range(nl))
        if len(L) >= 64 or (not rows_left and L):
            nq = min(len(L), 64)
            J = 1 if len(L) >= 64 else 64 // pow2ge(nq)
            popped, L = L[:nq], L[nq:]
            for v in popped:
                if v > UPX * J: push(v - UPX * J)
            lsteps += 1
        elif len(Sr) >= 64 or (not rows_left and Sr):
            Sr = Sr[64:]
            ssteps += 1
        elif not rows_left:
            break
    return psteps, lsteps, ssteps

for smax in (1, 2):
    res = [sim_two_rings(l, 4, smax) for l in chunks]
    sc_ = total_chunks / len(chunks)
    ps = sum(r[0] for r in res) * sc_; ls = sum(r[1] for r in res) * sc_; ss = sum(r[2] for r in res) * sc_
    cost_s = 17 + 17 * smax
    print(f"  two rings, short = {smax} px: producer {ps / 1e3:.1f} K, long steps {ls / 1e3:.1f} K, short steps {ss / 1e3:.1f} K  "
          f"-> consumer VALU {(ls * 93 + ss * cost_s) / 1e6:.2f} M  (one ring: {255.8e3 * 93 / 1e6:.2f} M)")

# ---- carry: fewer than CARRY entries left when the rows run out are taken into the next chunk (a small carry table keeps their
# owners' constants); the wave's last chunk drains.  Chunks are chained four at a time (a wave's share of a tile).
def sim_carry(chunk_list, UPX=4, CARRY=16, per_wave=2):
    ps = cs = 0
    ring = []
    for ci, lanes in enumerate(chunk_list):
        nl = len(lanes); ptr = [0] * nl
        last = (ci % per_wave) == per_wave - 1
        while True:
            while len(ring) < 64 and any(ptr[i] < len(lanes[i]) for i in range(nl)):
                ps += 1
                for i in range(nl):
                    if ptr[i] < len(lanes[i]):
                        if lanes[i][ptr[i]] > 0: ring.append(int(lanes[i][ptr[i]]))
                        ptr[i] += 1
            if not ring: break
            if len(ring) < 64 and not last and len(ring) <= CARRY: break          # carried
            nq = min(len(ring), 64)
            J = 64 // pow2ge(nq) if len(ring) < 64 else 1
            popped, ring = ring[:nq], ring[nq:]
            ring = ring + [v - UPX * J for v in popped if v > UPX * J]
            cs += 1
        if last: ring = []
    return ps, cs
for carry, pw in ((0, 2), (16, 2), (32, 2), (63, 2), (16, 1)):
    ps, cs = sim_carry(chunks, 4, carry, pw)
    sc_ = total_chunks / len(chunks)
    print(f"  carry <= {carry:2d} entries, {pw} chunks per wave: producer {ps * sc_ / 1e3:.1f} K, consumer steps {cs * sc_ / 1e3:.1f} K")

# ---- visit size chosen per chunk (2 / 4 / 8 pixels per lane per visit): consumer cost = visits x (OVH + PX x UPX) wave-instructions
OVH, PX = 25, 13
tot = {2: 0, 4: 0, 8: 0, "best": 0}
pick = {2: 0, 4: 0, 8: 0}
for lanes in chunks:
    c = {}
    for U in (2, 4, 8):
        _, cs, _, _ = sim_repush(lanes, U, False, True)
        c[U] = cs * (OVH + PX * U)
        tot[U] += c[U]
    b = min(c, key=c.get)
    pick[b] += 1
    tot["best"] += c[b]
sc = total_chunks / len(chunks) / 1e6
print(f"  consumer wave-instructions (model: {OVH} per visit + {PX} per pixel slot): UPX 2: {tot[2]*sc:.2f} M  4: {tot[4]*sc:.2f} M  8: {tot[8]*sc:.2f} M  "
      f"best per chunk: {tot['best']*sc:.2f} M  (chunks choosing 2 / 4 / 8: {pick[2]} / {pick[4]} / {pick[8]})")
