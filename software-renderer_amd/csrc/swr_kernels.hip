// swr_kernels.hip — gfx950 (CDNA4, wave64) kernels of the triangle hot path.
//
// Input of every frame is the triangle stream built once per scene by swr_upload.hip: primitives in
// Morton order of their centroid, de-indexed (tri_xyz / tri_rgb / tri_nrm per slot corner), with the
// bounding box of every 64-slot group.  Bins hold slots; visibility keys hold ORIGINAL indices.
//
// One frame (no host round trip; binning runs on its own stream, one frame ahead of the raster):
//   k_setup_hist  1 lane / triangle : first, one lane per owned 64-primitive group: groups whose projected box
//                                     provably misses the band / the framebuffer are dropped; then per
//                                     triangle: vertex_shader x3, /w, screen map, truncation (or round() under
//                                     the Metal rules), y-sort, validity via T(); 32-B GeomRec;
//                                     band-clipped pixel bbox (8 B/triangle); per-workgroup tile
//                                     histogram in LDS -> row of the (workgroup x tile) matrix
//   k_colscan     16 tiles / block  : exclusive prefix of every matrix column over workgroups
//   k_fill_lds    1 lane / triangle : every workgroup scans the tile totals in LDS, seeds its cursors
//                                     with tile_start + matrix row, bins[ds_add_rtn(cursor)] = prim|class
//                                     (no global atomics anywhere; k_setup_bin / k_scan / k_fill are the
//                                     global-atomic fallback for tile tables that do not fit LDS)
//   k_sort_bins   1 workgroup / tile: appends the frame's deferred triangles (k_bin: those that cover more than 128 tiles)
//                                     that touch the tile, then counting sort of the bin by size class (rows inside the tile)
//   k_raster      1 workgroup / tile: 64-bit visibility keys of the tile live in LDS.  Producer, lane =
//                                     triangle: two integer edge steppers (a DDA of Renderer.interpolate) give
//                                     the span of the lane's next row; ONE ring entry per span (owner lane, x,
//                                     y, length), placed by a ballot + v_mbcnt prefix.  Consumer, lane = entry:
//                                     the owner's constants from an LDS table, 4 x (weights, depth, ds_min_u64)
//                                     for the first four pixels of the span; the rest of a longer span goes
//                                     back into the ring.  Resolve: key -> winning primitive -> barycentric
//                                     colour -> fragment_shader -> one 16-B/lane framebuffer store per
//                                     4 pixels (clear fused: HBM sees each pixel exactly once).
//   k_raster<.., METAL, COLOR>      : the same frame under the Metal path's rules (SWR_FLAG_METAL_RULES); colour and depth-only
//                                     (SWR_FLAG_NO_COLOR) frames are separate kernels (own resolve, own register allocation)
//   k_points / k_points_resolve     : PrimitiveType .vertices;  k_clear_band: .line (reference stub)
//   k_raster<.., EXT>               : + the extended fragment stage at the resolve (normal / uv varyings,
//                                     Blinn-Phong, bilinear texture; swr_shaders.hip.h)
//   k_split_scene, k_validate_indices, k_texture_to_float: once per upload
//
// Semantics restated from renderer/Renderer.swift (reference file:line cited inline):
//   visibility without z-test = highest primitive index covering the pixel (painter's order of
//   the serial loop :222); with z-test = smallest depth, ties -> lowest primitive index (strict
//   '<' of :258).  Both are order-independent functions of the fragment set, so a commutative
//   atomic min over (orderable depth << 32 | prim) [z] or (~prim) [no z] reproduces the serial
//   loop bit for bit regardless of the order in which lanes, waves and bins deliver fragments.
//
// Float discipline: compiled with -ffp-contract=off; only + - * / in the order of the
// reference; IEEE-correct division; the same expressions in setup, raster and resolve.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "swr_internal.h"
#include "swr_shaders.hip.h"

namespace swr {

#define COORD_LIMIT 1073741824.0f  // 2^30, same skip rule as the oracle

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t orderable_depth(float d) {
    // monotone map float -> uint32 (valid for non-NaN): negative floats reversed, positive offset
    uint32_t u = __float_as_uint(d);
    return u ^ (uint32_t)(((int32_t)u >> 31) | 0x80000000);
}
__device__ __forceinline__ float depth_from_orderable(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(u);
}

// trunc(n / d) for d > 0, |n| < 2^31, |n/d| < 2^20: float estimate + exact integer correction.
__device__ __forceinline__ int tdiv_small(int n, int d, float rcp_d) {
    int an = n < 0 ? -n : n;
    int q = (int)((float)an * rcp_d);
    int r = an - q * d;
    if (r < 0) { q -= 1; r += d; }
    if (r >= d) { q += 1; }
    return n < 0 ? -q : q;
}

// Exact a / d and a % d for 0 <= a < 2^31, 0 < d, a / d < 2^20: float estimate (rcp_d ~ 1/d) + integer correction.
__device__ __forceinline__ void udivmod_small(int a, int d, float rcp_d, int& q, int& r) {
    q = (int)((float)a * rcp_d);
    r = a - q * d;
    if (r < 0) { q -= 1; r += d; }
    if (r >= d) { q += 1; r -= d; }
}

// One edge of Renderer.interpolate (:467-494) walked row by row: X(y) = x0 + trunc((x1 - x0) * (y - y0) / (y1 - y0)).
// The dividend grows by |x1 - x0| per row, so quotient and remainder are carried instead of divided again
// (a DDA): r += |D| % dy; on r >= dy the quotient takes one extra step.  Exactly the truncating division of :492
// because the sign of the dividend is the sign of D for every y >= y0.  All state is small integers (GEOM_SMALL).
struct EdgeStep {
    int X;          // x0 + sgn * floor(|D| * (y - y0) / dy) at the current row
    int r;          // (|D| * (y - y0)) % dy
    int ss, ss1;    // sgn * (|D| / dy) and that plus one more step of sgn
    int rs, dy;     // |D| % dy, dy
};
__device__ __forceinline__ void edge_consts(int xa, int ya, int xb, int yb, EdgeStep& e, float& rcp) {
    int D = xb - xa, dy = yb - ya;
    if (dy <= 0) { D = 0; dy = 1; }              // never evaluated by :467-494 (dy == 0 returns the start point)
    const int a = D < 0 ? -D : D, sgn = D < 0 ? -1 : 1;
    rcp = __builtin_amdgcn_rcpf((float)dy);
    int qs;
    udivmod_small(a, dy, rcp, qs, e.rs);
    e.dy = dy;
    e.ss = sgn * qs;
    e.ss1 = e.ss + sgn;
    e.X = xa;
    e.r = 0;
}
// move the edge to row ya + k (k >= 0, |D| * k < 2^31)
__device__ __forceinline__ void edge_jump(int xa, int xb, int k, float rcp, EdgeStep& e) {
    const int D = xb - xa;
    const int a = D < 0 ? -D : D, sgn = D < 0 ? -1 : 1;
    int q;
    udivmod_small(a * k, e.dy, rcp, q, e.r);
    e.X = xa + sgn * q;
}
__device__ __forceinline__ void edge_next_row(EdgeStep& e) {
    // (0 <= r < dy, 0 <= rs < dy: unsigned; the borrow of ONE subtraction is the comparison — five vector instructions per edge, not six)
    const uint32_t r2 = (uint32_t)e.r + (uint32_t)e.rs;
    uint32_t t;
    const bool below = __builtin_sub_overflow(r2, (uint32_t)e.dy, &t);
    e.r = (int)(below ? r2 : t);
    e.X += below ? e.ss : e.ss1;
}

// Sorted integer vertices + chain data of one triangle, as the span walker needs them.
struct Chains {
    int s0x, s0y, s1x, s1y, s2x, s2y;
    float r01, r12, r02;   // reciprocals of the three dy's (fast path only)
    bool small;
};

// Renderer.interpolate (:467-494) for both chains of row y, then the swap of :278-280.
// Returns the inclusive span [lo, hi].
__device__ __forceinline__ void row_span(const Chains& c, int y, int& lo, int& hi) {
    // left chain: 3 points [S0,S1,S2] (:469-475 base selection)
    const bool last = y >= c.s2y;
    const bool seg1 = y >= c.s1y;
    int x0 = last ? c.s2x : (seg1 ? c.s1x : c.s0x);
    int y0 = seg1 ? c.s1y : c.s0y;
    int dx = last ? 0 : (seg1 ? c.s2x - c.s1x : c.s1x - c.s0x);
    int dy = last ? 1 : (seg1 ? c.s2y - c.s1y : c.s1y - c.s0y);
    float rd = seg1 ? c.r12 : c.r01;
    // right chain: 2 points [S0,S2]; dy == 0 -> S0.x (:486-488)
    const int dyr = c.s2y - c.s0y;
    const int dxr = dyr ? c.s2x - c.s0x : 0;
    int L, R;
    if (c.small) {
        L = x0 + tdiv_small(dx * (y - y0), dy, last ? 1.0f : rd);
        R = c.s0x + tdiv_small(dxr * (y - c.s0y), dyr ? dyr : 1, dyr ? c.r02 : 1.0f);
    } else {
        L = x0 + (int)(((int64_t)dx * (int64_t)(y - y0)) / (int64_t)dy);
        R = c.s0x + (int)(((int64_t)dxr * (int64_t)(y - c.s0y)) / (int64_t)(dyr ? dyr : 1));
    }
    lo = L < R ? L : R;
    hi = L < R ? R : L;
}

// row_span for triangles flagged GEOM_SMALL only (phase 1 of k_raster): no 64-bit path.
__device__ __forceinline__ void row_span_small(const Chains& c, int y, int& lo, int& hi) {
    const bool last = y >= c.s2y;
    const bool seg1 = y >= c.s1y;
    const int x0 = last ? c.s2x : (seg1 ? c.s1x : c.s0x);
    const int y0 = seg1 ? c.s1y : c.s0y;
    const int dx = last ? 0 : (seg1 ? c.s2x - c.s1x : c.s1x - c.s0x);
    const int dy = last ? 1 : (seg1 ? c.s2y - c.s1y : c.s1y - c.s0y);
    const float rd = last ? 1.0f : (seg1 ? c.r12 : c.r01);
    const int dyr = c.s2y - c.s0y;
    const int dxr = dyr ? c.s2x - c.s0x : 0;
    const int L = x0 + tdiv_small(dx * (y - y0), dy, rd);
    const int R = c.s0x + tdiv_small(dxr * (y - c.s0y), dyr ? dyr : 1, dyr ? c.r02 : 1.0f);
    lo = min(L, R);
    hi = max(L, R);
}

// ------------------------------------------------------------------------------------------
// index validation (Swift would trap on an out-of-range index, Renderer.swift:226)
// ------------------------------------------------------------------------------------------
__global__ void k_validate_indices(const int64_t* __restrict__ idx, int64_t n, int64_t nv,
                                   uint32_t* counters) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (; i < n; i += stride) {
        int64_t v = idx[i];
        bad |= (v < 0) | (v >= nv);
    }
    if (bad) atomicOr(&counters[CNT_BAD_INDEX], 1u);
}

// swr_texture_upload: Pixel (b,g,r,a bytes) -> (r,g,b,a) floats, channel / 255.0f (IEEE division, once).
// (A layout with the 2 x 2 texels of a bilinear fetch stored together, 48 B per texel and one read per shaded pixel, measured
// 3 % slower on BASELINE config 5 textured: profiles/r04/texture_quads_ab.txt.)
__global__ void k_texture_to_float(const uint32_t* __restrict__ bgra, int64_t n, float4* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t t = bgra[i];
        out[i] = make_float4((float)((t >> 16) & 0xFFu) / 255.0f, (float)((t >> 8) & 0xFFu) / 255.0f,
                             (float)(t & 0xFFu) / 255.0f, (float)(t >> 24) / 255.0f);
    }
}

// ------------------------------------------------------------------------------------------
// k_setup_bin
// ------------------------------------------------------------------------------------------
struct SetupArgs {
    const float4* tri_xyz;  // [3*ntri] the triangle stream (swr_upload.hip): corner positions per sorted slot
    const float4* box64;    // [2*ceil(ntri/64)] object-space box of every 64-slot group
    int reordered;          // slots are a permutation of the primitives: original index -> GeomRec.flags
    int cull;               // test the group boxes against the band
    int64_t ntri;
    GeomRec* geo;
    GeomFull* geo_full;
    uint32_t* tile_count;
    uint2* ranges;
    Target tg;
    float4x4 m;
    int metal;              // SWR_FLAG_METAL_RULES: Shaders.metal / GpuRenderer.swift rules
};

constexpr uint32_t RANGE_NONE_X = 0x00000001u;   // tx0 = 1, tx1 = 0: empty rectangle

// T() (:95-100): columns (af - cf), (bf - cf) on (int + 0.5) floats; inverse = adjugate / determinant.
// One definition for setup (validity), raster (weights) and resolve, so the bits always agree.
__device__ __forceinline__ float tinv_of(int ax, int ay, int bx, int by, int cx, int cy,
                                         float& t00, float& t01, float& t10, float& t11) {
    const float cfx = (float)cx + 0.5f, cfy = (float)cy + 0.5f;
    const float m00 = ((float)ax + 0.5f) - cfx, m10 = ((float)ay + 0.5f) - cfy;
    const float m01 = ((float)bx + 0.5f) - cfx, m11 = ((float)by + 0.5f) - cfy;
    const float det = m00 * m11 - m01 * m10;
    // Four IEEE divisions by the same divisor.  The compiler's f32 division is: scale, rcp, one Newton step on the
    // reciprocal, quotient, two residual corrections (fma), unscale, fix-up of the special cases.  Here the operands
    // are integer-valued floats (|n| < 2^31, |det| < 2^62 for coordinates below 2^30): nothing needs scaling, no
    // intermediate leaves the normal range, and the reciprocal refinement is shared between the four quotients.
    // v_div_fixup restores what the refinement loses: the sign of a zero quotient (-0 / det), x / 0 = +-inf and
    // 0 / 0 = NaN of a degenerate triangle (det == 0), NaN propagation.  Correctly rounded, bit for bit what '/' gives
    // (the oracle divides plainly; every parity test goes through here).
    const float rc0 = __builtin_amdgcn_rcpf(det);
    const float rcp = __builtin_fmaf(__builtin_fmaf(-det, rc0, 1.0f), rc0, rc0);
    auto div_exact = [&](float n) {
        const float q0 = n * rcp;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-det, q0, n), rcp, q0);
        const float q2 = __builtin_fmaf(__builtin_fmaf(-det, q1, n), rcp, q1);
        return __builtin_amdgcn_div_fixupf(q2, det, n);
    };
    t00 = div_exact(m11); t01 = div_exact(-m01); t10 = div_exact(-m10); t11 = div_exact(m00);
    return det;
}

// Integer vertices of primitive `prim` from its compact (or, if not GEOM_SMALL, full) record.
__device__ __forceinline__ void decode_vertices(const GeomFull* __restrict__ full, uint32_t prim,
                                                const int4& q0, const float4& q1, int vx[3], int vy[3]);
__device__ __forceinline__ void load_vertices(const GeomRec* __restrict__ geo, const GeomFull* __restrict__ full,
                                              uint32_t prim, int4& q0, float4& q1, int vx[3], int vy[3]) {
    q0 = reinterpret_cast<const int4*>(geo + prim)[0];
    q1 = reinterpret_cast<const float4*>(geo + prim)[1];
    decode_vertices(full, prim, q0, q1, vx, vy);
}
__device__ __forceinline__ void decode_vertices(const GeomFull* __restrict__ full, uint32_t prim,
                                                const int4& q0, const float4& q1, int vx[3], int vy[3]) {
    vx[0] = q0.x; vy[0] = q0.y;
    vx[1] = q0.x + (int)(short)(q0.z & 0xFFFF); vy[1] = q0.y + (q0.z >> 16);
    vx[2] = q0.x + (int)(short)(q0.w & 0xFFFF); vy[2] = q0.y + (q0.w >> 16);
    if (!(__float_as_uint(q1.w) & GEOM_SMALL)) {
        const int4 f0 = reinterpret_cast<const int4*>(full + prim)[0];
        const int4 f1 = reinterpret_cast<const int4*>(full + prim)[1];
        vx[0] = f0.x; vy[0] = f0.y; vx[1] = f0.z; vy[1] = f0.w; vx[2] = f1.x; vy[2] = f1.y;
    }
}

// Per-triangle work of the setup stage: the three vertex_shader calls, /w, screen map,
// truncation, y-sort, T(); writes the 32-B GeomRec and returns the triangle's bbox
// clipped to the band, in pixels: x = x0 | x1 << 16, y = (y0 - row_begin) | (y1 - row_begin) << 16
// (RANGE_NONE_X when the bbox misses the band).
// (MT = the Metal rules, a compile-time flag in k_bin: as a run-time select hipcc evaluates round() for every vertex of
// every frame, 30 of the 385 vector instructions per triangle)
// (AFF = the transform's last row is (0, 0, 0, 1) — identity, orthographic, any affine map: w is then exactly 1 for every
// finite vertex (0*x + 0*y + 0*z + 1, :160) and x / 1 = x (:162), so the nine IEEE divisions per triangle are skipped; a
// non-finite vertex makes sx / sy non-finite either way and the triangle is skipped either way.  Chosen by the host, k_bin only.)
template <bool MT, bool AFF = false>
__device__ __forceinline__ uint2 setup_triangle_r(const SetupArgs& a, int64_t p, const float4& xa, const float4& xb, const float4& xc);
__device__ __forceinline__ uint2 setup_triangle(const SetupArgs& a, int64_t p, const float4& xa, const float4& xb, const float4& xc) {
    return a.metal ? setup_triangle_r<true>(a, p, xa, xb, xc) : setup_triangle_r<false>(a, p, xa, xb, xc);
}
__device__ __forceinline__ uint2 setup_triangle(const SetupArgs& a, int64_t p) {
    // :223-227 — the three vertex references of the primitive in slot p, in index order (de-indexed at upload)
    return setup_triangle(a, p, a.tri_xyz[3 * p + 0], a.tri_xyz[3 * p + 1], a.tri_xyz[3 * p + 2]);
}
// ... with the corners already loaded (k_setup_hist fetches those of its next group while it works on this one)
template <bool MT, bool AFF>
__device__ __forceinline__ uint2 setup_triangle_r(const SetupArgs& a, int64_t p, const float4& xa, const float4& xb, const float4& xc) {
    uint2 range = make_uint2(RANGE_NONE_X, 0u);
    const uint32_t orig = a.reordered ? __float_as_uint(xa.w) : 0u;
    // vertex colours are passed through by vertex_shader untouched (Shaders.metal:53) and are only
    // consumed by the resolve, which fetches them for the winning primitive through idx32 / rgb
    const float4 ca = make_float4(0, 0, 0, 0), cb = ca, cc = ca;

    const float fw = (float)a.tg.width, fh = (float)a.tg.height;
    float sx[3], sy[3], sz[3];
    {
        const float4 xs[3] = {xa, xb, xc};
        const float4 cs[3] = {ca, cb, cc};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            // Vertex.apply(transform:) (:159-163) through the vertex_shader hook
            VertexOut vo = vertex_shader(make_float3(xs[k].x, xs[k].y, xs[k].z),
                                         make_float3(cs[k].x, cs[k].y, cs[k].z), a.m);
            const float nx = AFF ? vo.pos.x : vo.pos.x / vo.pos.w;
            const float ny = AFF ? vo.pos.y : vo.pos.y / vo.pos.w;
            const float nz = AFF ? vo.pos.z : vo.pos.z / vo.pos.w;
            // convertedToScreen (:165-171)
            const float u = nx * 0.5f + 0.5f;
            const float v = ny * -0.5f + 0.5f;
            sx[k] = u * fw;
            sy[k] = v * fh;
            sz[k] = nz;
            if (MT) {                      // vertex_pass: pixels = round(uv * screen) (Shaders.metal:71)
                sx[k] = roundf(sx[k]);          // half away from zero, like Metal's round()
                sy[k] = roundf(sy[k]);
            }
        }
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; k++)
        ok = ok && (fabsf(sx[k]) < COORD_LIMIT) && (fabsf(sy[k]) < COORD_LIMIT);
    if (MT) {                              // uint2(pos.xy) (Shaders.metal:102-104): negative is undefined -> skip
#pragma unroll
        for (int k = 0; k < 3; k++) ok = ok && (sx[k] >= 0.0f) && (sy[k] >= 0.0f);
    }

    int ix[3] = {0, 0, 0}, iy[3] = {0, 0, 0};
    if (ok) {
#pragma unroll
        for (int k = 0; k < 3; k++) { ix[k] = (int)sx[k]; iy[k] = (int)sy[k]; }  // :251 truncation
    }
    // det == 0 (vertices collinear after truncation) is drawn like any other triangle under the CPU rules: T()
    // (:95-100) then holds +-inf / NaN, the clamp of :119-122 maps such colours to 0 / 1 and a NaN depth fails
    // the '<' of :258 — nothing traps in the reference.  (Under the Metal rules det is the shader's `divider`.)
    if (MT) {
        float t00, t01, t10, t11;
        const float det = tinv_of(ix[0], iy[0], ix[1], iy[1], ix[2], iy[2], t00, t01, t10, t11);
        ok = ok && (det != 0.0f) && (fabsf(det) < INFINITY);
    }

    // :271 stable 3-element insertion sort on FLOAT y
    int o0 = 0, o1 = 1, o2 = 2;
    if (sy[o1] < sy[o0]) { int t = o0; o0 = o1; o1 = t; }
    if (sy[o2] < sy[o1]) {
        int t = o1; o1 = o2; o2 = t;
        if (sy[o1] < sy[o0]) { t = o0; o0 = o1; o1 = t; }
    }
    int s0y = iy[o0], s2y = iy[o2];
    const int minx = min(ix[0], min(ix[1], ix[2])), maxx = max(ix[0], max(ix[1], ix[2]));
    if (MT) {
        // roi_pass (Shaders.metal:89-114): the bbox of the snapped vertices; the host skips ROIs whose
        // min-x or min-y is 0 (GpuRenderer.swift:122-124).  det == 0 here is the shader's `divider`
        // (same products): it would make every weight inf/NaN, i.e. no pixel inside -> skip as well.
        s0y = min(iy[0], min(iy[1], iy[2]));
        s2y = max(iy[0], max(iy[1], iy[2]));
        ok = ok && minx != 0 && s0y != 0;
    }
    // int16 vertex deltas and 32-bit span arithmetic are exact when the bbox extents are < 2^15
    const bool small = ((int64_t)maxx - (int64_t)minx < 32768) && ((int64_t)s2y - (int64_t)s0y < 32768);
    const uint32_t flags = (orig << GEOM_ORIG_SHIFT) | (ok ? GEOM_VALID : 0u) | (small ? GEOM_SMALL : 0u) |
                           ((uint32_t)o0 << GEOM_ORD_SHIFT) | ((uint32_t)o1 << (GEOM_ORD_SHIFT + 2)) |
                           ((uint32_t)o2 << (GEOM_ORD_SHIFT + 4));

    // bbox ∩ band -> tiles.  Every covered pixel lies in [minx,maxx] x [S0.y,S2.y] (spans are
    // integer interpolants between vertex x's, :467-494).
    const int x0 = max(minx, 0), x1 = min(maxx, a.tg.width - 1);
    const int y0 = max(s0y, a.tg.row_begin), y1 = min(s2y, a.tg.row_end - 1);
    const bool binned = ok && x0 <= x1 && y0 <= y1;
    // record stores: 2 x 16 B per lane (+ 2 for the rare non-small triangle) — only for triangles
    // that reach a bin: nobody ever gathers the record of a triangle that misses this GPU's band
    if (binned) {
        const uint32_t db = ((uint32_t)(ix[1] - ix[0]) & 0xFFFFu) | ((uint32_t)(iy[1] - iy[0]) << 16);
        const uint32_t dc = ((uint32_t)(ix[2] - ix[0]) & 0xFFFFu) | ((uint32_t)(iy[2] - iy[0]) << 16);
        int4* gp = reinterpret_cast<int4*>(a.geo + p);
        gp[0] = make_int4(ix[0], iy[0], (int)db, (int)dc);
        reinterpret_cast<float4*>(gp)[1] = make_float4(sz[0], sz[1], sz[2], __uint_as_float(flags));
        if (!small) {
            int4* fp = reinterpret_cast<int4*>(a.geo_full + p);
            fp[0] = make_int4(ix[0], iy[0], ix[1], iy[1]);
            fp[1] = make_int4(ix[2], iy[2], 0, 0);
        }
    }
    if (binned)
        range = make_uint2((uint32_t)x0 | ((uint32_t)x1 << 16),
                           (uint32_t)(y0 - a.tg.row_begin) | ((uint32_t)(y1 - a.tg.row_begin) << 16));
    return range;
}

// Bin entries carry a 6-bit size class above the primitive id (when ids fit 26 bits): the number
// of rows of the triangle inside the tile minus one (0..31), or CLASS_BIG when its clipped bbox
// area exceeds BIG_AREA.  k_raster counting-sorts its bin by this key so that the 64 triangles a
// wave walks together have the same height.
#ifndef SWR_BIG_AREA
#define SWR_BIG_AREA (1 << 20)
#endif
constexpr int BIG_AREA = SWR_BIG_AREA;
constexpr int LARGE_AREA = TILE_W * TILE_H / 2;   // clipped bbox area from which a triangle counts as large for its tile ...
constexpr int LARGE_MAX = 2;                      // ... and goes the cooperative way if the chunk has at most this many
constexpr uint32_t CLASS_SHIFT = 26;
constexpr uint32_t CLASS_BIG = 32;
constexpr int NUM_CLASSES = 33;
static_assert(TILE_H <= 32, "row-count classes assume TILE_H <= 32");

struct PixBox { int x0, x1, y0, y1; };   // y relative to the band
__device__ __forceinline__ PixBox unpack_box(uint2 r) {
    PixBox b; b.x0 = r.x & 0xFFFF; b.x1 = r.x >> 16; b.y0 = r.y & 0xFFFF; b.y1 = r.y >> 16; return b;
}
__device__ __forceinline__ uint32_t size_class(const PixBox& b, int tx, int ty) {
    const int rows = min(b.y1, ty * TILE_H + TILE_H - 1) - max(b.y0, ty * TILE_H) + 1;
    // (a box clipped to a tile never exceeds the default BIG_AREA: the column extent, the product and the compare are seven vector
    // instructions per (triangle, tile) pair of k_bin's second walk for a class nobody gets)
    if constexpr (BIG_AREA >= TILE_W * TILE_H) return (uint32_t)(rows - 1);
    const int cols = min(b.x1, tx * TILE_W + TILE_W - 1) - max(b.x0, tx * TILE_W) + 1;
    return rows * cols > BIG_AREA ? CLASS_BIG : (uint32_t)(rows - 1);
}

// Visit every tile of a triangle's tile rectangle.  Rectangles of up to COOP_TILES tiles are walked
// by the owning lane; larger ones (big triangles) are walked by the whole wave, one rectangle at a
// time, lanes striding over its tiles — a full-screen triangle would otherwise keep one lane busy
// for thousands of iterations.  Must be called by all 64 lanes of the wave (b.x0 > b.x1 = no tiles).
constexpr int COOP_TILES = 8;
struct NoBigTiles { __device__ bool operator()(const PixBox&, uint32_t, int, int) const { return false; } };
// big(box, p, owner lane, tiles): called once per triangle of more than COOP_TILES tiles, wave-uniformly; true = the caller has
// taken care of it (k_bin's deferred list), it is not walked
template <class F, class B = NoBigTiles>
__device__ __forceinline__ void for_each_tile(const PixBox& b, uint32_t p, F&& fn, B&& big = NoBigTiles{}) {
    const bool any = b.x0 <= b.x1;
    const int tx0 = b.x0 / TILE_W, ty0 = b.y0 / TILE_H;
    const int ntx = any ? b.x1 / TILE_W - tx0 + 1 : 0, nty = any ? b.y1 / TILE_H - ty0 + 1 : 0;
    const int n = ntx * nty;
    if (n > 0 && n <= COOP_TILES)
        for (int ty = ty0; ty < ty0 + nty; ty++)
            for (int tx = tx0; tx < tx0 + ntx; tx++) fn(b, p, tx, ty);
    unsigned long long mask = __ballot(n > COOP_TILES);
    const int lane = threadIdx.x & 63;
    while (mask) {
        const int src = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)mask) - 1);
        mask &= mask - 1;
        PixBox sb;
        sb.x0 = __builtin_amdgcn_readlane(b.x0, src); sb.x1 = __builtin_amdgcn_readlane(b.x1, src);
        sb.y0 = __builtin_amdgcn_readlane(b.y0, src); sb.y1 = __builtin_amdgcn_readlane(b.y1, src);
        const uint32_t sp = (uint32_t)__builtin_amdgcn_readlane((int)p, src);
        const int stx0 = sb.x0 / TILE_W, sty0 = sb.y0 / TILE_H;
        const int sntx = sb.x1 / TILE_W - stx0 + 1, sn = sntx * (sb.y1 / TILE_H - sty0 + 1);
        if (big(sb, sp, src, sn)) continue;
        // lane k walks tiles k, k + 64, ...: (column, row) advance by (64 % sntx, 64 / sntx) with one carry — a screen-filling
        // triangle is 64 steps of this loop in ONE wave, and the two divisions per step were most of each
        const int dq = 64 / sntx, dr = 64 % sntx;
        int cx = lane % sntx, cy = lane / sntx;
        for (int k = lane; k < sn; k += 64) {
            fn(sb, sp, stx0 + cx, sty0 + cy);
            cx += dr; cy += dq;
            if (cx >= sntx) { cx -= sntx; cy += 1; }
        }
    }
}

__device__ __forceinline__ int wave_incl_add(int v);

// Workgroup size of the two binning walks (template parameter BT of k_setup_hist / k_fill_lds): 256 threads.
// A 256-thread workgroup (44-52 VGPRs, one wave per SIMD, <= 16 KB of LDS) fits on a CU beside five resident
// raster workgroups, so the binning of frame N+1 really runs WHILE frame N is rasterised; a 1024-thread one
// (the former default, SWR_BIN_BT=1024) needs four free wave slots and 208 VGPRs per SIMD at once and only gets
// onto a CU in the raster's tail (cfg4: 132 -> 112 us per frame, tools/bt_g_sweep.sh).
// Tuning constants that used to be environment variables (rounds 1-2 swept them): compile-time now, so a stray variable
// cannot change how the product schedules.  A sweep builds side libraries: make ab NAME=g512 ABFLAGS=-DSWR_TUNE_BIN_G=512.
#ifndef SWR_TUNE_BIN_G
#define SWR_TUNE_BIN_G 256          // binning workgroups (256 threads each): one per CU
#endif
#ifndef SWR_TUNE_HIST16
#define SWR_TUNE_HIST16 1           // packed 16-bit LDS counters in k_setup_hist
#endif
#ifndef SWR_TUNE_VSPLIT
#define SWR_TUNE_VSPLIT (-1)        // log2 of the k_raster workgroups per tile; -1 = 4 per tile on grids of <= 320 tiles, else 1
#endif

// ---- binning, LDS path (default): no global atomics ------------------------------------------
// Can the 64 primitives of stream group `g` be skipped by this band?  True only when the projection of
// the group's object-space box provably misses [0,W) x [row_begin,row_end): all eight corners in front of
// the eye (w > 0, so the projected box is the hull of the projected corners), the hull at least `margin`
// pixels outside, where margin = 1 px (truncation / rounding of :251 / Shaders.metal:71 can move a vertex
// by less than one pixel) + a bound on the rounding error of the per-vertex transform (Σ|m_ij·c_j| terms at
// 2^-20 relative, propagated through the divide) + 2^-18 of the coordinate itself.  NaN anywhere (a
// non-finite vertex poisons its box) makes every comparison false: not culled.  Wave-uniform.
__device__ __forceinline__ bool group_culled(const SetupArgs& a, int64_t g) {
    const float4 lo = a.box64[2 * g], hi = a.box64[2 * g + 1];
    const float fw = (float)a.tg.width, fh = (float)a.tg.height;
    // error scale of one transformed coordinate: Σ_j |m_ij| * max|c_j| + |m_i3|
    const float ax = fmaxf(fabsf(lo.x), fabsf(hi.x)), ay = fmaxf(fabsf(lo.y), fabsf(hi.y)), az = fmaxf(fabsf(lo.z), fabsf(hi.z));
    const float4 c0 = a.m.columns[0], c1 = a.m.columns[1], c2 = a.m.columns[2], c3 = a.m.columns[3];
    const float d = 9.5367431640625e-07f;   // 2^-20
    const float ex = d * (fabsf(c0.x) * ax + fabsf(c1.x) * ay + fabsf(c2.x) * az + fabsf(c3.x));
    const float ey = d * (fabsf(c0.y) * ax + fabsf(c1.y) * ay + fabsf(c2.y) * az + fabsf(c3.y));
    const float ew = d * (fabsf(c0.w) * ax + fabsf(c1.w) * ay + fabsf(c2.w) * az + fabsf(c3.w));
    float wmin = INFINITY, nxlo = INFINITY, nxhi = -INFINITY, nylo = INFINITY, nyhi = -INFINITY;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float x = (k & 1) ? hi.x : lo.x, y = (k & 2) ? hi.y : lo.y, z = (k & 4) ? hi.z : lo.z;
        const float cx = c0.x * x + c1.x * y + c2.x * z + c3.x;
        const float cy = c0.y * x + c1.y * y + c2.y * z + c3.y;
        const float cw = c0.w * x + c1.w * y + c2.w * z + c3.w;
        ok = ok && (cw - 2.0f * ew > 0.0f) && fabsf(cx) < INFINITY && fabsf(cy) < INFINITY && cw < INFINITY;
        wmin = fminf(wmin, cw);
        const float nx = cx / cw, ny = cy / cw;
        nxlo = fminf(nxlo, nx); nxhi = fmaxf(nxhi, nx);
        nylo = fminf(nylo, ny); nyhi = fmaxf(nyhi, ny);
    }
    if (!ok) return false;
    const float wsafe = wmin - 2.0f * ew;                    // > 0
    const float nmax = fmaxf(fmaxf(fabsf(nxlo), fabsf(nxhi)), fmaxf(fabsf(nylo), fabsf(nyhi)));
    const float endc = (ex + ey + (1.0f + nmax) * ew) / wsafe + 4e-6f * (1.0f + nmax);   // |ndc error| bound
    // screen = (ndc * +-0.5 + 0.5) * size
    const float sx0 = (nxlo * 0.5f + 0.5f) * fw, sx1 = (nxhi * 0.5f + 0.5f) * fw;
    const float sy0 = (nyhi * -0.5f + 0.5f) * fh, sy1 = (nylo * -0.5f + 0.5f) * fh;
    const float smax = fmaxf(fmaxf(fabsf(sx0), fabsf(sx1)), fmaxf(fabsf(sy0), fabsf(sy1)));
    const float margin = 1.0f + endc * fmaxf(fw, fh) + smax * 3.8146972656e-06f + 0.25f;
    if (!(margin < INFINITY)) return false;
    return sx1 + margin < 0.0f || sx0 - margin >= fw || sy1 + margin < (float)a.tg.row_begin ||
           sy0 - margin >= (float)a.tg.row_end;
}

// Workgroup g owns the contiguous chunk [g*chunk, (g+1)*chunk) of the primitives in BOTH walks.
// k_setup_hist: per-workgroup tile histogram in LDS (ds_add), written as row g of the matrix
// M[G][tiles].  k_colscan turns every column into an exclusive prefix over g and emits the
// per-tile totals; k_scan scans the totals; k_fill_lds seeds its LDS cursors with
// tile_start[t] + M[g][t] and hands out bin positions with returning LDS atomics.
// Which stream groups (64 primitives, one wave's worth) does this band have to look at?  Workgroup w owns the
// groups w, w + G, w + 2G, ... — neighbouring groups (neighbours on screen, since the stream is Morton-ordered) go
// to different workgroups, so a band's few surviving groups spread over the whole GPU and one workgroup's LDS
// histogram is not hammered on one tile.  Phase 1: one lane per owned group projects the group's box
// (group_culled) and the survivors are compacted into an LDS list, which is also published to `live`
// ([G][1 + per]: count, group ids) for k_fill_lds to walk the same groups in the same order.  Phase 2: one wave
// per surviving group.  (The cull used to be a kernel of its own; a thin band's frame is bound by the host's
// launch rate, so it moved in here.)
// H16: two 16-bit counters per LDS word.  A workgroup's count for a tile is at most the number of primitives it owns
// (64 * per < 65536, checked by the launcher), so a half never carries into its neighbour.  A 4K histogram is then 8 KB
// instead of 16 KB and the workgroup fits into the LDS five resident k_raster workgroups leave free on a CU (12.5 KB)
// instead of displacing one of them for as long as it runs.
template <int BT, bool H16>
__global__ __launch_bounds__(BT) void k_setup_hist(SetupArgs a, uint32_t* __restrict__ M,
                                                     uint32_t* __restrict__ live, int per, int ntiles) {
    extern __shared__ uint32_t hist[];               // [ntiles] (or [ntiles/2]) histogram, [per] surviving groups, [1] their count
    const int hwords = H16 ? (ntiles + 1) >> 1 : ntiles;
    uint32_t* mylist = hist + hwords;                // (all dynamic: the kernel may ask for the whole 160 KB)
    uint32_t& nlive_s = mylist[per];
    if (threadIdx.x == 0) nlive_s = 0u;
    for (int e = threadIdx.x; e < hwords; e += blockDim.x) hist[e] = 0u;
    __syncthreads();
    {
        const int64_t groups = (a.ntri + 63) >> 6;
        const int lane = threadIdx.x & 63;
        for (int i0 = threadIdx.x & ~63; i0 < per; i0 += BT) {                 // wave-uniform trip count
            const int i = i0 + lane;
            const int64_t g = (int64_t)i * gridDim.x + blockIdx.x;
            const bool keep = i < per && g < groups && !(a.cull && group_culled(a, g));
            const unsigned long long mask = __ballot(keep);
            uint32_t base = 0u;
            if (lane == 0 && mask) base = atomicAdd(&nlive_s, (uint32_t)__popcll(mask));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (keep) mylist[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)g;
        }
    }
    __syncthreads();
    const uint32_t nlive = nlive_s;
    {
        uint32_t* out = live + (size_t)blockIdx.x * (size_t)(per + 1);
        if (threadIdx.x == 0) out[0] = nlive;
        for (uint32_t i = threadIdx.x; i < nlive; i += BT) out[1 + i] = mylist[i];
    }
    const int tiles_x = a.tg.tiles_x;
    // per wave; the 48 B per lane of the NEXT group are in flight while this one is transformed (the loop is a chain of
    // HBM round trips otherwise: four groups per wave, one wave per SIMD)
    float4 nxa = make_float4(0, 0, 0, 0), nxb = nxa, nxc = nxa;
    {
        const uint32_t j0 = threadIdx.x >> 6;
        const int64_t p0 = j0 < nlive ? ((int64_t)mylist[j0] << 6) + (threadIdx.x & 63) : a.ntri;
        if (p0 < a.ntri) { nxa = a.tri_xyz[3 * p0 + 0]; nxb = a.tri_xyz[3 * p0 + 1]; nxc = a.tri_xyz[3 * p0 + 2]; }
    }
    for (uint32_t j = threadIdx.x >> 6; j < nlive; j += BT / 64) {
        const int64_t p = ((int64_t)mylist[j] << 6) + (threadIdx.x & 63);
        const float4 xa = nxa, xb = nxb, xc = nxc;
        {
            const uint32_t jn = j + BT / 64;
            const int64_t pn = jn < nlive ? ((int64_t)mylist[jn] << 6) + (threadIdx.x & 63) : a.ntri;
            if (pn < a.ntri) { nxa = a.tri_xyz[3 * pn + 0]; nxb = a.tri_xyz[3 * pn + 1]; nxc = a.tri_xyz[3 * pn + 2]; }
        }
        uint2 r = make_uint2(RANGE_NONE_X, 0u);
        if (p < a.ntri) {
            r = setup_triangle(a, p, xa, xb, xc);
            a.ranges[p] = r;
        }
        // (one atomic per run of neighbouring lanes with the same tile instead of one per lane — the stream is
        // Morton-ordered, so these are 16- to 64-way same-address conflicts — was measured SLOWER: k_setup_hist 28.9 ->
        // 34.5 us, k_fill_lds 25.7 -> 33.8 us, profiles/r02/bin_aggregate_ab.txt: the LDS resolves same-address atomics
        // faster than the wave can find its runs)
        for_each_tile(unpack_box(r), (uint32_t)p, [&](const PixBox&, uint32_t, int tx, int ty) {
            const int tile = ty * tiles_x + tx;
            if (H16) atomicAdd(&hist[tile >> 1], 1u << ((tile & 1) << 4));
            else atomicAdd(&hist[tile], 1u);
        });
    }
    __syncthreads();
    uint32_t* row = M + (size_t)blockIdx.x * (size_t)ntiles;
    for (int e = threadIdx.x; e < ntiles; e += blockDim.x)
        row[e] = H16 ? (hist[e >> 1] >> ((e & 1) << 4)) & 0xFFFFu : hist[e];
}

// 256 threads = 16 tiles x 16 segments of the workgroup axis; a segment is ceil(G/16) rows, walked in
// register blocks of COLSEG rows (sum pass, then prefix pass over the same L2-resident columns).
constexpr int COLSEG = 32;
constexpr int MAX_BIN_G = 1024;
__global__ __launch_bounds__(256) void k_colscan(uint32_t* __restrict__ M, int G, int ntiles,
                                                 uint32_t* __restrict__ tile_count) {
    __shared__ uint32_t seg_sum[16][17];
    const int el = threadIdx.x & 15, seg = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const int seglen = (G + 15) / 16;
    const int g0 = seg * seglen, g1 = min(g0 + seglen, G);
    uint32_t sum = 0;
    if (e < ntiles)
        for (int gb = g0; gb < g1; gb += COLSEG) {
            uint32_t v[COLSEG];
#pragma unroll
            for (int k = 0; k < COLSEG; k++) v[k] = gb + k < g1 ? M[(size_t)(gb + k) * ntiles + e] : 0u;
#pragma unroll
            for (int k = 0; k < COLSEG; k++) sum += v[k];
        }
    seg_sum[el][seg] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int k = 0; k < seg; k++) run += seg_sum[el][k];
    if (e < ntiles)
        for (int gb = g0; gb < g1; gb += COLSEG) {
            uint32_t v[COLSEG];
#pragma unroll
            for (int k = 0; k < COLSEG; k++) v[k] = gb + k < g1 ? M[(size_t)(gb + k) * ntiles + e] : 0u;
#pragma unroll
            for (int k = 0; k < COLSEG; k++) {
                if (gb + k < g1) M[(size_t)(gb + k) * ntiles + e] = run;
                run += v[k];
            }
        }
    if (seg == 15 && e < ntiles) tile_count[e] = run;
}

template <int BT>
__global__ __launch_bounds__(BT) void k_fill_lds(const uint2* __restrict__ ranges, int64_t ntri,
                                                   const uint32_t* __restrict__ M,
                                                   const uint32_t* __restrict__ tile_count,
                                                   uint32_t* __restrict__ tile_start,
                                                   uint32_t* __restrict__ counters,
                                                   uint32_t* __restrict__ host_counters,
                                                   uint32_t* __restrict__ bins, uint32_t capacity,
                                                   const uint32_t* __restrict__ live, int gper,
                                                   int ntiles, int tiles_x, int tag_class, uint32_t* __restrict__ host_max) {
    extern __shared__ uint32_t lds[];
    uint32_t* cursor = lds;             // [ntiles]
    uint32_t* part = lds + ntiles;      // [BT]
    const int t = threadIdx.x;
    // Every workgroup scans the per-tile totals itself (16 KB from L2, ~1 us) instead of waiting
    // for a single-workgroup scan kernel; workgroup 0 publishes tile_start and the pair total.
    for (int e = t; e < ntiles; e += BT) cursor[e] = tile_count[e];
    __syncthreads();
    const int per = (ntiles + BT - 1) / BT;
    const int sb = t * per, se = min(sb + per, ntiles);
    uint32_t sum = 0, mx = 0;
    for (int i = sb; i < se; i++) { sum += cursor[i]; mx = max(mx, cursor[i]); }
    if (blockIdx.x == 0 && t == 0) part[BT / 64 + 1] = 0u;
    // exclusive prefix of the per-thread sums: a DPP scan inside every wave, then wave 0 scans the wave totals
    // (two barriers; the Hillis-Steele ladder over LDS this replaces took 2 log2(BT) = 20)
    static_assert(BT % 64 == 0 && BT / 64 <= 64, "one wave scans the wave totals");
    const uint32_t incl = (uint32_t)wave_incl_add((int)sum);
    if ((t & 63) == 63) part[t >> 6] = incl;
    __syncthreads();
    if (t < 64) {
        const uint32_t v = t < BT / 64 ? part[t] : 0u;
        const uint32_t wi = (uint32_t)wave_incl_add((int)v);
        if (t < BT / 64) part[t] = wi - v;
        if (t == BT / 64 - 1) part[BT / 64] = wi;
    }
    __syncthreads();
    const uint32_t total = part[BT / 64];
    uint32_t run = part[t >> 6] + incl - sum;
    for (int i = sb; i < se; i++) { const uint32_t c = cursor[i]; cursor[i] = run; run += c; }
    if (blockIdx.x == 0 && mx) atomicMax(&part[BT / 64 + 1], mx);      // the fullest bin of the frame (for the host: see launch_sort_bins)
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int e = t; e < ntiles; e += BT) tile_start[e] = cursor[e];
        if (t == 0) {
            if (host_max) *host_max = part[BT / 64 + 1];
            tile_start[ntiles] = total;
            counters[CNT_PAIRS] = total;                  // read by k_sort_bins / k_raster
            counters[CNT_OVERFLOW] = total > capacity ? 1u : 0u;
            host_counters[CNT_PAIRS] = total;             // pinned host word: no D2H copy per frame
        }
    }
    if (total > capacity) return;   // overflow: the host grows the bins and redraws
    const uint32_t* row = M + (size_t)blockIdx.x * (size_t)ntiles;
    for (int e = t; e < ntiles; e += BT) cursor[e] += row[e];
    __syncthreads();
    const uint32_t* mylist = live + (size_t)blockIdx.x * (size_t)(gper + 1);   // published by k_setup_hist
    const uint32_t nlive = mylist[0];
    // same groups, same order as k_setup_hist; the boxes of four groups are fetched together (the walk is otherwise one
    // HBM round trip per group, four or more per wave)
    constexpr int FB = 4;
    for (uint32_t j0 = t >> 6; j0 < nlive; j0 += FB * (BT / 64)) {
        int64_t pp[FB];
        uint2 rr[FB];
#pragma unroll
        for (int k = 0; k < FB; k++) {
            const uint32_t j = j0 + (uint32_t)k * (BT / 64);
            pp[k] = j < nlive ? ((int64_t)mylist[1 + j] << 6) + (t & 63) : ntri;
            rr[k] = pp[k] < ntri ? ranges[pp[k]] : make_uint2(RANGE_NONE_X, 0u);
        }
#pragma unroll
        for (int k = 0; k < FB; k++) {
            if (j0 + (uint32_t)k * (BT / 64) >= nlive) break;            // wave-uniform
            for_each_tile(unpack_box(rr[k]), (uint32_t)pp[k], [&](const PixBox& b, uint32_t prim, int tx, int ty) {
                const uint32_t pos = atomicAdd(&cursor[ty * tiles_x + tx], 1u);
                bins[pos] = prim | (tag_class ? size_class(b, tx, ty) << CLASS_SHIFT : 0u);
            });
        }
    }
}

// ---- binning in ONE kernel, fixed-stride bins (default) --------------------------------------------------------
// The bin of tile t is the fixed region bins[t * cap, (t + 1) * cap): there is no prefix over tiles, so nothing in the
// frame depends on the totals of ALL binning workgroups and the whole chain k_setup_hist -> k_colscan -> k_fill_lds
// collapses into one launch:
//   1  cull: one lane per owned 64-primitive group (as k_setup_hist)
//   2  per triangle: 3 x vertex_shader, /w, screen map, truncation, y-sort -> GeomRec, ranges; the workgroup's tile
//      histogram in LDS, two 16-bit counters per word
//   3  one RETURNING global atomic per (workgroup, tile it touches) reserves the workgroup's run inside the tile's
//      region: base = atomicAdd(fill[t], count).  The stream is Morton-ordered and a workgroup owns every G-th group,
//      so it touches a few hundred tiles, not all of them (cfg4: ~220 K atomics per frame, 16 or more in flight per
//      lane).  The LDS word now holds the two 16-bit cursors.
//   4  second walk over the same groups (ranges are L2-hot): bins[t * cap + ds_add_rtn(cursor[t])] = slot | class.
// fill[] must be zero on entry: the kernel zeroes the NEXT frame's array (four arrays rotate while the working sets
// rotate by three, so the array zeroed here was last read by a raster that finished before this kernel could start).
// A tile that receives more than cap entries overflows: the frame's largest fill goes to the host, which grows cap and
// redraws (or falls back to the exact-size path above when a tile needs more than FIXED_CAP_MAX entries).
struct BinArgs {
    SetupArgs a;
    uint32_t* fill;         // [CNT_WORDS counters][ntiles] of this frame (zero on entry)
    uint32_t* fill_next;    // the same block of the next frame: zeroed here
    uint32_t* bins;         // [ntiles * cap]
    uint32_t cap;           // entries per tile, <= FIXED_CAP_MAX
    int per;                // stream groups owned by one workgroup
    int ntiles;
    int tag_class;
    uint4* biglist;         // [BIGLIST_CAP] (slot, ranges.x, ranges.y, -) of the frame's deferred triangles
    int defer_ok;           // this frame's k_sort_bins will run
};
constexpr uint32_t FIXED_CAP_MAX = 61440u;   // cursor halves stay below 2^16: cap + primitives owned by one workgroup < 65536
enum { CNT_MAXFILL = 3, CNT_BIGLIST = 4, CNT_BIGSEEN = 5 };
// Triangles that cover more than BIN_BIG_TILES tiles (a wall, a ground plane, an occluder: a screen-filling one is 4 080 tiles
// at 4K) are not scattered into those tiles' bins by the wave that set them up — one 4-byte store per tile, each to another
// cache line, from ONE compute unit: 300 screen-filling triangles at 1080p took k_bin 145 us that way — but put on a short list
// (stream slot + box).  k_sort_bins, one workgroup per tile, appends the listed triangles that touch ITS tile to its own bin before
// it sorts it: 1 020 workgroups write 113 entries each instead of five waves writing 23 000.  Only frames whose k_sort_bins runs
// may defer (BinArgs::defer_ok); a full list, or a frame that skips the sort, bins them the old way.
constexpr int BIN_BIG_TILES = 128;
constexpr uint32_t BIGLIST_CAP = 1024u;
__device__ __forceinline__ int tiles_of_box(const PixBox& b) {
    return b.x0 <= b.x1 ? (b.x1 / TILE_W - b.x0 / TILE_W + 1) * (b.y1 / TILE_H - b.y0 / TILE_H + 1) : 0;
}

template <int BT, bool MT, bool DEFER, bool AFF = false>
// Register budget of the plain kernel: 56 VGPRs (tools/vgprs.sh) — with 58 the pipelined cfg4 frame measured 4 % slower (one of its waves has
// to fit beside five raster waves of 88); neither launch bounds nor amdgpu_waves_per_eu make this hipcc keep it, the source does.
__global__ __launch_bounds__(BT) void k_bin(BinArgs b) {
    extern __shared__ uint32_t hist[];               // [(ntiles + 1) / 2] counters -> cursors, [per] surviving groups, [1] their count, [2 * BT / 64] reduction, [1] deferred pairs
    const SetupArgs& a = b.a;
    const int ntiles = b.ntiles, per = b.per;
    const int hwords = (ntiles + 1) >> 1;
    uint32_t* mylist = hist + hwords;
    uint32_t& nlive_s = mylist[per];
    uint32_t* red = mylist + per + 1;
    uint32_t& bigp_s = red[2 * (BT / 64)];           // (triangle, tile) pairs this workgroup handed to the deferred list
    const int t = threadIdx.x, lane = t & 63;
    if (t == 0) { nlive_s = 0u; bigp_s = 0u; }
    for (int e = t; e < hwords; e += BT) hist[e] = 0u;
    {   // the next frame's counters and fill words
        uint32_t* fn = b.fill_next;
        for (int e = blockIdx.x * BT + t; e < CNT_WORDS + ntiles; e += gridDim.x * BT) fn[e] = 0u;
    }
    __syncthreads();
    {
        const int64_t groups = (a.ntri + 63) >> 6;
        for (int i0 = t & ~63; i0 < per; i0 += BT) {                 // wave-uniform trip count
            const int i = i0 + lane;
            const int64_t g = (int64_t)i * gridDim.x + blockIdx.x;
            const bool keep = i < per && g < groups && !(a.cull && group_culled(a, g));
            const unsigned long long mask = __ballot(keep);
            uint32_t base = 0u;
            if (lane == 0 && mask) base = atomicAdd(&nlive_s, (uint32_t)__popcll(mask));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (keep) mylist[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)g;
        }
    }
    __syncthreads();
    const uint32_t nlive = nlive_s;
    const int tiles_x = a.tg.tiles_x;
    // ---- 2: setup + histogram (the 48 B per lane of the NEXT group in flight while this one is transformed)
    {
        float4 nxa = make_float4(0, 0, 0, 0), nxb = nxa, nxc = nxa;
        {
            const uint32_t j0 = t >> 6;
            const int64_t p0 = j0 < nlive ? ((int64_t)mylist[j0] << 6) + lane : a.ntri;
            if (p0 < a.ntri) { nxa = a.tri_xyz[3 * p0 + 0]; nxb = a.tri_xyz[3 * p0 + 1]; nxc = a.tri_xyz[3 * p0 + 2]; }
        }
        for (uint32_t j = t >> 6; j < nlive; j += BT / 64) {
            const int64_t p = ((int64_t)mylist[j] << 6) + lane;
            const float4 xa = nxa, xb = nxb, xc = nxc;
            {
                const uint32_t jn = j + BT / 64;
                const int64_t pn = jn < nlive ? ((int64_t)mylist[jn] << 6) + lane : a.ntri;
                if (pn < a.ntri) { nxa = a.tri_xyz[3 * pn + 0]; nxb = a.tri_xyz[3 * pn + 1]; nxc = a.tri_xyz[3 * pn + 2]; }
            }
            uint2 r = make_uint2(RANGE_NONE_X, 0u);
            if (p < a.ntri) {
                r = setup_triangle_r<MT, AFF>(a, p, xa, xb, xc);
                if (DEFER && tiles_of_box(unpack_box(r)) > BIN_BIG_TILES) {
                    const uint32_t e = atomicAdd(&b.fill[CNT_BIGLIST], 1u);
                    if (e < BIGLIST_CAP) {                           // (a full list: walked like any other)
                        b.biglist[e] = make_uint4((uint32_t)p, r.x, r.y, 0u);
                        atomicAdd(&bigp_s, (uint32_t)tiles_of_box(unpack_box(r)));   // k_sort_bins appends exactly these pairs
                        r = make_uint2(RANGE_NONE_X, 0u);            // neither walk of this kernel sees it
                    }
                }
                a.ranges[p] = r;
            }
            for_each_tile(unpack_box(r), (uint32_t)p, [&](const PixBox&, uint32_t, int tx, int ty) {
                const int tile = ty * tiles_x + tx;
                atomicAdd(&hist[tile >> 1], 1u << ((tile & 1) << 4));
            });
        }
    }
    __syncthreads();
    // ---- 3: reserve this workgroup's run in every tile region it touches
    uint32_t psum = t == 0 ? bigp_s : 0u, pmax = 0u;
    {
        uint32_t* fill = b.fill + CNT_WORDS;
        constexpr int U = 4;
        for (int w0 = t; w0 < hwords; w0 += BT * U) {
            uint32_t cw[U], b0[U], b1[U];
#pragma unroll
            for (int k = 0; k < U; k++) {
                const int w = w0 + k * BT;
                cw[k] = w < hwords ? hist[w] : 0u;
                b0[k] = b1[k] = 0u;
            }
#pragma unroll
            for (int k = 0; k < U; k++) {             // up to 2 U independent returning atomics in flight per lane
                const int w = w0 + k * BT;
                const uint32_t c0 = cw[k] & 0xFFFFu, c1 = cw[k] >> 16;
                if (c0) b0[k] = atomicAdd(&fill[2 * w], c0);
                if (c1) b1[k] = atomicAdd(&fill[2 * w + 1], c1);
            }
#pragma unroll
            for (int k = 0; k < U; k++) {
                const int w = w0 + k * BT;
                const uint32_t c0 = cw[k] & 0xFFFFu, c1 = cw[k] >> 16;
                psum += c0 + c1;
                if (c0) pmax = max(pmax, b0[k] + c0);
                if (c1) pmax = max(pmax, b1[k] + c1);
                if (w < hwords) hist[w] = min(b0[k], b.cap) | (min(b1[k], b.cap) << 16);
            }
        }
    }
    {
        // totals of the frame: workgroup sums -> two device counters; the last workgroup to arrive hands them to the host
        const uint32_t ws = (uint32_t)wave_incl_add((int)psum);
        uint32_t wm = pmax;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wm = max(wm, (uint32_t)__shfl_xor((int)wm, off));
        if (lane == 63) red[t >> 6] = ws;
        if (lane == 0) red[BT / 64 + (t >> 6)] = wm;
    }
    __syncthreads();     // also orders the cursor words written above before the second walk
    if (t == 0) {
        // fire and forget: k_raster of this frame (which starts when this kernel has finished) hands the two totals to
        // the host.  (A ticket + last-workgroup-publishes protocol was built first: five dependent round trips to memory
        // in one lane put 8 us on the tail of the kernel: a thin band's k_bin 27 us against 20.)
        uint32_t s = 0u, m = 0u;
        for (int k = 0; k < BT / 64; k++) { s += red[k]; m = max(m, red[BT / 64 + k]); }
        uint32_t* cnt = b.fill;
        if (s) atomicAdd(&cnt[CNT_PAIRS], s);
        if (m) atomicMax(&cnt[CNT_MAXFILL], m);
    }
    // ---- 4: second walk, same groups, same order
    constexpr int FB = 4;
    for (uint32_t j0 = t >> 6; j0 < nlive; j0 += FB * (BT / 64)) {
        int64_t pp[FB];
        uint2 rr[FB];
#pragma unroll
        for (int k = 0; k < FB; k++) {
            const uint32_t j = j0 + (uint32_t)k * (BT / 64);
            pp[k] = j < nlive ? ((int64_t)mylist[j] << 6) + lane : a.ntri;
            rr[k] = pp[k] < a.ntri ? a.ranges[pp[k]] : make_uint2(RANGE_NONE_X, 0u);
        }
#pragma unroll
        for (int k = 0; k < FB; k++) {
            if (j0 + (uint32_t)k * (BT / 64) >= nlive) break;            // wave-uniform
            for_each_tile(unpack_box(rr[k]), (uint32_t)pp[k], [&](const PixBox& bx, uint32_t prim, int tx, int ty) {
                const int tile = ty * tiles_x + tx;
                const uint32_t sh = (uint32_t)(tile & 1) << 4;
                const uint32_t pos = (atomicAdd(&hist[tile >> 1], 1u << sh) >> sh) & 0xFFFFu;
                if (pos < b.cap)
                    b.bins[(size_t)tile * b.cap + pos] = prim | (b.tag_class ? size_class(bx, tx, ty) << CLASS_SHIFT : 0u);
            }, [&](const PixBox&, uint32_t, int, int sn) {
                // a triangle for the deferred list, binned the plain way: tells the host to take the deferring kernel next time
                // (here and not in the first walk, whose register budget it would break: 56, see above)
                if (!DEFER && sn > BIN_BIG_TILES && lane == 0) atomicOr(&b.fill[CNT_BIGSEEN], 1u);
                return false;
            });
        }
    }
}

// ---- binning, global-atomic path (fallback when the tile table does not fit LDS) --------------
__global__ __launch_bounds__(256) void k_setup_bin(SetupArgs a) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.ntri) return;
    const uint2 r = setup_triangle(a, p);
    a.ranges[p] = r;
    const PixBox b = unpack_box(r);
    if (b.x0 > b.x1) return;
    // count pass: fire-and-forget atomics (no return value -> no round trip)
    for (int ty = b.y0 / TILE_H; ty <= b.y1 / TILE_H; ty++)
        for (int tx = b.x0 / TILE_W; tx <= b.x1 / TILE_W; tx++)
            atomicAdd(&a.tile_count[ty * a.tg.tiles_x + tx], 1u);
}

// Second walk over the triangles' tile rectangles; a returning atomic on the tile's cursor
// (initialised to tile_start by k_scan) hands out the bin position.  Visibility keys make the
// raster order-independent, so bins need no particular order.
__global__ __launch_bounds__(256) void k_fill(const uint2* __restrict__ ranges, int64_t ntri,
                                              uint32_t* __restrict__ cursor,
                                              const uint32_t* __restrict__ counters,
                                              uint32_t* __restrict__ bins, uint32_t capacity,
                                              int tiles_x, int tag_class) {
    if (counters[CNT_PAIRS] > capacity) return;   // overflow: the host grows the bins and redraws
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= ntri) return;
    const PixBox b = unpack_box(ranges[p]);
    if (b.x0 > b.x1) return;
    for (int ty = b.y0 / TILE_H; ty <= b.y1 / TILE_H; ty++)
        for (int tx = b.x0 / TILE_W; tx <= b.x1 / TILE_W; tx++) {
            const uint32_t pos = atomicAdd(&cursor[ty * tiles_x + tx], 1u);
            bins[pos] = (uint32_t)p | (tag_class ? size_class(b, tx, ty) << CLASS_SHIFT : 0u);
        }
}

// ------------------------------------------------------------------------------------------
// k_scan: exclusive scan of tile_count -> tile_start[0..tiles], one workgroup
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan(const uint32_t* __restrict__ count,
                                               uint32_t* __restrict__ start,
                                               uint32_t* __restrict__ cursor, int n,
                                               uint32_t* counters, uint32_t* host_counters,
                                               uint32_t capacity) {
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b = t * per, e = min(b + per, n);
    uint32_t s = 0;
    for (int i = b; i < e; i++) s += count[i];
    part[t] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = (t >= off) ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - s;
    for (int i = b; i < e; i++) { start[i] = run; cursor[i] = run; run += count[i]; }
    if (t == 1023) {
        start[n] = part[1023];
        counters[CNT_PAIRS] = part[1023];                  // total (triangle,tile) pairs of the frame
        host_counters[CNT_PAIRS] = part[1023];             // pinned host word: no D2H copy per frame
        if (part[1023] > capacity) counters[CNT_OVERFLOW] = 1u;
    }
}

// ------------------------------------------------------------------------------------------
// k_sort_bins: per tile, counting sort of the bin by the 6-bit size class k_fill attached to
// every entry (heaviest class first), in place, class bits stripped.  After it the 64 triangles
// a raster wave walks together have the same number of rows inside the tile.  Bins longer than
// SORT_CAP are sorted in independent segments (still correct: order never affects the image).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_add(int v);
constexpr int SORT_THREADS = 256;
constexpr int SORT_CAP = 4 * SORT_THREADS;
__global__ __launch_bounds__(SORT_THREADS) void k_sort_bins(uint32_t* __restrict__ bins,
                                                            const uint32_t* __restrict__ tile_start,
                                                            const uint32_t* __restrict__ counters,
                                                            uint32_t capacity, int tag_class,
                                                            uint32_t* __restrict__ fill, uint32_t fixed_cap,
                                                            const uint4* __restrict__ biglist, int tiles_x) {
    __shared__ uint32_t cls_cnt[64];
    __shared__ uint32_t app_count;
    const int tid = threadIdx.x;
    uint32_t b0, b1;
    if (fixed_cap) {        // fixed-stride bins (k_bin): tile t owns [t * cap, t * cap + fill[t])
        b0 = blockIdx.x * fixed_cap;
        uint32_t count = fill[CNT_WORDS + blockIdx.x];
        // the frame's deferred triangles (k_bin, BIN_BIG_TILES) that touch this tile join its bin here
        const uint32_t nbig = biglist ? min(fill[CNT_BIGLIST], BIGLIST_CAP) : 0u;     // workgroup-uniform
        if (nbig) {
            if (tid == 0) app_count = count;
            __syncthreads();
            const int tx = (int)blockIdx.x % tiles_x, ty = (int)blockIdx.x / tiles_x;
            for (uint32_t base = 0; base < nbig; base += SORT_THREADS) {
                const uint32_t e = base + (uint32_t)tid;
                uint4 ent = make_uint4(0u, RANGE_NONE_X, 0u, 0u);
                if (e < nbig) ent = biglist[e];
                const PixBox bx = unpack_box(make_uint2(ent.y, ent.z));
                const bool hit = bx.x0 <= bx.x1 && tx >= bx.x0 / TILE_W && tx <= bx.x1 / TILE_W &&
                                 ty >= bx.y0 / TILE_H && ty <= bx.y1 / TILE_H;
                const unsigned long long mask = __ballot(hit);
                uint32_t wbase = 0u;
                if ((tid & 63) == 0 && mask) wbase = atomicAdd(&app_count, (uint32_t)__popcll(mask));
                wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
                if (hit) {
                    const uint32_t pos = wbase + (uint32_t)__popcll(mask & ((1ull << (tid & 63)) - 1ull));
                    if (pos < fixed_cap) bins[b0 + pos] = ent.x | (tag_class ? size_class(bx, tx, ty) << CLASS_SHIFT : 0u);
                }
            }
            __syncthreads();
            const uint32_t grown = app_count;
            // (no frame-wide counter is touched here — thousands of workgroups on one word would serialise in the L2: k_bin
            // has counted the deferred pairs — except by a tile that the appended triangles really push over its region: that
            // rare tile raises the frame's largest fill, which is what k_raster and the host test.  Round 3 bounded the fullest
            // bin by k_bin's maximum + the length of the list: a frame whose deferred triangles miss its fullest tile was then
            // declared overflowed, rastered empty and redrawn for nothing)
            if (tid == 0 && grown != count) fill[CNT_WORDS + blockIdx.x] = grown;
            if (tid == 0 && grown > fixed_cap) atomicMax(&fill[CNT_MAXFILL], grown);
            count = grown;
        }
        b1 = b0 + min(count, fixed_cap);
    } else {
        if (counters[CNT_PAIRS] > capacity) return;
        b0 = tile_start[blockIdx.x]; b1 = tile_start[blockIdx.x + 1];
    }
    for (uint32_t seg = b0; seg < b1; seg += SORT_CAP) {
        const uint32_t m = min((uint32_t)SORT_CAP, b1 - seg);
        if (tid < 64) cls_cnt[tid] = 0u;
        __syncthreads();
        uint32_t ent[SORT_CAP / SORT_THREADS], pos[SORT_CAP / SORT_THREADS];
#pragma unroll
        for (int k = 0; k < SORT_CAP / SORT_THREADS; k++) {
            const uint32_t i = tid + k * SORT_THREADS;
            ent[k] = 0u; pos[k] = 0u;
            if (i < m) {
                ent[k] = bins[seg + i];
                pos[k] = atomicAdd(&cls_cnt[tag_class ? ent[k] >> CLASS_SHIFT : 0u], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {   // wave 0: exclusive prefix over the classes in descending order
            const int c = NUM_CLASSES - 1 - tid;
            const uint32_t v = c >= 0 ? cls_cnt[c] : 0u;
            const uint32_t incl = (uint32_t)wave_incl_add((int)v);
            if (c >= 0) cls_cnt[c] = incl - v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SORT_CAP / SORT_THREADS; k++) {
            const uint32_t i = tid + k * SORT_THREADS;
            if (i < m) {
                const uint32_t c = tag_class ? ent[k] >> CLASS_SHIFT : 0u;
                bins[seg + cls_cnt[c] + pos[k]] = tag_class ? ent[k] & ((1u << CLASS_SHIFT) - 1u) : ent[k];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// k_raster
// ------------------------------------------------------------------------------------------
struct RasterArgs {
    const GeomRec* geo;
    const GeomFull* geo_full;
    const uint32_t* inv;        // [ntri] original primitive index -> stream slot (resolve; only when reordered)
    int reordered;              // bin entries are stream slots; keys carry the original index from GeomRec.flags
    const float4* tri_rgb;      // [3*ntri] vertex colours per slot corner, de-indexed at upload (r,g,b,v)
    const float4* tri_nrm;      // [3*ntri] (nx,ny,nz,u) — extended fragment stage only
    FragmentUniforms fs;        // material + texture of the extended fragment stage
    const uint32_t* tile_start;
    const uint32_t* bins;
    const uint32_t* counters;
    uint32_t capacity;
    uint8_t* color;     // band-local BGRA8
    float* depth;       // band-local f32
    Target tg;
    int tag_class;      // bin entries carry a size class above bit CLASS_SHIFT
    const uint32_t* fill;   // fixed-stride bins (k_bin): [CNT_WORDS counters][ntiles fills]; NULL = exact bins (tile_start)
    uint32_t fixed_cap;     // entries per tile region (0 = exact bins)
    uint32_t* host_pairs;   // fixed-stride bins: pinned host words for the frame's pair total, its largest fill (the overflow
    uint32_t* host_fill;    // test) and the same for the host's sort heuristic — k_bin leaves them in device counters
    uint32_t* host_max;
    int insort;         // 32-bit keys: the bins are unsorted and tagged (no k_sort_bins ran): the workgroup sorts its bin in LDS
    int pack_local;     // the scene has at most 2^WTAB_PRIM_BITS primitives: colour frames may carry the bin position in the key's low word (winner table)
    int vs_log;         // 2^vs_log workgroups per tile, each owning TILE_H >> vs_log of its rows (small grids, see launch_raster)
    uint32_t* redo_dev; // k_raster_depth: tiles (one in REDO_SAMPLE) that had to be rastered again with 64-bit keys, since the last launch
    uint32_t* host_redo;    // ... handed to the host's pinned word by the next launch (the host then switches the scene to the 64-bit kernel)
};
constexpr int REDO_SAMPLE = 8;

// ---- LDS of one raster workgroup -------------------------------------------------------------------------------
// 64-bit visibility keys (orderable depth << 32 | primitive, or ~primitive without z-test): every frame that needs the
// winner's identity — colour frames, painter's order, the Metal rules.
constexpr int RASTER_QCAP = 128;         // ring entries per wave (see raster_tile)
// Winner table of colour frames (raster_tile, resolve): the key's low word is  original primitive << WTAB_LOCAL_BITS | position in
// the tile's bin, so the resolve knows WHICH bin entries own pixels without a search: at most 2^20 primitives, bins of at most 4096 entries.
constexpr int WTAB_LOCAL_BITS = 12, WTAB_PRIM_BITS = 32 - WTAB_LOCAL_BITS;
constexpr int WTAB_WORDS = (1 << WTAB_LOCAL_BITS) / 32;
constexpr int RASTER_SORT_SEG = 1024;    // bin entries a raster workgroup can sort in its own LDS (fuller bins stay unsorted)
struct alignas(16) RasterLds64 {
    unsigned long long keys[TILE_W * TILE_H];
    float4 tabAB[2 * RASTER_THREADS];    // per triangle of the batch: (t00, t01, t10, t11) | (za, zb, zc, (C.x - X0) | (C.y - Y0) << 16)
    uint32_t tabP[RASTER_THREADS];       //                            original primitive index (the key's low word)
    uint32_t queue[RASTER_THREADS / 64][RASTER_QCAP];
    uint32_t next_chunk;                 // work-stealing cursor over the chunks of the sorted bin
    uint32_t pad_[3];
    uint2 winners[WTAB_WORDS];           // colour frames (winner table, raster_tile): x = which bin positions own a pixel (32 per word), y = how many before the word
};
// 32-bit keys: depth-only z-tested frames under the CPU rules (BASELINE config 4).  The image of such a frame is the
// per-pixel minimum of the fragment depths (Renderer.swift:257-261); WHICH primitive wins a tie cannot be seen in it, except
// through the sign of a zero (and never through anything else: equal non-zero floats have equal bits).  So the key is the
// depth itself and the atomic an LDS float minimum — no orderable map, no index word, half the LDS traffic — and a tile
// whose result holds a zero (below) is rastered again with the 64-bit keys.
struct alignas(16) RasterLds32 {
    float keys[TILE_W * TILE_H + 16];    // (+ padding: a 4-pixel visit that starts at the tile's last pixel may address three floats past it)
    float4 tabAB[2 * RASTER_THREADS];
    uint32_t tabP[4];                    // (unused: no index word)
    uint32_t queue[RASTER_THREADS / 64][RASTER_QCAP];
    uint32_t next_chunk;
    uint32_t redo;                       // a thread of the resolve met a key the 64-bit path has to decide
    // Part of what the narrow keys leave free holds the tile's bin, counting-sorted by size class by the workgroup itself
    // (raster_tile, insort — small grids only): such frames need no k_sort_bins launch in front of the raster.
    uint32_t cls_cnt[64];
    uint32_t sorted[RASTER_SORT_SEG];
};
static_assert(sizeof(RasterLds32) <= sizeof(RasterLds64), "the 32-bit path must not need more LDS than the 64-bit one (5 workgroups per CU)");
// The atomic is ds_min_f32.  What the LDS does with the special values was measured on the device
// (tools/micro/ds_min_f32_semantics.hip, profiles/r04/ds_min_f32_semantics.txt): IEEE minNum — a quiet NaN operand leaves the
// memory untouched (a NaN depth never passes '<', :258: exactly that), denormals compare exactly, -inf / +inf order as
// numbers, and -0 counts as smaller than +0 whatever the order of arrival.  The last point is the one thing the 32-bit key
// cannot get right — under '<' the two zeros are equal and the FIRST DRAWN keeps its sign — so a tile whose resolve meets a
// zero of either sign (or a NaN: only a signalling NaN input could put one there) is rastered again with the 64-bit keys.
// (A raw-bits ds_min_i32 variant was built first: exact for depths > 0 only, and cfg4's spans reach outside their triangles
// far enough for extrapolated depths below zero to win in most tiles — 700 us per frame in the fallback.)
__device__ __forceinline__ void k32_min(float* p, float d) {
    __hip_atomic_fetch_min(p, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// does the resolve have to hand this tile to the 64-bit path?
__device__ __forceinline__ bool k32_undecided(float k) {
    return !(k != 0.0f);                 // +-0 (the sign is the first-drawn winner's, :257-261) or NaN
}

constexpr unsigned long long KEY_EMPTY = ~0ull;
// A pixel has a winner iff the high word of its key is below the orderable image of +inf: the z-mode dense step
// stores NaN / +inf depths as +inf keys instead of testing every fragment (they lose against any real depth and
// are never resolved); painter's-mode keys have a zero high word; KEY_EMPTY has all ones.
constexpr uint32_t KEY_LIVE_BELOW = 0xFF800000u;

// Per-lane triangle state for the span walk.
struct TriState {
    Chains ch;
    float cfx, cfy;            // float(C) + 0.5 (:89)
    int cx, cy;                // C (integer), for the exact small-coordinate dx / dy
    float t00, t01, t10, t11;
    float za, zb, zc;
    uint32_t prim;
};

template <bool ZTEST, class KeyT>
__device__ __forceinline__ void fragment(KeyT* keys, const TriState& t, int x, int lidx,
                                         float r0, float r1) {
    // setPixel (:245-269): weights at the pixel centre (x+.5, y+.5) relative to cf
    unsigned long long key;
    if (ZTEST) {
        const float px = (float)x + 0.5f;
        const float dx = px - t.cfx;
        const float w0 = t.t00 * dx + r0;          // r0 = t01 * dy
        const float w1 = t.t10 * dx + r1;          // r1 = t11 * dy
        const float w2 = 1.0f - w0 - w1;           // :92
        float d = t.za * w0 + t.zb * w1 + t.zc * w2;   // :257
        if (!(d < INFINITY)) return;               // can never pass 'depth < buffer' (:258); NaN too
        if constexpr (std::is_same<KeyT, float>::value) { k32_min(&keys[lidx], d); return; }
        d = d + 0.0f;                              // -0 -> +0 for ordering only (== under '<')
        key = ((unsigned long long)orderable_depth(d) << 32) | (unsigned long long)t.prim;
    } else {
        key = (unsigned long long)(0xFFFFFFFFu - t.prim);   // painter's order: highest prim wins
    }
    if constexpr (!std::is_same<KeyT, float>::value) atomicMin(&keys[lidx], key);
}

__device__ __forceinline__ void load_tri(const GeomFull* __restrict__ full, uint32_t prim, const int4& q0,
                                         const float4& q1, TriState& t, int& minx, int& maxx) {
    int vx[3], vy[3];
    decode_vertices(full, prim, q0, q1, vx, vy);
    const uint32_t fl = __float_as_uint(q1.w);
    const int o0 = (fl >> GEOM_ORD_SHIFT) & 3, o1 = (fl >> (GEOM_ORD_SHIFT + 2)) & 3,
              o2 = (fl >> (GEOM_ORD_SHIFT + 4)) & 3;
    // select without dynamic indexing (keeps the arrays in registers)
    auto pick = [](int o, int a, int b, int c) { return o == 0 ? a : (o == 1 ? b : c); };
    t.ch.s0x = pick(o0, vx[0], vx[1], vx[2]); t.ch.s0y = pick(o0, vy[0], vy[1], vy[2]);
    t.ch.s1x = pick(o1, vx[0], vx[1], vx[2]); t.ch.s1y = pick(o1, vy[0], vy[1], vy[2]);
    t.ch.s2x = pick(o2, vx[0], vx[1], vx[2]); t.ch.s2y = pick(o2, vy[0], vy[1], vy[2]);
    t.ch.small = (fl & GEOM_SMALL) != 0;
    t.ch.r01 = t.ch.r12 = t.ch.r02 = 0.0f;     // only the cooperative walk uses them: computed there, once per large triangle
    t.cfx = (float)vx[2] + 0.5f;
    t.cfy = (float)vy[2] + 0.5f;
    t.cx = vx[2];
    t.cy = vy[2];
    tinv_of(vx[0], vy[0], vx[1], vy[1], vx[2], vy[2], t.t00, t.t01, t.t10, t.t11);
    t.za = q1.x; t.zb = q1.y; t.zc = q1.z;
    t.prim = prim;
    minx = min(vx[0], min(vx[1], vx[2]));
    maxx = max(vx[0], max(vx[1], vx[2]));
}

// Lane mask of a <= b as a scalar (v_cmp writes 0 for lanes outside EXEC).
#ifndef SWR_ASM_MASKS
#define SWR_ASM_MASKS 1
#endif
__device__ __forceinline__ unsigned long long lane_mask_le(int a, int b) {
#if SWR_ASM_MASKS
    unsigned long long m;
    asm("v_cmp_le_i32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
#else
    return __builtin_amdgcn_ballot_w64(a <= b);
#endif
}
__device__ __forceinline__ unsigned long long lane_mask_ne(uint32_t a, uint32_t b) {
#if SWR_ASM_MASKS
    unsigned long long m;
    asm("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
#else
    return __builtin_amdgcn_ballot_w64(a != b);
#endif
}
__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float bcast_f(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

struct MetalTri {
    float p3x, p3y, A0, B0, A1, B1, divider, z0, z1, z2;
};
__device__ __forceinline__ void metal_consts(const int vx[3], const int vy[3], float za, float zb, float zc,
                                             MetalTri& m) {
    const float p1x = (float)vx[0], p1y = (float)vy[0], p2x = (float)vx[1], p2y = (float)vy[1];
    m.p3x = (float)vx[2]; m.p3y = (float)vy[2];
    m.divider = (p1x - m.p3x) * (p2y - m.p3y) - (p2x - m.p3x) * (p1y - m.p3y);   // :143
    m.A0 = p2y - m.p3y; m.B0 = m.p3x - p2x;                                       // :144
    m.A1 = m.p3y - p1y; m.B1 = p1x - m.p3x;                                       // :147
    m.z0 = za; m.z1 = zb; m.z2 = zc;
}
__device__ __forceinline__ bool metal_weights(const MetalTri& m, int x, int y, float& w0, float& w1, float& w2) {
    const float sx = (float)x + 0.5f, sy = (float)y + 0.5f;                       // :133
    w0 = m.A0 * (sx - m.p3x) + m.B0 * (sy - m.p3y);
    w0 = w0 / m.divider;                                                          // :145
    w1 = m.A1 * (sx - m.p3x) + m.B1 * (sy - m.p3y);
    w1 = w1 / m.divider;                                                          // :148
    w2 = 1.0f - w0 - w1;                                                          // :149
    return 0.0f <= w0 && w0 <= 1.0f && 0.0f <= w1 && w1 <= 1.0f && 0.0f <= w2 && w2 <= 1.0f;   // :153
}

// The same two quotients with the divisor's share of the work done once (the winner table's shading pass): the division sequence of
// tinv_of — reciprocal, one Newton step, quotient, two residual corrections, v_div_fixup — whose argument holds here too: the divider
// is an integer-valued float below 2^62 in magnitude (or 0: fix-up gives the inf / NaN that '/' gives), the numerators are sums of two
// products of such values, nothing needs scaling.  Correctly rounded, bit for bit what metal_weights computes (every Metal-rules test
// with colour goes through here).
__device__ __forceinline__ void metal_weights_shared_rcp(const MetalTri& m, int x, int y, float& w0, float& w1, float& w2) {
    const float sx = (float)x + 0.5f, sy = (float)y + 0.5f;                       // :133
    const float rc0 = __builtin_amdgcn_rcpf(m.divider);
    const float rcp = __builtin_fmaf(__builtin_fmaf(-m.divider, rc0, 1.0f), rc0, rc0);
    auto div_exact = [&](float n) {
        const float q0 = n * rcp;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-m.divider, q0, n), rcp, q0);
        const float q2 = __builtin_fmaf(__builtin_fmaf(-m.divider, q1, n), rcp, q1);
        return __builtin_amdgcn_div_fixupf(q2, m.divider, n);
    };
    w0 = div_exact(m.A0 * (sx - m.p3x) + m.B0 * (sy - m.p3y));                    // :144-145
    w1 = div_exact(m.A1 * (sx - m.p3x) + m.B1 * (sy - m.p3y));                    // :147-148
    w2 = 1.0f - w0 - w1;                                                          // :149
}

// ---- wave64 scans on DPP (all 64 lanes must be active) ---------------------------------------
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp0(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xF, false);   // lanes without a source get 0
}
// One instruction per scan step: OP vdst, dpp(src0), src1 with vdst = src0 = src1; lanes whose DPP
// source is invalid or whose row is masked off keep their value (bound_ctrl off).  s_nop 1 covers
// the VALU-write -> DPP-read hazard (2 wait states), which hipcc cannot see inside the asm.
#define SWR_DPP_SCAN(OP, v)                                                                    \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"      \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"      \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"      \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"      \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"   \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"   \
                 "s_nop 1"                                                                     \
                 : "+v"(v))
__device__ __forceinline__ int wave_incl_add(int v) {
    SWR_DPP_SCAN("v_add_u32_dpp", v);
    return v;
}
// VAR > 0: timing-only ablations (results invalid), instantiated only under -DSWR_ABLATION (`make ablation`):
//   1 = no LDS atomic, 2 = no per-pixel maths, 3 = producer only (no unit consumed), 4 = no row walk at all,
//   8 = no resolve, 9 = no chunk at all, 10 = 4 + 8, 11 = 9 + 8 (tools/ablate.py)
// VAR_ALLCOOP is not an ablation: every triangle takes the cooperative walk (lanes = pixels, one triangle at a time) and the
// ring machinery is compiled out — the register-light, exact, ~6x slower way k_raster_depth re-rasters a tile whose 32-bit keys
// could not decide it.
constexpr int VAR_ALLCOOP = 20;
#ifndef SWR_RASTER_MIN_WAVES_EXT
#define SWR_RASTER_MIN_WAVES_EXT 5   // (4 while the extended stage resolved per thread: > 96 VGPRs; with the winner table 5 waves fit: cfg5 textured 0.303 -> 0.283 ms)
#endif
#ifndef SWR_RASTER_MIN_WAVES
#define SWR_RASTER_MIN_WAVES 5   // waves per SIMD the register allocator must allow (measured: 5 spill-free beats 6)
#endif
// METAL = the Metal path's rules (SWR_FLAG_METAL_RULES; Shaders.metal:123-167) on the same machinery:
// the "span" of every row of the ROI is the ROI's x-range, the per-pixel maths is the `divider`
// barycentric with the inside test, the store is bgra8Unorm.
// EXT = the extended fragment stage (normal / uv varyings, fragment_shader(vin, uniforms)) at the resolve.
// Register budget: 88 VGPRs, not the 96 that five waves per SIMD would allow.  Registers are allocated in blocks of 8, and
// the binning kernel of the NEXT frame (k_bin: 56) has to fit beside five raster waves on a SIMD (5 x 88 + 56 <= 512):
// at 91 VGPRs (allocated 96) k_raster alone is as fast, but k_bin finds no room beside it and the pipelined cfg4 frame
// goes from 0.086 to 0.091 ms (profiles/r03/vgpr_budget_ab.txt).  Check with `make asm` after every change: 86 now
// (Metal rules 88).
#ifndef SWR_RASTER_VGPRS
#define SWR_RASTER_VGPRS 88
#endif
// K32: 32-bit depth keys (RasterLds32).  Returns true (workgroup-uniform) when the tile has to be rastered again with the
// 64-bit keys — only a K32 instance ever does.
template <bool ZTEST, int VAR, bool METAL, bool EXT, bool COLOR, bool PLAIN = false, bool K32 = false>
__device__ __forceinline__ bool raster_tile(const RasterArgs& a, typename std::conditional<K32, RasterLds32, RasterLds64>::type& L) {
    static_assert(!METAL || ZTEST, "the Metal rules always z-test");
    static_assert(!EXT || COLOR, "the extended fragment stage only exists for colour frames");
    static_assert(!K32 || (ZTEST && !COLOR && !METAL), "32-bit keys: depth-only z-tested frames under the CPU rules");
#ifndef SWR_UNIT
#define SWR_UNIT 4
#endif
    constexpr int UNIT = SWR_UNIT;    // consecutive pixels of one span handled by one lane of a dense step

    uint32_t& next_chunk = L.next_chunk;           // work-stealing cursor over the chunks of the sorted bin
    float4* const tabA = L.tabAB;                  // per triangle of the batch: t00, t01, t10, t11
    float4* const tabB = L.tabAB + RASTER_THREADS; //                            za, zb, zc, (C.x - X0) | (C.y - Y0) << 16
    uint32_t* const slots = reinterpret_cast<uint32_t*>(L.tabAB);   // resolve only (see below)
    uint32_t* const tabP = L.tabP;                 //                            original primitive index (the key's low word)
    // per wave: ring of spans waiting for a lane.  entry = owner lane | xl << 6 | yl << 12 | (pixels - 1) << 17.
    // At most 63 entries wait when a producer step adds up to 64; a consumer step pops n and pushes back at most n.
    constexpr int QCAP = RASTER_QCAP;
    auto& queue = L.queue;
    auto* const keys = L.keys;
    // after the last chunk the per-triangle tables are dead: the resolve keeps the winners' stream slots there
    static_assert(sizeof(float4) * 2 * RASTER_THREADS >= sizeof(uint32_t) * TILE_W * TILE_H, "slots alias tabA + tabB");
    // (Early-z — sub-tile maxima of the keys, triangles dropped by a conservative depth bound over their clipped bounding
    // box — was built twice and measured slower both times: refreshed at every chunk in round 2 (profiles/r02/earlyz_ab.txt),
    // occluder-first with one barrier per tile in round 3 (profiles/r03/earlyz_occluder_first_ab.txt).  Not in the kernel.)

    const int tile = (int)(blockIdx.x >> a.vs_log);
    const int part = (int)(blockIdx.x & ((1u << a.vs_log) - 1u));     // which slice of the tile's rows this workgroup owns
    const int PROWS = TILE_H >> a.vs_log;
    const int tx = tile % a.tg.tiles_x, ty = tile / a.tg.tiles_x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // tile rectangle in full-image coordinates (inclusive)
    const int X0 = tx * TILE_W, Y0 = a.tg.row_begin + ty * TILE_H;
    const int X1 = min(X0 + TILE_W, a.tg.width) - 1;
    const int Y1 = min(Y0 + TILE_H, a.tg.row_end) - 1;
    // rows of the tile this workgroup walks and resolves (the whole tile unless the grid is split, vs_log > 0)
    const int Yp0 = Y0 + part * PROWS;
    const int Yp1 = min(Yp0 + PROWS - 1, Y1);
    if (Yp0 > Y1) return false;     // the band ends above this slice (workgroup-uniform)

    // an overflowed frame (more pairs than the bins hold / a tile region too small) is rastered empty: the host grows the
    // bins and redraws it
    uint32_t b0 = 0u, b1 = 0u;
    // the fullest bin of the frame: k_bin's maximum, raised by k_sort_bins if the deferred triangles it appended overfilled a tile
    const uint32_t max_fill = a.fixed_cap ? a.fill[CNT_MAXFILL] : 0u;
    if (a.fixed_cap && blockIdx.x == 0 && threadIdx.x == 0) {
        *a.host_pairs = a.fill[CNT_PAIRS];
        *a.host_fill = max_fill;
        // (bit 31: the frame had triangles for the deferred list — the host then keeps k_sort_bins running, which is what appends them)
        if (a.host_max) *a.host_max = a.fill[CNT_MAXFILL] | ((a.fill[CNT_BIGLIST] | a.fill[CNT_BIGSEEN]) ? 0x80000000u : 0u);
    }
    if (a.fixed_cap) {
        if (max_fill <= a.fixed_cap) { b0 = (uint32_t)tile * a.fixed_cap; b1 = b0 + a.fill[CNT_WORDS + tile]; }
    } else if (a.counters[CNT_PAIRS] <= a.capacity) {
        b0 = a.tile_start[tile]; b1 = a.tile_start[tile + 1];
    }
    const uint32_t m = b1 - b0;   // bin sorted by size class (k_sort_bins), heaviest first — unless the host skipped the sort
    // (sparse frames, launch_sort_bins): the entries then still carry k_fill_lds's class tag
    const uint32_t bin_mask = a.tag_class ? (1u << CLASS_SHIFT) - 1u : 0xFFFFFFFFu;
    // Colour frames with the reference's fragment stage: the key's low word also carries the triangle's position in this bin
    // (below the original index, which still decides ties), so the resolve can set up every WINNER once — see "winner table" there.
    // (PLAIN: the kernels for scenes of more than 2^20 primitives — no room for the bin position in the key: the per-thread resolve)
    constexpr bool WTAB_OK = COLOR && !K32 && VAR == 0 && !PLAIN;
    const bool wtab = WTAB_OK && a.pack_local != 0 && m <= (1u << WTAB_LOCAL_BITS);      // (workgroup-uniform)

    // the gather chain of the first batch (bin entry -> record) is issued before the LDS init so
    // that its latency overlaps the init and the barrier
    uint32_t prim_pre = 0u;
    int4 q0_pre = make_int4(0, 0, 0, 0);
    float4 q1_pre = make_float4(0, 0, 0, 0);
    // The sorted bin is cut into contiguous chunks of `csz` entries (one per lane): 64 when the grid
    // saturates the chip (fewest row steps in total), m/4 when it does not (small scenes: shortest
    // critical path).  The first 4 chunks go to the 4 waves statically (their gathers are issued
    // right here); after that a wave that finishes its chunk steals the next one from an LDS counter.
    // Chunks are heaviest-first, so the stolen ones are the light ones: the waves of a tile finish
    // together instead of waiting at the final barrier for the wave that drew the heavy chunks.
    // Few triangles in the tile (one or two chunks): chunk parallelism would leave waves idle while one of them
    // walks every row and shades every unit (BASELINE configs 2, 3 and 5: 20-30 triangles per tile).  Then the
    // waves split the ROWS instead: every wave loads every chunk and walks only its quarter of the tile's rows
    // (the row steppers jump to the first row of the quarter).  Costs the per-triangle setup four times, divides
    // the row steps and the pixel work by four.
#ifndef SWR_ROWSPLIT_MAX
#define SWR_ROWSPLIT_MAX 128
#endif
#ifndef SWR_ROWSPLIT_MAX_SPREAD
#define SWR_ROWSPLIT_MAX_SPREAD 192
#endif
    // (up to 192 on grids that do not fill the chip — 1080p is 1 020 tiles: 20 000 triangles of ~100 px 70 -> 54 us, 300
    // screen-filling ones 97 -> 80; cfg4's 4 080 tiles measured 1 % slower with it: profiles/r03/rowsplit_large_ab.txt)
    const bool rowsplit = m <= (uint32_t)(gridDim.x <= 1536 ? SWR_ROWSPLIT_MAX_SPREAD : SWR_ROWSPLIT_MAX);
    const int WROWS = PROWS / (RASTER_THREADS / 64);
    const int Yw0 = rowsplit ? Yp0 + (tid >> 6) * WROWS : Yp0;          // this wave's rows of the tile
    const int Yw1 = rowsplit ? min(Yw0 + WROWS - 1, Yp1) : Yp1;
    const bool spread = !rowsplit && gridDim.x <= 1536;   // fewer tiles than the chip holds workgroups (256 CUs x 6)
    const uint32_t csz = spread ? min(64u, max(1u, (m + RASTER_THREADS / 64 - 1) / (RASTER_THREADS / 64))) : 64u;
    const uint32_t nchunks = (m + csz - 1) / csz;
    uint32_t chunk = rowsplit ? 0u : (uint32_t)(tid >> 6);
    if (tid == 0) next_chunk = RASTER_THREADS / 64;
    const uint32_t slot0 = chunk * csz + (uint32_t)lane;
    const bool have0 = (uint32_t)lane < csz && slot0 < m;
    // 32-bit keys, small grids (a thin band: what one GPU of eight renders): the workgroup sorts its bin by size class itself
    // — what k_sort_bins does in a launch of its own — in the LDS the narrow keys leave free: entries -> registers, class counts
    // by returning LDS atomics, every wave scans the 33 counts for itself (no barrier for the prefix), entries -> LDS in class
    // order, heaviest first.  Two barriers more than the clear needs; the chunks then take their entries from LDS.  A frame of
    // a thin band is bound by its kernel chain (sort 5 us + a 5 us launch gap in front of a 21 us raster), so the launch
    // saved is worth more there than the sort costs inside the raster: worst band of 8, 27.8 -> 26.0-26.4 us per frame.  On a
    // grid that fills the chip it is the other way round — a raster workgroup holds 27 KB of LDS and 87 registers per thread
    // while it waits for its own sort's round trip, k_sort_bins' light workgroups hide theirs behind each other:
    // k_raster_depth alone 64.2 -> 68.5 us, the cfg4 frame 0.0775 -> 0.0792 ms — so the host asks for it on small grids only
    // (profiles/r04/insort_ab.txt; also measured there: wave 0 alone sorting while the others clear — same 68.5 —, and the
    // RECORDS exchanged through LDS too so that nothing is gathered after the sort — 72.7).  Row-split tiles (every wave walks
    // every chunk) do not care about the order and are left alone.
    constexpr int SORT_PER = RASTER_SORT_SEG / RASTER_THREADS;
    bool insort = false;
    uint32_t s_ent[K32 ? SORT_PER : 1];
    if constexpr (K32) {
        insort = a.insort != 0 && a.tag_class != 0 && !rowsplit && m <= (uint32_t)RASTER_SORT_SEG;
        if (insort) {
            if (tid < 64) L.cls_cnt[tid] = 0u;
#pragma unroll
            for (int k = 0; k < SORT_PER; k++) {
                const uint32_t i = (uint32_t)(tid + k * RASTER_THREADS);
                s_ent[k] = i < m ? a.bins[b0 + i] : 0u;
            }
        }
    }
    if (!insort && have0 && VAR != 9 && VAR != 11) {
        prim_pre = a.bins[b0 + slot0] & bin_mask;
        q0_pre = reinterpret_cast<const int4*>(a.geo + prim_pre)[0];
        q1_pre = reinterpret_cast<const float4*>(a.geo + prim_pre)[1];
    }

    // clear fused into the LDS init (Renderer.clear :205-206, :232-236)
    if constexpr (K32) {
        if (tid == 0) L.redo = 0u;
        for (int i = tid; i < TILE_W * TILE_H / 4; i += RASTER_THREADS)
            reinterpret_cast<float4*>(L.keys)[i] = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);   // (:206)
    } else {
        for (int i = tid; i < TILE_W * TILE_H; i += RASTER_THREADS) keys[i] = KEY_EMPTY;
        if constexpr (WTAB_OK) { if (tid < WTAB_WORDS) L.winners[tid] = make_uint2(0u, 0u); }
    }
    __syncthreads();
    if constexpr (K32) {
        if (insort) {                                                   // (workgroup-uniform)
            uint32_t s_pos[SORT_PER];
#pragma unroll
            for (int k = 0; k < SORT_PER; k++) {
                s_pos[k] = 0u;
                if ((uint32_t)(tid + k * RASTER_THREADS) < m) s_pos[k] = atomicAdd(&L.cls_cnt[s_ent[k] >> CLASS_SHIFT], 1u);
            }
            __syncthreads();
            // lane l holds class NUM_CLASSES - 1 - l: exclusive prefix in descending class order (heaviest first)
            const int cl = NUM_CLASSES - 1 - lane;
            const uint32_t cv = cl >= 0 ? L.cls_cnt[cl] : 0u;
            const uint32_t cbase = (uint32_t)wave_incl_add((int)cv) - cv;
#pragma unroll
            for (int k = 0; k < SORT_PER; k++) {
                const uint32_t cls = s_ent[k] >> CLASS_SHIFT;
                const uint32_t base = (uint32_t)__shfl((int)cbase, (int)(NUM_CLASSES - 1 - cls));
                if ((uint32_t)(tid + k * RASTER_THREADS) < m) L.sorted[base + s_pos[k]] = s_ent[k] & bin_mask;
            }
            __syncthreads();
            if (have0) {
                prim_pre = L.sorted[slot0];
                q0_pre = reinterpret_cast<const int4*>(a.geo + prim_pre)[0];
                q1_pre = reinterpret_cast<const float4*>(a.geo + prim_pre)[1];
            }
        }
    }

    while (VAR != 9 && VAR != 11 && chunk < nchunks) {
        const uint32_t e = chunk * csz + (uint32_t)lane;
        const bool have = (uint32_t)lane < csz && e < m;
        TriState t;
        int ya = 1, yb = 0, bxa = 0, bxb = -1;
        bool big = false, large = false;
        if (have) {
            int minx, maxx;
            // bin entry and record of this chunk were fetched ahead: the first chunk's before the LDS init, a later
            // one's while the previous chunk was being finished (steal_next below)
            const uint32_t prim = prim_pre;
            const int4 q0 = q0_pre;
            const float4 q1 = q1_pre;
            if (METAL) {
                // TriState re-used: t00..t11 = A0,B0,A1,B1 of the `divider` formula, cf.y = p3.y, cx = p3.x;
                // ch.s0..s2 = the snapped vertices in y order (the order bits of the record: a stable sort of the snapped
                // y's), for the row walk of the dense phase; the ROI's x-range is [minx, maxx]
                int vx[3], vy[3];
                decode_vertices(a.geo_full, prim, q0, q1, vx, vy);
                MetalTri mt;
                metal_consts(vx, vy, q1.x, q1.y, q1.z, mt);
                t.t00 = mt.A0; t.t01 = mt.B0; t.t10 = mt.A1; t.t11 = mt.B1;
                t.za = mt.z0; t.zb = mt.z1; t.zc = mt.z2;
                t.cfx = mt.p3x; t.cfy = mt.p3y; t.cx = vx[2]; t.cy = vy[2];
                t.prim = prim;
                const uint32_t fl = __float_as_uint(q1.w);
                t.ch.small = (fl & GEOM_SMALL) != 0;
                const int o0 = (fl >> GEOM_ORD_SHIFT) & 3, o1 = (fl >> (GEOM_ORD_SHIFT + 2)) & 3, o2 = (fl >> (GEOM_ORD_SHIFT + 4)) & 3;
                auto pick = [](int o, int a0, int b0, int c0) { return o == 0 ? a0 : (o == 1 ? b0 : c0); };
                t.ch.s0x = pick(o0, vx[0], vx[1], vx[2]); t.ch.s0y = pick(o0, vy[0], vy[1], vy[2]);
                t.ch.s1x = pick(o1, vx[0], vx[1], vx[2]); t.ch.s1y = pick(o1, vy[0], vy[1], vy[2]);
                t.ch.s2x = pick(o2, vx[0], vx[1], vx[2]); t.ch.s2y = pick(o2, vy[0], vy[1], vy[2]);
                t.ch.r01 = t.ch.r12 = t.ch.r02 = 0.0f;
                minx = min(vx[0], min(vx[1], vx[2])); maxx = max(vx[0], max(vx[1], vx[2]));
            } else {
                load_tri(a.geo_full, prim, q0, q1, t, minx, maxx);
            }
            // visibility keys order by the ORIGINAL primitive index (Renderer.swift:222,258)
            if (a.reordered) t.prim = __float_as_uint(q1.w) >> GEOM_ORIG_SHIFT;
            if (WTAB_OK && wtab) t.prim = (t.prim << WTAB_LOCAL_BITS) | e;     // ... and the position in the bin below it (winner table)
            ya = max(t.ch.s0y, Yw0);
            // Metal rules: the samples of the ROI's last row lie at max-y + 0.5, half a pixel below every vertex: never
            // inside (the float evaluation of :144-153 errs by ~2^-8 px at most for GEOM_SMALL extents), so it is not walked
            yb = min(METAL ? t.ch.s2y - 1 : t.ch.s2y, Yw1);
            bxa = max(minx, X0);
            bxb = min(maxx, X1);
            // the dense path below needs the exact small-coordinate arithmetic; everything else
            // (huge extents, large clipped area) is walked cooperatively in phase 2
            big = !t.ch.small || maxx - minx >= 16384 || (BIG_AREA < TILE_W * TILE_H && (yb - ya + 1) * (bxb - bxa + 1) > BIG_AREA);
            // "large" = half of what this WAVE walks of the tile (all of its rows, or its share of them when the waves split
            // the rows): its spans are then about 32 pixels or more, the length the wide visits below are made for.  (Measured
            // against half the TILE, a wave that walks 8 rows never saw a large triangle and sent 64-pixel spans through the
            // ring 4 pixels at a time: 300 screen-filling triangles 252 us, 93 with this line.)
            large = !big && (yb - ya + 1) * (bxb - bxa + 1) >= (LARGE_AREA / TILE_H) * (Yw1 - Yw0 + 1);
        }

        // A FEW large triangles among many small ones (a ground plane, a wall, an occluder: at most LARGE_MAX lanes of
        // the chunk cover half the tile or more) are walked cooperatively too: in the dense phase their lane would
        // queue QMAXU units per step and hold the whole wave back for 5-6 steps per row (occluded soup: 238 -> 108 us).
        // When every triangle of the chunk is large the dense phase is the faster one (300 screen-filling triangles:
        // 177 vs 257 us), so the count decides.
        {
            const unsigned long long lm = __ballot(have && large);
            if (lm != 0ull && __popcll(lm) <= LARGE_MAX) big = big || large;
        }
        if (VAR == VAR_ALLCOOP) big = true;
        // ---- big or huge-coordinate triangles first, one at a time, walked by the whole wave ----
        // (done before the dense phase so that its per-triangle registers die early)
        unsigned long long bigmask = __ballot(have && big);
        while (bigmask) {
            const int src = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)bigmask) - 1);
            bigmask &= bigmask - 1;
            if (METAL) {
                // ROI ∩ tile of one huge-coordinate triangle, lanes = pixels of a 16x4 / 32x2 / 64x1 chunk
                MetalTri mt;
                mt.A0 = bcast_f(t.t00, src); mt.B0 = bcast_f(t.t01, src);
                mt.A1 = bcast_f(t.t10, src); mt.B1 = bcast_f(t.t11, src);
                mt.z0 = bcast_f(t.za, src); mt.z1 = bcast_f(t.zb, src); mt.z2 = bcast_f(t.zc, src);
                mt.p3x = bcast_f(t.cfx, src); mt.p3y = bcast_f(t.cfy, src);
                mt.divider = mt.B1 * mt.A0 - mt.B0 * mt.A1;      // == (p1-p3)x(p2-p3): same products, same rounding
                const uint32_t uprim = (uint32_t)bcast_i((int)t.prim, src);
                const int xa = bcast_i(bxa, src), xb = bcast_i(bxb, src);              // ROI x-range ∩ tile
                const int uya = bcast_i(ya, src), uyb = bcast_i(yb, src);
                if (xa > xb) continue;
                const int w = xb - xa + 1;
                const int lw = w <= 16 ? 4 : (w <= 32 ? 5 : 6);
                const int cw = 1 << lw, rows_per = 64 >> lw;
                for (int yr = uya; yr <= uyb; yr += rows_per) {
                    const int y = yr + (lane >> lw);
                    for (int x = xa + (lane & (cw - 1)); x <= xb; x += cw) {
                        float w0, w1, w2;
                        if (y <= uyb && metal_weights(mt, x, y, w0, w1, w2)) {
                            float z = w0 * mt.z0 + w1 * mt.z1 + w2 * mt.z2;
                            if (z < INFINITY) {
                                z = z + 0.0f;
                                if constexpr (!K32)
                                    atomicMin(&keys[(y - Y0) * TILE_W + (x - X0)],
                                              ((unsigned long long)orderable_depth(z) << 32) | (unsigned long long)uprim);
                            }
                        }
                    }
                }
                continue;
            }
            TriState u;
            u.ch.s0x = bcast_i(t.ch.s0x, src); u.ch.s0y = bcast_i(t.ch.s0y, src);
            u.ch.s1x = bcast_i(t.ch.s1x, src); u.ch.s1y = bcast_i(t.ch.s1y, src);
            u.ch.s2x = bcast_i(t.ch.s2x, src); u.ch.s2y = bcast_i(t.ch.s2y, src);
            u.ch.r01 = __builtin_amdgcn_rcpf((float)(u.ch.s1y - u.ch.s0y));     // (wave-uniform; same values the per-lane setup used to compute)
            u.ch.r12 = __builtin_amdgcn_rcpf((float)(u.ch.s2y - u.ch.s1y));
            u.ch.r02 = __builtin_amdgcn_rcpf((float)(u.ch.s2y - u.ch.s0y));
            u.ch.small = bcast_i(t.ch.small ? 1 : 0, src) != 0;
            u.cfx = bcast_f(t.cfx, src); u.cfy = bcast_f(t.cfy, src);
            u.t00 = bcast_f(t.t00, src); u.t01 = bcast_f(t.t01, src);
            u.t10 = bcast_f(t.t10, src); u.t11 = bcast_f(t.t11, src);
            u.za = bcast_f(t.za, src); u.zb = bcast_f(t.zb, src); u.zc = bcast_f(t.zc, src);
            u.prim = (uint32_t)bcast_i((int)t.prim, src);
            const int uya = bcast_i(ya, src), uyb = bcast_i(yb, src);

            for (int yc = uya; yc <= uyb; yc += 64) {
                // lane = row: each lane computes the span of one row of this 64-row chunk
                const int yrow = yc + lane;
                int lo = 0, hi = -1;
                if (yrow <= uyb) {
                    row_span(u.ch, yrow, lo, hi);
                    lo = max(lo, X0);
                    hi = min(hi, X1);
                }
                // chunk shape from the widest span: 16x4, 32x2 or 64x1 pixels per wave step
                int wmax = hi - lo + 1;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off));
                const int lw = wmax <= 16 ? 4 : (wmax <= 32 ? 5 : 6);
                const int cw = 1 << lw, rows_per = 64 >> lw;
                const int nrows = min(64, uyb - yc + 1);
                for (int r0i = 0; r0i < nrows; r0i += rows_per) {
                    const int r = r0i + (lane >> lw);
                    const int rlo = __shfl(lo, r & 63), rhi = __shfl(hi, r & 63);
                    const int y = yc + r;
                    const bool rowok = r < nrows;
                    const float dy = ((float)y + 0.5f) - u.cfy;
                    const float q0 = u.t01 * dy, q1 = u.t11 * dy;
                    const int rowbase = (y - Y0) * TILE_W - X0;
                    if (rowok)
                        for (int x = rlo + (lane & (cw - 1)); x <= rhi; x += cw)
                            fragment<ZTEST>(keys, u, x, rowbase + x, q0, q1);
                }
            }
        }
        // ---- dense phase: lane = triangle for the row walk, lane = span for the pixel work -------------------------
        // Producer: every lane steps through the rows of ITS OWN (small) triangle and appends ONE entry per non-empty
        // span to the wave's ring in LDS (owner lane | x | y | length - 1; its position is the lane's rank in one
        // ballot).  There is no per-row limit: a row step never has to be repeated for a wide span.
        // Consumer: whenever 64 entries wait (or the rows have run out) every lane takes one: the owner's constants come
        // from the LDS tables (2 x ds_read_b128 + ds_read_b32), then UNIT x (weights, depth, ds_min_u64) for the first
        // UPX pixels of the span; what is left of a longer span goes back into the ring as a new entry (same owner,
        // x advanced), so consumer steps stay full while the rows last.  Tail of a chunk (fewer than 64 entries, no
        // rows left): an entry is shared by 64 / pow2(entries) lanes, lane group j taking the pixels [j UPX, (j + 1) UPX),
        // so the last long spans do not trickle out four pixels per step.
        // Wide chunks (at least half of the chunk's triangles cover half the tile or more: walls, ground planes,
        // screen-filling triangles): the same machinery with UPX = 32 pixels per visit (SL = 3) instead of 4 — the
        // consumer walks its share in eight groups of four and a 64-pixel row is two visits.  The two instantiations
        // are chosen per chunk (wave-uniform).
        // The next chunk of this wave (the following one when the waves split rows, else stolen from the counter) is
        // chosen as soon as the rows of this one have run out, and its bin entries are fetched while the ring is
        // drained — the gather chain bin entry -> record of a chunk is two dependent round trips to memory.  (Stealing
        // at the START of a chunk was measured in round 2 and lost: a chunk reserved by a wave that is still busy with a
        // heavy one is a chunk no idle wave can take.  Here the wave has at most a few consumer steps left.)
#ifndef SWR_STEAL_AHEAD
#define SWR_STEAL_AHEAD 0   // measured: choosing the next chunk when the rows run out (2-3 consumer steps early) costs k_raster 70.7 -> 73.7 us, profiles/r03/steal_ahead_ab.txt
#endif
        uint32_t chunk_next = nchunks;
        bool have_next = false;
        auto steal_next = [&]() {
            if (rowsplit) {
                chunk_next = chunk + 1u;
            } else {
                uint32_t nx = 0u;
                if (lane == 0) nx = atomicAdd(&next_chunk, 1u);
                chunk_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx);
            }
            const uint32_t e2 = chunk_next * csz + (uint32_t)lane;
            have_next = chunk_next < nchunks && (uint32_t)lane < csz && e2 < m;
            if constexpr (K32) {
                if (have_next) prim_pre = insort ? L.sorted[e2] : a.bins[b0 + e2] & bin_mask;
            } else {
                if (have_next) prim_pre = a.bins[b0 + e2] & bin_mask;
            }
        };
        auto dense = [&](auto SLc) {
            constexpr int SL = decltype(SLc)::value;            // log2 of the 4-pixel groups per visit
            constexpr int UPX = UNIT << SL;                     // pixels of a span one lane handles per visit
            const bool mine = have && !big;
            const int wbase = tid & ~63;
            // same-wave producers and consumers: LDS operations of one wave execute in order; the wavefront-scope
            // fences below only keep the compiler from moving the accesses across them
            if (ZTEST) {
                tabA[tid] = make_float4(t.t00, t.t01, t.t10, t.t11);
                tabB[tid] = make_float4(t.za, t.zb, t.zc,
                                        __int_as_float((int)(((uint32_t)(t.cx - X0) & 0xFFFFu) | ((uint32_t)(t.cy - Y0) << 16))));
            }
            if (!K32) tabP[tid] = t.prim;
            const int y = mine ? ya : 1;        // the first row; the walk itself counts in ebase (below)
            const int ye = mine ? yb : 0;
            uint32_t qhead = 0u, qcount = 0u;   // wave-uniform
            bool stolen = false;                // the next chunk has been chosen (wave-uniform)
            uint32_t* const q = queue[tid >> 6];
            // ring entry = owner lane | xl << 6 | yl << 12 | (pixels - 1) << 17, xl / yl tile-local
            static_assert(TILE_W == 64 && TILE_H == 32 && UNIT >= 1 && UNIT <= 8, "ring entry layout: 6 + 6 + 5 + 6 bits");
            uint32_t ebase = ((uint32_t)lane | ((uint32_t)((y - Y0) & (TILE_H - 1)) << 12)) - ((uint32_t)X0 << 6);
            // the row counter is ebase itself (its y field grows by one per row): the walk ends at ebase_stop, the kink is at ebase_kink
            const uint32_t ebase_stop = ebase + ((uint32_t)max(ye - y + 1, 0) << 12);
            const uint32_t ebase_kink = ebase + ((uint32_t)(t.ch.s1y - y) << 12);       // (a kink above the first row is never met)
            // The two chains of draw(triangle:) (:276-277) as row steppers.  Left chain [S0,S1,S2]: the segment the
            // first row of the tile falls into, switched to [S1,S2] at the row y == S1.y (:469-475); right chain [S0,S2].
            // At y == S2.y the interpolant returns S2.x (:469-471): the stepper of [S1,S2] arrives there by itself
            // (D * dy / dy = D exactly), except when S1.y == S2.y, where that segment is never interpolated — it then
            // starts (and stays) at S2.x.
            EdgeStep eL = {0, 0, 0, 0, 0, 1}, eK = eL, eR = eL;
            const int s1y = t.ch.s1y;
            // Metal rules: the same two chains bound the pixels that can pass the inside test of a row (below); the chain
            // there is the geometric one, so [S1,S2] starts at S1.x also when S1.y == S2.y.
            if (mine) {
                float rc0, rc1, rcr;
                EdgeStep e0;
                const int k1x = (!METAL && t.ch.s1y == t.ch.s2y) ? t.ch.s2x : t.ch.s1x;
                edge_consts(t.ch.s0x, t.ch.s0y, t.ch.s1x, t.ch.s1y, e0, rc0);
                edge_consts(k1x, t.ch.s1y, t.ch.s2x, t.ch.s2y, eK, rc1);
                edge_consts(t.ch.s0x, t.ch.s0y, t.ch.s2x, t.ch.s2y, eR, rcr);
                const bool in1 = y >= s1y;                       // the tile starts at or below the kink
                eL = in1 ? eK : e0;
                edge_jump(in1 ? k1x : t.ch.s0x, in1 ? t.ch.s2x : t.ch.s1x, y - (in1 ? t.ch.s1y : t.ch.s0y), in1 ? rc1 : rc0, eL);
                edge_jump(t.ch.s0x, t.ch.s2x, y - t.ch.s0y, rcr, eR);
            }
            auto rank_of = [](unsigned long long m) {
                return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            };
            // One consumer step.  TAIL = false: 64 entries, lane = entry.  TAIL = true: the qcount < 64 entries that are
            // left, each shared by 64 >> sh lanes.
            auto consume = [&](auto TAILc) {
                constexpr bool TAIL = decltype(TAILc)::value;
                const uint32_t n = TAIL ? qcount : 64u;
                const int sh = TAIL ? (n <= 1u ? 0 : 32 - __builtin_clz(n - 1u)) : 6;     // 1 << sh >= n (wave-uniform)
                const uint32_t idx = TAIL ? (uint32_t)lane & ((1u << sh) - 1u) : (uint32_t)lane;
                const int off = TAIL ? (lane >> sh) * UPX : 0;                            // first pixel of this lane's share
                const int visit = (64 >> sh) * UPX;                                       // pixels of the span handled by this step
                const uint32_t e = q[(qhead + idx) & (uint32_t)(QCAP - 1)];
                const int len = (int)((e >> 17) & 63u) + 1;
                const bool on = !TAIL || (idx < n && off < len);
                const int owner = (int)(e & 63u);
                const int lidx00 = (int)((e >> 6) & 2047u) + off;                // yl * TILE_W + xl
                const int xl00 = (int)((e >> 6) & 63u) + off, yl = (int)((e >> 12) & 31u);
                const int nvalid0 = on ? min(len - off, UPX) : 0;
                const uint32_t oprim = K32 ? 0u : tabP[wbase + owner];
                float4 ta = make_float4(0, 0, 0, 0), tb = make_float4(0, 0, 0, 0);
                if (ZTEST) { ta = tabA[wbase + owner]; tb = tabB[wbase + owner]; }
                const int cp = __float_as_int(tb.w);
                const int dyi = yl - (cp >> 16);                                 // y - C.y
                // Wide visits (screen-filling triangles: the lanes of a step sit in the same row and the same 32 pixels) walk
                // their groups in a lane-rotated order, so that one ds_min_u64 instruction meets eight addresses instead of
                // one: 300 screen-filling triangles 281 -> 252 us.  It costs three VGPRs: affordable only inside the 88-register
                // budget above (it once pushed the kernel to 91 and the pipelined cfg4 frame up by 6 %,
                // profiles/r03/vgpr_budget_ab.txt; moving the chains' reciprocals out of the per-lane setup made the room).
#ifndef SWR_WIDE_ROT
#define SWR_WIDE_ROT 1
#endif
                const bool rotated = SWR_WIDE_ROT && SL > 0 && __any(nvalid0 > UNIT);           // more than one group somewhere (wave-uniform)
                const int rot = rotated ? (lane & ((1 << SL) - 1)) : 0;
#pragma unroll 1
                for (int sg0 = 0; sg0 < (1 << SL); sg0++) {                      // the visit's groups of four pixels (one: SL == 0)
                    if (SL > 0 && !rotated && (SWR_WIDE_ROT ? sg0 > 0 : !__any(nvalid0 > UNIT * sg0))) break;   // wave-uniform
                    const int sg = SL > 0 ? ((sg0 + rot) & ((1 << SL) - 1)) : 0;
                    const int xl0 = xl00 + UNIT * sg, lidx0 = lidx00 + UNIT * sg;
                    const int nvalid = min(max(nvalid0 - UNIT * sg, 0), UNIT);
                    const int dxi = xl0 - (int)(short)(cp & 0xFFFF);             // x - C.x of the group's first pixel
                    if (VAR == 2 || VAR == 5) { asm volatile("" ::"v"(ta.x), "v"(ta.y), "v"(ta.z), "v"(ta.w), "v"(tb.x), "v"(tb.y), "v"(tb.z), "v"(dxi), "v"(dyi), "v"(nvalid), "v"(lidx0), "v"(oprim)); }
                    else if (METAL) {
                        // Shaders.metal:133-161 with ta = (A0,B0,A1,B1), tb = (z1,z2,z3, .); small integer coordinates:
                        // (x + .5) - p3.x == (x - p3.x) + .5 exactly
                        const float dxp0 = (float)dxi + 0.5f;
                        const float dyp = (float)dyi + 0.5f;
                        const float divider = ta.w * ta.x - ta.y * ta.z;               // :143
                        const float n0 = ta.y * dyp, n1 = ta.w * dyp;
                        // n / divider, correctly rounded, with the divisor's share of the work hoisted out of
                        // the pixel loop: this is the compiler's own f32 division sequence (rcp, one Newton
                        // step, quotient, two residual corrections) minus v_div_scale / v_div_fixup, which are
                        // identities here — GEOM_SMALL triangles have an integer divider with 1 <= |divider| <
                        // 2^31 and numerators that are 0 or multiples of 1/4 below 2^33, so nothing is scaled,
                        // denormal, infinite or NaN.  (The wave-cooperative path and the resolve divide plainly.)
                        const float rc0 = __builtin_amdgcn_rcpf(divider);
                        const float rcp = __builtin_fmaf(__builtin_fmaf(-divider, rc0, 1.0f), rc0, rc0);
                        auto div_exact = [&](float nn) {
                            const float q0 = nn * rcp;
                            const float q1 = __builtin_fmaf(__builtin_fmaf(-divider, q0, nn), rcp, q0);
                            return __builtin_fmaf(__builtin_fmaf(-divider, q1, nn), rcp, q1);
                        };
#pragma unroll
                        for (int qq = 0; qq < UNIT; qq++) {
                            const float dxp = dxp0 + (float)qq;
                            float w0 = ta.x * dxp + n0;                                // :144
                            w0 = div_exact(w0);                                        // :145
                            float w1 = ta.z * dxp + n1;                                // :147
                            w1 = div_exact(w1);                                        // :148
                            const float w2 = 1.0f - w0 - w1;                           // :149
                            const bool inside = 0.0f <= w0 && w0 <= 1.0f && 0.0f <= w1 && w1 <= 1.0f &&
                                                0.0f <= w2 && w2 <= 1.0f;              // :153
                            float d = w0 * tb.x + w1 * tb.y + w2 * tb.z;               // :157,:159
                            const bool live = qq < nvalid && inside && d < INFINITY;
                            d = d + 0.0f;
                            const unsigned long long key =
                                ((unsigned long long)orderable_depth(d) << 32) | (unsigned long long)oprim;
                            if constexpr (!K32) { if (live) atomicMin(&keys[lidx0 + qq], key); }
                        }
                    } else if (K32) {
                        // 32-bit keys: the depth is the key (ds_min_f32).  No -0 -> +0, no NaN / +inf clamp, no orderable map: a
                        // zero in the result sends the tile to the 64-bit path (k32_undecided, resolve)
                        float dx0 = (float)dxi;                              // (x + .5) - cf.x, exact: small integers
                        asm volatile("" : "+v"(dx0));
                        const float dy = (float)dyi;                         // (y + .5) - cf.y
                        const float r0 = ta.y * dy, r1 = ta.w * dy;          // t01*dy, t11*dy
                        // (Masking the pixels past the end of a span with a NaN offset — the LDS float minimum ignores quiet NaNs — instead
                        // of EXEC: one mask per visit, no compare / branch per pixel: the kernel alone 63.8 -> 61.4 us, the frame the same
                        // with or without it, 0.0702 / 0.0699 ms: profiles/r04/nanmask_ab.txt.  Not in the kernel.)
#pragma unroll
                        for (int qq = 0; qq < UNIT; qq++) {
                            const float dx = qq == 0 ? dx0 : dx0 + (float)qq;
                            const float w0 = ta.x * dx + r0;
                            const float w1 = ta.z * dx + r1;
                            const float w2 = 1.0f - w0 - w1;
                            const float d = tb.x * w0 + tb.y * w1 + tb.z * w2;     // :257
                            if constexpr (K32) { if (qq < nvalid) k32_min(&keys[lidx0 + qq], d); }
                        }
                    } else if (ZTEST) {
                        float dx0 = (float)dxi;                              // (x + .5) - cf.x, exact: small integers
                        asm volatile("" : "+v"(dx0));                        // keep dx0 + q a float add (2 cycles), not add + convert (6)
                        const float dy = (float)dyi;                         // (y + .5) - cf.y
                        const float r0 = ta.y * dy, r1 = ta.w * dy;          // t01*dy, t11*dy
#pragma unroll
                        for (int qq = 0; qq < UNIT; qq++) {
                            // exact: small integers.  (dx0 is never -0, so + 0.0f is the identity: spelled out, the compiler must keep the add)
                            const float dx = qq == 0 ? dx0 : dx0 + (float)qq;
                            const float w0 = ta.x * dx + r0;
                            const float w1 = ta.z * dx + r1;
                            const float w2 = 1.0f - w0 - w1;
                            float d = tb.x * w0 + tb.y * w1 + tb.z * w2;     // :257
                            // A NaN or +inf depth can never pass 'depth < buffer' (:258).  min(d, +inf) turns a NaN into
                            // +inf, and a key whose depth is +inf counts as "no fragment" at the resolve (KEY_LIVE_BELOW),
                            // so there is no per-pixel compare; d + 0 turns -0 into +0 (equal under '<').
                            d = fminf(d + 0.0f, INFINITY);
                            const unsigned long long key =
                                ((unsigned long long)orderable_depth(d) << 32) | (unsigned long long)oprim;
                            if (VAR == 1) { asm volatile("" ::"v"((uint32_t)key), "v"((uint32_t)(key >> 32))); continue; }
                            if constexpr (!K32) { if (qq < nvalid) atomicMin(&keys[lidx0 + qq], key); }
                        }
                    } else {
                        const unsigned long long key = (unsigned long long)(0xFFFFFFFFu - oprim);
#pragma unroll
                        for (int qq = 0; qq < UNIT; qq++)
                            if constexpr (!K32) { if (qq < nvalid) atomicMin(&keys[lidx0 + qq], key); }
                    }
                }   // sg
                // the rest of a span longer than this visit goes back into the ring (behind everything that waits)
                const bool more = (!TAIL || (lane >> sh) == 0) && (!TAIL || idx < n) && len > visit;
                const unsigned long long mb = __builtin_amdgcn_ballot_w64(more);
                if (more)
                    q[(qhead + qcount + rank_of(mb)) & (uint32_t)(QCAP - 1)] = e + ((uint32_t)visit << 6) - ((uint32_t)visit << 17);
                qhead += n;
                qcount = qcount - n + (uint32_t)__popcll(mb);
            };
            for (;;) {
                // (lane masks straight from v_cmp: hipcc turns a ballot inside this loop into v_cndmask + v_cmp_ne on top
                // of the compare; control flow is wave-uniform here, EXEC is all ones)
                unsigned long long am = lane_mask_ne(ebase, ebase_stop);     // lanes with rows left
                while (VAR != 4 && VAR != 10 && qcount < 64u && am != 0ull) {
                    int lo, hi;
                    if (METAL) {
                        // Shaders.metal:133-153 tests every pixel of the ROI row; only those between the two chains can pass.
                        // The samples of row y lie at y + 0.5, strictly between the integer rows y and y + 1, where each
                        // chain is ONE straight segment (vertices are integers).  Its truncated integer interpolant X(.)
                        // (the stepper) is within one pixel of the chain at y and at y + 1, the chain at y + 0.5 lies between
                        // the two, so a pixel whose centre x + 0.5 is inside has  min X - 1 <= x <= max X  over the four
                        // values; a pixel outside that range is at least half a pixel outside the triangle, far more than
                        // the float evaluation of the weights can err (~2^-8 px for extents below 2^15).
                        const int L0 = eL.X, R0 = eR.X;
                        EdgeStep nL = eL, nR = eR;
                        edge_next_row(nL);
                        edge_next_row(nR);
                        if (ebase + (1u << 12) == ebase_kink) nL = eK;
                        lo = max(min(min(L0, nL.X), min(R0, nR.X)) - 1, bxa);
                        hi = min(max(max(L0, nL.X), max(R0, nR.X)), bxb);
                        if (__builtin_amdgcn_inverse_ballot_w64(am)) { eL = nL; eR = nR; }
                    } else {
                        lo = max(min(eL.X, eR.X), X0);                       // :278-280 swap, then the tile's scissor
                        hi = min(max(eL.X, eR.X), X1);
                    }
                    // two single-compare ballots and a scalar AND: a ballot of (a && b) costs two more vector instructions
                    const unsigned long long pb = am & lane_mask_le(lo, hi);     // lanes with a non-empty span in this row
                    if (__builtin_amdgcn_inverse_ballot_w64(pb))
                        // (hi - lo) << 17 | lo << 6, + ebase (which holds -X0 << 6): lo * (2^6 - 2^17) as one 24-bit multiply-add (0 <= lo < 2^23)
                        q[(qhead + qcount + rank_of(pb)) & (uint32_t)(QCAP - 1)] =
                            (uint32_t)__mul24(lo, 64 - (1 << 17)) + (((uint32_t)hi << 17) + ebase);
                    if (__builtin_amdgcn_inverse_ballot_w64(am)) {
                        ebase += 1u << 12;
                        if (!METAL) {
                            edge_next_row(eL);
                            edge_next_row(eR);
                            if (ebase == ebase_kink) eL = eK;    // the kink: [S1,S2] starts at its first point
                        }
                    }
                    qcount += (uint32_t)__popcll(pb);
                    am = lane_mask_ne(ebase, ebase_stop);
                }
                if (SWR_STEAL_AHEAD && !stolen && am == 0ull) { stolen = true; steal_next(); }
                if (qcount == 0u) break;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (VAR == 3) { const uint32_t n = min(qcount, 64u); qhead += n; qcount -= n; continue; }
                if (qcount >= 64u) consume(std::false_type{});
                else consume(std::true_type{});
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            if (!stolen) steal_next();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the next chunk rewrites the tables
        };
        if constexpr (VAR == VAR_ALLCOOP) {
            steal_next();
        } else {
            // wide mode when at least half of the chunk's triangles that take the dense route are large for this tile
            const unsigned long long dm = __ballot(have && !big);
            const unsigned long long lg = __ballot(have && !big && large);
            if (dm != 0ull && 2 * __popcll(lg) >= __popcll(dm)) dense(std::integral_constant<int, 3>{});
            else dense(std::integral_constant<int, 0>{});
        }
        // the next chunk (wave-uniform, chosen by steal_next): its records, now that the bin entries have arrived
        chunk = chunk_next;
        if (have_next) {
            q0_pre = reinterpret_cast<const int4*>(a.geo + prim_pre)[0];
            q1_pre = reinterpret_cast<const float4*>(a.geo + prim_pre)[1];
        }
    }
    __syncthreads();

    // ---- resolve: key -> pixel, one coalesced write per pixel --------------------------------
    constexpr bool want_color = COLOR;          // a.color != nullptr; depth-only frames (SWR_FLAG_NO_COLOR) have their own kernels
    const int W = a.tg.width;
    const bool vec_ok = (W & 3) == 0;
    // 32-bit keys: the key is the depth (an untouched pixel still holds the +inf of the clear, :206); one 16-B LDS read and one
    // 16-B store per four pixels.  A zero of either sign (k32_undecided) is what the 64-bit path has to decide.
    if constexpr (K32) {
        bool undecided = false;
        for (int i = tid; i < TILE_W * TILE_H / 4; i += RASTER_THREADS) {
            const int ly = (i * 4) / TILE_W, lx = (i * 4) % TILE_W;
            const int y = Y0 + ly, x = X0 + lx;
            if (y < Yp0 || y > Yp1 || x > X1) continue;
            const float4 v = reinterpret_cast<const float4*>(L.keys)[i];
            undecided = undecided || k32_undecided(v.x) || k32_undecided(v.y) || k32_undecided(v.z) || k32_undecided(v.w);
            const size_t at = (size_t)(y - a.tg.row_begin) * (size_t)W + (size_t)x;   // App.swift:351-360
            if (vec_ok && x + 3 <= X1) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                f32x4 dv = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(dv, reinterpret_cast<f32x4*>(a.depth + at));
            } else {
                const float d4[4] = {v.x, v.y, v.z, v.w};
                for (int k = 0; k < 4 && x + k <= X1; k++) a.depth[at + k] = d4[k];
            }
        }
        if (undecided) L.redo = 1u;
        __syncthreads();
        return L.redo != 0u;
    } else {
    // ---- Colour frames, winner table.  A resolve thread that sets up the winner of its own pixels (the per-thread path below) runs
    // the winner setup — record gathers, vertex decode, the four exact divisions of T() — whenever ANY of its wave's 64 lanes meets
    // a new winner: at every one of a group's four pixels, 32 times per tile, for ~230 distinct winners (cfg4) or ~20 (cfg5).
    // Here the key's low word names the winner's position in the tile's bin, so
    //   1  every pixel keeps only that position (16 bits; the depth is recomputed from the winner the way the reference computes
    //      it, :257) — and, when the bin has more entries than records fit, the winners are marked in a bitmap over the bin and a
    //      prefix over the bitmap numbers them;
    //   2  one lane per bin entry (that owns a pixel) sets the winner up, ONCE, and leaves a record — T() or the Metal rules'
    //      constants, the three depths, the vertex colours (normals, texture coordinates) — where the keys were;
    //   3  every pixel shades from its winner's record.
    // No original-index -> slot gather, two gather round trips per tile (the first one in flight across step 1) instead of up to
    // nine, the setup runs ~6 times per tile instead of 32: cfg4 colour + depth 0.114 -> 0.101 ms (k_raster 57.6 -> 50.4 M vector
    // instructions), cfg5 0.168 -> 0.143, cfg5 textured 0.343 -> 0.284 (profiles/r04/winner_table_ab.txt).  A tile with more winners
    // than records fit (WTAB_RCAP) goes through steps 2 and 3 in rounds.
    if constexpr (WTAB_OK) {
        // record: [0..9] weights (CPU rules: T() 4, cf 2, z 3, - ; Metal rules: p3 2, A0 B0 A1 B1, divider, z 3), [10..18] colours a b c,
        // extended stage: [19..27] normals a b c, [28..33] (u, v) a b c.  An odd stride in words: neighbouring records in different banks.
        constexpr int REC_F = EXT ? 34 : 19;
        constexpr int REC_STRIDE = REC_F | 1;
        constexpr int WTAB_BYTES = (int)(sizeof(L.keys) + sizeof(L.tabAB) + sizeof(L.tabP) + sizeof(L.queue));   // contiguous, dead after the raster
        constexpr int PIX_OFF = WTAB_BYTES - TILE_W * TILE_H * 2;
        constexpr int WTAB_RCAP = PIX_OFF / (REC_STRIDE * 4);
        static_assert(offsetof(RasterLds64, queue) + sizeof(L.queue) == (size_t)WTAB_BYTES && offsetof(RasterLds64, keys) == 0, "records | positions alias keys .. queue");
        if (wtab) {
            uint16_t* const pix = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(&L) + PIX_OFF);
            float* const recs = reinterpret_cast<float*>(&L);
            const bool direct = m <= (uint32_t)WTAB_RCAP;        // every bin entry gets the record of its own position: no bitmap
            struct Gathered { int4 g0; float4 g1; float4 ca, cb, cc, na, nb, nc; };
            auto gather = [&](uint32_t slot) {
                Gathered G;
                G.g0 = reinterpret_cast<const int4*>(a.geo + slot)[0];
                G.g1 = reinterpret_cast<const float4*>(a.geo + slot)[1];
                G.na = G.nb = G.nc = make_float4(0, 0, 0, 0);
                if (EXT) {
                    G.ca = a.tri_rgb[3 * (size_t)slot + 0]; G.cb = a.tri_rgb[3 * (size_t)slot + 1]; G.cc = a.tri_rgb[3 * (size_t)slot + 2];
                    G.na = a.tri_nrm[3 * (size_t)slot + 0]; G.nb = a.tri_nrm[3 * (size_t)slot + 1]; G.nc = a.tri_nrm[3 * (size_t)slot + 2];
                } else {   // (lane 3 = the texture coordinate v: only the extended stage reads it)
                    const float* cp = reinterpret_cast<const float*>(a.tri_rgb + 3 * (size_t)slot);
                    G.ca = make_float4(cp[0], cp[1], cp[2], 0.0f); G.cb = make_float4(cp[4], cp[5], cp[6], 0.0f); G.cc = make_float4(cp[8], cp[9], cp[10], 0.0f);
                }
                return G;
            };
            auto build = [&](uint32_t rid, uint32_t slot, const Gathered& G) {
                int vx[3], vy[3];
                decode_vertices(a.geo_full, slot, G.g0, G.g1, vx, vy);
                float* r = recs + rid * REC_STRIDE;
                if (METAL) {
                    MetalTri mt;
                    metal_consts(vx, vy, G.g1.x, G.g1.y, G.g1.z, mt);
                    r[0] = mt.p3x; r[1] = mt.p3y; r[2] = mt.A0; r[3] = mt.B0; r[4] = mt.A1; r[5] = mt.B1; r[6] = mt.divider;
                } else {
                    float t00, t01, t10, t11;
                    tinv_of(vx[0], vy[0], vx[1], vy[1], vx[2], vy[2], t00, t01, t10, t11);
                    r[0] = t00; r[1] = t01; r[2] = t10; r[3] = t11; r[4] = (float)vx[2] + 0.5f; r[5] = (float)vy[2] + 0.5f; r[6] = 0.0f;
                }
                r[7] = G.g1.x; r[8] = G.g1.y; r[9] = G.g1.z;
                r[10] = G.ca.x; r[11] = G.ca.y; r[12] = G.ca.z; r[13] = G.cb.x; r[14] = G.cb.y; r[15] = G.cb.z; r[16] = G.cc.x; r[17] = G.cc.y; r[18] = G.cc.z;
                if (EXT) {
                    r[19] = G.na.x; r[20] = G.na.y; r[21] = G.na.z; r[22] = G.nb.x; r[23] = G.nb.y; r[24] = G.nb.z; r[25] = G.nc.x; r[26] = G.nc.y; r[27] = G.nc.z;
                    r[28] = G.na.w; r[29] = G.ca.w; r[30] = G.nb.w; r[31] = G.cb.w; r[32] = G.nc.w; r[33] = G.cc.w;
                }
            };
            // the first 256 bin entries' gathers go out now and land while the pixels are looked at
            uint32_t slot_first = (uint32_t)tid < m ? a.bins[b0 + (uint32_t)tid] & bin_mask : 0u;
            Gathered G_first = gather(slot_first);
            // 1: position of every pixel's winner (0xFFFF: none) ...
            uint32_t pos[2][4];
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const int p = (tid + g * RASTER_THREADS) * 4;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned long long key = keys[p + k];
                    const uint32_t low = ZTEST ? (uint32_t)key : 0xFFFFFFFFu - (uint32_t)key;
                    pos[g][k] = (uint32_t)(key >> 32) < KEY_LIVE_BELOW ? (low & ((1u << WTAB_LOCAL_BITS) - 1u)) : 0xFFFFu;
                }
                *reinterpret_cast<uint2*>(pix + p) = make_uint2(pos[g][0] | (pos[g][1] << 16), pos[g][2] | (pos[g][3] << 16));
            }
            __syncthreads();
            uint32_t nrec = m;
            if (!direct) {
                // ... and the winners marked in the bitmap — by the pixels whose left and upper neighbours belong to someone else only
                // (the top-left pixel of a winner's region always is one): a mark per pixel is 2 048 LDS atomics on a handful of words
                // (64 lanes of every instruction in turn on ONE word in a sparse tile: cfg5 0.169 -> 0.221 ms that way)
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    const int p = (tid + g * RASTER_THREADS) * 4;
                    const int lx = p % TILE_W;
                    uint32_t up[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu}, left = 0xFFFFu;
                    if (p >= TILE_W) {
                        const uint2 u = *reinterpret_cast<const uint2*>(pix + p - TILE_W);
                        up[0] = u.x & 0xFFFFu; up[1] = u.x >> 16; up[2] = u.y & 0xFFFFu; up[3] = u.y >> 16;
                    }
                    if (lx > 0) left = pix[p - 1];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t q = pos[g][k];
                        if (q != 0xFFFFu && q != (k ? pos[g][k - 1] : left) && q != up[k]) atomicOr(&L.winners[q >> 5].x, 1u << (q & 31u));
                    }
                }
                __syncthreads();
                // winners before every bitmap word — every wave for itself (same values, no barrier: a wave reads what it wrote)
                static_assert(WTAB_WORDS == 128, "two bitmap words per lane");
                const uint32_t bw0 = L.winners[2 * lane].x, bw1 = L.winners[2 * lane + 1].x;
                const uint32_t c0 = (uint32_t)__popc(bw0), c1 = (uint32_t)__popc(bw1);
                const uint32_t incl = (uint32_t)wave_incl_add((int)(c0 + c1));
                L.winners[2 * lane].y = incl - c0 - c1;
                L.winners[2 * lane + 1].y = incl - c1;
                nrec = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
            auto rid_of = [&](uint32_t q) {         // record of bin position q
                if (direct) return q;
                const uint2 wv = L.winners[q >> 5];
                return wv.y + (uint32_t)__popc(wv.x & ((1u << (q & 31u)) - 1u));
            };
            // one pixel from record rid
            auto shade = [&](uint32_t rid, int x, int y, uint32_t& c, float& d) {
                const float* r = recs + rid * REC_STRIDE;
                float v[REC_F];
#pragma unroll
                for (int j = 0; j < REC_F; j++) v[j] = r[j];
                float w0, w1, w2;
                if (METAL) {
                    MetalTri mt;
                    mt.p3x = v[0]; mt.p3y = v[1]; mt.A0 = v[2]; mt.B0 = v[3]; mt.A1 = v[4]; mt.B1 = v[5]; mt.divider = v[6];
                    mt.z0 = v[7]; mt.z1 = v[8]; mt.z2 = v[9];
                    metal_weights_shared_rcp(mt, x, y, w0, w1, w2);           // Shaders.metal:133-149
                } else {
                    const float dx = ((float)x + 0.5f) - v[4];
                    const float dy = ((float)y + 0.5f) - v[5];
                    w0 = v[0] * dx + v[1] * dy;
                    w1 = v[2] * dx + v[3] * dy;
                    w2 = 1.0f - w0 - w1;
                }
                d = INFINITY;                                                 // (:206; painter's order leaves the depth image alone)
                if (ZTEST) d = v[7] * w0 + v[8] * w1 + v[9] * w2;              // :257
                VertexOut vin;
                vin.pos = make_float4((float)x + 0.5f, (float)y + 0.5f, d, 1.0f);
                vin.color = make_float3(v[10] * w0 + v[13] * w1 + v[16] * w2,     // :266
                                        v[11] * w0 + v[14] * w1 + v[17] * w2,
                                        v[12] * w0 + v[15] * w1 + v[18] * w2);
                float4 f;
                if constexpr (EXT) {   // varyings interpolated like colour
                    vin.normal = make_float3(v[19] * w0 + v[22] * w1 + v[25] * w2,
                                             v[20] * w0 + v[23] * w1 + v[26] * w2,
                                             v[21] * w0 + v[24] * w1 + v[27] * w2);
                    vin.uv = make_float2(v[28] * w0 + v[30] * w1 + v[32] * w2,
                                         v[29] * w0 + v[31] * w1 + v[33] * w2);
                    f = fragment_shader(vin, a.fs);
                } else {
                    f = fragment_shader(vin);
                }
                // Pixel(float3:) -> .floats(b: z, g: y, r: x, a: 1) truncates (:116-128); the Metal path's bgra8Unorm store rounds to nearest even
                float ub = fminf(fmaxf(f.z, 0.0f), 1.0f) * 255.0f, ug = fminf(fmaxf(f.y, 0.0f), 1.0f) * 255.0f;
                float ur = fminf(fmaxf(f.x, 0.0f), 1.0f) * 255.0f, ua = fminf(fmaxf(f.w, 0.0f), 1.0f) * 255.0f;
                if (METAL) { ub = rintf(ub); ug = rintf(ug); ur = rintf(ur); ua = rintf(ua); }
                c = (uint32_t)ub | ((uint32_t)ug << 8) | ((uint32_t)ur << 16) | ((uint32_t)ua << 24);
            };
            // Passes 2 and 3, in ROUNDS of WTAB_RCAP winners: one round for every tile of the BASELINE scenes; a tile of ~1 000 two-pixel
            // triangles has ~600 winners and takes two.  A pixel is shaded and stored by the round its winner's number falls into
            // (pixels without a winner by the first); a tile of one round stores four pixels at a time.
            const bool one_round = nrec <= (uint32_t)WTAB_RCAP;
            for (uint32_t base = 0u; base < max(nrec, 1u); base += (uint32_t)WTAB_RCAP) {
                if (base) __syncthreads();       // the round before has read its records
                // 2: one lane per bin entry (that owns a pixel): its record
                for (uint32_t i = (uint32_t)tid; i < m; i += RASTER_THREADS) {
                    const bool mine = direct || ((L.winners[i >> 5].x >> (i & 31u)) & 1u);
                    const uint32_t rid = mine ? rid_of(i) - base : 0xFFFFFFFFu;
                    if (rid >= (uint32_t)WTAB_RCAP) continue;
                    if (i >= (uint32_t)RASTER_THREADS || base) {      // (else: gathered before pass 1)
                        slot_first = a.bins[b0 + i] & bin_mask;
                        G_first = gather(slot_first);
                    }
                    build(rid, slot_first, G_first);
                }
                __syncthreads();
                // 3: every pixel from its winner's record
#pragma unroll 1
                for (int g = 0; g < 2; g++) {
                    const int i = tid + g * RASTER_THREADS;
                    const int ly = (i * 4) / TILE_W, lx = (i * 4) % TILE_W;
                    const int y = Y0 + ly, x = X0 + lx;
                    if (y < Yp0 || y > Yp1 || x > X1) continue;
                    const uint2 pp = *reinterpret_cast<const uint2*>(pix + i * 4);
                    const uint32_t ps[4] = {pp.x & 0xFFFFu, pp.x >> 16, pp.y & 0xFFFFu, pp.y >> 16};
                    uint32_t cpix[4];
                    float dpix[4];
                    uint32_t stored = 0u;              // (rounds) which of the four pixels this round stores
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        cpix[k] = 0u;                  // Pixel(0,0,0,0) (:205)
                        dpix[k] = INFINITY;            // (:206)
                        if (x + k > X1) continue;
                        if (ps[k] == 0xFFFFu) { if (base == 0u) stored |= 1u << k; continue; }
                        const uint32_t rid = rid_of(ps[k]) - base;
                        if (rid < (uint32_t)WTAB_RCAP) { shade(rid, x + k, y, cpix[k], dpix[k]); stored |= 1u << k; }
                    }
                    const size_t at = (size_t)(y - a.tg.row_begin) * (size_t)W + (size_t)x;   // App.swift:351-360
                    if (one_round && vec_ok && x + 3 <= X1) {
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        typedef float f32x4 __attribute__((ext_vector_type(4)));
                        u32x4 cv = {cpix[0], cpix[1], cpix[2], cpix[3]};
                        __builtin_nontemporal_store(cv, reinterpret_cast<u32x4*>(a.color + at * 4));
                        f32x4 dv = {dpix[0], dpix[1], dpix[2], dpix[3]};
                        __builtin_nontemporal_store(dv, reinterpret_cast<f32x4*>(a.depth + at));
                    } else {
                        // (several rounds, or the ragged right edge: pixel by pixel — a round stores the pixels whose winner it holds, the first
                        // one also those without a winner)
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (!((stored >> k) & 1u)) continue;
                            reinterpret_cast<uint32_t*>(a.color)[at + k] = cpix[k];
                            a.depth[at + k] = dpix[k];
                        }
                    }
                }
            }
            return false;
        }
    }
    // Colour frames: the stream slots of ALL this thread's winners first (original index -> slot is a gather from a
    // 4 MB table; the keys hold the original index because it decides depth ties) — 8 independent loads in flight
    // instead of 4 + 4 behind each other — written back into the low words of the keys, which have done their job.
    static_assert(TILE_W * TILE_H / 4 == 2 * RASTER_THREADS, "two 4-pixel groups per thread");
    // (reached by colour frames only when the keys carry no bin position: a bin of more than 4 096 entries, or the PLAIN kernels)
    if (want_color && a.reordered && VAR != 8 && VAR != 10 && VAR != 11) {
        uint32_t sl[8];
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const int p = (tid + (g >> 2) * RASTER_THREADS) * 4 + (g & 3);
            const unsigned long long key = keys[p];
            const uint32_t prim = ZTEST ? (uint32_t)key : 0xFFFFFFFFu - (uint32_t)key;
            sl[g] = (uint32_t)(key >> 32) < KEY_LIVE_BELOW ? a.inv[prim] : 0u;
        }
#pragma unroll
        for (int g = 0; g < 8; g++) slots[(tid + (g >> 2) * RASTER_THREADS) * 4 + (g & 3)] = sl[g];
    }
    // Depth-only z-tested frames: the depth IS the key's high word (min() keys: empty / +inf / NaN keys all decode to +inf), so
    // the resolve is a decode and a store.  One exception: a depth of exactly zero — the key holds +0, the reference stores the
    // sign its arithmetic produced (:257) — sends the wave through the general path below, which re-evaluates the winner.
    if (ZTEST && !COLOR && (VAR == 0 || VAR == VAR_ALLCOOP)) {
        bool zero_seen = false;
        for (int i = tid; i < TILE_W * TILE_H / 4; i += RASTER_THREADS) {
            const int ly = (i * 4) / TILE_W, lx = (i * 4) % TILE_W;
            const int y = Y0 + ly, x = X0 + lx;
            if (y < Yp0 || y > Yp1 || x > X1) continue;
            float d4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t hi = (uint32_t)(keys[ly * TILE_W + lx + k] >> 32);
                d4[k] = depth_from_orderable(min(hi, KEY_LIVE_BELOW));
                zero_seen = zero_seen || (d4[k] == 0.0f);
            }
            const size_t at = (size_t)(y - a.tg.row_begin) * (size_t)W + (size_t)x;   // App.swift:351-360
            if (vec_ok && x + 3 <= X1) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                f32x4 dv = {d4[0], d4[1], d4[2], d4[3]};
                __builtin_nontemporal_store(dv, reinterpret_cast<f32x4*>(a.depth + at));
            } else {
                for (int k = 0; k < 4 && x + k <= X1; k++) a.depth[at + k] = d4[k];
            }
        }
        if (!__any(zero_seen)) return false;    // (wave-uniform; no barrier follows)
    }
    // The per-thread resolve: depth-only frames on the 64-bit keys (the Metal rules; scenes the host moved off the 32-bit keys;
    // only the rare d == 0 winner is looked up), and the colour tiles the winner table above could not take.
    // Depth-only: a thread's two 4-pixel groups are resolved TOGETHER, pixel by pixel: the gathers of a new winner in group 0 and
    // in group 1 go out back to back and are waited for once (profiles/r03/resolve_joint_ab.txt).  Colour (cold path): one group
    // at a time — the smaller register footprint.  (Rounds 2-3 chose per kernel, and the launch chose a joint walk for sparse
    // colour scenes, template parameter NGX: all of that was tuning of what is now the cold path.)
#ifndef SWR_NG_DEPTH
#define SWR_NG_DEPTH 2
#endif
    constexpr int NG = !COLOR ? SWR_NG_DEPTH : ((PLAIN && METAL && !EXT) ? 2 : 1);
    static_assert(NG == 1 || NG == 2, "a thread owns two groups");
    for (int i0 = tid; VAR != 8 && VAR != 10 && VAR != 11 && i0 < TILE_W * TILE_H / 4; i0 += NG * RASTER_THREADS) {
        int ly[NG], lx[NG], y[NG], x[NG];
        bool on[NG];
        uint32_t cpix[NG][4];
        float dpix[NG][4];
        // neighbouring pixels usually share the winning primitive: its record, T() and vertex
        // colours are fetched / computed once per run of equal primitives
        uint32_t cached_prim[NG], slot[NG];
        float4 q2[NG], q3[NG], ca[NG], cb[NG], cc[NG], na[NG], nb[NG], nc[NG];
        int4 g0[NG];
        float cfx[NG], cfy[NG];
        MetalTri mt[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int i = i0 + g * RASTER_THREADS;
            ly[g] = (i * 4) / TILE_W; lx[g] = (i * 4) % TILE_W;
            y[g] = Y0 + ly[g]; x[g] = X0 + lx[g];
            on[g] = !(y[g] < Yp0 || y[g] > Yp1 || x[g] > X1);
            cached_prim[g] = 0xFFFFFFFFu; slot[g] = 0u;
            q2[g] = q3[g] = ca[g] = cb[g] = cc[g] = na[g] = nb[g] = nc[g] = make_float4(0, 0, 0, 0);
            g0[g] = make_int4(0, 0, 0, 0);
            cfx[g] = cfy[g] = 0.0f;
            mt[g] = MetalTri{};
        }
        // (colour kernels reach this loop only for tiles the winner table could not take: kept small — one pixel at a time, pixels
        // stored one by one — so that this cold path does not set the kernel's register count)
        constexpr bool COLD = WTAB_OK;
#pragma unroll(COLD ? 1 : 4)
        for (int k = 0; k < 4; k++) {
            uint32_t prim[NG];
            float d[NG];
            bool need_rec[NG], miss[NG];
            // which groups meet a new winner at this pixel; their gathers go out together
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const unsigned long long key = keys[ly[g] * TILE_W + lx[g] + k];
                const uint32_t hi = (uint32_t)(key >> 32);
                const bool live = on[g] && hi < KEY_LIVE_BELOW && x[g] + k <= X1;
                prim[g] = ZTEST ? (uint32_t)key : 0xFFFFFFFFu - (uint32_t)key;
                d[g] = INFINITY;           // (:206)
                need_rec[g] = live && want_color;
                if (ZTEST && live) {
                    d[g] = depth_from_orderable(hi);
                    need_rec[g] = need_rec[g] || (d[g] == 0.0f);   // sign of zero comes from the winner
                }
                miss[g] = need_rec[g] && prim[g] != cached_prim[g];
            }
#pragma unroll
            for (int g = 0; g < NG; g++) {
                if (miss[g]) {
                    cached_prim[g] = prim[g];
                    // colour frames hold the winners' stream slots in LDS (above); depth-only: only the rare d == 0 winner
                    slot[g] = !a.reordered ? prim[g] : (want_color ? slots[ly[g] * TILE_W + lx[g] + k] : a.inv[prim[g]]);
                    g0[g] = reinterpret_cast<const int4*>(a.geo + slot[g])[0];
                    q3[g] = reinterpret_cast<const float4*>(a.geo + slot[g])[1];
                    if (want_color) {
                        // vertex colours of a,b,c (RenderPass.vertices[RenderPass.indices[3p+k]].color), one 48-B record
                        if (EXT) {
                            ca[g] = a.tri_rgb[3 * (size_t)slot[g] + 0];
                            cb[g] = a.tri_rgb[3 * (size_t)slot[g] + 1];
                            cc[g] = a.tri_rgb[3 * (size_t)slot[g] + 2];
                        } else {   // (lane 3 = the texture coordinate v: only the extended stage reads it — nine registers in flight, not twelve)
                            const float* cp = reinterpret_cast<const float*>(a.tri_rgb + 3 * (size_t)slot[g]);
                            ca[g] = make_float4(cp[0], cp[1], cp[2], 0.0f);
                            cb[g] = make_float4(cp[4], cp[5], cp[6], 0.0f);
                            cc[g] = make_float4(cp[8], cp[9], cp[10], 0.0f);
                        }
                        if (EXT) {
                            na[g] = a.tri_nrm[3 * (size_t)slot[g] + 0];
                            nb[g] = a.tri_nrm[3 * (size_t)slot[g] + 1];
                            nc[g] = a.tri_nrm[3 * (size_t)slot[g] + 2];
                        }
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < NG; g++) {
                if (miss[g]) {
                    int vx[3], vy[3];
                    decode_vertices(a.geo_full, slot[g], g0[g], q3[g], vx, vy);
                    if (METAL) {
                        metal_consts(vx, vy, q3[g].x, q3[g].y, q3[g].z, mt[g]);
                    } else {
                        // T() of the winning primitive, recomputed (same function, same bits)
                        tinv_of(vx[0], vy[0], vx[1], vy[1], vx[2], vy[2], q2[g].x, q2[g].y, q2[g].z, q2[g].w);
                        cfx[g] = (float)vx[2] + 0.5f;
                        cfy[g] = (float)vy[2] + 0.5f;
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < NG; g++) {
                uint32_t c = 0u;               // Pixel(0,0,0,0) (:205)
                if (need_rec[g]) {
                    float w0, w1, w2;
                    if (METAL) {
                        metal_weights(mt[g], x[g] + k, y[g], w0, w1, w2);          // Shaders.metal:133-149
                    } else {
                        const float dx = ((float)(x[g] + k) + 0.5f) - cfx[g];
                        const float dy = ((float)y[g] + 0.5f) - cfy[g];
                        w0 = q2[g].x * dx + q2[g].y * dy;
                        w1 = q2[g].z * dx + q2[g].w * dy;
                        w2 = 1.0f - w0 - w1;
                    }
                    if (ZTEST) d[g] = q3[g].x * w0 + q3[g].y * w1 + q3[g].z * w2;
                    if (want_color) {
                        VertexOut vin;
                        vin.pos = make_float4((float)(x[g] + k) + 0.5f, (float)y[g] + 0.5f, d[g], 1.0f);
                        vin.color = make_float3(ca[g].x * w0 + cb[g].x * w1 + cc[g].x * w2,     // :266
                                                ca[g].y * w0 + cb[g].y * w1 + cc[g].y * w2,
                                                ca[g].z * w0 + cb[g].z * w1 + cc[g].z * w2);
                        float4 f;
                        if (EXT) {   // varyings interpolated like colour; u = tri_nrm.w, v = tri_rgb.w
                            vin.normal = make_float3(na[g].x * w0 + nb[g].x * w1 + nc[g].x * w2,
                                                     na[g].y * w0 + nb[g].y * w1 + nc[g].y * w2,
                                                     na[g].z * w0 + nb[g].z * w1 + nc[g].z * w2);
                            vin.uv = make_float2(na[g].w * w0 + nb[g].w * w1 + nc[g].w * w2,
                                                 ca[g].w * w0 + cb[g].w * w1 + cc[g].w * w2);
                            f = fragment_shader(vin, a.fs);
                        } else {
                            f = fragment_shader(vin);
                        }
                        // Pixel(float3:) -> .floats(b: z, g: y, r: x, a: 1) truncates (:116-128);
                        // the Metal path's bgra8Unorm store rounds to nearest even
                        float ub = fminf(fmaxf(f.z, 0.0f), 1.0f) * 255.0f, ug = fminf(fmaxf(f.y, 0.0f), 1.0f) * 255.0f;
                        float ur = fminf(fmaxf(f.x, 0.0f), 1.0f) * 255.0f, ua = fminf(fmaxf(f.w, 0.0f), 1.0f) * 255.0f;
                        if (METAL) { ub = rintf(ub); ug = rintf(ug); ur = rintf(ur); ua = rintf(ua); }
                        c = (uint32_t)ub | ((uint32_t)ug << 8) | ((uint32_t)ur << 16) | ((uint32_t)ua << 24);
                    }
                }
                if constexpr (COLD) {
                    if (on[g] && x[g] + k <= X1) {
                        const size_t at1 = (size_t)(y[g] - a.tg.row_begin) * (size_t)W + (size_t)(x[g] + k);
                        reinterpret_cast<uint32_t*>(a.color)[at1] = c;
                        a.depth[at1] = d[g];
                    }
                } else {
                    cpix[g][k] = c;
                    dpix[g][k] = d[g];
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NG; g++) {
            if (COLD || !on[g]) continue;
            const size_t at = (size_t)(y[g] - a.tg.row_begin) * (size_t)W + (size_t)x[g];   // App.swift:351-360
            if (vec_ok && x[g] + 3 <= X1) {
                // streaming stores: nothing on the GPU reads the framebuffer again, and 33 MB of dirty lines left in the
                // L2s would be written back at the end of the kernel, in front of the next one
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                if (want_color) {
                    u32x4 cv = {cpix[g][0], cpix[g][1], cpix[g][2], cpix[g][3]};
                    __builtin_nontemporal_store(cv, reinterpret_cast<u32x4*>(a.color + at * 4));
                }
                f32x4 dv = {dpix[g][0], dpix[g][1], dpix[g][2], dpix[g][3]};
                __builtin_nontemporal_store(dv, reinterpret_cast<f32x4*>(a.depth + at));
            } else {
                for (int k = 0; k < 4 && x[g] + k <= X1; k++) {
                    if (want_color) reinterpret_cast<uint32_t*>(a.color)[at + k] = cpix[g][k];
                    a.depth[at + k] = dpix[g][k];
                }
            }
        }
    }
    }   // !K32
    return false;
}

// The kernels proper: the reference's fragment stage — colour and depth-only frames (SWR_FLAG_NO_COLOR) as separate kernels, so
// that each has its own register allocation (tools/vgprs.sh: 86 depth-only, 87 colour, 88 Metal rules; the budget of 88 above) —
// and the extended one.  PLAIN = the colour kernels of scenes with more than 2^20 primitives (no winner table: raster_tile).
template <bool ZTEST, int VAR = 0, bool METAL = false, bool COLOR = false, bool PLAIN = false>
__global__ __launch_bounds__(RASTER_THREADS, SWR_RASTER_MIN_WAVES) __attribute__((amdgpu_num_vgpr(SWR_RASTER_VGPRS)))
void k_raster(RasterArgs a) {
    __shared__ RasterLds64 L;
    raster_tile<ZTEST, VAR, METAL, false, COLOR, PLAIN>(a, L);
}
template <bool ZTEST, bool METAL = false, bool PLAIN = false>
__global__ __launch_bounds__(RASTER_THREADS, PLAIN ? 4 : SWR_RASTER_MIN_WAVES_EXT)
void k_raster_ext(RasterArgs a) {
    __shared__ RasterLds64 L;
    raster_tile<ZTEST, 0, METAL, true, true, PLAIN>(a, L);
}
// Depth-only z-tested frames under the CPU rules: 32-bit keys first; the rare tile whose result they cannot vouch for (a
// zero, whose sign is the first-drawn winner's) is rastered again, by the same workgroup, with the 64-bit keys.  One LDS block for both.
__global__ __launch_bounds__(RASTER_THREADS, SWR_RASTER_MIN_WAVES) __attribute__((amdgpu_num_vgpr(SWR_RASTER_VGPRS)))
void k_raster_depth(RasterArgs a) {
    constexpr size_t BYTES = sizeof(RasterLds64) > sizeof(RasterLds32) ? sizeof(RasterLds64) : sizeof(RasterLds32);
    __shared__ __attribute__((aligned(16))) unsigned char raw[BYTES];
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.host_redo = atomicExch(a.redo_dev, 0u);     // what the launches before this one counted
    if (raster_tile<true, 0, false, false, false, false, true>(a, *reinterpret_cast<RasterLds32*>(raw))) {
        if (threadIdx.x == 0 && blockIdx.x % REDO_SAMPLE == 0) atomicAdd(a.redo_dev, 1u);
        raster_tile<true, VAR_ALLCOOP, false, false, false, false, false>(a, *reinterpret_cast<RasterLds64*>(raw));
    }
}

// ------------------------------------------------------------------------------------------
// PrimitiveType .vertices (Renderer.swift:295-302) and .line (empty stub, :289-293)
// ------------------------------------------------------------------------------------------
// .vertices plots every transformed vertex reference at (Int(sx), Int(sy)) with its own colour, in
// index order, no z.  "Later overwrites" = the highest index wins, so an atomicMax of (index + 1)
// per pixel reproduces the serial loop; the band's depth buffer doubles as that u32 scratch and is
// reset to +inf by the resolve (the reference leaves depth at its cleared value).
__global__ void k_clear_band(uint32_t* __restrict__ color, uint32_t* __restrict__ depth_bits, int64_t n,
                             uint32_t depth_value) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (color) color[i] = 0u;                       // Pixel(0,0,0,0) (:205)
        depth_bits[i] = depth_value;
    }
}

__global__ void k_points(const swr_vertex* __restrict__ vtx, const int64_t* __restrict__ idx, int64_t ni,
                         float4x4 m, Target tg, uint32_t* __restrict__ order) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ni) return;
    const float4 p = reinterpret_cast<const float4*>(vtx)[2 * idx[i]];
    const VertexOut vo = vertex_shader(make_float3(p.x, p.y, p.z), make_float3(0, 0, 0), m);   // :160
    const float nx = vo.pos.x / vo.pos.w, ny = vo.pos.y / vo.pos.w;                             // :161
    const float sx = (nx * 0.5f + 0.5f) * (float)tg.width;                                      // :166-168
    const float sy = (ny * -0.5f + 0.5f) * (float)tg.height;
    if (!(fabsf(sx) < COORD_LIMIT) || !(fabsf(sy) < COORD_LIMIT)) return;   // Swift Int(NaN) would trap
    const int px = (int)sx, py = (int)sy;                                    // :298-299 truncation
    if (px < 0 || px >= tg.width || py < tg.row_begin || py >= tg.row_end) return;   // setter drops OOB (:30-36)
    atomicMax(&order[(size_t)(py - tg.row_begin) * (size_t)tg.width + (size_t)px], (uint32_t)(i + 1));
}

// PrimitiveType .line under SWR_FLAG_REAL_LINES (opt-in; the default is the reference's empty stub): the reference's own
// line DDA, draw(line:with:in:) (Renderer.swift:405-419), between the two transformed endpoints of every 2-index primitive,
// truncated like draw(vertices:) does (:298-299).  x and y are ACCUMULATED floats (x += xStep, :416-417), so the pixels of a
// line come out of a sequential loop: one lane per line.  Later lines overwrite earlier ones: atomicMax of (index of the
// line's first vertex reference + 1), resolved by k_points_resolve like .vertices — the colour is the first vertex's.
constexpr int LINE_MAX_STEPS = 1 << 20;     // longer lines are skipped (include/swr.h, SWR_FLAG_REAL_LINES)
__global__ void k_lines(const swr_vertex* __restrict__ vtx, const int64_t* __restrict__ idx, int64_t nlines,
                        float4x4 m, Target tg, uint32_t* __restrict__ order) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlines) return;
    int ex[2], ey[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float4 p = reinterpret_cast<const float4*>(vtx)[2 * idx[2 * l + k]];
        const VertexOut vo = vertex_shader(make_float3(p.x, p.y, p.z), make_float3(0, 0, 0), m);   // :160
        const float nx = vo.pos.x / vo.pos.w, ny = vo.pos.y / vo.pos.w;                             // :161
        const float sx = (nx * 0.5f + 0.5f) * (float)tg.width;                                      // :166-168
        const float sy = (ny * -0.5f + 0.5f) * (float)tg.height;
        if (!(fabsf(sx) < COORD_LIMIT) || !(fabsf(sy) < COORD_LIMIT)) return;   // Swift Int(NaN) would trap
        ex[k] = (int)sx; ey[k] = (int)sy;                                        // :298-299 truncation
    }
    const int dx = ex[1] - ex[0], dy = ey[1] - ey[0];                            // :406-407 (|coordinates| < 2^30: no overflow)
    const int steps = max(abs(dx), abs(dy));                                     // :408
    if (steps > LINE_MAX_STEPS) return;
    const float xs = (float)dx / (float)steps, ys = (float)dy / (float)steps;    // :409-410
    float x = (float)ex[0], y = (float)ey[0];                                    // :412-413
    const uint32_t tag = (uint32_t)(2 * l + 1);
    for (int k = 0; k < steps; k++) {                                            // :414
        const int px = (int)roundf(x), py = (int)roundf(y);                      // :415 rounded(): half away from zero
        if (px >= 0 && px < tg.width && py >= tg.row_begin && py < tg.row_end)   // the setter drops OOB (:30-36)
            atomicMax(&order[(size_t)(py - tg.row_begin) * (size_t)tg.width + (size_t)px], tag);
        x += xs;                                                                 // :416-417
        y += ys;
    }
}

__global__ void k_points_resolve(const swr_vertex* __restrict__ vtx, const int64_t* __restrict__ idx,
                                 uint32_t* __restrict__ color, uint32_t* __restrict__ depth_bits, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t o = depth_bits[i];
        uint32_t c = 0u;
        if (o) {
            const float4 v = reinterpret_cast<const float4*>(vtx)[2 * idx[o - 1] + 1];
            VertexOut vin;
            vin.pos = make_float4(0, 0, 0, 1);
            vin.color = make_float3(v.x, v.y, v.z);
            const float4 f = fragment_shader(vin);                            // Pixel(float3:) :126-128
            const uint32_t qb = (uint32_t)(fminf(fmaxf(f.z, 0.0f), 1.0f) * 255.0f);
            const uint32_t qg = (uint32_t)(fminf(fmaxf(f.y, 0.0f), 1.0f) * 255.0f);
            const uint32_t qr = (uint32_t)(fminf(fmaxf(f.x, 0.0f), 1.0f) * 255.0f);
            const uint32_t qa = (uint32_t)(fminf(fmaxf(f.w, 0.0f), 1.0f) * 255.0f);
            c = qb | (qg << 8) | (qr << 16) | (qa << 24);
        }
        color[i] = c;
        depth_bits[i] = 0x7F800000u;                                          // +inf (:206)
    }
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
void launch_points_or_lines(const DeviceFrame& f, int primitive_type, hipStream_t s) {
    const int64_t n = (int64_t)f.tg.width * (int64_t)(f.tg.row_end - f.tg.row_begin);
    if (n <= 0) return;
    const bool want_color = !(f.flags & SWR_FLAG_NO_COLOR);
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    const bool points = primitive_type == SWR_PRIMITIVE_VERTICES && want_color && f.index_count >= 3;
    const bool lines = primitive_type == SWR_PRIMITIVE_LINE && (f.flags & SWR_FLAG_REAL_LINES) && want_color && f.index_count >= 2;
    // .line (as written: an empty stub) and colour-less passes only clear; .vertices and real lines first zero the order
    // scratch (= depth bits)
    hipLaunchKernelGGL(k_clear_band, dim3(blocks), dim3(256), 0, s, want_color ? (uint32_t*)f.color : nullptr,
                       (uint32_t*)f.depth, n, (points || lines) ? 0u : 0x7F800000u);
    if (!points && !lines) return;
    const int64_t ni = f.index_count;
    float4x4 m;
    for (int c = 0; c < 4; c++)
        m.columns[c] = make_float4(f.m[4 * c + 0], f.m[4 * c + 1], f.m[4 * c + 2], f.m[4 * c + 3]);
    if (lines)
        hipLaunchKernelGGL(k_lines, dim3((unsigned)((ni / 2 + 63) / 64)), dim3(64), 0, s, f.vertices, f.indices, ni / 2, m, f.tg,
                           (uint32_t*)f.depth);
    else
        hipLaunchKernelGGL(k_points, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, s, f.vertices, f.indices, ni, m, f.tg,
                           (uint32_t*)f.depth);
    hipLaunchKernelGGL(k_points_resolve, dim3(blocks), dim3(256), 0, s, f.vertices, f.indices, (uint32_t*)f.color,
                       (uint32_t*)f.depth, n);
}

void launch_validate_indices(const int64_t* indices, int64_t count, int64_t vertex_count,
                             uint32_t* counters, hipStream_t s) {
    if (count <= 0) return;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_validate_indices, dim3((unsigned)blocks), dim3(256), 0, s, indices, count,
                       vertex_count, counters);
}

static SetupArgs make_setup_args(const DeviceFrame& f) {
    SetupArgs a;
    a.tri_xyz = f.tri_xyz; a.box64 = f.box64; a.reordered = f.reordered; a.ntri = f.ntri;
    a.cull = f.live_parity >= 0 ? 1 : 0;
    a.geo = f.geo; a.geo_full = f.geo_full;
    a.tile_count = f.tile_count; a.ranges = f.ranges; a.tg = f.tg;
    a.metal = (f.flags & SWR_FLAG_METAL_RULES) ? 1 : 0;
    for (int c = 0; c < 4; c++)
        a.m.columns[c] = make_float4(f.m[4 * c + 0], f.m[4 * c + 1], f.m[4 * c + 2], f.m[4 * c + 3]);
    return a;
}

// Stream groups owned by one binning workgroup (k_setup_hist / k_fill_lds): ceil(groups / G).
int live_groups_per_workgroup(int64_t ntri, int G) {
    const int64_t groups = (ntri + 63) / 64;
    return (int)((groups + G - 1) / (G > 0 ? G : 1));
}

// LDS binning geometry: G workgroups of BIN_THREADS threads, each owning `chunk` consecutive primitives.
BinPlan plan_binning(int64_t ntri, int ntiles, bool force_atomic) {
    BinPlan p{};
    p.lds_bytes = (size_t)ntiles * 4;
    p.use_lds = p.lds_bytes <= 136 * 1024 && !force_atomic;   // (SWR_DEBUG_BIN_MODE = 3 forces the fallback; the LDS also holds a workgroup's group list)
    p.threads = 256;
    int64_t g = (ntri + p.threads - 1) / p.threads;
    if (g > SWR_TUNE_BIN_G) g = SWR_TUNE_BIN_G;   // 1 per CU (measured best)
    if (g > MAX_BIN_G) g = MAX_BIN_G;
    if (g < 1) g = 1;
    p.G = (int)g;
    p.chunk = (int)((ntri + g - 1) / g);                      // (informational: the kernels walk their own group lists)
    if (p.chunk < 1) p.chunk = 1;
    // the workgroup's list of surviving stream groups shares the LDS with its tile histogram
    if (p.use_lds && p.lds_bytes + (size_t)(live_groups_per_workgroup(ntri, p.G) + 1) * 4 > 150 * 1024) p.use_lds = false;
    return p;
}

void launch_texture_to_float(const uint32_t* bgra, int64_t n, float4* out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_texture_to_float, dim3(2048), dim3(256), 0, s, bgra, n, out);
}

// The LDS binning kernels may ask for the whole 160 KB of a CU (8K-class tile tables).  The attribute is per device
// (per loaded code object): swr_context_create calls this after hipSetDevice for every device it opens.
hipError_t prepare_device() {
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void*)k_setup_hist<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_setup_hist<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
#define SWR_BIN_ATTR(MT, DF, AF) \
    if ((e = hipFuncSetAttribute((const void*)k_bin<256, MT, DF, AF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    SWR_BIN_ATTR(false, false, false) SWR_BIN_ATTR(true, false, false) SWR_BIN_ATTR(false, true, false) SWR_BIN_ATTR(true, true, false)
    SWR_BIN_ATTR(false, false, true) SWR_BIN_ATTR(true, false, true) SWR_BIN_ATTR(false, true, true) SWR_BIN_ATTR(true, true, true)
#undef SWR_BIN_ATTR
    return hipFuncSetAttribute((const void*)k_fill_lds<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// Launch with the completion of the kernel bound to `stop` (hipExtLaunchKernelGGL: the event is the kernel's own
// completion signal — no marker packet behind the kernel, which would cost the next kernel of the queue ~6.5 us).
#define SWR_LAUNCH(stop, kernel, grid, block, lds, stream, ...)                                         \
    do {                                                                                              \
        if (stop) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                        \
    } while (0)

// 16-bit LDS counters in the two binning walks: a workgroup's count for one tile is at most the primitives it owns
// (k_fill_lds with 16-bit running counts in LDS and its bin positions gathered from its row of M was built too: alone
// 23.2 -> 33.9 us, and with BOTH walks co-resident with the raster the frame went 0.096 -> 0.111 ms although k_raster
// itself got faster, 97 -> 93 us: profiles/r02/hist16_ab.txt.  Only k_setup_hist keeps the small histogram.)
static bool bin_h16(int per) { return SWR_TUNE_HIST16 && (int64_t)per * 64 < 65536; }

void launch_setup_bin(const DeviceFrame& f, hipStream_t s) {
    if (f.ntri <= 0) return;
    const SetupArgs a = make_setup_args(f);
    const int ntiles = f.tg.tiles_x * f.tg.tiles_y;
    if (f.plan.use_lds) {
        const int per = live_groups_per_workgroup(f.ntri, f.plan.G);
        const bool h16 = bin_h16(per);
        const size_t lds = (h16 ? (size_t)((ntiles + 1) / 2) * 4 : f.plan.lds_bytes) + (size_t)(per + 1) * 4;
        if (h16) hipLaunchKernelGGL((k_setup_hist<256, true>), dim3(f.plan.G), dim3(256), lds, s, a, f.bin_matrix, f.live, per, ntiles);
        else hipLaunchKernelGGL((k_setup_hist<256, false>), dim3(f.plan.G), dim3(256), lds, s, a, f.bin_matrix, f.live, per, ntiles);
        hipLaunchKernelGGL(k_colscan, dim3((ntiles + 15) / 16), dim3(256), 0, s, f.bin_matrix, f.plan.G, ntiles,
                           f.tile_count);
    } else {
        const unsigned blocks = (unsigned)((f.ntri + 255) / 256);
        hipLaunchKernelGGL(k_setup_bin, dim3(blocks), dim3(256), 0, s, a);
    }
}

// Can the frame be binned by the single-launch k_bin (fixed-stride bins), and how large may a tile region be?  Needs the
// LDS path's 16-bit histogram, class tags, 32-bit offsets into the region table, and cursor halves that stay below 2^16:
// region size + the primitives one workgroup owns < 65536.
uint32_t fixed_cap_max(int64_t ntri, int ntiles) {
    if (ntri <= 0 || ntiles <= 0 || ntri >= (1ll << CLASS_SHIFT)) return 0u;
    const BinPlan p = plan_binning(ntri, ntiles, false);
    if (!p.use_lds || p.threads != 256) return 0u;
    const int64_t own = (int64_t)live_groups_per_workgroup(ntri, p.G) * 64;
    if (own >= 65535 - 1024) return 0u;
    if ((size_t)((ntiles + 1) / 2) * 4 + (size_t)(own / 64 + 1) * 4 + 64 > 150 * 1024) return 0u;
    const uint64_t by_offset = 0xFFFFFFFFull / (uint64_t)ntiles;
    const uint64_t by_cursor = (uint64_t)(65535 - own);
    return (uint32_t)std::min<uint64_t>(std::min<uint64_t>(FIXED_CAP_MAX, by_offset), by_cursor) & ~63u;
}

bool launch_bin(const DeviceFrame& f, hipStream_t s, hipEvent_t stop) {
    BinArgs b;
    b.a = make_setup_args(f);
    b.fill = f.fill; b.fill_next = f.fill_next; b.bins = f.bins; b.cap = f.cap_tile;
    b.ntiles = f.tg.tiles_x * f.tg.tiles_y;
    b.per = live_groups_per_workgroup(f.ntri, f.plan.G);
    b.tag_class = f.ntri < (1ll << CLASS_SHIFT) ? 1 : 0;
    // the deferring kernel (three registers more: it would not fit beside five raster waves) only for frames
    // whose predecessor reported triangles for the list, and whose k_sort_bins runs
    b.biglist = f.biglist; b.defer_ok = (f.biglist && f.defer_big && !f.skip_sort) ? 1 : 0;
    const size_t lds = (size_t)((b.ntiles + 1) / 2) * 4 + (size_t)(b.per + 1) * 4 + 2 * (256 / 64) * 4 + 4;
    // the transform's last row is (0, 0, 0, 1): w == 1 for every finite vertex, no perspective divide (setup_triangle_r<.., AFF>)
    const bool aff = f.m[3] == 0.0f && f.m[7] == 0.0f && f.m[11] == 0.0f && f.m[15] == 1.0f;
#define SWR_BIN_GO2(MT, DF, AF) SWR_LAUNCH(stop, (k_bin<256, MT, DF, AF>), dim3(f.plan.G), dim3(256), (uint32_t)lds, s, b)
    if (aff) {
        if (b.defer_ok) { if (b.a.metal) SWR_BIN_GO2(true, true, true); else SWR_BIN_GO2(false, true, true); }
        else { if (b.a.metal) SWR_BIN_GO2(true, false, true); else SWR_BIN_GO2(false, false, true); }
    } else {
        if (b.defer_ok) { if (b.a.metal) SWR_BIN_GO2(true, true, false); else SWR_BIN_GO2(false, true, false); }
        else { if (b.a.metal) SWR_BIN_GO2(true, false, false); else SWR_BIN_GO2(false, false, false); }
    }
#undef SWR_BIN_GO2
    return stop != nullptr;
}

void launch_scan(const DeviceFrame& f, hipStream_t s) {
    if (f.plan.use_lds && f.ntri > 0) return;   // LDS path: the scan is fused into k_fill_lds
    const int n = f.tg.tiles_x * f.tg.tiles_y;
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, f.tile_count, f.tile_start, f.tile_cursor, n,
                       f.counters, f.host_counters, f.capacity);
}

bool launch_fill(const DeviceFrame& f, hipStream_t s, hipEvent_t stop) {
    if (f.ntri <= 0) return false;
    const int ntiles = f.tg.tiles_x * f.tg.tiles_y;
    if (f.plan.use_lds) {
        const int per = live_groups_per_workgroup(f.ntri, f.plan.G);
        const int tagged = f.ntri < (1ll << CLASS_SHIFT) ? 1 : 0;
        SWR_LAUNCH(stop, k_fill_lds<256>, dim3(f.plan.G), dim3(256), (uint32_t)(f.plan.lds_bytes + 4 * 256), s,
                   (const uint2*)f.ranges, f.ntri, (const uint32_t*)f.bin_matrix, (const uint32_t*)f.tile_count, f.tile_start,
                   f.counters, f.host_counters, f.bins, f.capacity, (const uint32_t*)f.live, per, ntiles, (int)f.tg.tiles_x, tagged, f.host_max);
    } else {
        const unsigned blocks = (unsigned)((f.ntri + 255) / 256);
        hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, s, f.ranges, f.ntri, f.tile_cursor, f.counters,
                           f.bins, f.capacity, f.tg.tiles_x, f.ntri < (1ll << CLASS_SHIFT) ? 1 : 0);
        return false;
    }
    return stop != nullptr;
}

bool launch_sort_bins(const DeviceFrame& f, hipStream_t s, hipEvent_t stop) {
    const unsigned tiles = (unsigned)(f.tg.tiles_x * f.tg.tiles_y);
    if (f.ntri <= 0 || tiles == 0 || f.skip_sort) return false;
    SWR_LAUNCH(stop, k_sort_bins, dim3(tiles), dim3(SORT_THREADS), 0, s, f.bins, (const uint32_t*)f.tile_start,
               (const uint32_t*)f.counters, f.capacity, f.ntri < (1ll << CLASS_SHIFT) ? 1 : 0,
               (uint32_t*)(f.fixed_bins ? f.fill : nullptr), f.fixed_bins ? f.cap_tile : 0u,
               (const uint4*)(f.fixed_bins ? f.biglist : nullptr), (int)f.tg.tiles_x);
    return stop != nullptr;
}

// Does this frame take k_raster_depth (32-bit depth keys)?  Depth-only, z-tested, CPU rules, and the scene has not been moved
// to the 64-bit kernel by the host (DeviceFrame::k32).
bool frame_uses_k32(const DeviceFrame& f) {
    return f.k32 && (f.flags & SWR_FLAG_DEPTH_TEST) && (f.flags & SWR_FLAG_NO_COLOR) && !(f.flags & SWR_FLAG_METAL_RULES);
}

bool launch_raster(const DeviceFrame& f, hipStream_t s, hipEvent_t stop) {
    RasterArgs a;
    a.geo = f.geo; a.geo_full = f.geo_full; a.tri_rgb = f.tri_rgb;
    a.inv = f.inv; a.reordered = f.reordered;
    a.tri_nrm = f.tri_nrm;
    a.fs.shader = f.material.shader; a.fs.shininess_log2 = f.material.shininess_log2;
    a.fs.light_dir = make_float3(f.material.light_dir[0], f.material.light_dir[1], f.material.light_dir[2]);
    a.fs.half_dir = make_float3(f.material.half_dir[0], f.material.half_dir[1], f.material.half_dir[2]);
    a.fs.ambient = f.material.ambient; a.fs.diffuse = f.material.diffuse; a.fs.specular = f.material.specular;
    a.fs.texels = f.texels; a.fs.tex_w = f.tex_w; a.fs.tex_h = f.tex_h;
    a.tile_start = f.tile_start; a.bins = f.bins;
    a.counters = f.counters; a.capacity = f.capacity;
    a.color = (f.flags & SWR_FLAG_NO_COLOR) ? nullptr : f.color;
    a.depth = f.depth; a.tg = f.tg;
    a.tag_class = f.ntri < (1ll << CLASS_SHIFT) ? 1 : 0;
    a.fill = f.fixed_bins ? f.fill : nullptr;
    a.fixed_cap = f.fixed_bins ? f.cap_tile : 0u;
    a.host_pairs = f.host_counters; a.host_fill = f.host_fill; a.host_max = f.host_max;
    a.redo_dev = f.redo_dev; a.host_redo = f.host_redo;
    a.insort = f.insort;
    a.pack_local = f.ntri <= (1ll << WTAB_PRIM_BITS) ? 1 : 0;
    const bool plain = !a.pack_local;      // more than 2^20 primitives: the colour kernels without the winner table
    const unsigned ntiles = (unsigned)(f.tg.tiles_x * f.tg.tiles_y);
    if (ntiles == 0) return false;
    // Small grids (a small window: the reference app's 512x512 is 128 tiles): four workgroups fit where one tile's
    // would run, so four share a tile, each walking and resolving its own 8 rows of it — the per-triangle setup is paid
    // four times, the row steps and the pixel work are divided (the app's sphere: k_raster 19.9 -> 12.8 us, Metal rules
    // 25.6 -> 14.3 us).  Two per tile for 320-640 tiles (1/8 band of cfg4: 510 dense tiles) measured no gain (23.3 vs
    // 24.6 us): those workgroups are bound by the gather -> setup latency chain of their chunks, not by their rows
    // (profiles/r02/vsplit_ab.txt).  -DSWR_TUNE_VSPLIT=0/1/2 forces the log2 of the split.
    constexpr int vs_mode = SWR_TUNE_VSPLIT;
    a.vs_log = vs_mode >= 0 ? std::min(vs_mode, 2) : (ntiles * 4 <= 1280 ? 2 : 0);
    const unsigned tiles = ntiles << a.vs_log;
    const bool ext = f.material.shader != SWR_SHADER_PASSTHROUGH && a.color != nullptr;
    if (f.flags & SWR_FLAG_METAL_RULES) {
        if (ext && plain) SWR_LAUNCH(stop, (k_raster_ext<true, true, true>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
        else if (ext) SWR_LAUNCH(stop, (k_raster_ext<true, true>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
        else if (a.color && plain) SWR_LAUNCH(stop, (k_raster<true, 0, true, true, true>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
        else if (a.color) SWR_LAUNCH(stop, (k_raster<true, 0, true, true>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
        else SWR_LAUNCH(stop, (k_raster<true, 0, true, false>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
        return stop != nullptr;
    }
    if (ext) {
        if (f.flags & SWR_FLAG_DEPTH_TEST) {
            if (plain) SWR_LAUNCH(stop, (k_raster_ext<true, false, true>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
            else SWR_LAUNCH(stop, (k_raster_ext<true, false>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
        } else {
            if (plain) SWR_LAUNCH(stop, (k_raster_ext<false, false, true>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
            else SWR_LAUNCH(stop, (k_raster_ext<false, false>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
        }
        return stop != nullptr;
    }
#ifdef SWR_ABLATION
    // timing-only ablations of k_raster<ztest> (results invalid): compiled only into lib/libswr_hip_ablation.so
    // (`make ablation`, used by tools/variants.sh); the product library has neither the kernels nor the switch
    static const int variant = getenv("SWR_DEBUG_VARIANT") ? atoi(getenv("SWR_DEBUG_VARIANT")) : 0;
    if ((f.flags & SWR_FLAG_DEPTH_TEST) && variant > 0 && !a.color) {
        switch (variant) {
#define SWR_V(N) case N: hipLaunchKernelGGL((k_raster<true, N>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a); return false;
            SWR_V(1) SWR_V(2) SWR_V(3) SWR_V(4) SWR_V(5) SWR_V(8) SWR_V(9) SWR_V(10) SWR_V(11)
#undef SWR_V
            default: break;
        }
    }
#endif
    if (f.flags & SWR_FLAG_DEPTH_TEST) {
        if (a.color && plain) SWR_LAUNCH(stop, (k_raster<true, 0, false, true, true>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
        else if (a.color) SWR_LAUNCH(stop, (k_raster<true, 0, false, true>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
        else if (frame_uses_k32(f)) SWR_LAUNCH(stop, k_raster_depth, dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
        else SWR_LAUNCH(stop, (k_raster<true, 0, false, false>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
    } else {
        if (a.color && plain) SWR_LAUNCH(stop, (k_raster<false, 0, false, true, true>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
        else if (a.color) SWR_LAUNCH(stop, (k_raster<false, 0, false, true>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
        else SWR_LAUNCH(stop, (k_raster<false, 0, false, false>), dim3(tiles), dim3(RASTER_THREADS), 0, s, a);
    }
    return stop != nullptr;
}

}  // namespace swr
