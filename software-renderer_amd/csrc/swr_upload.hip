// swr_upload.hip — once-per-scene preprocessing (swr_scene_upload / swr_scene_attributes).
//
// The frame kernels do not walk RenderPass.indices in their given order.  At upload the primitives are
// sorted by the Morton code of their object-space centroid and de-indexed into a "triangle stream":
//
//   tri_xyz[3s+k] = position of corner k of the primitive in sorted slot s (w of corner 0 = original index)
//   tri_rgb[3s+k] = its colour (lane 3: texture v),   tri_nrm[3s+k] = (normal, texture u)   [attributes]
//   inv[o]        = sorted slot of original primitive o
//   box64[2g..]   = object-space bounding box of the 64 primitives of group g = slots [64g, 64g+64)
//
// Why: (1) a rank that owns one tile-row band of the framebuffer culls whole 64-primitive groups whose
// projected box misses its band before touching their vertices, so the per-GPU setup cost shrinks with
// the band instead of staying at "all primitives" (DESIGN.md §7); (2) primitives that are neighbours on
// screen are neighbours in memory: setup reads are contiguous (no index gather), bin fills write runs,
// and the raster's record gathers hit fewer lines.  The image is unaffected: visibility keys carry the
// ORIGINAL primitive index (painter's order and z-tie order of Renderer.swift:222,258 refer to it), and
// the keys make the result independent of processing order.
//
// The sort itself is hipCUB's device radix sort (stable, so equal codes keep index order): it runs once
// per scene, not per frame.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "swr_internal.h"

namespace swr {

namespace {

__device__ __forceinline__ uint32_t ordered_bits(float f) {           // monotone float -> uint (non-NaN)
    const uint32_t u = __float_as_uint(f);
    return u ^ (uint32_t)(((int32_t)u >> 31) | 0x80000000);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// bounds[0..2] = min xyz, bounds[3..5] = max xyz over the finite vertices (ordered-uint encoding)
__global__ void k_bounds_init(uint32_t* bounds) {
    if (threadIdx.x < 3) bounds[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) bounds[threadIdx.x] = 0u;
}
__global__ void k_scene_bounds(const swr_vertex* __restrict__ vtx, int64_t nv, uint32_t* __restrict__ bounds) {
    const float4* xyz = reinterpret_cast<const float4*>(vtx);     // positions at even float4 slots of the AoS Vertex
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const float4 v = xyz[2 * i];
        const float c[3] = {v.x, v.y, v.z};
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (fabsf(c[k]) < INFINITY) { lo[k] = fminf(lo[k], c[k]); hi[k] = fmaxf(hi[k], c[k]); }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
        if ((threadIdx.x & 63) == 0) {
            if (lo[k] <= hi[k]) {
                atomicMin(&bounds[k], ordered_bits(lo[k]));
                atomicMax(&bounds[3 + k], ordered_bits(hi[k]));
            }
        }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {           // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// 30-bit Morton code of the centroid, normalised to the scene's bounding box; ids = 0..ntri-1.
// A primitive with a bad index or a non-finite vertex gets code 0 (its place does not matter).
__global__ void k_morton(const swr_vertex* __restrict__ vtx, int64_t nv, const int64_t* __restrict__ idx, int64_t ntri,
                         const uint32_t* __restrict__ bounds, uint32_t* __restrict__ codes,
                         uint32_t* __restrict__ ids, int sort) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float4* xyz = reinterpret_cast<const float4*>(vtx);
    float lo[3], scale[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        lo[k] = from_ordered_bits(bounds[k]);
        const float ext = from_ordered_bits(bounds[3 + k]) - lo[k];
        scale[k] = (ext > 0.0f && ext < INFINITY) ? 1023.0f / ext : 0.0f;
    }
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < ntri; p += stride) {
        uint32_t code = 0u;
        const int64_t i0 = idx[3 * p], i1 = idx[3 * p + 1], i2 = idx[3 * p + 2];
        if (sort && i0 >= 0 && i0 < nv && i1 >= 0 && i1 < nv && i2 >= 0 && i2 < nv) {
            const float4 a = xyz[2 * i0], b = xyz[2 * i1], c = xyz[2 * i2];
            const float cen[3] = {(a.x + b.x + c.x) * (1.0f / 3.0f), (a.y + b.y + c.y) * (1.0f / 3.0f),
                                  (a.z + b.z + c.z) * (1.0f / 3.0f)};
            uint32_t q[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float t = (cen[k] - lo[k]) * scale[k];
                q[k] = (t >= 0.0f && t <= 1023.0f) ? (uint32_t)t : 0u;      // NaN / out of range -> 0
            }
            code = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
        }
        codes[p] = code;
        ids[p] = (uint32_t)p;
    }
}

// slot s <- original primitive perm[s]
// (perm == NULL: identity order, slots [s0, ntri) only — the one-shot upload's chunks)
__global__ void k_gather_stream(const swr_vertex* __restrict__ v, int64_t nv, const int64_t* __restrict__ idx,
                                int64_t s0, int64_t ntri, const uint32_t* __restrict__ perm, float4* __restrict__ tri_xyz,
                                float4* __restrict__ tri_rgb, uint32_t* __restrict__ inv) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float4* vp = reinterpret_cast<const float4*>(v);
    for (int64_t s = s0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < ntri; s += stride) {
        const uint32_t o = perm ? perm[s] : (uint32_t)s;
        inv[o] = (uint32_t)s;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int64_t ix = idx[3 * (int64_t)o + k];
            float4 x = make_float4(0, 0, 0, 0), c = x;
            if (ix >= 0 && ix < nv) { x = vp[2 * ix]; c = vp[2 * ix + 1]; }   // bad indices are reported by the upload
            x.w = k == 0 ? __uint_as_float(o) : 0.0f;
            c.w = 0.0f;
            tri_xyz[3 * s + k] = x;
            tri_rgb[3 * s + k] = c;
        }
    }
}

// one wave per group of 64 slots: object-space box of their 192 vertices.  A non-finite coordinate
// poisons the box with NaN, which the frame's cull test reads as "cannot be culled".
__global__ void k_box64(const float4* __restrict__ tri_xyz, int64_t g0, int64_t ntri, float4* __restrict__ box64) {
    const int64_t g = g0 + (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int64_t groups = (ntri + 63) / 64;
    if (g >= groups) return;
    const int64_t s = g * 64 + (threadIdx.x & 63);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    if (s < ntri) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float4 x = tri_xyz[3 * s + k];
            const float c[3] = {x.x, x.y, x.z};
#pragma unroll
            for (int j = 0; j < 3; j++) {
                bad = bad || !(fabsf(c[j]) < INFINITY);
                lo[j] = fminf(lo[j], c[j]);
                hi[j] = fmaxf(hi[j], c[j]);
            }
        }
    }
    const bool any_bad = __ballot(bad) != 0ull;
#pragma unroll
    for (int j = 0; j < 3; j++)
        for (int off = 32; off > 0; off >>= 1) {
            lo[j] = fminf(lo[j], __shfl_xor(lo[j], off));
            hi[j] = fmaxf(hi[j], __shfl_xor(hi[j], off));
        }
    if ((threadIdx.x & 63) == 0) {
        const float nan = __uint_as_float(0x7FC00000u);
        box64[2 * g] = any_bad ? make_float4(nan, nan, nan, 0) : make_float4(lo[0], lo[1], lo[2], 0);
        box64[2 * g + 1] = any_bad ? make_float4(nan, nan, nan, 0) : make_float4(hi[0], hi[1], hi[2], 0);
    }
}

// swr_scene_attributes: the extra varyings, de-indexed into the same sorted slots.
// tri_nrm[3s+k] = (nx, ny, nz, u); v rides in the padding lane of tri_rgb[3s+k] (r, g, b, v).
__global__ void k_gather_attrs(const swr_vertex_attr* __restrict__ attrs, int64_t nv, const int64_t* __restrict__ idx,
                               int64_t ntri, const float4* __restrict__ tri_xyz, float4* __restrict__ tri_nrm,
                               float4* __restrict__ tri_rgb) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float4* ap = reinterpret_cast<const float4*>(attrs);
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < ntri; s += stride) {
        const uint32_t o = __float_as_uint(tri_xyz[3 * s].w);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int64_t ix = idx[3 * (int64_t)o + k];
            if (ix < 0 || ix >= nv) continue;        // cannot happen: the scene upload validated the indices
            const float4 n = ap[2 * ix], t = ap[2 * ix + 1];
            tri_nrm[3 * s + k] = make_float4(n.x, n.y, n.z, t.x);
            tri_rgb[3 * s + k].w = t.y;
        }
    }
}

}  // namespace

size_t stream_sort_temp_bytes(int64_t ntri) {
    if (ntri <= 0) return 0;
    size_t bytes = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                       (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)ntri, 0, 30, (hipStream_t)0);
    return bytes;
}

// scratch: 4 arrays of ntri uint32 (codes in/out, ids in/out) followed by 8 words of bounds.
hipError_t launch_build_stream(const StreamBuild& b, hipStream_t s) {
    if (b.ntri <= 0) return hipSuccess;
    uint32_t* codes_in = b.scratch;
    uint32_t* codes_out = b.scratch + b.ntri;
    uint32_t* ids_in = b.scratch + 2 * b.ntri;
    uint32_t* ids_out = b.scratch + 3 * b.ntri;
    uint32_t* bounds = b.scratch + 4 * b.ntri;
    hipLaunchKernelGGL(k_bounds_init, dim3(1), dim3(64), 0, s, bounds);
    if (b.nv > 0) hipLaunchKernelGGL(k_scene_bounds, dim3(128), dim3(256), 0, s, b.vertices, b.nv, bounds);   // 6 atomics per wave
    hipLaunchKernelGGL(k_morton, dim3(2048), dim3(256), 0, s, b.vertices, b.nv, b.indices, b.ntri, bounds, codes_in, ids_in,
                       b.sort ? 1 : 0);
    const uint32_t* perm = ids_in;
    if (b.sort) {
        size_t bytes = b.sort_temp_bytes;
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(b.sort_temp, bytes, codes_in, codes_out, ids_in, ids_out,
                                                          (int)b.ntri, 0, 30, s);
        if (e != hipSuccess) return e;
        perm = ids_out;
    }
    hipLaunchKernelGGL(k_gather_stream, dim3(2048), dim3(256), 0, s, b.vertices, b.nv, b.indices, (int64_t)0, b.ntri, perm,
                       b.tri_xyz, b.tri_rgb, b.inv);
    const int64_t groups = (b.ntri + 63) / 64;
    hipLaunchKernelGGL(k_box64, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, s, b.tri_xyz, (int64_t)0, b.ntri, b.box64);
    return hipGetLastError();
}

// The triangle stream of primitives [t0, t1) in index order (t0 a multiple of 64): what launch_build_stream does with
// sort == false, for one chunk of the index array — the one-shot upload runs it behind each chunk's copy.
hipError_t launch_build_stream_range(const StreamBuild& b, int64_t t0, int64_t t1, hipStream_t s) {
    if (t1 <= t0) return hipSuccess;
    const int64_t n = t1 - t0;
    hipLaunchKernelGGL(k_gather_stream, dim3((unsigned)std::min<int64_t>(2048, (n + 255) / 256)), dim3(256), 0, s, b.vertices, b.nv,
                       b.indices, t0, t1, (const uint32_t*)nullptr, b.tri_xyz, b.tri_rgb, b.inv);
    const int64_t g0 = t0 / 64, g1 = (t1 + 63) / 64;
    hipLaunchKernelGGL(k_box64, dim3((unsigned)((g1 - g0 + 3) / 4)), dim3(256), 0, s, b.tri_xyz, g0, t1, b.box64);
    return hipGetLastError();
}

void launch_gather_attrs(const swr_vertex_attr* attrs, int64_t nv, const int64_t* indices, int64_t ntri,
                         const float4* tri_xyz, float4* tri_nrm, float4* tri_rgb, hipStream_t s) {
    if (nv <= 0 || ntri <= 0) return;
    hipLaunchKernelGGL(k_gather_attrs, dim3(2048), dim3(256), 0, s, attrs, nv, indices, ntri, tri_xyz, tri_nrm, tri_rgb);
}

}  // namespace swr
