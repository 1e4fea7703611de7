// swr_shaders.hip.h — the two shader hooks of the reference, with their signatures kept
// (renderer/Shaders.metal:39-55 vertex_shader, :116-121 fragment_shader), as gfx950 device
// functions.  The raster pipeline calls them at the same two points the Metal pipeline does:
// vertex_shader once per vertex reference in the setup kernel (clip-space position, the
// divide by w happens in the caller like Shaders.metal:68 / Renderer.swift:161), and
// fragment_shader once per resolved pixel.
#pragma once
#include <hip/hip_runtime.h>

namespace swr {

// Metal float4x4: four float4 columns (Shaders.metal:47; Renderer.swift:199).
struct float4x4 {
    float4 columns[4];
};

// Shaders.metal:39-42 / GpuRenderer.swift:14-17 — 32 bytes.
struct VertexOut {
    float4 pos;
    float3 color;
};

// transform * float4(xyz, 1): column accumulation, one rounding per operation
// (col0*x, + col1*y, + col2*z, + col3*1) — Renderer.swift:160, Shaders.metal:50.
__device__ __forceinline__ VertexOut vertex_shader(float3 xyz, float3 color,
                                                   const float4x4& transform) {
    const float4 c0 = transform.columns[0], c1 = transform.columns[1];
    const float4 c2 = transform.columns[2], c3 = transform.columns[3];
    float4 r;
    r.x = c0.x * xyz.x; r.y = c0.y * xyz.x; r.z = c0.z * xyz.x; r.w = c0.w * xyz.x;
    r.x = r.x + c1.x * xyz.y; r.y = r.y + c1.y * xyz.y; r.z = r.z + c1.z * xyz.y; r.w = r.w + c1.w * xyz.y;
    r.x = r.x + c2.x * xyz.z; r.y = r.y + c2.y * xyz.z; r.z = r.z + c2.z * xyz.z; r.w = r.w + c2.w * xyz.z;
    r.x = r.x + c3.x * 1.0f;  r.y = r.y + c3.y * 1.0f;  r.z = r.z + c3.z * 1.0f;  r.w = r.w + c3.w * 1.0f;
    VertexOut out;
    out.pos = r;
    out.color = color;
    return out;
}

// Shaders.metal:116-121: return float4(vin.color, 1).
__device__ __forceinline__ float4 fragment_shader(VertexOut vin) {
    return make_float4(vin.color.x, vin.color.y, vin.color.z, 1.0f);
}

}  // namespace swr
