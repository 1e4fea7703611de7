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

// Shaders.metal:39-42 / GpuRenderer.swift:14-17 — pos + color (32 bytes in the reference).
// normal / uv are the extra varyings of the extended fragment stage (SURVEY.md §8(f) rank 2,
// include/swr.h swr_vertex_attr); the reference's own stage never reads them.
struct VertexOut {
    float4 pos;
    float3 color;
    float3 normal;
    float2 uv;
};

// What Metal would bind as [[buffer]] / [[texture]] arguments of the fragment function:
// include/swr.h swr_material + the texture of swr_texture_upload.
struct FragmentUniforms {
    int shader;              // SWR_SHADER_*
    int shininess_log2;
    float3 light_dir;
    float3 half_dir;
    float ambient, diffuse, specular;
    const float4* texels;    // tex_w*tex_h texels as (r,g,b,a) = channel / 255.0f, converted once at
                             // swr_texture_upload (k_texture_to_float) so the per-pixel fetch is one 16-B load
    int tex_w, tex_h;
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

// One texel as r,g,b floats in [0,1], repeat addressing.  x, y come from floor(frac(uv) * size - 0.5)
// and its +1 neighbour, i.e. lie in [-1, size]: one conditional add / subtract is the modulo.  The final
// clamp only matters for non-finite uv (outside the defined domain) and keeps the load in bounds.
__device__ __forceinline__ float3 fetch_texel(const FragmentUniforms& u, int x, int y) {
    x = x < 0 ? x + u.tex_w : (x >= u.tex_w ? x - u.tex_w : x);
    y = y < 0 ? y + u.tex_h : (y >= u.tex_h ? y - u.tex_h : y);
    x = min(max(x, 0), u.tex_w - 1);
    y = min(max(y, 0), u.tex_h - 1);
    // 12 of the texel's 16 bytes (alpha is not used): four texels in flight are 12 registers instead of 16
    const float* t = reinterpret_cast<const float*>(u.texels + ((size_t)y * (size_t)u.tex_w + (size_t)x));
    return make_float3(t[0], t[1], t[2]);
}

// The extended fragment stage (not in the reference; defined in include/swr.h at swr_material and
// DESIGN.md §10 — the test checker restates it with the same operations in the same order, one
// IEEE binary32 rounding each):
// per-pixel Blinn-Phong on the interpolated normal, optional bilinear texture on the base colour.
__device__ __forceinline__ float4 fragment_shader(VertexOut vin, const FragmentUniforms& u) {
    if (u.shader == 0) return fragment_shader(vin);
    const float3 n = vin.normal;
    const float len2 = n.x * n.x + n.y * n.y + n.z * n.z;
    float3 N = make_float3(0.0f, 0.0f, 0.0f);
    if (len2 > 0.0f) {
        const float len = sqrtf(len2);
        N = make_float3(n.x / len, n.y / len, n.z / len);
    }
    float ndl = N.x * u.light_dir.x + N.y * u.light_dir.y + N.z * u.light_dir.z;
    ndl = fmaxf(ndl, 0.0f);
    float ndh = N.x * u.half_dir.x + N.y * u.half_dir.y + N.z * u.half_dir.z;
    ndh = fmaxf(ndh, 0.0f);
    float s = ndh;
    for (int i = 0; i < u.shininess_log2; i++) s = s * s;
    float3 base = vin.color;
    if (u.shader == 2) {
        const float fu = vin.uv.x - floorf(vin.uv.x), fv = vin.uv.y - floorf(vin.uv.y);
        const float x = fu * (float)u.tex_w - 0.5f, y = fv * (float)u.tex_h - 0.5f;
        const float x0f = floorf(x), y0f = floorf(y);
        const float ax = x - x0f, ay = y - y0f;
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float3 t00 = fetch_texel(u, x0, y0), t10 = fetch_texel(u, x0 + 1, y0);
        const float3 t01 = fetch_texel(u, x0, y0 + 1), t11 = fetch_texel(u, x0 + 1, y0 + 1);
        const float3 top = make_float3(t00.x + (t10.x - t00.x) * ax, t00.y + (t10.y - t00.y) * ax,
                                       t00.z + (t10.z - t00.z) * ax);
        const float3 bot = make_float3(t01.x + (t11.x - t01.x) * ax, t01.y + (t11.y - t01.y) * ax,
                                       t01.z + (t11.z - t01.z) * ax);
        base.x = base.x * (top.x + (bot.x - top.x) * ay);
        base.y = base.y * (top.y + (bot.y - top.y) * ay);
        base.z = base.z * (top.z + (bot.z - top.z) * ay);
    }
    const float lit = u.ambient + u.diffuse * ndl;
    const float spec = u.specular * s;
    return make_float4(base.x * lit + spec, base.y * lit + spec, base.z * lit + spec, 1.0f);
}

}  // namespace swr
