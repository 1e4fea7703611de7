// Renderer.hpp — host-side mirror of the reference's draw-call interface, over the C-ABI.
//
// The reference's host language is Swift (renderer/Renderer.swift, renderer/GpuRenderer.swift);
// this image has no Swift toolchain, so the host side above include/swr.h is written in C++ with
// the SAME type and method names, argument meaning and defaults, so that call sites read like
// the reference's (App.swift:153-185):
//
//     RenderPass pass{image, depthImage, vertices, indices};   // Renderer.swift:191-200
//     pass.primitiveType = PrimitiveType::triangle;            // App.swift:164
//     pass.transform = projectionMatrix * transform.matrix;    // App.swift:183
//     renderer.render(pass);                                   // App.swift:185
//
// The Swift façade a maintainer would drop into the app is in host/swift/ (see INTEGRATION.md).
//
// Error behaviour: the reference has no error channel — it traps (fatalError / assert / try!,
// Renderer.swift:26,209,239; GpuRenderer.swift:20-38).  The closest C++ analogue is an exception:
// every non-zero swr status is thrown as swr_host::RenderError carrying swr_last_error().
// There is no CPU fallback: constructing a renderer without a HIP device throws.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/swr.h"

namespace swr_host {

struct RenderError : std::runtime_error {
    int code;
    RenderError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// Renderer.swift:44-49 — memory order b,g,r,a.
struct Pixel {
    uint8_t b, g, r, a;
};
static_assert(sizeof(Pixel) == 4, "Pixel is BGRA8");

// Renderer.swift:8-42 — a non-owning view of caller memory; element (x,y) at pointer[y*width+x]
// (App.swift:351-360: bytesPerRow is stored but addressing uses width).
template <class P>
class Image {
public:
    Image(P* pointer, long width, long height, long bytesPerRow)
        : pointer(pointer), width(width), height(height), bytesPerRow(bytesPerRow) {}
    P* pointer;
    const long width, height, bytesPerRow;
    bool contains(long x, long y) const { return x >= 0 && x < width && y >= 0 && y < height; }
    // get: the reference fatalError()s out of bounds (Renderer.swift:24-29)
    const P& at(long x, long y) const {
        if (!contains(x, y)) throw RenderError(SWR_ERR_BAD_ARG, "Image subscript out of bounds");
        return pointer[y * width + x];
    }
    // set: silently dropped out of bounds (Renderer.swift:30-36)
    void set(long x, long y, const P& v) {
        if (contains(x, y)) pointer[y * width + x] = v;
    }
};
using ColorImage = Image<Pixel>;   // Renderer.swift:5
using DepthImage = Image<float>;   // Renderer.swift:6

// Renderer.swift:154-157 — two SIMD3<Float>, 16 bytes each (lane 3 is padding).
struct Vertex {
    float xyz[4];
    float color[4];
    Vertex() : xyz{0, 0, 0, 0}, color{0, 0, 0, 0} {}
    Vertex(float x, float y, float z, float r, float g, float b) : xyz{x, y, z, 0}, color{r, g, b, 0} {}
};
static_assert(sizeof(Vertex) == sizeof(swr_vertex), "Vertex must match swr_vertex");

// Renderer.swift:174-189
enum class PrimitiveType { triangle, line, vertices };
inline int verticesCount(PrimitiveType t) { return t == PrimitiveType::line ? 2 : 3; }

// matrix_float4x4: column-major, columns[c][r] (Renderer.swift:199; default = identity).
struct matrix_float4x4 {
    float columns[4][4];
    static matrix_float4x4 identity() {
        matrix_float4x4 m{};
        for (int i = 0; i < 4; i++) m.columns[i][i] = 1.0f;
        return m;
    }
    // matrix_float4x4(rows:) as used at App.swift:176-181
    static matrix_float4x4 fromRows(const float r[4][4]) {
        matrix_float4x4 m{};
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) m.columns[j][i] = r[i][j];
        return m;
    }
    friend matrix_float4x4 operator*(const matrix_float4x4& a, const matrix_float4x4& b) {
        matrix_float4x4 o{};
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) {
                float s = 0.0f;
                for (int k = 0; k < 4; k++) s += a.columns[k][r] * b.columns[c][k];
                o.columns[c][r] = s;
            }
        return o;
    }
};

// ---- fragment-stage extensions (not in the reference; include/swr.h swr_vertex_attr / swr_material) ----
// Per-vertex normal + texture coordinate, parallel to RenderPass.vertices.
struct VertexAttributes {
    float normal[4];
    float uv[4];
    VertexAttributes() : normal{0, 0, 0, 0}, uv{0, 0, 0, 0} {}
    VertexAttributes(float nx, float ny, float nz, float u, float v) : normal{nx, ny, nz, 0}, uv{u, v, 0, 0} {}
};
static_assert(sizeof(VertexAttributes) == sizeof(swr_vertex_attr), "VertexAttributes must match swr_vertex_attr");

enum class Shader { passthrough = SWR_SHADER_PASSTHROUGH, phong = SWR_SHADER_PHONG,
                    texturedPhong = SWR_SHADER_TEXTURED_PHONG };

// What the Metal pipeline would bind as fragment-function arguments.
struct Material {
    Shader shader = Shader::passthrough;       // passthrough = Shaders.metal:116-121
    int shininessLog2 = 5;                     // specular exponent 2^k
    float lightDirection[3] = {0, 0, -1};      // unit, towards the light, in the space of the normals
    float halfDirection[3] = {0, 0, -1};       // unit Blinn half vector
    float ambient = 0.15f, diffuse = 0.8f, specular = 0.4f;
    const Image<Pixel>* texture = nullptr;     // texturedPhong only; caller-owned b,g,r,a texels
};

// Renderer.swift:191-200 (+ the optional extended fragment stage: empty attributes / passthrough material
// = the reference's RenderPass exactly)
struct RenderPass {
    ColorImage colorBuffer;
    DepthImage depthBuffer;
    std::vector<Vertex> vertices;
    std::vector<int64_t> indices;                       // Swift Int
    PrimitiveType primitiveType = PrimitiveType::triangle;
    matrix_float4x4 transform = matrix_float4x4::identity();
    std::vector<VertexAttributes> attributes = {};
    Material material = {};
};

namespace detail {
class Context {
public:
    // deviceCount > 1: one context drives that many tile-row bands, on as many GPUs as are visible (swr_config.device_count)
    explicit Context(uint32_t deviceCount = 0) {
        swr_config cfg{deviceCount > 1 ? 0 : -1, deviceCount, 0, 0};
        int rc = swr_context_create(&cfg, &ctx_);
        if (rc) throw RenderError(rc, swr_last_error(nullptr));
    }
    ~Context() { swr_context_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    // sceneId != 0: the caller's promise that p.vertices / p.indices / p.attributes / the texture hold what they held at
    // the last call with this id (swr_render_pass.scene_id): nothing is uploaded, the pass costs one resident frame + the gather
    void render(const RenderPass& p, uint32_t flags, uint64_t sceneId = 0) {
        swr_render_pass rp{};
        rp.scene_id = sceneId;
        rp.color = p.colorBuffer.pointer;
        rp.depth = p.depthBuffer.pointer;
        rp.width = p.colorBuffer.width;
        rp.height = p.colorBuffer.height;
        rp.color_bytes_per_row = p.colorBuffer.bytesPerRow;
        rp.depth_bytes_per_row = p.depthBuffer.bytesPerRow;
        rp.vertices = reinterpret_cast<const swr_vertex*>(p.vertices.data());
        rp.vertex_count = (int64_t)p.vertices.size();
        rp.indices = p.indices.data();
        rp.index_count = (int64_t)p.indices.size();
        rp.primitive_type = (int32_t)p.primitiveType;
        rp.flags = flags;
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) rp.transform[4 * c + r] = p.transform.columns[c][r];
        swr_material mat{};
        if (p.material.shader != Shader::passthrough) {
            if (p.attributes.size() != p.vertices.size())
                throw RenderError(SWR_ERR_BAD_ARG, "RenderPass.attributes must have one entry per vertex");
            mat.shader = (int32_t)p.material.shader;
            mat.shininess_log2 = p.material.shininessLog2;
            for (int k = 0; k < 3; k++) {
                mat.light_dir[k] = p.material.lightDirection[k];
                mat.half_dir[k] = p.material.halfDirection[k];
            }
            mat.ambient = p.material.ambient; mat.diffuse = p.material.diffuse; mat.specular = p.material.specular;
            rp.attributes = reinterpret_cast<const swr_vertex_attr*>(p.attributes.data());
            rp.material = &mat;
            if (p.material.shader == Shader::texturedPhong) {
                if (!p.material.texture) throw RenderError(SWR_ERR_BAD_ARG, "texturedPhong needs Material.texture");
                rp.texture = p.material.texture->pointer;
                rp.tex_width = (int32_t)p.material.texture->width;
                rp.tex_height = (int32_t)p.material.texture->height;
            }
        }
        int rc = swr_render(ctx_, &rp);
        if (rc) throw RenderError(rc, swr_last_error(ctx_));
    }
private:
    swr_context* ctx_ = nullptr;
};
}  // namespace detail

// final class Renderer { func render(renderPass:) } (Renderer.swift:202-230): the CPU renderer's
// semantics exactly as written — painter's order, depth image left at +inf — executed on the GPU.
class Renderer {
public:
    // The reference's caller draws the same mesh every display frame with a new transform (App.swift:153-185).
    // staticScene = true tells the renderer so: the mesh of the first render(renderPass:) stays resident on the GPU
    // (what GpuRenderer.swift:32-33,41-67 does with its MTLBuffers) until sceneVersion is changed or the vertex / index
    // counts differ.  false (the default) uploads on every call, whatever the arrays hold.
    bool staticScene = false;
    uint64_t sceneVersion = 1;          // bump after editing the arrays of a static scene in place
    // .line passes: false (default) = the reference as written — draw(line:colorBuffer:depthBuffer:) is an empty stub
    // (Renderer.swift:289-293), the pass only clears; true = draw every line with the reference's own DDA
    // (draw(line:with:in:), Renderer.swift:405-419; SWR_FLAG_REAL_LINES)
    bool realLines = false;
    void render(const RenderPass& renderPass) {
        const bool lines = realLines && renderPass.primitiveType == PrimitiveType::line;
        ctx_.render(renderPass, lines ? (uint32_t)SWR_FLAG_REAL_LINES : 0u, staticScene ? sceneVersion : 0);
    }
private:
    detail::Context ctx_;
};

// final class GpuRenderer { func render(renderPass:) } (GpuRenderer.swift:12,35): same pixel rules
// as Renderer (the parity target), with the z-test the Metal path has (Shaders.metal:158-165)
// restored from Renderer.swift:257-261.  depthTest = false gives Renderer's output.
class GpuRenderer {
public:
    GpuRenderer() = default;
    // One renderer, N GPUs: the framebuffer is cut into N tile-row bands and every band is copied straight into its rows
    // of RenderPass.colorBuffer / .depthBuffer (the reference has exactly one synchronous draw call, GpuRenderer.swift:35).
    explicit GpuRenderer(uint32_t deviceCount) : ctx_(deviceCount) {}
    bool depthTest = true;
    // true: the Metal kernels' own rules (round() snap, ROI threads + inside test, UNORM rounding,
    // ROI-min == 0 skip; Shaders.metal:57-167, GpuRenderer.swift:122-124) instead of the CPU renderer's
    bool metalRules = false;
    // see Renderer::staticScene: the resident mesh of GpuRenderer.swift:32-33,41-67
    bool staticScene = false;
    uint64_t sceneVersion = 1;
    bool realLines = false;             // see Renderer::realLines
    void render(const RenderPass& renderPass) {
        const bool lines = realLines && renderPass.primitiveType == PrimitiveType::line;
        ctx_.render(renderPass, (metalRules ? (uint32_t)SWR_FLAG_METAL_RULES : (depthTest ? (uint32_t)SWR_FLAG_DEPTH_TEST : 0u)) |
                                    (lines ? (uint32_t)SWR_FLAG_REAL_LINES : 0u),
                    staticScene ? sceneVersion : 0);
    }
private:
    detail::Context ctx_;
};

}  // namespace swr_host
