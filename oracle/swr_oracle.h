/*
 * swr_oracle.h — CPU restatement of the reference's CPU triangle path.  TEST INFRASTRUCTURE.
 *
 * PARITY UNPINNED: the reference (zhvrnkov/software-renderer) ships no tests, golden images
 * or fixtures, cannot be compiled here (Swift + Apple-only `simd`/`Metal` imports, no swift
 * toolchain), and its float 2x2 `inverse` / 4x4 `*` come from Apple's closed `simd` module.
 * This file follows the source text of renderer/Renderer.swift operation by operation; it is
 * pinned only by the hand-derived known answers of SURVEY.md Appendix C and by an independent
 * NumPy restatement (oracle/swr_oracle_np.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the reported CPU baseline.  The product (libswr_hip.so) never
 * links, loads or calls it.
 */
#ifndef SWR_ORACLE_H_
#define SWR_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as swr_vertex (Renderer.swift:154-157). */
typedef struct swro_vertex {
    float xyz[4];
    float color[4];
} swro_vertex;

enum {
    SWRO_DEPTH_TEST = 1u << 0,  /* restore Renderer.swift:257-261 */
    SWRO_NO_COLOR   = 1u << 1,  /* depth-only: colour image untouched */
    SWRO_REAL_LINES = 1u << 3,  /* swro_render_primitives, .line only: the DDA of Renderer.swift:405-419 between the two
                                   truncated endpoints instead of the empty stub of :289-293 (= SWR_FLAG_REAL_LINES) */
    /* oracle-only switches */
    SWRO_INV_RCP    = 1u << 8,  /* 2x2 inverse as adj * (1/det) instead of adj / det */
    SWRO_UNCLAMPED  = 1u << 9,  /* iterate rows/pixels exactly as Renderer.swift:275-283 does
                                   (off-screen included) and reject per pixel (:246-250) */
    SWRO_TINV_PER_TRIANGLE = 1u << 10, /* hoist T() out of the pixel loop (same values; the
                                   reference recomputes it per pixel, Renderer.swift:251-252) */
    SWRO_FMA_TRANSFORM = 1u << 11 /* Vertex.apply (Renderer.swift:160) with fused multiply-adds after the first
                                   column — the other plausible arm64 lowering of Apple's simd_mul; a sensitivity
                                   switch, never the parity target */
};

typedef struct swro_stats {
    int64_t fragments;        /* setPixel calls that passed the bounds check */
    int64_t fragments_written;/* ... that also passed the z-test (== fragments when off) */
    int64_t triangles_drawn;
    int64_t triangles_skipped;/* non-finite / out-of-range screen coordinate (documented deviation) */
} swro_stats;

/* Renderer.render(renderPass:) restricted to rows [row_begin,row_end) of the W x H images.
 * color: W*H*4 bytes BGRA (may be NULL with SWRO_NO_COLOR); depth: W*H floats.
 * Returns 0, or a negative code with the same meaning as swr.h's. */
int swro_render(uint8_t* color, float* depth, int64_t W, int64_t H,
                const swro_vertex* vertices, int64_t vertex_count,
                const int64_t* indices, int64_t index_count,
                const float transform[16], uint32_t flags,
                int64_t row_begin, int64_t row_end, swro_stats* stats);

/* Vertex stage alone: screen x / y (before truncation) and NDC z of every vertex (flags: SWRO_FMA_TRANSFORM). */
void swro_project(const swro_vertex* vertices, int64_t vertex_count, const float transform[16], int64_t W, int64_t H,
                  uint32_t flags, float* sx, float* sy, float* sz);

/* Same, with RenderPass.primitiveType (Renderer.swift:174-189, :210-219):
 *   0 = .triangle  -> swro_render
 *   1 = .line      -> draw(line:) is an empty stub (Renderer.swift:289-293): the frame is only cleared;
 *                     index_count must be a multiple of 2 (verticesCount, :183-184)
 *   2 = .vertices  -> draw(vertices:) (Renderer.swift:295-302): every transformed vertex of every
 *                     3-index primitive is plotted at (Int(sx), Int(sy)) with its own colour, in index
 *                     order (later vertices overwrite), no z; off-screen points are dropped (:30-36).
 *                     Deviation: a non-finite / |coord| >= 2^30 point is skipped (Swift would trap). */
int swro_render_primitives(uint8_t* color, float* depth, int64_t W, int64_t H,
                           const swro_vertex* vertices, int64_t vertex_count,
                           const int64_t* indices, int64_t index_count,
                           const float transform[16], uint32_t flags, int32_t primitive_type,
                           int64_t row_begin, int64_t row_end, swro_stats* stats);

/* The Metal path's rules (renderer/Shaders.metal:57-167 + renderer/GpuRenderer.swift:109-139) in IEEE
 * arithmetic — SURVEY.md §A.3, §8(f) rank 1.  PARITY UNPINNED twice over: the reference compiles its
 * shaders with MTL_FAST_MATH = YES (project.pbxproj:221,278), so the Metal binary is not
 * bit-reproducible even against itself; this restates the source text with one IEEE op per operator.
 *   vertex_pass   :57-75   pos = M*(xyz,1); pos.xyz /= pos.w; pos.xy = round(uv * screen)  (half away from 0)
 *   roi_pass      :89-114  bbox of the uint-cast snapped vertices
 *   host          GpuRenderer.swift:122-124  primitives whose ROI min-x or min-y is 0 are skipped
 *   rasterizer    :123-167 one thread per bbox pixel, sample at +0.5; barycentrics by the `divider`
 *                          formula; inside = all(0 <= ws <= 1); z = ws.pos.z; strict '<' z-test;
 *                          colour = ws.colours through fragment_shader; bgra8Unorm store
 * Documented choices: a negative / non-finite / >= 2^30 snapped coordinate (uint cast undefined in
 * MSL) skips the primitive; bbox pixels outside the textures are dropped; float -> unorm8 is
 * rint(clamp(v,0,1)*255) (round to nearest even, the MSL-preferred conversion).
 * flags: SWRO_NO_COLOR honoured; the z-test is always on (it is in the shader). */
int swro_render_metal(uint8_t* color, float* depth, int64_t W, int64_t H,
                      const swro_vertex* vertices, int64_t vertex_count,
                      const int64_t* indices, int64_t index_count,
                      const float transform[16], uint32_t flags,
                      int64_t row_begin, int64_t row_end, swro_stats* stats);

/* ---- fragment-stage extensions (include/swr.h swr_vertex_attr / swr_material) ----
 * NOT IN THE REFERENCE: its fragment stage is float4(color, 1) (Shaders.metal:116-121).  This is the
 * normative definition of the extended stage; the HIP kernels are checked against it bit for bit.
 * Varyings normal / uv are interpolated with the same weights and expression shape as colour
 * (a*w0 + b*w1 + c*w2, Renderer.swift:266).  swro_fragment, one IEEE binary32 op per operator:
 *   len2 = nx*nx + ny*ny + nz*nz;  N = len2 > 0 ? n / sqrtf(len2) : 0
 *   ndl = fmaxf(N.x*L.x + N.y*L.y + N.z*L.z, 0);  ndh = fmaxf(N.H, 0);  s = ndh squared shininess_log2 times
 *   base = color                      (shader 1)
 *        = color * bilinear(tex, uv)  (shader 2; repeat addressing, texel centres at +0.5, channel/255)
 *   rgb = base * (ambient + diffuse*ndl) + specular*s;  a = 1 */
typedef struct swro_vertex_attr {
    float normal[4];
    float uv[4];
} swro_vertex_attr;
typedef struct swro_material {
    int32_t shader;
    int32_t shininess_log2;
    float light_dir[4];
    float half_dir[4];
    float ambient, diffuse, specular, reserved;
} swro_material;
typedef struct swro_shading {
    const swro_vertex_attr* attrs;   /* vertex_count entries */
    swro_material material;
    const uint8_t* texture;          /* tex_w*tex_h*4 bytes b,g,r,a */
    int32_t tex_w, tex_h;
} swro_shading;

void swro_fragment(const swro_shading* sh, const float color[3], const float normal[3], const float uv[2],
                   float out_rgba[4]);

/* swro_render / swro_render_metal with the extended fragment stage (sh == NULL or shader 0: identical to
 * the plain entry points). */
int swro_render_shaded(uint8_t* color, float* depth, int64_t W, int64_t H,
                       const swro_vertex* vertices, int64_t vertex_count,
                       const int64_t* indices, int64_t index_count,
                       const float transform[16], uint32_t flags,
                       int64_t row_begin, int64_t row_end, swro_stats* stats, const swro_shading* sh);
int swro_render_metal_shaded(uint8_t* color, float* depth, int64_t W, int64_t H,
                             const swro_vertex* vertices, int64_t vertex_count,
                             const int64_t* indices, int64_t index_count,
                             const float transform[16], uint32_t flags,
                             int64_t row_begin, int64_t row_end, swro_stats* stats, const swro_shading* sh);

/* Renderer.interpolate(values:t:) (Renderer.swift:467-494), exposed for unit tests.
 * pts = n (x,y) pairs, n in {2,3}. */
int64_t swro_interpolate(const int64_t* pts_xy, int n, int64_t t);

/* Pixel.floats channel quantiser (Renderer.swift:116-124). */
uint8_t swro_quantise(float v);

#ifdef __cplusplus
}
#endif
#endif
