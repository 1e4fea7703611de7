/*
 * swr.h — C-ABI of the MI355X-native triangle rasterizer (drop-in boundary).
 *
 * This header is the whole boundary: plain pointers and sizes, no C++ / torch / HIP
 * types.  Every entry point names the reference interface (file:line relative to
 * zhvrnkov/software-renderer) that it stands in for.  The shared library that exports
 * these symbols is `software-renderer_amd/lib/libswr_hip.so` (built from
 * `software-renderer_amd/csrc/` with hipcc for gfx950).  There is no CPU fallback inside
 * the library: every entry point that computes needs a HIP device and returns
 * SWR_ERR_HIP (with a message in swr_last_error) when there is none.
 *
 * Semantics are those of the reference's CPU renderer (renderer/Renderer.swift:204-287,
 * 467-494, 88-100, 116-129, 159-171) — see DESIGN.md §2 for the normative restatement.
 */
#ifndef SWR_H_
#define SWR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWR_ABI_VERSION 5

/* ---- status codes (the reference has no error channel: it fatalError()s / try!s,
 *      Renderer.swift:26,209,239,497; GpuRenderer.swift:20-31,37-38) ------------------ */
enum {
    SWR_OK = 0,
    SWR_ERR_BAD_ARG = -1,       /* null pointer, non-positive size, bad band */
    SWR_ERR_INDEX_COUNT = -2,   /* index_count % verticesCount != 0   (assert, Renderer.swift:209) */
    SWR_ERR_INDEX_RANGE = -3,   /* an index outside [0, vertex_count)  (Swift array trap, Renderer.swift:226) */
    SWR_ERR_HIP = -4,           /* HIP runtime error / no device */
    SWR_ERR_UNSUPPORTED = -5,   /* unknown primitive type, too many primitives / vertices */
    SWR_ERR_NO_SCENE = -6,      /* swr_draw before swr_scene_upload / swr_target_set */
    SWR_ERR_NOMEM = -7,
    SWR_ERR_FRAME_DROPPED = -8  /* a frame that was copied to the host (swr_present) in an un-waited burst had overflowed
                                   the (triangle,tile) bins and was rastered empty; the bins have been grown, redraw it.
                                   (The last frame of a burst is always repaired silently; frames never presented are
                                   never reported: nobody could see them.) */
};

/* ---- PrimitiveType (Renderer.swift:174-189).  Only .triangle is on the hot path. ---- */
enum {
    SWR_PRIMITIVE_TRIANGLE = 0,
    SWR_PRIMITIVE_LINE = 1,      /* reference draw(line:) is an empty stub, Renderer.swift:289-293 */
    SWR_PRIMITIVE_VERTICES = 2   /* Renderer.swift:295-302: point plotting */
};

/* ---- draw flags --------------------------------------------------------------------- */
enum {
    /* 0 = the CPU renderer exactly as written: painter's order (highest primitive index
     * covering a pixel wins), depth image stays +inf (z-test commented out,
     * Renderer.swift:257-261). */
    SWR_FLAG_DEPTH_TEST = 1u << 0, /* restore Renderer.swift:257-261: depth = za*w0+zb*w1+zc*w2,
                                      strict '<', first-drawn wins ties (also :344-348,
                                      Shaders.metal:158-165) */
    SWR_FLAG_NO_COLOR   = 1u << 1, /* depth-only pass: the colour image is neither cleared nor
                                      written (BASELINE config 4) */
    SWR_FLAG_METAL_RULES = 1u << 2, /* the Metal path's rules instead of the CPU renderer's:
                                      vertices snapped with round() (Shaders.metal:71), one thread per ROI
                                      pixel with inside = all(0 <= ws <= 1) (:133-153), z-test always on
                                      (:158-161), bgra8Unorm store (round to nearest), ROIs whose min-x or
                                      min-y is 0 skipped (GpuRenderer.swift:122-124); IEEE arithmetic (the
                                      reference's MTL_FAST_MATH build is not bit-reproducible) */
    SWR_FLAG_REAL_LINES = 1u << 3  /* OPT-IN, .line primitives only.  Default (flag clear) = the reference as written:
                                      draw(line:colorBuffer:depthBuffer:) has an empty body (Renderer.swift:289-293), a
                                      .line pass only clears.  With the flag every 2-index primitive is drawn with the
                                      reference's own DDA (draw(line:with:in:), Renderer.swift:405-419) between its two
                                      transformed endpoints, truncated like .vertices does (:298-299): steps =
                                      max(|dx|, |dy|), float x / y advanced by dx/steps, dy/steps, pixel
                                      (Int(x.rounded()), Int(y.rounded())) for steps iterations (the end point itself is
                                      not plotted), in the colour of the FIRST vertex, later primitives overwrite
                                      earlier ones, no z-test, depth stays +inf.  Lines longer than 2^20 steps or with a
                                      non-finite endpoint are skipped (the reference would trap / never finish) */
};

/* ---- Vertex (Renderer.swift:154-157): two SIMD3<Float>, each padded to 16 B --------- */
typedef struct swr_vertex {
    float xyz[4];    /* x,y,z, lane 3 = padding (ignored) */
    float color[4];  /* r,g,b, lane 3 = padding (ignored) */
} swr_vertex;        /* 32 bytes */

/* ---- fragment-stage extensions (BASELINE configs 3 and 5) -------------------
 * The reference's fragment stage returns the interpolated vertex colour (Shaders.metal:116-121) and has
 * no normals, texture coordinates, lights or textures.  These additions sit behind the same
 * fragment_shader(VertexOut) hook: VertexOut gains `normal` and `uv` varyings, interpolated exactly like
 * `color` (screen-affine barycentric weights, Renderer.swift:266 / Shaders.metal:162), and the hook
 * evaluates a Blinn-Phong model per pixel.  Every operation is a single IEEE binary32 + - * / sqrt in a
 * fixed order, so the CPU oracle and the HIP kernels agree bit for bit; parity for these
 * modes is build-internal (there is nothing in the reference to compare with). */
typedef struct swr_vertex_attr {
    float normal[4]; /* nx,ny,nz (any length; normalised per pixel), lane 3 = padding */
    float uv[4];     /* u,v (repeat addressing), lanes 2,3 = padding */
} swr_vertex_attr;   /* 32 bytes, parallel to swr_vertex: attribute i belongs to vertex i */

enum {
    SWR_SHADER_PASSTHROUGH = 0,    /* float4(vin.color, 1)                      (Shaders.metal:116-121) */
    SWR_SHADER_PHONG = 1,          /* base = vin.color                          (BASELINE config 3)     */
    SWR_SHADER_TEXTURED_PHONG = 2  /* base = vin.color * bilinear(texture, uv)  (BASELINE config 5)     */
};

/* rgb = base * (ambient + diffuse * max(N.L, 0)) + specular * max(N.H, 0)^(2^shininess_log2),  a = 1,
 * N = vin.normal / |vin.normal| (zero vector when the length is 0).  light_dir and half_dir are given by
 * the caller in the space of the normals (object space: normals are passed through untransformed, like
 * colours, Shaders.metal:53); half_dir is the Blinn half vector of a distant light and viewer. */
typedef struct swr_material {
    int32_t shader;          /* SWR_SHADER_* */
    int32_t shininess_log2;  /* 0..16: the exponent is a power of two, evaluated by repeated squaring */
    float   light_dir[4];    /* unit vector towards the light, lane 3 ignored */
    float   half_dir[4];     /* unit half vector, lane 3 ignored */
    float   ambient, diffuse, specular, reserved;
} swr_material;              /* 56 bytes */

/* ---- RenderPass (Renderer.swift:191-200) + Image<T> (Renderer.swift:8-21) ------------
 * color: Pixel = {b,g,r,a} u8 (Renderer.swift:44-49); element (x,y) at color[y*width+x]
 * (App.swift:351-360: addressing uses width; bytesPerRow is stored but never read, so
 * the two *_bytes_per_row fields are carried for layout parity and ignored). */
typedef struct swr_render_pass {
    void*    color;                 /* width*height*4 bytes, BGRA8; may be NULL iff SWR_FLAG_NO_COLOR */
    float*   depth;                 /* width*height floats */
    int64_t  width;
    int64_t  height;
    int64_t  color_bytes_per_row;   /* ignored, see above */
    int64_t  depth_bytes_per_row;   /* ignored */
    const swr_vertex* vertices;
    int64_t  vertex_count;
    const int64_t* indices;         /* Swift Int (Renderer.swift:196) */
    int64_t  index_count;
    int32_t  primitive_type;        /* SWR_PRIMITIVE_* ; default .triangle (Renderer.swift:197) */
    uint32_t flags;                 /* SWR_FLAG_* */
    float    transform[16];         /* matrix_float4x4, column-major: column c = transform[4c..4c+3]
                                       (Renderer.swift:199) */
    /* extensions (all optional; NULL = the reference's passthrough fragment stage) */
    const swr_vertex_attr* attributes;  /* vertex_count entries */
    const swr_material*    material;
    const void*            texture;     /* tex_width*tex_height Pixels (b,g,r,a), row-major; SWR_SHADER_TEXTURED_PHONG */
    int32_t  tex_width, tex_height;
    /* Scene identity (ABI 4).  The reference's only caller draws the SAME mesh every display frame with a new transform
     * (App.swift:153-185) and its GpuRenderer keeps its device buffers across calls (GpuRenderer.swift:32-33,41-67).
     * 0 = no promise: vertices / indices / attributes / texture are uploaded and the device-side triangle stream is
     * rebuilt on every call (what ABI 3 did) — as a scene that lives for ONE frame: in index order (no Morton sort),
     * built behind the copy of the index array.  Non-zero = the caller promises that the CONTENT of `vertices`,
     * `indices`, `attributes` and `texture` (and their counts / sizes) equals that of the last swr_render on this
     * context that carried the same id: the upload is then skipped and the pass costs one resident frame plus the
     * gather.  A new id (or new counts) uploads.  transform, flags, primitive_type, material and the image pointers
     * may change freely between calls with the same id. */
    uint64_t scene_id;
} swr_render_pass;      /* 192 bytes */

typedef struct swr_config {
    int32_t  device;        /* HIP device ordinal (of the first device); -1 = current device */
    uint32_t device_count;  /* 0 or 1: one GPU.  N > 1: the context drives N tile-row bands of the framebuffer, band k
                               on device (device + k) % min(N, visible devices), one host thread and one set of HIP
                               streams per band, scene replicated, no collective.  With fewer
                               visible GPUs than N, several bands share a GPU (same code path). */
    uint32_t wait_budget_ms;/* Longest time any single wait inside the library may take (a helper thread polling for a
                               kernel's completion, a blocking call waiting for the streams): 0 = default (20 000 ms).
                               When it expires the context fails for good: the blocking call returns SWR_ERR_HIP with the
                               frame number and what was being waited for, every later call returns the same error at
                               once, swr_context_destroy still returns (it may leak what the GPU still owns).  The
                               reference has no counterpart: scheduleAndWait blocks forever (Metal+Extensions.swift:57-67). */
    uint32_t reserved;      /* 0 */
} swr_config;               /* 16 bytes */

/* Per-kernel device times of the last swr_draw / swr_render on this context, measured with
 * hipEvents recorded on the context's own stream (only filled when timing is enabled). */
typedef struct swr_timings {
    float setup_bin_ms;   /* vertex transform + triangle setup + tile binning (count/emit) */
    float scan_ms;        /* per-tile offsets */
    float scatter_ms;     /* bin fill */
    float raster_ms;      /* tile raster + resolve + framebuffer write (the dominant kernel) */
    float total_ms;       /* first kernel start -> last kernel end */
    int64_t tile_pairs;   /* (triangle,tile) pairs binned in the frame */
    int64_t tiles;        /* tiles in the band */
    int64_t triangles;    /* primitives submitted */
} swr_timings;

typedef struct swr_context swr_context;

/* Library / ABI identification; number of visible HIP devices (0 when there is none or no driver). */
int         swr_abi_version(void);
const char* swr_version(void);
int         swr_device_count(void);

/* GpuRenderer() / Renderer() default initialisers (App.swift:148-149) + MTLContext.shared
 * (Metal+Extensions.swift:5-45): device(s), streams, cached scratch buffers.  The reference's one synchronous
 * draw call (GpuRenderer.swift:35, caller App.swift:185) maps onto ONE context whatever the number of GPUs:
 * with cfg->device_count = N every entry point below fans out to N per-device sub-contexts. */
int  swr_context_create(const swr_config* cfg, swr_context** out);
void swr_context_destroy(swr_context* ctx);
/* Number of bands (sub-contexts) of the context, and where band `band` lives: its HIP device and the rows
 * [row_begin,row_end) it owns after swr_target_set.  Any out pointer may be NULL. */
int  swr_context_bands(const swr_context* ctx);
int  swr_context_band_info(const swr_context* ctx, int32_t band, int32_t* device, int64_t* row_begin, int64_t* row_end);

/* Last error text for this context (NULL ctx: last error of a failed swr_context_create).  Call it from the thread that
 * made the failing call; the text stays valid until that thread's next call on the context. */
const char* swr_last_error(const swr_context* ctx);

/* Wall-clock phases of the last swr_render on this context (host steady clock; the H2D / stream-build split inside the
 * upload comes from HIP events).  scene_cached = 1 when the pass carried the scene_id of the resident scene and nothing
 * was uploaded. */
typedef struct swr_render_times {
    float h2d_ms;           /* vertices / indices / attributes / texture: host -> device */
    float stream_build_ms;  /* index check, Morton order, de-indexed triangle stream, group boxes */
    float draw_ms;          /* target + one frame, until the raster has finished */
    float gather_ms;        /* swr_present + swr_present_wait: bands -> the caller's images */
    float total_ms;
    int32_t scene_cached;
    int32_t frames;         /* frames the call drew: 1, or 2 when a tile overflowed its bin region and the frame was redrawn (ABI 5;
                               was `reserved`) */
} swr_render_times;
int swr_render_timings(swr_context* ctx, swr_render_times* out);

/* Fault injection for the failure-path tests (tests/test_gpu_api.py, tests/host): the NEXT frame's raster share
 *   SWR_FAULT_LOST_EVENT   waits for a completion that never arrives (the wait budget must end it),
 *   SWR_FAULT_ENQUEUE      fails as if a HIP launch had returned an error.
 * Either way the context ends up failed (see swr_config.wait_budget_ms).  Never needed by a renderer. */
enum { SWR_FAULT_NONE = 0, SWR_FAULT_LOST_EVENT = 1, SWR_FAULT_ENQUEUE = 2 };
int swr_debug_fault(swr_context* ctx, int fault);

/* Test hooks (ABI 5; until round 4 these were environment variables read by the product's hot-path setup): force a code
 * path the library would otherwise choose by itself, for THIS context.  A hook takes effect at the next swr_scene_upload /
 * swr_target_set / swr_render (the stream order, the bin layout, the one-shot threshold) or at the next frame (the others);
 * results never depend on them — every parity test that sets one compares against the oracle.  Never needed by a renderer.
 *   SWR_DEBUG_STREAM_ORDER      1 (default) Morton-ordered triangle stream; 0 keep the caller's primitive order;
 *                               -1 behave as for scenes of >= 2^24 primitives (no reordering, slot == index)
 *   SWR_DEBUG_CULL              1 (default) per-band culling of 64-primitive groups; 0 off
 *   SWR_DEBUG_BIN_MODE          0 (default) fixed-stride bins (the single-launch k_bin) wherever they can hold the scene, exact-size
 *                               bins otherwise; 1 always exact-size bins (four-kernel chain); 2 = 0; 3 global-atomic binning fallback
 *   SWR_DEBUG_ONESHOT_MIN_TRIS  primitives from which a swr_render without a scene identity cuts its index copy in two
 *                               (default 2^18; minimum 64)
 *   SWR_DEBUG_DEPTH_KEYS32      1 (default) depth-only z-tested frames take 32-bit depth keys (k_raster_depth); 0 the 64-bit kernel
 *   SWR_DEBUG_RASTER_SORT       1 (default) such frames sort their bins inside the raster workgroups on small tile grids (thin
 *                               bands), by a k_sort_bins launch on large ones; 0 always the launch; 2 always inside the raster */
enum { SWR_DEBUG_STREAM_ORDER = 1, SWR_DEBUG_CULL = 2, SWR_DEBUG_BIN_MODE = 3, SWR_DEBUG_ONESHOT_MIN_TRIS = 4,
       SWR_DEBUG_DEPTH_KEYS32 = 5, SWR_DEBUG_RASTER_SORT = 6 };
int swr_debug_set(swr_context* ctx, int key, int64_t value);

/* Renderer.render(renderPass:) (Renderer.swift:204-230) and GpuRenderer.render(renderPass:)
 * (GpuRenderer.swift:35-90): caller-owned host memory in, colour + depth images filled on
 * return (synchronous, like scheduleAndWait, Metal+Extensions.swift:57-67). */
int swr_render(swr_context* ctx, const swr_render_pass* pass);

/* ---- resident path: what the app's frame loop does (App.swift:153-185) — same mesh every
 * frame, new transform — without re-uploading; inputs and outputs stay in HBM. ----------- */

/* RenderPass.vertices / .indices (Renderer.swift:195-196); replaces the per-frame
 * makeBuffer / setBytes uploads of GpuRenderer.swift:68-71,93-103.  Validates indices.
 * Also builds the device-side triangle stream once (primitives de-indexed and ordered by the Morton code
 * of their centroid, bounding boxes of 64-primitive groups) — invisible in the image:
 * painter's order and z-tie order always refer to the caller's index order. */
int swr_scene_upload(swr_context* ctx, const swr_vertex* vertices, int64_t vertex_count,
                     const int64_t* indices, int64_t index_count);

/* Fragment-stage extensions on the resident path.  swr_scene_attributes must follow the swr_scene_upload
 * it belongs to (vertex_count must match; a new swr_scene_upload discards the attributes).  The material
 * and the texture persist on the context until replaced; swr_material_set(ctx, NULL) restores the
 * reference's passthrough stage.  A draw with a Phong material but no attributes, or a textured material
 * but no texture, returns SWR_ERR_BAD_ARG. */
int swr_scene_attributes(swr_context* ctx, const swr_vertex_attr* attributes, int64_t vertex_count);
int swr_material_set(swr_context* ctx, const swr_material* material);
int swr_texture_upload(swr_context* ctx, const void* bgra8, int32_t width, int32_t height);

/* colorBuffer / depthBuffer size (Renderer.swift:192-193).  row_begin/row_end select the
 * tile-row band [row_begin,row_end) of the framebuffer this context owns; pass
 * 0,height for the whole image.  row_begin must be a multiple of swr_tile_rows().  A multi-device
 * context cuts [row_begin,row_end) into device_count bands of whole tile rows (like swr_band_rows);
 * a band may be empty when there are more devices than tile rows. */
int swr_target_set(swr_context* ctx, int64_t width, int64_t height,
                   int64_t row_begin, int64_t row_end);

/* One frame: clear + all triangles (Renderer.swift:204-230) into the device-resident band(s).
 * Asynchronous: the call validates its arguments and posts the frame to the context's helper threads, which
 * enqueue it on the HIP streams; swr_sync(), swr_present_wait() or a swr_read_* completes it.  A HIP failure while
 * enqueueing is returned by the next of those blocking calls; on a multi-device context the same holds for an
 * error of the draw itself (no scene, bad index count, ...). */
int swr_draw(swr_context* ctx, const float transform[16], uint32_t flags);
/* Same with RenderPass.primitiveType (Renderer.swift:197, :210-219): .triangle = swr_draw;
 * .vertices plots every vertex reference as a point (Renderer.swift:295-302); .line clears only
 * (the reference's draw(line:) is an empty stub, Renderer.swift:289-293). */
int swr_draw_primitives(swr_context* ctx, const float transform[16], uint32_t flags, int32_t primitive_type);
int swr_sync(swr_context* ctx);

/* ---- host-visible frames: the gather ("final image gathered with pinned hipMemcpyAsync") -------------------
 * The reference's images live in CPU/GPU-shared MTLBuffers (App.swift:59-60,80-101) and are complete on return
 * of render (scheduleAndWait, Metal+Extensions.swift:57-67).  Here every band is copied device -> host into its
 * rows of the caller's ONE full-size image; bands are disjoint, so there is nothing to merge.
 *
 * swr_host_alloc / swr_host_free: page-locked host memory every GPU can DMA into — allocate the colour and depth
 *   images with it (what makeBuffer(.storageModeShared) is to the reference).  swr_host_register /
 *   swr_host_unregister page-lock memory the caller already owns.  (Process-wide, no context needed.)
 *
 * swr_present(ctx, color_full, depth_full): enqueue the copy of the frame of the LAST swr_draw — rows
 *   [row_begin,row_end) of each band — into the caller's full-size images and return at once.  Per device: one
 *   hipMemcpyAsync per image, colour and depth in flight together on two copy streams, behind that frame's
 *   raster.  The device framebuffers are double-buffered: the next swr_draw renders into the other one, so the
 *   copy of frame N overlaps the raster of frame N+1.  Either pointer may be NULL (image not wanted; colour is
 *   skipped for SWR_FLAG_NO_COLOR frames).  A destination that is not page-locked still works: it is staged
 *   through pinned 8 MiB chunks by the context's helper thread (or, without helper threads, by the caller).
 * swr_present_wait(ctx): returns when every enqueued copy has landed: the pixels are host-visible.
 *
 * swr_read_color / swr_read_depth: swr_sync + the same copy of one image + wait (rows outside the band(s) are
 *   left untouched).  swr_render = upload + swr_draw + swr_present + swr_present_wait. */
void* swr_host_alloc(size_t bytes);
void  swr_host_free(void* p);
int   swr_host_register(void* p, size_t bytes);
int   swr_host_unregister(void* p);
int swr_present(swr_context* ctx, void* color_full_image, float* depth_full_image);
int swr_present_wait(swr_context* ctx);
int swr_read_color(swr_context* ctx, void* dst_full_image);
int swr_read_depth(swr_context* ctx, float* dst_full_image);

/* Timing instrumentation: hipEvents on the context stream.  level 0 = off, 1 = two events around
 * the dominant kernel (k_raster) only, 2 = around every stage (each event costs a few us of
 * stream time, so level 2 perturbs the frame it measures). */
int swr_timing_enable(swr_context* ctx, int level);
/* Level 1 only: bracket the k_raster of every n-th frame instead of every frame (default 1).  An event pair on
 * the raster stream is a synchronisation point that costs a pipelined 4K frame about 15 us; sampling keeps the
 * measurement live inside a timed region without slowing every frame of it.  swr_timing_totals' frame count is
 * the number of frames actually bracketed. */
int swr_timing_sample(swr_context* ctx, int every_nth);
int swr_get_timings(swr_context* ctx, swr_timings* out);          /* the last frame */
/* Sums over every frame drawn since swr_timing_reset (events are kept in a ring, so a whole
 * timed region of frames is measured without a host sync per frame). */
int swr_timing_totals(swr_context* ctx, swr_timings* sum_out, int64_t* frames_out);
int swr_timing_reset(swr_context* ctx);

/* Frame pipelining (on by default): the binning kernels of the next swr_draw run on a second stream,
 * over a triple-buffered working set, while the previous frame is still being rasterised.  Results
 * are identical either way; 0 serialises the two stages on one stream (clean per-stage timings). */
int swr_pipeline_enable(swr_context* ctx, int enable);

/* Tile geometry the band boundaries must respect. */
int swr_tile_rows(void);
int swr_tile_cols(void);

/* Helper: split `height` rows into `parts` contiguous bands aligned to swr_tile_rows();
 * writes row_begin/row_end of band `part`.  Pure host arithmetic (no device needed). */
int swr_band_rows(int64_t height, int32_t parts, int32_t part,
                  int64_t* row_begin, int64_t* row_end);

#ifdef __cplusplus
}
#endif
#endif /* SWR_H_ */
