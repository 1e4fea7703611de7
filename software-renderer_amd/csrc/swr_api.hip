// swr_api.hip — the C-ABI of include/swr.h over the gfx950 kernels (swr_kernels.hip).
//
// Stands in for GpuRenderer.render(renderPass:) (renderer/GpuRenderer.swift:35-141): where the
// reference allocates shared MTLBuffers, encodes one compute dispatch per triangle and blocks
// twice per frame in scheduleAndWait, this context keeps scene, records, bins and the
// framebuffer band resident in HBM and enqueues four kernels on one HIP stream per frame.
// There is no CPU fallback: without a HIP device every computing entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "swr_internal.h"

using namespace swr;

namespace {
thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};
}  // namespace

static_assert(sizeof(swr_render_pass) == 184 && sizeof(swr_material) == 56 && sizeof(swr_vertex_attr) == 32 &&
              sizeof(swr_vertex) == 32, "include/swr.h layouts (mirrored by the ctypes / Swift bindings)");

struct swr_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // scene (RenderPass.vertices / .indices)
    DevBuf vertices, indices, xyz, rgb, idx32, tri_rgb;
    // the triangle stream built at upload (swr_upload.hip)
    DevBuf tri_xyz, inv, box64, stream_scratch, sort_temp;
    bool reordered = false;
    int64_t nv = 0, ni = 0;
    bool has_scene = false;
    // extended fragment stage (swr_scene_attributes / swr_material_set / swr_texture_upload)
    DevBuf attrs, tri_nrm, texture, texture_bytes;   // texture: float4 texels; texture_bytes: upload staging
    bool has_attrs = false;
    swr_material material{};            // shader 0 = the reference's passthrough stage
    int32_t tex_w = 0, tex_h = 0;

    // target band (RenderPass.colorBuffer / .depthBuffer)
    Target tg{};
    bool has_target = false;
    DevBuf color, depth;
#ifndef SWR_NSLOT
#define SWR_NSLOT 3   // working sets in flight: binning may run up to two frames ahead of the raster (2 -> 3: -3 %)
#endif
    static constexpr int NSLOT = SWR_NSLOT;
    // Per-frame working set, multi-buffered: the binning kernels of frame N+1 run on `bin_stream`
    // while k_raster of frame N runs on `stream` (HBM-bound vs LDS/VALU-bound: they overlap well).
    struct Slot {
        DevBuf geo, geo_full, ranges, bins, bin_matrix;
        DevBuf live;           // per binning workgroup: count + surviving stream-group ids (k_setup_hist -> k_fill_lds)
        DevBuf tilebuf;        // [CNT_WORDS counters][tiles tile_count][tiles+1 tile_start][tiles cursor]
        hipEvent_t bin_done = nullptr, ras_done = nullptr;
        bool ras_recorded = false;
    } slot[NSLOT];
    hipStream_t bin_stream = nullptr;   // the stream binning is enqueued on (== stream when pipelining is off)
    hipStream_t bin_stream_own = nullptr;
    uint64_t frame_no = 0;
    int last_slot = 0;
    uint32_t capacity = 0;

    uint32_t* h_counters = nullptr;   // pinned, mapped into the device address space
    uint32_t* h_counters_dev = nullptr;
    // last draw (for the overflow redo and for swr_render)
    float last_m[16]{};
    uint32_t last_flags = 0;
    int last_prim = SWR_PRIMITIVE_TRIANGLE;
    bool draw_pending = false;

    // timing: a ring of hipEvent sets recorded on the context stream around each kernel, so a
    // whole timed region of frames can be measured without a host sync per frame
    static constexpr int RING = 64;
    int timing = 0;            // 0 off, 1 = events around k_raster only, 2 = around every stage
    int timing_every = 1;      // level 1: bracket only every n-th frame's k_raster (swr_timing_sample)
    hipEvent_t ev[RING][5]{};
    int ev_level[RING]{};
    bool ev_ok = false;
    uint64_t seq = 0;          // frames enqueued with timing on
    uint64_t harvested = 0;    // frames whose events have been read
    swr_timings last{};
    double sum_ms[5]{};
    int64_t sum_frames = 0;
};

namespace {

int fail(swr_context* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctx, SWR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

int ensure(swr_context* c, DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return SWR_OK;
    if (b.p) { HIP_TRY(c, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    // grow geometrically so a slowly growing scene does not reallocate every frame
    size_t want = bytes + bytes / 8;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(c, SWR_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.bytes = want;
    return SWR_OK;
}

int ensure_capacity(swr_context* c, uint32_t cap) {
    if (cap <= c->capacity) return SWR_OK;
    int rc;
    for (auto& sl : c->slot)
        if ((rc = ensure(c, sl.bins, (size_t)cap * 4))) return rc;
    c->capacity = cap;
    return SWR_OK;
}

int sync_streams(swr_context* c) {
    HIP_TRY(c, hipStreamSynchronize(c->bin_stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SWR_OK;
}

inline int tiles_of(const Target& t) { return t.tiles_x * t.tiles_y; }

DeviceFrame make_frame(swr_context* c, int si, const float m[16], uint32_t flags) {
    swr_context::Slot& sl = c->slot[si];
    DeviceFrame f{};
    f.vertices = (const swr_vertex*)c->vertices.p;
    f.indices = (const int64_t*)c->indices.p;
    f.xyz = (const float4*)c->xyz.p;
    f.rgb = (const float4*)c->rgb.p;
    f.idx32 = (const uint32_t*)c->idx32.p;
    f.tri_rgb = (const float4*)c->tri_rgb.p;
    f.tri_xyz = (const float4*)c->tri_xyz.p;
    f.inv = (const uint32_t*)c->inv.p;
    f.box64 = (const float4*)c->box64.p;
    f.reordered = c->reordered ? 1 : 0;
    f.tri_nrm = (const float4*)c->tri_nrm.p;
    f.material = c->material;
    f.texels = (const float4*)c->texture.p;
    f.tex_w = c->tex_w; f.tex_h = c->tex_h;
    f.vertex_count = c->nv;
    f.ntri = c->ni / 3;
    f.geo = (GeomRec*)sl.geo.p;
    f.geo_full = (GeomFull*)sl.geo_full.p;
    uint32_t* tb = (uint32_t*)sl.tilebuf.p;
    f.counters = tb;
    f.host_counters = c->h_counters_dev;
    f.tile_count = tb + CNT_WORDS;
    f.tile_start = tb + CNT_WORDS + tiles_of(c->tg);
    f.tile_cursor = tb + CNT_WORDS + 2 * tiles_of(c->tg) + 1;
    f.ranges = (uint2*)sl.ranges.p;
    f.plan = plan_binning(f.ntri, tiles_of(c->tg));
    f.bin_matrix = (uint32_t*)sl.bin_matrix.p;
    f.live = (uint32_t*)sl.live.p;
    // Groups of the stream whose projected box misses this context's band (or the framebuffer) are skipped by
    // k_setup_hist; the test costs one lane-pass per workgroup, so it is always on (SWR_CULL=0 switches it off).
    static const int cull_mode = getenv("SWR_CULL") ? atoi(getenv("SWR_CULL")) : 1;
    f.live_parity = cull_mode != 0 ? 0 : -1;
    f.bins = (uint32_t*)sl.bins.p;
    f.capacity = c->capacity;
    f.color = (uint8_t*)c->color.p;
    f.depth = (float*)c->depth.p;
    f.tg = c->tg;
    memcpy(f.m, m, sizeof f.m);
    f.flags = flags;
    return f;
}

// read the events of every completed-but-unread frame (caller has synchronised the stream)
void harvest(swr_context* c) {
    for (; c->harvested < c->seq; c->harvested++) {
        hipEvent_t* ev = c->ev[c->harvested % swr_context::RING];
        float ms[5] = {0, 0, 0, 0, 0};
        if (c->ev_level[c->harvested % swr_context::RING] >= 2) {
            hipEventElapsedTime(&ms[0], ev[0], ev[1]);
            hipEventElapsedTime(&ms[1], ev[1], ev[2]);
            hipEventElapsedTime(&ms[2], ev[2], ev[3]);
            hipEventElapsedTime(&ms[4], ev[0], ev[4]);
        }
        hipEventElapsedTime(&ms[3], ev[3], ev[4]);
        c->last.setup_bin_ms = ms[0]; c->last.scan_ms = ms[1]; c->last.scatter_ms = ms[2];
        c->last.raster_ms = ms[3]; c->last.total_ms = ms[4];
        for (int i = 0; i < 5; i++) c->sum_ms[i] += ms[i];
        c->sum_frames++;
    }
}

// SWR_HOST_PROFILE=1: wall time of the host side of a frame, split by HIP call group, printed at context destroy.
struct HostProfile {
    bool on = getenv("SWR_HOST_PROFILE") && atoi(getenv("SWR_HOST_PROFILE")) == 1;
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t frames = 0;
    std::chrono::steady_clock::time_point last;
    void begin() { if (on) last = std::chrono::steady_clock::now(); }
    void lap(int k) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        t[k] += std::chrono::duration<double, std::micro>(now - last).count();
        last = now;
    }
    ~HostProfile() {
        if (on && frames)
            fprintf(stderr, "[swr host profile] %llu frames, us/frame: prepare %.2f | wait(slot) %.2f | binning launches %.2f | "
                            "event record+wait %.2f | raster-stream launches %.2f | record(ras_done) %.2f\n",
                    (unsigned long long)frames, t[0] / frames, t[1] / frames, t[2] / frames, t[3] / frames, t[4] / frames,
                    t[5] / frames);
    }
};
static HostProfile g_host_profile;

int enqueue_frame(swr_context* c) {
    g_host_profile.begin();
    {
        const BinPlan plan = plan_binning(c->ni / 3, tiles_of(c->tg));
        if (plan.use_lds) {
            const size_t need = (size_t)plan.G * (size_t)tiles_of(c->tg) * 4;
            bool grow = false;
            for (auto& sl : c->slot) grow = grow || sl.bin_matrix.bytes < need;
            if (grow) {
                int rc = sync_streams(c);
                if (rc) return rc;
                for (auto& sl : c->slot)
                    if ((rc = ensure(c, sl.bin_matrix, need))) return rc;
            }
        }
    }
    if (c->last_prim != SWR_PRIMITIVE_TRIANGLE) {
        // .vertices / .line: three small kernels, no binning; the pair counter reads 0
        int rc = sync_streams(c);
        if (rc) return rc;
        DeviceFrame f = make_frame(c, 0, c->last_m, c->last_flags);
        c->h_counters[CNT_PAIRS] = 0;
        launch_points_or_lines(f, c->last_prim, c->stream);
        HIP_TRY(c, hipGetLastError());
        c->draw_pending = true;
        return SWR_OK;
    }
    const int si = (int)(c->frame_no++ % swr_context::NSLOT);
    c->last_slot = si;
    swr_context::Slot& sl = c->slot[si];
    DeviceFrame f = make_frame(c, si, c->last_m, c->last_flags);
    hipEvent_t* ev = nullptr;
    if (c->timing >= 2 || (c->timing == 1 && (c->frame_no % (uint64_t)c->timing_every) == 0)) {
        if (c->seq - c->harvested >= (uint64_t)swr_context::RING) {   // ring full: drain it
            int rc = sync_streams(c);
            if (rc) return rc;
            harvest(c);
        }
        ev = c->ev[c->seq % swr_context::RING];
        c->ev_level[c->seq % swr_context::RING] = c->timing;
        c->seq++;
    }
    hipStream_t sb = c->bin_stream, sr = c->stream;
    g_host_profile.lap(0);
    // this slot's buffers are free again once the raster of NSLOT frames ago has read them
    if (sl.ras_recorded && sb != sr) HIP_TRY(c, hipStreamWaitEvent(sb, sl.ras_done, 0));
    g_host_profile.lap(1);
    if (!f.plan.use_lds || f.ntri <= 0) {
        // global-atomic fallback: counters must start at zero.  Empty scene: no binning kernel
        // runs at all, so the tile table (counts, starts, counters) is simply zeroed.
        if (f.ntri <= 0) { int rc = sync_streams(c); if (rc) return rc; c->h_counters[CNT_PAIRS] = 0; }
        const size_t zero_bytes = (size_t)(CNT_WORDS + 3 * tiles_of(c->tg) + 1) * 4;
        HIP_TRY(c, hipMemsetAsync(sl.tilebuf.p, 0, zero_bytes, sb));
    }
    const bool all = c->timing >= 2;
    if (ev && all) HIP_TRY(c, hipEventRecord(ev[0], sb));
    launch_setup_bin(f, sb);
    if (ev && all) HIP_TRY(c, hipEventRecord(ev[1], sb));
    launch_scan(f, sb);
    if (ev && all) HIP_TRY(c, hipEventRecord(ev[2], sb));
    launch_fill(f, sb);
    // Where k_sort_bins runs.  On the binning stream it is part of the chain that runs ahead of the raster; on
    // the raster stream (right before k_raster) the binning stream is free one kernel earlier.  Alternating A/B on
    // one box (tools/ab_sort_stream.py, tools/bt_bands.sh; untimed frames of cfg4, us per frame):
    //   whole 4K frame (4 080 tiles): 112 vs 122 -> binning stream;  half / quarter / eighth bands: 83 / 59 / 41
    //   vs 73 / 49 / 40 -> raster stream;  light frames (cfg2, 6 k triangles): 37 vs 43 -> binning stream.
    // SWR_SORT_STREAM=0/1 forces either.
    static const int sort_stream_mode = getenv("SWR_SORT_STREAM") ? atoi(getenv("SWR_SORT_STREAM")) : -1;
    const bool sort_on_raster_stream = sort_stream_mode >= 0 ? sort_stream_mode == 1
                                                            : (sb != sr && f.ntri >= 200000 && tiles_of(c->tg) < 3000);
    if (!sort_on_raster_stream) launch_sort_bins(f, sb);
    g_host_profile.lap(2);
    if (sb != sr) {
        HIP_TRY(c, hipEventRecord(sl.bin_done, sb));
        HIP_TRY(c, hipStreamWaitEvent(sr, sl.bin_done, 0));
    }
    g_host_profile.lap(3);
    if (sort_on_raster_stream) launch_sort_bins(f, sr);
    if (ev) HIP_TRY(c, hipEventRecord(ev[3], sr));
    launch_raster(f, sr);
    if (ev) HIP_TRY(c, hipEventRecord(ev[4], sr));
    g_host_profile.lap(4);
    if (sb != sr) { HIP_TRY(c, hipEventRecord(sl.ras_done, sr)); sl.ras_recorded = true; }
    g_host_profile.lap(5);
    g_host_profile.frames++;
    HIP_TRY(c, hipGetLastError());
    c->draw_pending = true;   // the pair total lands in h_counters[CNT_PAIRS] (written by the scan)
    return SWR_OK;
}

}  // namespace

extern "C" {

int swr_draw_primitives(swr_context* c, const float transform[16], uint32_t flags, int32_t primitive_type);

int swr_abi_version(void) { return SWR_ABI_VERSION; }
const char* swr_version(void) { return "swr-hip gfx950 0.1 (tile 64x32, wave64 LDS visibility keys)"; }
int swr_tile_rows(void) { return TILE_H; }
int swr_tile_cols(void) { return TILE_W; }

const char* swr_last_error(const swr_context* ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int swr_band_rows(int64_t height, int32_t parts, int32_t part, int64_t* row_begin, int64_t* row_end) {
    if (height <= 0 || parts <= 0 || part < 0 || part >= parts || !row_begin || !row_end)
        return SWR_ERR_BAD_ARG;
    const int64_t trows = (height + TILE_H - 1) / TILE_H;
    const int64_t t0 = trows * part / parts, t1 = trows * (part + 1) / parts;
    *row_begin = std::min<int64_t>(t0 * TILE_H, height);
    *row_end = std::min<int64_t>(t1 * TILE_H, height);
    return SWR_OK;
}

int swr_context_create(const swr_config* cfg, swr_context** out) {
    if (!out) return fail(nullptr, SWR_ERR_BAD_ARG, "swr_context_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SWR_ERR_HIP, "no HIP device (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    int dev = cfg ? cfg->device : -1;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) return fail(nullptr, SWR_ERR_BAD_ARG, "device %d out of range (%d devices)", dev, ndev);
    swr_context* c = new swr_context();
    c->device = dev;
    if ((e = hipSetDevice(dev)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipHostMalloc((void**)&c->h_counters, CNT_WORDS * 4, hipHostMallocMapped)) != hipSuccess ||
        (e = hipHostGetDevicePointer((void**)&c->h_counters_dev, c->h_counters, 0)) != hipSuccess) {
        int rc = fail(nullptr, SWR_ERR_HIP, "context init failed: %s", hipGetErrorString(e));
        delete c;
        return rc;
    }
    memset(c->h_counters, 0, CNT_WORDS * 4);
    {
        // SWR_PIPELINE=0: binning and raster share one stream (no overlap of consecutive frames)
        const char* pl = getenv("SWR_PIPELINE");
        if (pl && pl[0] == '0') c->bin_stream = c->stream;
        else {
            // a plain second stream: stream priorities (binning highest or lowest) were measured to make
            // no difference to how the two queues share the CUs on this platform
            if (hipStreamCreateWithFlags(&c->bin_stream, hipStreamNonBlocking) != hipSuccess) c->bin_stream = c->stream;
            else c->bin_stream_own = c->bin_stream;
        }
        for (auto& sl : c->slot) {
            hipEventCreateWithFlags(&sl.bin_done, hipEventDisableTiming);
            hipEventCreateWithFlags(&sl.ras_done, hipEventDisableTiming);
        }
    }
    for (int r = 0; r < swr_context::RING; r++)
        for (int i = 0; i < 5; i++) hipEventCreate(&c->ev[r][i]);
    c->ev_ok = true;
    *out = c;
    return SWR_OK;
}

void swr_context_destroy(swr_context* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->bin_stream) hipStreamSynchronize(c->bin_stream);
    if (c->stream) hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->vertices, &c->indices, &c->xyz, &c->rgb, &c->idx32, &c->tri_rgb, &c->tri_xyz, &c->inv, &c->box64, &c->stream_scratch, &c->sort_temp, &c->attrs, &c->tri_nrm, &c->texture, &c->texture_bytes, &c->color, &c->depth};
    for (DevBuf* b : bufs) if (b->p) hipFree(b->p);
    for (auto& sl : c->slot) {
        DevBuf* sb[] = {&sl.geo, &sl.geo_full, &sl.ranges, &sl.bins, &sl.bin_matrix, &sl.live, &sl.tilebuf};
        for (DevBuf* b : sb) if (b->p) hipFree(b->p);
        if (sl.bin_done) hipEventDestroy(sl.bin_done);
        if (sl.ras_done) hipEventDestroy(sl.ras_done);
    }
    if (c->bin_stream_own) hipStreamDestroy(c->bin_stream_own);
    if (c->h_counters) hipHostFree(c->h_counters);
    if (c->ev_ok)
        for (int r = 0; r < swr_context::RING; r++)
            for (int i = 0; i < 5; i++) hipEventDestroy(c->ev[r][i]);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int swr_scene_upload(swr_context* c, const swr_vertex* vertices, int64_t vertex_count,
                     const int64_t* indices, int64_t index_count) {
    if (!c) return SWR_ERR_BAD_ARG;
    if (vertex_count < 0 || index_count < 0 || (index_count > 0 && (!indices || !vertices)))
        return fail(c, SWR_ERR_BAD_ARG, "swr_scene_upload: bad vertex/index arguments");
    if (index_count / 3 >= 0xFFFFFFFFll || vertex_count > 0xFFFFFFFFll)
        return fail(c, SWR_ERR_UNSUPPORTED, "more than 2^32-2 primitives or 2^32 vertices");
    HIP_TRY(c, hipSetDevice(c->device));
    { int rcs = sync_streams(c); if (rcs) return rcs; }
    c->has_scene = false;
    c->has_attrs = false;
    c->draw_pending = false;
    int rc;
    if ((rc = ensure(c, c->vertices, (size_t)vertex_count * sizeof(swr_vertex)))) return rc;
    if ((rc = ensure(c, c->indices, (size_t)index_count * 8))) return rc;
    if ((rc = ensure(c, c->xyz, (size_t)vertex_count * 16))) return rc;
    if ((rc = ensure(c, c->rgb, (size_t)vertex_count * 16))) return rc;
    if ((rc = ensure(c, c->idx32, (size_t)index_count * 4))) return rc;
    if ((rc = ensure(c, c->tri_rgb, (size_t)index_count * 16))) return rc;
    const int64_t ntri = index_count / 3;
    // SWR_SORT=0: keep index order (the original index still travels in GeomRec.flags);
    // SWR_SORT=-1: behave as for a scene of 2^24 primitives or more (no reordering, slot == index) — test hook
    static const int sort_mode = getenv("SWR_SORT") ? atoi(getenv("SWR_SORT")) : 1;
    const bool reorder = sort_mode == 1 && ntri > 1 && ntri < SORT_MAX_TRIS;
    const size_t sort_bytes = reorder ? stream_sort_temp_bytes(ntri) : 0;
    if ((rc = ensure(c, c->tri_xyz, (size_t)index_count * 16))) return rc;
    if ((rc = ensure(c, c->inv, (size_t)ntri * 4))) return rc;
    if ((rc = ensure(c, c->box64, (size_t)((ntri + 63) / 64) * 32))) return rc;
    if ((rc = ensure(c, c->stream_scratch, ((size_t)ntri * 4 + 8) * 4))) return rc;
    if ((rc = ensure(c, c->sort_temp, sort_bytes))) return rc;
    for (auto& sl : c->slot) {
        if ((rc = ensure(c, sl.geo, (size_t)(index_count / 3) * sizeof(GeomRec)))) return rc;
        if ((rc = ensure(c, sl.geo_full, (size_t)(index_count / 3) * sizeof(GeomFull)))) return rc;
        if ((rc = ensure(c, sl.ranges, (size_t)(index_count / 3) * sizeof(uint2)))) return rc;
        if ((rc = ensure(c, sl.live, (size_t)((index_count / 3 + 63) / 64 + 2 * 1024 + 2) * 4))) return rc;   // [G <= 1024][1 + per]
        if ((rc = ensure(c, sl.tilebuf, (size_t)(CNT_WORDS + 3 * std::max(1, tiles_of(c->tg)) + 1) * 4))) return rc;
    }
    if (vertex_count)
        HIP_TRY(c, hipMemcpyAsync(c->vertices.p, vertices, (size_t)vertex_count * sizeof(swr_vertex),
                                  hipMemcpyHostToDevice, c->stream));
    if (index_count)
        HIP_TRY(c, hipMemcpyAsync(c->indices.p, indices, (size_t)index_count * 8, hipMemcpyHostToDevice, c->stream));
    // index range check (Swift array subscript would trap, Renderer.swift:226)
    HIP_TRY(c, hipMemsetAsync(c->slot[0].tilebuf.p, 0, CNT_WORDS * 4, c->stream));
    launch_validate_indices((const int64_t*)c->indices.p, index_count, vertex_count, (uint32_t*)c->slot[0].tilebuf.p, c->stream);
    launch_split_scene((const swr_vertex*)c->vertices.p, vertex_count, (const int64_t*)c->indices.p, index_count,
                       (float4*)c->xyz.p, (float4*)c->rgb.p, (uint32_t*)c->idx32.p, c->stream);
    HIP_TRY(c, hipGetLastError());
    {
        StreamBuild b{};
        b.vertices = (const swr_vertex*)c->vertices.p; b.nv = vertex_count;
        b.indices = (const int64_t*)c->indices.p; b.ntri = ntri;
        b.xyz = (const float4*)c->xyz.p;
        b.sort = reorder;
        b.scratch = (uint32_t*)c->stream_scratch.p;
        b.sort_temp = c->sort_temp.p; b.sort_temp_bytes = sort_bytes;
        b.tri_xyz = (float4*)c->tri_xyz.p; b.tri_rgb = (float4*)c->tri_rgb.p;
        b.inv = (uint32_t*)c->inv.p; b.box64 = (float4*)c->box64.p;
        HIP_TRY(c, launch_build_stream(b, c->stream));
        c->reordered = ntri > 0 && ntri < SORT_MAX_TRIS && sort_mode != -1;   // original index travels in GeomRec.flags
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_counters, c->slot[0].tilebuf.p, CNT_WORDS * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->h_counters[CNT_BAD_INDEX])
        return fail(c, SWR_ERR_INDEX_RANGE, "an index is outside [0, %lld)", (long long)vertex_count);
    c->nv = vertex_count;
    c->ni = index_count;
    c->has_scene = true;
    const uint64_t want = (uint64_t)(index_count / 3) * 2 + 65536;
    return ensure_capacity(c, (uint32_t)std::min<uint64_t>(want, 0xFFFFFFF0ull));
}

int swr_scene_attributes(swr_context* c, const swr_vertex_attr* attributes, int64_t vertex_count) {
    if (!c) return SWR_ERR_BAD_ARG;
    if (!c->has_scene) return fail(c, SWR_ERR_NO_SCENE, "swr_scene_attributes needs swr_scene_upload first");
    if (vertex_count != c->nv || (vertex_count > 0 && !attributes))
        return fail(c, SWR_ERR_BAD_ARG, "swr_scene_attributes: %lld attributes for %lld vertices",
                    (long long)vertex_count, (long long)c->nv);
    HIP_TRY(c, hipSetDevice(c->device));
    { int rcs = sync_streams(c); if (rcs) return rcs; }
    int rc;
    if ((rc = ensure(c, c->attrs, (size_t)vertex_count * sizeof(swr_vertex_attr)))) return rc;
    if ((rc = ensure(c, c->tri_nrm, (size_t)c->ni * 16))) return rc;
    if (vertex_count)
        HIP_TRY(c, hipMemcpyAsync(c->attrs.p, attributes, (size_t)vertex_count * sizeof(swr_vertex_attr),
                                  hipMemcpyHostToDevice, c->stream));
    launch_gather_attrs((const swr_vertex_attr*)c->attrs.p, vertex_count, (const int64_t*)c->indices.p, c->ni / 3,
                        (const float4*)c->tri_xyz.p, (float4*)c->tri_nrm.p, (float4*)c->tri_rgb.p, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->has_attrs = true;
    return SWR_OK;
}

int swr_material_set(swr_context* c, const swr_material* m) {
    if (!c) return SWR_ERR_BAD_ARG;
    if (!m) { c->material = swr_material{}; return SWR_OK; }
    if (m->shader != SWR_SHADER_PASSTHROUGH && m->shader != SWR_SHADER_PHONG && m->shader != SWR_SHADER_TEXTURED_PHONG)
        return fail(c, SWR_ERR_UNSUPPORTED, "unknown shader %d", m->shader);
    if (m->shininess_log2 < 0 || m->shininess_log2 > 16)
        return fail(c, SWR_ERR_BAD_ARG, "shininess_log2 %d outside [0,16]", m->shininess_log2);
    c->material = *m;      // read at the next swr_draw (by value into the kernel arguments)
    return SWR_OK;
}

int swr_texture_upload(swr_context* c, const void* bgra8, int32_t width, int32_t height) {
    if (!c) return SWR_ERR_BAD_ARG;
    if (!bgra8 || width <= 0 || height <= 0 || width > 16384 || height > 16384)
        return fail(c, SWR_ERR_BAD_ARG, "swr_texture_upload: bad texture %dx%d", width, height);
    HIP_TRY(c, hipSetDevice(c->device));
    { int rcs = sync_streams(c); if (rcs) return rcs; }
    int rc;
    const size_t n = (size_t)width * (size_t)height;
    if ((rc = ensure(c, c->texture_bytes, n * 4))) return rc;
    if ((rc = ensure(c, c->texture, n * 16))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->texture_bytes.p, bgra8, n * 4, hipMemcpyHostToDevice, c->stream));
    launch_texture_to_float((const uint32_t*)c->texture_bytes.p, (int64_t)n, (float4*)c->texture.p, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->tex_w = width; c->tex_h = height;
    return SWR_OK;
}

int swr_target_set(swr_context* c, int64_t width, int64_t height, int64_t row_begin, int64_t row_end) {
    if (!c) return SWR_ERR_BAD_ARG;
    if (width <= 0 || height <= 0 || width > 65535 || height > 65535)
        return fail(c, SWR_ERR_BAD_ARG, "bad framebuffer size %lldx%lld", (long long)width, (long long)height);
    if (row_begin < 0 || row_end > height || row_begin > row_end || (row_begin % TILE_H) != 0)
        return fail(c, SWR_ERR_BAD_ARG, "bad band [%lld,%lld): row_begin must be a multiple of %d",
                    (long long)row_begin, (long long)row_end, TILE_H);
    HIP_TRY(c, hipSetDevice(c->device));
    { int rcs = sync_streams(c); if (rcs) return rcs; }
    c->draw_pending = false;
    Target t;
    t.width = (int32_t)width; t.height = (int32_t)height;
    t.row_begin = (int32_t)row_begin; t.row_end = (int32_t)row_end;
    t.tiles_x = (int32_t)((width + TILE_W - 1) / TILE_W);
    t.tiles_y = (int32_t)((row_end - row_begin + TILE_H - 1) / TILE_H);
    const size_t px = (size_t)width * (size_t)(row_end - row_begin);
    int rc;
    if ((rc = ensure(c, c->color, px * 4))) return rc;
    if ((rc = ensure(c, c->depth, px * 4))) return rc;
    for (auto& sl : c->slot)
        if ((rc = ensure(c, sl.tilebuf, (size_t)(CNT_WORDS + 3 * std::max(1, tiles_of(t)) + 1) * 4))) return rc;
    c->tg = t;
    c->has_target = true;
    return SWR_OK;
}

int swr_draw(swr_context* c, const float transform[16], uint32_t flags) {
    return swr_draw_primitives(c, transform, flags, SWR_PRIMITIVE_TRIANGLE);
}

int swr_draw_primitives(swr_context* c, const float transform[16], uint32_t flags, int32_t primitive_type) {
    if (!c || !transform) return SWR_ERR_BAD_ARG;
    if (primitive_type != SWR_PRIMITIVE_TRIANGLE && primitive_type != SWR_PRIMITIVE_LINE &&
        primitive_type != SWR_PRIMITIVE_VERTICES)
        return fail(c, SWR_ERR_UNSUPPORTED, "unknown primitive type %d", primitive_type);
    {
        const int per = primitive_type == SWR_PRIMITIVE_LINE ? 2 : 3;          // verticesCount, Renderer.swift:179-188
        if (c->has_scene && c->ni % per != 0)                                  // assert, Renderer.swift:209
            return fail(c, SWR_ERR_INDEX_COUNT, "index_count %lld is not a multiple of %d", (long long)c->ni, per);
    }
    if (!c->has_scene || !c->has_target)
        return fail(c, SWR_ERR_NO_SCENE, "swr_draw needs swr_scene_upload and swr_target_set first");
    if (flags & ~(uint32_t)(SWR_FLAG_DEPTH_TEST | SWR_FLAG_NO_COLOR | SWR_FLAG_METAL_RULES))
        return fail(c, SWR_ERR_BAD_ARG, "unknown flag bits 0x%x", flags);
    if (primitive_type == SWR_PRIMITIVE_TRIANGLE && !(flags & SWR_FLAG_NO_COLOR) &&
        c->material.shader != SWR_SHADER_PASSTHROUGH) {
        if (!c->has_attrs)
            return fail(c, SWR_ERR_BAD_ARG, "the material needs vertex attributes (swr_scene_attributes)");
        if (c->material.shader == SWR_SHADER_TEXTURED_PHONG && c->tex_w <= 0)
            return fail(c, SWR_ERR_BAD_ARG, "the material needs a texture (swr_texture_upload)");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    memcpy(c->last_m, transform, sizeof c->last_m);
    c->last_flags = flags;
    c->last_prim = primitive_type;
    return enqueue_frame(c);
}

int swr_sync(swr_context* c) {
    if (!c) return SWR_ERR_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    for (int attempt = 0; attempt < 8; attempt++) {
        { int rcs = sync_streams(c); if (rcs) return rcs; }
        if (!c->draw_pending) { harvest(c); return SWR_OK; }
        const uint32_t pairs = c->h_counters[CNT_PAIRS];
        if (pairs <= c->capacity) {
            c->draw_pending = false;
            c->last.tile_pairs = pairs;
            c->last.tiles = tiles_of(c->tg);
            c->last.triangles = c->ni / 3;
            harvest(c);
            return SWR_OK;
        }
        // the (triangle,tile) pair list overflowed: grow and redraw the same frame
        const uint64_t want = (uint64_t)pairs + pairs / 4 + 1024;
        if (want > 0xFFFFFFF0ull) return fail(c, SWR_ERR_UNSUPPORTED, "too many (triangle,tile) pairs: %u", pairs);
        int rc = ensure_capacity(c, (uint32_t)want);
        if (rc) return rc;
        if ((rc = enqueue_frame(c))) return rc;
    }
    return fail(c, SWR_ERR_HIP, "pair list kept overflowing");
}

int swr_read_color(swr_context* c, void* dst) {
    if (!c || !dst) return SWR_ERR_BAD_ARG;
    int rc = swr_sync(c);
    if (rc) return rc;
    const size_t row = (size_t)c->tg.width * 4;
    const size_t rows = (size_t)(c->tg.row_end - c->tg.row_begin);
    if (rows)
        HIP_TRY(c, hipMemcpy((uint8_t*)dst + (size_t)c->tg.row_begin * row, c->color.p, rows * row, hipMemcpyDeviceToHost));
    return SWR_OK;
}

int swr_read_depth(swr_context* c, float* dst) {
    if (!c || !dst) return SWR_ERR_BAD_ARG;
    int rc = swr_sync(c);
    if (rc) return rc;
    const size_t row = (size_t)c->tg.width * 4;
    const size_t rows = (size_t)(c->tg.row_end - c->tg.row_begin);
    if (rows)
        HIP_TRY(c, hipMemcpy((uint8_t*)dst + (size_t)c->tg.row_begin * row, c->depth.p, rows * row, hipMemcpyDeviceToHost));
    return SWR_OK;
}

int swr_timing_enable(swr_context* c, int enable) {
    if (!c) return SWR_ERR_BAD_ARG;
    int rc = swr_sync(c);
    if (rc) return rc;
    c->timing = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    return SWR_OK;
}

int swr_timing_sample(swr_context* c, int every_nth) {
    if (!c || every_nth < 1) return SWR_ERR_BAD_ARG;
    int rc = swr_sync(c);
    if (rc) return rc;
    c->timing_every = every_nth;
    return SWR_OK;
}

int swr_pipeline_enable(swr_context* c, int enable) {
    if (!c) return SWR_ERR_BAD_ARG;
    int rc = swr_sync(c);
    if (rc) return rc;
    c->bin_stream = (enable && c->bin_stream_own) ? c->bin_stream_own : c->stream;
    return SWR_OK;
}

int swr_timing_totals(swr_context* c, swr_timings* sum, int64_t* frames) {
    if (!c || !sum || !frames) return SWR_ERR_BAD_ARG;
    int rc = swr_sync(c);
    if (rc) return rc;
    *sum = c->last;
    sum->setup_bin_ms = (float)c->sum_ms[0]; sum->scan_ms = (float)c->sum_ms[1];
    sum->scatter_ms = (float)c->sum_ms[2]; sum->raster_ms = (float)c->sum_ms[3];
    sum->total_ms = (float)c->sum_ms[4];
    *frames = c->sum_frames;
    return SWR_OK;
}

int swr_timing_reset(swr_context* c) {
    if (!c) return SWR_ERR_BAD_ARG;
    int rc = swr_sync(c);
    if (rc) return rc;
    for (double& v : c->sum_ms) v = 0.0;
    c->sum_frames = 0;
    return SWR_OK;
}

int swr_get_timings(swr_context* c, swr_timings* out) {
    if (!c || !out) return SWR_ERR_BAD_ARG;
    int rc = swr_sync(c);
    if (rc) return rc;
    *out = c->last;
    return SWR_OK;
}

int swr_render(swr_context* c, const swr_render_pass* p) {
    if (!c || !p) return SWR_ERR_BAD_ARG;
    if (p->primitive_type != SWR_PRIMITIVE_TRIANGLE && p->primitive_type != SWR_PRIMITIVE_LINE &&
        p->primitive_type != SWR_PRIMITIVE_VERTICES)
        return fail(c, SWR_ERR_UNSUPPORTED, "unknown primitive type %d", p->primitive_type);
    if (p->index_count >= 0 && p->index_count % (p->primitive_type == SWR_PRIMITIVE_LINE ? 2 : 3) != 0)   // Renderer.swift:209
        return fail(c, SWR_ERR_INDEX_COUNT, "index_count %lld is not a multiple of %d", (long long)p->index_count,
                    p->primitive_type == SWR_PRIMITIVE_LINE ? 2 : 3);
    if (!p->depth || (!(p->flags & SWR_FLAG_NO_COLOR) && !p->color))
        return fail(c, SWR_ERR_BAD_ARG, "swr_render: colour/depth image pointer is NULL");
    int rc;
    if ((rc = swr_scene_upload(c, p->vertices, p->vertex_count, p->indices, p->index_count))) return rc;
    // the pass carries its own fragment stage: NULL material = the reference's passthrough
    if ((rc = swr_material_set(c, p->material))) return rc;
    if (p->attributes && (rc = swr_scene_attributes(c, p->attributes, p->vertex_count))) return rc;
    if (p->texture && (rc = swr_texture_upload(c, p->texture, p->tex_width, p->tex_height))) return rc;
    if ((rc = swr_target_set(c, p->width, p->height, 0, p->height))) return rc;
    if ((rc = swr_draw_primitives(c, p->transform, p->flags, p->primitive_type))) return rc;
    if (!(p->flags & SWR_FLAG_NO_COLOR) && (rc = swr_read_color(c, p->color))) return rc;
    return swr_read_depth(c, p->depth);
}

}  // extern "C"
