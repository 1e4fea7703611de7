// swr_api.hip — the C-ABI of include/swr.h over the gfx950 kernels (swr_kernels.hip).
//
// Stands in for GpuRenderer.render(renderPass:) (renderer/GpuRenderer.swift:35-141): where the
// reference allocates shared MTLBuffers, encodes one compute dispatch per triangle and blocks
// twice per frame in scheduleAndWait, this context keeps scene, records, bins and the
// framebuffer band resident in HBM and enqueues five kernels on two HIP streams per frame.
//
// One swr_context is either
//   * a single-device context: one GPU, one tile-row band of the framebuffer, or
//   * a group (swr_config.device_count = N > 1): N single-device sub-contexts, sub-context k owning band k of the
//     group's target on device (first + k) % devices, each driven by its own host thread.  Every entry point fans
//     out; the scene is replicated; there is no collective — each device copies its band straight into its rows
//     of the caller's one host image (swr_present / swr_read_*: hipMemcpyAsync on the device's copy streams).
//
// There is no CPU fallback: without a HIP device every computing entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "swr_internal.h"

using namespace swr;

namespace {
thread_local std::string g_create_error;
// true on a context's own helper threads (bin_worker / ras_worker): an error there cannot be written into the context's
// error string — the caller may be reading it (swr_last_error) — and is fatal for the context anyway
thread_local bool tl_helper_thread = false;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// One host thread per sub-context of a group: HIP calls of different devices are enqueued in parallel and the
// caller's thread pays a few microseconds per fan-out instead of N x (nine HIP calls).  Jobs run in FIFO order;
// asynchronous posts (swr_draw, swr_present) return at once and a failure is kept until the next blocking call.
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv_job, cv_idle;
    std::deque<std::function<int()>> q;
    bool busy = false, quit = false;
    int sticky_rc = 0;

    void start(int device, bool helper = false) {
        th = std::thread([this, device, helper] {
            tl_helper_thread = helper;
            (void)hipSetDevice(device);
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                if (q.empty()) {
                    // spin briefly before sleeping: frames arrive every few tens of microseconds
                    lk.unlock();
#ifndef SWR_TUNE_HELPER_SPIN
#define SWR_TUNE_HELPER_SPIN 4000
#endif
                    for (int spin = 0; spin < SWR_TUNE_HELPER_SPIN; spin++) {
                        __builtin_ia32_pause();
                        if (pending.load(std::memory_order_acquire)) break;
                    }
                    lk.lock();
                    cv_job.wait(lk, [this] { return quit || !q.empty(); });
                }
                if (q.empty() && quit) return;
                std::function<int()> job = std::move(q.front());
                q.pop_front();
                pending.store(!q.empty(), std::memory_order_release);
                busy = true;
                lk.unlock();
                const int rc = job();
                lk.lock();
                busy = false;
                if (rc && !sticky_rc) sticky_rc = rc;
                if (q.empty()) cv_idle.notify_all();
            }
        });
    }
    void post(std::function<int()> job) {
        {
            std::lock_guard<std::mutex> lk(m);
            q.push_back(std::move(job));
            pending.store(true, std::memory_order_release);
        }
        cv_job.notify_one();
    }
    // wait until every posted job has run; returns (and clears) the first failure since the last drain
    int drain() {
        std::unique_lock<std::mutex> lk(m);
        cv_idle.wait(lk, [this] { return q.empty() && !busy; });
        const int rc = sticky_rc;
        sticky_rc = 0;
        return rc;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv_job.notify_one();
        if (th.joinable()) th.join();
    }
    std::atomic<bool> pending{false};
};
}  // namespace

static_assert(sizeof(swr_render_pass) == 192 && sizeof(swr_material) == 56 && sizeof(swr_vertex_attr) == 32 &&
              sizeof(swr_vertex) == 32 && sizeof(swr_config) == 16 && sizeof(swr_render_times) == 28,
              "include/swr.h layouts (mirrored by the ctypes / Swift bindings)");

struct swr_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;                    // written and read on the caller's thread only (swr_last_error)

    // ---- failure state: a HIP error while a helper thread enqueues, or a wait that outlives the budget, fails the
    // context for good.  `failed` is what every spin loop, every later call and the destructor look at; the text is
    // written once under the mutex (by whichever thread failed) and copied into `err` by the caller's thread.
    std::atomic<int> failed{0};
    std::mutex err_m;
    std::string failed_msg;
    uint32_t wait_budget_ms = 20000;
    std::atomic<int> inject{0};         // swr_debug_fault: consumed by the next frame's raster share
    // swr_debug_set (test hooks; the defaults are what a renderer gets)
    int dbg_stream_order = 1;           // 1 Morton-ordered stream, 0 caller's order, -1 as for >= 2^24 primitives
    int dbg_cull = 1;                   // per-band culling of 64-primitive groups
    int dbg_bin_mode = 0;               // 0 auto, 1 exact-size bins, 2 fixed-stride bins everywhere, 3 global-atomic fallback
    int64_t dbg_oneshot_min_tris = (int64_t)1 << 18;
    bool dbg_k32 = true;                // depth-only z-tested frames on 32-bit depth keys
    int dbg_insort = 1;                 // ... which sort their bins inside the raster workgroups: 0 never, 1 on small grids, 2 always
    // scene identity of swr_render (swr_render_pass.scene_id): what is resident
    uint64_t scene_id = 0;
    int64_t scene_nv = -1, scene_ni = -1;
    bool scene_attrs = false;
    const void* scene_tex = nullptr; int32_t scene_tw = 0, scene_th = 0;
    swr_render_times rt{};              // phases of the last swr_render
    float up_h2d_ms = 0.0f, up_build_ms = 0.0f;   // of the last swr_scene_upload (HIP events)
    hipEvent_t up_ev[3] = {nullptr, nullptr, nullptr};

    // ---- group (device_count > 1): the sub-contexts and their host threads; nothing below is used by a group ----
    std::vector<swr_context*> kids;
    std::vector<Worker*> workers;
    Target group_tg{};
    bool group_has_target = false;

    // scene (RenderPass.vertices / .indices)
    DevBuf vertices, indices, tri_rgb;
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

    // target band (RenderPass.colorBuffer / .depthBuffer), double-buffered in HBM: swr_present copies the frame just
    // drawn to the host while the next swr_draw renders into the other buffer
    Target tg{};
    bool has_target = false;
#ifndef SWR_NSLOT
#define SWR_NSLOT 4   // working sets = frame lanes = frames in flight: 3 -> 4 lanes: worst band of 8 16.2 -> 13.5 us, of 2 35.8 -> 33.7, the whole frame the same; 5: slower everywhere (more streams than hardware queues), profiles/r04/lanes_probe.txt
#endif
    static constexpr int NSLOT = SWR_NSLOT;
    // (one per lane: with frame lanes — below — every frame in flight renders into its own buffer; the two-stream pipeline uses two)
    static constexpr int NFB = NSLOT;
    static_assert(NFB >= 2, "double-buffered at least");
    DevBuf color[NFB], depth[NFB];
    int fb_cur = 0;                     // the next swr_draw renders into this buffer
    int fb_last = 0;                    // the buffer of the last swr_draw (what swr_present / swr_read_* copy)
    hipStream_t last_stream = nullptr;  // the stream that carries the last frame's raster (swr_present records frame_done behind it)
    hipStream_t copy_stream[2] = {nullptr, nullptr};   // colour, depth: both images in flight together
    hipEvent_t frame_done[NFB] = {};     // raster stream -> copy streams, per framebuffer
    hipEvent_t copy_done[NFB][2] = {};   // [fb][image]: the raster into fb waits for these
    bool copy_recorded[NFB][2] = {};
    // pinned staging for destinations that are not page-locked (two 8 MiB chunks per image, D2H / memcpy pipelined)
    static constexpr size_t STAGE_BYTES = 8u << 20;
    void* stage[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    hipEvent_t stage_ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    // the last swr_present (for the redo of a frame whose pair list overflowed)
    void* present_color = nullptr;
    float* present_depth = nullptr;
    bool present_pending = false;

#ifndef SWR_RAS_EVERY
#define SWR_RAS_EVERY 1   // every n-th k_raster carries a completion event (see RAS_EVERY)
#endif
    // A kernel that carries a completion signal (ras_done) costs the NEXT kernel of its queue ~5 us
    // (profiles/r02/c_kernel_trace_pipelined.txt).  The binning of frame N needs "k_raster(N - NSLOT) has finished"; it
    // waits for the first event-carrying raster at or after that frame instead (same queue, in order), so only every
    // RAS_EVERY-th raster needs one, at the price of RAS_EVERY - 1 extra working sets.  Measured (NSLOT / RAS_EVERY = 3 / 1,
    // 4 / 2, 6 / 4; profiles/r02/ras_every_ab.txt): cfg4 0.0977 / 0.0993 / 0.0981 ms, 1/8 band 29.2 / 27.7 / 27.0 us — the
    // gap on the raster queue is filled by the binning queue's kernels (the frame is bound by the sum of the work), so
    // the default stays 3 / 1.
    static constexpr int RAS_EVERY = SWR_RAS_EVERY;
    static_assert(RAS_EVERY >= 1 && NSLOT > RAS_EVERY, "the event-carrying raster must be older than the frame being binned");
    // Per-frame working set, multi-buffered: the binning kernels of frame N+1 run on `bin_stream`
    // while k_raster of frame N runs on `stream` (HBM-bound vs LDS/VALU-bound: they overlap well).
    struct Slot {
        DevBuf geo, geo_full, ranges, bins, bin_matrix;
        DevBuf biglist;        // fixed-stride bins: the frame's deferred large triangles (k_bin -> k_sort_bins)
        DevBuf live;           // per binning workgroup: count + surviving stream-group ids (k_setup_hist -> k_fill_lds)
        DevBuf tilebuf;        // [CNT_WORDS counters][tiles tile_count][tiles+1 tile_start][tiles cursor]
        hipEvent_t bin_done = nullptr, ras_done = nullptr;
        uint64_t ras_event_frame = ~0ull;   // the frame whose k_raster this slot's ras_done tracks (none yet)
    } slot[NSLOT];
    hipStream_t bin_stream = nullptr;   // the stream binning is enqueued on (== stream when pipelining is off)
    hipStream_t bin_stream_own = nullptr;
    hipEvent_t up_chunk_ev = nullptr;   // one-shot upload: index chunk k resident / its stream built
    uint64_t frame_no = 0;
    uint64_t synced_upto = 0;           // every frame below this has completed (full stream sync seen by the caller)
    int last_slot = 0;
    uint32_t capacity = 0;              // exact bins: (triangle,tile) pairs every slot's `bins` holds
    // Fixed-stride bins (k_bin: ONE binning launch per frame): tile t owns bins[t * cap_tile, (t + 1) * cap_tile).  The
    // per-tile fill counters live in four rotating blocks (frame % 4) while the working sets rotate by three: k_bin of
    // frame N zeroes the block of frame N + 1, which was last read by the raster of frame N - 3 — the raster whose
    // completion lets the binning of frame N start.
    static constexpr int NFILL = NSLOT + 1;      // k_bin(N) zeroes the fill block of frame N + 1: its last reader must be the raster of frame N - NSLOT
    DevBuf fillbuf[NFILL];              // [CNT_WORDS counters][tiles fills]
    bool fill_dirty[NFILL] = {};        // block was used and nothing has zeroed it since (then the frame memsets it first)
    // ---- FRAME LANES (round 4; the default whenever frames may overlap).  The binning and the raster of ONE frame go back to
    // back onto ONE stream — kernels of a queue follow each other without a gap, and nothing else orders them: no event, no
    // wait packet, no completion signal, no helper thread that polls — and consecutive frames onto NSLOT different streams.
    // Everything a frame writes belongs to its lane: the working set (slot = lane), its two fill blocks (k_bin of the lane's
    // n-th frame zeroes the block of its (n + 1)-th, last read by a raster that is earlier on the same stream) and its
    // framebuffer (fb advances with every draw; the buffer a frame writes was last written three frames earlier — same lane).
    // Frames of different lanes overlap freely on the chip.  What the two-stream pipeline of rounds 1-3 paid per frame — a
    // 5 us gap behind every kernel that carries a completion signal, the polls of two helper threads — is gone: cfg4 76 -> 67 us
    // per frame, and a thin band (1/8 of the frame: bound by its kernel chain, not by the chip) 24.5 -> 15.9 us
    // (profiles/r04/lanes_probe.txt).  SWR_LANES=0 keeps the two-stream pipeline.
    hipStream_t lane_stream[NSLOT] = {};
    bool lanes_ok = false;
    DevBuf lanefill[NSLOT][2];
    bool lanefill_dirty[NSLOT][2] = {};
    // pacing of an un-waited burst: every PACE_EVERY-th frame leaves a marker behind its raster; a draw waits for the marker of
    // 3 * PACE_EVERY frames ago (the per-frame pinned words are a ring of PAIR_RING frames)
    static constexpr int PACE_EVERY = 24;
    hipEvent_t pace_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t pace_frame[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    uint32_t cap_tile = 0;              // entries per tile region
    bool fixed_mode = false;            // frames are binned by k_bin (else: exact-size bins, four kernels)
    bool fixed_allowed = true;          // false once a tile of this scene / target needed more than the fixed path can give
    size_t bins_entries = 0;            // entries every slot's `bins` holds

    // Pinned, device-mapped words.  h_pairs[f % PAIR_RING] receives the (triangle,tile) pair total of frame f from
    // the binning kernels (no D2H copy, no host sync per frame); a whole burst of un-waited frames can be checked
    // for bin overflow afterwards.  h_misc is the upload's index-check word.
    // (Fixed-stride bins: words [PAIR_RING + 1, 2 PAIR_RING + 1) receive the LARGEST tile fill of frame f — the overflow
    // test there is per tile region.)
    static constexpr int PAIR_RING = 256;
    uint32_t* h_pairs = nullptr;
    uint32_t* h_pairs_dev = nullptr;
    bool frame_presented[PAIR_RING] = {};   // was frame f copied to the host (swr_present)?  Only then can an overflow be seen
    uint32_t* h_misc = nullptr;
    // Depth-only z-tested frames: 32-bit depth keys (k_raster_depth) until the scene shows that too many of its tiles have
    // to be rastered again with the 64-bit keys (depths that are not > +0: a 2-D scene at z = 0, geometry in front of the
    // near plane); word 2 PAIR_RING + 1 of h_pairs receives the sampled count of such tiles, one launch late.
    bool k32_ok = true;                 // reset by a scene of another size / another tile grid
    int64_t sized_ntri = -1;            // what size_bins last sized the bins for
    int sized_tiles = -1;
    DevBuf redo_cnt;                    // the device counter behind that word
    uint64_t frames_checked = 0;        // frames [frames_checked, frame_no) have not had their pair total looked at
    // Two host threads per context: every frame is ~9 HIP calls (3.5 us each); the binning stream's share (slot wait,
    // three or four launches, event record) is enqueued by `bin_worker` while the caller's thread enqueues the raster
    // stream's share (event wait, launches, event record) of the PREVIOUS frame — a thin band or a small scene is
    // bound by the host's enqueue rate, not by the GPU.
    Worker* bin_worker = nullptr;
    // ... and a third one for the raster stream's share, for a different reason: a cross-stream hipStreamWaitEvent in
    // front of a kernel costs that kernel ~10 us after its predecessor on the queue has ended, even when the event
    // completed long before (profiles/r02/gaps_pipelined_before.txt) — a wait that is never enqueued costs nothing.
    // `ras_worker` polls hipEventQuery(bin_done) of the frame and only then enqueues k_raster, with no wait in front
    // (and `bin_worker` does the same with ras_done before it re-uses a working set).  swr_draw / swr_present only post.
    // SWR_EVENT_WAITS=1: the caller's thread enqueues the raster share behind event waits (round-2 first half).
    Worker* ras_worker = nullptr;
    // bin_done / ras_done are bound to the kernels they follow (hipExtLaunchKernelGGL's stop event) instead of being
    // recorded behind them: a marker packet between two kernels of a queue costs the second one ~6.5 us.  SWR_BIND_EVENTS=0: record.
    bool bind_events = true;
    // One frame at a time (an interactive app: swr_draw, swr_sync / swr_present_wait, idle, ...) does not need the
    // helpers: when every earlier frame is known to be complete, the frame is enqueued by the caller's thread itself,
    // ordered by event waits on the streams — the helpers may be asleep by then, and waking two threads costs more than
    // the two wait packets (draw -> sync of cfg4 after 5 ms of idle: 205 -> 180 us; cfg2 94 -> 70 us,
    // profiles/r02/cold_latency.txt).  Frames drawn while others are in flight go through the helpers.  SWR_INLINE_IDLE=0: never.
    bool inline_idle = true;
    std::atomic<uint64_t> bin_enqueued{0};      // frames whose binning (incl. the bin_done record) is on the binning stream
    std::atomic<uint64_t> ras_enqueued{0};      // frames whose raster (incl. the ras_done record) is on the raster stream
    std::atomic<int> bin_error{0};
    uint64_t posted = 0;                        // frames whose binning share has been handed to the helper (or run inline)
    static constexpr int RAS_RING = 64;          // frames whose raster share may be waiting for ras_worker; swr_draw blocks beyond that
    struct RasJob { DeviceFrame f; hipEvent_t ev3 = nullptr, ev4 = nullptr; int si = 0; bool sort_here = false; int fb = 0; bool paced = false; } ras_job[RAS_RING];
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

    // SWR_HOST_PROFILE=1: wall time of the host side of a frame, split by HIP call group, printed at destroy
    bool hp_on = false;
    double hp_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t hp_frames = 0;
    std::chrono::steady_clock::time_point hp_last;
    std::chrono::steady_clock::time_point hp_last_r;    // the raster share's own clock (it may run on ras_worker)
    void hp_begin_r() { if (hp_on) hp_last_r = std::chrono::steady_clock::now(); }
    void hp_lap_r(int k) {
        if (!hp_on) return;
        const auto now = std::chrono::steady_clock::now();
        hp_t[k] += std::chrono::duration<double, std::micro>(now - hp_last_r).count();
        hp_last_r = now;
    }
    void hp_begin() { if (hp_on) hp_last = std::chrono::steady_clock::now(); }
    void hp_lap(int k) {
        if (!hp_on) return;
        const auto now = std::chrono::steady_clock::now();
        hp_t[k] += std::chrono::duration<double, std::micro>(now - hp_last).count();
        hp_last = now;
    }
};

namespace {

// The context has failed for good (first failure wins).  Callable from any thread.
int fatal_msg(swr_context* c, int code, const char* text) {
    std::lock_guard<std::mutex> lk(c->err_m);
    if (!c->failed.load(std::memory_order_relaxed)) {
        c->failed_msg = text;
        c->failed.store(code, std::memory_order_release);
    }
    return code;
}

int fail(swr_context* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (!c) g_create_error = buf;
    else if (tl_helper_thread) return fatal_msg(c, code, buf);    // never touch c->err from a helper (the caller may be reading it)
    else c->err = buf;
    return code;
}

int fatal(swr_context* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return fatal_msg(c, code, buf);
}

// Caller's thread: has the context failed?  Then its text becomes the last error and every entry point returns the code.
int sticky(swr_context* c) {
    const int f = c->failed.load(std::memory_order_acquire);
    if (f) {
        std::lock_guard<std::mutex> lk(c->err_m);
        c->err = c->failed_msg;
    }
    return f;
}

// Bounded waiting: spin, then yield, never beyond the context's budget; gives up at once when the context has failed.
struct Deadline {
    std::chrono::steady_clock::time_point end;
    unsigned spins = 0;
    explicit Deadline(const swr_context* c) : end(std::chrono::steady_clock::now() + std::chrono::milliseconds(c->wait_budget_ms)) {}
    // false: the budget is used up
    bool pause() {
        if (++spins < 2000) { for (int i = 0; i < 8; i++) __builtin_ia32_pause(); return true; }
        std::this_thread::yield();
        return (spins & 63u) != 0 || std::chrono::steady_clock::now() < end;
    }
};

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

int ensure_bins(swr_context* c, size_t entries) {
    if (entries <= c->bins_entries) return SWR_OK;
    int rc;
    for (auto& sl : c->slot)
        if ((rc = ensure(c, sl.bins, entries * 4))) return rc;
    c->bins_entries = entries;
    return SWR_OK;
}

int ensure_capacity(swr_context* c, uint32_t cap) {
    if (cap <= c->capacity) return SWR_OK;
    const int rc = ensure_bins(c, cap);
    if (rc) return rc;
    c->capacity = cap;
    return SWR_OK;
}

int flush_raster(swr_context* c, uint64_t upto_frame_count);
int enqueue_raster_shares(swr_context* c, uint64_t upto_frame_count);

// hipEventQuery until the event has completed (host-paced ordering: see swr_context::ras_worker) — or the context has
// failed, or the wait budget is used up: then the context fails with what was being waited for.
int poll_event(swr_context* c, hipEvent_t ev, const char* what, uint64_t frame, bool never = false) {
    Deadline d(c);
    for (;;) {
        if (const int f = c->failed.load(std::memory_order_acquire)) return f;
        const hipError_t e = never ? hipErrorNotReady : hipEventQuery(ev);
        if (e == hipSuccess) return SWR_OK;
        if (e != hipErrorNotReady) return fatal(c, SWR_ERR_HIP, "hipEventQuery(%s of frame %llu) failed: %s", what, (unsigned long long)frame, hipGetErrorString(e));
        if (!never) (void)hipGetLastError();                  // hipErrorNotReady is sticky in hipGetLastError
        if (!d.pause())
            return fatal(c, SWR_ERR_HIP, "frame %llu: %s did not complete within %u ms (device %d); the context has failed",
                         (unsigned long long)frame, what, c->wait_budget_ms, c->device);
    }
}

// hipStreamSynchronize with the same bound (hipStreamQuery polls)
int wait_stream(swr_context* c, hipStream_t s, const char* what) {
    Deadline d(c);
    for (;;) {
        if (const int f = c->failed.load(std::memory_order_acquire)) return f;
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return SWR_OK;
        if (e != hipErrorNotReady) return fatal(c, SWR_ERR_HIP, "hipStreamQuery(%s) failed: %s", what, hipGetErrorString(e));
        (void)hipGetLastError();
        if (!d.pause())
            return fatal(c, SWR_ERR_HIP, "the %s did not drain within %u ms (device %d, %llu frames drawn); the context has failed",
                         what, c->wait_budget_ms, c->device, (unsigned long long)c->frame_no);
    }
}

// wait until an atomic frame counter has passed `g` (the other helper's progress)
int wait_counter(swr_context* c, const std::atomic<uint64_t>& ctr, uint64_t g, const char* what) {
    Deadline d(c);
    while (ctr.load(std::memory_order_acquire) <= g) {
        if (const int f = c->failed.load(std::memory_order_acquire)) return f;
        if (!d.pause())
            return fatal(c, SWR_ERR_HIP, "frame %llu: %s was not enqueued within %u ms; the context has failed", (unsigned long long)g, what, c->wait_budget_ms);
    }
    return SWR_OK;
}

// frames go onto the lanes (swr_context::lane_stream) whenever frames may overlap at all (pipelining on)
inline bool lane_mode(const swr_context* c) { return c->lanes_ok && c->bin_stream != c->stream; }

int sync_streams(swr_context* c) {
    int rc = flush_raster(c, c->frame_no);      // everything drawn so far is on the streams
    if (!rc) rc = wait_stream(c, c->bin_stream, "binning stream");
    if (!rc) rc = wait_stream(c, c->stream, "raster stream");
    if (c->lanes_ok)
        for (hipStream_t ls : c->lane_stream)
            if (!rc && ls) rc = wait_stream(c, ls, "frame lane");
    if (rc) return sticky(c) ? sticky(c) : rc;
    c->synced_upto = c->posted;      // NOT frame_no: enqueue_frame may sync after it has numbered the frame it is about to post
    return SWR_OK;
}

int sync_copies(swr_context* c) {
    int rc = SWR_OK;
    if (c->ras_worker) rc = c->ras_worker->drain();    // posted swr_present copies
    if (!rc) rc = wait_stream(c, c->copy_stream[0], "colour copy stream");
    if (!rc) rc = wait_stream(c, c->copy_stream[1], "depth copy stream");
    if (rc) return sticky(c) ? sticky(c) : rc;
    return SWR_OK;
}

inline int tiles_of(const Target& t) { return t.tiles_x * t.tiles_y; }
inline uint32_t& pair_word(swr_context* c, uint64_t frame) { return c->h_pairs[frame % swr_context::PAIR_RING]; }
inline uint32_t& fill_word(swr_context* c, uint64_t frame) { return c->h_pairs[swr_context::PAIR_RING + 1 + frame % swr_context::PAIR_RING]; }

// Size the bins for the scene / target pair (both known; every stream idle).  Fixed-stride bins where k_bin can be used:
// a first guess of six times the mean load per tile (the host grows it when a frame overflows), exact bins otherwise.
int size_bins(swr_context* c) {
    if (!c->has_scene || !c->has_target) return SWR_OK;
    const int tiles = tiles_of(c->tg);
    const int64_t ntri = c->ni / 3;
    {   // the redo counter of the 32-bit depth keys starts from zero (swr_context::k32_ok is reset by the callers)
        int rc0 = ensure(c, c->redo_cnt, 16);
        if (rc0) return rc0;
        HIP_TRY(c, hipMemsetAsync(c->redo_cnt.p, 0, 16, c->stream));
        if ((rc0 = wait_stream(c, c->stream, "raster stream (redo counter)"))) return sticky(c) ? sticky(c) : rc0;
        c->h_pairs[2 * swr_context::PAIR_RING + 1] = 0u;
        c->sized_ntri = ntri; c->sized_tiles = tiles;
    }

    // Bands of large scenes (what one GPU of N renders) took round 2's chain of three short binning kernels in round 3: the one
    // long k_bin shared the chip with the band's raster worse.  With the perspective divide gone for affine transforms (k_bin
    // <.., AFF>) and no k_sort_bins launch on thin depth-only bands (the raster sorts its own bins) it is the other way round:
    // worst band of N = 2 / 4 / 8 on one GPU 47.1 / 29.3 / 24.5 us per frame against 52.8 / 35.7 / 26.3 with the chain
    // (profiles/r04/band_binmode_ab.txt) — k_bin everywhere.  Colour frames: N = 2 / 4 equal, N = 8 32.9 (chain) vs 36.4.
    // swr_debug_set(SWR_DEBUG_BIN_MODE, 1) forces the chain.
    const bool no_fixed = c->dbg_bin_mode == 1 || c->dbg_bin_mode == 3;
    const uint32_t cmax = (c->fixed_allowed && !no_fixed) ? fixed_cap_max(ntri, tiles) : 0u;
    int rc;
    if (cmax) {
        // a primitive enters a tile's region at most once, so a region of `ntri` entries can never overflow: small scenes
        // get that; large ones six times the mean load of a tile, at least 1024 entries (a fuller tile makes the host grow
        // the regions and redraw once)
        uint64_t want = std::max<uint64_t>(std::max<uint64_t>(64, (uint64_t)(6 * ntri / tiles)), (uint64_t)std::min<int64_t>(ntri, 1024));
        want = std::min<uint64_t>((want + 63) & ~63ull, cmax);
        if (!c->fixed_mode || c->cap_tile < want || c->cap_tile > cmax) c->cap_tile = (uint32_t)want;   // a grown region survives a new transform
        if ((rc = ensure_bins(c, (size_t)tiles * c->cap_tile))) return rc;
        const size_t fb = (size_t)(CNT_WORDS + tiles) * 4;
        for (int k = 0; k < swr_context::NFILL; k++) {
            if ((rc = ensure(c, c->fillbuf[k], fb))) return rc;
            HIP_TRY(c, hipMemsetAsync(c->fillbuf[k].p, 0, c->fillbuf[k].bytes, c->stream));
            c->fill_dirty[k] = false;
        }
        for (int l = 0; l < swr_context::NSLOT; l++)
            for (int k = 0; k < 2; k++) {
                if ((rc = ensure(c, c->lanefill[l][k], fb))) return rc;
                HIP_TRY(c, hipMemsetAsync(c->lanefill[l][k].p, 0, c->lanefill[l][k].bytes, c->stream));
                c->lanefill_dirty[l][k] = false;
            }
        if ((rc = wait_stream(c, c->stream, "raster stream (fill counters)"))) return sticky(c) ? sticky(c) : rc;
        c->fixed_mode = true;
        return SWR_OK;
    }
    c->fixed_mode = false;
    const uint64_t want = (uint64_t)ntri * 2 + 65536;
    return ensure_capacity(c, (uint32_t)std::min<uint64_t>(want, 0xFFFFFFF0ull));
}

DeviceFrame make_frame(swr_context* c, int si, uint64_t frame, const float m[16], uint32_t flags) {
    swr_context::Slot& sl = c->slot[si];
    DeviceFrame f{};
    f.vertices = (const swr_vertex*)c->vertices.p;
    f.indices = (const int64_t*)c->indices.p;
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
    f.index_count = c->ni;
    f.ntri = c->ni / 3;
    f.geo = (GeomRec*)sl.geo.p;
    f.geo_full = (GeomFull*)sl.geo_full.p;
    uint32_t* tb = (uint32_t*)sl.tilebuf.p;
    f.counters = tb;
    f.host_counters = c->h_pairs_dev + (frame % swr_context::PAIR_RING);   // word CNT_PAIRS (= 0) of this frame
    f.host_max = c->h_pairs_dev + swr_context::PAIR_RING;                  // one word, overwritten by every frame
    f.skip_sort = 0;
    f.defer_big = 0;
    f.redo_dev = (uint32_t*)c->redo_cnt.p;
    f.host_redo = c->h_pairs_dev + 2 * swr_context::PAIR_RING + 1;
    f.k32 = (c->k32_ok && c->dbg_k32 && f.redo_dev) ? 1 : 0;
    f.tile_count = tb + CNT_WORDS;
    f.tile_start = tb + CNT_WORDS + tiles_of(c->tg);
    f.tile_cursor = tb + CNT_WORDS + 2 * tiles_of(c->tg) + 1;
    f.ranges = (uint2*)sl.ranges.p;
    f.plan = plan_binning(f.ntri, tiles_of(c->tg), c->dbg_bin_mode == 3);
#ifndef SWR_TUNE_IDLE_BIN_G
#define SWR_TUNE_IDLE_BIN_G 512
#endif
    // An idle context (every earlier frame complete: a single frame, or the first of a burst) bins with twice the
    // workgroups: with nothing else on the chip k_bin is a chain of round trips at one wave per SIMD, and two waves per
    // SIMD overlap them; beside a raster the extra waves only take issue slots from it (round 2's sweeps), so frames
    // posted while others are in flight keep one workgroup per CU.
    if (c->fixed_mode && f.ntri > 0 && c->synced_upto == c->posted) {
        const int64_t groups = (f.ntri + 63) / 64;
        f.plan.G = (int)std::max<int64_t>(1, std::min<int64_t>(SWR_TUNE_IDLE_BIN_G, std::max<int64_t>(f.plan.G, (groups + 7) / 8)));
    }
    f.bin_matrix = (uint32_t*)sl.bin_matrix.p;
    f.live = (uint32_t*)sl.live.p;
    // Groups of the stream whose projected box misses this context's band (or the framebuffer) are skipped by
    // k_setup_hist; the test costs one lane-pass per workgroup, so it is always on (SWR_DEBUG_CULL = 0 switches it off).
    f.live_parity = c->dbg_cull != 0 ? 0 : -1;
    f.bins = (uint32_t*)sl.bins.p;
    f.capacity = c->capacity;
    f.fixed_bins = (c->fixed_mode && f.ntri > 0) ? 1 : 0;
    f.cap_tile = c->cap_tile;
    if (lane_mode(c)) {       // the lane's two blocks alternate: this frame's, and the one its k_bin zeroes for the lane's next frame
        const int par = (int)((frame / (uint64_t)swr_context::NSLOT) & 1u);
        f.fill = (uint32_t*)c->lanefill[si][par].p;
        f.fill_next = (uint32_t*)c->lanefill[si][par ^ 1].p;
    } else {
        f.fill = (uint32_t*)c->fillbuf[frame % swr_context::NFILL].p;
        f.fill_next = (uint32_t*)c->fillbuf[(frame + 1) % swr_context::NFILL].p;
    }
    f.host_fill = c->h_pairs_dev + swr_context::PAIR_RING + 1 + (frame % swr_context::PAIR_RING);
    f.biglist = (uint4*)sl.biglist.p;
    f.color = (uint8_t*)c->color[c->fb_cur].p;
    f.depth = (float*)c->depth[c->fb_cur].p;
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

// the raster (or point / clear pass) about to write framebuffer `fb` must not overtake a copy still reading it
int wait_for_copies_of(swr_context* c, int fb, hipStream_t s) {
    for (int img = 0; img < 2; img++)
        if (c->copy_recorded[fb][img]) HIP_TRY(c, hipStreamWaitEvent(s, c->copy_done[fb][img], 0));
    return SWR_OK;
}

int enqueue_frame(swr_context* c) {
    if (const int f = sticky(c)) return f;      // a failed context posts nothing more
    c->hp_begin();
    {
        const BinPlan plan = plan_binning(c->ni / 3, tiles_of(c->tg), c->dbg_bin_mode == 3);
        if (plan.use_lds && !c->fixed_mode) {
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
    c->fb_last = c->fb_cur;
    if (tiles_of(c->tg) == 0) {
        // an empty band (a group with more sub-contexts than tile rows hands these out): nothing to bin, raster or copy
        int rc = flush_raster(c, c->frame_no);
        if (rc) return rc;
        fill_word(c, c->frame_no) = 0;
        pair_word(c, c->frame_no++) = 0;
        c->posted = c->frame_no;
        c->bin_enqueued.store(c->frame_no); c->ras_enqueued.store(c->frame_no);
        c->draw_pending = true;
        return SWR_OK;
    }
    if (c->last_prim != SWR_PRIMITIVE_TRIANGLE) {
        // .vertices / .line: three small kernels, no binning; the pair counter reads 0
        int rc = sync_streams(c);
        if (rc) return rc;
        const uint64_t frame = c->frame_no++;
        DeviceFrame f = make_frame(c, 0, frame, c->last_m, c->last_flags);
        pair_word(c, frame) = 0;
        fill_word(c, frame) = 0;
        if ((rc = wait_for_copies_of(c, c->fb_cur, c->stream))) return rc;
        launch_points_or_lines(f, c->last_prim, c->stream);
        HIP_TRY(c, hipGetLastError());
        c->posted = c->frame_no;
        c->bin_enqueued.store(c->frame_no); c->ras_enqueued.store(c->frame_no);
        c->draw_pending = true;
        c->last_stream = c->stream;
        if (lane_mode(c)) c->fb_cur = (c->fb_last + 1) % swr_context::NFB;
        return SWR_OK;
    }
    const uint64_t frame = c->frame_no++;
    c->frame_presented[frame % swr_context::PAIR_RING] = false;
    const int si = (int)(frame % swr_context::NSLOT);
    c->last_slot = si;
    DeviceFrame f = make_frame(c, si, frame, c->last_m, c->last_flags);
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
    c->hp_lap(0);
    // Where k_sort_bins runs.  On the binning stream it is part of the chain that runs ahead of the raster; on
    // the raster stream (right before k_raster) the binning stream is free one kernel earlier.  Alternating A/B on
    // one box (tools/ab_sort_stream.py, tools/bt_bands.sh; untimed frames of cfg4, us per frame):
    //   whole 4K frame (4 080 tiles): 112 vs 122 -> binning stream;  half / quarter / eighth bands: 83 / 59 / 41
    //   vs 73 / 49 / 40 -> raster stream;  light frames (cfg2, 6 k triangles): 37 vs 43 -> binning stream.
    // -DSWR_TUNE_SORT_STREAM=0/1 forces either.
#ifndef SWR_TUNE_SORT_STREAM
#define SWR_TUNE_SORT_STREAM (-1)     // 0 / 1: k_sort_bins always on the binning / raster stream (tuning builds)
#endif
#ifndef SWR_TUNE_SORT_SPARSE
#define SWR_TUNE_SORT_SPARSE 0        // 1: launch k_sort_bins on sparse frames too
#endif
    constexpr int sort_stream_mode = SWR_TUNE_SORT_STREAM;
    const bool sort_on_raster_stream = sort_stream_mode >= 0 ? sort_stream_mode == 1
                                                            : (sb != sr && f.ntri >= 200000 && tiles_of(c->tg) < 3000);
    // Sparse frames need no k_sort_bins: when the fullest bin of the latest binned frame (a pinned word k_fill_lds
    // overwrites every frame; 0xFFFFFFFF after a new scene or target) fits two chunks, every tile of THIS frame is
    // expected to be walked by all four waves together (row-split mode), where the order inside the bin does not
    // matter — k_raster masks the class tags itself.  A wrong guess costs time, never pixels.  (cfg5: 0.321 -> 0.313 ms,
    // the app's sphere 18.6 -> 17.1 us; -DSWR_TUNE_SORT_SPARSE=1 sorts always.)
    constexpr bool sort_sparse = SWR_TUNE_SORT_SPARSE != 0;
    // (a word the GPU overwrites while frames are in flight: read it as what it is, a relaxed atomic)
    const uint32_t fullest = __atomic_load_n(&c->h_pairs[swr_context::PAIR_RING], __ATOMIC_RELAXED);
    if (!sort_sparse && sort_stream_mode < 0 && fullest <= 128u) f.skip_sort = 1;
    // bit 31 of that word: the latest rastered frame met triangles that cover hundreds of tiles — this frame's k_bin puts them on
    // the deferred list and k_sort_bins appends them per tile (swr_kernels.hip, BIN_BIG_TILES); 0xFFFFFFFF = nothing known yet
    f.defer_big = (fullest != 0xFFFFFFFFu && (fullest & 0x80000000u)) ? 1 : 0;
    // 32-bit depth keys: one tile in REDO_SAMPLE (8) reports when it had to be rastered again (~6x the time of a tile that did
    // not); from 2 % of the tiles on the 64-bit kernel is the cheaper one for this scene
    if (f.k32) {
        const uint32_t redo = __atomic_load_n(&c->h_pairs[2 * swr_context::PAIR_RING + 1], __ATOMIC_RELAXED);
        if ((uint64_t)redo * 8u * 50u > (uint64_t)tiles_of(c->tg)) { c->k32_ok = false; f.k32 = 0; }
    }
    // ... and on a small grid (a thin band: the frame is bound by its kernel chain, not by the chip) such a frame sorts its bins
    // inside the raster workgroups, in the LDS the narrow keys leave free: no k_sort_bins launch — unless the frame defers
    // triangles that cover hundreds of tiles, which k_sort_bins appends to the bins.  On a grid that fills the chip the launch
    // is the cheaper place for the sort (swr_kernels.hip, raster_tile; profiles/r04/insort_ab.txt).
    // (swr_debug_set SWR_DEBUG_RASTER_SORT: 0 never, 1 by grid size, 2 always)
    const bool insort_grid = c->dbg_insort == 2 || (c->dbg_insort == 1 && tiles_of(c->tg) <= 1536);
    if (insort_grid && frame_uses_k32(f) && !f.defer_big && !f.skip_sort && sort_stream_mode < 0) { f.insort = 1; f.skip_sort = 1; }
    const bool all = c->timing >= 2;
    if (f.ntri <= 0) { int rc = sync_streams(c); if (rc) return rc; pair_word(c, frame) = 0; }
    if (!f.fixed_bins) fill_word(c, frame) = 0;
    // fixed-stride bins: this frame's fill block must be zero when k_bin starts (the k_bin before it did that, unless that
    // frame took another path); k_bin leaves it dirty and zeroes the next frame's
    bool fill_memset = false;
    const bool lanes = lane_mode(c);
    if (f.fixed_bins && lanes) {
        const int par = (int)((frame / (uint64_t)swr_context::NSLOT) & 1u);
        fill_memset = c->lanefill_dirty[si][par];
        c->lanefill_dirty[si][par] = true;
        c->lanefill_dirty[si][par ^ 1] = false;
    } else if (f.fixed_bins) {
        const int fbk = (int)(frame % swr_context::NFILL);
        fill_memset = c->fill_dirty[fbk];
        c->fill_dirty[fbk] = true;
        c->fill_dirty[(fbk + 1) % swr_context::NFILL] = false;
    }
    const bool zero_tables = !f.fixed_bins && (!f.plan.use_lds || f.ntri <= 0);
    const size_t zero_bytes = (size_t)(CNT_WORDS + 3 * tiles_of(c->tg) + 1) * 4;
    hipEvent_t e0 = (ev && all) ? ev[0] : nullptr, e1 = (ev && all) ? ev[1] : nullptr, e2 = (ev && all) ? ev[2] : nullptr;
    if (lanes) {
        // ---- FRAME LANES: the whole frame, binning and raster, back to back on the lane's stream (see swr_context::lane_stream)
        hipStream_t S = c->lane_stream[si];
        const int inj = c->inject.exchange(0, std::memory_order_relaxed);      // swr_debug_fault
        auto fail_frame = [&](int rc) {      // nothing was enqueued for this frame: the context has failed (fatal) or the call reports
            c->posted = frame + 1;
            c->bin_enqueued.store(frame + 1, std::memory_order_release);
            c->ras_enqueued.store(frame + 1, std::memory_order_release);
            return rc;
        };
        if (inj == SWR_FAULT_ENQUEUE) return fail_frame(fatal(c, SWR_ERR_HIP, "frame %llu: injected enqueue failure (swr_debug_fault)", (unsigned long long)frame));
        if (inj == SWR_FAULT_LOST_EVENT) {
            const int rc = poll_event(c, c->slot[si].bin_done, "the lane's previous frame (injected: a completion that never arrives)", frame, true);
            if (rc) return fail_frame(rc);
        }
        // an un-waited burst is paced by markers: at most ~4 * PACE_EVERY frames in flight
        if (frame % (uint64_t)swr_context::PACE_EVERY == 0) {
            const uint64_t old = frame - 3 * (uint64_t)swr_context::PACE_EVERY;
            const int k = (int)((frame / (uint64_t)swr_context::PACE_EVERY) % 4);
            const int ko = (k + 1) % 4;          // (k - 3) mod 4
            if (frame >= 3 * (uint64_t)swr_context::PACE_EVERY && c->pace_frame[ko] == old && old >= c->synced_upto) {
                const int rc = poll_event(c, c->pace_ev[ko], "an earlier frame of the burst", old);
                if (rc) return fail_frame(rc);
            }
        }
        int rc = wait_for_copies_of(c, c->fb_cur, S);
        if (rc) return fail_frame(rc);
        auto enq = [&]() -> int {
            if (zero_tables) HIP_TRY(c, hipMemsetAsync(c->slot[si].tilebuf.p, 0, zero_bytes, S));
            if (fill_memset) HIP_TRY(c, hipMemsetAsync(f.fill, 0, (size_t)(CNT_WORDS + tiles_of(f.tg)) * 4, S));
            if (e0) HIP_TRY(c, hipEventRecord(e0, S));
            if (f.fixed_bins) {
                launch_bin(f, S, nullptr);
                if (e1) HIP_TRY(c, hipEventRecord(e1, S));
                if (e2) HIP_TRY(c, hipEventRecord(e2, S));
            } else {
                launch_setup_bin(f, S);
                if (e1) HIP_TRY(c, hipEventRecord(e1, S));
                launch_scan(f, S);
                if (e2) HIP_TRY(c, hipEventRecord(e2, S));
                launch_fill(f, S, nullptr);
            }
            if (!f.skip_sort) launch_sort_bins(f, S, nullptr);
            if (ev) HIP_TRY(c, hipEventRecord(ev[3], S));
            launch_raster(f, S, nullptr);
            if (ev) HIP_TRY(c, hipEventRecord(ev[4], S));
            if (frame % (uint64_t)swr_context::PACE_EVERY == 0) {
                const int k = (int)((frame / (uint64_t)swr_context::PACE_EVERY) % 4);
                HIP_TRY(c, hipEventRecord(c->pace_ev[k], S));
                c->pace_frame[k] = frame;
            }
            HIP_TRY(c, hipGetLastError());
            return SWR_OK;
        };
        rc = enq();
        if (rc) return fail_frame(fatal(c, rc, "enqueueing frame %llu failed: %s", (unsigned long long)frame, c->err.c_str()));
        c->draw_pending = true;
        c->posted = frame + 1;
        c->bin_enqueued.store(frame + 1, std::memory_order_release);
        c->ras_enqueued.store(frame + 1, std::memory_order_release);
        c->last_stream = S;
        c->fb_cur = (c->fb_last + 1) % swr_context::NFB;      // the next frame (another lane) renders into its own buffer
        return SWR_OK;
    }
    c->last_stream = sr;
    // ---- the binning stream's share of the frame ----
    // this slot's buffers are free again once the raster of NSLOT frames ago has read them (a full sync since then
    // settles it: every non-pipelined path syncs first)
    const bool slot_wait = sb != sr && frame >= (uint64_t)swr_context::NSLOT && frame - (uint64_t)swr_context::NSLOT >= c->synced_upto;
    // helpers or the caller's own thread?  (idle context: every earlier frame is complete)
    const bool streaming = sb != sr && (c->bin_worker || c->ras_worker) && !(c->inline_idle && c->synced_upto == c->posted);
    const bool paced = streaming && c->ras_worker != nullptr;      // cross-stream order by host polls instead of event waits
    auto bin_share = [c, f, si, sb, sr, frame, zero_tables, zero_bytes, e0, e1, e2, sort_on_raster_stream, slot_wait, paced, fill_memset]() -> int {
        swr_context::Slot& sl = c->slot[si];
        if (slot_wait) {
            // the first event-carrying raster at or after that frame (RAS_EVERY); its event must have been bound /
            // recorded (by ras_worker or the caller's thread) before it is polled / waited for
            const uint64_t prev = frame - (uint64_t)swr_context::NSLOT;
            const uint64_t ef = prev + (uint64_t)(swr_context::RAS_EVERY - 1) - prev % (uint64_t)swr_context::RAS_EVERY;
            { const int rc = wait_counter(c, c->ras_enqueued, ef, "the raster of an earlier frame"); if (rc) return rc; }
            swr_context::Slot& es = c->slot[ef % (uint64_t)swr_context::NSLOT];
            if (es.ras_event_frame == ef) {
                if (paced) { const int rc = poll_event(c, es.ras_done, "k_raster (its working set is needed again)", ef); if (rc) return rc; }
                else HIP_TRY(c, hipStreamWaitEvent(sb, es.ras_done, 0));
            }
        }
        // global-atomic fallback: counters must start at zero.  Empty scene: no binning kernel runs at all, so the
        // tile table (counts, starts, counters) is simply zeroed.
        if (zero_tables) HIP_TRY(c, hipMemsetAsync(sl.tilebuf.p, 0, zero_bytes, sb));
        if (fill_memset) HIP_TRY(c, hipMemsetAsync(f.fill, 0, (size_t)(CNT_WORDS + tiles_of(f.tg)) * 4, sb));
        if (e0) HIP_TRY(c, hipEventRecord(e0, sb));
        // bin_done = the completion of the chain's last kernel itself (bound at launch) where there is one
        hipEvent_t stop = (sb != sr && c->bind_events) ? sl.bin_done : nullptr;
        bool bound;
        if (f.fixed_bins) {
            // ONE launch: cull, setup, histogram, region reservation, fill (k_bin)
            bound = launch_bin(f, sb, (sort_on_raster_stream || f.skip_sort) ? stop : nullptr);
            if (e1) HIP_TRY(c, hipEventRecord(e1, sb));
            if (e2) HIP_TRY(c, hipEventRecord(e2, sb));
        } else {
            launch_setup_bin(f, sb);
            if (e1) HIP_TRY(c, hipEventRecord(e1, sb));
            launch_scan(f, sb);
            if (e2) HIP_TRY(c, hipEventRecord(e2, sb));
            bound = launch_fill(f, sb, (sort_on_raster_stream || f.skip_sort) ? stop : nullptr);
        }
        if (!sort_on_raster_stream && !f.skip_sort) bound = launch_sort_bins(f, sb, stop);
        if (sb != sr && !bound) HIP_TRY(c, hipEventRecord(sl.bin_done, sb));
        HIP_TRY(c, hipGetLastError());
        return SWR_OK;
    };
    if (frame >= (uint64_t)swr_context::RAS_RING) {
        const int rc = wait_counter(c, c->ras_enqueued, frame - (uint64_t)swr_context::RAS_RING, "the raster share of an earlier frame");
        if (rc) { c->frame_no = frame; return sticky(c) ? sticky(c) : rc; }     // nothing was posted for this frame
    }
    swr_context::RasJob& rj = c->ras_job[frame % swr_context::RAS_RING];
    rj.f = f; rj.ev3 = ev ? ev[3] : nullptr; rj.ev4 = ev ? ev[4] : nullptr; rj.si = si; rj.sort_here = sort_on_raster_stream;
    rj.fb = c->fb_cur;
    rj.paced = paced;
    c->draw_pending = true;   // the pair total lands in the frame's pinned word (written by the scan)
    c->posted = frame + 1;
    if (streaming) {
        // two helpers: one enqueues this frame's binning while the other enqueues the previous frame's raster.  One
        // helper (sub-contexts of a group: the group's per-device thread is this thread): the binning share runs here.
        auto bin_job = [c, bin_share, frame]() -> int {
            int rc = c->failed.load(std::memory_order_acquire);
            if (!rc) rc = bin_share();
            if (rc) {
                fatal(c, rc, "enqueueing the binning of frame %llu failed: %s", (unsigned long long)frame, tl_helper_thread ? "(helper thread)" : c->err.c_str());
                c->bin_error.store(rc, std::memory_order_relaxed);
            }
            c->bin_enqueued.store(frame + 1, std::memory_order_release);
            return rc;
        };
        if (c->bin_worker) c->bin_worker->post(bin_job);
        else (void)bin_job();                    // a failure is picked up by the raster share (bin_error) and by the next blocking call
        c->hp_lap(2);
        if (c->ras_worker) {
            c->ras_worker->post([c, frame]() -> int { return enqueue_raster_shares(c, frame + 1); });
            return SWR_OK;
        }
        return flush_raster(c, frame);            // frames < frame; this one follows with the next draw / sync / present
    }
    { int rc = bin_share(); if (rc) return rc; }
    c->bin_enqueued.store(frame + 1, std::memory_order_release);
    c->hp_lap(2);
    return flush_raster(c, frame + 1);
}

// Every frame below `upto` has its raster share on the raster stream when this returns.
int flush_raster(swr_context* c, uint64_t upto) {
    if (c->ras_worker) {                                  // every posted share (and present) has been enqueued
        const int rc = c->ras_worker->drain();
        if (rc) return rc;
    }
    return enqueue_raster_shares(c, std::min(upto, c->posted));   // frames of the one-stream path (never posted)
}

// The raster stream's share of frame g (caller's thread or ras_worker).
int raster_share(swr_context* c, uint64_t g) {
    { const int rc = wait_counter(c, c->bin_enqueued, g, "the binning share"); if (rc) return rc; }
    if (c->bin_error.load(std::memory_order_relaxed)) {
        const int rc = c->bin_error.exchange(0);
        return rc ? rc : SWR_ERR_HIP;          // the text was recorded by the thread that failed
    }
    swr_context::RasJob& rj = c->ras_job[g % swr_context::RAS_RING];
    swr_context::Slot& sl = c->slot[rj.si];
    hipStream_t sb = c->bin_stream, sr = c->stream;
    const int inj = c->inject.exchange(0, std::memory_order_relaxed);      // swr_debug_fault
    if (inj == SWR_FAULT_ENQUEUE) return fatal(c, SWR_ERR_HIP, "frame %llu: injected enqueue failure (swr_debug_fault)", (unsigned long long)g);
    c->hp_begin_r();
    if (sb != sr) {
        if (rj.paced || inj == SWR_FAULT_LOST_EVENT) {
            const int rc = poll_event(c, sl.bin_done, "the binning (k_bin / k_sort_bins)", g, inj == SWR_FAULT_LOST_EVENT);
            if (rc) return rc;
        } else HIP_TRY(c, hipStreamWaitEvent(sr, sl.bin_done, 0));
    } else if (inj == SWR_FAULT_LOST_EVENT) {
        const int rc = poll_event(c, sl.bin_done, "the binning (k_bin / k_sort_bins)", g, true);
        if (rc) return rc;
    }
    c->hp_lap_r(3);
    { int rc = wait_for_copies_of(c, rj.fb, sr); if (rc) return rc; }
    if (rj.sort_here) launch_sort_bins(rj.f, sr);
    if (rj.ev3) HIP_TRY(c, hipEventRecord(rj.ev3, sr));
    const bool carries = sb != sr && g % (uint64_t)swr_context::RAS_EVERY == (uint64_t)(swr_context::RAS_EVERY - 1);
    const bool bound = launch_raster(rj.f, sr, (carries && c->bind_events) ? sl.ras_done : nullptr);
    if (rj.ev4) HIP_TRY(c, hipEventRecord(rj.ev4, sr));
    c->hp_lap_r(4);
    if (carries) { if (!bound) HIP_TRY(c, hipEventRecord(sl.ras_done, sr)); sl.ras_event_frame = g; }
    c->hp_lap_r(5);
    c->hp_frames++;
    HIP_TRY(c, hipGetLastError());
    return SWR_OK;
}

// The raster shares of every frame below `upto` that have not been enqueued yet.  ANY failure fails the context and
// still advances ras_enqueued: the binning helper (slot wait), swr_draw (back-pressure) and the destructor wait on it.
int enqueue_raster_shares(swr_context* c, uint64_t upto) {
    for (;;) {
        const uint64_t g = c->ras_enqueued.load(std::memory_order_relaxed);
        if (g >= upto) return SWR_OK;
        int rc = c->failed.load(std::memory_order_acquire);
        if (!rc) rc = raster_share(c, g);
        if (rc) {
            // (no-op when the cause has been recorded already: a helper's HIP_TRY, an expired wait)
            fatal(c, rc, "enqueueing the raster of frame %llu failed: %s", (unsigned long long)g, tl_helper_thread ? "(helper thread)" : c->err.c_str());
            c->ras_enqueued.store(std::max(upto, g + 1), std::memory_order_release);
            return rc;
        }
        c->ras_enqueued.store(g + 1, std::memory_order_release);
    }
}

// Is `p` page-locked memory HIP knows (hipHostMalloc / hipHostRegister)?  Then hipMemcpyAsync into it is a true
// asynchronous DMA; anything else would be staged by the runtime and block the caller.
bool is_pinned(const void* p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();   // unknown pointer = plain pageable memory: not an error
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

int ensure_stage(swr_context* c, int img) {
    for (int k = 0; k < 2; k++) {
        if (!c->stage[img][k]) HIP_TRY(c, hipHostMalloc(&c->stage[img][k], swr_context::STAGE_BYTES, hipHostMallocDefault));
        if (!c->stage_ev[img][k]) HIP_TRY(c, hipEventCreateWithFlags(&c->stage_ev[img][k], hipEventDisableTiming));
    }
    return SWR_OK;
}

// Enqueue the copy of image `img` (0 colour, 1 depth) of framebuffer `fb`, this context's band only, into rows
// [row_begin, row_end) of the caller's full-size host image, on the image's own copy stream, behind frame_done[fb].
// Page-locked destination: ONE hipMemcpyAsync straight into the caller's rows (returns at once).  Pageable
// destination: pipelined through two pinned 8 MiB chunks (D2H of chunk k+1 overlaps the memcpy of chunk k; blocks).
int copy_band(swr_context* c, int fb, int img, void* dst_full, uint64_t frame) {
    const size_t row = (size_t)c->tg.width * 4;
    const size_t bytes = (size_t)(c->tg.row_end - c->tg.row_begin) * row;
    if (!bytes) return SWR_OK;
    uint8_t* dst = (uint8_t*)dst_full + (size_t)c->tg.row_begin * row;
    const uint8_t* src = (const uint8_t*)(img == 0 ? c->color[fb].p : c->depth[fb].p);
    hipStream_t s = c->copy_stream[img];
    HIP_TRY(c, hipStreamWaitEvent(s, c->frame_done[fb], 0));
    if (is_pinned(dst)) {
        HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
    } else {
        int rc = ensure_stage(c, img);
        if (rc) return rc;
        const size_t CH = swr_context::STAGE_BYTES;
        const size_t n = (bytes + CH - 1) / CH;
        for (size_t k = 0; k <= n; k++) {
            if (k < n) {
                const size_t len = std::min(CH, bytes - k * CH);
                HIP_TRY(c, hipMemcpyAsync(c->stage[img][k & 1], src + k * CH, len, hipMemcpyDeviceToHost, s));
                HIP_TRY(c, hipEventRecord(c->stage_ev[img][k & 1], s));
            }
            if (k > 0) {
                const size_t j = k - 1, len = std::min(CH, bytes - j * CH);
                { const int rcw = poll_event(c, c->stage_ev[img][j & 1], "a staged copy to the host", frame); if (rcw) return rcw; }
                memcpy(dst + j * CH, c->stage[img][j & 1], len);
            }
        }
    }
    HIP_TRY(c, hipEventRecord(c->copy_done[fb][img], s));
    c->copy_recorded[fb][img] = true;
    return SWR_OK;
}

// ---- single-device implementations of the entry points ----------------------------------------------------------
int check_frames(swr_context* c);
int enqueue_present(swr_context* c, void* color_full, float* depth_full);

// oneshot: the scene is uploaded for ONE frame (swr_render without a scene identity, the reference's calling pattern):
// the triangle stream keeps index order — the Morton sort costs more than it saves a single frame (cfg4: build 0.49 ->
// 0.23 ms, the frame 0.154 -> 0.177 ms) — and it is built chunk by chunk behind the copy of the index array.
int single_scene_upload(swr_context* c, const swr_vertex* vertices, int64_t vertex_count,
                        const int64_t* indices, int64_t index_count, bool oneshot = false) {
    if (const int f = sticky(c)) return f;
    if (vertex_count < 0 || index_count < 0 || (index_count > 0 && (!indices || !vertices)))
        return fail(c, SWR_ERR_BAD_ARG, "swr_scene_upload: bad vertex/index arguments");
    if (index_count / 3 >= 0xFFFFFFFFll || vertex_count > 0xFFFFFFFFll)
        return fail(c, SWR_ERR_UNSUPPORTED, "more than 2^32-2 primitives or 2^32 vertices");
    HIP_TRY(c, hipSetDevice(c->device));
    { int rcs = check_frames(c); if (rcs) return rcs; }     // (an overflowed frame presented just before is repaired / reported first)
    c->has_scene = false;
    c->has_attrs = false;
    c->draw_pending = false;
    c->present_pending = false;
    int rc;
    if ((rc = ensure(c, c->vertices, (size_t)vertex_count * sizeof(swr_vertex)))) return rc;
    if ((rc = ensure(c, c->indices, (size_t)index_count * 8))) return rc;
    if ((rc = ensure(c, c->tri_rgb, (size_t)index_count * 16))) return rc;
    const int64_t ntri = index_count / 3;
    // SWR_DEBUG_STREAM_ORDER 0: keep index order (the original index still travels in GeomRec.flags);
    // -1: behave as for a scene of 2^24 primitives or more (no reordering, slot == index) — test hook
    const int sort_mode = c->dbg_stream_order;
    const bool reorder = sort_mode == 1 && !oneshot && ntri > 1 && ntri < SORT_MAX_TRIS;
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
        if ((rc = ensure(c, sl.biglist, (size_t)1024 * 16))) return rc;
        if ((rc = ensure(c, sl.live, (size_t)((index_count / 3 + 63) / 64 + 2 * 1024 + 2) * 4))) return rc;   // [G <= 1024][1 + per]
        if ((rc = ensure(c, sl.tilebuf, (size_t)(CNT_WORDS + 3 * std::max(1, tiles_of(c->tg)) + 1) * 4))) return rc;
    }
    for (auto& e : c->up_ev) if (!e) HIP_TRY(c, hipEventCreate(&e));
    HIP_TRY(c, hipEventRecord(c->up_ev[0], c->stream));
    if (vertex_count)
        HIP_TRY(c, hipMemcpyAsync(c->vertices.p, vertices, (size_t)vertex_count * sizeof(swr_vertex),
                                  hipMemcpyHostToDevice, c->stream));
    StreamBuild b{};
    b.vertices = (const swr_vertex*)c->vertices.p; b.nv = vertex_count;
    b.indices = (const int64_t*)c->indices.p; b.ntri = ntri;
    b.sort = reorder;
    b.scratch = (uint32_t*)c->stream_scratch.p;
    b.sort_temp = c->sort_temp.p; b.sort_temp_bytes = sort_bytes;
    b.tri_xyz = (float4*)c->tri_xyz.p; b.tri_rgb = (float4*)c->tri_rgb.p;
    b.inv = (uint32_t*)c->inv.p; b.box64 = (float4*)c->box64.p;
    uint32_t* const bad = (uint32_t*)c->slot[0].tilebuf.p;     // CNT_BAD_INDEX lives in the first slot's counter words
    HIP_TRY(c, hipMemsetAsync(bad, 0, CNT_WORDS * 4, c->stream));
    // One-shot scenes with an index array worth cutting up (>= 4 MiB): the index check and the stream of chunk k run on
    // the binning stream while chunk k+1 is on the link (a copy from pageable memory returns when its bytes are staged)
    hipStream_t side = c->bin_stream_own && c->bin_stream_own != c->stream ? c->bin_stream_own : nullptr;
    const int64_t chunk_min = c->dbg_oneshot_min_tris;                   // (SWR_DEBUG_ONESHOT_MIN_TRIS: chunk small scenes too)
    const bool chunked = oneshot && side && sort_mode != -1 && ntri >= chunk_min && ntri < SORT_MAX_TRIS;
    if (chunked) {
        // two chunks, the second one small: every extra copy call from pageable memory costs the link ~30 us, the last
        // chunk's build is the tail (cfg4, whole call: {100} 3.10 ms, {70,100} 3.06, {50,82,100} 3.07, four quarters 3.12)
#ifndef SWR_TUNE_ONESHOT_CUTS
#define SWR_TUNE_ONESHOT_CUTS {70, 100}
#endif
        constexpr int cuts[] = SWR_TUNE_ONESHOT_CUTS;          // per cent of the primitives
        if (!c->up_chunk_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->up_chunk_ev, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->up_chunk_ev, c->stream));           // (vertices resident, counter words zero)
        HIP_TRY(c, hipStreamWaitEvent(side, c->up_chunk_ev, 0));
        int64_t t0 = 0;
        for (const int cut : cuts) {
            const int64_t t1 = cut >= 100 ? ntri : std::min(ntri, (ntri * cut / 100 + 63) / 64 * 64);
            if (t1 <= t0) continue;
            HIP_TRY(c, hipMemcpyAsync((int64_t*)c->indices.p + 3 * t0, indices + 3 * t0, (size_t)(t1 - t0) * 24, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipEventRecord(c->up_chunk_ev, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(side, c->up_chunk_ev, 0));
            launch_validate_indices((const int64_t*)c->indices.p + 3 * t0, 3 * (t1 - t0), vertex_count, bad, side);
            HIP_TRY(c, launch_build_stream_range(b, t0, t1, side));
            t0 = t1;
        }
        HIP_TRY(c, hipEventRecord(c->up_ev[1], c->stream));
        HIP_TRY(c, hipEventRecord(c->up_chunk_ev, side));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->up_chunk_ev, 0));
        c->reordered = true;                                          // (identity order; the original index travels in GeomRec.flags)
    } else {
        if (index_count)
            HIP_TRY(c, hipMemcpyAsync(c->indices.p, indices, (size_t)index_count * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(c->up_ev[1], c->stream));
        // index range check (Swift array subscript would trap, Renderer.swift:226)
        launch_validate_indices((const int64_t*)c->indices.p, index_count, vertex_count, bad, c->stream);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, launch_build_stream(b, c->stream));
        c->reordered = ntri > 0 && ntri < SORT_MAX_TRIS && sort_mode != -1;   // original index travels in GeomRec.flags
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_misc, c->slot[0].tilebuf.p, CNT_WORDS * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipEventRecord(c->up_ev[2], c->stream));
    if ((rc = wait_stream(c, c->stream, "scene upload"))) return sticky(c) ? sticky(c) : rc;
    (void)hipEventElapsedTime(&c->up_h2d_ms, c->up_ev[0], c->up_ev[1]);
    (void)hipEventElapsedTime(&c->up_build_ms, c->up_ev[1], c->up_ev[2]);
    if (c->h_misc[CNT_BAD_INDEX])
        return fail(c, SWR_ERR_INDEX_RANGE, "an index is outside [0, %lld)", (long long)vertex_count);
    c->nv = vertex_count;
    c->ni = index_count;
    c->has_scene = true;
    c->h_pairs[swr_context::PAIR_RING] = 0xFFFFFFFFu;       // fullest bin of the new scene: unknown (sort)
    // What earlier frames taught the context about this (primitive count, tile grid) pair survives the upload — the regions a
    // crowded tile made the host grow, a scene that needed exact bins, a scene whose depth frames belong on the 64-bit keys: the
    // reference's calling pattern uploads the SAME mesh on every call (GpuRenderer.swift:41-71), and re-learning it would cost
    // every such call a second frame.  A different primitive count starts from the first guess again.
    if (index_count / 3 != c->sized_ntri) {
        c->fixed_allowed = true;
        c->fixed_mode = false;                              // (re-sized below or at swr_target_set)
        c->k32_ok = true;
    }
    if (!c->has_target) {
        const uint64_t want = (uint64_t)(index_count / 3) * 2 + 65536;
        return ensure_capacity(c, (uint32_t)std::min<uint64_t>(want, 0xFFFFFFF0ull));
    }
    return size_bins(c);
}

int single_scene_attributes(swr_context* c, const swr_vertex_attr* attributes, int64_t vertex_count) {
    if (const int f = sticky(c)) return f;
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
    if ((rc = wait_stream(c, c->stream, "attribute upload"))) return sticky(c) ? sticky(c) : rc;
    c->has_attrs = true;
    return SWR_OK;
}

int single_material_set(swr_context* c, const swr_material* m) {
    if (!m) { c->material = swr_material{}; return SWR_OK; }
    if (m->shader != SWR_SHADER_PASSTHROUGH && m->shader != SWR_SHADER_PHONG && m->shader != SWR_SHADER_TEXTURED_PHONG)
        return fail(c, SWR_ERR_UNSUPPORTED, "unknown shader %d", m->shader);
    if (m->shininess_log2 < 0 || m->shininess_log2 > 16)
        return fail(c, SWR_ERR_BAD_ARG, "shininess_log2 %d outside [0,16]", m->shininess_log2);
    c->material = *m;      // read at the next swr_draw (by value into the kernel arguments)
    return SWR_OK;
}

int single_texture_upload(swr_context* c, const void* bgra8, int32_t width, int32_t height) {
    if (const int f = sticky(c)) return f;
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
    if ((rc = wait_stream(c, c->stream, "texture upload"))) return sticky(c) ? sticky(c) : rc;
    c->tex_w = width; c->tex_h = height;
    return SWR_OK;
}

int check_target_args(swr_context* c, int64_t width, int64_t height, int64_t row_begin, int64_t row_end) {
    if (width <= 0 || height <= 0 || width > 65535 || height > 65535)
        return fail(c, SWR_ERR_BAD_ARG, "bad framebuffer size %lldx%lld", (long long)width, (long long)height);
    if (row_begin < 0 || row_end > height || row_begin > row_end || (row_begin % TILE_H) != 0)
        return fail(c, SWR_ERR_BAD_ARG, "bad band [%lld,%lld): row_begin must be a multiple of %d",
                    (long long)row_begin, (long long)row_end, TILE_H);
    return SWR_OK;
}

int single_target_set(swr_context* c, int64_t width, int64_t height, int64_t row_begin, int64_t row_end) {
    if (const int f = sticky(c)) return f;
    int rc = check_target_args(c, width, height, row_begin, row_end);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->has_target && c->tg.width == width && c->tg.height == height && c->tg.row_begin == row_begin && c->tg.row_end == row_end)
        return SWR_OK;          // the same target again (swr_render every frame): nothing to resize, nothing to wait for
    if ((rc = check_frames(c)) || (rc = sync_copies(c))) return rc;      // (an overflowed frame presented just before is repaired / reported first)
    c->draw_pending = false;
    c->present_pending = false;
    Target t;
    t.width = (int32_t)width; t.height = (int32_t)height;
    t.row_begin = (int32_t)row_begin; t.row_end = (int32_t)row_end;
    t.tiles_x = (int32_t)((width + TILE_W - 1) / TILE_W);
    t.tiles_y = (int32_t)((row_end - row_begin + TILE_H - 1) / TILE_H);
    const size_t px = (size_t)width * (size_t)(row_end - row_begin);
    for (int fb = 0; fb < swr_context::NFB; fb++) {
        if ((rc = ensure(c, c->color[fb], px * 4))) return rc;
        if ((rc = ensure(c, c->depth[fb], px * 4))) return rc;
        c->copy_recorded[fb][0] = c->copy_recorded[fb][1] = false;
    }
    c->fb_cur = c->fb_last = 0;
    for (auto& sl : c->slot)
        if ((rc = ensure(c, sl.tilebuf, (size_t)(CNT_WORDS + 3 * std::max(1, tiles_of(t)) + 1) * 4))) return rc;
    c->tg = t;
    c->has_target = true;
    c->h_pairs[swr_context::PAIR_RING] = 0xFFFFFFFFu;       // fullest bin on the new target: unknown (sort)
    if (tiles_of(t) != c->sized_tiles) {                    // (see single_scene_upload)
        c->fixed_allowed = true;
        c->fixed_mode = false;
        c->k32_ok = true;
    }
    return size_bins(c);
}

int check_draw_args(swr_context* c, uint32_t flags, int32_t primitive_type) {
    if (primitive_type != SWR_PRIMITIVE_TRIANGLE && primitive_type != SWR_PRIMITIVE_LINE &&
        primitive_type != SWR_PRIMITIVE_VERTICES)
        return fail(c, SWR_ERR_UNSUPPORTED, "unknown primitive type %d", primitive_type);
    if (flags & ~(uint32_t)(SWR_FLAG_DEPTH_TEST | SWR_FLAG_NO_COLOR | SWR_FLAG_METAL_RULES | SWR_FLAG_REAL_LINES))
        return fail(c, SWR_ERR_BAD_ARG, "unknown flag bits 0x%x", flags);
    if ((flags & SWR_FLAG_REAL_LINES) && primitive_type != SWR_PRIMITIVE_LINE)
        return fail(c, SWR_ERR_BAD_ARG, "SWR_FLAG_REAL_LINES only applies to .line primitives");
    return SWR_OK;
}

int single_draw(swr_context* c, const float transform[16], uint32_t flags, int32_t primitive_type) {
    int rc = check_draw_args(c, flags, primitive_type);
    if (rc) return rc;
    {
        const int per = primitive_type == SWR_PRIMITIVE_LINE ? 2 : 3;          // verticesCount, Renderer.swift:179-188
        if (c->has_scene && c->ni % per != 0)                                  // assert, Renderer.swift:209
            return fail(c, SWR_ERR_INDEX_COUNT, "index_count %lld is not a multiple of %d", (long long)c->ni, per);
    }
    if (!c->has_scene || !c->has_target)
        return fail(c, SWR_ERR_NO_SCENE, "swr_draw needs swr_scene_upload and swr_target_set first");
    if (primitive_type == SWR_PRIMITIVE_TRIANGLE && !(flags & SWR_FLAG_NO_COLOR) &&
        c->material.shader != SWR_SHADER_PASSTHROUGH) {
        if (!c->has_attrs)
            return fail(c, SWR_ERR_BAD_ARG, "the material needs vertex attributes (swr_scene_attributes)");
        if (c->material.shader == SWR_SHADER_TEXTURED_PHONG && c->tex_w <= 0)
            return fail(c, SWR_ERR_BAD_ARG, "the material needs a texture (swr_texture_upload)");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->frame_no - c->frames_checked >= (uint64_t)swr_context::PAIR_RING - 2) {
        // the ring of per-frame pair totals is about to wrap over frames nobody has looked at: look now (before
        // the new transform replaces the one a redo of the last frame would need)
        if ((rc = check_frames(c))) return rc;
    }
    memcpy(c->last_m, transform, sizeof c->last_m);
    c->last_flags = flags;
    c->last_prim = primitive_type;
    return enqueue_frame(c);
}

// swr_present of the frame in fb_last: both images in flight together, each on its own copy stream
int enqueue_present(swr_context* c, void* color_full, float* depth_full) {
    const int fb = c->fb_last;
    if (tiles_of(c->tg) == 0) return SWR_OK;
    const bool want_color = color_full && !(c->last_flags & SWR_FLAG_NO_COLOR);
    const uint64_t frame = c->frame_no ? c->frame_no - 1 : 0;      // the frame being presented (for messages)
    hipStream_t fs = c->last_stream ? c->last_stream : c->stream;        // the stream that carries the frame's raster
    auto copies = [c, fb, want_color, color_full, depth_full, frame, fs]() -> int {
        int rc;
        HIP_TRY(c, hipEventRecord(c->frame_done[fb], fs));
        if (want_color && (rc = copy_band(c, fb, 0, color_full, frame))) return rc;
        if (depth_full && (rc = copy_band(c, fb, 1, depth_full, frame))) return rc;
        return SWR_OK;
    };
    if (lane_mode(c)) {
        // frame lanes: the frame is on its stream already, and the next draw went (or goes) to another buffer anyway
        int rc = flush_raster(c, c->frame_no);
        if (rc || (rc = copies())) return rc;
        if (c->frame_no) c->frame_presented[(c->frame_no - 1) % swr_context::PAIR_RING] = true;
        return SWR_OK;
    }
    if (c->ras_worker) c->ras_worker->post(copies);   // behind the frame's raster share, in order
    else {
        int rc = flush_raster(c, c->frame_no);        // the frame being presented must be on the raster stream
        if (rc || (rc = copies())) return rc;
    }
    c->fb_cur = fb == 0 ? 1 : 0;     // the next frame renders into the other framebuffer while this one is being copied
    if (c->frame_no) c->frame_presented[(c->frame_no - 1) % swr_context::PAIR_RING] = true;
    return SWR_OK;
}

int single_present(swr_context* c, void* color_full, float* depth_full) {
    if (const int f = sticky(c)) return f;
    if (!c->has_target) return fail(c, SWR_ERR_NO_SCENE, "swr_present needs swr_target_set and a swr_draw first");
    if (!color_full && !depth_full) return fail(c, SWR_ERR_BAD_ARG, "swr_present: both image pointers are NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    c->present_color = color_full;
    c->present_depth = depth_full;
    c->present_pending = true;
    return enqueue_present(c, color_full, depth_full);
}

// Look at the pair totals of every frame since the last look.  The LAST frame is repaired here (bins grown, frame
// redrawn and, if it had been presented, copied again); an earlier frame of an un-waited burst that overflowed was
// rastered empty — the bins are grown and SWR_ERR_FRAME_DROPPED is returned once everything else is in order.
int check_frames(swr_context* c) {
    bool dropped = false;
    uint64_t dropped_frame = 0;
    uint32_t dropped_pairs = 0;
    auto finish = [&]() -> int {
        harvest(c);
        if (dropped)
            return fail(c, SWR_ERR_FRAME_DROPPED, "frame %llu of an un-waited burst overflowed the bin capacity (%u entries) and "
                        "was rastered empty; the bins have been grown — redraw it", (unsigned long long)dropped_frame, dropped_pairs);
        return SWR_OK;
    };
    // what a frame needed against what the bins give: exact bins count (triangle,tile) pairs, fixed-stride bins the
    // entries of the fullest tile region
    auto used = [&](uint64_t f) { return c->fixed_mode ? fill_word(c, f) : pair_word(c, f); };
    auto limit = [&]() { return c->fixed_mode ? c->cap_tile : c->capacity; };
    for (int attempt = 0; attempt < 8; attempt++) {
        int rc;
        if ((rc = sync_streams(c))) return rc;
        if (!c->draw_pending || c->frame_no == 0) { c->frames_checked = c->frame_no; return finish(); }
        const uint64_t L = c->frame_no - 1;
        const uint32_t pairs = used(L);
        uint32_t need = pairs;
        uint32_t total_pairs = pair_word(c, L);               // (for the switch to exact bins)
        for (uint64_t f = c->frames_checked; f < L; f++) {
            const uint32_t pf = used(f);
            if (pf > limit()) { need = std::max(need, pf); total_pairs = std::max(total_pairs, pair_word(c, f)); }
            // an earlier frame that overflowed was rastered empty: that matters only if it was copied to the host
            if (pf > limit() && c->frame_presented[f % swr_context::PAIR_RING]) {
                if (!dropped || pf > dropped_pairs) { dropped_frame = f; dropped_pairs = pf; }
                dropped = true;
            }
        }
        c->frames_checked = L;
        const uint32_t old_limit = limit();
        if (need > old_limit) {
            if ((rc = sync_copies(c))) return rc;
            if (c->fixed_mode) {
                const uint32_t cmax = fixed_cap_max(c->ni / 3, tiles_of(c->tg));
                if (need > cmax) {
                    // a tile needs more than a fixed region can hold: exact-size bins (four binning kernels) from here on
                    c->fixed_mode = false;
                    c->fixed_allowed = false;
                    const uint64_t want = (uint64_t)total_pairs + total_pairs / 4 + 1024;
                    if (want > 0xFFFFFFF0ull) return fail(c, SWR_ERR_UNSUPPORTED, "too many (triangle,tile) pairs: %u", total_pairs);
                    if ((rc = ensure_capacity(c, (uint32_t)want))) return rc;
                } else {
                    const uint64_t want = std::min<uint64_t>(cmax, (((uint64_t)need + need / 4 + 64) + 63) & ~63ull);
                    if ((rc = ensure_bins(c, (size_t)tiles_of(c->tg) * (size_t)want))) return rc;
                    c->cap_tile = (uint32_t)want;
                }
            } else {
                const uint64_t want = (uint64_t)need + need / 4 + 1024;
                if (want > 0xFFFFFFF0ull) return fail(c, SWR_ERR_UNSUPPORTED, "too many (triangle,tile) pairs: %u", need);
                if ((rc = ensure_capacity(c, (uint32_t)want))) return rc;
            }
        }
        if (pairs <= old_limit) {
            c->draw_pending = false;
            c->present_pending = false;        // the last frame is verified: a later repair must not copy into a stale destination
            c->frames_checked = c->frame_no;
            c->last.tile_pairs = pair_word(c, L);
            c->last.tiles = tiles_of(c->tg);
            c->last.triangles = c->ni / 3;
            return finish();
        }
        // the last frame overflowed: redraw it into the same framebuffer, copy it again
        const bool was_presented = c->present_pending && c->frame_presented[L % swr_context::PAIR_RING];
        c->fb_cur = c->fb_last;
        if ((rc = enqueue_frame(c))) return rc;
        if (was_presented && (rc = enqueue_present(c, c->present_color, c->present_depth))) return rc;
    }
    return fail(c, SWR_ERR_HIP, "pair list kept overflowing");
}

int single_sync(swr_context* c) {
    if (const int f = sticky(c)) return f;
    HIP_TRY(c, hipSetDevice(c->device));
    return check_frames(c);
}

int single_present_wait(swr_context* c) {
    if (const int f = sticky(c)) return f;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = check_frames(c);
    if (rc) return rc;
    if ((rc = sync_copies(c))) return rc;
    c->present_pending = false;
    return SWR_OK;
}

int single_read(swr_context* c, int img, void* dst) {
    if (const int f = sticky(c)) return f;
    if (!c->has_target) return fail(c, SWR_ERR_NO_SCENE, "swr_read_* needs swr_target_set first");
    int rc = single_sync(c);
    if (rc) return rc;
    if (tiles_of(c->tg) == 0) return SWR_OK;
    const int fb = c->fb_last;
    HIP_TRY(c, hipEventRecord(c->frame_done[fb], c->stream));
    if ((rc = copy_band(c, fb, img, dst, c->frame_no ? c->frame_no - 1 : 0))) return rc;
    if ((rc = wait_stream(c, c->copy_stream[img], "copy stream"))) return sticky(c) ? sticky(c) : rc;
    return SWR_OK;
}

void destroy_single(swr_context* c) {
    hipSetDevice(c->device);
    // helper jobs are bounded (every wait in them is) and give up at once on a failed context: the drains return
    if (c->ras_worker) { c->ras_worker->drain(); c->ras_worker->stop(); delete c->ras_worker; c->ras_worker = nullptr; }
    if (c->bin_worker) { c->bin_worker->drain(); c->bin_worker->stop(); delete c->bin_worker; c->bin_worker = nullptr; }
    // bounded too: a GPU that never finishes must not hang the destructor (what it still owns is then leaked)
    c->wait_budget_ms = std::min<uint32_t>(c->wait_budget_ms, 5000u);
    auto drain = [c](hipStream_t st) {           // (looks at the stream, not at the failed flag: the GPU may be fine)
        if (!st) return true;
        Deadline d(c);
        for (;;) {
            const hipError_t e = hipStreamQuery(st);
            if (e == hipSuccess) return true;
            if (e != hipErrorNotReady) return false;
            (void)hipGetLastError();
            if (!d.pause()) return false;
        }
    };
    bool drained = drain(c->bin_stream) && drain(c->stream) && drain(c->copy_stream[0]) && drain(c->copy_stream[1]);
    for (hipStream_t ls : c->lane_stream) drained = drained && drain(ls);
    if (!drained) {
        fprintf(stderr, "[swr] context on device %d destroyed in a failed state (%s); device memory it may still be using is not freed\n",
                c->device, c->failed_msg.c_str());
        delete c;
        return;
    }
    if (c->hp_on && c->hp_frames)
        fprintf(stderr, "[swr host profile] device %d, %llu frames, us/frame: prepare %.2f | wait(slot) %.2f | binning launches %.2f | "
                        "event record+wait %.2f | raster-stream launches %.2f | record(ras_done) %.2f\n", c->device,
                (unsigned long long)c->hp_frames, c->hp_t[0] / c->hp_frames, c->hp_t[1] / c->hp_frames, c->hp_t[2] / c->hp_frames,
                c->hp_t[3] / c->hp_frames, c->hp_t[4] / c->hp_frames, c->hp_t[5] / c->hp_frames);
    DevBuf* bufs[] = {&c->redo_cnt, &c->vertices, &c->indices, &c->tri_rgb, &c->tri_xyz, &c->inv, &c->box64, &c->stream_scratch, &c->sort_temp,
                      &c->attrs, &c->tri_nrm, &c->texture, &c->texture_bytes};
    for (DevBuf* b : bufs) if (b->p) hipFree(b->p);
    for (DevBuf& b : c->color) if (b.p) hipFree(b.p);
    for (DevBuf& b : c->depth) if (b.p) hipFree(b.p);
    for (DevBuf& b : c->fillbuf) if (b.p) hipFree(b.p);
    for (auto& lf : c->lanefill) for (DevBuf& b : lf) if (b.p) hipFree(b.p);
    for (hipStream_t ls : c->lane_stream) if (ls) hipStreamDestroy(ls);
    for (hipEvent_t e : c->pace_ev) if (e) hipEventDestroy(e);
    for (auto& sl : c->slot) {
        DevBuf* sb[] = {&sl.geo, &sl.geo_full, &sl.ranges, &sl.bins, &sl.bin_matrix, &sl.live, &sl.tilebuf, &sl.biglist};
        for (DevBuf* b : sb) if (b->p) hipFree(b->p);
        if (sl.bin_done) hipEventDestroy(sl.bin_done);
        if (sl.ras_done) hipEventDestroy(sl.ras_done);
    }
    for (int i = 0; i < swr_context::NFB; i++) {
        if (c->frame_done[i]) hipEventDestroy(c->frame_done[i]);
        for (int j = 0; j < 2; j++) if (c->copy_done[i][j]) hipEventDestroy(c->copy_done[i][j]);
    }
    for (int i = 0; i < 2; i++) {
        for (int j = 0; j < 2; j++) {
            if (c->stage[i][j]) hipHostFree(c->stage[i][j]);
            if (c->stage_ev[i][j]) hipEventDestroy(c->stage_ev[i][j]);
        }
        if (c->copy_stream[i]) hipStreamDestroy(c->copy_stream[i]);
    }
    if (c->bin_stream_own) hipStreamDestroy(c->bin_stream_own);
    if (c->h_pairs) hipHostFree(c->h_pairs);
    if (c->h_misc) hipHostFree(c->h_misc);
    if (c->ev_ok)
        for (int r = 0; r < swr_context::RING; r++)
            for (int i = 0; i < 5; i++) hipEventDestroy(c->ev[r][i]);
    for (auto& e : c->up_ev) if (e) hipEventDestroy(e);
    if (c->up_chunk_ev) hipEventDestroy(c->up_chunk_ev);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

// helpers: 2 = a binning and a raster helper (a single-device context), 1 = the raster helper only — the thread that
// calls swr_draw enqueues the binning itself (sub-contexts of a group: that thread is the group's per-device worker, so
// a device has ONE thread that polls for completions, not two), 0 = none.
int create_single(int dev, swr_context** out, int helpers, uint32_t wait_budget_ms) {
    swr_context* c = new swr_context();
    c->device = dev;
    if (wait_budget_ms) c->wait_budget_ms = wait_budget_ms;
    c->hp_on = getenv("SWR_HOST_PROFILE") && atoi(getenv("SWR_HOST_PROFILE")) == 1;
    hipError_t e;
    if ((e = hipSetDevice(dev)) != hipSuccess || (e = prepare_device()) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->copy_stream[0], hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->copy_stream[1], hipStreamNonBlocking)) != hipSuccess ||
        (e = hipHostMalloc((void**)&c->h_pairs, (2 * swr_context::PAIR_RING + 2) * 4, hipHostMallocMapped)) != hipSuccess ||
        (e = hipHostGetDevicePointer((void**)&c->h_pairs_dev, c->h_pairs, 0)) != hipSuccess ||
        (e = hipHostMalloc((void**)&c->h_misc, CNT_WORDS * 4, hipHostMallocDefault)) != hipSuccess) {
        int rc = fail(nullptr, SWR_ERR_HIP, "context init on device %d failed: %s", dev, hipGetErrorString(e));
        destroy_single(c);
        return rc;
    }
    memset(c->h_pairs, 0, (2 * swr_context::PAIR_RING + 2) * 4);
    c->h_pairs[swr_context::PAIR_RING] = 0xFFFFFFFFu;     // fullest bin: unknown
    memset(c->h_misc, 0, CNT_WORDS * 4);
    {
        // SWR_PIPELINE=0: binning and raster share one stream (no overlap of consecutive frames)
        const char* pl = getenv("SWR_PIPELINE");
        if (pl && pl[0] == '0') c->bin_stream = c->stream;
        else {
            // a plain second stream: stream priorities (binning highest or lowest) were measured to make
            // no difference to how the two queues share the CUs on this platform
            if (hipStreamCreateWithFlags(&c->bin_stream, hipStreamNonBlocking) != hipSuccess) c->bin_stream = c->stream;
            else c->bin_stream_own = c->bin_stream;
            // SWR_HOST_THREADS=1: enqueue everything from the caller's thread (no helper)
            const char* ht = getenv("SWR_HOST_THREADS");
            if (ht && ht[0] == '1') helpers = 0;
            { const char* be = getenv("SWR_BIND_EVENTS"); c->bind_events = !(be && be[0] == '0'); }
            { const char* ii = getenv("SWR_INLINE_IDLE"); c->inline_idle = !(ii && ii[0] == '0'); }
            const char* ew = getenv("SWR_EVENT_WAITS");     // =1: no raster helper; the caller's thread orders the streams by event waits
            if (c->bin_stream_own && helpers >= 2) { c->bin_worker = new Worker(); c->bin_worker->start(dev, true); }
            if (c->bin_stream_own && helpers >= 1 && !(ew && ew[0] == '1')) { c->ras_worker = new Worker(); c->ras_worker->start(dev, true); }
        }
        for (auto& sl : c->slot) {
            hipEventCreateWithFlags(&sl.bin_done, hipEventDisableTiming);
            hipEventCreateWithFlags(&sl.ras_done, hipEventDisableTiming);
        }
        for (int i = 0; i < swr_context::NFB; i++) {
            hipEventCreateWithFlags(&c->frame_done[i], hipEventDisableTiming);
            for (int j = 0; j < 2; j++) hipEventCreateWithFlags(&c->copy_done[i][j], hipEventDisableTiming);
        }
        // frame lanes (swr_context::lane_stream): one stream per working set.  SWR_LANES=0: the two-stream pipeline of rounds 1-3
        {
            const char* ln = getenv("SWR_LANES");
            c->lanes_ok = c->bin_stream_own != nullptr && !(ln && ln[0] == '0');
            for (int l = 0; l < swr_context::NSLOT && c->lanes_ok; l++)
                if (hipStreamCreateWithFlags(&c->lane_stream[l], hipStreamNonBlocking) != hipSuccess) c->lanes_ok = false;
            for (hipEvent_t& e : c->pace_ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
        }
    }
    for (int r = 0; r < swr_context::RING; r++)
        for (int i = 0; i < 5; i++) hipEventCreate(&c->ev[r][i]);
    c->ev_ok = true;
    *out = c;
    return SWR_OK;
}

// ---- group fan-out -----------------------------------------------------------------------------------------------
inline bool is_group(const swr_context* c) { return !c->kids.empty(); }

// run fn(kid) on every sub-context's own thread and wait; the first failure (and its text) becomes the group's
template <class F>
int group_run(swr_context* g, F fn) {
    const size_t n = g->kids.size();
    for (size_t k = 0; k < n; k++) {
        swr_context* kid = g->kids[k];
        g->workers[k]->post([kid, fn] { return fn(kid); });
    }
    int first = SWR_OK;
    for (size_t k = 0; k < n; k++) {
        const int rc = g->workers[k]->drain();
        if (rc && !first) {
            first = rc;
            g->err = "device " + std::to_string(g->kids[k]->device) + " (band " + std::to_string(k) + "): " + g->kids[k]->err;
        }
    }
    return first;
}

// fire and forget (swr_draw / swr_present on a group): errors surface at the next group_run
template <class F>
void group_post(swr_context* g, F fn) {
    for (size_t k = 0; k < g->kids.size(); k++) {
        swr_context* kid = g->kids[k];
        g->workers[k]->post([kid, fn] { return fn(kid); });
    }
}

// tile-row band k of n inside [row_begin, row_end) (row_begin is tile-aligned)
void sub_band(int64_t row_begin, int64_t row_end, int n, int k, int64_t* r0, int64_t* r1) {
    const int64_t trows = (row_end - row_begin + TILE_H - 1) / TILE_H;
    *r0 = std::min<int64_t>(row_begin + trows * k / n * TILE_H, row_end);
    *r1 = std::min<int64_t>(row_begin + trows * (k + 1) / n * TILE_H, row_end);
}

}  // namespace

extern "C" {

int swr_abi_version(void) { return SWR_ABI_VERSION; }
const char* swr_version(void) { return "swr-hip gfx950 0.4 (tile 64x32, wave64 LDS visibility keys: 32-bit depth keys + winner table, span ring, one-launch binning, frame lanes, multi-device bands)"; }
int swr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
int swr_tile_rows(void) { return TILE_H; }
int swr_tile_cols(void) { return TILE_W; }

const char* swr_last_error(const swr_context* ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int swr_band_rows(int64_t height, int32_t parts, int32_t part, int64_t* row_begin, int64_t* row_end) {
    if (height <= 0 || parts <= 0 || part < 0 || part >= parts || !row_begin || !row_end)
        return SWR_ERR_BAD_ARG;
    sub_band(0, height, parts, part, row_begin, row_end);
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
    const uint32_t n = cfg ? cfg->device_count : 0;
    if (n > 64) return fail(nullptr, SWR_ERR_BAD_ARG, "device_count %u > 64", n);
    const uint32_t budget = cfg ? cfg->wait_budget_ms : 0u;
    if (n <= 1) return create_single(dev, out, 2, budget);
    // group: sub-context k on device (dev + k) % min(n, visible devices) — with fewer GPUs than bands several bands
    // share a GPU (same code path; how a 1-GPU box tests the 8-band layout)
    swr_context* g = new swr_context();
    g->device = dev;
    if (budget) g->wait_budget_ms = budget;
    const int span = std::min<int>((int)n, ndev);
    for (uint32_t k = 0; k < n; k++) {
        swr_context* kid = nullptr;
        int rc = create_single((dev + (int)(k % (uint32_t)span)) % ndev, &kid, 1, budget);
        if (rc) {
            for (swr_context* x : g->kids) destroy_single(x);
            delete g;
            return rc;
        }
        g->kids.push_back(kid);
    }
    for (uint32_t k = 0; k < n; k++) {
        Worker* w = new Worker();
        w->start(g->kids[k]->device);
        g->workers.push_back(w);
    }
    *out = g;
    return SWR_OK;
}

void swr_context_destroy(swr_context* c) {
    if (!c) return;
    if (is_group(c)) {
        for (Worker* w : c->workers) { w->drain(); w->stop(); delete w; }
        for (swr_context* kid : c->kids) destroy_single(kid);
        delete c;
        return;
    }
    destroy_single(c);
}

int swr_context_bands(const swr_context* c) { return c ? (is_group(c) ? (int)c->kids.size() : 1) : 0; }

int swr_context_band_info(const swr_context* c, int32_t band, int32_t* device, int64_t* row_begin, int64_t* row_end) {
    if (!c || band < 0 || band >= swr_context_bands(c)) return SWR_ERR_BAD_ARG;
    const swr_context* k = is_group(c) ? c->kids[band] : c;
    if (device) *device = k->device;
    if (row_begin) *row_begin = k->has_target ? k->tg.row_begin : 0;
    if (row_end) *row_end = k->has_target ? k->tg.row_end : 0;
    return SWR_OK;
}

int swr_scene_upload(swr_context* c, const swr_vertex* vertices, int64_t vertex_count,
                     const int64_t* indices, int64_t index_count) {
    if (!c) return SWR_ERR_BAD_ARG;
    c->scene_id = 0;      // the resident scene changes: a later swr_render must not take it for its own
    if (is_group(c))   // replicated: every device reads the caller's arrays itself, over its own PCIe link
        return group_run(c, [=](swr_context* k) { return single_scene_upload(k, vertices, vertex_count, indices, index_count); });
    return single_scene_upload(c, vertices, vertex_count, indices, index_count);
}

int swr_scene_attributes(swr_context* c, const swr_vertex_attr* attributes, int64_t vertex_count) {
    if (!c) return SWR_ERR_BAD_ARG;
    c->scene_id = 0;      // the resident scene changes: a later swr_render must not take it for its own
    if (is_group(c)) return group_run(c, [=](swr_context* k) { return single_scene_attributes(k, attributes, vertex_count); });
    return single_scene_attributes(c, attributes, vertex_count);
}

int swr_material_set(swr_context* c, const swr_material* m) {
    if (!c) return SWR_ERR_BAD_ARG;
    if (is_group(c)) {
        const bool has = m != nullptr;
        const swr_material copy = has ? *m : swr_material{};
        return group_run(c, [=](swr_context* k) { return single_material_set(k, has ? &copy : nullptr); });
    }
    return single_material_set(c, m);
}

int swr_texture_upload(swr_context* c, const void* bgra8, int32_t width, int32_t height) {
    if (!c) return SWR_ERR_BAD_ARG;
    c->scene_id = 0;      // the resident scene changes: a later swr_render must not take it for its own
    if (is_group(c)) return group_run(c, [=](swr_context* k) { return single_texture_upload(k, bgra8, width, height); });
    return single_texture_upload(c, bgra8, width, height);
}

int swr_target_set(swr_context* c, int64_t width, int64_t height, int64_t row_begin, int64_t row_end) {
    if (!c) return SWR_ERR_BAD_ARG;
    if (!is_group(c)) return single_target_set(c, width, height, row_begin, row_end);
    int rc = check_target_args(c, width, height, row_begin, row_end);
    if (rc) return rc;
    const int n = (int)c->kids.size();
    for (int k = 0; k < n; k++) {
        int64_t r0, r1;
        sub_band(row_begin, row_end, n, k, &r0, &r1);
        swr_context* kid = c->kids[k];
        c->workers[k]->post([=] { return single_target_set(kid, width, height, r0, r1); });
    }
    rc = group_run(c, [](swr_context*) { return SWR_OK; });
    if (rc) return rc;
    c->group_tg = Target{(int32_t)width, (int32_t)height, (int32_t)row_begin, (int32_t)row_end, 0, 0};
    c->group_has_target = true;
    return SWR_OK;
}

int swr_draw_primitives(swr_context* c, const float transform[16], uint32_t flags, int32_t primitive_type) {
    if (!c || !transform) return SWR_ERR_BAD_ARG;
    if (!is_group(c)) return single_draw(c, transform, flags, primitive_type);
    int rc = check_draw_args(c, flags, primitive_type);
    if (rc) return rc;
    std::array<float, 16> m;
    memcpy(m.data(), transform, sizeof(float) * 16);
    group_post(c, [=](swr_context* k) { return single_draw(k, m.data(), flags, primitive_type); });
    return SWR_OK;      // asynchronous, like the single-device draw: a failure is reported by the next blocking call
}

int swr_draw(swr_context* c, const float transform[16], uint32_t flags) {
    return swr_draw_primitives(c, transform, flags, SWR_PRIMITIVE_TRIANGLE);
}

int swr_sync(swr_context* c) {
    if (!c) return SWR_ERR_BAD_ARG;
    if (is_group(c)) return group_run(c, [](swr_context* k) { return single_sync(k); });
    return single_sync(c);
}

int swr_present(swr_context* c, void* color_full, float* depth_full) {
    if (!c) return SWR_ERR_BAD_ARG;
    if (!is_group(c)) return single_present(c, color_full, depth_full);
    if (!color_full && !depth_full) return fail(c, SWR_ERR_BAD_ARG, "swr_present: both image pointers are NULL");
    group_post(c, [=](swr_context* k) { return single_present(k, color_full, depth_full); });
    return SWR_OK;
}

int swr_present_wait(swr_context* c) {
    if (!c) return SWR_ERR_BAD_ARG;
    if (is_group(c)) return group_run(c, [](swr_context* k) { return single_present_wait(k); });
    return single_present_wait(c);
}

int swr_read_color(swr_context* c, void* dst) {
    if (!c || !dst) return SWR_ERR_BAD_ARG;
    if (is_group(c)) return group_run(c, [=](swr_context* k) { return single_read(k, 0, dst); });
    return single_read(c, 0, dst);
}

int swr_read_depth(swr_context* c, float* dst) {
    if (!c || !dst) return SWR_ERR_BAD_ARG;
    if (is_group(c)) return group_run(c, [=](swr_context* k) { return single_read(k, 1, dst); });
    return single_read(c, 1, dst);
}

// ---- page-locked host images ---------------------------------------------------------------------------------------
void* swr_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void swr_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int swr_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return SWR_ERR_BAD_ARG;
    if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, SWR_ERR_HIP, "hipHostRegister(%p, %zu) failed", p, bytes);
    }
    return SWR_OK;
}

int swr_host_unregister(void* p) {
    if (!p) return SWR_ERR_BAD_ARG;
    if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return SWR_ERR_HIP; }
    return SWR_OK;
}

// ---- instrumentation -------------------------------------------------------------------------------------------------
int swr_timing_enable(swr_context* c, int enable) {
    if (!c) return SWR_ERR_BAD_ARG;
    if (is_group(c)) return group_run(c, [=](swr_context* k) { return swr_timing_enable(k, enable); });
    int rc = swr_sync(c);
    if (rc) return rc;
    c->timing = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    return SWR_OK;
}

int swr_timing_sample(swr_context* c, int every_nth) {
    if (!c || every_nth < 1) return SWR_ERR_BAD_ARG;
    if (is_group(c)) return group_run(c, [=](swr_context* k) { return swr_timing_sample(k, every_nth); });
    int rc = swr_sync(c);
    if (rc) return rc;
    c->timing_every = every_nth;
    return SWR_OK;
}

int swr_pipeline_enable(swr_context* c, int enable) {
    if (!c) return SWR_ERR_BAD_ARG;
    if (is_group(c)) return group_run(c, [=](swr_context* k) { return swr_pipeline_enable(k, enable); });
    int rc = swr_sync(c);
    if (rc) return rc;
    c->bin_stream = (enable && c->bin_stream_own) ? c->bin_stream_own : c->stream;
    return SWR_OK;
}

// A group reports the slowest band per stage (the frame is done when the last band is) and the sums of the counts.
static void merge_timings(swr_timings* acc, const swr_timings& t, bool first) {
    if (first) { *acc = t; return; }
    acc->setup_bin_ms = std::max(acc->setup_bin_ms, t.setup_bin_ms);
    acc->scan_ms = std::max(acc->scan_ms, t.scan_ms);
    acc->scatter_ms = std::max(acc->scatter_ms, t.scatter_ms);
    acc->raster_ms = std::max(acc->raster_ms, t.raster_ms);
    acc->total_ms = std::max(acc->total_ms, t.total_ms);
    acc->tile_pairs += t.tile_pairs;
    acc->tiles += t.tiles;
}

int swr_timing_totals(swr_context* c, swr_timings* sum, int64_t* frames) {
    if (!c || !sum || !frames) return SWR_ERR_BAD_ARG;
    if (is_group(c)) {
        int rc = swr_sync(c);
        if (rc) return rc;
        for (size_t k = 0; k < c->kids.size(); k++) {
            swr_timings t; int64_t f = 0;
            if ((rc = swr_timing_totals(c->kids[k], &t, &f))) return rc;
            merge_timings(sum, t, k == 0);
            if (k == 0) *frames = f;
        }
        return SWR_OK;
    }
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
    if (is_group(c)) return group_run(c, [](swr_context* k) { return swr_timing_reset(k); });
    int rc = swr_sync(c);
    if (rc) return rc;
    for (double& v : c->sum_ms) v = 0.0;
    c->sum_frames = 0;
    return SWR_OK;
}

int swr_get_timings(swr_context* c, swr_timings* out) {
    if (!c || !out) return SWR_ERR_BAD_ARG;
    if (is_group(c)) {
        int rc = swr_sync(c);
        if (rc) return rc;
        for (size_t k = 0; k < c->kids.size(); k++) {
            swr_timings t;
            if ((rc = swr_get_timings(c->kids[k], &t))) return rc;
            merge_timings(out, t, k == 0);
        }
        return SWR_OK;
    }
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
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<float, std::milli>(b - a).count(); };
    const auto t0 = clk::now();
    int rc;
    // Scene identity: the caller's promise that the arrays hold what they held at the last call with this id — the
    // resident copy (and the triangle stream built from it) is then used as it is, like the reference's GpuRenderer
    // keeps its MTLBuffers across calls (GpuRenderer.swift:32-33,41-67) for the app's one mesh (App.swift:153-185).
    const bool cached = p->scene_id != 0 && p->scene_id == c->scene_id && p->vertex_count == c->scene_nv &&
                        p->index_count == c->scene_ni && (p->attributes != nullptr) == c->scene_attrs &&
                        (p->texture == nullptr ? c->scene_tex == nullptr
                                               : (c->scene_tex != nullptr && p->tex_width == c->scene_tw && p->tex_height == c->scene_th));
    c->rt = swr_render_times{};
    c->rt.scene_cached = cached ? 1 : 0;
    if (!cached) {
        c->scene_id = 0;
        // no identity = the scene lives for this frame only: uploaded in index order, built behind the copy
        const bool oneshot = p->scene_id == 0;
        if (is_group(c))
            rc = group_run(c, [=](swr_context* k) { return single_scene_upload(k, p->vertices, p->vertex_count, p->indices, p->index_count, oneshot); });
        else
            rc = single_scene_upload(c, p->vertices, p->vertex_count, p->indices, p->index_count, oneshot);
        if (rc) return rc;
        if (p->attributes && (rc = swr_scene_attributes(c, p->attributes, p->vertex_count))) return rc;
        if (p->texture && (rc = swr_texture_upload(c, p->texture, p->tex_width, p->tex_height))) return rc;
        c->scene_id = p->scene_id; c->scene_nv = p->vertex_count; c->scene_ni = p->index_count;
        c->scene_attrs = p->attributes != nullptr;
        c->scene_tex = p->texture; c->scene_tw = p->tex_width; c->scene_th = p->tex_height;
        // the split inside the upload: HIP events of (the slowest band of) the upload itself
        const swr_context* k0 = c->kids.empty() ? c : c->kids[0];
        float h2d = k0->up_h2d_ms, build = k0->up_build_ms;
        for (const swr_context* k : c->kids) { h2d = std::max(h2d, k->up_h2d_ms); build = std::max(build, k->up_build_ms); }
        c->rt.h2d_ms = h2d;
        c->rt.stream_build_ms = ms(t0, clk::now()) - h2d;       // index check, sort, gather, attribute / texture passes, allocation
        (void)build;
    }
    const auto t1 = clk::now();
    // the pass carries its own fragment stage: NULL material = the reference's passthrough
    if ((rc = swr_material_set(c, p->material))) return rc;
    if ((rc = swr_target_set(c, p->width, p->height, 0, p->height))) return rc;
    const swr_context* kf = c->kids.empty() ? c : c->kids[0];
    const uint64_t frames0 = kf->frame_no;
    if ((rc = swr_draw_primitives(c, p->transform, p->flags, p->primitive_type))) return rc;
    // colour and depth leave every device together (two copy streams each); synchronous on return like
    // scheduleAndWait (Metal+Extensions.swift:57-67)
    if ((rc = swr_present(c, (p->flags & SWR_FLAG_NO_COLOR) ? nullptr : p->color, p->depth))) return rc;
    if ((rc = swr_sync(c))) return rc;                           // (the frame itself; repairs an overflow before the copy is waited for)
    const auto t2 = clk::now();
    if ((rc = swr_present_wait(c))) return rc;
    const auto t3 = clk::now();
    c->rt.draw_ms = ms(t1, t2);
    c->rt.frames = (int32_t)(kf->frame_no - frames0);            // 1, or 2 when the bins had to grow and the frame was redrawn
    c->rt.gather_ms = ms(t2, t3);
    c->rt.total_ms = ms(t0, t3);
    return SWR_OK;
}

int swr_render_timings(swr_context* c, swr_render_times* out) {
    if (!c || !out) return SWR_ERR_BAD_ARG;
    *out = c->rt;
    return SWR_OK;
}

int swr_debug_set(swr_context* c, int key, int64_t value) {
    if (!c) return SWR_ERR_BAD_ARG;
    if (is_group(c)) {
        int rc = SWR_OK;
        for (swr_context* k : c->kids) { const int r = swr_debug_set(k, key, value); if (r && !rc) { rc = r; c->err = k->err; } }
        return rc;
    }
    switch (key) {
        case SWR_DEBUG_STREAM_ORDER:
            if (value < -1 || value > 1) break;
            c->dbg_stream_order = (int)value; return SWR_OK;
        case SWR_DEBUG_CULL:
            if (value < 0 || value > 1) break;
            c->dbg_cull = (int)value; return SWR_OK;
        case SWR_DEBUG_BIN_MODE: {
            if (value < 0 || value > 3) break;
            c->dbg_bin_mode = (int)value;
            // the layout of the bins follows the mode: size them again for what is resident (streams idle first)
            if (c->has_scene && c->has_target) {
                int rc = swr_sync(c);
                if (rc) return rc;
                c->fixed_allowed = true; c->fixed_mode = false;
                return size_bins(c);
            }
            return SWR_OK;
        }
        case SWR_DEBUG_ONESHOT_MIN_TRIS:
            if (value < 0) break;
            c->dbg_oneshot_min_tris = std::max<int64_t>(64, value); return SWR_OK;
        case SWR_DEBUG_DEPTH_KEYS32:
            if (value < 0 || value > 1) break;
            c->dbg_k32 = value != 0; return SWR_OK;
        case SWR_DEBUG_RASTER_SORT:
            if (value < 0 || value > 2) break;
            c->dbg_insort = (int)value; return SWR_OK;
        default:
            return fail(c, SWR_ERR_BAD_ARG, "swr_debug_set: unknown key %d", key);
    }
    return fail(c, SWR_ERR_BAD_ARG, "swr_debug_set: value %lld out of range for key %d", (long long)value, key);
}

int swr_debug_fault(swr_context* c, int fault) {
    if (!c || fault < SWR_FAULT_NONE || fault > SWR_FAULT_ENQUEUE) return SWR_ERR_BAD_ARG;
    if (is_group(c)) { for (swr_context* k : c->kids) k->inject.store(fault); return SWR_OK; }
    c->inject.store(fault);
    return SWR_OK;
}

}  // extern "C"
