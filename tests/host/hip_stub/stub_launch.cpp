// Stand-ins for the kernel launch wrappers of swr_kernels.hip / swr_upload.hip on the fake HIP runtime: they enqueue host
// lambdas that leave what the host layer looks at afterwards (pair totals, largest fill, a frame tag in the framebuffer).
// Test infrastructure of tests/host/tsan_host_test.cpp only.
#include <algorithm>
#include <atomic>
#include <cstring>

#include "../../../software-renderer_amd/csrc/swr_internal.h"

namespace swr {
std::atomic<uint32_t> g_fake_fill{7};        // largest tile fill the fake k_bin reports (tests raise it to force a regrow)
std::atomic<uint32_t> g_fake_pairs{1000};

int live_groups_per_workgroup(int64_t ntri, int G) { return (int)(((ntri + 63) / 64 + G - 1) / (G > 0 ? G : 1)); }
BinPlan plan_binning(int64_t ntri, int ntiles, bool) {
    BinPlan p{};
    p.use_lds = true; p.threads = 256; p.G = (int)std::max<int64_t>(1, std::min<int64_t>(256, (ntri + 255) / 256));
    p.chunk = (int)((ntri + p.G - 1) / p.G); p.lds_bytes = (size_t)ntiles * 4;
    return p;
}
uint32_t fixed_cap_max(int64_t ntri, int ntiles) { return (ntri <= 0 || ntiles <= 0) ? 0u : 61440u; }
hipError_t prepare_device() { return hipSuccess; }
size_t stream_sort_temp_bytes(int64_t) { return 64; }

void launch_validate_indices(const int64_t* idx, int64_t n, int64_t nv, uint32_t* counters, hipStream_t s) {
    fake_enqueue(s, [=] { for (int64_t i = 0; i < n; i++) if (idx[i] < 0 || idx[i] >= nv) counters[CNT_BAD_INDEX] = 1u; });
}
hipError_t launch_build_stream(const StreamBuild&, hipStream_t s) { fake_enqueue(s, nullptr); return hipSuccess; }
hipError_t launch_build_stream_range(const StreamBuild&, int64_t, int64_t, hipStream_t s) { fake_enqueue(s, nullptr); return hipSuccess; }
void launch_gather_attrs(const swr_vertex_attr*, int64_t, const int64_t*, int64_t, const float4*, float4*, float4*, hipStream_t s) { fake_enqueue(s, nullptr); }
void launch_texture_to_float(const uint32_t*, int64_t, float4*, hipStream_t s) { fake_enqueue(s, nullptr); }

bool launch_bin(const DeviceFrame& f, hipStream_t s, hipEvent_t stop) {
    const DeviceFrame ff = f;
    fake_enqueue(s, [ff] {
        const int ntiles = ff.tg.tiles_x * ff.tg.tiles_y;
        memset(ff.fill_next, 0, (size_t)(CNT_WORDS + ntiles) * 4);
        ff.fill[CNT_PAIRS] = g_fake_pairs.load();
        ff.fill[3] = g_fake_fill.load();                    // CNT_MAXFILL
    }, stop);
    return stop != nullptr;
}
void launch_setup_bin(const DeviceFrame&, hipStream_t s) { fake_enqueue(s, nullptr); }
void launch_scan(const DeviceFrame&, hipStream_t) {}
bool launch_fill(const DeviceFrame& f, hipStream_t s, hipEvent_t stop) {
    const DeviceFrame ff = f;
    fake_enqueue(s, [ff] { ff.counters[CNT_PAIRS] = g_fake_pairs.load(); *ff.host_counters = g_fake_pairs.load(); if (ff.host_max) __atomic_store_n(ff.host_max, g_fake_fill.load(), __ATOMIC_RELAXED); }, stop);
    return stop != nullptr;
}
bool launch_sort_bins(const DeviceFrame& f, hipStream_t s, hipEvent_t stop) {
    if (f.skip_sort) return false;
    fake_enqueue(s, nullptr, stop);
    return stop != nullptr;
}
bool frame_uses_k32(const DeviceFrame& f) {
    return f.k32 && (f.flags & SWR_FLAG_DEPTH_TEST) && (f.flags & SWR_FLAG_NO_COLOR) && !(f.flags & SWR_FLAG_METAL_RULES);
}
bool launch_raster(const DeviceFrame& f, hipStream_t s, hipEvent_t stop) {
    const DeviceFrame ff = f;
    if (ff.tg.tiles_x * ff.tg.tiles_y == 0) return false;
    fake_enqueue(s, [ff] {
        bool overflow;
        if (ff.fixed_bins) {
            *ff.host_counters = ff.fill[CNT_PAIRS]; *ff.host_fill = ff.fill[3]; if (ff.host_max) __atomic_store_n(ff.host_max, ff.fill[3], __ATOMIC_RELAXED);
            overflow = ff.fill[3] > ff.cap_tile;
        } else overflow = ff.counters[CNT_PAIRS] > ff.capacity;
        // the "image": every pixel of the band carries the frame's tag (transform[0]); an overflowed frame is rastered empty
        const size_t n = (size_t)ff.tg.width * (size_t)(ff.tg.row_end - ff.tg.row_begin);
        const float tag = overflow ? -1.0f : ff.m[0];
        for (size_t i = 0; i < n; i++) ff.depth[i] = tag;
        if (ff.color && !(ff.flags & SWR_FLAG_NO_COLOR)) memset(ff.color, (int)tag & 0xFF, n * 4);
    }, stop);
    return stop != nullptr;
}
void launch_points_or_lines(const DeviceFrame& f, int, hipStream_t s) {
    const DeviceFrame ff = f;
    fake_enqueue(s, [ff] {
        const size_t n = (size_t)ff.tg.width * (size_t)(ff.tg.row_end - ff.tg.row_begin);
        for (size_t i = 0; i < n; i++) ff.depth[i] = ff.m[0];
    });
}
}  // namespace swr
