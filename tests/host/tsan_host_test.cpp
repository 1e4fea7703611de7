// Thread-sanitizer run of the C-ABI's HOST layer (software-renderer_amd/csrc/swr_api.hip: helper threads, host-paced
// stream ordering, the present / regrow / failure protocols) on a fake HIP runtime (tests/host/hip_stub/): streams are
// host threads, kernels are stand-ins that tag the framebuffer with the frame's transform[0].  No GPU, no pixels: what is
// checked is ORDER (every host image shows the frame it was presented for), the error protocol, and that TSan stays silent.
//   g++ -std=c++17 -O1 -g -fsanitize=thread -Itests/host/hip_stub -x c++ software-renderer_amd/csrc/swr_api.hip \
//       tests/host/hip_stub/stub_runtime.cpp tests/host/hip_stub/stub_launch.cpp tests/host/tsan_host_test.cpp -lpthread
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/swr.h"
#include <hip/hip_runtime.h>

namespace swr { extern std::atomic<uint32_t> g_fake_fill, g_fake_pairs; }
static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

static const int W = 128, H = 96;
static std::vector<swr_vertex> verts(300);
static std::vector<int64_t> idx(300);
static void tagm(float m[16], float tag) { memset(m, 0, 64); m[0] = tag; m[5] = m[10] = m[15] = 1.0f; }
static bool all_eq(const float* d, float v, int r0 = 0, int r1 = H) {
    for (int i = r0 * W; i < r1 * W; i++) if (d[i] != v) return false;
    return true;
}

static void resident(uint32_t devices) {
    swr_config cfg{0, devices, 2000, 0};
    swr_context* c = nullptr;
    CHECK(swr_context_create(&cfg, &c) == SWR_OK);
    CHECK(swr_scene_upload(c, verts.data(), 300, idx.data(), 300) == SWR_OK);
    CHECK(swr_target_set(c, W, H, 0, H) == SWR_OK);
    float m[16];
    std::vector<float> d(W * H);
    // a long un-waited burst (crosses the working-set ring and the 64-deep raster ring), then one read
    for (int f = 1; f <= 300; f++) { tagm(m, (float)f); CHECK(swr_draw(c, m, SWR_FLAG_DEPTH_TEST | SWR_FLAG_NO_COLOR) == SWR_OK); }
    CHECK(swr_read_depth(c, d.data()) == SWR_OK);
    CHECK(all_eq(d.data(), 300.0f));
    // draw + present into four page-locked images and a pageable one, no wait in between
    float* img[5];
    for (int k = 0; k < 4; k++) img[k] = (float*)swr_host_alloc(W * H * 4);
    std::vector<float> pageable(W * H);
    img[4] = pageable.data();
    for (int round = 0; round < 6; round++) {
        for (int k = 0; k < 5; k++) {
            tagm(m, (float)(1000 + 10 * round + k));
            CHECK(swr_draw(c, m, SWR_FLAG_DEPTH_TEST | SWR_FLAG_NO_COLOR) == SWR_OK);
            CHECK(swr_present(c, nullptr, img[k]) == SWR_OK);
        }
        CHECK(swr_present_wait(c) == SWR_OK);
        for (int k = 0; k < 5; k++) CHECK(all_eq(img[k], (float)(1000 + 10 * round + k)));
    }
    // one frame at a time (the idle-inline path), timing on and off, pipelining off and on
    for (int f = 0; f < 20; f++) {
        if (f == 5) CHECK(swr_timing_enable(c, 2) == SWR_OK);
        if (f == 10) CHECK(swr_pipeline_enable(c, 0) == SWR_OK);
        if (f == 15) { CHECK(swr_pipeline_enable(c, 1) == SWR_OK); CHECK(swr_timing_enable(c, 0) == SWR_OK); }
        tagm(m, (float)(2000 + f));
        CHECK(swr_draw(c, m, SWR_FLAG_DEPTH_TEST | SWR_FLAG_NO_COLOR) == SWR_OK);
        CHECK(swr_sync(c) == SWR_OK);
    }
    CHECK(swr_read_depth(c, d.data()) == SWR_OK && all_eq(d.data(), 2019.0f));
    // a tile region overflows in the middle of a presented burst: the earlier frame is reported, the last one repaired
    swr::g_fake_fill.store(5000);
    tagm(m, 3001.0f); CHECK(swr_draw(c, m, SWR_FLAG_DEPTH_TEST | SWR_FLAG_NO_COLOR) == SWR_OK); CHECK(swr_present(c, nullptr, img[0]) == SWR_OK);
    tagm(m, 3002.0f); CHECK(swr_draw(c, m, SWR_FLAG_DEPTH_TEST | SWR_FLAG_NO_COLOR) == SWR_OK); CHECK(swr_present(c, nullptr, img[1]) == SWR_OK);
    CHECK(swr_present_wait(c) == SWR_ERR_FRAME_DROPPED);
    CHECK(swr_present_wait(c) == SWR_OK);
    CHECK(all_eq(img[1], 3002.0f));
    swr::g_fake_fill.store(7);
    // swr_render with a scene identity
    swr_render_pass rp{};
    rp.depth = img[2]; rp.width = W; rp.height = H; rp.vertices = verts.data(); rp.vertex_count = 300; rp.indices = idx.data(); rp.index_count = 300;
    rp.flags = SWR_FLAG_DEPTH_TEST | SWR_FLAG_NO_COLOR; rp.scene_id = 42;
    swr_render_times rt{};
    for (int f = 0; f < 4; f++) {
        tagm(rp.transform, (float)(4000 + f));
        CHECK(swr_render(c, &rp) == SWR_OK);
        CHECK(swr_render_timings(c, &rt) == SWR_OK && rt.scene_cached == (f > 0));
        CHECK(all_eq(img[2], (float)(4000 + f)));
    }
    // ... and without one: the one-shot upload builds the stream on the binning stream, chunk by chunk behind the index copy
    setenv("SWR_ONESHOT_MIN_TRIS", "64", 1);
    rp.scene_id = 0; rp.index_count = 300;
    for (int f = 0; f < 3; f++) {
        tagm(rp.transform, (float)(5000 + f));
        CHECK(swr_render(c, &rp) == SWR_OK);
        CHECK(swr_render_timings(c, &rt) == SWR_OK && rt.scene_cached == 0);
        CHECK(all_eq(img[2], (float)(5000 + f)));
    }
    unsetenv("SWR_ONESHOT_MIN_TRIS");
    // destroy with work in flight
    for (int f = 0; f < 40; f++) { tagm(m, 1.0f); swr_draw(c, m, SWR_FLAG_DEPTH_TEST | SWR_FLAG_NO_COLOR); if (f % 3 == 0) swr_present(c, nullptr, img[3]); }
    swr_context_destroy(c);
    for (int k = 0; k < 4; k++) swr_host_free(img[k]);
}

static void failure(uint32_t devices, int fault) {
    swr_config cfg{0, devices, 150, 0};
    swr_context* c = nullptr;
    CHECK(swr_context_create(&cfg, &c) == SWR_OK);
    CHECK(swr_scene_upload(c, verts.data(), 300, idx.data(), 300) == SWR_OK);
    CHECK(swr_target_set(c, W, H, 0, H) == SWR_OK);
    float m[16];
    tagm(m, 1.0f);
    for (int f = 0; f < 6; f++) CHECK(swr_draw(c, m, SWR_FLAG_DEPTH_TEST | SWR_FLAG_NO_COLOR) == SWR_OK);
    CHECK(swr_sync(c) == SWR_OK);
    CHECK(swr_debug_fault(c, fault) == SWR_OK);
    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < 80; f++) swr_draw(c, m, SWR_FLAG_DEPTH_TEST | SWR_FLAG_NO_COLOR);      // beyond the 64-deep raster ring: must not hang
    CHECK(swr_sync(c) == SWR_ERR_HIP);
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    CHECK(s < 8.0);
    CHECK(strlen(swr_last_error(c)) > 10);
    CHECK(swr_sync(c) == SWR_ERR_HIP);                       // sticky
    std::vector<float> d(W * H);
    CHECK(swr_read_depth(c, d.data()) == SWR_ERR_HIP);
    swr_context_destroy(c);                                  // returns
}

int main() {
    for (int i = 0; i < 300; i++) { idx[i] = i; verts[i] = swr_vertex{{0.1f * (float)(i % 7), 0.2f, 0.5f, 0}, {1, 1, 1, 0}}; }
    for (const char* lanes : {"1", "0"}) {       // frame lanes (the default) and the two-stream pipeline with its helper threads
        setenv("SWR_LANES", lanes, 1);
        for (int delay : {0, 15, 150}) {         // kernels that finish before / while / long after the host enqueues the next
            fake_kernel_delay_us(delay);
            resident(0);
            resident(3);
        }
        fake_kernel_delay_us(15);
        for (int fault = 1; fault <= 2; fault++) { failure(0, fault); failure(2, fault); }
    }
    std::printf(fails ? "tsan host test: %d failures\n" : "tsan host test: ok\n", fails);
    return fails ? 1 : 0;
}
