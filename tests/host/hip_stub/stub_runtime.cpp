// The fake HIP runtime (see hip/hip_runtime.h in this directory): in-order streams on host threads.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <set>
#include <thread>

struct fake_event {
    std::atomic<uint64_t> recorded{0}, completed{0};      // generation counters: complete when completed >= recorded
    std::chrono::steady_clock::time_point when;
    std::mutex m;
};
struct fake_stream {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    std::atomic<uint64_t> enq{0}, done{0};
    bool quit = false;
    std::thread th;
    fake_stream() {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                cv.wait(lk, [this] { return quit || !q.empty(); });
                if (q.empty() && quit) return;
                auto op = std::move(q.front());
                q.pop_front();
                lk.unlock();
                op();
                done.fetch_add(1, std::memory_order_release);
                lk.lock();
            }
        });
    }
    void push(std::function<void()> op) {
        { std::lock_guard<std::mutex> lk(m); q.push_back(std::move(op)); enq.fetch_add(1, std::memory_order_relaxed); }
        cv.notify_one();
    }
    ~fake_stream() {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv.notify_one();
        th.join();
    }
};
static std::atomic<int> g_delay_us{20};
static std::mutex g_pin_m;
static std::set<const void*> g_pinned;
static thread_local hipError_t tl_last = hipSuccess;

void fake_kernel_delay_us(int us) { g_delay_us.store(us); }
static void complete(fake_event* e, uint64_t gen) {
    { std::lock_guard<std::mutex> lk(e->m); e->when = std::chrono::steady_clock::now(); }
    uint64_t c = e->completed.load(std::memory_order_relaxed);
    while (c < gen && !e->completed.compare_exchange_weak(c, gen, std::memory_order_release)) {}
}
void fake_enqueue(hipStream_t s, std::function<void()> op, hipEvent_t stop) {
    uint64_t gen = 0;
    if (stop) gen = stop->recorded.fetch_add(1, std::memory_order_acq_rel) + 1;
    s->push([op, stop, gen] {
        const int us = g_delay_us.load();
        if (us > 0) std::this_thread::sleep_for(std::chrono::microseconds(us));
        if (op) op();
        if (stop) complete(stop, gen);
    });
}

hipError_t hipGetDeviceCount(int* n) { *n = getenv("FAKE_HIP_DEVICES") ? atoi(getenv("FAKE_HIP_DEVICES")) : 2; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetLastError() { const hipError_t e = tl_last; tl_last = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : (e == hipErrorNotReady ? "hipErrorNotReady" : "fake HIP error"); }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
    *p = calloc(1, n ? n : 1);
    std::lock_guard<std::mutex> lk(g_pin_m); g_pinned.insert(*p);
    return hipSuccess;
}
hipError_t hipHostFree(void* p) { { std::lock_guard<std::mutex> lk(g_pin_m); g_pinned.erase(p); } free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipHostRegister(void* p, size_t, unsigned) { std::lock_guard<std::mutex> lk(g_pin_m); g_pinned.insert(p); return hipSuccess; }
hipError_t hipHostUnregister(void* p) { std::lock_guard<std::mutex> lk(g_pin_m); g_pinned.erase(p); return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
    std::lock_guard<std::mutex> lk(g_pin_m);
    if (g_pinned.count(p)) { a->type = hipMemoryTypeHost; return hipSuccess; }
    tl_last = hipErrorInvalidValue;
    return hipErrorInvalidValue;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new fake_stream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) {
    if (s->done.load(std::memory_order_acquire) >= s->enq.load(std::memory_order_acquire)) return hipSuccess;
    tl_last = hipErrorNotReady;
    return hipErrorNotReady;
}
hipError_t hipStreamSynchronize(hipStream_t s) { while (hipStreamQuery(s) != hipSuccess) std::this_thread::yield(); tl_last = hipSuccess; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new fake_event(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new fake_event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    const uint64_t gen = e->recorded.fetch_add(1, std::memory_order_acq_rel) + 1;
    s->push([e, gen] { complete(e, gen); });
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) {
    if (e->completed.load(std::memory_order_acquire) >= e->recorded.load(std::memory_order_acquire)) return hipSuccess;   // never recorded: complete
    tl_last = hipErrorNotReady;
    return hipErrorNotReady;
}
hipError_t hipEventSynchronize(hipEvent_t e) { while (hipEventQuery(e) != hipSuccess) std::this_thread::yield(); tl_last = hipSuccess; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    const uint64_t gen = e->recorded.load(std::memory_order_acquire);
    s->push([e, gen] { while (e->completed.load(std::memory_order_acquire) < gen) std::this_thread::yield(); });
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    std::lock_guard<std::mutex> la(a->m);
    std::chrono::steady_clock::time_point ta = a->when;
    std::chrono::steady_clock::time_point tb;
    if (a == b) tb = ta; else { std::lock_guard<std::mutex> lb(b->m); tb = b->when; }
    *ms = std::chrono::duration<float, std::milli>(tb - ta).count();
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t s) {
    s->push([dst, src, n] { memcpy(dst, src, n); });
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s) {
    s->push([dst, v, n] { memset(dst, v, n); });
    return hipSuccess;
}
