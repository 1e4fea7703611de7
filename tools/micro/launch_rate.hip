// Host cost of one kernel launch on this runtime: a near-empty kernel with a kernarg block the size of the raster's, launched
// back to back (the GPU keeps up: the kernel takes ~2 us) on one stream and round-robin on four.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/launch_rate tools/micro/launch_rate.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char bytes[320]; int* out; };
__global__ void k_null(Big b) { if (b.out && threadIdx.x == 1000) *b.out = b.bytes[0]; }
__global__ void k_small(int* out) { if (out && threadIdx.x == 1000) *out = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s[4];
    for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    Big b{}; b.out = nullptr;
    const int N = 20000;
    for (int mode = 0; mode < 6; mode++) {
        for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(k_null, dim3(64), dim3(256), 0, s[0], b);
        hipDeviceSynchronize();
        const double t0 = now();
        for (int i = 0; i < N; i++) {
            hipStream_t st = (mode & 1) ? s[i & 3] : s[0];
            if (mode < 2) hipLaunchKernelGGL(k_null, dim3(64), dim3(256), 0, st, b);
            else if (mode < 4) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, (int*)nullptr);
            else { hipLaunchKernelGGL(k_null, dim3(64), dim3(256), 0, st, b); hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, (int*)nullptr); }
        }
        const double t1 = now();
        hipDeviceSynchronize();
        const double t2 = now();
        const char* names[] = {"328-B kernarg, one stream", "328-B kernarg, four streams", "8-B kernarg, one stream", "8-B kernarg, four streams",
                               "pair (328 B + 8 B) on one stream", "pair, stream per pair of four"};
        printf("%-34s enqueue %.2f us per launch-call   drained %.2f us\n", names[mode], (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
    }
    // the same through hipModuleLaunchKernel with the kernarg block handed over as one buffer (no per-argument marshalling, no host-function lookup)
    hipFunction_t fn = nullptr;
    if (hipGetFuncBySymbol(&fn, (const void*)k_null) == hipSuccess && fn) {
        for (int mode = 0; mode < 2; mode++) {
            size_t sz = sizeof(b);
            void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &b, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            for (int i = 0; i < 2000; i++) hipModuleLaunchKernel(fn, 64, 1, 1, 256, 1, 1, 0, s[0], nullptr, extra);
            hipDeviceSynchronize();
            const double t0 = now();
            for (int i = 0; i < N; i++) hipModuleLaunchKernel(fn, 64, 1, 1, 256, 1, 1, 0, mode ? s[i & 3] : s[0], nullptr, extra);
            const double t1 = now();
            hipDeviceSynchronize();
            const double t2 = now();
            printf("%-34s enqueue %.2f us per launch-call   drained %.2f us\n", mode ? "hipModuleLaunchKernel, four streams" : "hipModuleLaunchKernel, one stream", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
        }
    } else printf("hipGetFuncBySymbol failed\n");
    return 0;
}
