// Cost of one ds_min_u64 (no return) per wave for the address patterns the raster's consumer produces.
// Built as a shared object and driven from Python (a bare executable is not allowed to use the GPU on the dev boxes):
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/micro/lds_atomic_patterns.hip -o /tmp/lap.so && python3 -c "import ctypes; ctypes.CDLL('/tmp/lap.so').run()"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(int pattern, int iters, unsigned long long* out) {
    __shared__ unsigned long long keys[2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2048; i += blockDim.x) keys[i] = ~0ull;
    __syncthreads();
    int idx;
    switch (pattern) {
        case 0: idx = lane; break;                                        // 64 consecutive pixels of one row
        case 1: idx = 4 * (lane & 7); break;                              // 8 addresses, 8 lanes each (rotated groups)
        case 2: idx = 4 * (lane & 7) + ((lane >> 3) & 3); break;          // 32 addresses, 2 lanes each
        case 3: idx = 0; break;                                           // one address
        case 4: idx = 4 * lane; break;                                    // 64 addresses 32 B apart (lane = 4-pixel unit)
        case 5: idx = (lane * 37 + 11) & 2047; break;                     // scattered
        case 6: idx = 4 * (lane & 15); break;                             // 16 addresses 32 B apart, 4 lanes each (round 2's units)
        case 7: idx = 64 * ((lane >> 3) & 7) + 4 * (lane & 7); break;     // 8 rows x 8 groups: 64 addresses, 8 bank pairs
        default: idx = lane; break;
    }
    idx += 64 * wave;                                                      // the four waves on different rows
    unsigned long long key = 0x7000000000000000ull + (unsigned)tid;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int q = 0; q < 4; q++) { atomicMin(&keys[(idx + q) & 2047], key); key -= 64; }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 8 + wave] = (unsigned long long)(t1 - t0);
    if (keys[tid] == 12345ull) out[63] = 1;
}
extern "C" int run() {
    unsigned long long* out; hipMalloc(&out, 64 * 8);
    const char* names[8] = {"64 consecutive px", "8 addr x 8 lanes", "32 addr x 2 lanes", "1 addr x 64 lanes", "64 addr, 32 B apart", "scattered", "16 addr x 4 lanes (32 B apart)", "8 rows x 8 groups"};
    for (int waves = 1; waves <= 4; waves *= 4)
        for (int p = 0; p < 8; p++) {
            hipMemset(out, 0, 64 * 8);
            hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, p, 2000, out);
            unsigned long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
            printf("%d wave(s)  %-32s %6.1f clocks per ds_min_u64 (per wave)\n", waves, names[p], (double)h[0] / (2000.0 * 4));
        }
    return 0;
}
