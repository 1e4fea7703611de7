// VALU issue-rate microbenchmark for gfx950: cycles per wave64 VALU instruction per SIMD, by op type, dependency
// structure and waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITERS = 2000;
constexpr int PER = 64;   // instructions per loop iteration

// KIND: 0 independent f32 mul/add (8 chains), 1 one dependent f32 chain, 2 independent int add/and/xor, 3 v_mul_lo_u32 indep,
//       4 v_cvt_f32_i32 / v_cvt_i32_f32 indep, 5 v_cndmask indep (vcc), 6 two dependent chains, 7 four dependent chains,
//       8 v_fma_f32 independent, 9 v_pk_mul_f32 independent, 10 v_cmp_lt_f32 to sgpr pairs indep
template <int KIND>
__global__ void k(float* out, long long* cyc, float seed) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + (float)(threadIdx.x + i);
    int b[8];
    for (int i = 0; i < 8; i++) b[i] = (int)threadIdx.x * 7 + i;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
        if (KIND == 0) {
#pragma unroll
            for (int j = 0; j < PER / 8; j++) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            }
        } else if (KIND == 1) {
#pragma unroll
            for (int j = 0; j < PER; j++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(seed));
        } else if (KIND == 2) {
#pragma unroll
            for (int j = 0; j < PER / 8; j++) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(b[i]) : "v"(b[(i + 1) & 7]));
            }
        } else if (KIND == 3) {
#pragma unroll
            for (int j = 0; j < PER / 8; j++) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b[i]) : "v"(b[(i + 1) & 7]));
            }
        } else if (KIND == 4) {
#pragma unroll
            for (int j = 0; j < PER / 8; j++) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
                    asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(b[i + 1]) : "v"(a[i + 1]));
                }
            }
        } else if (KIND == 5) {
#pragma unroll
            for (int j = 0; j < PER / 8; j++) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b[i]) : "v"(b[(i + 1) & 7]));
            }
        } else if (KIND == 6) {
#pragma unroll
            for (int j = 0; j < PER / 2; j++) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(seed));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[1]) : "v"(seed));
            }
        } else if (KIND == 7) {
#pragma unroll
            for (int j = 0; j < PER / 4; j++) {
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            }
        } else if (KIND == 8) {
#pragma unroll
            for (int j = 0; j < PER / 8; j++) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
            }
        } else if (KIND == 9) {
            float2* p = reinterpret_cast<float2*>(a);
            float2 s2 = make_float2(seed, seed);
#pragma unroll
            for (int j = 0; j < PER / 4; j++) {
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
            }
        } else if (KIND == 11) {   // v_cmp e64 -> sgpr pair, v_cndmask e64 reading it (the span-walk pattern), 4 independent pairs
#pragma unroll
            for (int j = 0; j < PER / 8; j++) {
                asm volatile("v_cmp_lt_i32 s[20:21], %0, %1\n\tv_cmp_lt_i32 s[22:23], %1, %2\n\tv_cmp_lt_i32 s[24:25], %2, %3\n\tv_cmp_lt_i32 s[26:27], %3, %0\n\t"
                             "v_cndmask_b32 %0, %0, %1, s[20:21]\n\tv_cndmask_b32 %1, %1, %2, s[22:23]\n\tv_cndmask_b32 %2, %2, %3, s[24:25]\n\tv_cndmask_b32 %3, %3, %0, s[26:27]"
                             : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) :: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            }
        } else if (KIND == 12) {   // v_cmp e32 -> vcc, v_cndmask e32 reading vcc, dependent pair
#pragma unroll
            for (int j = 0; j < PER / 2; j++) {
                asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(b[j & 3]) : "v"(b[4 + (j & 3)]) : "vcc");
            }
        } else if (KIND == 13) {   // VOP3 integer ops: v_bfe_u32, v_and_or_b32, v_lshl_add_u32, v_add3_u32
#pragma unroll
            for (int j = 0; j < PER / 4; j++) {
                asm volatile("v_bfe_u32 %0, %0, 3, 9\n\tv_and_or_b32 %1, %1, 63, %0\n\tv_lshl_add_u32 %2, %2, 2, %1\n\tv_add3_u32 %3, %3, %2, %0"
                             : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
            }
        } else if (KIND == 14) {   // s_and_saveexec / one VALU / s_or exec (the per-pixel predication pattern): counts 1 VALU + 2 SALU per group
#pragma unroll
            for (int j = 0; j < PER; j++) {
                asm volatile("s_mov_b64 s[22:23], -1\n\ts_and_saveexec_b64 s[20:21], s[22:23]\n\tv_add_f32 %0, %0, %1\n\ts_or_b64 exec, exec, s[20:21]" : "+v"(a[j & 7]) : "v"(seed) : "s20", "s21", "s22", "s23");
            }
        } else if (KIND == 15) {   // v_readlane_b32
#pragma unroll
            for (int j = 0; j < PER / 4; j++) {
                asm volatile("v_readlane_b32 s20, %0, 63\n\tv_readlane_b32 s21, %1, 5\n\tv_readlane_b32 s22, %2, 7\n\tv_readlane_b32 s23, %3, 9" :: "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]) : "s20", "s21", "s22", "s23");
            }
        } else if (KIND == 16) {   // mbcnt pair
#pragma unroll
            for (int j = 0; j < PER / 2; j++) {
                asm volatile("v_mbcnt_lo_u32_b32 %0, s22, 0\n\tv_mbcnt_hi_u32_b32 %0, s23, %0" : "=v"(b[j & 7]));
            }
        } else if (KIND == 17) {   // f32 mul + add pairs as the per-pixel maths: 12 ops, 3-deep dependency, 4 pixels interleaved
#pragma unroll
            for (int j = 0; j < PER / 16; j++) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    asm volatile("v_mul_f32 %0, %2, %0\n\tv_mul_f32 %1, %2, %1" : "+v"(a[i]), "+v"(a[4 + i]) : "v"(seed));
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    asm volatile("v_add_f32 %0, %0, %1\n\tv_sub_f32 %1, %2, %0" : "+v"(a[i]), "+v"(a[4 + i]) : "v"(seed));
                }
            }
        } else if (KIND == 10) {
#pragma unroll
            for (int j = 0; j < PER / 4; j++) {
                asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cmp_lt_f32 s[22:23], %1, %0\n\tv_cmp_lt_f32 s[24:25], %0, %0\n\tv_cmp_lt_f32 s[26:27], %1, %1"
                             :: "v"(a[0]), "v"(a[1]) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; int sb = 0;
    for (int i = 0; i < 8; i++) { s += a[i]; sb += b[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)sb;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
int run(const char* name, int factor) {
    float* out; long long* cyc;
    const int CUS = 256;
    CHECK(hipMalloc(&out, sizeof(float) * CUS * 2048));
    CHECK(hipMalloc(&cyc, sizeof(long long) * CUS * 8));
    printf("%-34s", name);
    for (int wps : {1, 2, 3, 4, 5, 8}) {   // waves per SIMD: one workgroup per CU of wps*4 waves
        const int threads = wps * 4 * 64;
        if (threads > 1024) {              // two workgroups per CU
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k<KIND>, dim3(CUS * 2), dim3(threads / 2), 0, 0, out, cyc, 1.0001f);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<KIND>, dim3(CUS * 2), dim3(threads / 2), 0, 0, out, cyc, 1.0001f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // total wave-instructions per SIMD = wps * ITERS * PER * factor
            const double cyc_per = ms * 1e-3 * 2.4e9 / ((double)wps * ITERS * PER * factor);
            printf("  %dw: %.2f", wps, cyc_per);
            continue;
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(CUS), dim3(threads), 0, 0, out, cyc, 1.0001f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(CUS), dim3(threads), 0, 0, out, cyc, 1.0001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(CUS);
        hipMemcpy(h.data(), cyc, sizeof(long long) * CUS, hipMemcpyDeviceToHost);
        const double cyc_per = ms * 1e-3 * 2.4e9 / ((double)wps * ITERS * PER * factor);
        printf("  %dw: %.2f", wps, cyc_per);
    }
    printf("   (cycles @2.4GHz per wave-instruction per SIMD, from wall time)\n");
    hipFree(out); hipFree(cyc);
    return 0;
}

int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int only = argc > 1 ? atoi(argv[1]) : -1;
#define RUN(K, NAME) if (only < 0 || only == K) run<K>(NAME, 1);
    RUN(0, "f32 mul, 8 independent chains")
    RUN(8, "f32 fma, 8 independent chains")
    RUN(1, "f32 add, 1 dependent chain")
    RUN(6, "f32 add, 2 dependent chains")
    RUN(7, "f32 add, 4 dependent chains")
    RUN(2, "u32 add, 8 chains")
    RUN(3, "v_mul_lo_u32, 8 chains")
    RUN(4, "v_cvt f32<->i32")
    RUN(5, "v_cndmask (vcc)")
    RUN(9, "v_pk_mul_f32 (per pk instr)")
    RUN(10, "v_cmp_lt_f32 -> sgpr pair")
    RUN(11, "v_cmp e64->sgpr + v_cndmask e64 x4")
    RUN(12, "v_cmp vcc + v_cndmask vcc (dep)")
    RUN(13, "VOP3 int: bfe/and_or/lshl_add/add3")
    RUN(14, "saveexec + v_add + or exec (per grp)")
    RUN(15, "v_readlane_b32")
    RUN(16, "v_mbcnt lo/hi")
    RUN(17, "f32 mul/add pairs, 4 px interleaved")
    return 0;
}
