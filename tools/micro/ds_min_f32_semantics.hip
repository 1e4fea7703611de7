// Semantics of ds_min_f32 / ds_min_i32 on gfx950 for the cases the depth keys can meet.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
__global__ void k(const float* mem, const float* val, float* out_f, float* out_vmin, int n) {
    __shared__ float s[64];
    const int t = threadIdx.x;
    s[t] = mem[t];
    __syncthreads();
    __hip_atomic_fetch_min(&s[t], val[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    out_f[t] = s[t];
    out_vmin[t] = fminf(mem[t], val[t]);
}
static float f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
static uint32_t u(float x) { uint32_t v; memcpy(&v, &x, 4); return v; }
int main() {
    const uint32_t cases[][2] = {
        {0x7F800000, 0x7FC00000}, {0x3F800000, 0x7FC00000}, {0x3F800000, 0xFFC00000}, {0x7F800000, 0x7F800001},
        {0x00000000, 0x80000000}, {0x80000000, 0x00000000}, {0x00000005, 0x00000003}, {0x00000003, 0x00000005},
        {0x00000003, 0x00000000}, {0x00000000, 0x00000003}, {0x80000003, 0x80000005}, {0x80000005, 0x80000003},
        {0xBF800000, 0xC0000000}, {0xC0000000, 0xBF800000}, {0x7F800000, 0x7F800000}, {0x7F800000, 0x3F000000},
        {0x3F000000, 0x00000007}, {0x00000007, 0x3F000000}, {0x80000007, 0x00000007}, {0x00000007, 0x80000007},
        {0x7F800000, 0xFF800000}, {0x00800000, 0x007FFFFF}, {0x007FFFFF, 0x00800000}, {0x7FC00000, 0x3F800000},
    };
    const int n = sizeof(cases) / sizeof(cases[0]);
    float hm[64] = {0}, hv[64] = {0}, ho[64], hvm[64];
    for (int i = 0; i < n; i++) { hm[i] = f(cases[i][0]); hv[i] = f(cases[i][1]); }
    float *dm, *dv, *dof, *dvm;
    hipMalloc(&dm, 256); hipMalloc(&dv, 256); hipMalloc(&dof, 256); hipMalloc(&dvm, 256);
    hipMemcpy(dm, hm, 256, hipMemcpyHostToDevice); hipMemcpy(dv, hv, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dm, dv, dof, dvm, n);
    hipMemcpy(ho, dof, 256, hipMemcpyDeviceToHost); hipMemcpy(hvm, dvm, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++)
        printf("mem %08x val %08x -> ds_min_f32 %08x   v_min_f32 %08x\n", cases[i][0], cases[i][1], u(ho[i]), u(hvm[i]));
    return 0;
}
