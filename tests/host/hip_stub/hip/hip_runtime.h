// A FAKE HIP runtime for the thread-sanitizer build of the C-ABI's host layer (tests/host/tsan_host_test.cpp):
// just enough of <hip/hip_runtime.h> for software-renderer_amd/csrc/swr_api.hip to compile as plain C++.
// Streams are in-order queues drained by one host thread each; events complete when their stream reaches them;
// "device memory" is host memory; kernels are the stand-ins of stub_launch.cpp.  Test infrastructure only: the
// product is built by hipcc against the real runtime and has no CPU path.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct int4 { int x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotReady = 600, hipErrorInvalidValue = 1, hipErrorLaunchFailure = 719 };
typedef struct fake_stream* hipStream_t;
typedef struct fake_event* hipEvent_t;
enum { hipStreamNonBlocking = 1 };
enum { hipEventDisableTiming = 2, hipEventDisableSystemFence = 0x20000000 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
enum { hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostMallocMapped = 2, hipHostRegisterPortable = 1 };
enum { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeUnregistered = 0 };
struct hipPointerAttribute_t { int type; };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned flags);
hipError_t hipHostRegister(void* p, size_t n, unsigned flags);
hipError_t hipHostUnregister(void* p);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s);

// test-side hooks of the fake runtime
#include <functional>
void fake_enqueue(hipStream_t s, std::function<void()> op, hipEvent_t stop = nullptr);   // a "kernel"
void fake_kernel_delay_us(int us);                                                           // how long a stand-in kernel takes
