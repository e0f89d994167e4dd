// hip_stub.cpp — TEST INFRASTRUCTURE: a stand-in for the HIP runtime on host memory, for the CPU-only sanitizer build of the PRODUCT's host code
// (tests/test_host_sanitizers.py).  The library's translation units are compiled host-only (`hipcc --cuda-host-only -fsanitize=address,undefined`)
// and linked against this file instead of libamdhip64: "device" memory is calloc'ed host memory (so AddressSanitizer sees every byte the host
// code uploads, every table it builds and every staging copy), copies are memcpy, kernel launches return success without running anything
// (the device code is not part of this build).  What runs for real is everything the host side does: geometry, planners (pyramid fusion /
// chains, FAST work items, quadtree key tables), workspace sizing, staging, the ingest ticket state machine, the vocabulary loaders.
// Nothing of this is linked into the product.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>

extern "C" {
typedef int hipError_t;                     // hipSuccess = 0
typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;
struct dim3s { uint32_t x, y, z; };

static std::atomic<long> g_launches{0};
long hip_stub_launches() { return g_launches.load(); }

hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
hipError_t hipSetDevice(int) { return 0; }
hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
hipError_t hipDeviceSynchronize() { return 0; }
hipError_t hipGetLastError() { return 0; }
const char* hipGetErrorString(hipError_t) { return "hip_stub"; }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? 0 : 2; }
hipError_t hipFree(void* p) { free(p); return 0; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(n ? n : 1, 1); return *p ? 0 : 2; }
hipError_t hipHostFree(void* p) { free(p); return 0; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int) { for (size_t y = 0; y < h; y++) memcpy((char*)d + y * dp, (const char*)s + y * sp, w); return 0; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int k, hipStream_t) { return hipMemcpy2D(d, dp, s, sp, w, h, k); }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)calloc(1, 8); return 0; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return 0; }
hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)calloc(1, 8); return 0; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)calloc(1, 8); return 0; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return 0; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
hipError_t hipEventQuery(hipEvent_t) { return 0; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
hipError_t hipMemcpyFromSymbol(void*, const void*, size_t, size_t, int) { return 0; }
hipError_t hipMemcpyToSymbol(const void*, const void*, size_t, size_t, int) { return 0; }
hipError_t hipGetSymbolAddress(void** p, const void*) { static char z[4096]; *p = z; return 0; }
// kernel launches: the host-side stubs clang generates for __global__ functions call these three
hipError_t __hipPushCallConfiguration(dim3s, dim3s, size_t, hipStream_t) { return 0; }
hipError_t __hipPopCallConfiguration(dim3s*, dim3s*, size_t*, hipStream_t*) { return 0; }
hipError_t hipLaunchKernel(const void*, dim3s, dim3s, void**, size_t, hipStream_t) { g_launches++; return 0; }
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipRegisterManagedVar(void**, void*, void*, const char*, size_t, unsigned) {}
void __hipUnregisterFatBinary(void**) {}
}
