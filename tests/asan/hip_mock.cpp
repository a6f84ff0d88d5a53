// Test double of the HIP runtime for the host-side sanitizer build (tests/asan/Makefile): device memory is
// host memory, copies are memcpy, kernel launches do nothing.  It exists so that the HOST logic of libgmhip --
// program validation and lowering, the per-device program cache, the caching allocator, the specialiser's
// source generator -- can run under AddressSanitizer / UBSan on a box without a GPU (GPU-side sanitizers are
// not available on the pool).  Nothing here is product code.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>

#include "gm_vm.h"

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)0x1; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
  memset(p, 0, sizeof(*p));
  p->multiProcessorCount = 256;
  strcpy(p->name, "mock");
  strcpy(p->gcnArchName, "gfx950");
  return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return hipSuccess; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)1 << 34; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "mock"; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipModuleLoadData(hipModule_t*, const void*) { return hipErrorNoBinaryForGpu; }
hipError_t hipModuleGetFunction(hipFunction_t*, hipModule_t, const char*) { return hipErrorNotFound; }
hipError_t hipModuleLaunchKernel(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t,
                                 void**, void**) { return hipSuccess; }
}

// the kernel launchers of the translation units that hold device code (not linked into the sanitizer build)
namespace gmvm {
template <typename RC>
int launch_class(const VmLaunch&, int, int, int, dim3, dim3, hipStream_t) { return GM_OK; }
template int launch_class<float>(const VmLaunch&, int, int, int, dim3, dim3, hipStream_t);
template int launch_class<double>(const VmLaunch&, int, int, int, dim3, dim3, hipStream_t);
template int launch_class<int>(const VmLaunch&, int, int, int, dim3, dim3, hipStream_t);
template int launch_class<long long>(const VmLaunch&, int, int, int, dim3, dim3, hipStream_t);
}  // namespace gmvm
namespace gm {
int lut8_launch(int, const void* const*, const uint32_t*, void* const*, const uint32_t*, size_t, size_t, const void*, int, hipStream_t) { return GM_OK; }
int lookup32_launch(int, const void* const*, const uint32_t*, void* const*, const uint32_t*, size_t, size_t, const Lookup32&, int, hipStream_t) { return GM_OK; }
}  // namespace gm
