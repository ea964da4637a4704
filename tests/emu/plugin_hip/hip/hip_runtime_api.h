// TEST INFRASTRUCTURE ONLY.  A host-memory stand-in for the handful of HIP runtime calls the Arrow
// registration shim (arrow_amd/csrc/arrow_plugin.cc) makes, so that the shim can be built against the
// emulated kernel library (tests/emu) and its device-resident code paths — kROCM buffers, device-aware
// kernels, the Acero node — can run in the GPU-less CPU tier.  "Device" memory is ordinary host memory
// that Arrow nevertheless sees as non-CPU (the shim's RocmBuffer reports kROCM), streams are
// synchronous.  Never shipped, never used by the product build.
#pragma once
#define ARX_EMULATED_HIP_RUNTIME 1   // the SIMT emulator behind it runs one kernel at a time
#include <cstddef>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2,
                     hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1 };

static inline const char* hipGetErrorString(hipError_t) { return "emulated HIP runtime"; }
static inline hipError_t hipMalloc(void** p, size_t n) {
  *p = std::aligned_alloc(256, (n + 255) / 256 * 256 + 256);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
enum { hipHostMallocDefault = 0 };
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(1); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum { hipErrorNotReady = 600 };
typedef void* hipEvent_t;
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(1); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }   // synchronous streams are always idle
#ifndef HIPEMU_DEVICE_QUERY
#define HIPEMU_DEVICE_QUERY
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
#endif
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
