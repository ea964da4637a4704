// hip_emu.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A tiny SIMT emulator that lets the *unchanged* kernel sources under arrow_amd/csrc/ be
// compiled with g++ and executed on the CPU, so their logic (wave collectives, LDS staging,
// tile bookkeeping, atomics) can be checked against the oracle in this GPU-less container
// before GPU minutes are spent.  Nothing under arrow_amd/ knows about it and the product
// never loads the emulated library: only tests/ builds and dlopens it.
//
// Model: one workgroup at a time; every HIP thread is a fiber on its own stack; wave collectives
// (__shfl*, __ballot, __any, wave_barrier) and __syncthreads are rendezvous points at which
// fibers yield to a round-robin scheduler.  Lanes do NOT run in lockstep, so any cross-lane
// LDS dependency that the real hardware gets "for free" must be marked in the source with
// __builtin_amdgcn_wave_barrier() — which is also what pins the compiler on the GPU.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <type_traits>
#include <vector>

#define __global__
#define __device__
#define __host__
#ifndef __forceinline__
#define __forceinline__ inline __attribute__((always_inline))
#endif
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct __attribute__((aligned(16))) uint4 { uint32_t x, y, z, w; };
struct __attribute__((aligned(16))) double2 { double x, y; };
struct __attribute__((aligned(16))) longlong2 { long long x, y; };
struct __attribute__((aligned(8))) float2 { float x, y; };
struct __attribute__((aligned(8))) uint2 { uint32_t x, y; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }

// ---------------------------------------------------------------- runtime API stubs
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef void* hipStream_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
#ifndef HIPEMU_DEVICE_QUERY
#define HIPEMU_DEVICE_QUERY
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
// few "CUs": persistent kernels walk several work items per workgroup here, as they do on the 256-CU part
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 2; return hipSuccess; }
#endif
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }

// ---------------------------------------------------------------- emulator core
namespace hipemu {

struct Wave {
  uint64_t slots[64];
  int nactive = 0;
  int arrived = 0;
  unsigned gen = 0;
};

// A fiber's context is its saved stack pointer: the switch (hip_emu_runtime.cpp) pushes the callee-saved
// registers and swaps stacks — no signal-mask system call per switch, which is what made
// swapcontext() the dominant cost of the CPU tier.
struct Fiber {
  void* sp = nullptr;
  dim3 tidx;
  int tid = 0;
  bool done = false;
};

struct State {
  dim3 grid, block, bidx;
  Fiber* cur = nullptr;
  void* sched_sp = nullptr;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  int block_nactive = 0, block_arrived = 0;
  unsigned block_gen = 0;
  const std::function<void()>* body = nullptr;
};

State& st();
void yield();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);

inline Wave& my_wave() { return st().waves[st().cur->tid >> 6]; }
inline int my_lane() { return st().cur->tid & 63; }

inline void wave_sync() {
  Wave& w = my_wave();
  const unsigned gen = w.gen;
  if (++w.arrived >= w.nactive) {
    w.arrived = 0;
    ++w.gen;
  } else {
    while (w.gen == gen) yield();
  }
}

inline void block_sync() {
  State& s = st();
  const unsigned gen = s.block_gen;
  if (++s.block_arrived >= s.block_nactive) {
    s.block_arrived = 0;
    ++s.block_gen;
  } else {
    while (s.block_gen == gen) yield();
  }
}

template <typename T>
inline T exchange(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "shuffle of <= 64-bit values");
  Wave& w = my_wave();
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  w.slots[my_lane()] = raw;
  wave_sync();
  T r;
  raw = w.slots[src_lane & 63];
  memcpy(&r, &raw, sizeof(T));
  wave_sync();
  return r;
}

inline uint64_t ballot(bool pred) {
  Wave& w = my_wave();
  w.slots[my_lane()] = pred ? 1 : 0;
  wave_sync();
  uint64_t m = 0;
  const int lanes = std::min<int>(64, static_cast<int>(st().fibers.size()) - (st().cur->tid & ~63));
  for (int i = 0; i < lanes; ++i) {
    if (w.slots[i] && !st().fibers[(st().cur->tid & ~63) + i].done) m |= (uint64_t(1) << i);
  }
  wave_sync();
  return m;
}

}  // namespace hipemu

#define threadIdx (hipemu::st().cur->tidx)
#define blockIdx (hipemu::st().bidx)
#define blockDim (hipemu::st().block)
#define gridDim (hipemu::st().grid)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })

#define __syncthreads() hipemu::block_sync()
#define __builtin_amdgcn_wave_barrier() hipemu::wave_sync()
static inline void __threadfence_block() {}
// A device-scope fence is where a protocol hands data to other threads: the emulator lets the other fibers of the
// workgroup run there, so that readers do see the state between "claimed" and "published" (grouper.hip).
static inline void __threadfence() { hipemu::yield(); }
#define __builtin_amdgcn_fence(...) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(ptr, order, scope) (*(ptr))
#define __hip_atomic_store(ptr, value, order, scope) (*(ptr) = (value))

template <typename T> inline T __shfl(T v, int src, int = 64) { return hipemu::exchange(v, src); }
template <typename T> inline T __shfl_up(T v, unsigned d, int = 64) {
  const int lane = hipemu::my_lane();
  const int src = lane - static_cast<int>(d);
  return hipemu::exchange(v, src < 0 ? lane : src);
}
template <typename T> inline T __shfl_down(T v, unsigned d, int = 64) {
  const int lane = hipemu::my_lane();
  const int src = lane + static_cast<int>(d);
  return hipemu::exchange(v, src > 63 ? lane : src);
}
template <typename T> inline T __shfl_xor(T v, int m, int = 64) {
  return hipemu::exchange(v, hipemu::my_lane() ^ m);
}
// buffer descriptors: out-of-range offsets read as zero without touching memory
struct __amdgpu_buffer_rsrc_t { const uint8_t* base; uint32_t num; };
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int num, int) {
  return __amdgpu_buffer_rsrc_t{static_cast<const uint8_t*>(p), static_cast<uint32_t>(num)};
}
inline uint4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, int) {
  uint4 v{0, 0, 0, 0};
  const uint64_t o = static_cast<uint64_t>(voff) + soff;
  if (o + 16 <= r.num) memcpy(&v, r.base + o, 16);
  return v;
}
// only ever applied to wave-uniform values in the kernels
// non-temporal accesses are ordinary accesses here; the native-vector carriers of arx_common.h become GCC vectors
// (4-byte elements only)
#define ext_vector_type(n) vector_size(4 * (n))
template <typename T> inline T __builtin_nontemporal_load(const T* p) { return *p; }
template <typename T> inline void __builtin_nontemporal_store(T v, T* p) { *p = v; }
inline void __builtin_amdgcn_sched_barrier(int) {}   // an instruction-scheduling fence: nothing to emulate
template <typename T> inline T __builtin_amdgcn_readfirstlane(T v) { return v; }
inline uint32_t __builtin_amdgcn_readlane(uint32_t v, int src) { return hipemu::exchange(v, src); }
inline uint64_t __ballot(bool p) { return hipemu::ballot(p); }
inline bool __any(bool p) { return hipemu::ballot(p) != 0; }
inline bool __all(bool p) { return hipemu::ballot(!p) == 0; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll(static_cast<long long>(x)); }

// fibers of one workgroup are scheduled cooperatively on one OS thread: plain RMW is atomic
template <typename T> using emu_id_t = typename std::type_identity<T>::type;
template <typename T> inline T atomicAdd(T* p, emu_id_t<T> v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicOr(T* p, emu_id_t<T> v) { T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicMin(T* p, emu_id_t<T> v) { T o = *p; *p = std::min<T>(o, v); return o; }
template <typename T> inline T atomicMax(T* p, emu_id_t<T> v) { T o = *p; *p = std::max<T>(o, v); return o; }
template <typename T> inline T atomicExch(T* p, emu_id_t<T> v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicCAS(T* p, emu_id_t<T> c, emu_id_t<T> v) { T o = *p; if (o == c) *p = v; return o; }
