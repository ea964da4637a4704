// Shared device/host helpers for the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>

#include "../../include/arrow_amd.h"

namespace arx {

// A tuning knob (arx_set_option): written on any thread while launches on other threads read it.  Relaxed atomics — a
// call in flight sees the old or the new value of each knob, never a torn one, and knobs never change results.
template <typename T>
struct Knob {
  std::atomic<T> v;
  constexpr explicit Knob(T init) : v(init) {}
  operator T() const { return v.load(std::memory_order_relaxed); }
  Knob& operator=(T x) {
    v.store(x, std::memory_order_relaxed);
    return *this;
  }
};

constexpr int kWave = 64;
constexpr int kBlock = 256;             // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kTileRows = 4096;         // one wave tile = 64 lanes x one 64-bit mask word
constexpr int kTilesPerGroup = 64;      // count-kernel workgroup = 64 wave tiles

// ---------------------------------------------------------------------------
// Error plumbing (host)
// ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
// Per-file tuning-knob hooks used by arx_set_option (return 1 if the name was recognised).
int set_selection_option(const char* name, int64_t value);
int set_sort_option(const char* name, int64_t value);
int set_groupby_option(const char* name, int64_t value);
int set_parquet_option(const char* name, int64_t value);
// Per-file diagnostic counters read by arx_get_counter (return 1 if the name was recognised).
int get_groupby_counter(const char* name, int64_t* out);
int get_sort_counter(const char* name, int64_t* out);

#define ARX_HIP(call)                                          \
  do {                                                         \
    hipError_t e__ = (call);                                   \
    if (e__ != hipSuccess) return ::arx::hip_fail(e__, #call); \
  } while (0)

#define ARX_CHECK_LAUNCH(name)                                   \
  do {                                                           \
    hipError_t e__ = hipGetLastError();                          \
    if (e__ != hipSuccess) return ::arx::hip_fail(e__, name);    \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------------------
// Bitmap view: a bitmap with an arbitrary bit offset, read as aligned u64 words.
// `logical word i` = bits [64 i, 64 i + 64) of the logical bitmap (after offset).
// A NULL bitmap reads as all ones.  Bits at positions >= length read as zero.
// (Twin of the CPU-side BitBlockCounter/ShiftWord machinery,
//  cpp/src/arrow/util/bit_block_counter.h:42-47.)
// ---------------------------------------------------------------------------
struct Bits {
  const uint64_t* base;  // 8-byte aligned word holding the first logical bit (or NULL)
  int64_t nphys;         // number of physical words covering the logical range
  int64_t length;        // logical length in bits
  int shift;             // position of logical bit 0 inside base[0]
};

static inline Bits make_bits(const void* bitmap, int64_t bit_offset, int64_t length) {
  Bits b;
  b.length = length;
  if (bitmap == nullptr) {
    b.base = nullptr;
    b.nphys = 0;
    b.shift = 0;
    return b;
  }
  const uint64_t byte_addr = reinterpret_cast<uint64_t>(bitmap) + static_cast<uint64_t>(bit_offset >> 3);
  const int bit_in_byte = static_cast<int>(bit_offset & 7);
  const uint64_t aligned = byte_addr & ~uint64_t(7);
  b.base = reinterpret_cast<const uint64_t*>(aligned);
  b.shift = static_cast<int>((byte_addr - aligned) * 8) + bit_in_byte;
  b.nphys = (static_cast<int64_t>(b.shift) + length + 63) >> 6;
  return b;
}

// make_bits on the device (the bitmap's address comes out of a device-side table)
__device__ __forceinline__ Bits make_bits_device(const void* bitmap, int64_t bit_offset, int64_t length) {
  Bits b;
  b.length = length;
  if (bitmap == nullptr) {
    b.base = nullptr;
    b.nphys = 0;
    b.shift = 0;
    return b;
  }
  const uint64_t byte_addr = reinterpret_cast<uint64_t>(bitmap) + static_cast<uint64_t>(bit_offset >> 3);
  const int bit_in_byte = static_cast<int>(bit_offset & 7);
  const uint64_t aligned = byte_addr & ~uint64_t(7);
  b.base = reinterpret_cast<const uint64_t*>(aligned);
  b.shift = static_cast<int>((byte_addr - aligned) * 8) + bit_in_byte;
  b.nphys = (static_cast<int64_t>(b.shift) + length + 63) >> 6;
  return b;
}

__device__ __forceinline__ uint64_t low_mask64(int n) {  // n in [0,64]
  return n >= 64 ? ~uint64_t(0) : ((uint64_t(1) << n) - 1);
}

// Logical word `w` of the bitmap (see Bits).  Safe for any w >= 0.
__device__ __forceinline__ uint64_t load_word(const Bits& b, int64_t w) {
  const int64_t first = w << 6;
  if (first >= b.length) return 0;
  const int64_t remain = b.length - first;
  uint64_t v;
  if (b.base == nullptr) {
    v = ~uint64_t(0);
  } else {
    v = b.base[w] >> b.shift;
    if (b.shift != 0 && (w + 1) < b.nphys) v |= b.base[w + 1] << (64 - b.shift);
  }
  if (remain < 64) v &= low_mask64(static_cast<int>(remain));
  return v;
}

// Branch-free variant for hot loops (no control flow => the compiler can keep several of these
// loads in flight).  Requires b.base != NULL and b.nphys >= 1.
__device__ __forceinline__ uint64_t load_word_nb(const Bits& b, int64_t w) {
  const int64_t last = b.nphys - 1;
  const int64_t i0 = w < last ? w : last;
  const int64_t i1 = (w + 1) < last ? (w + 1) : last;
  const uint64_t lo = b.base[i0];
  const uint64_t hi = b.base[i1];
  const uint64_t v = (lo >> b.shift) | ((hi << 1) << (63 - b.shift));
  const int64_t remain = b.length - (w << 6);
  const uint64_t m = remain >= 64 ? ~uint64_t(0)
                                  : (remain <= 0 ? uint64_t(0) : ((uint64_t(1) << remain) - 1));
  return v & m;
}

// ---------------------------------------------------------------------------
// XCD-aware work mapping.  The dispatcher places workgroup b on XCD b % 8 (observed, not promised: only speed depends on
// it).  xcd_contiguous(b, n) renumbers the n workgroups so that XCD x works on ONE contiguous range of work items
// instead of every eighth one: work items that write the same region (the tiles of one partition) then share an L2.
// Bijective for any n (cdna_hip_programming.md, "XCD swizzle must be bijective").
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t b, uint32_t n) {
  const uint32_t xcd = b & 7u, q = n >> 3, r = n & 7u;
  const uint32_t first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return first + (b >> 3);
}

// The ORDER KEY of a double: the int64 whose signed order is the numeric order of the doubles (-0.0 just below +0.0, NaN
// patterns at both ends) — float extrema ride on the integer min / max machinery through it (groupby.hip, reduce_float.hip).
__device__ __forceinline__ long long float_order_key(double v) {
  const unsigned long long u = static_cast<unsigned long long>(__builtin_bit_cast(long long, v));
  return static_cast<long long>((u >> 63) ? (~u ^ 0x8000000000000000ull) : u);
}
__device__ __forceinline__ double float_from_order_key(long long k) {
  const unsigned long long s = static_cast<unsigned long long>(k);
  const double v = __builtin_bit_cast(double, k >= 0 ? s : ~(s ^ 0x8000000000000000ull));
  return v != v ? __builtin_bit_cast(double, 0x7FF8000000000000ull) : v;
}


// ---------------------------------------------------------------------------
// Wave-level primitives (64 lanes)
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t n = __shfl_up(v, d, 64);
    if (lane >= d) v += n;
  }
  return v;
}

__device__ __forceinline__ uint64_t wave_inclusive_scan_u64(uint64_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t n = __shfl_up(v, d, 64);
    if (lane >= d) v += n;
  }
  return v;
}

__device__ __forceinline__ uint32_t wave_reduce_sum_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__device__ __forceinline__ uint64_t wave_reduce_sum_u64(uint64_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
  uint32_t lo = __shfl(static_cast<uint32_t>(v), src, 64);
  uint32_t hi = __shfl(static_cast<uint32_t>(v >> 32), src, 64);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

// Software parallel-bit-extract: gathers the bits of x selected by m into the low
// popcount(m) bits, preserving order (Hacker's Delight 7-4 "compress").
__device__ __forceinline__ uint64_t pext64(uint64_t x, uint64_t m) {
  x &= m;
  uint64_t mk = ~m << 1;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    uint64_t mp = mk ^ (mk << 1);
    mp ^= mp << 2;
    mp ^= mp << 4;
    mp ^= mp << 8;
    mp ^= mp << 16;
    mp ^= mp << 32;
    const uint64_t mv = mp & m;
    m = (m ^ mv) | (mv >> (1 << i));
    const uint64_t t = x & mv;
    x = (x ^ t) | (t >> (1 << i));
    mk &= ~mp;
  }
  return x;
}

// Raw buffer descriptor over [base, base + bytes): loads whose offset falls outside return 0
// WITHOUT touching memory, which turns "skip this lane" into a plain offset select (no exec
// masking, no branch) and lets the compiler keep several loads in flight with counted vmcnt.
// `base` must be wave-uniform (the descriptor lives in SGPRs).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), /*stride=*/0,
                                           static_cast<int>(bytes), /*flags=*/0x00020000);
}
constexpr uint32_t kBufferSkip = 0x80000000u;  // an offset no descriptor of ours covers
__device__ __forceinline__ uint4 buffer_load_b128(__amdgpu_buffer_rsrc_t r, uint32_t byte_offset) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_offset, 0, 0));
}
// the same load with the non-temporal policy (aux bit 1 = nt on gfx950)
__device__ __forceinline__ uint4 buffer_load_b128_nt(__amdgpu_buffer_rsrc_t r, uint32_t byte_offset) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_offset, 0, 2));
}

// Non-temporal (streaming) loads and stores: `global_load/store ... nt`.  A one-pass stream read or written with the
// default policy competes for the L2 with itself; with nt a one-shot grid reads at 7.3 TB/s instead of 6.6 and the
// f64 -> f32 cast moves 6.4 TB/s instead of 5.6-6.1 (scripts/micro/stream_bench.hip, profiles/r03_a_*).  The raw
// 16- / 8-byte carriers are native vectors (the builtin does not take HIP's struct vector types).
typedef uint32_t arx_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t arx_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 nt_load16(const void* p) {
  const arx_u32x4 v = __builtin_nontemporal_load(static_cast<const arx_u32x4*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store16(void* p, uint4 q) {
  arx_u32x4 v;
  v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  __builtin_nontemporal_store(v, static_cast<arx_u32x4*>(p));
}
__device__ __forceinline__ uint2 nt_load8(const void* p) {
  const arx_u32x2 v = __builtin_nontemporal_load(static_cast<const arx_u32x2*>(p));
  return make_uint2(v[0], v[1]);
}
__device__ __forceinline__ void nt_store8(void* p, uint2 q) {
  arx_u32x2 v;
  v[0] = q.x; v[1] = q.y;
  __builtin_nontemporal_store(v, static_cast<arx_u32x2*>(p));
}
template <typename T>
__device__ __forceinline__ T nt_load(const T* p) {
  return __builtin_nontemporal_load(p);
}
template <typename T>
__device__ __forceinline__ void nt_store(T* p, T v) {
  __builtin_nontemporal_store(v, p);
}

// 16 bytes from an address of any alignment (one global_load_dwordx4: global memory takes the unaligned address)
__device__ __forceinline__ uint4 load16_unaligned(const uint8_t* p) {
  uint4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

// Spread the low 32 bits of x to the even bit positions of a 64-bit word.
__device__ __forceinline__ uint64_t spread32(uint64_t x) {
  x &= 0xffffffffull;
  x = (x | (x << 16)) & 0x0000ffff0000ffffull;
  x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
  x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace arx
