// What does a streaming kernel reach on THIS box, and which form gets the cast / greater kernels there?
// (every streaming kernel tops out at ~5.2 TB/s; the guide's float4 copy measures 6.29.)
//   copy     float4 -> float4                       (8 B read + 8 B written per 8 B: "copy ceiling")
//   read     float4 -> nothing (xor-reduced)        (read-only ceiling)
//   cast2    double2 -> float2   (today's cast_f64_f32_kernel: 16-B loads, 8-B stores)
//   cast4s   the lane's 4 consecutive doubles (2 x 16-B loads, 32-B lane stride) -> one float4 store
//   cast4x   2 contiguous 16-B loads per lane, pairs exchanged with ds_bpermute -> one float4 store
//   gt       two double2 loads per operand -> 64-lane ballots -> u64 words
// each with U independent loads in flight, plain or non-temporal, grid-stride (blocks = 256 CUs x k) or one-shot.
//   usage: stream_bench [log2 elements of 8 bytes = 30]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT, typename T> __device__ __forceinline__ T ld(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NT, typename T> __device__ __forceinline__ void st(T* p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

__global__ void fill(double* a, int64_t n, uint64_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t z = seed + (uint64_t)i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 27;
    a[i] = (double)(int64_t)z * 1e-10;
  }
}

// n16 = number of 16-byte units
template <int U, bool NT, int THREADS>
__global__ __launch_bounds__(THREADS) void copy_kernel(const f4* __restrict__ in, f4* __restrict__ out, int64_t n16) {
  const int64_t per = (int64_t)THREADS * U;
  for (int64_t base = (int64_t)blockIdx.x * per; base < n16; base += (int64_t)gridDim.x * per) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(in + base + u * THREADS + threadIdx.x);
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(out + base + u * THREADS + threadIdx.x, v[u]);
  }
}
template <int U, bool NT, int THREADS>
__global__ __launch_bounds__(THREADS) void read_kernel(const f4* __restrict__ in, f4* __restrict__ out, int64_t n16) {
  const int64_t per = (int64_t)THREADS * U;
  f4 acc = {0, 0, 0, 0};
  for (int64_t base = (int64_t)blockIdx.x * per; base < n16; base += (int64_t)gridDim.x * per) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(in + base + u * THREADS + threadIdx.x);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x == 1.2345f) out[0] = acc;
}
// n2 = number of double pairs
template <int U, bool NT, int THREADS>
__global__ __launch_bounds__(THREADS) void cast2_kernel(const d2* __restrict__ in, f2* __restrict__ out, int64_t n2) {
  const int64_t per = (int64_t)THREADS * U;
  for (int64_t base = (int64_t)blockIdx.x * per; base < n2; base += (int64_t)gridDim.x * per) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(in + base + u * THREADS + threadIdx.x);
#pragma unroll
    for (int u = 0; u < U; ++u) { f2 o = {(float)v[u].x, (float)v[u].y}; st<NT>(out + base + u * THREADS + threadIdx.x, o); }
  }
}
// n4 = number of double quads; lane owns quad q: loads in[2q], in[2q+1]
template <int U, bool NT, int THREADS>
__global__ __launch_bounds__(THREADS) void cast4s_kernel(const d2* __restrict__ in, f4* __restrict__ out, int64_t n4) {
  const int64_t per = (int64_t)THREADS * U;
  for (int64_t base = (int64_t)blockIdx.x * per; base < n4; base += (int64_t)gridDim.x * per) {
    d2 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t q = base + u * THREADS + threadIdx.x;
      a[u] = ld<NT>(in + 2 * q); b[u] = ld<NT>(in + 2 * q + 1);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f4 o = {(float)a[u].x, (float)a[u].y, (float)b[u].x, (float)b[u].y};
      st<NT>(out + base + u * THREADS + threadIdx.x, o);
    }
  }
}
// contiguous loads: per wave and step, load A covers pairs [w0, w0+64), load B pairs [w0+64, w0+128); output quad
// j of the wave's 64 quads = pairs 2j, 2j+1 -> lane j takes them from lanes (2j)&63, (2j+1)&63 of A (j<32) or B
template <int U, bool NT, int THREADS>
__global__ __launch_bounds__(THREADS) void cast4x_kernel(const d2* __restrict__ in, f4* __restrict__ out, int64_t n4) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t per = (int64_t)THREADS * U;   // quads per block iteration
  for (int64_t base = (int64_t)blockIdx.x * per; base < n4; base += (int64_t)gridDim.x * per) {
    d2 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t q0 = base + (int64_t)(u * (THREADS / 64) + wave) * 64;   // first quad of this wave's step
      a[u] = ld<NT>(in + 2 * q0 + lane); b[u] = ld<NT>(in + 2 * q0 + 64 + lane);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t q0 = base + (int64_t)(u * (THREADS / 64) + wave) * 64;
      const f2 fa = {(float)a[u].x, (float)a[u].y}, fb = {(float)b[u].x, (float)b[u].y};
      const int s0 = (2 * lane) & 63, s1 = (2 * lane + 1) & 63;
      const float ax0 = __shfl(fa.x, s0, 64), ay0 = __shfl(fa.y, s0, 64), ax1 = __shfl(fa.x, s1, 64), ay1 = __shfl(fa.y, s1, 64);
      const float bx0 = __shfl(fb.x, s0, 64), by0 = __shfl(fb.y, s0, 64), bx1 = __shfl(fb.x, s1, 64), by1 = __shfl(fb.y, s1, 64);
      f4 o;
      if (lane < 32) o = f4{ax0, ay0, ax1, ay1}; else o = f4{bx0, by0, bx1, by1};
      st<NT>(out + q0 + lane, o);
    }
  }
}
// greater: per wave and step U x 2 x 64 pairs; ballot per component, words interleaved so that lane pairs stay in order:
// rows of load u: pair p = w0 + lane -> rows 2p, 2p+1.  bit (2*lane + c) of a 128-bit group -> two u64 words
template <int U, bool NT, int THREADS>
__global__ __launch_bounds__(THREADS) void gt_kernel(const d2* __restrict__ l, const d2* __restrict__ r, uint64_t* __restrict__ out, int64_t n2) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t per = (int64_t)THREADS * U;   // pairs per block iteration
  for (int64_t base = (int64_t)blockIdx.x * per; base < n2; base += (int64_t)gridDim.x * per) {
    d2 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p0 = base + (int64_t)(u * (THREADS / 64) + wave) * 64;
      a[u] = ld<NT>(l + p0 + lane); b[u] = ld<NT>(r + p0 + lane);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p0 = base + (int64_t)(u * (THREADS / 64) + wave) * 64;
      const uint64_t mx = __ballot(a[u].x > b[u].x), my = __ballot(a[u].y > b[u].y);
      // interleave: bit 2i = mx bit i, bit 2i+1 = my bit i; lanes 0/1 build the low/high word
      if (lane < 2) {
        uint32_t hx = (uint32_t)(mx >> (32 * lane)), hy = (uint32_t)(my >> (32 * lane));
        uint64_t x = hx, y = hy;
        x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
        x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full; x = (x | (x << 2)) & 0x3333333333333333ull; x = (x | (x << 1)) & 0x5555555555555555ull;
        y = (y | (y << 16)) & 0x0000FFFF0000FFFFull; y = (y | (y << 8)) & 0x00FF00FF00FF00FFull;
        y = (y | (y << 4)) & 0x0F0F0F0F0F0F0F0Full; y = (y | (y << 2)) & 0x3333333333333333ull; y = (y | (y << 1)) & 0x5555555555555555ull;
        out[(p0 >> 5) + lane] = x | (y << 1);
      }
    }
  }
}

enum Op { COPY, READ, CAST2, CAST4S, CAST4X, GT };
static const char* op_name[] = {"copy", "read", "cast2", "cast4s", "cast4x", "gt"};

template <int U, bool NT, int THREADS>
static float launch(Op op, int grid, const void* a, const void* b, void* o, int64_t n8) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    switch (op) {
      case COPY: copy_kernel<U, NT, THREADS><<<grid, THREADS>>>((const f4*)a, (f4*)o, n8 / 2); break;
      case READ: read_kernel<U, NT, THREADS><<<grid, THREADS>>>((const f4*)a, (f4*)o, n8 / 2); break;
      case CAST2: cast2_kernel<U, NT, THREADS><<<grid, THREADS>>>((const d2*)a, (f2*)o, n8 / 2); break;
      case CAST4S: cast4s_kernel<U, NT, THREADS><<<grid, THREADS>>>((const d2*)a, (f4*)o, n8 / 4); break;
      case CAST4X: cast4x_kernel<U, NT, THREADS><<<grid, THREADS>>>((const d2*)a, (f4*)o, n8 / 4); break;
      case GT: gt_kernel<U, NT, THREADS><<<grid, THREADS>>>((const d2*)a, (const d2*)b, (uint64_t*)o, n8 / 2); break;
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best;
}

static double bytes_of(Op op, int64_t n8) {
  switch (op) {
    case COPY: return 16.0 * n8;
    case READ: return 8.0 * n8;
    case CAST2: case CAST4S: case CAST4X: return 12.0 * n8;
    case GT: return 16.125 * n8;
  }
  return 0;
}

template <int U, bool NT, int THREADS>
static void row(Op op, const void* a, const void* b, void* o, int64_t n8) {
  // units per block iteration in the op's own element (see kernels): all are THREADS * U of {16 B, pair, quad}
  const int64_t units = op == CAST4S || op == CAST4X ? n8 / 4 : n8 / 2;
  const int64_t oneshot = (units + (int64_t)THREADS * U - 1) / ((int64_t)THREADS * U);
  for (int64_t grid : {(int64_t)256 * 4 * 256 / THREADS, (int64_t)256 * 8 * 256 / THREADS, (int64_t)256 * 16 * 256 / THREADS, oneshot}) {
    if (grid > oneshot) continue;
    const float ms = launch<U, NT, THREADS>(op, (int)grid, a, b, o, n8);
    printf("%-7s U=%d %-5s threads=%4d grid=%9lld%s  %8.3f ms  %8.1f GB/s\n", op_name[op], U, NT ? "nt" : "plain", THREADS,
           (long long)grid, grid == oneshot ? " (one-shot)" : "           ", ms, bytes_of(op, n8) / ms * 1e-6);
    fflush(stdout);
  }
}

template <int U, bool NT>
static void rows(Op op, const void* a, const void* b, void* o, int64_t n8) {
  row<U, NT, 256>(op, a, b, o, n8);
  row<U, NT, 1024>(op, a, b, o, n8);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 30;
  const int64_t n8 = (int64_t)1 << lg;
  double *a, *b; void* o;
  CK(hipMalloc(&a, n8 * 8)); CK(hipMalloc(&b, n8 * 8)); CK(hipMalloc(&o, n8 * 8));
  fill<<<4096, 256>>>(a, n8, 1); fill<<<4096, 256>>>(b, n8, 2);
  CK(hipDeviceSynchronize());
  printf("elements=2^%d of 8 bytes; bytes counted: copy 16/el, read 8/el, cast 12/el, gt 16.125/el\n", lg);
  for (Op op : {COPY, READ, CAST2, CAST4S, CAST4X, GT}) {
    rows<1, false>(op, a, b, o, n8); rows<2, false>(op, a, b, o, n8); rows<4, false>(op, a, b, o, n8); rows<8, false>(op, a, b, o, n8);
    rows<2, true>(op, a, b, o, n8); rows<4, true>(op, a, b, o, n8);
  }
  // hipMemcpyAsync device-to-device for reference
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0)); CK(hipMemcpyAsync(o, a, n8 * 8, hipMemcpyDeviceToDevice, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("hipMemcpyAsync d2d                                      %8.3f ms  %8.1f GB/s\n", best, 16.0 * n8 / best * 1e-6);
  return 0;
}
