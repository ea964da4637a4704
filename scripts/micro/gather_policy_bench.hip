// take's gather: 8-byte values[idx[i]] for 1e8 indices into 2^30 rows.  Monotonic indices at 10 % density touch 81 % of
// the column's 128-B lines for one value each (FETCH 7.07 GB for 0.8 GB of values).  Does any cache
// policy of the load make the miss fetch a sector (32 / 64 B) instead of the line?  One kernel per policy (distinct
// names: a FETCH_SIZE pass of rocprofv3 attributes the bytes), monotonic and random indices, U = 8 gathers per lane.
//   policies: plain | nt | sc0 | sc1 | sc0 sc1 | sc0 nt | sc1 nt | sc0 sc1 nt, as written in the instruction
//   usage: gather_policy_bench [log2 rows = 30] [indices = 100000000] [both|mono|random]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void fill_values(uint64_t* v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (uint64_t)i * 0x9E3779B97F4A7C15ull;
}
// monotonic: index j = j * stride + hash(j) % stride (one per `stride` rows); random: hash(j) % n
__global__ void fill_indices(uint32_t* idx, int64_t m, int64_t n, int mono) {
  const int64_t stride = n / m;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
    uint64_t z = (uint64_t)j * 0xBF58476D1CE4E5B9ull + 12345;
    z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
    idx[j] = mono ? (uint32_t)(j * stride + (int64_t)(z % (uint64_t)stride)) : (uint32_t)(z % (uint64_t)n);
  }
}

template <int P> __device__ __forceinline__ uint64_t gather8(const uint64_t* p) {
  uint64_t v;
  if constexpr (P == 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if constexpr (P == 1) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if constexpr (P == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  if constexpr (P == 3) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if constexpr (P == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  if constexpr (P == 5) asm volatile("global_load_dwordx2 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
  if constexpr (P == 6) asm volatile("global_load_dwordx2 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
  if constexpr (P == 7) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int P>
__global__ __launch_bounds__(256) void gather_policy(const uint64_t* __restrict__ values, const uint32_t* __restrict__ idx,
                                                     int64_t m, uint64_t* __restrict__ out) {
  constexpr int U = 8;
  const int64_t per = 256 * U;
  for (int64_t base = (int64_t)blockIdx.x * per; base < m; base += (int64_t)gridDim.x * per) {
    uint32_t ix[U];
    uint64_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = base + u * 256 + threadIdx.x;
      ix[u] = idx[j < m ? j : m - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = gather8<P>(values + ix[u]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = base + u * 256 + threadIdx.x;
      if (j < m) out[j] = v[u];
    }
  }
}

template <int P>
static void run(const char* what, const char* pol, const uint64_t* values, const uint32_t* idx, int64_t m, uint64_t* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 32;
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((gather_policy<P>), dim3(grid), dim3(256), 0, 0, values, idx, m, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  uint64_t probe[2];
  CK(hipMemcpy(probe, out + m / 2, 16, hipMemcpyDeviceToHost));
  printf("%-9s %-10s %8.3f ms  %6.1f G gathers/s   probe %016llx\n", what, pol, best, m / best / 1e6, (unsigned long long)(probe[0] ^ probe[1]));
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 30;
  const int64_t n = int64_t(1) << lg;
  const int64_t m = argc > 2 ? atoll(argv[2]) : 100000000;
  uint64_t *values, *out;
  uint32_t* idx;
  CK(hipMalloc(&values, n * 8)); CK(hipMalloc(&out, m * 8)); CK(hipMalloc(&idx, m * 4));
  hipLaunchKernelGGL(fill_values, dim3(4096), dim3(256), 0, 0, values, n);
  const char* which = argc > 3 ? argv[3] : "both";
  for (int mono = 1; mono >= 0; --mono) {
    if ((mono && which[0] == 'r') || (!mono && which[0] == 'm')) continue;
    hipLaunchKernelGGL(fill_indices, dim3(4096), dim3(256), 0, 0, idx, m, n, mono);
    CK(hipDeviceSynchronize());
    const char* what = mono ? "monotonic" : "random";
    run<0>(what, "plain", values, idx, m, out);
    run<1>(what, "nt", values, idx, m, out);
    run<2>(what, "sc0", values, idx, m, out);
    run<3>(what, "sc1", values, idx, m, out);
    run<4>(what, "sc0_sc1", values, idx, m, out);
    run<5>(what, "sc0_nt", values, idx, m, out);
    run<6>(what, "sc1_nt", values, idx, m, out);
    run<7>(what, "sc0sc1nt", values, idx, m, out);
  }
  return 0;
}
