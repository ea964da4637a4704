// Could a SWEEPING filter for 8-byte values — every value read coalesced (16 B per lane), output positions from two
// ballots per 128 rows, selected values stored straight to the wave's contiguous output range (no LDS, the L2 combines
// the partial lines) — beat the gather form at 25 / 50 % selectivity?  (the product's sweeping form goes
// through an LDS ring and loses to the gather form at every selectivity.)
// One wave per 4096-row tile as in selection.hip; per-tile output offsets precomputed (the product's count pass).
//   variants: direct  = two predicated 8-byte stores per lane per step
//             nt      = the same with non-temporal loads
//   usage: sweep_compact_bench [log2 rows = 30]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ void fill_values(uint64_t* v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (uint64_t)i * 0x9E3779B97F4A7C15ull + 1;
}
__global__ void fill_mask(uint64_t* m, int64_t nwords, uint32_t threshold) {   // bit set with probability threshold / 2^32
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * blockDim.x) {
    uint64_t word = 0;
    for (int b = 0; b < 64; ++b) {
      uint64_t z = (uint64_t)(w * 64 + b) * 0xBF58476D1CE4E5B9ull + 777;
      z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
      if ((uint32_t)z < threshold) word |= 1ull << b;
    }
    m[w] = word;
  }
}
// per tile (64 words = 4096 rows): number of set bits; then an exclusive scan on the host side of the bench (thrust-free:
// a single-block scan kernel over up to 2^18 tiles)
__global__ void tile_counts(const uint64_t* m, int64_t ntiles, uint32_t* counts) {
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ntiles) return;
  uint32_t k = __popcll(m[t * 64 + (threadIdx.x & 63)]);
  for (int d = 32; d >= 1; d >>= 1) k += __shfl_xor(k, d, 64);
  if ((threadIdx.x & 63) == 0) counts[t] = k;
}
__global__ void scan_tiles(const uint32_t* counts, int64_t ntiles, int64_t* offs) {   // one block of 1024, serial chunks
  __shared__ int64_t part[1024];
  const int64_t per = (ntiles + 1023) / 1024;
  const int64_t b = threadIdx.x * per, e = b + per < ntiles ? b + per : ntiles;
  int64_t s = 0;
  for (int64_t i = b; i < e; ++i) s += counts[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { int64_t run = 0; for (int i = 0; i < 1024; ++i) { const int64_t c = part[i]; part[i] = run; run += c; } }
  __syncthreads();
  int64_t run = part[threadIdx.x];
  for (int64_t i = b; i < e; ++i) { offs[i] = run; run += counts[i]; }
  if (threadIdx.x == 1023) offs[ntiles] = run;
}

template <bool NT, int U>
__global__ __launch_bounds__(256) void sweep_compact(const uint64_t* __restrict__ values, const uint64_t* __restrict__ mask,
                                                     const int64_t* __restrict__ offs, int64_t ntiles, uint64_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ntiles) return;
  const uint64_t Ew = mask[t * 64 + lane];
  int64_t pos = offs[t];
  const u4* __restrict__ v16 = reinterpret_cast<const u4*>(values + t * 4096);
  const uint64_t lt = (1ull << lane) - 1;
  for (int it0 = 0; it0 < 32; it0 += U) {     // 128 rows per step: lane holds rows 2 * lane, 2 * lane + 1 of the step
    u4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u4* p = v16 + (it0 + u) * 64 + lane;
      q[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = it0 + u;
      const uint64_t word = __shfl(Ew, it * 2 + (lane >> 5), 64);
      const uint32_t bits = (uint32_t)(word >> ((lane & 31) * 2)) & 3u;
      const uint64_t b0 = __ballot(bits & 1u), b1 = __ballot(bits & 2u);
      const int r0 = __popcll(b0 & lt) + __popcll(b1 & lt);
      if (bits & 1u) out[pos + r0] = ((uint64_t)q[u].y << 32) | q[u].x;
      if (bits & 2u) out[pos + r0 + (bits & 1u)] = ((uint64_t)q[u].w << 32) | q[u].z;
      pos += __popcll(b0) + __popcll(b1);
    }
  }
}

template <bool NT, int U>
static void run(const char* name, const uint64_t* values, const uint64_t* mask, const int64_t* offs, int64_t ntiles, uint64_t* out,
                int64_t n, int64_t selected) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((sweep_compact<NT, U>), dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, 0, values, mask, offs, ntiles, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  uint64_t probe[3];
  CK(hipMemcpy(probe, out, 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(probe + 1, out + selected / 2, 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(probe + 2, out + selected - 1, 8, hipMemcpyDeviceToHost));
  const double bytes = (double)n * 8 + (double)n / 8 + (double)selected * 8;
  printf("  %-14s U=%d  %7.3f ms   %6.1f GB/s of (values + mask + output)   probe %016llx\n", name, U, best, bytes / best / 1e6,
         (unsigned long long)(probe[0] ^ probe[1] ^ probe[2]));
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 30;
  const int64_t n = int64_t(1) << lg, nwords = n / 64, ntiles = n / 4096;
  uint64_t *values, *mask, *out;
  uint32_t* counts;
  int64_t* offs;
  CK(hipMalloc(&values, n * 8)); CK(hipMalloc(&mask, nwords * 8)); CK(hipMalloc(&out, n * 8));
  CK(hipMalloc(&counts, ntiles * 4)); CK(hipMalloc(&offs, (ntiles + 1) * 8));
  hipLaunchKernelGGL(fill_values, dim3(4096), dim3(256), 0, 0, values, n);
  for (double sel : {0.10, 0.25, 0.50, 1.0}) {
    const uint32_t thr = sel >= 1.0 ? 0xFFFFFFFFu : (uint32_t)(sel * 4294967296.0);
    hipLaunchKernelGGL(fill_mask, dim3(4096), dim3(256), 0, 0, mask, nwords, thr);
    hipLaunchKernelGGL(tile_counts, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, 0, mask, ntiles, counts);
    hipLaunchKernelGGL(scan_tiles, dim3(1), dim3(1024), 0, 0, counts, ntiles, offs);
    int64_t selected = 0;
    CK(hipMemcpy(&selected, offs + ntiles, 8, hipMemcpyDeviceToHost));
    printf("selectivity %.2f: %lld of %lld rows\n", sel, (long long)selected, (long long)n);
    run<false, 4>("direct", values, mask, offs, ntiles, out, n, selected);
    run<true, 4>("nt loads", values, mask, offs, ntiles, out, n, selected);
    run<true, 8>("nt loads", values, mask, offs, ntiles, out, n, selected);
    run<true, 2>("nt loads", values, mask, offs, ntiles, out, n, selected);
  }
  return 0;
}
