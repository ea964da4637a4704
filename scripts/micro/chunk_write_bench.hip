// What a software-write-combining radix scatter could reach at best: the MEMORY side only.  Every wave reads its
// keys linearly (8 B/row) and writes 12-byte records as aligned chunks of C records to the frontiers of B bins spread
// over the whole output (a chunk = what a workgroup would flush from an LDS staging buffer when a bin's buffer is
// full); the bin of a chunk is pseudo-random, the position comes from one returning atomic per chunk.  No LDS work, no
// digit extraction: an upper bound for the flat level of the sort / group-by partition.
//   usage: chunk_write_bench [log2 rows]      build: hipcc --offload-arch=gfx950 -O3 -o build/chunk_write_bench ...
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct __attribute__((packed, aligned(4))) Rec12 { uint32_t lo, hi, idx; };

__global__ void fill(uint64_t* k, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t z = 0x1234 + (uint64_t)i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    k[i] = z ^ (z >> 27);
  }
}

// one wave = one chunk of C records per step (C multiple of 64: C/64 records per lane)
template <int C>
__global__ __launch_bounds__(256) void chunk_write(const uint64_t* __restrict__ keys, int64_t n, int bins, int64_t bin_rows,
                                                   uint32_t* __restrict__ cursor, Rec12* __restrict__ out, int linear) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t nchunks = n / C;
  for (int64_t c = wave; c < nchunks; c += nwaves) {
    const int64_t row0 = c * C;
    uint64_t k[C / 64];
#pragma unroll
    for (int u = 0; u < C / 64; ++u) k[u] = keys[row0 + u * 64 + lane];
    int64_t base;
    if (linear) {
      base = row0;
    } else {
      const uint32_t bin = (uint32_t)((uint64_t)c * 0x9E3779B97F4A7C15ull >> 40) % (uint32_t)bins;
      uint32_t pos = 0;
      if (lane == 0) pos = atomicAdd(&cursor[bin], (uint32_t)C);
      pos = __shfl(pos, 0, 64);
      base = (int64_t)bin * bin_rows + (pos % (uint32_t)(bin_rows - C + 1)) / C * C;   // stays inside the bin, chunk-aligned
    }
#pragma unroll
    for (int u = 0; u < C / 64; ++u) {
      Rec12 r;
      r.lo = (uint32_t)k[u]; r.hi = (uint32_t)(k[u] >> 32); r.idx = (uint32_t)(row0 + u * 64 + lane);
      out[base + u * 64 + lane] = r;
    }
  }
}

// the same pattern without the atomic (which serialises when few bins share many chunks): chunk c goes to bin c % B
// at slot c / B — B write frontiers advancing in lock step; chunks of C >= 16 records (a wave covers 64 * U rows)
template <int U>
__global__ __launch_bounds__(256) void chunk_write_rr(const uint64_t* __restrict__ keys, int64_t n, int C, int bins,
                                                      int64_t bin_rows, Rec12* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t nsteps = n / (64 * U);
  for (int64_t s = wave; s < nsteps; s += nwaves) {
    const int64_t row0 = s * 64 * U;
    uint64_t k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) k[u] = keys[row0 + u * 64 + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + u * 64 + lane;
      const int64_t chunk = row / C;
      const int64_t dst = (chunk % bins) * bin_rows + (chunk / bins) * C + (row % C);
      Rec12 r;
      r.lo = (uint32_t)k[u]; r.hi = (uint32_t)(k[u] >> 32); r.idx = (uint32_t)row;
      out[dst] = r;
    }
  }
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 30;
  const int64_t n = (int64_t)1 << lg;
  uint64_t* keys; Rec12* out; uint32_t* cursor;
  CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&out, n * 12)); CK(hipMalloc(&cursor, 4096 * 4));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, keys, n);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("rows=2^%d: read 8 B/row linearly, write 12 B/row in aligned chunks of C records to B bin frontiers\n", lg);
  auto run = [&](auto kern, int C, int bins, int linear) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemset(cursor, 0, 4096 * 4));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(8192), dim3(256), 0, 0, keys, n, bins, n / bins, cursor, out, linear);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best = ms < best ? ms : best;
    }
    printf("C=%4d records (%5d B)  bins=%5d %-8s %8.3f ms  %7.1f GB/s\n", C, C * 12, bins, linear ? "linear" : "frontier", best,
           n * 20.0 / best / 1e6);
  };
  for (int bins : {0, 64, 128, 512, 1024}) {
    const int linear = bins == 0;
    const int b = linear ? 1 : bins;
    run(chunk_write<64>, 64, b, linear);
    run(chunk_write<128>, 128, b, linear);
    run(chunk_write<256>, 256, b, linear);
  }
  printf("-- no atomics: chunk c -> bin c %% B, slot c / B (B frontiers advancing together)\n");
  for (int bins : {64, 128, 256, 512, 1024}) {
    for (int C : {8, 16, 32, 64, 128, 256}) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(chunk_write_rr<4>, dim3(8192), dim3(256), 0, 0, keys, n, C, bins, n / bins, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = ms < best ? ms : best;
      }
      printf("C=%4d records (%5d B)  bins=%5d round-robin %8.3f ms  %7.1f GB/s\n", C, C * 12, bins, best, n * 20.0 / best / 1e6);
    }
  }
  return 0;
}
