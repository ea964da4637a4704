// Does data written by one kernel come back cheaper to the next one when the chunk fits the 256 MB memory-side cache
// (MALL / Infinity Cache)?  The sort's level 2 writes 12-byte records the finish kernel reads right afterwards, the
// group-by's scatter writes what the aggregate reads: if a chunk-sized produce -> consume pair runs faster than
// "produce everything, then consume everything", running those passes chunk by chunk is worth building.
//   produce: copy  src[chunk] -> tmp[chunk]      (16 B moved per 8 B element)
//   consume: read  tmp[chunk] -> xor-reduce      (8 B per element)
// whole = produce(all) ; consume(all).  chunked(S) = for each chunk of S bytes: produce ; consume  (one stream, or
// two streams so that chunk i's consume overlaps chunk i+1's produce).
//   usage: mall_bench [log2 total bytes = 33]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void fill(f4* a, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = (float)(i & 1023);
    a[i] = f4{x, x + 1, x + 2, x + 3};
  }
}
template <int U>
__global__ __launch_bounds__(256) void produce(const f4* __restrict__ in, f4* __restrict__ out, int64_t n16) {
  const int64_t base = (int64_t)blockIdx.x * 256 * U;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const int64_t i = base + u * 256 + threadIdx.x; if (i < n16) v[u] = in[i]; }
#pragma unroll
  for (int u = 0; u < U; ++u) { const int64_t i = base + u * 256 + threadIdx.x; if (i < n16) out[i] = v[u]; }
}
template <int U>
__global__ __launch_bounds__(256) void consume(const f4* __restrict__ in, f4* __restrict__ sink, int64_t n16) {
  const int64_t base = (int64_t)blockIdx.x * 256 * U;
  f4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < U; ++u) { const int64_t i = base + u * 256 + threadIdx.x; if (i < n16) acc += in[i]; }
  if (acc.x == 1.2345f) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 33;
  const int64_t total = 1ll << lg, n16 = total / 16;
  constexpr int U = 4;
  f4 *src, *tmp, *sink;
  CK(hipMalloc(&src, total)); CK(hipMalloc(&tmp, total)); CK(hipMalloc(&sink, 64));
  fill<<<4096, 256>>>(src, n16);
  CK(hipDeviceSynchronize());
  hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto grid = [&](int64_t n) { return (unsigned)((n + 256 * U - 1) / (256 * U)); };
  printf("total %lld MB per pass (produce moves 2x that, consume 1x)\n", (long long)(total >> 20));
  for (int rep = 0; rep < 2; ++rep) {
    // whole
    {
      float best = 1e9f, best_p = 0, best_c = 0;
      for (int it = 0; it < 3; ++it) {
        hipEvent_t m; CK(hipEventCreate(&m));
        CK(hipEventRecord(e0, s0));
        produce<U><<<grid(n16), 256, 0, s0>>>(src, tmp, n16);
        CK(hipEventRecord(m, s0));
        consume<U><<<grid(n16), 256, 0, s0>>>(tmp, sink, n16);
        CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
        float ms, p; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&p, e0, m));
        if (ms < best) { best = ms; best_p = p; best_c = ms - p; }
        CK(hipEventDestroy(m));
      }
      printf("whole                         %8.3f ms  (produce %.3f = %.0f GB/s, consume %.3f = %.0f GB/s)\n", best, best_p,
             2.0 * total / best_p * 1e-6, best_c, 1.0 * total / best_c * 1e-6);
    }
    for (int64_t chunk_mb : {16, 32, 64, 96, 128, 192, 256, 512, 1024}) {
      const int64_t c16 = (chunk_mb << 20) / 16;
      if (c16 > n16) continue;
      for (int two = 0; two < 2; ++two) {
        float best = 1e9f;
        for (int it = 0; it < 3; ++it) {
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0, s0));
          if (two) CK(hipStreamWaitEvent(s1, e0, 0));
          int k = 0;
          hipEvent_t done[2]; CK(hipEventCreate(&done[0])); CK(hipEventCreate(&done[1]));
          for (int64_t at = 0; at < n16; at += c16, ++k) {
            const int64_t n = (n16 - at < c16) ? n16 - at : c16;
            produce<U><<<grid(n), 256, 0, s0>>>(src + at, tmp + at, n);
            if (two) {
              CK(hipEventRecord(done[k & 1], s0));
              CK(hipStreamWaitEvent(s1, done[k & 1], 0));
              consume<U><<<grid(n), 256, 0, s1>>>(tmp + at, sink, n);
            } else {
              consume<U><<<grid(n), 256, 0, s0>>>(tmp + at, sink, n);
            }
          }
          if (two) { CK(hipEventRecord(done[0], s1)); CK(hipStreamWaitEvent(s0, done[0], 0)); }
          CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
          CK(hipEventDestroy(done[0])); CK(hipEventDestroy(done[1]));
        }
        printf("chunks of %5lld MB  %s   %8.3f ms   %.0f GB/s over the 3x bytes\n", (long long)chunk_mb, two ? "two streams" : "one stream ",
               best, 3.0 * total / best * 1e-6);
      }
    }
  }
  return 0;
}
