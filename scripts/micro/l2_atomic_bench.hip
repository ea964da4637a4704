// The load-bearing assumption of the group-by design: "global atomics run at ~12 Grows/s whatever the
// table size, so aggregate in LDS behind two partition levels".  That number was measured once with returning
// device-scope atomics.  This bench re-measures the aggregation step in the forms a ONE-level partition would use:
//   dev    device(agent)-scope atomics (sc1: executed behind the L2, at the fabric) into a table of T bytes
//   l2     workgroup-scope atomics (no sc1: executed IN the XCD's L2) into a table slice that only workgroups of
//          ONE XCD touch — the workgroup reads its XCC_ID and takes tiles from that XCD's queue, so correctness
//          does not depend on the dispatcher's block->XCD map
// each with {sum u64 + count u32} per row (what hash_sum needs), sum only, returning / non-returning, and with keys
// + values read linearly from HBM (12 B/row, the real aggregate pass) or generated in registers (pure atomic rate).
//   usage: l2_atomic_bench [log2 rows]    build: hipcc --offload-arch=gfx950 -O3 -o build/l2_atomic_bench ...
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void fill(uint32_t* k, int64_t* v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t z = mix(0x1234 + (uint64_t)i * 0x9E3779B97F4A7C15ull);
    k[i] = (uint32_t)(z >> 32);
    v[i] = (int64_t)mix(z);
  }
}

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

struct Slot { unsigned long long sum; uint32_t count; uint32_t tag; };   // 16 B: both atomics of a row in one line

// MODE bit0: 1 = workgroup scope (L2), 0 = agent scope; bit1: returning; bit2: sum only; bit3: keys from registers
template <int MODE>
__global__ __launch_bounds__(256) void agg(const uint32_t* __restrict__ keys, const int64_t* __restrict__ vals, int64_t n,
                                           Slot* __restrict__ table, uint32_t slots_per_part, int parts_per_xcd,
                                           uint32_t* __restrict__ queue, int tile_rows, unsigned long long* sink) {
  constexpr bool L2 = MODE & 1, RET = MODE & 2, SUMONLY = MODE & 4, REGKEYS = MODE & 8;
  __shared__ int64_t s_tile;
  const uint32_t xcd = L2 ? xcc_id() : (blockIdx.x & 7);
  const int64_t ntiles = n / tile_rows;            // tiles are dealt to XCD (tile % 8)
  const int64_t tiles_x = ntiles / 8;
  unsigned long long acc = 0;
  for (;;) {
    if (threadIdx.x == 0) s_tile = atomicAdd(&queue[xcd * 32], 1u);
    __syncthreads();
    const int64_t t = s_tile;
    __syncthreads();
    if (t >= tiles_x) break;
    // the tile's rows all belong to one "partition" of this XCD: a table slice of slots_per_part slots
    const uint32_t part = (uint32_t)(t % parts_per_xcd);
    Slot* tab = table + ((size_t)xcd * parts_per_xcd + part) * slots_per_part;
    const int64_t row0 = (t * 8 + xcd) * (int64_t)tile_rows;
    for (int i = threadIdx.x; i < tile_rows; i += 256 * 4) {
      uint32_t k[4]; int64_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = row0 + i + u * 256;
        if (REGKEYS) { const uint64_t z = mix((uint64_t)r * 0x9E3779B97F4A7C15ull); k[u] = (uint32_t)(z >> 32); v[u] = (int64_t)z; }
        else { k[u] = keys[r]; v[u] = vals[r]; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        Slot* s = tab + (uint32_t)(((uint64_t)k[u] * slots_per_part) >> 32);
        if (L2) {
          if (RET) {
            acc += __hip_atomic_fetch_add(&s->sum, (unsigned long long)v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!SUMONLY) acc += __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            __hip_atomic_fetch_add(&s->sum, (unsigned long long)v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!SUMONLY) __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        } else {
          if (RET) {
            acc += __hip_atomic_fetch_add(&s->sum, (unsigned long long)v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!SUMONLY) acc += __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            __hip_atomic_fetch_add(&s->sum, (unsigned long long)v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!SUMONLY) __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
  }
  if (RET && acc == 0x123456789abcull) *sink = acc;
}

// which XCD does block b land on, and how many blocks of a persistent grid does each XCD get
__global__ void census(uint32_t* per_xcd, uint32_t* mismatch) {
  if (threadIdx.x == 0) {
    const uint32_t x = xcc_id();
    atomicAdd(&per_xcd[x], 1u);
    if (x != (blockIdx.x & 7)) atomicAdd(mismatch, 1u);
  }
}

__global__ void check(const Slot* table, size_t nslots, unsigned long long* out) {
  unsigned long long s = 0, c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (size_t)gridDim.x * blockDim.x) {
    s += table[i].sum; c += table[i].count;
  }
  atomicAdd(&out[0], s); atomicAdd(&out[1], c);
}

template <int MODE>
static void run(const char* name, const uint32_t* keys, const int64_t* vals, int64_t n, Slot* table, size_t table_slots_total,
                uint32_t slots_per_part, int parts_per_xcd, uint32_t* queue, unsigned long long* sink, int grid,
                unsigned long long want_sum) {
  const int tile_rows = 8192;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(queue, 0, 8 * 32 * 4));
    CK(hipMemset(table, 0, table_slots_total * sizeof(Slot)));
    CK(hipEventRecord(e0));
    agg<MODE><<<grid, 256>>>(keys, vals, n, table, slots_per_part, parts_per_xcd, queue, tile_rows, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  unsigned long long* chk; CK(hipMalloc(&chk, 16)); CK(hipMemset(chk, 0, 16));
  check<<<1024, 256>>>(table, table_slots_total, chk);
  unsigned long long h[2]; CK(hipMemcpy(h, chk, 16, hipMemcpyDeviceToHost)); CK(hipFree(chk));
  const int64_t rows = n / tile_rows / 8 * 8 * tile_rows;
  const bool sumonly = MODE & 4, regk = MODE & 8;
  const char* ok = regk ? "-" : ((h[0] == want_sum && (sumonly || h[1] == (unsigned long long)rows)) ? "ok" : "BAD");
  printf("%-44s part=%8.2f MB x%3d/xcd grid=%5d  %8.3f ms  %8.2f Grows/s  %s\n", name,
         slots_per_part * 16.0 / 1048576.0, parts_per_xcd, grid, best, rows / best * 1e-6, ok);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 28;
  const int64_t n = (int64_t)1 << lg;
  uint32_t* keys; int64_t* vals; CK(hipMalloc(&keys, n * 4)); CK(hipMalloc(&vals, n * 8));
  fill<<<4096, 256>>>(keys, vals, n);
  uint32_t* queue; CK(hipMalloc(&queue, 8 * 32 * 4));
  unsigned long long* sink; CK(hipMalloc(&sink, 8));
  uint32_t* cen; CK(hipMalloc(&cen, 64)); CK(hipMemset(cen, 0, 64));
  census<<<2048, 256>>>(cen, cen + 8);
  uint32_t hc[9]; CK(hipMemcpy(hc, cen, 36, hipMemcpyDeviceToHost));
  printf("census of a 2048-block grid: per XCD");
  for (int i = 0; i < 8; ++i) printf(" %u", hc[i]);
  printf("; blocks whose XCC_ID != blockIdx %% 8: %u\n", hc[8]);
  // reference checksum of all values of whole tiles
  std::vector<int64_t> hv((size_t)n);
  CK(hipMemcpy(hv.data(), vals, n * 8, hipMemcpyDeviceToHost));
  unsigned long long want = 0;
  const int64_t rows = n / 8192 / 8 * 8 * 8192;
  for (int64_t i = 0; i < rows; ++i) want += (unsigned long long)hv[i];
  hv.clear(); hv.shrink_to_fit();

  const size_t max_slots = (size_t)1 << 26;     // 1 GiB of 16-byte slots
  Slot* table; CK(hipMalloc(&table, max_slots * sizeof(Slot)));
  printf("rows=2^%d; one row = key u32 + value i64; slot = {sum u64, count u32, tag u32}\n", lg);
  // per-partition table sizes: 64 KB ... 4 MB (L2 = 4 MB per XCD); parts_per_xcd chosen so the whole table is bounded
  const uint32_t part_slots[] = {4096, 16384, 65536, 131072, 262144};
  for (int grid : {2048, 1024}) {
    for (uint32_t ps : part_slots) {
      const int ppx = 32;      // 32 partitions per XCD, tiles visit them round-robin (worst case: no temporal locality)
      const size_t total = (size_t)8 * ppx * ps;
      run<1>("l2  nonret sum+count  hbm rows", keys, vals, n, table, total, ps, ppx, queue, sink, grid, want);
      run<5>("l2  nonret sum only   hbm rows", keys, vals, n, table, total, ps, ppx, queue, sink, grid, want);
      run<3>("l2  return sum+count  hbm rows", keys, vals, n, table, total, ps, ppx, queue, sink, grid, want);
      run<9>("l2  nonret sum+count  reg rows", keys, vals, n, table, total, ps, ppx, queue, sink, grid, want);
      run<0>("dev nonret sum+count  hbm rows", keys, vals, n, table, total, ps, ppx, queue, sink, grid, want);
    }
  }
  // one partition per XCD at a time (what a partition-ordered aggregate gives: all tiles of a partition are consecutive)
  for (uint32_t ps : part_slots) {
    const size_t total = (size_t)8 * ps;
    run<1>("l2  nonret sum+count  hbm rows  1 part", keys, vals, n, table, total, ps, 1, queue, sink, 2048, want);
    run<9>("l2  nonret sum+count  reg rows  1 part", keys, vals, n, table, total, ps, 1, queue, sink, 2048, want);
    run<0>("dev nonret sum+count  hbm rows  1 part", keys, vals, n, table, total, ps, 1, queue, sink, 2048, want);
    run<2>("dev return sum+count  hbm rows  1 part", keys, vals, n, table, total, ps, 1, queue, sink, 2048, want);
  }
  // the old experiment's shape: one big table, device scope
  for (size_t slots : {(size_t)1 << 16, (size_t)1 << 20, (size_t)1 << 24, (size_t)1 << 26}) {
    run<0>("dev nonret sum+count  hbm rows  whole table", keys, vals, n, table, slots, (uint32_t)(slots / 8), 1, queue, sink, 2048, want);
    run<2>("dev return sum+count  hbm rows  whole table", keys, vals, n, table, slots, (uint32_t)(slots / 8), 1, queue, sink, 2048, want);
  }
  return 0;
}
