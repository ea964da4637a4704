// Can ONE flat radix level replace the group-by's two?  The two-level plan exists because a flat scatter
// into 2048+ bins writes runs of 2-4 records per (tile, bin), and those partial 128-byte lines are shared by workgroups
// on all eight XCDs: every L2 holds a fragment of every frontier line until eviction.  This bench gives every XCD its
// OWN frontier per bin — cursor[(blockIdx & 7)][bin], exact offsets from a histogram with the same block -> class rule —
// so that (with the dispatcher's observed block b -> XCD b % 8 placement; speed only, never correctness) a frontier
// line fills inside ONE write-back L2 whatever the run length.  Records are the group-by's: key u32 + value i64 in,
// 12-byte {key, value} out.
//   forms : tile   = stage the tile in LDS, reorder by bin, write runs      (what gbp_scatter1g does)
//           direct = every row straight to its slot, no reorder (L2 merges)
//   bins  : 256, 2048, 8192         cursors : shared by all blocks | private per (blockIdx & 7)
//   usage: wide_scatter_bench [log2 rows=28]     build: hipcc --offload-arch=gfx950 -O3 -o build/wide_scatter_bench ...
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct __attribute__((packed, aligned(4))) Rec12 { uint32_t key, vlo, vhi; };

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
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { uint32_t n = __shfl_up(v, d, 64); if (lane >= d) v += n; }
  return v;
}

// counts[cls][bin], cls = blockIdx & (ncls - 1); block b covers tiles [b * tpb, (b + 1) * tpb)
template <int LOGB>
__global__ __launch_bounds__(1024) void hist_kernel(const uint32_t* __restrict__ keys, int64_t n, int64_t rows_per_block,
                                                    int ncls, uint32_t* __restrict__ counts) {
  constexpr int BINS = 1 << LOGB;
  __shared__ uint32_t s[BINS];
  for (int b = threadIdx.x; b < BINS; b += 1024) s[b] = 0;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  for (int64_t r = r0 + threadIdx.x; r < r1; r += 1024) atomicAdd(&s[keys[r] >> (32 - LOGB)], 1u);
  __syncthreads();
  const int cls = blockIdx.x & (ncls - 1);
  for (int b = threadIdx.x; b < BINS; b += 1024) if (s[b]) atomicAdd(&counts[(size_t)cls * BINS + b], s[b]);
}

// starts[cls][bin] laid out bin-major: bin 0 {cls 0..}, bin 1 ... ; one block, serial per thread chunk (tiny)
__global__ void scan_kernel(const uint32_t* counts, uint32_t* cursor, uint32_t* bin_start, int bins, int ncls) {
  if (threadIdx.x || blockIdx.x) return;
  uint32_t run = 0;
  for (int b = 0; b < bins; ++b) {
    bin_start[b] = run;
    for (int c = 0; c < ncls; ++c) { cursor[(size_t)c * bins + b] = run; run += counts[(size_t)c * bins + b]; }
  }
  bin_start[bins] = run;
}

template <int LOGB, int RPT, bool DIRECT>
__global__ __launch_bounds__(1024) void scatter_kernel(const uint32_t* __restrict__ keys, const int64_t* __restrict__ vals,
                                                       int64_t n, int tiles_per_block, int ncls, uint32_t* __restrict__ cursor,
                                                       Rec12* __restrict__ out) {
  constexpr int BINS = 1 << LOGB, THREADS = 1024, TILE = THREADS * RPT;
  constexpr int BPT = (BINS + THREADS - 1) / THREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* s_cnt = (uint32_t*)smem;                 // counts, then local starts
  uint32_t* s_gbase = s_cnt + BINS;
  uint32_t* s_wtot = s_gbase + BINS;                 // 16
  uint64_t* s_val = (uint64_t*)(s_wtot + 16);        // TILE   (tile form only)
  uint32_t* s_key = (uint32_t*)(s_val + (DIRECT ? 0 : TILE));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cls = blockIdx.x & (ncls - 1);
  uint32_t* cur = cursor + (size_t)cls * BINS;
  for (int t = 0; t < tiles_per_block; ++t) {
    const int64_t row0 = ((int64_t)blockIdx.x * tiles_per_block + t) * TILE;
    if (row0 >= n) break;
    const int nrows = (int)(n - row0 < TILE ? n - row0 : TILE);
    for (int b = tid; b < BINS; b += THREADS) s_cnt[b] = 0;
    uint32_t key[RPT]; int64_t val[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int p = i * THREADS + tid;
      const int64_t r = row0 + (p < nrows ? p : nrows - 1);
      key[i] = keys[r]; val[i] = vals[r];
    }
    __syncthreads();
    uint32_t rank[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int p = i * THREADS + tid;
      rank[i] = p < nrows ? atomicAdd(&s_cnt[key[i] >> (32 - LOGB)], 1u) : 0u;
    }
    __syncthreads();
    uint32_t c[BPT], mine = 0;
#pragma unroll
    for (int k = 0; k < BPT; ++k) { const int b = tid * BPT + k; c[k] = b < BINS ? s_cnt[b] : 0u; mine += c[k]; }
    const uint32_t incl = wave_incl_scan(mine);
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - mine;
    for (int k = 0; k < wave; ++k) pre += s_wtot[k];
#pragma unroll
    for (int k = 0; k < BPT; ++k) {
      const int b = tid * BPT + k;
      if (b < BINS) {
        s_cnt[b] = pre;                                // local start of bin b inside the tile
        s_gbase[b] = c[k] ? atomicAdd(&cur[b], c[k]) : 0u;
      }
      pre += c[k];
    }
    __syncthreads();
    if (DIRECT) {
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const int p = i * THREADS + tid;
        if (p < nrows) {
          Rec12 r; r.key = key[i]; r.vlo = (uint32_t)val[i]; r.vhi = (uint32_t)((uint64_t)val[i] >> 32);
          out[s_gbase[key[i] >> (32 - LOGB)] + rank[i]] = r;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const int p = i * THREADS + tid;
        if (p < nrows) {
          const uint32_t pos = s_cnt[key[i] >> (32 - LOGB)] + rank[i];
          s_key[pos] = key[i]; s_val[pos] = (uint64_t)val[i];
        }
      }
      __syncthreads();
      for (int p = tid; p < nrows; p += THREADS) {
        const uint32_t k = s_key[p];
        const uint32_t d = k >> (32 - LOGB);
        const uint64_t v = s_val[p];
        Rec12 r; r.key = k; r.vlo = (uint32_t)v; r.vhi = (uint32_t)(v >> 32);
        out[s_gbase[d] + ((uint32_t)p - s_cnt[d])] = r;
      }
    }
    __syncthreads();
  }
}


// Register-staged big tiles: a thread keeps RPT rows in registers, the tile (THREADS * RPT rows, up to 32K) is ranked
// with LDS atomics as before, but the reorder goes through an LDS buffer of TILE / ROUNDS records in ROUNDS rounds —
// so a (tile, bin) run is ROUNDS times longer than what one LDS-resident tile gives.  NT: non-temporal input loads
// (the input stream should not push half-filled frontier lines out of the L2).
template <int LOGB, int THREADS, int RPT, int ROUNDS, bool NT>
__global__ __launch_bounds__(THREADS) void scatter_big_kernel(const uint32_t* __restrict__ keys, const int64_t* __restrict__ vals,
                                                              int64_t n, int ncls, uint32_t* __restrict__ cursor,
                                                              Rec12* __restrict__ out) {
  constexpr int BINS = 1 << LOGB, TILE = THREADS * RPT, CHUNK = TILE / ROUNDS;
  constexpr int BPT = (BINS + THREADS - 1) / THREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* s_cnt = (uint32_t*)smem;
  uint32_t* s_gbase = s_cnt + BINS;
  uint32_t* s_wtot = s_gbase + BINS;
  uint64_t* s_val = (uint64_t*)(s_wtot + 16);
  uint32_t* s_key = (uint32_t*)(s_val + CHUNK);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cls = blockIdx.x & (ncls - 1);
  uint32_t* cur = cursor + (size_t)cls * BINS;
  const int64_t row0 = (int64_t)blockIdx.x * TILE;
  if (row0 >= n) return;
  const int nrows = (int)(n - row0 < TILE ? n - row0 : TILE);
  for (int b = tid; b < BINS; b += THREADS) s_cnt[b] = 0;
  uint32_t key[RPT]; int64_t val[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int p = i * THREADS + tid;
    const int64_t r = row0 + (p < nrows ? p : nrows - 1);
    if (NT) { key[i] = __builtin_nontemporal_load(keys + r); val[i] = __builtin_nontemporal_load(vals + r); }
    else { key[i] = keys[r]; val[i] = vals[r]; }
  }
  __syncthreads();
  uint32_t pos[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int p = i * THREADS + tid;
    pos[i] = p < nrows ? atomicAdd(&s_cnt[key[i] >> (32 - LOGB)], 1u) : 0u;
  }
  __syncthreads();
  uint32_t c[BPT], mine = 0;
#pragma unroll
  for (int k = 0; k < BPT; ++k) { const int b = tid * BPT + k; c[k] = b < BINS ? s_cnt[b] : 0u; mine += c[k]; }
  const uint32_t incl = wave_incl_scan(mine);
  if (lane == 63) s_wtot[wave] = incl;
  __syncthreads();
  uint32_t pre = incl - mine;
  for (int k = 0; k < wave; ++k) pre += s_wtot[k];
#pragma unroll
  for (int k = 0; k < BPT; ++k) {
    const int b = tid * BPT + k;
    if (b < BINS) {
      s_cnt[b] = pre;
      s_gbase[b] = c[k] ? atomicAdd(&cur[b], c[k]) : 0u;
    }
    pre += c[k];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int p = i * THREADS + tid;
    pos[i] = p < nrows ? s_cnt[key[i] >> (32 - LOGB)] + pos[i] : 0xFFFFFFFFu;
  }
  for (int r = 0; r < ROUNDS; ++r) {
    const uint32_t lo = (uint32_t)r * CHUNK;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const uint32_t q = pos[i] - lo;
      if (q < (uint32_t)CHUNK) { s_key[q] = key[i]; s_val[q] = (uint64_t)val[i]; }
    }
    __syncthreads();
    const int cnt = nrows - (int)lo < CHUNK ? nrows - (int)lo : CHUNK;
    for (int p = tid; p < cnt; p += THREADS) {
      const uint32_t k = s_key[p];
      const uint32_t d = k >> (32 - LOGB);
      const uint64_t v = s_val[p];
      Rec12 rr; rr.key = k; rr.vlo = (uint32_t)v; rr.vhi = (uint32_t)(v >> 32);
      out[s_gbase[d] + (lo + (uint32_t)p - s_cnt[d])] = rr;
    }
    __syncthreads();
  }
}

__global__ void verify_kernel(const Rec12* out, const uint32_t* bin_start, int logb, int64_t n, unsigned long long* res) {
  unsigned long long bad = 0, sum = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const Rec12 r = out[i];
    const uint32_t d = r.key >> (32 - logb);
    if (i < bin_start[d] || i >= bin_start[d + 1]) ++bad;
    sum += ((unsigned long long)r.vhi << 32 | r.vlo) + r.key;
  }
  atomicAdd(&res[0], bad); atomicAdd(&res[1], sum);
}
__global__ void sum_in_kernel(const uint32_t* k, const int64_t* v, int64_t n, unsigned long long* res) {
  unsigned long long sum = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    sum += (unsigned long long)v[i] + k[i];
  atomicAdd(res, sum);
}

template <int LOGB, int RPT, bool DIRECT>
static void run(const uint32_t* keys, const int64_t* vals, int64_t n, Rec12* out, int ncls, int tiles_per_block,
                unsigned long long want) {
  constexpr int BINS = 1 << LOGB, TILE = 1024 * RPT;
  uint32_t *counts, *cursor, *bin_start; unsigned long long* res;
  CK(hipMalloc(&counts, (size_t)8 * BINS * 4)); CK(hipMalloc(&cursor, (size_t)8 * BINS * 4));
  CK(hipMalloc(&bin_start, (BINS + 1) * 4)); CK(hipMalloc(&res, 16));
  const int64_t rows_per_block = (int64_t)tiles_per_block * TILE;
  const int grid = (int)((n + rows_per_block - 1) / rows_per_block);
  const size_t lds = (size_t)(2 * BINS + 16) * 4 + (DIRECT ? 0 : (size_t)TILE * 12);
  CK(hipFuncSetAttribute((const void*)scatter_kernel<LOGB, RPT, DIRECT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  float best_h = 1e30f, best_s = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(counts, 0, (size_t)8 * BINS * 4));
    CK(hipEventRecord(e0));
    hist_kernel<LOGB><<<grid, 1024>>>(keys, n, rows_per_block, ncls, counts);
    CK(hipEventRecord(e1));
    scan_kernel<<<1, 64>>>(counts, cursor, bin_start, BINS, ncls);
    scatter_kernel<LOGB, RPT, DIRECT><<<grid, 1024, lds>>>(keys, vals, n, tiles_per_block, ncls, cursor, out);
    CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
    float h, s; CK(hipEventElapsedTime(&h, e0, e1)); CK(hipEventElapsedTime(&s, e1, e2));
    if (h < best_h) best_h = h;
    if (s < best_s) best_s = s;
  }
  CK(hipMemset(res, 0, 16));
  verify_kernel<<<2048, 256>>>(out, bin_start, LOGB, n, res);
  unsigned long long h[2]; CK(hipMemcpy(h, res, 16, hipMemcpyDeviceToHost));
  printf("%-6s bins=%5d tile=%5d tiles/block=%2d cursors=%-9s  hist %7.3f ms  scan+scatter %7.3f ms  %7.1f Grows/s  %7.1f GB/s  %s\n",
         DIRECT ? "direct" : "tile", BINS, TILE, tiles_per_block, ncls == 1 ? "shared" : "per-xcd", best_h, best_s,
         n / best_s * 1e-6, n * 24.0 / best_s * 1e-6, (h[0] == 0 && h[1] == want) ? "ok" : "BAD");
  fflush(stdout);
  CK(hipFree(counts)); CK(hipFree(cursor)); CK(hipFree(bin_start)); CK(hipFree(res));
}

template <int LOGB, int THREADS, int RPT, int ROUNDS, bool NT>
static void run_big(const uint32_t* keys, const int64_t* vals, int64_t n, Rec12* out, int ncls, unsigned long long want) {
  constexpr int BINS = 1 << LOGB, TILE = THREADS * RPT, CHUNK = TILE / ROUNDS;
  uint32_t *counts, *cursor, *bin_start; unsigned long long* res;
  CK(hipMalloc(&counts, (size_t)8 * BINS * 4)); CK(hipMalloc(&cursor, (size_t)8 * BINS * 4));
  CK(hipMalloc(&bin_start, (BINS + 1) * 4)); CK(hipMalloc(&res, 16));
  const int grid = (int)((n + TILE - 1) / TILE);
  const size_t lds = (size_t)(2 * BINS + 16) * 4 + (size_t)CHUNK * 12;
  CK(hipFuncSetAttribute((const void*)scatter_big_kernel<LOGB, THREADS, RPT, ROUNDS, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e1, e2; CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  float best_s = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(counts, 0, (size_t)8 * BINS * 4));
    hist_kernel<LOGB><<<grid, 1024>>>(keys, n, TILE, ncls, counts);
    CK(hipEventRecord(e1));
    scan_kernel<<<1, 64>>>(counts, cursor, bin_start, BINS, ncls);
    scatter_big_kernel<LOGB, THREADS, RPT, ROUNDS, NT><<<grid, THREADS, lds>>>(keys, vals, n, ncls, cursor, out);
    CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
    float s; CK(hipEventElapsedTime(&s, e1, e2));
    if (s < best_s) best_s = s;
  }
  CK(hipMemset(res, 0, 16));
  verify_kernel<<<2048, 256>>>(out, bin_start, LOGB, n, res);
  unsigned long long h[2]; CK(hipMemcpy(h, res, 16, hipMemcpyDeviceToHost));
  printf("big    bins=%5d tile=%5d = %4d thr x %2d, %d round(s) of %5d, %s loads, cursors=%-7s  scan+scatter %7.3f ms  %7.1f Grows/s  %7.1f GB/s  %s\n",
         BINS, TILE, THREADS, RPT, ROUNDS, CHUNK, NT ? "nt" : "plain", ncls == 1 ? "shared" : "per-xcd", best_s,
         n / best_s * 1e-6, n * 24.0 / best_s * 1e-6, (h[0] == 0 && h[1] == want) ? "ok" : "BAD");
  fflush(stdout);
  CK(hipFree(counts)); CK(hipFree(cursor)); CK(hipFree(bin_start)); CK(hipFree(res));
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 28;
  const int64_t n = (int64_t)1 << lg;
  uint32_t* keys; int64_t* vals; Rec12* out;
  CK(hipMalloc(&keys, n * 4)); CK(hipMalloc(&vals, n * 8)); CK(hipMalloc(&out, n * 12));
  fill<<<4096, 256>>>(keys, vals, n);
  unsigned long long* res; CK(hipMalloc(&res, 8)); CK(hipMemset(res, 0, 8));
  sum_in_kernel<<<2048, 256>>>(keys, vals, n, res);
  unsigned long long want; CK(hipMemcpy(&want, res, 8, hipMemcpyDeviceToHost));
  printf("rows=2^%d, key u32 + value i64 -> 12-byte records; 24 B/row moved by the scatter\n", lg);
  const char* what = argc > 2 ? argv[2] : "big";
  if (what[0] == 'a') {   // round 3 call 1: LDS-resident tiles, shared vs per-XCD cursors, tile vs direct
  for (int ncls : {1, 8}) {
    for (int tpb : {1, 4}) {
      run<8, 8, false>(keys, vals, n, out, ncls, tpb, want);
      run<11, 8, false>(keys, vals, n, out, ncls, tpb, want);
      run<13, 6, false>(keys, vals, n, out, ncls, tpb, want);
      run<8, 8, true>(keys, vals, n, out, ncls, tpb, want);
      run<11, 8, true>(keys, vals, n, out, ncls, tpb, want);
      run<13, 8, true>(keys, vals, n, out, ncls, tpb, want);
    }
  }
  } else {
    if (what[0] == 'o') {   // two workgroups per CU: small LDS chunks so that one's loads overlap the other's stores
      for (int ncls : {1, 8}) {
        run_big<11, 1024, 24, 3, false>(keys, vals, n, out, ncls, want);   // the product's form (112 KB: one per CU)
        run_big<11, 512, 48, 6, false>(keys, vals, n, out, ncls, want);    // 64 KB
        run_big<11, 512, 40, 5, false>(keys, vals, n, out, ncls, want);
        run_big<11, 512, 32, 4, false>(keys, vals, n, out, ncls, want);
        run_big<11, 512, 32, 8, false>(keys, vals, n, out, ncls, want);    // 40 KB: three per CU
        run_big<11, 512, 24, 3, false>(keys, vals, n, out, ncls, want);
        run_big<11, 256, 64, 4, false>(keys, vals, n, out, ncls, want);    // 256 threads x 64 rows, 64 KB
        run_big<11, 1024, 24, 6, false>(keys, vals, n, out, ncls, want);
        run_big<10, 512, 48, 6, false>(keys, vals, n, out, ncls, want);
        run_big<10, 512, 32, 4, false>(keys, vals, n, out, ncls, want);
        run_big<9, 512, 32, 4, false>(keys, vals, n, out, ncls, want);
        run_big<9, 1024, 8, 1, false>(keys, vals, n, out, ncls, want);
      }
      return 0;
    }
    for (int ncls : {8, 1}) {
      run_big<11, 1024, 8, 1, false>(keys, vals, n, out, ncls, want);
      run_big<11, 1024, 8, 1, true>(keys, vals, n, out, ncls, want);
      run_big<11, 512, 16, 1, true>(keys, vals, n, out, ncls, want);
      run_big<11, 512, 32, 2, false>(keys, vals, n, out, ncls, want);
      run_big<11, 512, 32, 2, true>(keys, vals, n, out, ncls, want);
      run_big<11, 512, 48, 3, true>(keys, vals, n, out, ncls, want);
      run_big<11, 1024, 16, 2, true>(keys, vals, n, out, ncls, want);
      run_big<11, 1024, 24, 3, true>(keys, vals, n, out, ncls, want);
      run_big<10, 1024, 8, 1, true>(keys, vals, n, out, ncls, want);
      run_big<10, 512, 32, 2, true>(keys, vals, n, out, ncls, want);
      run_big<12, 512, 32, 2, true>(keys, vals, n, out, ncls, want);
      run_big<12, 512, 48, 3, true>(keys, vals, n, out, ncls, want);
      run_big<8, 1024, 8, 1, true>(keys, vals, n, out, ncls, want);
      run_big<5, 1024, 8, 1, true>(keys, vals, n, out, ncls, want);
      run_big<5, 1024, 8, 1, false>(keys, vals, n, out, ncls, want);
    }
  }
  return 0;
}
