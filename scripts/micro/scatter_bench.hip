// Micro-benchmark: high-radix partition (scatter) of 12-byte records on gfx950.
// The MSD sort (key u64 + row u32) and the hash_sum group-by (key i32 + value i64) both spend most of
// their time in radix-partition passes; this program measures, for one pass over N rows, which
// scatter structure keeps HBM busy as the number of bins grows (128 ... 2048):
//   tile   : stage a tile in LDS, reorder by bin, write coalesced runs (chunked exact offsets = MODE 0,
//            or one returning global atomic per (tile, bin) = MODE 1)
//   direct : no LDS reorder — LDS cursors hold the exact global offset of (bin, chunk); every row is
//            stored straight to its slot and the L2 merges the partial lines of a bin's run
// Record layout: SoA (u64 keys[] + u32 idx[]) or AoS (12-byte {lo, hi, idx}).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/scatter_bench.hip -o build/scatter_bench
// Run  : build/scatter_bench [log2_rows=28] [filter-substring]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

struct __attribute__((packed, aligned(4))) Rec12 {
  uint32_t lo, hi, idx;
};

struct Args {
  const uint64_t* keys;   // SoA input keys (or NULL when rec_in)
  const uint32_t* idx_in; // SoA input row ids (NULL: row id = position)
  const Rec12* rec_in;    // AoS input
  int64_t n;
  int shift;              // digit = (key >> shift) & (BINS-1)
  int64_t chunk_rows;
  int nchunks;
  const uint32_t* chunk_off;  // [BINS][nchunks] exclusive global offsets (MODE 0 / direct)
  uint32_t* gcursor;          // [BINS] running cursors (MODE 1); [nparts][BINS] when part_rows != 0
  int64_t part_rows;          // != 0: the input is partition-major (part_rows rows each) and a tile scatters inside its partition
  uint64_t* kout;
  uint32_t* iout;
  Rec12* rout;
};

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t n = __shfl_up(v, d, 64);
    if (lane >= d) v += n;
  }
  return v;
}

__global__ void fill_keys_part(uint64_t* k, int64_t n, uint64_t seed, int64_t part_rows) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint64_t z = seed + (uint64_t)i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    k[i] = ((uint64_t)(i / part_rows) << 56) | (z >> 8);
  }
}

__global__ void fill_keys(uint64_t* k, int64_t n, uint64_t seed) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint64_t z = seed + (uint64_t)i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    k[i] = z ^ (z >> 31);
  }
}

template <int BINS>
__global__ __launch_bounds__(512) void hist_kernel(Args a, uint32_t* __restrict__ hist /*[BINS][nchunks]*/) {
  __shared__ uint32_t h[BINS];
  for (int i = threadIdx.x; i < BINS; i += 512) h[i] = 0;
  __syncthreads();
  const int64_t begin = (int64_t)blockIdx.x * a.chunk_rows;
  const int64_t end = begin + a.chunk_rows < a.n ? begin + a.chunk_rows : a.n;
  for (int64_t r = begin + threadIdx.x; r < end; r += 512) {
    const uint64_t k = a.rec_in ? (((uint64_t)a.rec_in[r].hi << 32) | a.rec_in[r].lo) : a.keys[r];
    atomicAdd(&h[(uint32_t)(k >> a.shift) & (BINS - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BINS; i += 512) hist[(int64_t)i * a.nchunks + blockIdx.x] = h[i];
}

// ---------------------------------------------------------------------------------------------
// tile: LDS reorder.  TILE = THREADS * RPT rows.  MODE 0: chunked exact offsets (one workgroup per chunk);
// MODE 1: one returning global atomic per (tile, bin); the workgroup handles `a.chunk_rows / TILE` consecutive
// tiles and (PF) loads tile t+1 into registers before tile t goes through LDS.
template <int BINS, int THREADS, int RPT, bool AOS_IN, bool AOS_OUT, int MODE, bool PF>
__global__ __launch_bounds__(THREADS) void scatter_tile_kernel(Args a) {
  constexpr int TILE = THREADS * RPT;
  constexpr int BPT = (BINS + THREADS - 1) / THREADS;  // bins per thread in the scan
  __shared__ __attribute__((aligned(16))) uint64_t s_key[TILE];
  __shared__ uint32_t s_idx[TILE];
  __shared__ uint32_t s_cnt[BINS];
  __shared__ uint32_t s_start[BINS];
  __shared__ uint32_t s_gbase[BINS];
  __shared__ uint32_t s_cursor[BINS];
  __shared__ uint32_t s_wtot[THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (MODE == 0) {
    for (int b = tid; b < BINS; b += THREADS) s_cursor[b] = a.chunk_off[(int64_t)b * a.nchunks + blockIdx.x];
  }
  const int64_t begin = (int64_t)blockIdx.x * a.chunk_rows;
  const int64_t end = begin + a.chunk_rows < a.n ? begin + a.chunk_rows : a.n;
  uint64_t key[RPT], nkey[RPT];
  uint32_t idx[RPT], nidx[RPT];
  auto load_tile = [&](int64_t row0, int nrows, uint64_t* k, uint32_t* ix) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int p = i * THREADS + tid;
      const int64_t r = row0 + (p < nrows ? p : nrows - 1);
      if (AOS_IN) {
        const Rec12 rr = a.rec_in[r];
        k[i] = ((uint64_t)rr.hi << 32) | rr.lo;
        ix[i] = rr.idx;
      } else {
        k[i] = a.keys[r];
        ix[i] = (uint32_t)r;
      }
    }
  };
  if (PF && begin < end) load_tile(begin, (int)(end - begin < TILE ? end - begin : TILE), nkey, nidx);
  for (int64_t row0 = begin; row0 < end; row0 += TILE) {
    const int nrows = (int)(end - row0 < TILE ? end - row0 : TILE);
    for (int b = tid; b < BINS; b += THREADS) s_cnt[b] = 0;
    if (PF) {
#pragma unroll
      for (int i = 0; i < RPT; ++i) { key[i] = nkey[i]; idx[i] = nidx[i]; }
    } else {
      load_tile(row0, nrows, key, idx);
    }
    __syncthreads();
    int dig[RPT];
    uint32_t rank[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int p = i * THREADS + tid;
      dig[i] = p < nrows ? (int)((uint32_t)(key[i] >> a.shift) & (BINS - 1)) : -1;
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      rank[i] = 0;
      if (dig[i] >= 0) rank[i] = atomicAdd(&s_cnt[dig[i]], 1u);
    }
    if (PF && row0 + TILE < end) {
      const int64_t nr0 = row0 + TILE;
      load_tile(nr0, (int)(end - nr0 < TILE ? end - nr0 : TILE), nkey, nidx);
    }
    __syncthreads();
    // exclusive scan over BINS counters, BPT consecutive bins per thread
    uint32_t c[BPT], mine = 0;
#pragma unroll
    for (int k = 0; k < BPT; ++k) {
      const int b = tid * BPT + k;
      c[k] = b < BINS ? s_cnt[b] : 0u;
      mine += c[k];
    }
    const uint32_t incl = wave_incl_scan(mine);
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - mine;
    for (int k = 0; k < wave; ++k) pre += s_wtot[k];
#pragma unroll
    for (int k = 0; k < BPT; ++k) {
      const int b = tid * BPT + k;
      if (b < BINS) {
        s_start[b] = pre;
        if (MODE == 0) {
          const uint32_t g = s_cursor[b];
          s_gbase[b] = g;
          s_cursor[b] = g + c[k];
        } else {
          const int64_t pbase = a.part_rows ? (row0 / a.part_rows) * BINS : 0;
          s_gbase[b] = c[k] != 0 ? atomicAdd(&a.gcursor[pbase + b], c[k]) : 0u;
        }
        pre += c[k];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      if (dig[i] >= 0) {
        const uint32_t pos = s_start[dig[i]] + rank[i];
        s_key[pos] = key[i];
        s_idx[pos] = idx[i];
      }
    }
    __syncthreads();
    for (int p = tid; p < nrows; p += THREADS) {
      const uint64_t k = s_key[p];
      const uint32_t d = (uint32_t)(k >> a.shift) & (BINS - 1);
      const uint32_t dst = s_gbase[d] + ((uint32_t)p - s_start[d]);
      if (AOS_OUT) {
        Rec12 r;
        r.lo = (uint32_t)k; r.hi = (uint32_t)(k >> 32); r.idx = s_idx[p];
        a.rout[dst] = r;
      } else {
        a.kout[dst] = k;
        a.iout[dst] = s_idx[p];
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// direct: LDS cursors = exact global offsets; rows go straight to their slot.
template <int BINS, int THREADS, int U, bool AOS_IN, bool AOS_OUT>
__global__ __launch_bounds__(THREADS) void scatter_direct_kernel(Args a) {
  __shared__ uint32_t s_cursor[BINS];
  const int tid = threadIdx.x;
  for (int b = tid; b < BINS; b += THREADS) s_cursor[b] = a.chunk_off[(int64_t)b * a.nchunks + blockIdx.x];
  __syncthreads();
  const int64_t begin = (int64_t)blockIdx.x * a.chunk_rows;
  const int64_t end = begin + a.chunk_rows < a.n ? begin + a.chunk_rows : a.n;
  for (int64_t row0 = begin; row0 < end; row0 += THREADS * U) {
    uint64_t key[U];
    uint32_t idx[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = row0 + u * THREADS + tid;
      ok[u] = r < end;
      key[u] = 0; idx[u] = 0;
      if (ok[u]) {
        if (AOS_IN) {
          const Rec12 rr = a.rec_in[r];
          key[u] = ((uint64_t)rr.hi << 32) | rr.lo;
          idx[u] = rr.idx;
        } else {
          key[u] = a.keys[r];
          idx[u] = a.idx_in ? a.idx_in[r] : (uint32_t)r;
        }
      }
    }
    uint32_t pos[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      pos[u] = 0;
      if (ok[u]) pos[u] = atomicAdd(&s_cursor[(uint32_t)(key[u] >> a.shift) & (BINS - 1)], 1u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ok[u]) {
        if (AOS_OUT) {
          Rec12 r;
          r.lo = (uint32_t)key[u]; r.hi = (uint32_t)(key[u] >> 32); r.idx = idx[u];
          a.rout[pos[u]] = r;
        } else {
          a.kout[pos[u]] = key[u];
          a.iout[pos[u]] = idx[u];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// verification: every output slot of bin b holds a key of digit b whose idx points back at it
__global__ void verify_kernel(Args a, const uint64_t* src_keys, const uint32_t* bin_start, int bins,
                              unsigned long long* bad, unsigned long long* keysum) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long s = 0;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.n; p += stride) {
    uint64_t k;
    uint32_t id;
    if (a.rout) {
      k = ((uint64_t)a.rout[p].hi << 32) | a.rout[p].lo;
      id = a.rout[p].idx;
    } else {
      k = a.kout[p];
      id = a.iout[p];
    }
    const uint32_t d = (uint32_t)(k >> a.shift) & (bins - 1);
    bool ok = p >= bin_start[d] && p < bin_start[d + 1] && src_keys[id] == k;
    if (!ok) atomicAdd(bad, 1ull);
    s += k;
  }
  atomicAdd(keysum, s);
}

struct Variant {
  std::string name;
  int bins, threads, rows_per_step;
  bool aos_in, aos_out;
  int mode;  // 0 tile/exact, 1 tile/global cursor, 2 direct
  void (*launch)(Args, int grid, hipStream_t);
  void (*hist)(Args, uint32_t*, int grid, hipStream_t);
};

template <int BINS, int THREADS, int RPT, bool AI, bool AO, int MODE, bool PF = false>
static void launch_tile(Args a, int grid, hipStream_t st) {
  hipLaunchKernelGGL((scatter_tile_kernel<BINS, THREADS, RPT, AI, AO, MODE, PF>), dim3(grid), dim3(THREADS), 0, st, a);
}
template <int BINS, int THREADS, int U, bool AI, bool AO>
static void launch_direct(Args a, int grid, hipStream_t st) {
  hipLaunchKernelGGL((scatter_direct_kernel<BINS, THREADS, U, AI, AO>), dim3(grid), dim3(THREADS), 0, st, a);
}
template <int BINS>
static void launch_hist(Args a, uint32_t* h, int grid, hipStream_t st) {
  hipLaunchKernelGGL((hist_kernel<BINS>), dim3(grid), dim3(512), 0, st, a, h);
}

#define TILE_V(B, T, R, AI, AO, M)                                                                   \
  Variant {                                                                                          \
    std::string("tile") + (M ? "G" : "X") + " b" #B " t" #T "x" #R + (AI ? " aosin" : " soain") +    \
        (AO ? " aosout" : " soaout"),                                                                \
        B, T, T * R, AI, AO, M, launch_tile<B, T, R, AI, AO, M>, launch_hist<B>                      \
  }
#define TILE_PF(B, T, R, AI, AO, M)                                                                  \
  Variant {                                                                                          \
    std::string("tile") + (M ? "G" : "X") + "pf b" #B " t" #T "x" #R + (AI ? " aosin" : " soain") +  \
        (AO ? " aosout" : " soaout"),                                                                \
        B, T, T * R, AI, AO, M, launch_tile<B, T, R, AI, AO, M, true>, launch_hist<B>                \
  }
#define DIRECT_V(B, T, U, AI, AO)                                                                    \
  Variant {                                                                                          \
    std::string("direct b" #B " t" #T "x" #U) + (AI ? " aosin" : " soain") + (AO ? " aosout" : " soaout"), \
        B, T, T * U, AI, AO, 2, launch_direct<B, T, U, AI, AO>, launch_hist<B>                       \
  }

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 28;
  const char* filter = argc > 2 ? argv[2] : "";
  const int64_t n = (int64_t)1 << lg;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  uint64_t *keys, *kout;
  uint32_t *iout, *d_off, *d_cursor, *d_binstart;
  Rec12 *rec_a, *rec_b;
  unsigned long long* d_stats;
  CK(hipMalloc(&keys, n * 8));
  CK(hipMalloc(&kout, n * 8));
  CK(hipMalloc(&iout, n * 4));
  CK(hipMalloc(&rec_a, n * 12));
  CK(hipMalloc(&rec_b, n * 12));
  const int kMaxBins = 1024, kMaxChunks = 65536;
  CK(hipMalloc(&d_off, (size_t)kMaxBins * kMaxChunks * 4));
  CK(hipMalloc(&d_cursor, kMaxBins * 4));
  uint32_t* d_cursor_big;
  CK(hipMalloc(&d_cursor_big, 256 * 1024 * 4));
  CK(hipMalloc(&d_binstart, (kMaxBins + 1) * 4));
  CK(hipMalloc(&d_stats, 16));
  hipLaunchKernelGGL(fill_keys, dim3(4096), dim3(256), 0, st, keys, n, 0x1234ull);
  CK(hipStreamSynchronize(st));
  // an AoS copy of the input for the aos-in variants: 128-bin scatter of the keys (so it is "level-1 output" shaped)
  std::vector<Variant> vs = {
      // chunked exact offsets (round-1 level-1 shape) vs global cursors, SoA vs AoS
      TILE_V(128, 512, 8, false, false, 0), TILE_V(128, 512, 8, false, true, 0),
      TILE_V(128, 512, 8, false, false, 1), TILE_V(128, 512, 8, false, true, 1), TILE_V(128, 512, 8, true, true, 1),
      TILE_V(64, 512, 8, false, true, 1), TILE_V(64, 512, 8, true, true, 1),
      TILE_V(256, 512, 8, false, true, 1), TILE_V(256, 512, 8, true, true, 1),
      TILE_V(512, 512, 8, false, true, 0), TILE_V(512, 512, 8, false, true, 1), TILE_V(512, 512, 8, true, true, 1),
      TILE_V(1024, 512, 8, false, true, 1), TILE_V(1024, 512, 8, true, true, 1),
      TILE_V(512, 1024, 8, false, true, 1), TILE_V(512, 1024, 8, true, true, 1),
      TILE_V(1024, 1024, 8, false, true, 1), TILE_V(1024, 1024, 8, true, true, 1),
      TILE_V(128, 256, 8, false, true, 1), TILE_V(128, 256, 16, false, true, 1), TILE_V(256, 256, 16, true, true, 1),
      TILE_V(128, 1024, 4, false, true, 1),
      // register prefetch of the next tile (the workgroup walks several consecutive tiles)
      TILE_PF(128, 512, 8, false, true, 1), TILE_PF(128, 512, 8, true, true, 1), TILE_PF(512, 512, 8, false, true, 1),
      TILE_PF(512, 512, 8, true, true, 1), TILE_PF(1024, 1024, 8, true, true, 1), TILE_PF(512, 1024, 8, true, true, 1),
      TILE_PF(128, 512, 8, false, true, 0), TILE_PF(512, 512, 8, false, true, 0),
  };
  const int grids[] = {1024, 2048, 4096, 16384, 65536};
  printf("rows=2^%d (%lld); bytes moved per row: soain 8, aosin 12; out 12\n", lg, (long long)n);
  printf("%-44s %6s %9s %9s %9s %8s\n", "variant", "grid", "ms", "Grows/s", "GB/s", "check");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<uint32_t> h_hist, h_off;
  for (const Variant& v : vs) {
    if (filter[0] && !strstr(v.name.c_str(), filter)) continue;
    for (int grid : grids) {
      if (v.mode == 0 && grid > 4096) continue;
      Args a{};
      a.n = n;
      a.shift = 64 - 9 - 11;  // a digit somewhere in the middle of the (uniform) key
      a.chunk_rows = (n + grid - 1) / grid;
      a.chunk_rows = (a.chunk_rows + v.rows_per_step - 1) / v.rows_per_step * v.rows_per_step;
      a.nchunks = (int)((n + a.chunk_rows - 1) / a.chunk_rows);
      a.keys = v.aos_in ? nullptr : keys;
      a.rec_in = v.aos_in ? rec_a : nullptr;
      a.kout = v.aos_out ? nullptr : kout;
      a.iout = v.aos_out ? nullptr : iout;
      a.rout = v.aos_out ? rec_b : nullptr;
      a.chunk_off = d_off;
      a.gcursor = d_cursor;
      if (v.aos_in) {
        // build rec_a = {key, row} in row order (any order would do: the digit used here is independent)
        Args b = a;
        b.keys = keys; b.rec_in = nullptr; b.rout = rec_a; b.kout = nullptr; b.iout = nullptr;
        // identity "scatter": one bin per chunk trick is overkill; use a tiny kernel via direct variant with exact offsets = row
        // (cheap way: 1 bin) — handled by hist on 1 bin below
        b.shift = 63; // digits 0/1 -> use 128-bin kernel with shift 63 (only bins 0,1 used)
        h_hist.assign((size_t)128 * a.nchunks, 0);
        CK(hipMemsetAsync(d_off, 0, (size_t)kMaxBins * kMaxChunks * 4, st));
        b.chunk_off = d_off;
        launch_hist<128>(b, d_off, a.nchunks, st);
        CK(hipStreamSynchronize(st));
        h_hist.resize((size_t)128 * a.nchunks);
        CK(hipMemcpy(h_hist.data(), d_off, h_hist.size() * 4, hipMemcpyDeviceToHost));
        uint32_t run = 0;
        for (size_t i = 0; i < h_hist.size(); ++i) { uint32_t c = h_hist[i]; h_hist[i] = run; run += c; }
        CK(hipMemcpy(d_off, h_hist.data(), h_hist.size() * 4, hipMemcpyHostToDevice));
        launch_tile<128, 512, 8, false, true, 0, false>(b, a.nchunks, st);
        CK(hipStreamSynchronize(st));
      }
      // histogram + exclusive scan (bin-major, then chunk) on the host: not timed
      v.hist(a, d_off, a.nchunks, st);
      CK(hipStreamSynchronize(st));
      h_hist.resize((size_t)v.bins * a.nchunks);
      CK(hipMemcpy(h_hist.data(), d_off, h_hist.size() * 4, hipMemcpyDeviceToHost));
      std::vector<uint32_t> binstart(v.bins + 1);
      uint32_t run = 0;
      for (int b = 0; b < v.bins; ++b) {
        binstart[b] = run;
        for (int c = 0; c < a.nchunks; ++c) {
          uint32_t x = h_hist[(size_t)b * a.nchunks + c];
          h_hist[(size_t)b * a.nchunks + c] = run;
          run += x;
        }
      }
      binstart[v.bins] = run;
      CK(hipMemcpy(d_off, h_hist.data(), h_hist.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(d_binstart, binstart.data(), (v.bins + 1) * 4, hipMemcpyHostToDevice));
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemcpyAsync(d_cursor, d_binstart, v.bins * 4, hipMemcpyDeviceToDevice, st));
        CK(hipEventRecord(e0, st));
        v.launch(a, a.nchunks, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      CK(hipMemsetAsync(d_stats, 0, 16, st));
      hipLaunchKernelGGL(verify_kernel, dim3(2048), dim3(256), 0, st, a, keys, d_binstart, v.bins, d_stats, d_stats + 1);
      unsigned long long stats[2];
      CK(hipMemcpyAsync(stats, d_stats, 16, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      const double bytes = (double)n * ((v.aos_in ? 12 : 8) + 12);
      printf("%-44s %6d %9.3f %9.2f %9.1f %8s\n", v.name.c_str(), a.nchunks, best, n / best / 1e6, bytes / best / 1e6,
             stats[0] == 0 ? "ok" : "BAD");
      fflush(stdout);
    }
  }
  // ---- locality test: the input is partition-major (P partitions); a tile scatters inside its own partition
  // (what a second radix level does) vs the flat scatter above whose bins span the whole output
  for (int P : {16, 64, 256}) {
    const int64_t part_rows = n / P;
    hipLaunchKernelGGL(fill_keys_part, dim3(4096), dim3(256), 0, st, keys, n, 0x77ull, part_rows);
    CK(hipStreamSynchronize(st));
    for (int which = 0; which < 3; ++which) {
      const int bins = which == 0 ? 64 : (which == 1 ? 128 : 512);
      Args a{};
      a.n = n;
      a.shift = 56 - (which == 0 ? 6 : (which == 1 ? 7 : 9));
      a.keys = keys;
      a.rout = rec_b;
      a.gcursor = d_cursor_big;
      a.part_rows = part_rows;
      const int tile = which == 2 ? 8192 : 4096;
      a.chunk_rows = tile;
      a.nchunks = (int)((n + tile - 1) / tile);
      // counts per (partition, bin): histogram with chunk == partition
      Args h = a;
      h.chunk_rows = part_rows;
      h.nchunks = P;
      if (bins == 64) launch_hist<64>(h, d_off, P, st); else if (bins == 128) launch_hist<128>(h, d_off, P, st); else launch_hist<512>(h, d_off, P, st);
      CK(hipStreamSynchronize(st));
      h_hist.resize((size_t)bins * P);
      CK(hipMemcpy(h_hist.data(), d_off, h_hist.size() * 4, hipMemcpyDeviceToHost));
      std::vector<uint32_t> cur((size_t)P * bins);
      uint32_t run = 0;
      for (int pp = 0; pp < P; ++pp)
        for (int b = 0; b < bins; ++b) { cur[(size_t)pp * bins + b] = run; run += h_hist[(size_t)b * P + pp]; }
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemcpy(d_cursor_big, cur.data(), cur.size() * 4, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, st));
        if (bins == 64) launch_tile<64, 512, 8, false, true, 1>(a, a.nchunks, st);
        else if (bins == 128) launch_tile<128, 512, 8, false, true, 1>(a, a.nchunks, st);
        else launch_tile<512, 1024, 8, false, true, 1>(a, a.nchunks, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      printf("tileG inside %3d partitions, b%d tile %d soain aosout            %9.3f ms %9.2f Grows/s %9.1f GB/s\n", P, bins, tile,
             best, n / best / 1e6, (double)n * 20 / best / 1e6);
      fflush(stdout);
    }
  }
  return 0;
}
