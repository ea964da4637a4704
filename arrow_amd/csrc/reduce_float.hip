// Scalar aggregates of float32 / float64 columns for gfx950: min_max by order keys, and a sum that equals the
// reference's bit for bit.
//
// What it restates (semantics only):
//   MinMaxState<floating>  cpp/src/arrow/compute/kernels/aggregate_basic.inc.cc:681-701   fmin / fmax over NaN anti-extrema
//   SumArray (floating)    cpp/src/arrow/compute/kernels/aggregate_internal.h:155-232      pairwise summation
//   SumImpl / MeanImpl     cpp/src/arrow/compute/kernels/aggregate_basic.inc.cc:49-110, 262-286
//
// The reference's float sum is not "any order": SumArray adds the valid values of every RUN of valid slots in BLOCKS of 16
// (left to right, a run's last block may be shorter, a new run starts a new block) and merges the block sums with a binary
// counter — block i joins the partial sum of its level, two partial sums of one level are added and move up —, folding the
// levels left over at the end from the lowest up.  That is a fixed tree over the sequence of blocks, so a parallel machine
// can evaluate exactly the same additions:
//   * blocks: one thread per block start adds its <= 16 values in order (fsum_leaf_* / fsum_block_sums);
//   * the tree: an aligned group of 2^11 consecutive entries of one level is one complete subtree — a workgroup adds
//     neighbours pairwise in LDS (fsum_tree): level L entries -> level L + 11 entries; the entries behind the last complete
//     group (< 2^11 of them) are handed to the host as they are;
//   * the host replays the binary counter over what is left (a few thousand doubles): complete subtrees enter at their
//     level, which leaves the counter in exactly the state 2^L single blocks would have.
// With nulls the block boundaries depend on where the runs start: per 64-row validity word the carry "rows of the run so
// far, mod 16" (64 = 0 mod 16, so an all-valid word passes it on unchanged) is resolved by a two-level scan, block
// starts are counted and numbered the same way, and every word's thread then adds the blocks that start in it.
// HBM traffic: the values once (8 or 4 B/row) + 8 B per block (0.5 B/row without nulls).
#include <cmath>
#include "arx_common.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace arx {

// ------------------------------------------------------------------ min_max of floats: order keys, NaNs skipped
// acc = {unused, count of valid values, min key, max key} (4 x int64, arx_reduce_i64_init); a column of NaNs only
// leaves the anti-extrema INT64_MAX / INT64_MIN, whose own pattern as an order key is a NaN.
template <typename T>
__global__ __launch_bounds__(kBlock) void reduce_float_minmax_kernel(const T* __restrict__ in, Bits valid, int64_t n,
                                                                     unsigned long long* __restrict__ acc) {
  __shared__ unsigned long long s_cnt[kWavesPerBlock];
  __shared__ long long s_min[kWavesPerBlock], s_max[kWavesPerBlock];
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  unsigned long long cnt = 0;
  long long mn = INT64_MAX, mx = INT64_MIN;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool ok = (load_word(valid, i >> 6) >> (i & 63)) & 1ull;
    const double d = static_cast<double>(in[i]);
    if (ok) {
      ++cnt;
      if (d == d) {
        const long long k = float_order_key(d);
        mn = k < mn ? k : mn;
        mx = k > mx ? k : mx;
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    cnt += __shfl_xor(cnt, d, 64);
    const long long omn = __shfl_xor(mn, d, 64), omx = __shfl_xor(mx, d, 64);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
  }
  const int wave = threadIdx.x >> 6;
  if (lane_id() == 0) {
    s_cnt[wave] = cnt;
    s_min[wave] = mn;
    s_max[wave] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kWavesPerBlock; ++k) {
      cnt += s_cnt[k];
      mn = s_min[k] < mn ? s_min[k] : mn;
      mx = s_max[k] > mx ? s_max[k] : mx;
    }
    if (cnt != 0) {
      atomicAdd(&acc[1], cnt);
      atomicMin(reinterpret_cast<long long*>(&acc[2]), mn);
      atomicMax(reinterpret_cast<long long*>(&acc[3]), mx);
    }
  }
}

// ------------------------------------------------------------------ the pairwise sum
constexpr int kFsumBlock = 16;                   // kBlockSize, aggregate_internal.h:169
constexpr int kFsumLg = 11;                      // levels one tree pass climbs
constexpr int kFsumGroup = 1 << kFsumLg;         // entries of one complete subtree
constexpr int kFsumThreads = 256;
constexpr int kFsumTileWords = 256;              // validity words per workgroup of the scan kernels

// adjacent-pair tree over the kFsumGroup doubles in `e` (LDS); the result ends up in e[0]
__device__ __forceinline__ void fsum_tree_in_lds(double* e, int tid) {
  for (int d = 1; d < kFsumGroup; d <<= 1) {
    __syncthreads();
    for (int j = tid; j < kFsumGroup / (2 * d); j += kFsumThreads) {
      const int i = 2 * d * j;
      e[i] = e[i] + e[i + d];
    }
  }
  __syncthreads();
}

// no nulls: block b = rows [16 b, 16 b + 16).  One workgroup per group of kFsumGroup blocks: complete groups leave one
// level-11 entry in out[g], the blocks of the last, incomplete group go to tail[] as they are.
template <typename T>
__global__ __launch_bounds__(kFsumThreads) void fsum_leaf_dense_kernel(const T* __restrict__ in, int64_t n, int64_t nblocks,
                                                                       double* __restrict__ out, double* __restrict__ tail) {
  __shared__ double e[kFsumGroup];
  const int tid = threadIdx.x;
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * kFsumGroup;
  const int64_t have = nblocks - b0 < kFsumGroup ? nblocks - b0 : kFsumGroup;
  for (int j = tid; j < kFsumGroup; j += kFsumThreads) {
    double s = 0;
    if (j < have) {
      const int64_t r0 = (b0 + j) * kFsumBlock;
      const int len = n - r0 < kFsumBlock ? static_cast<int>(n - r0) : kFsumBlock;
      for (int k = 0; k < len; ++k) s += static_cast<double>(in[r0 + k]);
    }
    e[j] = s;
  }
  if (have < kFsumGroup) {   // (workgroup-uniform) the last group: no tree
    __syncthreads();
    for (int j = tid; j < have; j += kFsumThreads) tail[j] = e[j];
    return;
  }
  fsum_tree_in_lds(e, tid);
  if (tid == 0) out[blockIdx.x] = e[0];
}

// one tree pass: entries of one level -> entries 11 levels up (+ the tail)
__global__ __launch_bounds__(kFsumThreads) void fsum_tree_kernel(const double* __restrict__ in, int64_t count,
                                                                 double* __restrict__ out, double* __restrict__ tail) {
  __shared__ double e[kFsumGroup];
  const int tid = threadIdx.x;
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * kFsumGroup;
  const int64_t have = count - b0 < kFsumGroup ? count - b0 : kFsumGroup;
  if (have < kFsumGroup) {
    for (int j = tid; j < have; j += kFsumThreads) tail[j] = in[b0 + j];
    return;
  }
  for (int j = tid; j < kFsumGroup; j += kFsumThreads) e[j] = in[b0 + j];
  fsum_tree_in_lds(e, tid);
  if (tid == 0) out[blockIdx.x] = e[0];
}

// ---- with nulls.  Per validity word w (rows 64 w ..): carry(w) = rows of the run that reaches into w, mod 16.
// A word with a null slot "resets": the carry it hands on is its number of leading (top) valid slots mod 16.
struct FsumScan {
  Bits valid;
  int64_t nwords, ntiles;
  uint8_t* word_carry;     // [nwords] carry from a reset inside the tile, 0xFF = the tile's carry-in
  uint8_t* tile_last;      // [ntiles] carry the tile hands on, 0xFF = none of its words resets
  uint8_t* tile_carry;     // [ntiles] carry into the tile
  uint32_t* word_blocks;   // [nwords] block starts before the word, inside its tile
  uint32_t* tile_blocks;   // [ntiles] block starts of the tile
  int64_t* tile_base;      // [ntiles] block starts before the tile
  int64_t* totals;         // {blocks, valid rows}
};

__device__ __forceinline__ int fsum_top_ones16(uint64_t m) { return __builtin_clzll(~m) & 15; }   // (m != all ones)

__global__ __launch_bounds__(kFsumTileWords) void fsum_word_carry_kernel(FsumScan a) {
  __shared__ int last_reset[kFsumTileWords];
  __shared__ uint8_t top[kFsumTileWords];
  const int t = threadIdx.x;
  const int64_t w = static_cast<int64_t>(blockIdx.x) * kFsumTileWords + t;
  const uint64_t m = w < a.nwords ? load_word(a.valid, w) : ~uint64_t(0);   // (words past the end pass the carry on: unused)
  const bool reset = m != ~uint64_t(0);
  top[t] = reset ? static_cast<uint8_t>(fsum_top_ones16(m)) : 0;
  last_reset[t] = reset ? t : -1;
  // inclusive max-scan of the reset positions (Hillis-Steele over the tile)
  for (int d = 1; d < kFsumTileWords; d <<= 1) {
    __syncthreads();
    const int other = t >= d ? last_reset[t - d] : -1;
    __syncthreads();
    if (other > last_reset[t]) last_reset[t] = other;
  }
  __syncthreads();
  const int before = t > 0 ? last_reset[t - 1] : -1;
  if (w < a.nwords) a.word_carry[w] = before >= 0 ? top[before] : 0xFF;
  if (t == kFsumTileWords - 1) a.tile_last[blockIdx.x] = last_reset[t] >= 0 ? top[last_reset[t]] : 0xFF;
}

// carries into the tiles: tile_carry[i] = the last tile_last[j] != 0xFF with j < i (0 when there is none: the column's
// first run starts at row 0).  1024 threads, each walks a contiguous chunk; thread 0 chains the chunk summaries.
__global__ __launch_bounds__(1024) void fsum_tile_carry_kernel(FsumScan a) {
  __shared__ uint8_t chunk_last[1024];
  __shared__ uint8_t chunk_in[1024];
  const int t = threadIdx.x;
  const int64_t per = (a.ntiles + 1023) / 1024;
  const int64_t lo = t * per, hi = lo + per < a.ntiles ? lo + per : a.ntiles;
  uint8_t last = 0xFF;
  for (int64_t i = lo; i < hi; ++i) {
    const uint8_t v = a.tile_last[i];
    if (v != 0xFF) last = v;
  }
  chunk_last[t] = last;
  __syncthreads();
  if (t == 0) {
    uint8_t run = 0;
    for (int k = 0; k < 1024; ++k) {
      chunk_in[k] = run;
      if (chunk_last[k] != 0xFF) run = chunk_last[k];
    }
  }
  __syncthreads();
  uint8_t run = chunk_in[t];
  for (int64_t i = lo; i < hi; ++i) {
    a.tile_carry[i] = run;
    const uint8_t v = a.tile_last[i];
    if (v != 0xFF) run = v;
  }
}

// block starts of the first run of a word (length L0 from bit 0, carry c): positions p < L0 with (c + p) % 16 == 0
__device__ __forceinline__ int fsum_first_run_starts(int c, int L0) { return (c + L0 + 15) / 16 - (c > 0 ? 1 : 0); }

__device__ __forceinline__ int fsum_run_length(uint64_t m, int s) {   // valid slots from bit s on (bit s is set)
  const uint64_t inv = ~(m >> s);
  const int l = inv == 0 ? 64 : __builtin_ctzll(inv);
  return l < 64 - s ? l : 64 - s;
}

__global__ __launch_bounds__(kFsumTileWords) void fsum_word_blocks_kernel(FsumScan a) {
  __shared__ uint32_t cnt[kFsumTileWords];
  __shared__ unsigned long long s_valid;
  const int t = threadIdx.x;
  const int64_t w = static_cast<int64_t>(blockIdx.x) * kFsumTileWords + t;
  if (t == 0) s_valid = 0;
  uint32_t starts = 0;
  uint64_t m = 0;
  if (w < a.nwords) {
    m = load_word(a.valid, w);
    const uint8_t wc = a.word_carry[w];
    const int c = wc == 0xFF ? a.tile_carry[blockIdx.x] : wc;
    const int L0 = (m & 1) ? fsum_run_length(m, 0) : 0;
    starts = static_cast<uint32_t>(fsum_first_run_starts(c, L0));
    uint64_t x = L0 >= 64 ? 0 : (m >> L0) << L0;   // the runs behind the first
    while (x != 0) {
      const int s = __builtin_ctzll(x);
      const int l = fsum_run_length(m, s);
      starts += static_cast<uint32_t>((l + 15) / 16);
      x = s + l >= 64 ? 0 : (x >> (s + l)) << (s + l);
    }
  }
  cnt[t] = starts;
  __syncthreads();
  const unsigned long long pc = wave_reduce_sum_u64(static_cast<unsigned long long>(__popcll(m)));
  if (lane_id() == 0 && pc != 0) atomicAdd(&s_valid, pc);
  // exclusive scan of the starts inside the tile
  for (int d = 1; d < kFsumTileWords; d <<= 1) {
    const uint32_t other = t >= d ? cnt[t - d] : 0;
    __syncthreads();
    cnt[t] += other;
    __syncthreads();
  }
  if (w < a.nwords) a.word_blocks[w] = cnt[t] - starts;
  if (t == kFsumTileWords - 1) {
    a.tile_blocks[blockIdx.x] = cnt[t];
    if (s_valid != 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.totals + 1), s_valid);
  }
}

__global__ __launch_bounds__(1024) void fsum_tile_base_kernel(FsumScan a) {
  __shared__ unsigned long long chunk_sum[1024];
  const int t = threadIdx.x;
  const int64_t per = (a.ntiles + 1023) / 1024;
  const int64_t lo = t * per, hi = lo + per < a.ntiles ? lo + per : a.ntiles;
  unsigned long long s = 0;
  for (int64_t i = lo; i < hi; ++i) s += a.tile_blocks[i];
  chunk_sum[t] = s;
  __syncthreads();
  if (t == 0) {
    unsigned long long run = 0;
    for (int k = 0; k < 1024; ++k) {
      const unsigned long long c = chunk_sum[k];
      chunk_sum[k] = run;
      run += c;
    }
    a.totals[0] = static_cast<int64_t>(run);
  }
  __syncthreads();
  unsigned long long run = chunk_sum[t];
  for (int64_t i = lo; i < hi; ++i) {
    a.tile_base[i] = static_cast<int64_t>(run);
    run += a.tile_blocks[i];
  }
}

// every word's thread adds the blocks that START in it (a block of <= 16 rows may reach into the next word)
template <typename T>
__global__ __launch_bounds__(kFsumTileWords) void fsum_block_sums_kernel(FsumScan a, const T* __restrict__ in,
                                                                         double* __restrict__ block_sums) {
  const int t = threadIdx.x;
  const int64_t w = static_cast<int64_t>(blockIdx.x) * kFsumTileWords + t;
  if (w >= a.nwords) return;
  const uint64_t m = load_word(a.valid, w);
  if (m == 0) return;
  const uint64_t m2 = w + 1 < a.nwords ? load_word(a.valid, w + 1) : 0;
  const uint8_t wc = a.word_carry[w];
  const int c = wc == 0xFF ? a.tile_carry[blockIdx.x] : wc;
  int64_t o = a.tile_base[blockIdx.x] + a.word_blocks[w];
  const int64_t row0 = w << 6;
  auto emit = [&](int p) {   // the block that starts at bit p: its valid slots in a row, at most 16
    const uint64_t seq = p == 0 ? m : ((m >> p) | (m2 << (64 - p)));
    const uint64_t inv = ~seq;
    int len = inv == 0 ? 64 : __builtin_ctzll(inv);
    len = len < kFsumBlock ? len : kFsumBlock;
    double s = 0;
    for (int k = 0; k < len; ++k) s += static_cast<double>(in[row0 + p + k]);
    block_sums[o++] = s;
  };
  const int L0 = (m & 1) ? fsum_run_length(m, 0) : 0;
  for (int p = (16 - c) & 15; p < L0; p += kFsumBlock) emit(p);
  uint64_t x = L0 >= 64 ? 0 : (m >> L0) << L0;
  while (x != 0) {
    const int s = __builtin_ctzll(x);
    const int l = fsum_run_length(m, s);
    for (int p = s; p < s + l; p += kFsumBlock) emit(p);
    x = s + l >= 64 ? 0 : (x >> (s + l)) << (s + l);
  }
}

// the reference's binary counter (aggregate_internal.h:170-231), with entries that may enter above level 0
struct PairwiseCounter {
  double sum[80];
  uint64_t mask = 0;
  int root_level = 0;
  PairwiseCounter() { std::memset(sum, 0, sizeof(sum)); }
  void reduce(int level, double v) {
    int cur = level;
    uint64_t cur_mask = uint64_t(1) << cur;
    sum[cur] += v;
    mask ^= cur_mask;
    while ((mask & cur_mask) == 0) {
      v = sum[cur];
      sum[cur] = 0;
      ++cur;
      cur_mask <<= 1;
      sum[cur] += v;
      mask ^= cur_mask;
    }
    root_level = std::max(root_level, cur);
  }
  double finish() {
    for (int i = 1; i <= root_level; ++i) sum[i] += sum[i - 1];
    return sum[root_level];
  }
};

static inline size_t fsum_align(size_t x) { return (x + 255) & ~size_t(255); }

static int64_t fsum_max_blocks(int64_t n, int64_t null_count) {
  if (null_count == 0) return ceil_div(n, kFsumBlock);
  const int64_t nulls = null_count < 0 ? n : null_count;   // unknown: every slot may be one
  return std::min<int64_t>(ceil_div(n, kFsumBlock) + nulls + 1, (n + 1) / 2 + 1);
}

// ---- hash_sum / hash_mean of float32 / float64 values over dense group ids.  GroupedReducingAggregator<FloatType /
// DoubleType, GroupedSumImpl | GroupedMeanImpl> (hash_aggregate_numeric.cc:44-152,352-430) adds every row to its group's
// DOUBLE accumulator in ROW ORDER (Consume :70-83 -> VisitGroupedValues; Reduce = double(u) + double(v)) — a sum whose
// value depends on the order, so "the same result" means the same order.  Here: the rows are stably sorted by group id
// (arx_sort_indices keeps equal keys in row order), which makes every group one run of the permutation in row order, and
// ONE thread walks a group's run adding from the group's running sum — the additions of the reference, group by group
// instead of row by row.  No atomics: a group has one owner per call; batches continue where the last one stopped.
constexpr int64_t kFsumLongRun = 1024;

// what a walker does with a row: A = double (sums and products of floats, in row order) or uint64_t (hash_product of the
// integer types: MultiplyTraits multiplies in the unsigned type, wrapping — base_arithmetic_internal.h:303-325);
// skip() is the value a row that does not count is replaced by, the operation's identity (x + -0.0 == x and x * 1.0 == x
// bit for bit, for every x)
template <typename A, bool PRODUCT>
struct WalkOp;
template <>
struct WalkOp<double, false> {
  static __device__ __forceinline__ double skip() { return -0.0; }
  static __device__ __forceinline__ double apply(double acc, double v) { return acc + v; }
};
template <>
struct WalkOp<double, true> {
  static __device__ __forceinline__ double skip() { return 1.0; }
  static __device__ __forceinline__ double apply(double acc, double v) { return acc * v; }
};
template <>
struct WalkOp<uint64_t, true> {
  static __device__ __forceinline__ uint64_t skip() { return 1; }
  static __device__ __forceinline__ uint64_t apply(uint64_t acc, uint64_t v) { return acc * v; }
};

// the rows in (group id, row) order: group id, value as a double, validity — what the walkers then read sequentially
__global__ __launch_bounds__(kBlock) void fill_u64_kernel(uint64_t* __restrict__ out, uint64_t value, int64_t n) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) out[i] = value;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void hash_fsum_gather_kernel(const T* __restrict__ values, Bits vvalid, int is_scalar, double scalar,
                                                                  int scalar_valid, const uint32_t* __restrict__ gids,
                                                                  const uint64_t* __restrict__ perm, int64_t n,
                                                                  uint32_t* __restrict__ gs, double* __restrict__ vs,
                                                                  uint8_t* __restrict__ oks) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint64_t r = perm[i];
    gs[i] = gids[r];
    if (is_scalar) {
      vs[i] = scalar;
      oks[i] = scalar_valid != 0 ? 1 : 0;
    } else {
      vs[i] = static_cast<double>(values[r]);
      oks[i] = (vvalid.base == nullptr || ((load_word(vvalid, static_cast<int64_t>(r) >> 6) >> (r & 63)) & 1ull)) ? 1 : 0;
    }
  }
}

// the same for the integer types: the value widened to 64 bits (sign- or zero-extended: the product's accumulator type)
template <typename T>
__global__ __launch_bounds__(kBlock) void hash_iprod_gather_kernel(const T* __restrict__ values, Bits vvalid, const uint32_t* __restrict__ gids,
                                                                   const uint64_t* __restrict__ perm, int64_t n, uint32_t* __restrict__ gs,
                                                                   uint64_t* __restrict__ vs, uint8_t* __restrict__ oks) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint64_t r = perm[i];
    gs[i] = gids[r];
    vs[i] = static_cast<uint64_t>(static_cast<int64_t>(values[r]));   // (unsigned T: zero-extended by the first cast)
    oks[i] = (vvalid.base == nullptr || ((load_word(vvalid, static_cast<int64_t>(r) >> 6) >> (r & 63)) & 1ull)) ? 1 : 0;
  }
}

template <typename A, bool PRODUCT>
__global__ __launch_bounds__(kBlock) void hash_fsum_walk_kernel(const uint32_t* __restrict__ gs, const A* __restrict__ vs,
                                                                const uint8_t* __restrict__ oks, int64_t n, A* __restrict__ sums,
                                                                long long* __restrict__ counts, uint32_t* __restrict__ null_seen,
                                                                unsigned long long* __restrict__ long_runs) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint32_t g = gs[i];
    if (i > 0 && gs[i - 1] == g) continue;   // not the first row of its group's run
    // a run of more than kFsumLongRun rows is left to a whole wave (hash_fsum_walk_long_kernel): long_runs[0] counts them
    if (i + kFsumLongRun < n && gs[i + kFsumLongRun] == g) {
      long_runs[1 + atomicAdd(&long_runs[0], 1ull)] = static_cast<unsigned long long>(i);
      continue;
    }
    A acc = sums[g];
    long long cnt = 0;
    bool saw_null = false;
    bool more = true;
    for (int64_t j = i; j < n && more; j += 4) {
      // four rows' loads in flight; the adds stay in row order
      uint32_t gg[4];
      A v[4];
      uint8_t ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t q = j + k < n ? j + k : n - 1;
        gg[k] = gs[q];
        v[k] = vs[q];
        ok[k] = oks[q];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!more || j + k >= n || gg[k] != g) {
          more = false;
        } else if (ok[k]) {
          acc = WalkOp<A, PRODUCT>::apply(acc, v[k]);
          ++cnt;
        } else {
          saw_null = true;
        }
      }
    }
    sums[g] = acc;
    if (cnt != 0) counts[g] += cnt;
    if (saw_null) null_seen[g] |= 1u;
  }
}

// The long runs (few groups, many rows each): the same additions in the same order, by ONE wave per run that streams it —
// kFsumAhead chunks of 64 rows in flight.  A row that does not count (null, or past the run's end) is replaced by -0.0, the
// identity of IEEE addition (x + -0.0 == x bit for bit, for every x), so the serial part is nothing but a chain of adds: the
// 64 values of a chunk go through LDS and come back as broadcast reads, all issued before the first add.  (Reading them lane by
// lane with v_readlane cost ~120 cycles a row, with ds_bpermute more: 37 / 46 ms for 6.7e5-row runs.)
constexpr int kFsumAhead = 8;
template <typename A, bool PRODUCT>
__global__ __launch_bounds__(64) void hash_fsum_walk_long_kernel(const uint32_t* __restrict__ gs, const A* __restrict__ vs,
                                                                 const uint8_t* __restrict__ oks, int64_t n, A* __restrict__ sums,
                                                                 long long* __restrict__ counts, uint32_t* __restrict__ null_seen,
                                                                 const unsigned long long* __restrict__ long_runs) {
  __shared__ A stage[kFsumAhead][64];
  const int lane = threadIdx.x;
  const int64_t nruns = static_cast<int64_t>(long_runs[0]);
  for (int64_t e = blockIdx.x; e < nruns; e += gridDim.x) {
    const int64_t i = static_cast<int64_t>(long_runs[1 + e]);
    const uint32_t g = gs[i];
    A acc = sums[g];
    long long cnt = 0;
    bool saw_null = false;
    bool more = true;
    // the next round's rows are loaded into registers before this round's chain of adds runs (the barrier in between would
    // otherwise keep the loads behind it)
    uint32_t gg[kFsumAhead];
    A vv[kFsumAhead];
    uint8_t oo[kFsumAhead];
    auto fetch = [&](int64_t j0) {
#pragma unroll
      for (int c = 0; c < kFsumAhead; ++c) {
        const int64_t q = j0 + c * 64 + lane < n ? j0 + c * 64 + lane : n - 1;
        gg[c] = gs[q];
        vv[c] = vs[q];
        oo[c] = oks[q];
      }
    };
    fetch(i);
    for (int64_t j = i; j < n && more; j += 64 * kFsumAhead) {
      uint64_t run_mask[kFsumAhead], ok_mask[kFsumAhead];
      __syncthreads();   // (one wave: the previous round's reads of the stage are done)
#pragma unroll
      for (int c = 0; c < kFsumAhead; ++c) {
        const bool in_run = j + c * 64 + lane < n && gg[c] == g;
        const bool ok = oo[c] != 0;
        run_mask[c] = __ballot(in_run);
        ok_mask[c] = __ballot(ok);
        stage[c][lane] = (in_run && ok) ? vv[c] : WalkOp<A, PRODUCT>::skip();
      }
      __syncthreads();
      if (j + 64 * kFsumAhead < n) fetch(j + 64 * kFsumAhead);
#pragma unroll
      for (int c = 0; c < kFsumAhead; ++c) {
        if (!more) continue;   // (wave-uniform)
        const int take = run_mask[c] == ~0ull ? 64 : __builtin_ctzll(~run_mask[c]);   // the run's rows are a prefix of the 64
        const uint64_t prefix = take == 64 ? ~0ull : ((1ull << take) - 1ull);
        // (a staged value past `take` is -0.0: the rows are sorted by group, the group does not come back)
        A x[64];
#pragma unroll
        for (int l = 0; l < 64; ++l) x[l] = stage[c][l];
#pragma unroll
        for (int l = 0; l < 64; ++l) acc = WalkOp<A, PRODUCT>::apply(acc, x[l]);
        cnt += __popcll(ok_mask[c] & prefix);
        saw_null = saw_null || ((~ok_mask[c] & prefix) != 0);
        if (take < 64) more = false;
      }
    }
    if (lane == 0) {
      sums[g] = acc;
      if (cnt != 0) counts[g] += cnt;
      if (saw_null) null_seen[g] |= 1u;
    }
  }
}

__global__ __launch_bounds__(kBlock) void hash_fsum_merge_kernel(double* __restrict__ sums, long long* __restrict__ counts,
                                                                 uint32_t* __restrict__ null_seen, const double* __restrict__ other_sums,
                                                                 const long long* __restrict__ other_counts,
                                                                 const uint32_t* __restrict__ other_null_seen,
                                                                 const uint32_t* __restrict__ mapping, int64_t m) {
  // Merge (:85-107): group g of the other state lands on mapping[g], each target at most once per call
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; g < m; g += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint32_t t = mapping[g];
    sums[t] = sums[t] + other_sums[g];
    counts[t] += other_counts[g];
    if (other_null_seen[g] & 1u) null_seen[t] |= 1u;
  }
}

__global__ __launch_bounds__(kBlock) void hash_fmean_finalize_kernel(const double* __restrict__ sums, const long long* __restrict__ counts,
                                                                     int64_t m, double* __restrict__ out) {
  // GroupedMeanImpl::DoMean (:381-385): double(reduced) / count; an empty group reads 0 (its slot is null)
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; g < m; g += static_cast<int64_t>(gridDim.x) * kBlock) {
    out[g] = counts[g] > 0 ? sums[g] / static_cast<double>(counts[g]) : 0.0;
  }
}

// ---- hash_variance / hash_stddev / hash_skew / hash_kurtosis: the second pass of the two-pass moments
// (GroupedStatisticImpl::ConsumeGeneric, kernels/hash_aggregate_numeric.cc:555-615: mean = sum / count, then the sums of
// (x - mean)^k).  out[i] = (x_i - mean of its group)^power for the valid rows (0 for the others: their validity bit keeps
// them out of the sum that follows).
template <typename T>
__global__ __launch_bounds__(kBlock) void group_central_power_kernel(const T* __restrict__ in, Bits valid, const uint32_t* __restrict__ ids,
                                                                     int64_t n, const double* __restrict__ sums,
                                                                     const long long* __restrict__ counts, int power,
                                                                     double* __restrict__ out) {
#pragma clang fp contract(off)   // (every product and difference rounded on its own, as the reference's scalar code)
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool ok = (load_word(valid, i >> 6) >> (i & 63)) & 1ull;
    double r = 0.0;
    if (ok) {
      const uint32_t g = ids[i];
      const double d = static_cast<double>(in[i]) - sums[g] / static_cast<double>(counts[g]);   // (a valid row: counts[g] >= 1)
      const double d2 = d * d;
      r = power == 2 ? d2 : power == 3 ? d2 * d : d2 * d2;
    }
    out[i] = r;
  }
}

// Moments::Variance / Stddev / Skew / Kurtosis (kernels/aggregate_var_std_internal.h:83-116) of (count, m2, m3, m4); a group
// the reference leaves null (GroupedStatisticImpl::Finalize :747-775) reads 0
__global__ __launch_bounds__(kBlock) void hash_moments_finalize_kernel(const long long* __restrict__ counts, const double* __restrict__ m2s,
                                                                       const double* __restrict__ m3s, const double* __restrict__ m4s,
                                                                       int64_t m, int stat, int ddof, int biased, double* __restrict__ out) {
#pragma clang fp contract(off)
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; g < m; g += static_cast<int64_t>(gridDim.x) * kBlock) {
    const long long c = counts[g];
    const double count = static_cast<double>(c);
    double r = 0.0;
    const bool defined = c > ddof && (stat != ARX_STAT_SKEW || biased || c > 2) && (stat != ARX_STAT_KURTOSIS || biased || c > 3);
    if (defined) {
      const double m2 = m2s[g];
      if (stat == ARX_STAT_VARIANCE || stat == ARX_STAT_STDDEV) {
        r = m2 / static_cast<double>(c - ddof);
        if (stat == ARX_STAT_STDDEV) r = sqrt(r);
      } else if (stat == ARX_STAT_SKEW) {
        const double m3 = m3s[g];
        if (biased) {
          r = sqrt(count) * m3 / sqrt(m2 * m2 * m2);
        } else {
          const double m2_avg = m2 / count;
          r = sqrt(count * (count - 1)) / (count - 2) * (m3 / count) / sqrt(m2_avg * m2_avg * m2_avg);
        }
      } else {
        const double m4 = m4s[g];
        if (biased) {
          r = count * m4 / (m2 * m2) - 3;
        } else {
          const double m2_avg = m2 / count;
          r = 1.0 / ((count - 2) * (count - 3)) * (((count * count) - 1.0) * (m4 / count) / (m2_avg * m2_avg) - 3 * ((count - 1) * (count - 1)));
        }
      }
    }
    out[g] = r;
  }
}

extern "C" {

int arx_group_central_power(const ArxSpan* values, int num_type, const uint32_t* group_ids, int64_t length, const double* sums,
                            const int64_t* counts, int power, double* out, void* stream) {
  if (values == nullptr || length < 0 || power < 2 || power > 4 || (num_type != ARX_NUM_FLOAT32 && num_type != ARX_NUM_FLOAT64)) {
    set_error("bad arguments to arx_group_central_power (float32 / float64 values, power 2 to 4)");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  if (values->data == nullptr || group_ids == nullptr || sums == nullptr || counts == nullptr || out == nullptr) {
    set_error("arx_group_central_power: NULL buffer");
    return ARX_INVALID;
  }
  const Bits valid = make_bits(values->null_count != 0 ? values->validity : nullptr, values->offset, length);
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((length + kBlock - 1) / kBlock, 256 * 16)));
  if (num_type == ARX_NUM_FLOAT32) {
    hipLaunchKernelGGL(group_central_power_kernel<float>, dim3(grid), dim3(kBlock), 0, as_stream(stream),
                       static_cast<const float*>(values->data) + values->offset, valid, group_ids, length, sums,
                       reinterpret_cast<const long long*>(counts), power, out);
  } else {
    hipLaunchKernelGGL(group_central_power_kernel<double>, dim3(grid), dim3(kBlock), 0, as_stream(stream),
                       static_cast<const double*>(values->data) + values->offset, valid, group_ids, length, sums,
                       reinterpret_cast<const long long*>(counts), power, out);
  }
  ARX_CHECK_LAUNCH("group_central_power_kernel");
  return ARX_OK;
}

int arx_hash_moments_finalize(const int64_t* counts, const double* m2, const double* m3, const double* m4, int64_t num_groups, int stat,
                              int ddof, int biased, double* out, void* stream) {
  if (num_groups < 0 || stat < ARX_STAT_VARIANCE || stat > ARX_STAT_KURTOSIS || ddof < 0) {
    set_error("bad arguments to arx_hash_moments_finalize");
    return ARX_INVALID;
  }
  if (num_groups == 0) return ARX_OK;
  if (counts == nullptr || m2 == nullptr || out == nullptr || (stat == ARX_STAT_SKEW && m3 == nullptr) || (stat == ARX_STAT_KURTOSIS && m4 == nullptr)) {
    set_error("arx_hash_moments_finalize: NULL buffer");
    return ARX_INVALID;
  }
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((num_groups + kBlock - 1) / kBlock, 256 * 8)));
  hipLaunchKernelGGL(hash_moments_finalize_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const long long*>(counts),
                     m2, m3, m4, num_groups, stat, ddof, biased, out);
  ARX_CHECK_LAUNCH("hash_moments_finalize_kernel");
  return ARX_OK;
}

int arx_reduce_float_minmax(const ArxSpan* values, int num_type, void* acc, void* stream) {
  if (values == nullptr || acc == nullptr || values->length < 0 || (num_type != ARX_NUM_FLOAT32 && num_type != ARX_NUM_FLOAT64)) {
    set_error("bad arguments to arx_reduce_float_minmax");
    return ARX_INVALID;
  }
  const int64_t n = values->length;
  if (n == 0) return ARX_OK;
  if (values->data == nullptr) {
    set_error("NULL data buffer passed to arx_reduce_float_minmax");
    return ARX_INVALID;
  }
  const Bits valid = make_bits(values->null_count != 0 ? values->validity : nullptr, values->offset, n);
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, kBlock * 8), int64_t(1) << 20)));
  if (num_type == ARX_NUM_FLOAT64) {
    hipLaunchKernelGGL(reduce_float_minmax_kernel<double>, dim3(grid), dim3(kBlock), 0, as_stream(stream),
                       static_cast<const double*>(values->data) + values->offset, valid, n, static_cast<unsigned long long*>(acc));
  } else {
    hipLaunchKernelGGL(reduce_float_minmax_kernel<float>, dim3(grid), dim3(kBlock), 0, as_stream(stream),
                       static_cast<const float*>(values->data) + values->offset, valid, n, static_cast<unsigned long long*>(acc));
  }
  ARX_CHECK_LAUNCH("reduce_float_minmax_kernel");
  return ARX_OK;
}

size_t arx_sum_float_workspace_bytes(int64_t length, int64_t null_count) {
  if (length <= 0) return 256;
  const int64_t nwords = ceil_div(length, 64), ntiles = ceil_div(nwords, kFsumTileWords);
  const int64_t blocks = fsum_max_blocks(length, null_count);
  size_t bytes = 256;                                            // totals
  if (null_count != 0) {
    bytes += fsum_align(nwords) + 2 * fsum_align(ntiles);        // word_carry, tile_last, tile_carry
    bytes += fsum_align(nwords * 4) + fsum_align(ntiles * 4) + fsum_align(ntiles * 8);
    bytes += fsum_align(blocks * 8);                             // the block sums
  }
  int64_t count = blocks;
  while (count > 0) {                                            // one output + one tail per pass
    bytes += fsum_align((count / kFsumGroup + 1) * 8) + fsum_align(kFsumGroup * 8);
    if (count < kFsumGroup) break;
    count /= kFsumGroup;
  }
  return bytes + 1024;
}

int arx_sum_float(const ArxSpan* values, int num_type, void* ws, size_t ws_bytes, double* out_sum, int64_t* out_count,
                  void* stream) {
  if (values == nullptr || out_sum == nullptr || out_count == nullptr || values->length < 0 ||
      (num_type != ARX_NUM_FLOAT32 && num_type != ARX_NUM_FLOAT64)) {
    set_error("bad arguments to arx_sum_float");
    return ARX_INVALID;
  }
  const int64_t n = values->length;
  *out_sum = 0;
  *out_count = 0;
  if (n == 0) return ARX_OK;
  const bool has_nulls = values->null_count != 0 && values->validity != nullptr;
  if (values->data == nullptr || ws == nullptr || ws_bytes < arx_sum_float_workspace_bytes(n, has_nulls ? values->null_count : 0)) {
    set_error("arx_sum_float: NULL buffer or a workspace below arx_sum_float_workspace_bytes");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  uint8_t* p = static_cast<uint8_t*>(ws);
  auto take = [&](size_t bytes) {
    uint8_t* r = p;
    p += fsum_align(bytes);
    return r;
  };
  int64_t* totals = reinterpret_cast<int64_t*>(take(256));
  const double* level = nullptr;   // the entries of the current level
  int64_t count = 0;
  // what the host gets: per pass its tail (level, entries), last the top array
  struct Piece { int level; const double* dev; int64_t n; };
  std::vector<Piece> pieces;
  int cur_level = 0;
  const int64_t max_blocks = fsum_max_blocks(n, has_nulls ? values->null_count : 0);
  if (!has_nulls) {
    count = ceil_div(n, kFsumBlock);
    const int64_t groups = ceil_div(count, kFsumGroup);
    double* out = reinterpret_cast<double*>(take((count / kFsumGroup + 1) * 8));
    double* tail = reinterpret_cast<double*>(take(kFsumGroup * 8));
    if (num_type == ARX_NUM_FLOAT64) {
      hipLaunchKernelGGL(fsum_leaf_dense_kernel<double>, dim3(static_cast<unsigned>(groups)), dim3(kFsumThreads), 0, st,
                         static_cast<const double*>(values->data) + values->offset, n, count, out, tail);
    } else {
      hipLaunchKernelGGL(fsum_leaf_dense_kernel<float>, dim3(static_cast<unsigned>(groups)), dim3(kFsumThreads), 0, st,
                         static_cast<const float*>(values->data) + values->offset, n, count, out, tail);
    }
    ARX_CHECK_LAUNCH("fsum_leaf_dense_kernel");
    if (count % kFsumGroup != 0) pieces.push_back({0, tail, count % kFsumGroup});
    level = out;
    count /= kFsumGroup;
    cur_level = kFsumLg;
    *out_count = n;
  } else {
    FsumScan a{};
    a.valid = make_bits(values->validity, values->offset, n);
    a.nwords = ceil_div(n, 64);
    a.ntiles = ceil_div(a.nwords, kFsumTileWords);
    a.word_carry = take(a.nwords);
    a.tile_last = take(a.ntiles);
    a.tile_carry = take(a.ntiles);
    a.word_blocks = reinterpret_cast<uint32_t*>(take(a.nwords * 4));
    a.tile_blocks = reinterpret_cast<uint32_t*>(take(a.ntiles * 4));
    a.tile_base = reinterpret_cast<int64_t*>(take(a.ntiles * 8));
    a.totals = totals;
    double* block_sums = reinterpret_cast<double*>(take(max_blocks * 8));
    ARX_HIP(hipMemsetAsync(totals, 0, 16, st));
    const unsigned tiles = static_cast<unsigned>(a.ntiles);
    hipLaunchKernelGGL(fsum_word_carry_kernel, dim3(tiles), dim3(kFsumTileWords), 0, st, a);
    hipLaunchKernelGGL(fsum_tile_carry_kernel, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(fsum_word_blocks_kernel, dim3(tiles), dim3(kFsumTileWords), 0, st, a);
    hipLaunchKernelGGL(fsum_tile_base_kernel, dim3(1), dim3(1024), 0, st, a);
    if (num_type == ARX_NUM_FLOAT64) {
      hipLaunchKernelGGL(fsum_block_sums_kernel<double>, dim3(tiles), dim3(kFsumTileWords), 0, st, a,
                         static_cast<const double*>(values->data) + values->offset, block_sums);
    } else {
      hipLaunchKernelGGL(fsum_block_sums_kernel<float>, dim3(tiles), dim3(kFsumTileWords), 0, st, a,
                         static_cast<const float*>(values->data) + values->offset, block_sums);
    }
    ARX_CHECK_LAUNCH("fsum_block_sums_kernel");
    int64_t h[2];
    ARX_HIP(hipMemcpyAsync(h, totals, 16, hipMemcpyDeviceToHost, st));
    ARX_HIP(hipStreamSynchronize(st));
    if (h[0] < 0 || h[0] > max_blocks) {
      set_error("arx_sum_float: %lld blocks for a workspace of %lld (null_count understated?)", static_cast<long long>(h[0]),
                static_cast<long long>(max_blocks));
      return ARX_INVALID;
    }
    count = h[0];
    *out_count = h[1];
    level = block_sums;
    cur_level = 0;
  }
  while (count >= kFsumGroup) {
    const int64_t groups = ceil_div(count, kFsumGroup);
    double* out = reinterpret_cast<double*>(take((count / kFsumGroup + 1) * 8));
    double* tail = reinterpret_cast<double*>(take(kFsumGroup * 8));
    hipLaunchKernelGGL(fsum_tree_kernel, dim3(static_cast<unsigned>(groups)), dim3(kFsumThreads), 0, st, level, count, out, tail);
    ARX_CHECK_LAUNCH("fsum_tree_kernel");
    if (count % kFsumGroup != 0) pieces.push_back({cur_level, tail, count % kFsumGroup});
    level = out;
    count /= kFsumGroup;
    cur_level += kFsumLg;
  }
  if (count > 0) pieces.push_back({cur_level, level, count});
  // block order: the top array's entries come first, then the tails from the highest level down
  std::vector<std::vector<double>> host(pieces.size());
  for (size_t i = 0; i < pieces.size(); ++i) {
    host[i].resize(static_cast<size_t>(pieces[i].n));
    ARX_HIP(hipMemcpyAsync(host[i].data(), pieces[i].dev, static_cast<size_t>(pieces[i].n) * 8, hipMemcpyDeviceToHost, st));
  }
  ARX_HIP(hipStreamSynchronize(st));
  if (*out_count == 0) return ARX_OK;   // (data_size == 0: the sum is 0, :164-166)
  PairwiseCounter counter;
  for (size_t i = pieces.size(); i-- > 0;) {
    for (double v : host[i]) counter.reduce(pieces[i].level, v);
  }
  *out_sum = counter.finish();
  return ARX_OK;
}

size_t arx_hash_sum_float_workspace_bytes(int64_t length) {
  if (length <= 0) return 0;
  const size_t n = static_cast<size_t>(length);
  return fsum_align(arx_sort_indices_workspace_bytes(length)) + fsum_align(n * 8) /* perm */ + fsum_align(n * 4) /* group ids */ +
         fsum_align(n * 8) /* values */ + fsum_align(n) /* validity */ + fsum_align((n / kFsumLongRun + 2) * 8) + 512;
}

int arx_hash_sum_float_consume(const ArxSpan* values, int num_type, int values_is_scalar, double scalar_value,
                               const uint32_t* group_ids, int64_t length, void* ws, size_t ws_bytes, double* sums,
                               int64_t* counts, uint32_t* null_seen, void* stream) {
  if (values == nullptr || length < 0 || (num_type != ARX_NUM_FLOAT32 && num_type != ARX_NUM_FLOAT64)) {
    set_error("bad arguments to arx_hash_sum_float_consume");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  if (group_ids == nullptr || sums == nullptr || counts == nullptr || null_seen == nullptr || ws == nullptr ||
      ws_bytes < arx_hash_sum_float_workspace_bytes(length) || (!values_is_scalar && values->data == nullptr)) {
    set_error("arx_hash_sum_float_consume: NULL buffer or a workspace below arx_hash_sum_float_workspace_bytes");
    return ARX_INVALID;
  }
  const size_t n = static_cast<size_t>(length);
  uint8_t* p = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~uintptr_t(255));
  const size_t sort_bytes = arx_sort_indices_workspace_bytes(length);
  uint8_t* q = p + fsum_align(sort_bytes);
  uint64_t* perm = reinterpret_cast<uint64_t*>(q); q += fsum_align(n * 8);
  uint32_t* gs = reinterpret_cast<uint32_t*>(q); q += fsum_align(n * 4);
  double* vs = reinterpret_cast<double*>(q); q += fsum_align(n * 8);
  uint8_t* oks = q; q += fsum_align(n);
  unsigned long long* long_runs = reinterpret_cast<unsigned long long*>(q);
  const ArxSpan keys{nullptr, group_ids, 0, length, 0};
  const int rc = arx_sort_indices(&keys, ARX_KEY_UINT32, ARX_SORT_ASCENDING, ARX_NULLS_AT_END, p, sort_bytes, perm, stream);
  if (rc != ARX_OK) return rc;
  hipStream_t st = as_stream(stream);
  ARX_HIP(hipMemsetAsync(long_runs, 0, 8, st));
  const bool has_nulls = !values_is_scalar && values->null_count != 0 && values->validity != nullptr;
  const Bits vvalid = has_nulls ? make_bits(values->validity, values->offset, length) : Bits{};
  const int scalar_valid = values->null_count == 0 ? 1 : 0;
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((length + kBlock - 1) / kBlock, 256 * 16)));
  if (num_type == ARX_NUM_FLOAT64) {
    const double* v = values_is_scalar ? nullptr : static_cast<const double*>(values->data) + values->offset;
    hipLaunchKernelGGL((hash_fsum_gather_kernel<double>), dim3(grid), dim3(kBlock), 0, st, v, vvalid, values_is_scalar, scalar_value,
                       scalar_valid, group_ids, perm, length, gs, vs, oks);
  } else {
    const float* v = values_is_scalar ? nullptr : static_cast<const float*>(values->data) + values->offset;
    hipLaunchKernelGGL((hash_fsum_gather_kernel<float>), dim3(grid), dim3(kBlock), 0, st, v, vvalid, values_is_scalar, scalar_value,
                       scalar_valid, group_ids, perm, length, gs, vs, oks);
  }
  ARX_CHECK_LAUNCH("hash_fsum_gather_kernel");
  hipLaunchKernelGGL((hash_fsum_walk_kernel<double, false>), dim3(grid), dim3(kBlock), 0, st, gs, static_cast<const double*>(vs), oks, length, sums,
                     reinterpret_cast<long long*>(counts), null_seen, long_runs);
  ARX_CHECK_LAUNCH("hash_fsum_walk_kernel");
  if (length > kFsumLongRun) {
    hipLaunchKernelGGL((hash_fsum_walk_long_kernel<double, false>), dim3(256 * 8), dim3(64), 0, st, gs, static_cast<const double*>(vs), oks, length,
                       sums, reinterpret_cast<long long*>(counts), null_seen, long_runs);
    ARX_CHECK_LAUNCH("hash_fsum_walk_long_kernel");
  }
  return ARX_OK;
}

// hash_product — GroupedProductImpl (kernels/hash_aggregate_numeric.cc:311-347): per group the product of the valid values IN
// ROW ORDER from 1, in the accumulator type of the sum (int64 / uint64 wrapping, double).  Integer products do not depend on
// the order, products of doubles do (and of two NaNs: the first payload wins), so every type takes the walkers above: rows
// stably sorted by group id, one owner per group.  `products`: 8 bytes per group, filled with 1 / 1.0 by
// arx_hash_product_init before the first batch; counts / null_seen as for the sums (the same finalize).
int arx_hash_product_init(void* products, int num_type, int64_t num_groups, void* stream) {
  if (num_groups <= 0) return ARX_OK;
  if (products == nullptr) {
    set_error("arx_hash_product_init: NULL buffer");
    return ARX_INVALID;
  }
  const bool is_float = num_type == ARX_NUM_FLOAT32 || num_type == ARX_NUM_FLOAT64;
  const double one = 1.0;
  uint64_t bits = 1;
  if (is_float) memcpy(&bits, &one, 8);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(fill_u64_kernel, dim3(static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((num_groups + kBlock - 1) / kBlock, 4096)))),
                     dim3(kBlock), 0, st, static_cast<uint64_t*>(products), bits, num_groups);
  ARX_CHECK_LAUNCH("fill_u64_kernel");
  return ARX_OK;
}

int arx_hash_product_consume(const ArxSpan* values, int num_type, const uint32_t* group_ids, int64_t length, void* ws, size_t ws_bytes,
                             void* products, int64_t* counts, uint32_t* null_seen, void* stream) {
  if (values == nullptr || length < 0 || num_type < ARX_NUM_INT8 || num_type > ARX_NUM_FLOAT64) {
    set_error("bad arguments to arx_hash_product_consume");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  if (group_ids == nullptr || products == nullptr || counts == nullptr || null_seen == nullptr || ws == nullptr ||
      ws_bytes < arx_hash_sum_float_workspace_bytes(length) || values->data == nullptr) {
    set_error("arx_hash_product_consume: NULL buffer or a workspace below arx_hash_sum_float_workspace_bytes");
    return ARX_INVALID;
  }
  const size_t n = static_cast<size_t>(length);
  uint8_t* p = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~uintptr_t(255));
  const size_t sort_bytes = arx_sort_indices_workspace_bytes(length);
  uint8_t* q = p + fsum_align(sort_bytes);
  uint64_t* perm = reinterpret_cast<uint64_t*>(q); q += fsum_align(n * 8);
  uint32_t* gs = reinterpret_cast<uint32_t*>(q); q += fsum_align(n * 4);
  void* vs = q; q += fsum_align(n * 8);
  uint8_t* oks = q; q += fsum_align(n);
  unsigned long long* long_runs = reinterpret_cast<unsigned long long*>(q);
  const ArxSpan keys{nullptr, group_ids, 0, length, 0};
  const int rc = arx_sort_indices(&keys, ARX_KEY_UINT32, ARX_SORT_ASCENDING, ARX_NULLS_AT_END, p, sort_bytes, perm, stream);
  if (rc != ARX_OK) return rc;
  hipStream_t st = as_stream(stream);
  ARX_HIP(hipMemsetAsync(long_runs, 0, 8, st));
  const bool has_nulls = values->null_count != 0 && values->validity != nullptr;
  const Bits vvalid = has_nulls ? make_bits(values->validity, values->offset, length) : Bits{};
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((length + kBlock - 1) / kBlock, 256 * 16)));
  const bool is_float = num_type == ARX_NUM_FLOAT32 || num_type == ARX_NUM_FLOAT64;
#define ARX_IPROD_GATHER(T)                                                                                                            \
  hipLaunchKernelGGL((hash_iprod_gather_kernel<T>), dim3(grid), dim3(kBlock), 0, st, static_cast<const T*>(values->data) + values->offset, vvalid, \
                     group_ids, perm, length, gs, static_cast<uint64_t*>(vs), oks)
  switch (num_type) {
    case ARX_NUM_INT8: ARX_IPROD_GATHER(int8_t); break;
    case ARX_NUM_UINT8: ARX_IPROD_GATHER(uint8_t); break;
    case ARX_NUM_INT16: ARX_IPROD_GATHER(int16_t); break;
    case ARX_NUM_UINT16: ARX_IPROD_GATHER(uint16_t); break;
    case ARX_NUM_INT32: ARX_IPROD_GATHER(int32_t); break;
    case ARX_NUM_UINT32: ARX_IPROD_GATHER(uint32_t); break;
    case ARX_NUM_INT64: ARX_IPROD_GATHER(int64_t); break;
    case ARX_NUM_UINT64: ARX_IPROD_GATHER(uint64_t); break;
    case ARX_NUM_FLOAT32:
      hipLaunchKernelGGL((hash_fsum_gather_kernel<float>), dim3(grid), dim3(kBlock), 0, st, static_cast<const float*>(values->data) + values->offset,
                         vvalid, 0, 0.0, 1, group_ids, perm, length, gs, static_cast<double*>(vs), oks);
      break;
    default:
      hipLaunchKernelGGL((hash_fsum_gather_kernel<double>), dim3(grid), dim3(kBlock), 0, st, static_cast<const double*>(values->data) + values->offset,
                         vvalid, 0, 0.0, 1, group_ids, perm, length, gs, static_cast<double*>(vs), oks);
      break;
  }
#undef ARX_IPROD_GATHER
  ARX_CHECK_LAUNCH("hash product gather kernel");
  if (is_float) {
    hipLaunchKernelGGL((hash_fsum_walk_kernel<double, true>), dim3(grid), dim3(kBlock), 0, st, gs, static_cast<const double*>(vs), oks, length,
                       static_cast<double*>(products), reinterpret_cast<long long*>(counts), null_seen, long_runs);
  } else {
    hipLaunchKernelGGL((hash_fsum_walk_kernel<uint64_t, true>), dim3(grid), dim3(kBlock), 0, st, gs, static_cast<const uint64_t*>(vs), oks, length,
                       static_cast<uint64_t*>(products), reinterpret_cast<long long*>(counts), null_seen, long_runs);
  }
  ARX_CHECK_LAUNCH("hash_fsum_walk_kernel");
  if (length > kFsumLongRun) {
    if (is_float) {
      hipLaunchKernelGGL((hash_fsum_walk_long_kernel<double, true>), dim3(256 * 8), dim3(64), 0, st, gs, static_cast<const double*>(vs), oks, length,
                         static_cast<double*>(products), reinterpret_cast<long long*>(counts), null_seen, long_runs);
    } else {
      hipLaunchKernelGGL((hash_fsum_walk_long_kernel<uint64_t, true>), dim3(256 * 8), dim3(64), 0, st, gs, static_cast<const uint64_t*>(vs), oks, length,
                         static_cast<uint64_t*>(products), reinterpret_cast<long long*>(counts), null_seen, long_runs);
    }
    ARX_CHECK_LAUNCH("hash_fsum_walk_long_kernel");
  }
  return ARX_OK;
}

int arx_hash_sum_f64_merge(double* sums, int64_t* counts, uint32_t* null_seen, const double* other_sums, const int64_t* other_counts,
                           const uint32_t* other_null_seen, const uint32_t* group_id_mapping, int64_t other_num_groups, void* stream) {
  if (other_num_groups < 0) {
    set_error("bad arguments to arx_hash_sum_f64_merge");
    return ARX_INVALID;
  }
  if (other_num_groups == 0) return ARX_OK;
  if (sums == nullptr || counts == nullptr || null_seen == nullptr || other_sums == nullptr || other_counts == nullptr ||
      other_null_seen == nullptr || group_id_mapping == nullptr) {
    set_error("arx_hash_sum_f64_merge: NULL buffer");
    return ARX_INVALID;
  }
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((other_num_groups + kBlock - 1) / kBlock, 256 * 8)));
  hipLaunchKernelGGL(hash_fsum_merge_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), sums, reinterpret_cast<long long*>(counts),
                     null_seen, other_sums, reinterpret_cast<const long long*>(other_counts), other_null_seen, group_id_mapping,
                     other_num_groups);
  ARX_CHECK_LAUNCH("hash_fsum_merge_kernel");
  return ARX_OK;
}

int arx_hash_mean_f64_finalize(const double* sums, const int64_t* counts, int64_t num_groups, double* out_means, void* stream) {
  if (num_groups < 0 || (num_groups > 0 && (sums == nullptr || counts == nullptr || out_means == nullptr))) {
    set_error("bad arguments to arx_hash_mean_f64_finalize");
    return ARX_INVALID;
  }
  if (num_groups == 0) return ARX_OK;
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((num_groups + kBlock - 1) / kBlock, 256 * 8)));
  hipLaunchKernelGGL(hash_fmean_finalize_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), sums,
                     reinterpret_cast<const long long*>(counts), num_groups, out_means);
  ARX_CHECK_LAUNCH("hash_fmean_finalize_kernel");
  return ARX_OK;
}

}  // extern "C"

}  // namespace arx
