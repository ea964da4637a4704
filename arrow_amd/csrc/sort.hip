// array_sort_indices for 64-bit integer keys on gfx950: stable LSB radix sort,
// 8 passes x 8-bit digits over (key, 32-bit row id) pairs.
//
// What it restates (semantics only):
//   ArraySortIndices<UInt64Type,*>::Exec       cpp/src/arrow/compute/kernels/vector_array_sort.cc:524-540
//   ArrayCompareSorter (std::stable_sort)       same file :144-178
//   ArrayCountOrCompareSorter                   same file :404-446  (both branches are stable and
//                                               give the same permutation)
//   PartitionNullsOnly<StablePartitioner>       cpp/src/arrow/compute/kernels/vector_sort_internal.h:225-293
// The result is the unique stable argsort: ascending by key with ties in row order;
// descending = ascending on ~key (ties still in row order, as `rhs < lhs` + stable_sort gives);
// nulls keep row order and sit at the end or the start.
//
// Per pass (digit = bits [shift, shift+8)):
//   radix_hist_kernel    : one workgroup per chunk of tiles -> hist[digit][chunk]
//   radix_scan_kernel    : exclusive scan of hist in digit-major order (single workgroup)
//   radix_scatter_kernel : per tile of 4096 keys: wave-level multi-split ranking (8 ballots per
//                          key), workgroup scan of the 4 x 256 wave counters, reorder through LDS,
//                          then coalesced runs to the destination.
// HBM traffic per row per pass: 8 (hist) + 12 (read) + 12 (write) bytes.
#include "arx_common.h"

#include <string.h>

#include <algorithm>
#include <atomic>
#include <type_traits>
#include <vector>

namespace arx {

constexpr int kSortItems = 16;
constexpr int kSortTile = kBlock * kSortItems;  // 4096 keys
constexpr int kDigits = 256;
constexpr int kMaxChunks = 16384;  // upper bound (sizes the histogram table)
static Knob<int> g_sort_msd{-1};            // -1 = auto (n >= g_sort_msd_min_rows), 0 = never, 1 = whenever possible
static Knob<int> g_sort_msd_min_rows{1 << 22};
static Knob<int> g_sort_msd_sampled{1};     // skewed keys: bucket boundaries from a sorted sample
static Knob<int> g_sort_msd_fused{1};       // finish LDS-sized level-2 buckets in one workgroup (msd_bucket_kernel)
static Knob<int64_t> g_sort_msd_segment_rows{int64_t(1) << 27};  // above this: an extra top-bits level cuts segments
static Knob<int> g_sort_msd_final_rows_log2{1};  // log2 of the rows aimed at per final sub-bucket (rank loop length); with the 4096-bin finish 1 beats 2 / 3 / 4 by 3 / 9 / 18 % at 2e9 rows
static Knob<int> g_sort_msd_small_bucket{1};  // 512-thread / 5120-row bucket kernel when every bucket fits it
static Knob<int> g_sort_msd_wide_sample_shift{6};  // wide form: level-1 capacities from a histogram of 1 tile in 2^shift (0 = exact histogram of every row)
static Knob<int> g_sort_xcd_map{1};                // wide form, XCD-contiguous work numbering: bit 0 level 2 (-2.1 ms at 2e9 rows: a bucket's runs meet in one L2), bit 1 bucket finish, bit 2 level 1 (both: no effect)
static Knob<int> g_sort_msd_wide_bits{0};           // wide form: partition bits (0 = from the row count: buckets of 2048..4096 rows; tests force many bins on few rows)
static Knob<int> g_sort_msd_wide_b2max{11};        // wide form: most partition bits given to level 2 (<= 12; 0 = the even split).  2e9 rows, 20 bits: 9 + 11 with 16-row level-2 tiles 31.4 ms, 10 + 10 with 8-row tiles 34.2, 8 + 12 37.2 (profiles/r03_h, r03_i); 2^28 rows: 10 and 11 within 2 %
static Knob<int> g_sort_msd_wide_rpt1{24};         // wide form, level 1: rows per thread of a scatter tile (8: the tile lives in LDS; 16 / 24: in registers, moved through LDS in rounds)
static Knob<int> g_sort_msd_wide_rpt2{16};          // level 2 likewise
static Knob<int> g_sort_msd_tiny_bucket{2};        // wide form: 256-thread / 2560-row bucket finish when every bucket fits it
static Knob<int> g_sort_msd_bucket_cpt{4};         // wide form's bucket finish: sub-bucket counters per thread (4: <= 4 T sub-buckets, 8: <= 8 T)
static Knob<int> g_sort_msd_prefix{1};             // MSD forms take their digits below the bits that ALL keys share (ids, timestamps, small ints: the top bits are equal)
static Knob<int> g_sort_msd_wide_gap2{1};          // wide form: level-2 buckets get a fixed room each (bucket mean + 6 sigma + 64) instead of an exact histogram pass
static Knob<int> g_sort_msd_wide_rec8{1};           // wide form over the caller's own column: 8-byte {32 key bits below the level-1 digit, row id} records through both levels and the finish (48.5 B/row instead of 64.5); rows whose 32 bits tie read their full keys from the column
static Knob<int> g_sort_msd_wide_rec8_tie_shift{4};  // ... given up (and repeated with 12-byte records) once more than (rows >> shift) rows of ONE bucket tied (duplicate-heavy keys: every tie is two random 8-byte reads; 2e9 uniform keys: 0.9 per 1000)
constexpr int kMsdwWcRDefault = 4;
static Knob<int> g_sort_msd_wide_wc{256};          // rec8 form: level 1 write-combined by this many persistent workgroups (0 = the tile-at-a-time level 1)
static Knob<int> g_sort_records_in_place{1};       // arx_sort_records: the wide form's level 1 reads the records themselves (0: split into key / row arrays first; A/B knob sort_records_in_place)
static Knob<int> g_sort_vary_sample_shift{6};      // the shared-prefix probe reads one tile of 8192 rows in 2^shift first (knob sort_vary_sample_shift; round 6: 4 -> 6, 0.34 -> 0.22 ms at 2e9 rows)
static Knob<int> g_sort_msd_wide_wc_typed{1};      // the append kernel compiled for the key type (uint64 / int64 / double); 0: the run-time switch (A/B knob sort_msd_wide_wc_typed)
static Knob<int> g_sort_msd_wide_wc_form{2};       // 2: round 6's append kernel (every store a whole line); 1: round 5's rank-and-stage kernel (A/B knob sort_msd_wide_wc_form)
static Knob<int> g_sort_msd_wide_wc_rows{kMsdwWcRDefault};   // form 2: rows per thread and batch, 4 / 6 / 8 (A/B knob sort_msd_wide_wc_rows)
static Knob<int> g_sort_msd_wide_wc_min_rows{1 << 17};   // form 2: rows a persistent workgroup must have (fewer workgroups for small inputs; knob sort_msd_wide_wc_min_rows — tests)
static Knob<int> g_sort_msd_wide_wc_prefetch{1};   // ... with 16-row tiles and the next tile's keys requested before the current one's words leave
static Knob<int> g_sort_msd_wide_l2w{3};           // rec8 form, level 2 in small workgroups (msdw_scatter2w_kernel): 1 = 512 threads x 16 rows, 4096-word stage (2 per CU); 2 = 512 x 16, 2048-word stage (3); 3 = 1024 x 8, 4096 (2); 0 = msdw_scatter2_kernel
static Knob<int> g_sort_msd_wide_sample_strict{0}; // tests: a sampled attempt that overflows is an error instead of a silent exact re-run
static Knob<int> g_sort_msd_wide{1};          // inputs beyond sort_msd_segment_rows: the wide two-level form (run_msd_sort_wide) before the segmented one
static Knob<int> g_sort_msd_bucket_v2{1};     // single-atomic-pass bucket finish with up to 4096 sub-buckets (msd_bucket2_kernel)
static Knob<int> g_sort_msd_seg_min_bits{1};  // floor of the segment level's bits (more bins = fewer LDS atomic collisions)
static Knob<int> g_sort_msd_global_bits{14};  // (= kMsdMaxBits) cap of the two global levels (tests lower it to reach level 3)
static Knob<int> g_sort_fuse_prep{1};    // first pass reads the caller's column directly (no prep pass)
static Knob<int> g_sort_chunks{2048};    // chunks actually used (arx_set_option "sort_chunks")

// Provided by selection.hip: ascending row numbers of the set (or clear) bits of a bitmap.
int selection_bit_positions(const void* bitmap, int64_t bit_offset, int64_t length, bool invert,
                            void* ws, size_t ws_bytes, uint32_t* out, int64_t* out_count_host,
                            hipStream_t st);
size_t selection_workspace_bytes(int64_t length);

// Order transform: any supported key -> uint64 whose unsigned order is the requested order.
// `xf` = kXfRaw | (descending ? kXfDesc : 0) | key_type << 4   (key_type: ARX_KEY_* of the C ABI).
// Signed integers flip the sign bit; 32-bit keys sit in the top half; floats use the usual
// sign-magnitude trick with -0.0 canonicalised to +0.0 (they compare equal in the reference, so
// they must tie); NaNs never get here (they are partitioned next to the nulls first).
// Descending sorts ~key, which keeps ties in row order.
constexpr int kXfRaw = 1;
constexpr int kXfDesc = 4;
__host__ __device__ __forceinline__ int make_xf(int key_type, bool descending) {
  return kXfRaw | (descending ? kXfDesc : 0) | (key_type << 4);
}
__device__ __forceinline__ uint64_t load_key_typed(const void* base, int64_t i, int xf) {
  uint64_t k;
  switch ((xf >> 4) & 7) {
    case 0: k = static_cast<const uint64_t*>(base)[i]; break;
    case 1: k = static_cast<const uint64_t*>(base)[i] ^ 0x8000000000000000ull; break;
    case 2: k = static_cast<uint64_t>(static_cast<const uint32_t*>(base)[i]) << 32; break;
    case 3: k = static_cast<uint64_t>(static_cast<const uint32_t*>(base)[i] ^ 0x80000000u) << 32; break;
    case 4: {
      uint64_t b = static_cast<const uint64_t*>(base)[i];
      if ((b << 1) == 0) b = 0;
      k = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
      break;
    }
    default: {
      uint32_t b = static_cast<const uint32_t*>(base)[i];
      if ((b << 1) == 0) b = 0;
      const uint32_t t = (b >> 31) ? ~b : (b | 0x80000000u);
      k = static_cast<uint64_t>(t) << 32;
      break;
    }
  }
  if (xf & kXfDesc) k = ~k;
  return k;
}
// the same transform applied to bits already loaded (64-bit types: the value; 32-bit types: the value zero-extended)
__device__ __forceinline__ uint64_t key_from_bits(uint64_t b, int xf) {
  uint64_t k;
  switch ((xf >> 4) & 7) {
    case 0: k = b; break;
    case 1: k = b ^ 0x8000000000000000ull; break;
    case 2: k = b << 32; break;
    case 3: k = (b ^ 0x80000000ull) << 32; break;
    case 4: {
      if ((b << 1) == 0) b = 0;
      k = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
      break;
    }
    default: {
      uint32_t w = static_cast<uint32_t>(b);
      if ((w << 1) == 0) w = 0;
      const uint32_t t = (w >> 31) ? ~w : (w | 0x80000000u);
      k = static_cast<uint64_t>(t) << 32;
      break;
    }
  }
  if (xf & kXfDesc) k = ~k;
  return k;
}
__device__ __forceinline__ bool key_type_is_64bit(int xf) {
  const int kt = (xf >> 4) & 7;
  return kt == 0 || kt == 1 || kt == 4;
}
// legacy form for the 64-bit-only multi-GPU helpers
__device__ __forceinline__ uint64_t key_transform(uint64_t k, bool is_signed, bool descending) {
  if (is_signed) k ^= 0x8000000000000000ull;
  if (descending) k = ~k;
  return k;
}

// One wave per 64 rows: bit = valid && !isnan (the rows that are sorted) / valid && isnan.
__global__ __launch_bounds__(kBlock) void sort_float_classify_kernel(const void* __restrict__ values,
                                                                     int is_f32, Bits valid, int64_t n,
                                                                     uint64_t* __restrict__ sortable_bits,
                                                                     uint64_t* __restrict__ nan_bits) {
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwords = (n + 63) >> 6;
  for (int64_t w = wave_g; w < nwords; w += nwaves) {
    const int64_t i = (w << 6) + lane;
    bool ok = false, nan = false;
    if (i < n) {
      ok = (load_word(valid, w) >> lane) & 1ull;
      if (is_f32) {
        const uint32_t b = static_cast<const uint32_t*>(values)[i];
        nan = (b & 0x7fffffffu) > 0x7f800000u;
      } else {
        const uint64_t b = static_cast<const uint64_t*>(values)[i];
        nan = (b & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
      }
    }
    const uint64_t s_bits = __ballot(ok && !nan);
    const uint64_t n_bits = __ballot(ok && nan);
    if (lane == 0) {
      sortable_bits[w] = s_bits;
      nan_bits[w] = n_bits;
    }
  }
}

// keys_out[i] = transform(values[rows[i]]) , idx_out[i] = rows[i]  (rows == NULL -> identity)
__global__ __launch_bounds__(kBlock) void sort_prep_kernel(const void* __restrict__ values,
                                                           const uint32_t* __restrict__ rows,
                                                           int64_t n, int xf, int /*unused*/,
                                                           uint64_t* __restrict__ keys_out,
                                                           uint32_t* __restrict__ idx_out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t r = rows ? rows[i] : static_cast<uint32_t>(i);
    keys_out[i] = load_key_typed(values, r, xf);
    idx_out[i] = r;
  }
}

__global__ __launch_bounds__(kBlock) void widen_kernel(const uint32_t* __restrict__ in, int64_t n,
                                                       uint64_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    out[i] = in[i];
  }
}

// raw: 0 = keys are already order-transformed; else bit0 set = raw column values, transformed on
// the fly (bit1 = signed, bit2 = descending) — the first pass reads the caller's array directly.
__device__ __forceinline__ uint64_t load_sort_key(const uint64_t* __restrict__ keys, int64_t i, int raw) {
  return raw ? load_key_typed(keys, i, raw) : keys[i];
}

__global__ __launch_bounds__(kBlock) void radix_hist_kernel(const uint64_t* __restrict__ keys,
                                                            int64_t n, int shift,
                                                            int64_t chunk_keys, int64_t nchunks,
                                                            uint32_t* __restrict__ hist, int raw) {
  __shared__ uint32_t h[kDigits];
  const int64_t chunk = blockIdx.x;
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t begin = chunk * chunk_keys;
  const int64_t end = begin + chunk_keys < n ? begin + chunk_keys : n;
  for (int64_t i = begin + threadIdx.x; i < end; i += kBlock) {
    const uint32_t d = static_cast<uint32_t>(load_sort_key(keys, i, raw) >> shift) & 255u;
    atomicAdd(&h[d], 1u);
  }
  __syncthreads();
  hist[static_cast<int64_t>(threadIdx.x) * nchunks + chunk] = h[threadIdx.x];
}

// In-place exclusive scan of the 256 x nchunks counters (digit-major), two small kernels:
//   radix_digit_totals_kernel : one workgroup per digit -> total of its row of counters
//   radix_scan_kernel         : one workgroup per digit: exclusive scan of the 256 totals (every
//                               workgroup redoes it, 256 values) + coalesced scan of its own row.
__global__ __launch_bounds__(64) void radix_digit_totals_kernel(const uint32_t* __restrict__ hist,
                                                                 int64_t nchunks,
                                                                 uint32_t* __restrict__ totals) {
  const int lane = threadIdx.x;
  const uint32_t* row = hist + static_cast<int64_t>(blockIdx.x) * nchunks;
  uint32_t s = 0;
  for (int64_t i = lane; i < nchunks; i += 64) s += row[i];
  s = wave_reduce_sum_u32(s);
  if (lane == 0) totals[blockIdx.x] = s;
}

// one wave per digit
__global__ __launch_bounds__(64) void radix_scan_kernel(uint32_t* __restrict__ hist, int64_t nchunks,
                                                        const uint32_t* __restrict__ totals) {
  const int lane = threadIdx.x;
  // start of this digit = sum of the totals of the smaller digits
  uint32_t t = 0;
  for (int d = lane; d < static_cast<int>(blockIdx.x); d += 64) t += totals[d];
  uint32_t carry = wave_reduce_sum_u32(t);
  uint32_t* row = hist + static_cast<int64_t>(blockIdx.x) * nchunks;
  for (int64_t base = 0; base < nchunks; base += 64) {
    const int64_t i = base + lane;
    const uint32_t x = i < nchunks ? row[i] : 0u;
    const uint32_t incl = wave_inclusive_scan_u32(x);
    if (i < nchunks) row[i] = carry + incl - x;
    carry += __shfl(incl, 63, 64);
  }
}

struct __attribute__((aligned(16))) SortLds {
  uint64_t keys[kSortTile];
  uint32_t idx[kSortTile];
  uint32_t wave_cnt[kWavesPerBlock][kDigits];
  uint32_t cursor[kDigits];
  uint32_t wave_tot[kWavesPerBlock];
};

// last_pass: write idx widened to uint64 into out_final (keys are no longer needed)
// RAW: first pass over the caller's column (transform on load, row id = position); `raw` then carries
// the transform flags (bit1 = signed, bit2 = descending).
template <bool RAW>
__global__ __launch_bounds__(kBlock, 3) void radix_scatter_kernel(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ idx_in, int64_t n, int shift,
    int64_t chunk_tiles, int64_t nchunks, const uint32_t* __restrict__ hist_scanned,
    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out, uint64_t* __restrict__ out_final,
    int last_pass, int raw) {
  __shared__ SortLds lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t chunk = blockIdx.x;
  lds.cursor[tid] = hist_scanned[static_cast<int64_t>(tid) * nchunks + chunk];
  __syncthreads();

  const int64_t tile_begin = chunk * chunk_tiles;
  const int64_t ntiles_total = (n + kSortTile - 1) / kSortTile;
  const int64_t tile_end =
      tile_begin + chunk_tiles < ntiles_total ? tile_begin + chunk_tiles : ntiles_total;

  for (int64_t tile = tile_begin; tile < tile_end; ++tile) {
    const int64_t base = tile * kSortTile;
    const int64_t remain = n - base;
    const int nv = remain < kSortTile ? static_cast<int>(remain) : kSortTile;

    // 1. load: wave-striped so that (wave, item, lane) order == row order (stability)
    uint64_t key[kSortItems];
    uint32_t idx[kSortItems];
    uint32_t rank[kSortItems];
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
      const int p = wave * (kSortItems * 64) + i * 64 + lane;
      if (p < nv) {
        if constexpr (RAW) {
          key[i] = load_key_typed(keys_in, base + p, raw);
          idx[i] = static_cast<uint32_t>(base + p);  // first pass: row id = position
        } else {
          key[i] = keys_in[base + p];
          idx[i] = idx_in[base + p];
        }
      } else {
        key[i] = ~uint64_t(0);  // sorts last inside the tile, never written
        idx[i] = 0;
      }
    }
    // 2. zero the per-wave digit counters
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) lds.wave_cnt[w][tid] = 0;
    __syncthreads();

    // 3. wave-level multi-split: rank of each key among equal digits seen so far by this wave.
    // Lanes holding the same digit find each other with 8 ballots; the lowest such lane adds the
    // group's size to the wave's digit counter with ONE returning LDS atomic and the group reads
    // the old value back with a shuffle.  LDS operations of a wave execute in order, so the four
    // atomics of a batch are issued back to back (one round trip per 4 keys, not two per key) and
    // still see each other's updates; lane order inside a group keeps the sort stable.
#pragma unroll
    for (int i0 = 0; i0 < kSortItems; i0 += 4) {
      uint32_t prev[4];
      int leader[4];
      uint32_t below[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t d = static_cast<uint32_t>(key[i0 + j] >> shift) & 255u;
        uint64_t peers = ~uint64_t(0);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const bool bit = (d >> b) & 1u;
          const uint64_t bal = __ballot(bit);
          peers &= bit ? bal : ~bal;
        }
        leader[j] = __ffsll(static_cast<unsigned long long>(peers)) - 1;
        below[j] = __popcll(peers & ((uint64_t(1) << lane) - 1));
        prev[j] = 0;
        if (lane == leader[j]) prev[j] = atomicAdd(&lds.wave_cnt[wave][d], static_cast<uint32_t>(__popcll(peers)));
        __builtin_amdgcn_wave_barrier();  // keeps the four updates in program order
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) rank[i0 + j] = __shfl(prev[j], leader[j], 64) + below[j];
    }
    __syncthreads();

    // 4. workgroup scan: thread t owns digit t
    uint32_t c[kWavesPerBlock];
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
      c[w] = lds.wave_cnt[w][tid];
      tot += c[w];
    }
    const uint32_t incl = wave_inclusive_scan_u32(tot);
    if (lane == 63) lds.wave_tot[wave] = incl;
    __syncthreads();
    uint32_t dbase = incl - tot;
    for (int k = 0; k < wave; ++k) dbase += lds.wave_tot[k];
    uint32_t run = dbase;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
      lds.wave_cnt[w][tid] = run;
      run += c[w];
    }
    __syncthreads();

    // 5. reorder through LDS
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
      const uint32_t d = static_cast<uint32_t>(key[i] >> shift) & 255u;
      const uint32_t pos = lds.wave_cnt[wave][d] + rank[i];
      lds.keys[pos] = key[i];
      lds.idx[pos] = idx[i];
    }
    __syncthreads();

    // 6. coalesced runs to the destination
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
      const int p = k * kBlock + tid;
      if (p < nv) {
        const uint64_t kk = lds.keys[p];
        const uint32_t d = static_cast<uint32_t>(kk >> shift) & 255u;
        const uint32_t dst = lds.cursor[d] + (static_cast<uint32_t>(p) - lds.wave_cnt[0][d]);  // wave 0's run start = the digit's start
        if (last_pass) {
          out_final[dst] = lds.idx[p];
        } else {
          keys_out[dst] = kk;
          idx_out[dst] = lds.idx[p];
        }
      }
    }
    __syncthreads();
    // 7. advance the per-digit cursors of this chunk
    lds.cursor[tid] += tot;
    __syncthreads();
  }
}


// ---- multi-GPU sort support: splitter histogram + stable partition by destination
// The splitter bins are taken inside the WINDOW of the keys that exist: bin = top `bits` bits of (key - base) << shift,
// base = the smallest transformed key of all shards, shift = leading zeros of (largest - smallest).  Ids, timestamps
// and small integers share their top bits; bins of the raw key would send every row to one rank.
// A key OUTSIDE the window (a window taken from a sample of the rows, round 6) falls into the first / the last bin: the bin
// stays a non-decreasing function of the key, which is all the partition needs.
__device__ __forceinline__ uint32_t sort_window_bin(uint64_t tk, uint64_t base, int shift, int bits) {
  if (tk < base) return 0u;
  const uint64_t d = tk - base;
  if (shift > 0 && (d >> (64 - shift)) != 0) return (1u << bits) - 1u;
  return static_cast<uint32_t>((d << shift) >> (64 - bits));
}

// row i of a grid-stride loop over a SAMPLE of the rows: one tile of 8192 rows in 2^sample_shift, the tile picked inside its
// group by a hash of the group (a periodic input cannot line up with the sample).  sample_shift 0: every row.
__device__ __forceinline__ int64_t sort_sampled_row(int64_t i, int sample_shift) {
  if (sample_shift == 0) return i;
  const int64_t g = i >> 13;
  const uint32_t pick = (static_cast<uint32_t>(g) * 2654435761u) >> (32 - sample_shift);
  return (((g << sample_shift) + pick) << 13) + (i & 8191);
}

// range[0] = max of ~key (= ~min), range[1] = max key over the non-null rows (both zero-initialised by the caller,
// both combined by MAX — one all-reduce with one operator gives the global range)
__global__ __launch_bounds__(kBlock) void sort_key_range_kernel(const uint64_t* __restrict__ values, Bits valid, int64_t n,
                                                                int is_signed, int descending,
                                                                unsigned long long* __restrict__ range, int sample_shift = 0) {
  __shared__ unsigned long long s_lo[kWavesPerBlock], s_hi[kWavesPerBlock];
  unsigned long long lo = 0, hi = 0;   // lo accumulates ~key
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t work = sample_shift == 0 ? n : ((((n + 8191) >> 13) + (int64_t(1) << sample_shift) - 1) >> sample_shift) << 13;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < work; j += stride) {
    const int64_t i = sort_sampled_row(j, sample_shift);
    if (i >= n) continue;
    const bool ok = (load_word(valid, i >> 6) >> (i & 63)) & 1ull;
    if (ok) {
      const unsigned long long tk = key_transform(values[i], is_signed != 0, descending != 0);
      lo = lo > ~tk ? lo : ~tk;
      hi = hi > tk ? hi : tk;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long l2 = shfl_u64(lo, lane_id() ^ d), h2 = shfl_u64(hi, lane_id() ^ d);
    lo = lo > l2 ? lo : l2;
    hi = hi > h2 ? hi : h2;
  }
  if (lane_id() == 0) {
    s_lo[threadIdx.x >> 6] = lo;
    s_hi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kWavesPerBlock; ++w) {
      lo = lo > s_lo[w] ? lo : s_lo[w];
      hi = hi > s_hi[w] ? hi : s_hi[w];
    }
    if (lo != 0) atomicMax(&range[0], lo);
    if (hi != 0) atomicMax(&range[1], hi);
  }
}

__global__ __launch_bounds__(kBlock) void sort_key_hist_kernel(const uint64_t* __restrict__ values,
                                                               Bits valid, int64_t n, int is_signed,
                                                               int descending, int bits, uint64_t base, int shift,
                                                               unsigned long long* __restrict__ hist, int sample_shift = 0) {
  __shared__ uint32_t h[4096];
  const int nb = 1 << bits;
  for (int i = threadIdx.x; i < nb; i += kBlock) h[i] = 0;
  __syncthreads();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t work = sample_shift == 0 ? n : ((((n + 8191) >> 13) + (int64_t(1) << sample_shift) - 1) >> sample_shift) << 13;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < work; j += stride) {
    const int64_t i = sort_sampled_row(j, sample_shift);
    if (i >= n) continue;
    const bool ok = (load_word(valid, i >> 6) >> (i & 63)) & 1ull;
    if (ok) {
      const uint64_t tk = key_transform(values[i], is_signed != 0, descending != 0);
      atomicAdd(&h[sort_window_bin(tk, base, shift, bits)], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += kBlock) {
    if (h[i] != 0) atomicAdd(&hist[i], static_cast<unsigned long long>(h[i]));
  }
}

// dkeys[i] = destination of row rows[i] (number of splitter bins <= its bin), idx[i] = the row
__global__ __launch_bounds__(kBlock) void sort_dest_prep_kernel(const uint64_t* __restrict__ values,
                                                                const uint32_t* __restrict__ rows,
                                                                int64_t n, int is_signed, int descending,
                                                                int bits, uint64_t base, int shift,
                                                                const uint32_t* __restrict__ split,
                                                                int nsplit, uint64_t* __restrict__ dkeys,
                                                                uint32_t* __restrict__ idx_out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t r = rows ? rows[i] : static_cast<uint32_t>(i);
    const uint64_t tk = key_transform(values[r], is_signed != 0, descending != 0);
    const uint32_t bin = sort_window_bin(tk, base, shift, bits);
    uint32_t d = 0;
    for (int j = 0; j < nsplit; ++j) d += (split[j] <= bin) ? 1u : 0u;
    dkeys[i] = d;
    idx_out[i] = r;
  }
}

// ---- exchange records of the sharded sort: {order-transformed key, local row id}, 12 bytes
// records[i] = {transform(values[rows[i]]), rows[i]}  (rows in destination-major, row-order-preserving order)
__global__ __launch_bounds__(kBlock) void sort_pack_records_kernel(const uint64_t* __restrict__ values,
                                                                   const uint32_t* __restrict__ rows, int64_t n,
                                                                   int is_signed, int descending, int keyless,
                                                                   ArxSortRecord* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t r = rows[i];
    const uint64_t tk = keyless ? 0 : key_transform(values[r], is_signed != 0, descending != 0);
    ArxSortRecord rec;
    rec.key_lo = static_cast<uint32_t>(tk);
    rec.key_hi = static_cast<uint32_t>(tk >> 32);
    rec.row = r;
    out[i] = rec;
  }
}

// Receiver side: the concatenation of the blocks every source rank sent.  Block s holds valid[s] key records
// and nulls[s] null-row records (nulls first when nulls_first); base[s] = global row number of the source's row 0.
// Valid records are compacted in source order (keys + global rows), null rows likewise.
constexpr int kSortMaxBlocks = 1024;
__global__ __launch_bounds__(kBlock) void sort_unpack_records_kernel(const ArxSortRecord* __restrict__ rec, int64_t n,
                                                                     const int64_t* __restrict__ meta, int nblocks,
                                                                     int nulls_first, uint64_t* __restrict__ out_keys,
                                                                     int64_t* __restrict__ out_rows,
                                                                     int64_t* __restrict__ out_null_rows) {
  __shared__ int64_t s_start[kSortMaxBlocks + 1], s_vpre[kSortMaxBlocks], s_npre[kSortMaxBlocks];
  __shared__ int64_t s_valid[kSortMaxBlocks], s_nulls[kSortMaxBlocks], s_base[kSortMaxBlocks];
  if (threadIdx.x == 0) {
    int64_t st = 0, vp = 0, np = 0;
    for (int b = 0; b < nblocks; ++b) {
      s_start[b] = st;
      s_vpre[b] = vp;
      s_npre[b] = np;
      s_valid[b] = meta[3 * b];
      s_nulls[b] = meta[3 * b + 1];
      s_base[b] = meta[3 * b + 2];
      st += s_valid[b] + s_nulls[b];
      vp += s_valid[b];
      np += s_nulls[b];
    }
    s_start[nblocks] = st;
  }
  __syncthreads();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    int lo = 0, hi = nblocks;  // last block with start <= i
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_start[mid] <= i) lo = mid; else hi = mid;
    }
    const int64_t j = i - s_start[lo];
    const ArxSortRecord r = rec[i];
    const int64_t grow = s_base[lo] + static_cast<int64_t>(r.row);
    const bool is_null = nulls_first ? j < s_nulls[lo] : j >= s_valid[lo];
    if (is_null) {
      out_null_rows[s_npre[lo] + (nulls_first ? j : j - s_valid[lo])] = grow;
    } else {
      const int64_t q = s_vpre[lo] + (nulls_first ? j - s_nulls[lo] : j);
      out_keys[q] = (static_cast<uint64_t>(r.key_hi) << 32) | r.key_lo;
      out_rows[q] = grow;
    }
  }
}

// ---- round 6: the sharded sort's partition WITHOUT a stable pass.  The records carry GLOBAL row numbers and the receiver
// sorts them by (key, row) (arx_sort_records), so the order inside a destination's block is free: rows go straight from the
// column to their destination's block.  (arx_sort_partition_records orders rows by destination with a stable LSD pass over
// {destination, row} pairs + a gather: 5 GB moved at 0.8 TB/s for 2.5e8 rows, profiles/r04_u_*.)
constexpr int kSortPartMax = 64;          // destinations
constexpr int kSortPartThreads = 1024;
constexpr int kSortPartRows = 8;          // rows per thread: 8192-row tiles

__device__ __forceinline__ uint32_t sort_dest_of(uint32_t bin, const uint32_t* __restrict__ split, int nsplit) {
  uint32_t d = 0;
  for (int j = 0; j < nsplit; ++j) d += (split[j] <= bin) ? 1u : 0u;
  return d;
}

// counts[d] += rows of destination d (one LDS histogram per workgroup)
__global__ __launch_bounds__(kSortPartThreads) void sort_part_count_kernel(const uint64_t* __restrict__ values, int64_t n, int is_signed,
                                                                          int descending, int bits, uint64_t base, int shift,
                                                                          const uint32_t* __restrict__ split, int num_parts,
                                                                          unsigned long long* __restrict__ counts) {
  __shared__ uint32_t h[kSortPartMax], sp[kSortPartMax];
  const int tid = threadIdx.x;
  if (tid < kSortPartMax) {
    h[tid] = 0;
    sp[tid] = tid < num_parts - 1 ? split[tid] : 0xFFFFFFFFu;
  }
  __syncthreads();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kSortPartThreads;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kSortPartThreads + tid; i < n; i += stride) {
    const uint64_t tk = key_transform(values[i], is_signed != 0, descending != 0);
    atomicAdd(&h[sort_dest_of(sort_window_bin(tk, base, shift, bits), sp, num_parts - 1)], 1u);
  }
  __syncthreads();
  if (tid < num_parts && h[tid] != 0) atomicAdd(&counts[tid], static_cast<unsigned long long>(h[tid]));
}

// cursor[d] = first record of destination d's block (exclusive scan of the counts)
__global__ void sort_part_scan_kernel(const unsigned long long* __restrict__ counts, int num_parts, unsigned long long* __restrict__ cursor) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    unsigned long long run = 0;
    for (int d = 0; d < num_parts; ++d) {
      cursor[d] = run;
      run += counts[d];
    }
  }
}

// one tile of 8192 rows per workgroup: ranks inside the tile by LDS atomics, ONE returning global atomic per (tile,
// destination) for the tile's run in the destination's block, records {transformed key, row_base + row} written in runs
__global__ __launch_bounds__(kSortPartThreads) void sort_part_scatter_kernel(const uint64_t* __restrict__ values, int64_t n, int is_signed,
                                                                            int descending, int bits, uint64_t base, int shift,
                                                                            const uint32_t* __restrict__ split, int num_parts,
                                                                            uint32_t row_base, unsigned long long* __restrict__ cursor,
                                                                            ArxSortRecord* __restrict__ out) {
  __shared__ uint32_t h[kSortPartMax], sp[kSortPartMax];
  __shared__ unsigned long long gbase[kSortPartMax];
  const int tid = threadIdx.x;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * (kSortPartThreads * kSortPartRows);
  if (tid < kSortPartMax) {
    h[tid] = 0;
    sp[tid] = tid < num_parts - 1 ? split[tid] : 0xFFFFFFFFu;
  }
  __syncthreads();
  uint64_t tk[kSortPartRows];
  uint32_t dest[kSortPartRows], rank[kSortPartRows];
#pragma unroll
  for (int i = 0; i < kSortPartRows; ++i) {
    const int64_t r = row0 + i * kSortPartThreads + tid;
    tk[i] = key_transform(values[r < n ? r : n - 1], is_signed != 0, descending != 0);
  }
#pragma unroll
  for (int i = 0; i < kSortPartRows; ++i) {
    const int64_t r = row0 + i * kSortPartThreads + tid;
    dest[i] = sort_dest_of(sort_window_bin(tk[i], base, shift, bits), sp, num_parts - 1);
    rank[i] = r < n ? atomicAdd(&h[dest[i]], 1u) : 0u;
  }
  __syncthreads();
  if (tid < num_parts) gbase[tid] = h[tid] != 0 ? atomicAdd(&cursor[tid], static_cast<unsigned long long>(h[tid])) : 0ull;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kSortPartRows; ++i) {
    const int64_t r = row0 + i * kSortPartThreads + tid;
    if (r < n) {
      ArxSortRecord rec;
      rec.key_lo = static_cast<uint32_t>(tk[i]);
      rec.key_hi = static_cast<uint32_t>(tk[i] >> 32);
      rec.row = row_base + static_cast<uint32_t>(r);
      out[gbase[dest[i]] + rank[i]] = rec;
    }
  }
}

// {row as the key, position} of every record / the records at the given positions: what puts the records in row order
// before the LSD passes of arx_sort_records (stable on the key alone)
__global__ __launch_bounds__(kBlock) void sort_records_rows_kernel(const ArxSortRecord* __restrict__ rec, int64_t n,
                                                                   uint64_t* __restrict__ keys, uint32_t* __restrict__ pos) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    keys[i] = rec[i].row;
    pos[i] = static_cast<uint32_t>(i);
  }
}
__global__ __launch_bounds__(kBlock) void sort_records_gather_kernel(const ArxSortRecord* __restrict__ rec, const uint32_t* __restrict__ pos,
                                                                     int64_t n, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const ArxSortRecord r = rec[pos[i]];
    keys[i] = (static_cast<uint64_t>(r.key_hi) << 32) | r.key_lo;
    idx[i] = r.row;
  }
}

// records -> the (keys, row ids) pair of arrays the sorts read
__global__ __launch_bounds__(kBlock) void sort_split_records_kernel(const ArxSortRecord* __restrict__ rec, int64_t n,
                                                                    uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const ArxSortRecord r = rec[i];
    keys[i] = (static_cast<uint64_t>(r.key_hi) << 32) | r.key_lo;
    idx[i] = r.row;
  }
}

__global__ void widen_counts_kernel(const uint32_t* __restrict__ in, int n, int64_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}


// =====================================================================================
// MSD-hybrid path of arx_sort_indices_64 (large inputs).
//
// The stable argsort is the unique ascending order of the COMPOSITE (transformed key, row id), so
// no individual pass has to be stable as long as the last step orders by the composite.  That
// allows most-significant-digit partitioning with cheap LDS-atomic ranking (the group-by's
// scatter, ~4 TB/s) instead of eight stable least-significant-digit passes (256 B/row of traffic):
//   M1 msd_hist      top `bits` (<= 14) bits: per-chunk level-1 counts + global bucket counts (8 B/row)
//   M2 msd_scan_a/b  bucket starts, level-1 chunk offsets, level-2 cursors and tile map
//   M3 msd_scatter1  level 1 (b1 bits), chunked, exact offsets                       (12+12 B/row)
//   M4 msd_scatter2  level 2 (b2 bits) inside every level-1 bucket, cursor atomics   (12+12 B/row)
//   M5 msd_local     level 3 (b3 <= 9 bits): one workgroup per level-2 bucket partitions it in place
//                    of its own range (histogram pass + scatter pass, second read L2-warm)
//   M6 msd_final     buckets now hold ~32-256 rows sharing their top bits: every row counts the
//                    bucket members with a smaller (key, row id) — its final position — inside an
//                    LDS window with a halo, and writes its row id there as uint64      (12+8 B/row)
// A bucket larger than the halo (heavily duplicated keys) raises a flag and the caller falls back
// to the LSD path, which has no such limit.
// =====================================================================================
constexpr int kMsdThreads = 512;
constexpr int kMsdTile = 4096;
constexpr int kMsdRows = kMsdTile / kMsdThreads;  // 8 per thread
constexpr int kMsdMaxBits = 14;
constexpr int kMsdMaxChunks = 2048;
constexpr int kMsdCore = 2048;   // rows finalised per workgroup of msd_final
constexpr int kMsdHalo = 256;    // a bucket must fit in the halo on either side
constexpr int kMsdWindow = kMsdCore + 2 * kMsdHalo;

struct MsdArgs {
  const uint64_t* src_keys;   // level-1 input: raw column (raw != 0) or transformed keys
  const uint32_t* src_idx;    // level-1 input row ids (unused when raw)
  int raw;                    // 0, or make_xf(key_type, descending)
  int64_t n;
  int bits, b1, b2, b3;       // bits = b1 + b2 (global levels), b3 = local level
  int kshift;                 // top bits already equal inside this segment: digits are taken from key << kshift
  const uint64_t* spl;        // sampled-splitter mode: 2^bits - 1 ascending splitters (else NULL)
  int64_t chunk_rows, nchunks;
  uint32_t* part_count;       // [2^bits]
  uint32_t* part_start;       // [2^bits + 1]
  uint32_t* cursor2;          // [2^bits]
  uint32_t* hist1;            // [2^b1 * nchunks]
  uint32_t* l1_start;         // [2^b1 + 1]
  uint32_t* l2_tile_start;    // [2^b1 + 1]
  uint64_t* keys_x;           // level-1 output, level-3 output
  uint32_t* idx_x;
  uint64_t* keys_y;           // level-2 output
  uint32_t* idx_y;
  uint64_t* out_final;
  unsigned int* overflow;
  const uint32_t* part_in;    // record input of the bucket finish (AOS form): first record of every bucket
  int xcd_map;                // XCD-contiguous work numbering: bit 0 the level-2 scatter, bit 1 the bucket finish
  int tie_shift;              // rec8 finish: a bucket gives up (overflow bit 32) once more than (its rows >> tie_shift) rows tied
};

// (key, row id) as one 12-byte record: what the wide form's two scatter levels write and read (one output stream per
// bin instead of two: 5-20 % faster scatters, scripts/micro/scatter_bench.hip)
struct __attribute__((packed, aligned(4))) MsdRec {
  uint32_t lo, hi, idx;
};
__device__ __forceinline__ uint64_t msd_rec_key(const MsdRec& r) { return (static_cast<uint64_t>(r.hi) << 32) | r.lo; }

template <bool RAW>
__device__ __forceinline__ uint64_t msd_load_key(const MsdArgs& a, int64_t i) {
  if constexpr (RAW) {
    return load_key_typed(a.src_keys, i, a.raw);
  } else {
    return a.src_keys[i];
  }
}

template <bool RAW>
__global__ __launch_bounds__(kMsdThreads) void msd_hist_kernel(MsdArgs a) {
  __shared__ uint32_t h[1 << kMsdMaxBits];
  const int tid = threadIdx.x;
  const int nparts = 1 << a.bits;
  for (int i = tid; i < nparts; i += kMsdThreads) h[i] = 0;
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * a.chunk_rows;
  const int64_t end = begin + a.chunk_rows < a.n ? begin + a.chunk_rows : a.n;
  const int shift = 64 - a.bits;
  constexpr int U = 8;
  int64_t r = begin + tid;
  for (; r + (U - 1) * kMsdThreads < end; r += U * kMsdThreads) {
    uint64_t kk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) kk[u] = msd_load_key<RAW>(a, r + u * kMsdThreads);
#pragma unroll
    for (int u = 0; u < U; ++u) atomicAdd(&h[(kk[u] << a.kshift) >> shift], 1u);
  }
  for (; r < end; r += kMsdThreads) atomicAdd(&h[(msd_load_key<RAW>(a, r) << a.kshift) >> shift], 1u);
  __syncthreads();
  for (int i = tid; i < nparts; i += kMsdThreads) {
    const uint32_t c = h[i];
    if (c != 0) atomicAdd(&a.part_count[i], c);
  }
  const int nb1 = 1 << a.b1;
  const int per = 1 << a.b2;
  for (int d = tid; d < nb1; d += kMsdThreads) {
    uint32_t sum = 0;
    for (int j = 0; j < per; ++j) sum += h[(d << a.b2) + j];
    a.hist1[static_cast<int64_t>(d) * a.nchunks + blockIdx.x] = sum;
  }
}

// one workgroup: bucket starts (exclusive scan), level-1 starts, level-2 cursors + tile map
__global__ __launch_bounds__(1024) void msd_scan_a_kernel(MsdArgs a) {
  __shared__ uint32_t ps[(1 << 15) + 1];  // up to 2^15 buckets (sampled-splitter mode)
  __shared__ uint32_t wave_tot[16];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nparts = 1 << a.bits;
  const int per = (nparts + 1023) / 1024;
  const int b = tid * per < nparts ? tid * per : nparts;
  const int e = b + per < nparts ? b + per : nparts;
  uint32_t sum = 0;
  for (int i = b; i < e; ++i) sum += a.part_count[i];
  const uint32_t incl = wave_inclusive_scan_u32(sum);
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t prefix = incl - sum;
  for (int k = 0; k < wave; ++k) prefix += wave_tot[k];
  for (int i = b; i < e; ++i) {
    ps[i] = prefix;
    prefix += a.part_count[i];
  }
  if (tid == 1023) ps[nparts] = prefix;
  __syncthreads();
  for (int i = tid; i <= nparts; i += 1024) {
    const uint32_t s0 = ps[i];
    a.part_start[i] = s0;
    if (i < nparts) a.cursor2[i] = s0;
  }
  const int nb1 = 1 << a.b1;
  for (int d = tid; d <= nb1; d += 1024) a.l1_start[d] = ps[d == nb1 ? nparts : (d << a.b2)];
  __syncthreads();
  uint32_t tiles = 0;
  if (tid < nb1) {
    const uint32_t lo = ps[tid << a.b2];
    const uint32_t hi = ps[(tid + 1) == nb1 ? nparts : ((tid + 1) << a.b2)];
    tiles = (hi - lo + kMsdTile - 1) / kMsdTile;
  }
  const uint32_t tincl = wave_inclusive_scan_u32(tiles);
  __syncthreads();
  if (lane == 63) wave_tot[wave] = tincl;
  __syncthreads();
  uint32_t tprefix = tincl - tiles;
  for (int k = 0; k < wave; ++k) tprefix += wave_tot[k];
  if (tid < nb1) a.l2_tile_start[tid] = tprefix;
  if (tid == nb1 - 1) a.l2_tile_start[nb1] = tprefix + tiles;
  // largest bucket -> overflow[1]: lets the host skip the scatter passes when the top bits are
  // too skewed for the LDS-resident finish (it then goes straight to the LSD passes)
  uint32_t mx = 0;
  for (int i = b; i < e; ++i) mx = mx > a.part_count[i] ? mx : a.part_count[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = __shfl_xor(mx, d, 64);
    mx = mx > o ? mx : o;
  }
  __syncthreads();
  if (lane == 0) wave_tot[wave] = mx;
  __syncthreads();
  if (tid == 0) {
    uint32_t m = 0;
    for (int k = 0; k < 16; ++k) m = m > wave_tot[k] ? m : wave_tot[k];
    a.overflow[1] = m;
  }
}

// one workgroup per level-1 digit: hist1[d][*] -> exclusive offsets (+ l1_start[d])
__global__ __launch_bounds__(64) void msd_scan_b_kernel(MsdArgs a) {
  const int lane = threadIdx.x;
  uint32_t* row = a.hist1 + static_cast<int64_t>(blockIdx.x) * a.nchunks;
  uint32_t carry = a.l1_start[blockIdx.x];
  for (int64_t base = 0; base < a.nchunks; base += 64) {
    const int64_t i = base + lane;
    const uint32_t x = i < a.nchunks ? row[i] : 0u;
    const uint32_t incl = wave_inclusive_scan_u32(x);
    if (i < a.nchunks) row[i] = carry + incl - x;
    carry += __shfl(incl, 63, 64);
  }
}

template <int MAXBINS>
struct __attribute__((aligned(16))) MsdScatterLds {
  uint64_t keys[kMsdTile];
  uint32_t idx[kMsdTile];
  uint32_t cnt[MAXBINS];
  uint32_t start[MAXBINS];
  uint32_t gbase[MAXBINS];
  uint32_t cursor[MAXBINS];
  uint32_t wave_tot[kMsdThreads / 64];
  uint64_t spl[256];  // sampled-splitter mode: the nb - 1 splitters of this level / bucket
};

// number of splitters <= key (upper bound) among spl[0, nsp), nsp = 2^k - 1
__device__ __forceinline__ uint32_t msd_search(const uint64_t* spl, int nsp, uint64_t key) {
  uint32_t lo = 0;
  for (int half = (nsp + 1) >> 1; half >= 1; half >>= 1) {
    if (spl[lo + half - 1] <= key) lo += half;
  }
  return lo;
}

// Scatter one tile of <= 4096 (key, row id) pairs by digit = (key >> dshift) & (nb - 1).
// MODE 0: run bases from lds.cursor (advanced per tile);  MODE 1: from a global cursor array
// (one returning atomic per digit);  the destination arrays are absolute.
template <bool RAW, int MODE, int MAXBINS, bool SPL = false>
__device__ __forceinline__ void msd_scatter_tile(const MsdArgs& a, MsdScatterLds<MAXBINS>& lds,
                                                 const uint64_t* __restrict__ kin,
                                                 const uint32_t* __restrict__ iin, int64_t row0,
                                                 int nrows, int nb, int dshift,
                                                 uint32_t* __restrict__ gcursor, uint32_t dst0,
                                                 uint64_t* __restrict__ kout, uint32_t* __restrict__ iout) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t dmask = static_cast<uint32_t>(nb - 1);
  if (tid < nb) lds.cnt[tid] = 0;
  __syncthreads();
  uint64_t key[kMsdRows];
  uint32_t idx[kMsdRows];
  int dig[kMsdRows];
  uint32_t rank[kMsdRows];
  // Loads are unconditional (rows past the tile's end re-read its last row): a guarded load makes
  // the compiler wait for each one before the next is issued, i.e. one HBM round trip per row.
#pragma unroll
  for (int i = 0; i < kMsdRows; ++i) {
    const int p = i * kMsdThreads + tid;
    const int64_t r = row0 + (p < nrows ? p : nrows - 1);
    if constexpr (RAW) {
      key[i] = load_key_typed(kin, r, a.raw);
      idx[i] = static_cast<uint32_t>(r);
    } else {
      key[i] = kin[r];
      idx[i] = iin[r];
    }
  }
#pragma unroll
  for (int i = 0; i < kMsdRows; ++i) {
    const int p = i * kMsdThreads + tid;
    if constexpr (SPL) {
      dig[i] = static_cast<int>(msd_search(lds.spl, nb - 1, key[i]));
    } else {
      dig[i] = static_cast<int>(static_cast<uint32_t>((key[i] << a.kshift) >> dshift) & dmask);
    }
    if (p >= nrows) dig[i] = -1;
  }
#pragma unroll
  for (int i = 0; i < kMsdRows; ++i) {
    rank[i] = 0;
    if (dig[i] >= 0) rank[i] = atomicAdd(&lds.cnt[dig[i]], 1u);
  }
  __syncthreads();
  uint32_t c = 0;
  if (tid < nb) c = lds.cnt[tid];
  const uint32_t incl = wave_inclusive_scan_u32(c);
  if (lane == 63) lds.wave_tot[wave] = incl;
  __syncthreads();
  if (tid < nb) {
    uint32_t pre = incl - c;
    for (int k = 0; k < wave; ++k) pre += lds.wave_tot[k];
    lds.start[tid] = pre;
    if constexpr (MODE == 0) {
      const uint32_t g = lds.cursor[tid];
      lds.gbase[tid] = g;
      lds.cursor[tid] = g + c;
    } else {
      lds.gbase[tid] = c != 0 ? atomicAdd(&gcursor[tid], c) : 0u;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kMsdRows; ++i) {
    if (dig[i] >= 0) {
      const uint32_t pos = lds.start[dig[i]] + rank[i];
      lds.keys[pos] = key[i];
      lds.idx[pos] = idx[i];
    }
  }
  __syncthreads();
  for (int p = tid; p < nrows; p += kMsdThreads) {
    const uint64_t k = lds.keys[p];
    const uint32_t d = SPL ? msd_search(lds.spl, nb - 1, k)
                           : (static_cast<uint32_t>((k << a.kshift) >> dshift) & dmask);
    const uint32_t dst = dst0 + lds.gbase[d] + (static_cast<uint32_t>(p) - lds.start[d]);
    kout[dst] = k;
    iout[dst] = lds.idx[p];
  }
  __syncthreads();
}

template <bool RAW>
__global__ __launch_bounds__(kMsdThreads, 6) void msd_scatter1_kernel(MsdArgs a) {
  __shared__ MsdScatterLds<128> lds;
  const int tid = threadIdx.x;
  const int nb = 1 << a.b1;
  if (tid < nb) lds.cursor[tid] = a.hist1[static_cast<int64_t>(tid) * a.nchunks + blockIdx.x];
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * a.chunk_rows;
  const int64_t end = begin + a.chunk_rows < a.n ? begin + a.chunk_rows : a.n;
  for (int64_t row0 = begin; row0 < end; row0 += kMsdTile) {
    const int nrows = static_cast<int>(end - row0 < kMsdTile ? end - row0 : kMsdTile);
    msd_scatter_tile<RAW, 0, 128>(a, lds, a.src_keys, a.src_idx, row0, nrows, nb, 64 - a.b1, nullptr, 0u,
                                  a.keys_x, a.idx_x);
  }
}

__global__ __launch_bounds__(kMsdThreads, 6) void msd_scatter2_kernel(MsdArgs a) {
  __shared__ MsdScatterLds<128> lds;
  __shared__ uint32_t part_s;
  const int tid = threadIdx.x;
  const int nb1 = 1 << a.b1;
  // (a.xcd_map bit 0: XCD x takes a contiguous eighth of the tiles — a level-1 bucket's runs meet in one L2)
  const uint32_t g = (a.xcd_map & 1) ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  if (g >= a.l2_tile_start[nb1]) return;  // over-provisioned grid
  if (tid < 64) {
    uint32_t below = 0;
    for (int p = tid; p < nb1; p += 64) below += (a.l2_tile_start[p] <= g) ? 1u : 0u;
    below = wave_reduce_sum_u32(below);
    if (tid == 0) part_s = below - 1;
  }
  __syncthreads();
  const uint32_t p = part_s;
  const int64_t lo = a.l1_start[p];
  const int64_t hi = a.l1_start[p + 1];
  const int64_t row0 = lo + static_cast<int64_t>(g - a.l2_tile_start[p]) * kMsdTile;
  const int nrows = static_cast<int>(hi - row0 < kMsdTile ? hi - row0 : kMsdTile);
  msd_scatter_tile<false, 1, 128>(a, lds, a.keys_x, a.idx_x, row0, nrows, 1 << a.b2, 64 - a.bits,
                                  a.cursor2 + (static_cast<size_t>(p) << a.b2), 0u, a.keys_y, a.idx_y);
}

// ---------------------------------------------------------------- sampled-splitter mode
// When the top bits are skewed (normally distributed floats, clustered ids) equal-width buckets
// overflow.  Then the bucket boundaries come from the data: a regular sample of 32 keys per bucket
// is sorted (LSD passes on <= 1M keys) and every 32nd becomes a splitter; bucket(key) = number of
// splitters <= key, found by a 7/8-step search over the 127/255 splitters of one level held in LDS.
//   S1 msd_sample        sample keys (jittered stride)
//   S2 (LSD passes)      sort the sample;  msd_pick_splitters: every (m / 2^bits)-th
//   S3 msd_hist1_s       level-1 counts per chunk (search among the 2^b1 - 1 level-1 splitters)
//   S4 msd_scatter1_s    level 1;   msd_hist2_s: per level-1 bucket, counts of its 2^b2 sub-buckets
//   S5 msd_scatter2_s    level 2;   msd_bucket<SPL>: sub-buckets by linear interpolation inside the
//                        bucket's key range (monotonic, so sub-bucket order = key order)
template <bool RAW>
__global__ __launch_bounds__(kBlock) void msd_sample_kernel(MsdArgs a, int64_t m, uint64_t* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= m) return;
  const int64_t stride = a.n / m;  // >= 4
  const uint64_t jitter = (static_cast<uint64_t>(i) * 0x9E3779B97F4A7C15ull) >> 33;
  const int64_t pos = i * stride + static_cast<int64_t>(jitter % static_cast<uint64_t>(stride));
  out[i] = msd_load_key<RAW>(a, pos);
}

__global__ __launch_bounds__(kBlock) void msd_pick_splitters_kernel(const uint64_t* __restrict__ sorted,
                                                                    int64_t m, int nparts,
                                                                    uint64_t* __restrict__ spl) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j < nparts - 1) spl[j] = sorted[static_cast<int64_t>(j + 1) * (m / nparts)];
}

// level-1 histogram: like msd_hist with bits = b1, digit by search
template <bool RAW>
__global__ __launch_bounds__(kMsdThreads) void msd_hist1_s_kernel(MsdArgs a) {
  __shared__ uint32_t h[256];
  __shared__ uint64_t l1[256];
  const int tid = threadIdx.x;
  const int nb1 = 1 << a.b1;
  const int per = 1 << a.b2;
  if (tid < nb1) h[tid] = 0;
  if (tid < nb1 - 1) l1[tid] = a.spl[(tid + 1) * per - 1];
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * a.chunk_rows;
  const int64_t end = begin + a.chunk_rows < a.n ? begin + a.chunk_rows : a.n;
  constexpr int U = 8;
  int64_t r = begin + tid;
  for (; r + (U - 1) * kMsdThreads < end; r += U * kMsdThreads) {
    uint64_t kk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) kk[u] = msd_load_key<RAW>(a, r + u * kMsdThreads);
#pragma unroll
    for (int u = 0; u < U; ++u) atomicAdd(&h[msd_search(l1, nb1 - 1, kk[u])], 1u);
  }
  for (; r < end; r += kMsdThreads) atomicAdd(&h[msd_search(l1, nb1 - 1, msd_load_key<RAW>(a, r))], 1u);
  __syncthreads();
  if (tid < nb1) {
    const uint32_t c = h[tid];
    if (c != 0) atomicAdd(&a.part_count[tid], c);
    a.hist1[static_cast<int64_t>(tid) * a.nchunks + blockIdx.x] = c;
  }
}

template <bool RAW>
__global__ __launch_bounds__(kMsdThreads, 6) void msd_scatter1_s_kernel(MsdArgs a) {
  __shared__ MsdScatterLds<128> lds;
  const int tid = threadIdx.x;
  const int nb = 1 << a.b1;
  const int per = 1 << a.b2;
  if (tid < nb) lds.cursor[tid] = a.hist1[static_cast<int64_t>(tid) * a.nchunks + blockIdx.x];
  if (tid < nb - 1) lds.spl[tid] = a.spl[(tid + 1) * per - 1];
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * a.chunk_rows;
  const int64_t end = begin + a.chunk_rows < a.n ? begin + a.chunk_rows : a.n;
  for (int64_t row0 = begin; row0 < end; row0 += kMsdTile) {
    const int nrows = static_cast<int>(end - row0 < kMsdTile ? end - row0 : kMsdTile);
    msd_scatter_tile<RAW, 0, 128, true>(a, lds, a.src_keys, a.src_idx, row0, nrows, nb, 0, nullptr, 0u,
                                        a.keys_x, a.idx_x);
  }
}

// which level-1 bucket owns tile g (shared by the level-2 kernels)
__device__ __forceinline__ uint32_t msd_tile_owner(const MsdArgs& a, uint32_t g, uint32_t* part_s) {
  const int tid = threadIdx.x;
  const int nb1 = 1 << a.b1;
  if (tid < 64) {
    uint32_t below = 0;
    for (int p = tid; p < nb1; p += 64) below += (a.l2_tile_start[p] <= g) ? 1u : 0u;
    below = wave_reduce_sum_u32(below);
    if (tid == 0) *part_s = below - 1;
  }
  __syncthreads();
  return *part_s;
}

// level-2 counts: one tile of one level-1 bucket per workgroup -> part_count[p * 2^b2 + d]
__global__ __launch_bounds__(kMsdThreads) void msd_hist2_s_kernel(MsdArgs a) {
  __shared__ uint32_t h[256];
  __shared__ uint64_t l2[256];
  __shared__ uint32_t part_s;
  const int tid = threadIdx.x;
  const int nb1 = 1 << a.b1;
  const int nb2 = 1 << a.b2;
  const uint32_t g = blockIdx.x;
  if (g >= a.l2_tile_start[nb1]) return;
  const uint32_t p = msd_tile_owner(a, g, &part_s);
  if (tid < nb2) h[tid] = 0;
  if (tid < nb2 - 1) l2[tid] = a.spl[(static_cast<size_t>(p) << a.b2) + tid];
  __syncthreads();
  const int64_t lo = a.l1_start[p];
  const int64_t hi = a.l1_start[p + 1];
  const int64_t row0 = lo + static_cast<int64_t>(g - a.l2_tile_start[p]) * kMsdTile;
  const int64_t end = row0 + kMsdTile < hi ? row0 + kMsdTile : hi;
  for (int64_t r = row0 + tid; r < end; r += kMsdThreads) atomicAdd(&h[msd_search(l2, nb2 - 1, a.keys_x[r])], 1u);
  __syncthreads();
  if (tid < nb2 && h[tid] != 0) atomicAdd(&a.part_count[(static_cast<size_t>(p) << a.b2) + tid], h[tid]);
}

__global__ __launch_bounds__(kMsdThreads) void msd_scatter2_s_kernel(MsdArgs a) {
  __shared__ MsdScatterLds<256> lds;
  __shared__ uint32_t part_s;
  const int tid = threadIdx.x;
  const int nb1 = 1 << a.b1;
  const int nb2 = 1 << a.b2;
  const uint32_t g = blockIdx.x;
  if (g >= a.l2_tile_start[nb1]) return;
  const uint32_t p = msd_tile_owner(a, g, &part_s);
  if (tid < nb2 - 1) lds.spl[tid] = a.spl[(static_cast<size_t>(p) << a.b2) + tid];
  __syncthreads();
  const int64_t lo = a.l1_start[p];
  const int64_t hi = a.l1_start[p + 1];
  const int64_t row0 = lo + static_cast<int64_t>(g - a.l2_tile_start[p]) * kMsdTile;
  const int nrows = static_cast<int>(hi - row0 < kMsdTile ? hi - row0 : kMsdTile);
  msd_scatter_tile<false, 1, 256, true>(a, lds, a.keys_x, a.idx_x, row0, nrows, nb2, 0,
                                        a.cursor2 + (static_cast<size_t>(p) << a.b2), 0u, a.keys_y, a.idx_y);
}

// level 3: one workgroup per level-2 bucket, (keys_y, idx_y) -> (keys_x, idx_x) inside the
// bucket's own range
__global__ __launch_bounds__(kMsdThreads) void msd_local_kernel(MsdArgs a) {
  __shared__ MsdScatterLds<512> lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t q = blockIdx.x;
  const int64_t lo = a.part_start[q];
  const int64_t hi = a.part_start[q + 1];
  if (lo == hi) return;  // workgroup-uniform
  const int nb = 1 << a.b3;
  const int dshift = 64 - a.bits - a.b3;
  const uint32_t dmask = static_cast<uint32_t>(nb - 1);
  // histogram of the bucket -> running cursors (relative to the bucket start)
  if (tid < nb) lds.cnt[tid] = 0;
  __syncthreads();
  for (int64_t r = lo + tid; r < hi; r += kMsdThreads) {
    atomicAdd(&lds.cnt[static_cast<uint32_t>((a.keys_y[r] << a.kshift) >> dshift) & dmask], 1u);
  }
  __syncthreads();
  uint32_t c = 0;
  if (tid < nb) c = lds.cnt[tid];
  const uint32_t incl = wave_inclusive_scan_u32(c);
  if (lane == 63) lds.wave_tot[wave] = incl;
  __syncthreads();
  if (tid < nb) {
    uint32_t pre = incl - c;
    for (int k = 0; k < wave; ++k) pre += lds.wave_tot[k];
    lds.cursor[tid] = pre;
  }
  __syncthreads();
  for (int64_t row0 = lo; row0 < hi; row0 += kMsdTile) {
    const int nrows = static_cast<int>(hi - row0 < kMsdTile ? hi - row0 : kMsdTile);
    msd_scatter_tile<false, 0, 512>(a, lds, a.keys_y, a.idx_y, row0, nrows, nb, dshift, nullptr,
                                    static_cast<uint32_t>(lo), a.keys_x, a.idx_x);
  }
}

struct __attribute__((aligned(16))) MsdFinalLds {
  uint64_t keys[kMsdWindow];
  uint32_t idx[kMsdWindow];
  uint64_t head[kMsdWindow / 64 + 1];  // bit i of word w: window row 64 w + i starts a new bucket
};

// final: position of every row inside its bucket = number of members with a smaller (key, id).
// Bucket boundaries come from a bitmap of "prefix changes here" flags (one ballot per 64 rows):
// the start of row i's bucket is the last set bit at or below i, its end the first set bit above.
__global__ __launch_bounds__(256) void msd_final_kernel(MsdArgs a, const uint64_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ idx, int pshift) {
  __shared__ MsdFinalLds w;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int64_t core0 = static_cast<int64_t>(blockIdx.x) * kMsdCore;
  const int64_t wb = core0 - kMsdHalo > 0 ? core0 - kMsdHalo : 0;
  const int64_t we = core0 + kMsdCore + kMsdHalo < a.n ? core0 + kMsdCore + kMsdHalo : a.n;
  const int wlen = static_cast<int>(we - wb);
  for (int i = tid; i < wlen; i += 256) {
    w.keys[i] = keys[wb + i];
    w.idx[i] = idx[wb + i];
  }
  __syncthreads();
  constexpr int kWords = kMsdWindow / 64;
  for (int base = 0; base < kWords * 64; base += 256) {
    const int i = base + tid;
    const bool head = i < wlen && (i == 0 || ((w.keys[i] << a.kshift) >> pshift) != ((w.keys[i - 1] << a.kshift) >> pshift));
    const uint64_t bal = __ballot(head);
    if (lane == 0) w.head[i >> 6] = bal;
  }
  __syncthreads();
  const int nwords = (wlen + 63) >> 6;
  const int c_lo = static_cast<int>(core0 - wb);
  const int c_hi = static_cast<int>((core0 + kMsdCore < a.n ? core0 + kMsdCore : a.n) - wb);
  bool bad = false;
  for (int i = c_lo + tid; i < c_hi; i += 256) {
    // bucket start: last head bit at or below i
    int wd = i >> 6;
    const int bit = i & 63;
    uint64_t m = w.head[wd] & (bit == 63 ? ~uint64_t(0) : ((uint64_t(2) << bit) - 1));
    while (m == 0) m = w.head[--wd];  // terminates: row 0 of the window is always a head
    const int bs = (wd << 6) + 63 - __builtin_clzll(m);
    // bucket end: first head bit above i (or the window end)
    wd = i >> 6;
    m = bit == 63 ? 0 : (w.head[wd] & ~((uint64_t(2) << bit) - 1));
    while (m == 0 && ++wd < nwords) m = w.head[wd];
    const int be = m != 0 ? (wd << 6) + (__ffsll(static_cast<unsigned long long>(m)) - 1) : wlen;
    if ((bs == 0 && wb > 0) || (be == wlen && we < a.n)) {
      bad = true;  // the bucket may continue outside the window
      continue;
    }
    const uint64_t ki = w.keys[i];
    const uint32_t ii = w.idx[i];
    int rank = 0;
    // four members per step: the LDS reads are independent (rows past the bucket end re-read its
    // last member and are masked out), so their latency overlaps
    for (int j = bs; j < be; j += 4) {
      uint64_t kj[4];
      uint32_t ij[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jj = (j + u) < be ? (j + u) : (be - 1);
        kj[u] = w.keys[jj];
        ij[u] = w.idx[jj];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool less = kj[u] < ki || (kj[u] == ki && ij[u] < ii);
        rank += (less && (j + u) < be) ? 1 : 0;
      }
    }
    a.out_final[wb + bs + rank] = ii;
  }
  if (__any(bad) && lane == 0) atomicOr(a.overflow, 1u);
}

// M5+M6 fused, for level-2 buckets that fit LDS (<= kBktCap rows, i.e. inputs / segments up to 2^27
// rows): one workgroup loads its bucket into registers, partitions it by the next b3 (<= 10) bits
// straight into LDS (count -> scan -> LDS-atomic cursors), then every row counts the members of its
// sub-bucket with a smaller (key, row id) and writes its row id at that position of the output.
// No level-3 round trip through HBM, no halo re-reads: 12 B/row in, 8 B/row out.
// Two sizes: 1024 threads / 10240 rows (120 KiB of LDS, one workgroup per CU) and 512 threads /
// 5120 rows (two workgroups per CU, whose load / LDS / store phases overlap) — the host picks the
// small one whenever the largest bucket fits it.
constexpr int kBktThreads = 1024;
constexpr int kBktRows = 10;                       // per thread
constexpr int kBktCap = kBktThreads * kBktRows;    // 10240 rows = 120 KiB of LDS
constexpr int kBktThreadsSmall = 512;
constexpr int kBktCapSmall = kBktThreadsSmall * kBktRows;
constexpr int kBktThreadsTiny = 256;   // wide form with 2^20 buckets of ~2K rows: 34 KB of LDS, four finishes per CU
constexpr int kBktCapTiny = kBktThreadsTiny * kBktRows;
constexpr int kBktRowsTiny9 = 9;       // 2304 rows: under 32 KB of LDS, five finishes per CU
constexpr int kBktCapTiny9 = kBktThreadsTiny * kBktRowsTiny9;
constexpr int kBktMaxBins = 1024;

template <int T>
struct __attribute__((aligned(16))) MsdBucketLds {
  uint64_t keys[T * kBktRows];
  uint32_t idx[T * kBktRows];
  uint32_t cnt[kBktMaxBins];     // counts, then running cursors
  uint32_t start[kBktMaxBins + 1];
  uint32_t wave_tot[T / 64];
};

template <bool SPL, int T>
__global__ __launch_bounds__(T) void msd_bucket_kernel(MsdArgs a, const uint64_t* __restrict__ keys,
                                                       const uint32_t* __restrict__ idx) {
  __shared__ MsdBucketLds<T> w;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t q = blockIdx.x;
  const int64_t lo = a.part_start[q];
  const int m = static_cast<int>(static_cast<int64_t>(a.part_start[q + 1]) - lo);
  if (m == 0) return;  // workgroup-uniform
  if (m > T * kBktRows) {
    if (tid == 0) atomicOr(a.overflow, 2u);
    return;
  }
  const int nb = 1 << a.b3;
  const int dshift = 64 - a.bits - a.b3;
  const uint32_t dmask = static_cast<uint32_t>(nb - 1);
  // sampled-splitter mode: sub-bucket = linear interpolation inside [lo, hi), the bucket's key range
  uint64_t klo = 0;
  double kinv = 0.0;
  if constexpr (SPL) {
    const uint32_t nparts = 1u << a.bits;
    klo = q > 0 ? a.spl[q - 1] : 0;
    const uint64_t khi = q + 1 < nparts ? a.spl[q] : ~uint64_t(0);
    const double range = static_cast<double>(khi - klo) + 1.0;
    kinv = static_cast<double>(nb) / range;
  }
  auto digit_of = [&](uint64_t k) -> uint32_t {
    if constexpr (SPL) {
      const uint32_t d = static_cast<uint32_t>(static_cast<double>(k - klo) * kinv);
      return d < static_cast<uint32_t>(nb) ? d : static_cast<uint32_t>(nb - 1);
    } else {
      return a.b3 == 0 ? 0u : (static_cast<uint32_t>((k << a.kshift) >> dshift) & dmask);
    }
  };
  for (int i = tid; i < nb; i += T) w.cnt[i] = 0;
  __syncthreads();
  uint64_t key[kBktRows];
  uint32_t id[kBktRows];
  int dig[kBktRows];
  // unconditional loads (clamped to the bucket's last row): all kBktRows round trips overlap
#pragma unroll
  for (int i = 0; i < kBktRows; ++i) {
    const int p = i * T + tid;
    const int64_t r = lo + (p < m ? p : m - 1);
    key[i] = keys[r];
    id[i] = idx[r];
  }
#pragma unroll
  for (int i = 0; i < kBktRows; ++i) {
    const int p = i * T + tid;
    dig[i] = p < m ? static_cast<int>(digit_of(key[i])) : -1;
  }
#pragma unroll
  for (int i = 0; i < kBktRows; ++i) {
    if (dig[i] >= 0) atomicAdd(&w.cnt[dig[i]], 1u);
  }
  __syncthreads();
  // exclusive scan of nb <= 1024 counters: kBktMaxBins / T consecutive counters per thread
  constexpr int CPT = kBktMaxBins / T;
  uint32_t c[CPT];
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int b = tid * CPT + k;
    c[k] = b < nb ? w.cnt[b] : 0u;
    mine += c[k];
  }
  const uint32_t incl = wave_inclusive_scan_u32(mine);
  if (lane == 63) w.wave_tot[wave] = incl;
  __syncthreads();
  uint32_t pre = incl - mine;
  for (int k = 0; k < wave; ++k) pre += w.wave_tot[k];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int b = tid * CPT + k;
    if (b < nb) {
      w.start[b] = pre;
      w.cnt[b] = pre;  // running cursor
    }
    pre += c[k];
  }
  if (tid == 0) w.start[nb] = static_cast<uint32_t>(m);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kBktRows; ++i) {
    if (dig[i] >= 0) {
      const uint32_t pos = atomicAdd(&w.cnt[dig[i]], 1u);
      w.keys[pos] = key[i];
      w.idx[pos] = id[i];
    }
  }
  __syncthreads();
  for (int i = tid; i < m; i += T) {
    const uint64_t ki = w.keys[i];
    const uint32_t ii = w.idx[i];
    const uint32_t d = digit_of(ki);
    const int bs = static_cast<int>(w.start[d]);
    const int be = static_cast<int>(w.start[d + 1]);
    int rank = 0;
    for (int j = bs; j < be; j += 4) {
      uint64_t kj[4];
      uint32_t ij[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jj = (j + u) < be ? (j + u) : (be - 1);
        kj[u] = w.keys[jj];
        ij[u] = w.idx[jj];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool less = kj[u] < ki || (kj[u] == ki && ij[u] < ii);
        rank += (less && (j + u) < be) ? 1 : 0;
      }
    }
    a.out_final[lo + bs + rank] = ii;
  }
}

// Second form of the LDS-resident finish (default; sort_msd_bucket_v2 = 0 selects the one above).  Same result,
// cheaper phases:
//   * ONE pass of LDS atomics: the returning add that counts a sub-bucket also hands the row its arrival rank
//     inside it, so after the scan the row's LDS slot is start[digit] + rank (no second cursor pass);
//   * up to 4096 (T = 1024) / 2048 (T = 512) sub-buckets instead of 1024, counters scanned in place: sub-buckets
//     of ~2-4 rows make the final ranking loop one short iteration for almost every row (it was the dominant
//     phase at ~8 rows: LDS reads grow with the square of the sub-bucket size).
template <int T, int CPT = 4, int R = kBktRows>
struct __attribute__((aligned(16))) MsdBucket2Lds {
  uint64_t keys[T * R];
  uint32_t idx[T * R];
  uint32_t start[CPT * T + 1];   // counts, then exclusive starts (+ sentinel)
  uint32_t wave_tot[T / 64];
};

// AOS: the bucket's rows are 12-byte records starting at record part_in[q] of `keys` (idx unused)
// CPT: sub-bucket counters per thread of the scan (sub-buckets <= CPT * T); R: rows per thread (bucket <= T * R rows)
template <bool SPL, int T, bool AOS = false, int CPT = 4, int R = kBktRows>
__global__ __launch_bounds__(T) void msd_bucket2_kernel(MsdArgs a, const uint64_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ idx) {
  __shared__ MsdBucket2Lds<T, CPT, R> w;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t q = (a.xcd_map & 2) ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int64_t lo = a.part_start[q];
  const int m = static_cast<int>(static_cast<int64_t>(a.part_start[q + 1]) - lo);
  const int64_t lo_in = AOS ? static_cast<int64_t>(a.part_in[q]) : lo;
  const MsdRec* __restrict__ recs = reinterpret_cast<const MsdRec*>(keys);
  if (m == 0) return;  // workgroup-uniform
  if (m > T * R) {
    if (tid == 0) atomicOr(a.overflow, 2u);
    return;
  }
  const int nb = 1 << a.b3;   // <= NB (host)
  const int dshift = 64 - a.bits - a.b3;
  const uint32_t dmask = static_cast<uint32_t>(nb - 1);
  uint64_t klo = 0;
  double kinv = 0.0;
  if constexpr (SPL) {
    const uint32_t nparts = 1u << a.bits;
    klo = q > 0 ? a.spl[q - 1] : 0;
    const uint64_t khi = q + 1 < nparts ? a.spl[q] : ~uint64_t(0);
    const double range = static_cast<double>(khi - klo) + 1.0;
    kinv = static_cast<double>(nb) / range;
  }
  auto digit_of = [&](uint64_t k) -> uint32_t {
    if constexpr (SPL) {
      const uint32_t d = static_cast<uint32_t>(static_cast<double>(k - klo) * kinv);
      return d < static_cast<uint32_t>(nb) ? d : static_cast<uint32_t>(nb - 1);
    } else {
      return a.b3 == 0 ? 0u : (static_cast<uint32_t>((k << a.kshift) >> dshift) & dmask);
    }
  };
  for (int i = tid; i < nb; i += T) w.start[i] = 0;
  uint64_t key[R];
  uint32_t id[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {   // unconditional loads (clamped): all round trips overlap
    const int p = i * T + tid;
    const int64_t r = lo_in + (p < m ? p : m - 1);
    if constexpr (AOS) {
      const MsdRec rr = recs[r];
      key[i] = msd_rec_key(rr);
      id[i] = rr.idx;
    } else {
      key[i] = keys[r];
      id[i] = idx[r];
    }
  }
  __syncthreads();
  uint32_t dig[R], rank[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    dig[i] = digit_of(key[i]);
    rank[i] = 0;
    if (i * T + tid < m) rank[i] = atomicAdd(&w.start[dig[i]], 1u);
  }
  __syncthreads();
  // in-place exclusive scan of nb <= CPT * T counters: CPT consecutive counters per thread
  uint32_t c[CPT];
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int b = tid * CPT + k;
    c[k] = b < nb ? w.start[b] : 0u;
    mine += c[k];
  }
  const uint32_t incl = wave_inclusive_scan_u32(mine);
  if (lane == 63) w.wave_tot[wave] = incl;
  __syncthreads();
  uint32_t pre = incl - mine;
  for (int k = 0; k < wave; ++k) pre += w.wave_tot[k];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int b = tid * CPT + k;
    if (b < nb) w.start[b] = pre;
    pre += c[k];
  }
  if (tid == 0) w.start[nb] = static_cast<uint32_t>(m);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < R; ++i) {
    if (i * T + tid < m) {
      const uint32_t pos = w.start[dig[i]] + rank[i];
      w.keys[pos] = key[i];
      w.idx[pos] = id[i];
    }
  }
  __syncthreads();
  // rank inside the sub-bucket (~2 rows): keys only, two per step; row ids are read only where a key repeats.  The
  // finish is bound by its LDS traffic (profiles/r03_i): this loop was 9 of the ~13 LDS accesses per row when it read
  // four (key, row id) pairs per step.
  for (int i = tid; i < m; i += T) {
    const uint64_t ki = w.keys[i];
    const uint32_t d = digit_of(ki);
    const int bs = static_cast<int>(w.start[d]);
    const int be = static_cast<int>(w.start[d + 1]);
    int rk = 0;
    bool tie = false;
    for (int j = bs; j < be; j += 2) {
      const bool two = j + 1 < be;
      const uint64_t k0 = w.keys[j];
      const uint64_t k1 = w.keys[two ? j + 1 : j];
      rk += (k0 < ki ? 1 : 0) + ((two && k1 < ki) ? 1 : 0);
      tie = tie || (k0 == ki && j != i) || (two && k1 == ki && j + 1 != i);
    }
    const uint32_t ii = w.idx[i];
    if (tie) {   // equal keys keep the order of their row ids
      for (int j = bs; j < be; ++j) rk += (w.keys[j] == ki && w.idx[j] < ii) ? 1 : 0;
    }
    a.out_final[lo + bs + rk] = ii;
  }
}

// The same finish over rec8 words (msdw_word: 32 key bits above the row id).  A bucket is T * R words of LDS — a third
// less than (key, row id) pairs, one LDS access per row where the pair form made two — and ranking compares whole words:
// inside a sub-bucket that is the order of (key bits in the word, row id).  Rows whose 32 key bits tie with another row
// of their sub-bucket (2e9 uniform keys: 0.9 per 1000) read the full keys of both from the caller's column and correct
// their rank by the bits below; when the word already holds every remaining key bit (32-bit key types, shared prefixes)
// a tie is a tie.  A bucket in which more than (its rows >> tie_shift) rows tie, or a row with more than kMsdwMaxTied tied
// neighbours, gives the attempt up (bit 32: later workgroups return at once) and the host repeats the sort with full
// records — duplicate-heavy keys would pay 1 + (tied neighbours) random 8-byte reads per row here.
constexpr int kMsdwTieSlot0 = 64;   // flags[64 + 32 k], k < 32: tie counts (one 128-byte line each)
constexpr int kMsdwFlagWords = kMsdwTieSlot0 + 32 * 32;
constexpr int kMsdwMaxTied = 16;   // rec8 finish: a row with more tied neighbours than this gives the attempt up
template <int T, int CPT = 4, int R = kBktRows>
struct __attribute__((aligned(16))) MsdBucket2wLds {
  uint64_t words[T * R];
  uint32_t start[CPT * T + 1];   // counts, then exclusive starts (+ sentinel)
  uint32_t wave_tot[T / 64];
  uint32_t ties;                 // rows of this bucket that read their full key
};

template <int T, int CPT = 4, int R = kBktRows>
__global__ __launch_bounds__(T) void msd_bucket2w_kernel(MsdArgs a, const uint64_t* __restrict__ words_in) {
  __shared__ MsdBucket2wLds<T, CPT, R> w;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // given up: the call is repeated.  ONE thread looks and tells the others — the bit may appear between the loads of two
  // threads, and a workgroup of which some threads left ranks garbage and reads the column at garbage row ids
  // (profiles/r05_e: "illegal memory access" on a few runs in ten)
  if (tid == 0) w.ties = __atomic_load_n(a.overflow, __ATOMIC_RELAXED) & 32u;
  __syncthreads();
  if (w.ties != 0) return;   // workgroup-uniform
  __syncthreads();           // (everyone has read it before thread 0 reuses the word as the tie counter)
  const uint32_t q = (a.xcd_map & 2) ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int64_t lo = a.part_start[q];
  const int m = static_cast<int>(static_cast<int64_t>(a.part_start[q + 1]) - lo);
  const int64_t lo_in = static_cast<int64_t>(a.part_in[q]);
  if (m == 0) return;  // workgroup-uniform
  if (m > T * R) {
    if (tid == 0) atomicOr(a.overflow, 2u);
    return;
  }
  const int nb = 1 << a.b3;   // <= CPT * T (host)
  const int dsh = 64 - a.b3;
  auto digit_of = [&](uint64_t wd) -> uint32_t { return a.b3 == 0 ? 0u : static_cast<uint32_t>((wd << a.b2) >> dsh); };
  const bool bits_below = a.kshift + a.b1 + 32 < 64;   // key bits the words do not hold
  // this bucket's share of the tie budget: m >> tie_shift rows (one counter per address serialises at ~40 ns per
  // returning atomic — 1.8e6 tied rows of 2e9 cost 69 ms that way, profiles/r05_a — so the budget is kept per bucket, in LDS)
  // (a floor of 64 rows — m >> shift is 0 for a bucket of a few rows, and ONE duplicated outlier key in such a
  //  bucket would then abandon rec8 for the whole input; tie_shift >= 31 keeps its meaning of "no budget at all": tests)
  const uint32_t tie_share = static_cast<uint32_t>(m) >> (a.tie_shift < 31 ? a.tie_shift : 31);
  const uint32_t tie_limit = a.tie_shift >= 31 ? 0u : (tie_share > 64u ? tie_share : 64u);
  for (int i = tid; i < nb; i += T) w.start[i] = 0;
  // (round 6: the kernel's VALU is its bound — SQ_ACTIVE_INST_VALU is 14 % of the wave cycles with seven waves a SIMD,
  //  profiles/r06_j_* — and a bucket fills 40 % of the T * R slots on average: the slot loops skip the rounds i * T >= m,
  //  a workgroup-uniform test, instead of running all R of them over clamped duplicates)
  uint64_t wd[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {   // unconditional loads (clamped) inside a round: all round trips overlap
    const int p = i * T + tid;
    wd[i] = 0;
    if (i * T < m) wd[i] = words_in[lo_in + (p < m ? p : m - 1)];
  }
  __syncthreads();
  uint32_t dig[R], rank[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    dig[i] = 0;
    rank[i] = 0;
    if (i * T < m) {
      dig[i] = digit_of(wd[i]);
      if (i * T + tid < m) rank[i] = atomicAdd(&w.start[dig[i]], 1u);
    }
  }
  __syncthreads();
  uint32_t c[CPT];
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int b = tid * CPT + k;
    c[k] = b < nb ? w.start[b] : 0u;
    mine += c[k];
  }
  const uint32_t incl = wave_inclusive_scan_u32(mine);
  if (lane == 63) w.wave_tot[wave] = incl;
  __syncthreads();
  uint32_t pre = incl - mine;
  for (int k = 0; k < wave; ++k) pre += w.wave_tot[k];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int b = tid * CPT + k;
    if (b < nb) w.start[b] = pre;
    pre += c[k];
  }
  if (tid == 0) w.start[nb] = static_cast<uint32_t>(m);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < R; ++i) {
    if (i * T < m) {
      if (i * T + tid < m) w.words[w.start[dig[i]] + rank[i]] = wd[i];
    }
  }
  __syncthreads();
  for (int i = tid; i < m; i += T) {
    const uint64_t wi = w.words[i];
    const uint32_t d = digit_of(wi);
    const int bs = static_cast<int>(w.start[d]);
    const int be = static_cast<int>(w.start[d + 1]);
    int rk = 0;
    int tied = 0;   // other rows of the sub-bucket with the same 32 key bits
    for (int j = bs; j < be; j += 2) {
      const bool two = j + 1 < be;
      const uint64_t w0 = w.words[j];
      const uint64_t w1 = w.words[two ? j + 1 : j];
      rk += (w0 < wi ? 1 : 0) + ((two && w1 < wi) ? 1 : 0);
      tied += ((((w0 ^ wi) >> 32) == 0 && j != i) ? 1 : 0) + ((two && ((w1 ^ wi) >> 32) == 0 && j + 1 != i) ? 1 : 0);
    }
    if (tied > kMsdwMaxTied && bits_below) {   // a run of equal words: quadratic in column reads — not this form's input
      if ((__atomic_load_n(a.overflow, __ATOMIC_RELAXED) & 32u) == 0) atomicOr(a.overflow, 32u);
    } else if (tied != 0 && bits_below) {   // (rare) the bits below decide before the row ids do
      const uint64_t ki = load_key_typed(a.src_keys, static_cast<int64_t>(static_cast<uint32_t>(wi)), a.raw);
      for (int j = bs; j < be; ++j) {
        const uint64_t wj = w.words[j];
        if (j != i && ((wj ^ wi) >> 32) == 0) {
          const uint64_t kj = load_key_typed(a.src_keys, static_cast<int64_t>(static_cast<uint32_t>(wj)), a.raw);
          const bool counted = wj < wi;
          const bool before = kj < ki || (kj == ki && counted);
          rk += (before ? 1 : 0) - (counted ? 1 : 0);
        }
      }
      if (atomicAdd(&w.ties, 1u) == tie_limit && (__atomic_load_n(a.overflow, __ATOMIC_RELAXED) & 32u) == 0) {
        atomicOr(a.overflow, 32u);   // (once per bucket at most)
      }
    }
    a.out_final[lo + bs + rk] = static_cast<uint32_t>(wi);
  }
  __syncthreads();
  // statistics only (arx_get_counter "sort_wide_rec8_ties"): one atomic per bucket with ties, spread over 32 lines
  if (tid == 0 && w.ties != 0) atomicAdd(&a.overflow[kMsdwTieSlot0 + 32 * (q & 31u)], w.ties);
}

// (A persistent form of this finish — workgroups walking buckets q, q + grid, ... and loading the next bucket's rows
// into registers before ranking the current one — measured SLOWER: 13.7 ms against 11.4 ms for 2^19 buckets, and a
// grid of one workgroup per CU instead of two 43 ms end to end instead of 36: the finish is bound by its LDS phases and
// by how many workgroups interleave them, not by the latency of its loads.  profiles/r03_i_sort_persistent_bucket_ab.txt)
// part_count + part_start + cursor2 (2^14 + 1 each), hist1 (128 x 2048), l1_start, l2_tile_start, flag
constexpr int kMsdSplBits = 15;  // sampled-splitter mode: up to 2^15 buckets
constexpr size_t kMsdTableWords = (size_t(1) << kMsdSplBits) + 64;
constexpr size_t kMsdTableBytes = (3 * kMsdTableWords + size_t(128) * kMsdMaxChunks + 2 * 192 + 64) * 4 + (size_t(1) << kMsdSplBits) * 8;

// tables of the wide two-level form (run_msd_sort_wide): 3 arrays of 2^20 buckets + 10 of 1024 level-1 entries
constexpr int kMsdwMaxBins = 1024;    // level 1 (one thread per bin in the one-workgroup scans)
constexpr int kMsdwMaxBins2 = 4096;   // level 2 (inside a level-1 bucket: short runs are fine there, xcd_contiguous)
constexpr int kMsdwMaxBits = 20;
constexpr size_t kMsdwTableBytes = (3 * ((size_t(1) << kMsdwMaxBits) + 64) + 10 * (kMsdwMaxBins + 64) + kMsdwMaxBins * 32) * 4;   // (+ the write-combined level 1's cursors, a 128-byte line each)

// The wide form lays its buckets out with room to spare instead of counting them exactly first: level-1 buckets
// sized from a sampled histogram get est/32 + min(est/8 + 1, 16384) more rows, level-2 buckets a fixed room of
// mean + 6 sqrt(mean) + 64 rows (mean > 2048: at most 22 % more).  Both fit n + this many records, checked on the
// device; what does not fit falls back to exact counts.
static inline int64_t msdw_slack_rows(int64_t n) { return n / 4 + 4096; }

struct SortPlan {
  int64_t n;          // rows to sort (non-null)
  int64_t ntiles;
  int64_t chunk_tiles;
  int64_t nchunks;
  size_t off_keys_a, off_keys_b, off_idx_a, off_idx_b, off_hist, off_totals, off_split, off_msd, off_float_bits, off_rows, off_sel_ws, total;
};

static SortPlan make_plan(int64_t length) {
  SortPlan p{};
  p.n = length;
  p.ntiles = ceil_div(std::max<int64_t>(length, 1), kSortTile);
  p.chunk_tiles = std::max<int64_t>(1, ceil_div(p.ntiles, int(g_sort_chunks)));
  p.nchunks = ceil_div(p.ntiles, p.chunk_tiles);
  auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
  const size_t n = static_cast<size_t>(std::max<int64_t>(length, 1));
  size_t o = 0;
  // (keys_a, idx_a) and (keys_b, idx_b) are adjacent: the wide form uses either pair as ONE array of nx 12-byte records
  const size_t nx = n + static_cast<size_t>(msdw_slack_rows(static_cast<int64_t>(n)));
  p.off_keys_a = o; o = align(o + nx * 8);
  p.off_idx_a = o; o = align(o + nx * 4);
  p.off_keys_b = o; o = align(o + nx * 8);
  p.off_idx_b = o; o = align(o + nx * 4);
  p.off_hist = o; o = align(o + static_cast<size_t>(kDigits) * kMaxChunks * 4);
  p.off_totals = o; o = align(o + static_cast<size_t>(kDigits) * 4);
  p.off_split = o; o = align(o + static_cast<size_t>(kDigits) * 4);
  p.off_msd = o; o = align(o + std::max(kMsdTableBytes, kMsdwTableBytes));
  p.off_float_bits = o; o = align(o + 2 * (n / 64 + 2) * 8);  // sortable / NaN bitmaps of float keys
  p.off_rows = o; o = align(o + n * 4);  // row ids of the non-null / null partitions
  p.off_sel_ws = o; o = align(o + selection_workspace_bytes(length));
  p.total = o;
  return p;
}

// which record form the wide sorts of this process ran with (arx_get_counter; tests and the bench's parity leg)
static std::atomic<int64_t> g_sort_wide_runs{0}, g_sort_wide_rec8_runs{0}, g_sort_wide_rec8_ties{0}, g_sort_wide_rec8_given_up{0},
    g_sort_wide_wc_runs{0}, g_sort_wide_wc_given_up{0};

int get_sort_counter(const char* name, int64_t* out) {
  if (strcmp(name, "sort_wide_runs") == 0) *out = g_sort_wide_runs.load();
  else if (strcmp(name, "sort_wide_rec8_runs") == 0) *out = g_sort_wide_rec8_runs.load();
  else if (strcmp(name, "sort_wide_rec8_ties") == 0) *out = g_sort_wide_rec8_ties.load();
  else if (strcmp(name, "sort_wide_rec8_given_up") == 0) *out = g_sort_wide_rec8_given_up.load();
  else if (strcmp(name, "sort_wide_wc_runs") == 0) *out = g_sort_wide_wc_runs.load();
  else if (strcmp(name, "sort_wide_wc_given_up") == 0) *out = g_sort_wide_wc_given_up.load();
  else return 0;
  return 1;
}

int set_sort_option(const char* name, int64_t value) {
  if (strcmp(name, "sort_msd") == 0) {
    g_sort_msd = value < 0 ? -1 : (value != 0);
    return 1;
  }
  if (strcmp(name, "sort_msd_sampled") == 0) {
    g_sort_msd_sampled = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(value, 2)));
    return 1;
  }
  if (strcmp(name, "sort_msd_fused") == 0) {
    g_sort_msd_fused = value != 0;
    return 1;
  }
  if (strcmp(name, "sort_msd_final_rows_log2") == 0) {
    g_sort_msd_final_rows_log2 = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(value, 8)));
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_sample_shift") == 0) {
    g_sort_msd_wide_sample_shift = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(value, 8)));
    return 1;
  }
  if (strcmp(name, "sort_xcd_map") == 0) {
    g_sort_xcd_map = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(value, 7)));
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_bits") == 0) {
    g_sort_msd_wide_bits = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(value, kMsdwMaxBits)));
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_b2max") == 0) {
    g_sort_msd_wide_b2max = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(value, 12)));
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_rpt1") == 0 || strcmp(name, "sort_msd_wide_rpt2") == 0) {
    const int rpt = value >= 24 ? 24 : value >= 16 ? 16 : 8;
    (name[17] == '1' ? g_sort_msd_wide_rpt1 : g_sort_msd_wide_rpt2) = rpt;
    return 1;
  }
  if (strcmp(name, "sort_msd_bucket_cpt") == 0) {
    g_sort_msd_bucket_cpt = value >= 8 ? 8 : 4;
    return 1;
  }
  if (strcmp(name, "sort_msd_tiny_bucket") == 0) {
    g_sort_msd_tiny_bucket = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(value, 2)));   // 2: also the 9-rows-per-thread form
    return 1;
  }
  if (strcmp(name, "sort_msd_prefix") == 0) {
    g_sort_msd_prefix = value != 0;
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_gap2") == 0) {
    g_sort_msd_wide_gap2 = value != 0;
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_rec8") == 0) {
    g_sort_msd_wide_rec8 = value != 0;
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_wc") == 0) {
    g_sort_msd_wide_wc = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(value, 4096)));
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_l2w") == 0) {
    g_sort_msd_wide_l2w = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(value, 3)));
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_wc_prefetch") == 0) {
    g_sort_msd_wide_wc_prefetch = value != 0;
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_wc_rows") == 0) {
    g_sort_msd_wide_wc_rows = value == 8 ? 8 : value == 6 ? 6 : 4;
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_wc_min_rows") == 0) {
    g_sort_msd_wide_wc_min_rows = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(value, 1 << 30)));
    return 1;
  }
  if (strcmp(name, "sort_records_in_place") == 0) {
    g_sort_records_in_place = value != 0 ? 1 : 0;
    return 1;
  }
  if (strcmp(name, "sort_vary_sample_shift") == 0) {
    g_sort_vary_sample_shift = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(value, 8)));
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_wc_typed") == 0) {
    g_sort_msd_wide_wc_typed = value == 2 ? 2 : (value != 0 ? 1 : 0);
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_wc_form") == 0) {
    g_sort_msd_wide_wc_form = value == 1 ? 1 : 2;
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_rec8_tie_shift") == 0) {
    g_sort_msd_wide_rec8_tie_shift = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(value, 40)));
    return 1;
  }
  if (strcmp(name, "sort_msd_wide_sample_strict") == 0) {
    g_sort_msd_wide_sample_strict = value != 0;
    return 1;
  }
  if (strcmp(name, "sort_msd_wide") == 0) {
    g_sort_msd_wide = value != 0;
    return 1;
  }
  if (strcmp(name, "sort_msd_bucket_v2") == 0) {
    g_sort_msd_bucket_v2 = value != 0;
    return 1;
  }
  if (strcmp(name, "sort_msd_small_bucket") == 0) {
    g_sort_msd_small_bucket = value != 0;
    return 1;
  }
  if (strcmp(name, "sort_msd_seg_min_bits") == 0) {
    g_sort_msd_seg_min_bits = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(value, 7)));
    return 1;
  }
  if (strcmp(name, "sort_msd_segment_rows") == 0) {
    g_sort_msd_segment_rows = std::max<int64_t>(1024, value);
    return 1;
  }
  if (strcmp(name, "sort_msd_global_bits") == 0) {
    g_sort_msd_global_bits = static_cast<int>(std::max<int64_t>(2, std::min<int64_t>(value, kMsdMaxBits)));
    return 1;
  }
  if (strcmp(name, "sort_msd_min_rows") == 0) {
    g_sort_msd_min_rows = static_cast<int>(std::max<int64_t>(256, std::min<int64_t>(value, INT32_MAX)));
    return 1;
  }
  if (strcmp(name, "sort_fuse_prep") == 0) {
    g_sort_fuse_prep = value != 0;
    return 1;
  }
  if (strcmp(name, "sort_chunks") == 0) {
    g_sort_chunks = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(value, kMaxChunks)));
    return 1;
  }
  return 0;
}

// Runs the MSD-hybrid path.  *overflowed = 1 if a bucket did not fit the final window (the caller
// then falls back to the LSD path).  Synchronous (reads the flag back).
// Bits in which the (order-transformed) keys differ from key 0, OR-ed into *out: the leading zeros of the result are
// bits EVERY key shares — row ids, timestamps, small integers share most of their top bits, and an MSD digit taken
// there puts every row into one bucket.  sample_shift > 0: only one tile of 8192 rows in 2^sample_shift is read (plus
// the first and the last tile) — enough to say "no shared prefix" for free; a shared prefix seen in the sample is then
// confirmed over all rows before anything relies on it.
// (keys == nullptr, !RAW: the keys are those of the 12-byte records `recs` — the sharded sort's receiver, round 6)
template <bool RAW>
__global__ __launch_bounds__(kBlock) void sort_vary_kernel(const uint64_t* __restrict__ keys, int xf, int64_t n, int sample_shift,
                                                           unsigned long long* __restrict__ out, const MsdRec* __restrict__ recs = nullptr) {
  constexpr int64_t kTile = 8192;
  const int64_t ntiles = (n + kTile - 1) / kTile;
  const uint64_t ref = RAW ? load_key_typed(keys, 0, xf) : (keys != nullptr ? keys[0] : msd_rec_key(recs[0]));
  unsigned long long vary = 0;
  for (int64_t g = blockIdx.x; ; g += gridDim.x) {
    int64_t tile = g;
    if (sample_shift > 0) {
      const int64_t groups = (ntiles + (int64_t(1) << sample_shift) - 1) >> sample_shift;
      if (g >= groups + 1) break;
      tile = g == groups ? ntiles - 1
                         : (g << sample_shift) + ((static_cast<uint32_t>(g) * 2654435761u) >> (32 - sample_shift));
      if (tile >= ntiles) tile = ntiles - 1;
    } else if (g >= ntiles) {
      break;
    }
    const int64_t begin = tile * kTile;
    const int64_t end = begin + kTile < n ? begin + kTile : n;
    for (int64_t r = begin + threadIdx.x; r < end; r += kBlock) {
      const uint64_t k = RAW ? load_key_typed(keys, r, xf) : (keys != nullptr ? keys[r] : msd_rec_key(recs[r]));
      vary |= k ^ ref;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) vary |= __shfl_xor(vary, d, 64);
  if ((threadIdx.x & 63) == 0 && vary != 0) atomicOr(out, vary);
}

// Leading bits shared by all n keys (0 when sort_msd_prefix is off, when there are none, or when all keys are equal).
static int sort_shared_prefix_bits(const uint64_t* keys, int raw_xf, int64_t n, unsigned long long* d_word, hipStream_t st,
                                   int* out_bits, const MsdRec* recs = nullptr) {
  *out_bits = 0;
  if (!g_sort_msd_prefix || n < 2) return ARX_OK;
  for (int pass = 0; pass < 2; ++pass) {
    const int sample_shift = pass == 0 && n > (int64_t(1) << 20) ? int(g_sort_vary_sample_shift) : 0;
    ARX_HIP(hipMemsetAsync(d_word, 0, 8, st));
    const int64_t ntiles = ceil_div(n, 8192);
    const int64_t work = sample_shift ? (ntiles >> sample_shift) + 2 : ntiles;
    const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(work, 256 * 16)));
    if (raw_xf != 0) {
      hipLaunchKernelGGL((sort_vary_kernel<true>), dim3(grid), dim3(kBlock), 0, st, keys, raw_xf, n, sample_shift, d_word);
    } else {
      hipLaunchKernelGGL((sort_vary_kernel<false>), dim3(grid), dim3(kBlock), 0, st, keys, 0, n, sample_shift, d_word, recs);
    }
    ARX_CHECK_LAUNCH("sort_vary_kernel");
    unsigned long long vary = 0;
    ARX_HIP(hipMemcpyAsync(&vary, d_word, 8, hipMemcpyDeviceToHost, st));
    ARX_HIP(hipStreamSynchronize(st));
    const int shared = vary == 0 ? 64 : __builtin_clzll(vary);
    if (shared == 0) return ARX_OK;            // (also what a sample says for full-range keys: no second pass)
    if (sample_shift == 0) {                   // exact
      *out_bits = shared >= 64 ? 0 : shared;   // all keys equal: nothing to gain, the LSD passes sort them
      return ARX_OK;
    }
  }
  return ARX_OK;
}

static int run_msd_sort(const uint64_t* src_keys, const uint32_t* src_idx, int raw, int64_t n,
                        uint64_t* keys_x, uint32_t* idx_x, uint64_t* keys_y, uint32_t* idx_y,
                        uint8_t* tables, uint64_t* out_final, hipStream_t st, int* overflowed,
                        int kshift = 0) {
  *overflowed = 0;
  if (n == 0) return ARX_OK;
  MsdArgs a{};
  a.kshift = kshift;
  a.xcd_map = g_sort_xcd_map;
  a.src_keys = src_keys;
  a.src_idx = src_idx;
  a.raw = raw;
  a.n = n;
  int lg = 0;
  while ((int64_t(1) << (lg + 1)) <= n) ++lg;
  int total = lg - g_sort_msd_final_rows_log2;  // default 3: ~8-16 rows per final bucket
  total = std::max(2, std::min(total, std::min(kMsdMaxBits + 9, 64 - kshift)));
  a.bits = std::max(2, std::min(total, int(g_sort_msd_global_bits)));
  total = std::min(total, a.bits + 9);
  a.b1 = (a.bits + 1) / 2;
  a.b2 = a.bits - a.b1;
  a.b3 = total - a.bits;
  const int64_t ntiles = ceil_div(n, kMsdTile);
  const int64_t chunk_tiles = std::max<int64_t>(1, ceil_div(ntiles, kMsdMaxChunks));
  a.chunk_rows = chunk_tiles * kMsdTile;
  a.nchunks = ceil_div(ntiles, chunk_tiles);
  const size_t np = kMsdTableWords;
  uint32_t* t = reinterpret_cast<uint32_t*>(tables);
  a.part_count = t;
  a.part_start = t + np;
  a.cursor2 = t + 2 * np;
  a.hist1 = t + 3 * np;
  a.l1_start = a.hist1 + size_t(128) * kMsdMaxChunks;
  a.l2_tile_start = a.l1_start + 192;
  a.overflow = a.l2_tile_start + 192;
  a.keys_x = keys_x;
  a.idx_x = idx_x;
  a.keys_y = keys_y;
  a.idx_y = idx_y;
  a.out_final = out_final;
  const int nparts = 1 << a.bits;
  ARX_HIP(hipMemsetAsync(a.part_count, 0, static_cast<size_t>(nparts) * 4, st));
  ARX_HIP(hipMemsetAsync(a.overflow, 0, 4, st));
  const unsigned nch = static_cast<unsigned>(a.nchunks);
  if (raw) {
    hipLaunchKernelGGL((msd_hist_kernel<true>), dim3(nch), dim3(kMsdThreads), 0, st, a);
  } else {
    hipLaunchKernelGGL((msd_hist_kernel<false>), dim3(nch), dim3(kMsdThreads), 0, st, a);
  }
  ARX_CHECK_LAUNCH("msd_hist_kernel");
  hipLaunchKernelGGL(msd_scan_a_kernel, dim3(1), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(msd_scan_b_kernel, dim3(1u << a.b1), dim3(64), 0, st, a);
  ARX_CHECK_LAUNCH("msd_scan kernels");
  const bool fused = g_sort_msd_fused != 0 && (n >> a.bits) <= 8192;
  unsigned int max_part = 0;
  if (fused) {
    // skewed top bits (e.g. normally distributed floats): a bucket would not fit LDS -> do not
    // waste the scatter passes, the caller runs the LSD passes instead
    ARX_HIP(hipMemcpyAsync(&max_part, a.overflow + 1, 4, hipMemcpyDeviceToHost, st));
    ARX_HIP(hipStreamSynchronize(st));
    if (max_part > static_cast<unsigned int>(kBktCap)) {
      *overflowed = 1;
      return ARX_OK;
    }
  }
  if (raw) {
    hipLaunchKernelGGL((msd_scatter1_kernel<true>), dim3(nch), dim3(kMsdThreads), 0, st, a);
  } else {
    hipLaunchKernelGGL((msd_scatter1_kernel<false>), dim3(nch), dim3(kMsdThreads), 0, st, a);
  }
  ARX_CHECK_LAUNCH("msd_scatter1_kernel");
  const unsigned grid2 = static_cast<unsigned>(ceil_div(n, kMsdTile) + (int64_t(1) << a.b1));
  hipLaunchKernelGGL(msd_scatter2_kernel, dim3(grid2), dim3(kMsdThreads), 0, st, a);
  ARX_CHECK_LAUNCH("msd_scatter2_kernel");
  // level-2 buckets that fit LDS: finish each one in a single workgroup (b3 may use 10 bits there)
  if (fused) {
    const bool small = g_sort_msd_small_bucket != 0 && max_part <= static_cast<unsigned int>(kBktCapSmall);
    const int want_b3 = lg - (g_sort_msd_final_rows_log2 - 1) - a.bits;   // default: ~4 rows per sub-bucket
    if (g_sort_msd_bucket_v2) {
      a.b3 = std::max(0, std::min(std::min(want_b3, small ? 11 : 12), 64 - kshift - a.bits));
      if (small) {
        hipLaunchKernelGGL((msd_bucket2_kernel<false, kBktThreadsSmall>), dim3(static_cast<unsigned>(nparts)),
                           dim3(kBktThreadsSmall), 0, st, a, keys_y, idx_y);
      } else {
        hipLaunchKernelGGL((msd_bucket2_kernel<false, kBktThreads>), dim3(static_cast<unsigned>(nparts)),
                           dim3(kBktThreads), 0, st, a, keys_y, idx_y);
      }
    } else {
      a.b3 = std::max(0, std::min(std::min(want_b3, 10), 64 - kshift - a.bits));
      if (small) {
        hipLaunchKernelGGL((msd_bucket_kernel<false, kBktThreadsSmall>), dim3(static_cast<unsigned>(nparts)),
                           dim3(kBktThreadsSmall), 0, st, a, keys_y, idx_y);
      } else {
        hipLaunchKernelGGL((msd_bucket_kernel<false, kBktThreads>), dim3(static_cast<unsigned>(nparts)),
                           dim3(kBktThreads), 0, st, a, keys_y, idx_y);
      }
    }
    ARX_CHECK_LAUNCH("msd_bucket_kernel");
  } else {
    const uint64_t* fk = keys_y;
    const uint32_t* fi = idx_y;
    if (a.b3 > 0) {
      hipLaunchKernelGGL(msd_local_kernel, dim3(static_cast<unsigned>(nparts)), dim3(kMsdThreads), 0, st, a);
      ARX_CHECK_LAUNCH("msd_local_kernel");
      fk = keys_x;
      fi = idx_x;
    }
    hipLaunchKernelGGL(msd_final_kernel, dim3(static_cast<unsigned>(ceil_div(n, kMsdCore))), dim3(256), 0, st, a,
                       fk, fi, 64 - (a.bits + a.b3));
    ARX_CHECK_LAUNCH("msd_final_kernel");
  }
  unsigned int flag = 0;
  ARX_HIP(hipMemcpyAsync(&flag, a.overflow, 4, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  *overflowed = flag != 0;
  return ARX_OK;
}


// The sampled-splitter pipeline (n <= 2^27 + slack rows, >= 2^18).  `lsd_*` = scratch of the LSD
// kernels used to sort the sample.  Synchronous.
static int run_msd_sort_sampled(const uint64_t* src_keys, const uint32_t* src_idx, int raw, int64_t n,
                                uint64_t* keys_x, uint32_t* idx_x, uint64_t* keys_y, uint32_t* idx_y,
                                uint8_t* tables, uint32_t* lsd_hist, uint32_t* lsd_totals, uint64_t* out_final,
                                hipStream_t st, int* overflowed) {
  *overflowed = 0;
  if (n == 0) return ARX_OK;
  int bits = 2;
  while (bits < kMsdSplBits && (n >> bits) > 4096) ++bits;
  if ((n >> bits) > 6144) {  // even 2^15 buckets would average too close to the LDS capacity
    *overflowed = 1;
    return ARX_OK;
  }
  const int nparts = 1 << bits;
  int64_t m = 32 * static_cast<int64_t>(nparts);   // sample size: 32 keys per bucket
  while (m * 4 > n && m > nparts) m >>= 1;
  if (m * 4 > n) {
    *overflowed = 1;
    return ARX_OK;
  }
  MsdArgs a{};
  a.src_keys = src_keys;
  a.src_idx = src_idx;
  a.raw = raw;
  a.n = n;
  a.kshift = 0;
  a.bits = bits;
  a.b2 = std::min(8, bits / 2 + (bits & 1));
  a.b1 = bits - a.b2;            // <= 7
  int per_bucket_lg = 0;
  while ((int64_t(8) << per_bucket_lg) < (n >> bits) && per_bucket_lg < 10) ++per_bucket_lg;
  a.b3 = per_bucket_lg;          // ~8 rows per interpolated sub-bucket
  const int64_t ntiles = ceil_div(n, kMsdTile);
  const int64_t chunk_tiles = std::max<int64_t>(1, ceil_div(ntiles, kMsdMaxChunks));
  a.chunk_rows = chunk_tiles * kMsdTile;
  a.nchunks = ceil_div(ntiles, chunk_tiles);
  const size_t np = kMsdTableWords;
  uint32_t* t = reinterpret_cast<uint32_t*>(tables);
  a.part_count = t;
  a.part_start = t + np;
  a.cursor2 = t + 2 * np;
  a.hist1 = t + 3 * np;
  a.l1_start = a.hist1 + size_t(128) * kMsdMaxChunks;
  a.l2_tile_start = a.l1_start + 192;
  a.overflow = a.l2_tile_start + 192;
  uint64_t* spl = reinterpret_cast<uint64_t*>(a.overflow + 64);
  a.spl = spl;
  a.keys_x = keys_x;
  a.idx_x = idx_x;
  a.keys_y = keys_y;
  a.idx_y = idx_y;
  a.out_final = out_final;

  // ---- S1/S2: sample, sort the sample with the LSD kernels (scratch: the not-yet-used x buffers)
  // (x is written by level 1 only after the splitters have been picked; y may alias the source)
  uint64_t* s0 = keys_x;
  uint64_t* s1 = keys_x + m;
  uint32_t* i0 = idx_x;
  uint32_t* i1 = idx_x + m;
  const unsigned gs = static_cast<unsigned>(ceil_div(m, kBlock));
  if (raw) {
    hipLaunchKernelGGL((msd_sample_kernel<true>), dim3(gs), dim3(kBlock), 0, st, a, m, s0);
  } else {
    hipLaunchKernelGGL((msd_sample_kernel<false>), dim3(gs), dim3(kBlock), 0, st, a, m, s0);
  }
  ARX_CHECK_LAUNCH("msd_sample_kernel");
  {
    const int64_t stiles = ceil_div(m, kSortTile);
    const int64_t sct = std::max<int64_t>(1, ceil_div(stiles, int(g_sort_chunks)));
    const int64_t snch = ceil_div(stiles, sct);
    for (int pass = 0; pass < 8; ++pass) {
      hipLaunchKernelGGL(radix_hist_kernel, dim3(static_cast<unsigned>(snch)), dim3(kBlock), 0, st, s0, m, pass * 8,
                         sct * kSortTile, snch, lsd_hist, 0);
      hipLaunchKernelGGL(radix_digit_totals_kernel, dim3(kDigits), dim3(64), 0, st, lsd_hist, snch, lsd_totals);
      hipLaunchKernelGGL(radix_scan_kernel, dim3(kDigits), dim3(64), 0, st, lsd_hist, snch, lsd_totals);
      hipLaunchKernelGGL((radix_scatter_kernel<false>), dim3(static_cast<unsigned>(snch)), dim3(kBlock), 0, st, s0,
                         i0, m, pass * 8, sct, snch, lsd_hist, s1, i1, static_cast<uint64_t*>(nullptr), 0, 0);
      std::swap(s0, s1);
      std::swap(i0, i1);
    }
    ARX_CHECK_LAUNCH("sample sort");
  }
  hipLaunchKernelGGL(msd_pick_splitters_kernel, dim3(static_cast<unsigned>(ceil_div(nparts, kBlock))), dim3(kBlock),
                     0, st, s0, m, nparts, spl);
  ARX_CHECK_LAUNCH("msd_pick_splitters_kernel");

  // ---- S3/S4: level 1
  MsdArgs a1 = a;   // the scan kernels see level 1 as a single-level plan
  a1.bits = a.b1;
  a1.b2 = 0;
  const unsigned nch = static_cast<unsigned>(a.nchunks);
  const int nb1 = 1 << a.b1;
  ARX_HIP(hipMemsetAsync(a.part_count, 0, static_cast<size_t>(nb1) * 4, st));
  ARX_HIP(hipMemsetAsync(a.overflow, 0, 8, st));
  if (raw) {
    hipLaunchKernelGGL((msd_hist1_s_kernel<true>), dim3(nch), dim3(kMsdThreads), 0, st, a);
  } else {
    hipLaunchKernelGGL((msd_hist1_s_kernel<false>), dim3(nch), dim3(kMsdThreads), 0, st, a);
  }
  hipLaunchKernelGGL(msd_scan_a_kernel, dim3(1), dim3(1024), 0, st, a1);
  hipLaunchKernelGGL(msd_scan_b_kernel, dim3(static_cast<unsigned>(nb1)), dim3(64), 0, st, a1);
  if (raw) {
    hipLaunchKernelGGL((msd_scatter1_s_kernel<true>), dim3(nch), dim3(kMsdThreads), 0, st, a);
  } else {
    hipLaunchKernelGGL((msd_scatter1_s_kernel<false>), dim3(nch), dim3(kMsdThreads), 0, st, a);
  }
  ARX_CHECK_LAUNCH("msd sampled level 1");

  // ---- level-2 counts, bucket starts, cursors; bail out if a bucket cannot fit LDS
  const unsigned grid2 = static_cast<unsigned>(ceil_div(n, kMsdTile) + nb1);
  ARX_HIP(hipMemsetAsync(a.part_count, 0, static_cast<size_t>(nparts) * 4, st));
  hipLaunchKernelGGL(msd_hist2_s_kernel, dim3(grid2), dim3(kMsdThreads), 0, st, a);
  hipLaunchKernelGGL(msd_scan_a_kernel, dim3(1), dim3(1024), 0, st, a);
  ARX_CHECK_LAUNCH("msd sampled level-2 counts");
  unsigned int max_part = 0;
  ARX_HIP(hipMemcpyAsync(&max_part, a.overflow + 1, 4, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  if (max_part > static_cast<unsigned int>(kBktCap)) {  // > 10240 copies of one key, typically
    *overflowed = 1;
    return ARX_OK;
  }
  hipLaunchKernelGGL(msd_scatter2_s_kernel, dim3(grid2), dim3(kMsdThreads), 0, st, a);
  ARX_CHECK_LAUNCH("msd_scatter2_s_kernel");
  if (g_sort_msd_small_bucket != 0 && max_part <= static_cast<unsigned int>(kBktCapSmall)) {
    hipLaunchKernelGGL((msd_bucket_kernel<true, kBktThreadsSmall>), dim3(static_cast<unsigned>(nparts)),
                       dim3(kBktThreadsSmall), 0, st, a, keys_y, idx_y);
  } else {
    hipLaunchKernelGGL((msd_bucket_kernel<true, kBktThreads>), dim3(static_cast<unsigned>(nparts)),
                       dim3(kBktThreads), 0, st, a, keys_y, idx_y);
  }
  ARX_CHECK_LAUNCH("msd_bucket_kernel");
  unsigned int flag = 0;
  ARX_HIP(hipMemcpyAsync(&flag, a.overflow, 4, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  *overflowed = flag != 0;
  return ARX_OK;
}

// =====================================================================================
// Wide two-level form for inputs beyond 2^27 rows (configs[4]: 2e9 rows).  The segmented form below moves
// 104 B/row (an extra histogram + scatter level per segment and one pipeline launch sequence per segment); this
// one covers bits = ceil(log2 n) - 12 (<= 20) key bits with TWO scatter levels of up to 1024 bins each and moves
// 8 + 20 + 8 + 24 + 20 = 80 B/row in seven launches:
//   W1 msdw_hist0     top b1 bits (LDS histogram per chunk, <= 1024 atomics per workgroup)          8 B/row
//   W2 msdw_scan0     level-1 starts + cursors, tile map and histogram-unit map of level 2
//   W3 msdw_scatter1  level 1 over the whole input: 8192-row tiles (1024 threads), one returning global atomic
//                     per (tile, digit) — consecutive tiles extend the same <= 1024 output runs    8 + 12 B/row
//   W4 msdw_hist1     next b2 bits inside every level-1 bucket, work units of <= 2^19 rows of ONE bucket   8 B/row
//   W5 msdw_scan1     one workgroup per level-1 bucket: bucket starts, cursors, largest bucket
//   W6 msdw_scatter2  level 2 INSIDE each level-1 bucket (reads and writes stay within the bucket's few tens of
//                     MB — measured 25 % faster per byte than a scatter whose bins span the whole array) 12 + 12 B/row
//   W7 msd_bucket2    LDS finish of the 2^bits buckets (~2-4K rows each)                          12 + 8 B/row
// Neither level is stable: the finish orders by (key, row id) (see the header of the MSD-hybrid path).
// Skewed keys: the largest bucket is read back after W5; one that does not fit LDS sends the caller to the
// segmented / sampled / LSD forms.
// =====================================================================================
constexpr int kMsdwThreads = 1024;
constexpr int kMsdwRows = 8;
constexpr int kMsdwTile = kMsdwThreads * kMsdwRows;   // 8192 rows
constexpr int64_t kMsdwUnit = int64_t(1) << 19;        // rows per level-2 histogram work unit
// the append form of the write-combined level 1 (msdw_scatter1wc2_kernel)
constexpr int kMsdwWcK = 4;                  // lines per chunk (one returning global atomic each)
constexpr int kMsdwWc2MinBins = 256;
constexpr int kMsdwWcCursorStride = 32;      // u32 between two buckets' cursors: a 128-byte line each

struct MsdwArgs {
  const uint64_t* src_keys;
  const uint32_t* src_idx;
  const MsdRec* src_rec;   // !raw, src_keys == NULL: level 1 reads {transformed key, row id} records instead of the two arrays (arx_sort_records)
  int raw;
  int64_t n;
  int bits, b1, b2;
  int64_t chunk_rows;      // hist0 chunk
  int kshift;              // leading bits that every key shares: digits are taken from key << kshift
  int sample_shift;        // hist0 reads one chunk in 2^sample_shift (0: every row, exact level-1 sizes)
  int gap2;                // level-2 buckets get a fixed room each (no level-2 histogram)
  int xcd_map;             // level 2: XCD-contiguous tile numbering
  int xcd_map1;            // level 1 likewise (A/B)
  int tile1, tile2;        // rows per scatter tile of level 1 / level 2: kMsdwTile, or 2x / 3x that held in registers
  int rec8;                // records are 8-byte words {the 32 key bits below the level-1 digit, row id} (msdw_word)
  int wc1;                 // rec8, level 1 write-combined (msdw_scatter1wc_kernel): > 0 = its persistent workgroups; buckets
                           // start on 128-byte lines, hold kMsdwPad words, and l1_count[] becomes their exact row counts
  int wc_form;             // 1: round 5's rank-and-stage kernel; 2: round 6's append kernel (msdw_scatter1wc2_kernel): every
                           // store a whole line, places from chunks of kMsdwWcK lines, l1_count[] = the PAD words of a bucket
  uint32_t* wc_cursor;     // form 2: [2^b1 * kMsdwWcCursorStride] next free LINE of every bucket (a 128-byte line per cursor)
  int64_t wc_rows_per_wg;  // form 2: rows of a workgroup's contiguous share
  int wc_k;                // form 2: lines per chunk (1, 2 or 4: the largest whose pads fit the buffers' slack)
  int64_t capacity;        // records rec_x / rec_y can hold
  uint32_t* l1_count;      // [2^b1] histogram (of the sample)
  uint32_t* l1_start;      // [2^b1] first record of a level-1 bucket in rec_x (buckets may be followed by unused room)
  uint32_t* l1_end;        // [2^b1] end of its room (until msdw_scan0b), then end of its rows
  uint32_t* l1_out;        // [2^b1 + 1] first FINAL position of the bucket's rows
  uint32_t* cursor1;       // [2^b1]
  uint32_t* l2_tile_start; // [2^b1 + 1]
  uint32_t* unit_start;    // [2^b1 + 1]
  uint32_t* room2;         // [2^b1] gap2: room of every level-2 bucket of this level-1 bucket
  uint32_t* y_base;        // [2^b1] gap2: first record of the level-1 bucket's level-2 rooms in rec_y
  uint32_t* count2;        // [2^bits] exact form: level-2 histogram;  gap2: part_in (first record of the bucket in rec_y)
  uint32_t* part_start;    // [2^bits + 1] first FINAL position of every level-2 bucket
  uint32_t* cursor2;       // [2^bits]
  uint32_t* flags;         // [0] bits: 2 a bucket does not fit LDS, 4 a level-1 bucket outgrew its room, 8 fixed level-2
                           //     rooms are not possible here, 16 a level-2 bucket outgrew its room, 32 rec8: too many rows
                           //     tied in their 32 record bits; [1] largest bucket; [2] sampled rows; [3] gap2: largest
                           //     level-2 room; [64 + 32 k] rec8: rows that read their full key (32 partial counts)
  MsdRec* rec_x;           // level-1 output
  MsdRec* rec_y;           // level-2 output
};

template <bool RAW>
__global__ __launch_bounds__(kMsdThreads) void msdw_hist0_kernel(MsdwArgs a) {
  __shared__ uint32_t h[kMsdwMaxBins];
  const int tid = threadIdx.x;
  const int nb = 1 << a.b1;
  for (int i = tid; i < nb; i += kMsdThreads) h[i] = 0;
  __syncthreads();
  // sampled: workgroup g reads chunk (g << s) + a pseudo-random offset inside its group of 2^s chunks, so a
  // periodic input cannot line up with the sample
  int64_t chunk = blockIdx.x;
  if (a.sample_shift > 0) {
    const uint32_t mix = (static_cast<uint32_t>(blockIdx.x) * 2654435761u) >> (32 - a.sample_shift);
    chunk = (chunk << a.sample_shift) + mix;
  }
  const int64_t begin = chunk * a.chunk_rows;
  if (begin >= a.n) return;  // workgroup-uniform (last group of a sampled run)
  const int64_t end = begin + a.chunk_rows < a.n ? begin + a.chunk_rows : a.n;
  if (a.sample_shift > 0 && tid == 0) atomicAdd(&a.flags[2], static_cast<uint32_t>(end - begin));
  const int shift = 64 - a.b1;
  constexpr int U = 8;
  int64_t r = begin + tid;
  for (; r + (U - 1) * kMsdThreads < end; r += U * kMsdThreads) {
    uint64_t kk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      kk[u] = RAW ? load_key_typed(a.src_keys, r + u * kMsdThreads, a.raw)
                  : (a.src_keys != nullptr ? a.src_keys[r + u * kMsdThreads] : msd_rec_key(a.src_rec[r + u * kMsdThreads]));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) atomicAdd(&h[(kk[u] << a.kshift) >> shift], 1u);
  }
  for (; r < end; r += kMsdThreads) {
    const uint64_t k = RAW ? load_key_typed(a.src_keys, r, a.raw) : (a.src_keys != nullptr ? a.src_keys[r] : msd_rec_key(a.src_rec[r]));
    atomicAdd(&h[(k << a.kshift) >> shift], 1u);
  }
  __syncthreads();
  for (int i = tid; i < nb; i += kMsdThreads) {
    const uint32_t c = h[i];
    if (c != 0) atomicAdd(&a.l1_count[i], c);
  }
}

// One workgroup of 1024 threads (one level-1 bucket per thread): room for every level-1 bucket in rec_x.  Exact
// histogram: room = rows.  Sampled histogram: rows are estimated (count * n / sampled rows) and get est/32 +
// min(est/8 + 1, 16384) more; if that does not fit the buffer nothing gets room, every tile of level 1 reports the
// overflow and the host repeats the level with the exact histogram.
__global__ __launch_bounds__(1024) void msdw_scan0_kernel(MsdwArgs a) {
  __shared__ uint64_t wt[16];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nb = 1 << a.b1;
  const uint32_t c = tid < nb ? a.l1_count[tid] : 0u;
  uint64_t room = c;
  if (a.sample_shift > 0) {
    const uint32_t sampled = a.flags[2];
    uint64_t est = 0;
    if (sampled != 0 && c != 0) {
      est = static_cast<uint64_t>(static_cast<double>(c) * static_cast<double>(a.n) / static_cast<double>(sampled)) + 1;
      if (est > static_cast<uint64_t>(a.n)) est = static_cast<uint64_t>(a.n);
    }
    const uint64_t extra = est / 8 + 1 < 16384 ? est / 8 + 1 : 16384;
    room = tid < nb ? est + est / 32 + extra : 0;
  }
  if (a.wc1 > 0 && a.wc_form == 2) {   // whole chunks of lines; a workgroup leaves at most one padded line + one chunk's tail per bucket
    const uint64_t chunk = 16u * static_cast<uint64_t>(a.wc_k);
    room = tid < nb ? (room + chunk - 1) / chunk * chunk + chunk * 2 * static_cast<uint64_t>(a.wc1) : 0;
    if (tid < nb) a.l1_count[tid] = 0;   // from here on: PAD words of the bucket (msdw_scatter1wc2_kernel adds them up)
  } else if (a.wc1 > 0) {   // whole lines; every workgroup ends with at most one padded line per bucket
    room = tid < nb ? ((room + 15) & ~uint64_t(15)) + 16u * static_cast<uint64_t>(a.wc1) : 0;
    if (tid < nb) a.l1_count[tid] = 0;   // from here on: rows that arrived (msdw_scatter1wc_kernel adds them up)
  }
  const uint64_t incl = wave_inclusive_scan_u64(room);
  if (lane == 63) wt[wave] = incl;
  __syncthreads();
  uint64_t pre = incl - room, total = 0;
  for (int k = 0; k < 16; ++k) {
    if (k < wave) pre += wt[k];
    total += wt[k];
  }
  const bool fits = total <= static_cast<uint64_t>(a.capacity);
  if (tid < nb) {
    a.l1_start[tid] = fits ? static_cast<uint32_t>(pre) : 0u;
    a.cursor1[tid] = fits ? static_cast<uint32_t>(pre) : 0u;
    a.l1_end[tid] = fits ? static_cast<uint32_t>(pre + room) : 0u;
    if (a.wc1 > 0 && a.wc_form == 2) a.wc_cursor[tid * kMsdwWcCursorStride] = fits ? static_cast<uint32_t>(pre >> 4) : 0u;
  }
}

// After level 1 (one workgroup, one level-1 bucket per thread): rows that arrived in every bucket -> tile / unit maps
// of the level-2 kernels and the bucket's first final position.  gap2: the room of the bucket's level-2 buckets
// (their mean + 6 sqrt(mean) + 64 rows; keys that are uniform inside the level-1 bucket stay below it) and where the
// rooms start in rec_y — flags bit 8 when a room would not fit LDS or all rooms do not fit the buffer.
__global__ __launch_bounds__(1024) void msdw_scan0b_kernel(MsdwArgs a) {
  __shared__ uint32_t wt[3][16];
  __shared__ uint64_t wy[16];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nb = 1 << a.b1;
  uint32_t c = 0;      // rows of the bucket
  uint32_t span = 0;   // records it occupies (write-combined level 1: its rows + pads)
  if (tid < nb) {
    const bool form2 = a.wc1 > 0 && a.wc_form == 2;
    const uint32_t lo = a.l1_start[tid], room_end = a.l1_end[tid];
    const uint64_t cur64 = form2 ? static_cast<uint64_t>(a.wc_cursor[tid * kMsdwWcCursorStride]) << 4 : a.cursor1[tid];
    const uint32_t cur = cur64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : static_cast<uint32_t>(cur64);
    if (cur > room_end || cur < lo) {
      atomicOr(&a.flags[0], 4u);
      span = room_end - lo;
    } else {
      span = cur - lo;
    }
    a.l1_end[tid] = lo + span;
    c = form2 ? span - (a.l1_count[tid] < span ? a.l1_count[tid] : span) : a.wc1 > 0 ? a.l1_count[tid] : span;
  }
  const uint32_t tiles = (span + static_cast<uint32_t>(a.tile2) - 1) / static_cast<uint32_t>(a.tile2);
  const uint32_t units = static_cast<uint32_t>((static_cast<int64_t>(span) + kMsdwUnit - 1) / kMsdwUnit);
  uint32_t room = 0;
  if (a.gap2 && c != 0) {
    const uint32_t mean = (c + (1u << a.b2) - 1) >> a.b2;
    uint32_t root = 0;   // ceil(sqrt(mean)), integer
    for (int b = 15; b >= 0; --b) {
      const uint32_t t = root | (1u << b);
      if (static_cast<uint64_t>(t) * t < mean) root = t;
    }
    root += 1;
    room = mean > static_cast<uint32_t>(kBktCapSmall) ? 0xFFFFFFFFu : mean + 6u * root + 64u;
    if (room > static_cast<uint32_t>(kBktCapSmall)) {
      atomicOr(&a.flags[0], 8u);
      room = 0;
    }
    atomicMax(&a.flags[3], room);   // picks the bucket finish (the smallest workgroup whose LDS holds every room)
  }
  const uint64_t yrows = static_cast<uint64_t>(room) << a.b2;
  const uint32_t i0 = wave_inclusive_scan_u32(c);
  const uint32_t i1 = wave_inclusive_scan_u32(tiles);
  const uint32_t i2 = wave_inclusive_scan_u32(units);
  const uint64_t i3 = wave_inclusive_scan_u64(yrows);
  if (lane == 63) {
    wt[0][wave] = i0;
    wt[1][wave] = i1;
    wt[2][wave] = i2;
    wy[wave] = i3;
  }
  __syncthreads();
  uint32_t p0 = i0 - c, p1 = i1 - tiles, p2 = i2 - units;
  uint64_t p3 = i3 - yrows, ytotal = 0;
  for (int k = 0; k < 16; ++k) {
    if (k < wave) {
      p0 += wt[0][k];
      p1 += wt[1][k];
      p2 += wt[2][k];
      p3 += wy[k];
    }
    ytotal += wy[k];
  }
  const bool yfits = ytotal <= static_cast<uint64_t>(a.capacity);
  if (tid < nb) {
    a.l1_out[tid] = p0;
    a.l2_tile_start[tid] = p1;
    a.unit_start[tid] = p2;
    a.room2[tid] = yfits ? room : 0u;
    a.y_base[tid] = yfits ? static_cast<uint32_t>(p3) : 0u;
  }
  if (tid == nb - 1) {
    a.l1_out[nb] = p0 + c;
    a.l2_tile_start[nb] = p1 + tiles;
    a.unit_start[nb] = p2 + units;
    if (static_cast<int64_t>(p0) + c != a.n) atomicOr(&a.flags[0], 4u);   // rows were dropped by an overflowing tile
    if (a.gap2 && !yfits) atomicOr(&a.flags[0], 8u);
  }
}

struct __attribute__((aligned(16))) MsdwScatterLds {
  uint64_t keys[kMsdwTile];
  uint32_t idx[kMsdwTile];
  uint32_t cnt[kMsdwMaxBins2];
  uint32_t start[kMsdwMaxBins2];
  uint32_t gbase[kMsdwMaxBins2];
  uint32_t wave_tot[kMsdwThreads / 64];
  uint32_t part;
};

// Where the tile's run of every digit starts in the output: ONE returning 64-bit atomic per PAIR of adjacent digits
// (cursor[2j] in the low half, cursor[2j + 1] in the high half; positions are < 2^32 and stay so, so the low half never
// carries).  Global atomics run at 24 G/s chip-wide whatever they touch (profiles/r03_a_atomics_rate_*) and a level of
// 16384-row tiles over 2048 bins issues 2.5e8 of them for 2e9 rows; halving them was worth 0.2 ms of that level's 10
// and 0.3 ms end to end (profiles/r03_s_sort_paired_cursor_atomics_ab.txt) — they mostly overlap the tile's loads.
// cc[k] = rows of digit tid * per + k (0 beyond nb); per = 1, 2 or 4; cursor must be 8-byte aligned.
__device__ __forceinline__ void msdw_reserve_runs(uint32_t* __restrict__ cursor, const uint32_t (&cc)[kMsdwMaxBins2 / kMsdwThreads],
                                                  int per, int nb, uint32_t (&base)[kMsdwMaxBins2 / kMsdwThreads]) {
  constexpr int K = kMsdwMaxBins2 / kMsdwThreads;
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < K; ++k) base[k] = 0;
  if (per == 1) {   // (workgroup-uniform) the pair's digits sit in neighbouring lanes
    const uint32_t c = cc[0];
    const uint32_t cn = __shfl_xor(c, 1, 64);
    unsigned long long old = 0;
    if ((tid & 1) == 0 && tid < nb && (c | cn) != 0) {
      old = atomicAdd(reinterpret_cast<unsigned long long*>(cursor + tid),
                      static_cast<unsigned long long>(c) | (static_cast<unsigned long long>(cn) << 32));
    }
    const uint32_t from_even = __shfl_xor(static_cast<uint32_t>(old >> 32), 1, 64);
    base[0] = (tid & 1) == 0 ? static_cast<uint32_t>(old) : from_even;
  } else {
#pragma unroll
    for (int k = 0; k + 1 < K; k += 2) {
      const int b = tid * per + k;
      if (k < per && b < nb && (cc[k] | cc[k + 1]) != 0) {
        const unsigned long long old =
            atomicAdd(reinterpret_cast<unsigned long long*>(cursor + b),
                      static_cast<unsigned long long>(cc[k]) | (static_cast<unsigned long long>(cc[k + 1]) << 32));
        base[k] = static_cast<uint32_t>(old);
        base[k + 1] = static_cast<uint32_t>(old >> 32);
      }
    }
  }
}

// rec8: a row as ONE 8-byte word — the 32 key bits below the level-1 digit (of key << kshift) above its row id.  Inside a
// level-1 bucket the unsigned order of the words is the order of (those 41 + kshift leading key bits, row id): level 2
// and the finish take their digits from the top of the word, the finish ranks whole words, and only rows whose 32 bits
// tie inside a sub-bucket go back to the column for the bits below (msd_bucket2w_kernel).
constexpr uint64_t kMsdwPad = ~uint64_t(0);   // no row: a row id of 2^32 - 1 is never a row (the forms stop below 2^32 - 4096 rows)
__device__ __forceinline__ uint64_t msdw_word(uint64_t key, uint32_t id, int kshift, int b1) {
  const int sh = kshift + b1;
  const uint64_t below = sh < 64 ? key << sh : 0;
  return (below & 0xFFFFFFFF00000000ull) | id;
}

// Scatter one tile of <= 8192 rows by digit = (key >> dshift) & (nb - 1); run bases from one returning atomic per
// digit on gcursor[]; output = 12-byte records.  1024 threads, nb <= 1024 (one counter per thread in the scan).
// SRC: 0 raw column (row id = position), 1 transformed keys + row ids, 2 records, 3 rec8 words (digits from the word's
// top, no shared-prefix shift: msdw_word already applied it).  OUT8: the output is rec8 words.
// CHECK: a run that does not fit its bucket's room is not written and sets `overflow_bit` in flags[0] — 1: room ends
// at gend[digit]; 2: digit d owns [room_base + d * room, + room).
template <int SRC, int CHECK, bool OUT8 = false>
__device__ __forceinline__ void msdw_scatter_tile(const MsdwArgs& a, MsdwScatterLds& lds, const uint64_t* __restrict__ kin,
                                                  const uint32_t* __restrict__ iin, const MsdRec* __restrict__ rin,
                                                  int64_t row0, int nrows, int nb, int dshift,
                                                  uint32_t* __restrict__ gcursor, const uint32_t* __restrict__ gend,
                                                  uint32_t room_base, uint32_t room, uint32_t overflow_bit,
                                                  MsdRec* __restrict__ rout) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t dmask = static_cast<uint32_t>(nb - 1);
  for (int b = tid; b < nb; b += kMsdwThreads) lds.cnt[b] = 0;
  uint64_t key[kMsdwRows];
  uint32_t idx[kMsdwRows];
#pragma unroll
  for (int i = 0; i < kMsdwRows; ++i) {   // unconditional (clamped) loads: all in flight together
    const int p = i * kMsdwThreads + tid;
    const int64_t r = row0 + (p < nrows ? p : nrows - 1);
    if constexpr (SRC == 0) {
      key[i] = load_key_typed(kin, r, a.raw);
      idx[i] = static_cast<uint32_t>(r);
    } else if constexpr (SRC == 1) {
      key[i] = kin[r];
      idx[i] = iin[r];
    } else if constexpr (SRC == 2) {
      const MsdRec rr = rin[r];
      key[i] = msd_rec_key(rr);
      idx[i] = rr.idx;
    } else {
      key[i] = reinterpret_cast<const uint64_t*>(rin)[r];
      idx[i] = 0;
    }
  }
  __syncthreads();
  const int ksh = SRC == 3 ? 0 : a.kshift;
  uint32_t dig[kMsdwRows], rank[kMsdwRows];
#pragma unroll
  for (int i = 0; i < kMsdwRows; ++i) {
    dig[i] = static_cast<uint32_t>((key[i] << ksh) >> dshift) & dmask;
    rank[i] = 0xFFFFFFFFu;   // no row: past the tile's end, or (words) a pad of the write-combined level 1
    if (i * kMsdwThreads + tid < nrows && (SRC != 3 || key[i] != kMsdwPad)) rank[i] = atomicAdd(&lds.cnt[dig[i]], 1u);
  }
  __syncthreads();
  // exclusive scan of nb <= 4096 counters: `per` consecutive counters per thread (1 up to 1024 bins)
  const int per = (nb + kMsdwThreads - 1) / kMsdwThreads;
  uint32_t cc[kMsdwMaxBins2 / kMsdwThreads];
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < kMsdwMaxBins2 / kMsdwThreads; ++k) {
    const int b = tid * per + k;
    cc[k] = (k < per && b < nb) ? lds.cnt[b] : 0u;
    mine += cc[k];
  }
  const uint32_t incl = wave_inclusive_scan_u32(mine);
  if (lane == 63) lds.wave_tot[wave] = incl;
  __syncthreads();
  uint32_t pre = incl - mine;
  for (int k = 0; k < wave; ++k) pre += lds.wave_tot[k];
  uint32_t run_base[kMsdwMaxBins2 / kMsdwThreads];
  msdw_reserve_runs(gcursor, cc, per, nb, run_base);
#pragma unroll
  for (int k = 0; k < kMsdwMaxBins2 / kMsdwThreads; ++k) {
    const int b = tid * per + k;
    if (k < per && b < nb) {
      const uint32_t c = cc[k];
      lds.start[b] = pre;
      uint32_t base = run_base[k];
      if constexpr (CHECK != 0) {
        const uint32_t end = CHECK == 1 ? gend[b] : room_base + (static_cast<uint32_t>(b) + 1u) * room;
        if (c != 0 && (base + c > end || base + c < base)) {
          base = 0xFFFFFFFFu;
          atomicOr(&a.flags[0], overflow_bit);
        }
      }
      lds.gbase[b] = base;
      pre += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kMsdwRows; ++i) {
    if (rank[i] != 0xFFFFFFFFu) {
      const uint32_t pos = lds.start[dig[i]] + rank[i];
      lds.keys[pos] = key[i];
      if constexpr (SRC != 3) lds.idx[pos] = idx[i];
    }
  }
  __syncthreads();
  const int nplaced = SRC == 3 ? static_cast<int>(lds.start[nb - 1] + lds.cnt[nb - 1]) : nrows;   // (pads take no place)
  for (int p = tid; p < nplaced; p += kMsdwThreads) {
    const uint64_t k = lds.keys[p];
    const uint32_t d = static_cast<uint32_t>((k << ksh) >> dshift) & dmask;
    const uint32_t gb = lds.gbase[d];
    if (CHECK != 0 && gb == 0xFFFFFFFFu) continue;
    const uint32_t at = gb + (static_cast<uint32_t>(p) - lds.start[d]);
    if constexpr (OUT8) {
      reinterpret_cast<uint64_t*>(rout)[at] = SRC == 3 ? k : msdw_word(k, lds.idx[p], a.kshift, a.b1);
    } else {
      MsdRec rr;
      rr.lo = static_cast<uint32_t>(k);
      rr.hi = static_cast<uint32_t>(k >> 32);
      rr.idx = lds.idx[p];
      rout[at] = rr;
    }
  }
}

// The same scatter over a tile of RPT * 1024 rows (RPT = 16 or 24) held in REGISTERS: ranked with LDS atomics, then
// moved through the 8192-record LDS buffer in RPT / 8 rounds.  A (tile, digit) run is RPT / 8 times as long as the
// LDS-resident tile's — 48 records = 576 B at 512 bins and RPT 24 — which is what sets the rate of a scatter whose bins
// span the whole array (scripts/micro/wide_scatter_bench.hip, profiles/r03_c_wide_scatter_small_lds_chunks.txt; the
// group-by's flat level is the same kernel shape, groupby.hip K3w).
template <int SRC, int CHECK, int RPT, bool OUT8 = false>
__device__ __forceinline__ void msdw_scatter_big_tile(const MsdwArgs& a, MsdwScatterLds& lds, const uint64_t* __restrict__ kin,
                                                      const uint32_t* __restrict__ iin, const MsdRec* __restrict__ rin,
                                                      int64_t row0, int nrows, int nb, int dshift,
                                                      uint32_t* __restrict__ gcursor, const uint32_t* __restrict__ gend,
                                                      uint32_t room_base, uint32_t room, uint32_t overflow_bit,
                                                      MsdRec* __restrict__ rout) {
  static_assert(RPT % kMsdwRows == 0, "whole rounds of the LDS buffer");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t dmask = static_cast<uint32_t>(nb - 1);
  for (int b = tid; b < nb; b += kMsdwThreads) lds.cnt[b] = 0;
  uint64_t key[RPT];
  uint32_t idx[(SRC == 0 || SRC == 3) ? 1 : RPT];
  const int ksh = SRC == 3 ? 0 : a.kshift;
  // unconditional loads (rows past the tile's end re-read its last row), all in flight together; addresses = one
  // uniform base + a 32-bit offset per row.  The caller's column is read once: non-temporal.
  if constexpr (SRC == 0) {
    if (key_type_is_64bit(a.raw)) {   // workgroup-uniform
      const uint64_t* __restrict__ base = kin + row0;
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const uint32_t p = static_cast<uint32_t>(i * kMsdwThreads + tid);
        key[i] = __builtin_nontemporal_load(base + (p < static_cast<uint32_t>(nrows) ? p : static_cast<uint32_t>(nrows - 1)));
      }
    } else {
      const uint32_t* __restrict__ base = reinterpret_cast<const uint32_t*>(kin) + row0;
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const uint32_t p = static_cast<uint32_t>(i * kMsdwThreads + tid);
        key[i] = __builtin_nontemporal_load(base + (p < static_cast<uint32_t>(nrows) ? p : static_cast<uint32_t>(nrows - 1)));
      }
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) key[i] = key_from_bits(key[i], a.raw);
  } else {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const uint32_t p0 = static_cast<uint32_t>(i * kMsdwThreads + tid);
      const uint32_t p = p0 < static_cast<uint32_t>(nrows) ? p0 : static_cast<uint32_t>(nrows - 1);
      if constexpr (SRC == 1) {
        key[i] = (kin + row0)[p];
        idx[i] = (iin + row0)[p];
      } else if constexpr (SRC == 2) {
        const MsdRec rr = (rin + row0)[p];
        key[i] = msd_rec_key(rr);
        idx[i] = rr.idx;
      } else {
        key[i] = (reinterpret_cast<const uint64_t*>(rin) + row0)[p];
      }
    }
  }
  __syncthreads();
  // a row's place in the tile (< 24576) takes 16 bits: two per register, 0xFFFF = no row (past the tile's end)
  static_assert(RPT % 2 == 0 && RPT * kMsdwThreads < 0xFFFF - kMsdwTile, "packed 16-bit tile positions");
  uint32_t pos2[RPT / 2];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const uint32_t d = static_cast<uint32_t>((key[i] << ksh) >> dshift) & dmask;
    const uint32_t pr = (i * kMsdwThreads + tid < nrows && (SRC != 3 || key[i] != kMsdwPad)) ? atomicAdd(&lds.cnt[d], 1u) : 0xFFFFu;
    pos2[i / 2] = (i & 1) ? (pos2[i / 2] | (pr << 16)) : pr;
    if (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);   // eight atomics in flight, not RPT (their addresses and results are registers)
  }
  __syncthreads();
  const int per = (nb + kMsdwThreads - 1) / kMsdwThreads;
  uint32_t cc[kMsdwMaxBins2 / kMsdwThreads];
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < kMsdwMaxBins2 / kMsdwThreads; ++k) {
    const int b = tid * per + k;
    cc[k] = (k < per && b < nb) ? lds.cnt[b] : 0u;
    mine += cc[k];
  }
  const uint32_t incl = wave_inclusive_scan_u32(mine);
  if (lane == 63) lds.wave_tot[wave] = incl;
  __syncthreads();
  uint32_t pre = incl - mine;
  for (int k = 0; k < wave; ++k) pre += lds.wave_tot[k];
  uint32_t run_base[kMsdwMaxBins2 / kMsdwThreads];
  msdw_reserve_runs(gcursor, cc, per, nb, run_base);
#pragma unroll
  for (int k = 0; k < kMsdwMaxBins2 / kMsdwThreads; ++k) {
    const int b = tid * per + k;
    if (k < per && b < nb) {
      const uint32_t c = cc[k];
      lds.start[b] = pre;
      uint32_t base = run_base[k];
      if constexpr (CHECK != 0) {
        const uint32_t end = CHECK == 1 ? gend[b] : room_base + (static_cast<uint32_t>(b) + 1u) * room;
        if (c != 0 && (base + c > end || base + c < base)) {
          base = 0xFFFFFFFFu;
          atomicOr(&a.flags[0], overflow_bit);
        }
      }
      lds.gbase[b] = base;
      pre += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const uint32_t d = static_cast<uint32_t>((key[i] << ksh) >> dshift) & dmask;
    const uint32_t pr = (pos2[i / 2] >> ((i & 1) * 16)) & 0xFFFFu;
    const uint32_t at = pr != 0xFFFFu ? pr + lds.start[d] : 0xFFFFu;
    pos2[i / 2] = (i & 1) ? ((pos2[i / 2] & 0xFFFFu) | (at << 16)) : ((pos2[i / 2] & 0xFFFF0000u) | at);
    if (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);
  }
  const int nplaced = SRC == 3 ? static_cast<int>(lds.start[nb - 1] + lds.cnt[nb - 1]) : nrows;   // (pads take no place)
  for (int r = 0; r < RPT / kMsdwRows; ++r) {
    const uint32_t lo = static_cast<uint32_t>(r) * kMsdwTile;
    if (static_cast<int>(lo) >= nplaced) break;   // workgroup-uniform
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const uint32_t q = ((pos2[i / 2] >> ((i & 1) * 16)) & 0xFFFFu) - lo;   // (no row: 0xFFFF - lo is never inside the buffer)
      if (q < static_cast<uint32_t>(kMsdwTile)) {
        lds.keys[q] = key[i];
        if constexpr (SRC == 0) {
          lds.idx[q] = static_cast<uint32_t>(row0) + static_cast<uint32_t>(i * kMsdwThreads + tid);
        } else if constexpr (SRC != 3) {
          lds.idx[q] = idx[i];
        }
      }
    }
    __syncthreads();
    const int cnt = nplaced - static_cast<int>(lo) < kMsdwTile ? nplaced - static_cast<int>(lo) : kMsdwTile;
    for (int p = tid; p < cnt; p += kMsdwThreads) {
      const uint64_t k = lds.keys[p];
      const uint32_t d = static_cast<uint32_t>((k << ksh) >> dshift) & dmask;
      const uint32_t gb = lds.gbase[d];
      if (CHECK != 0 && gb == 0xFFFFFFFFu) continue;
      const uint32_t at = gb + (lo + static_cast<uint32_t>(p) - lds.start[d]);
      if constexpr (OUT8) {
        reinterpret_cast<uint64_t*>(rout)[at] = SRC == 3 ? k : msdw_word(k, lds.idx[p], a.kshift, a.b1);
      } else {
        MsdRec rr;
        rr.lo = static_cast<uint32_t>(k);
        rr.hi = static_cast<uint32_t>(k >> 32);
        rr.idx = lds.idx[p];
        rout[at] = rr;
      }
    }
    __syncthreads();
  }
}

template <int SRC, int CHECK, int RPT, bool OUT8 = false>
__device__ __forceinline__ void msdw_scatter_any_tile(const MsdwArgs& a, MsdwScatterLds& lds, const uint64_t* __restrict__ kin,
                                                      const uint32_t* __restrict__ iin, const MsdRec* __restrict__ rin,
                                                      int64_t row0, int nrows, int nb, int dshift,
                                                      uint32_t* __restrict__ gcursor, const uint32_t* __restrict__ gend,
                                                      uint32_t room_base, uint32_t room, uint32_t overflow_bit,
                                                      MsdRec* __restrict__ rout) {
  if constexpr (RPT == kMsdwRows) {
    msdw_scatter_tile<SRC, CHECK, OUT8>(a, lds, kin, iin, rin, row0, nrows, nb, dshift, gcursor, gend, room_base, room,
                                        overflow_bit, rout);
  } else {
    msdw_scatter_big_tile<SRC, CHECK, RPT, OUT8>(a, lds, kin, iin, rin, row0, nrows, nb, dshift, gcursor, gend, room_base,
                                                 room, overflow_bit, rout);
  }
}

template <bool RAW, int RPT, bool OUT8 = false>
__global__ __launch_bounds__(kMsdwThreads) void msdw_scatter1_kernel(MsdwArgs a) {
  __shared__ MsdwScatterLds lds;
  // an earlier tile already found a bucket without room: the level will be repeated, do not finish this attempt
  // (e.g. pre-sorted input, where a sample of tiles says little about where the bucket boundaries fall).  One thread
  // looks for the whole workgroup: the bit may appear between two threads' loads, and half a workgroup scatters garbage
  if (a.sample_shift > 0) {
    if (threadIdx.x == 0) lds.part = __atomic_load_n(&a.flags[0], __ATOMIC_RELAXED) & 4u;
    __syncthreads();
    if (lds.part != 0) return;   // workgroup-uniform
    __syncthreads();
  }
  constexpr int kTile = RPT * kMsdwThreads;
  const uint32_t tile = a.xcd_map1 ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int64_t row0 = static_cast<int64_t>(tile) * kTile;
  const int nrows = static_cast<int>(a.n - row0 < kTile ? a.n - row0 : kTile);
  msdw_scatter_any_tile<RAW ? 0 : 1, 1, RPT, OUT8>(a, lds, a.src_keys, a.src_idx, nullptr, row0, nrows, 1 << a.b1, 64 - a.b1,
                                                   a.cursor1, a.l1_end, 0u, 0u, 4u, a.rec_x);
}

// Level 1 reading {transformed key, row id} RECORDS (a.src_rec): what the receiver of the sharded sort holds (arx_sort_records) —
// the same tile scatter with the source form level 2 uses; no split of the records into a key and a row array first
// (2.5 ms of a rank's 13.2 at 5e8 records, profiles/r06_t_*).
template <int RPT>
__global__ __launch_bounds__(kMsdwThreads) void msdw_scatter1r_kernel(MsdwArgs a) {
  __shared__ MsdwScatterLds lds;
  if (a.sample_shift > 0) {   // (as msdw_scatter1_kernel: an earlier tile found a bucket without room)
    if (threadIdx.x == 0) lds.part = __atomic_load_n(&a.flags[0], __ATOMIC_RELAXED) & 4u;
    __syncthreads();
    if (lds.part != 0) return;   // workgroup-uniform
    __syncthreads();
  }
  constexpr int kTile = RPT * kMsdwThreads;
  const uint32_t tile = a.xcd_map1 ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int64_t row0 = static_cast<int64_t>(tile) * kTile;
  const int nrows = static_cast<int>(a.n - row0 < kTile ? a.n - row0 : kTile);
  msdw_scatter_any_tile<2, 1, RPT, false>(a, lds, nullptr, nullptr, a.src_rec, row0, nrows, 1 << a.b1, 64 - a.b1, a.cursor1, a.l1_end,
                                          0u, 0u, 4u, a.rec_x);
}

// Level 1 of the rec8 form, WRITE-COMBINED (VERDICT r4 "Next round" 3).  What bounds a scatter whose bins span the whole
// array is the partial 128-byte lines at both ends of every (tile, bin) run, not its bytes: the same level ran 12.6 ms
// writing 8-byte words and 11.9 ms writing 12-byte records (profiles/r05_a).  Here a PERSISTENT workgroup keeps one
// line (16 words) per bin in LDS: a tile's run of bin d follows what the bin has left over from the workgroup's earlier
// tiles, whole lines go out at line-aligned positions (buckets start on lines, the cursors move in lines), the rest
// stays for the next tile.  After its last tile a workgroup writes each left-over as ONE line filled up with kMsdwPad
// words, which level 2 and the exact level-2 histogram skip; the buckets' exact row counts are added up in l1_count[].
// nb <= kMsdwWcBins (one thread per bin; the lines are 64 KB of LDS).
constexpr int kMsdwWcBins = 512;
struct __attribute__((aligned(16))) MsdwWcBin {
  uint32_t start;     // the bin's first place in the sorted tile
  uint32_t oldleft;   // words it had left over before this tile
  uint32_t full;      // words of (left-over ++ run) that leave as whole lines
  uint32_t gbase;     // where those go (0xFFFFFFFF: no room — dropped, flags bit 4)
};
struct __attribute__((aligned(16))) MsdwWcLds {
  uint64_t words[kMsdwTile];          // one round of the tile, sorted by bin
  uint64_t line[kMsdwWcBins][16];     // what every bin has left over
  MsdwWcBin info[kMsdwWcBins];        // (one 16-byte read per emitted word)
  uint16_t bin[kMsdwTile];            // bin of words[q]
  uint32_t cnt[kMsdwWcBins];
  uint32_t left[kMsdwWcBins];         // words left over after this tile
  uint32_t wave_tot[kMsdwThreads / 64];
  uint32_t stop;
};

// keys of one tile -> registers (unconditional loads, rows past the tile's end re-read its last row; the caller's column
// is read once: non-temporal)
template <int RPT>
__device__ __forceinline__ void msdw_wc_load(const MsdwArgs& a, int64_t row0, int nrows, uint64_t (&key)[RPT]) {
  const int tid = threadIdx.x;
  if (key_type_is_64bit(a.raw)) {   // workgroup-uniform
    const uint64_t* __restrict__ base = a.src_keys + row0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const uint32_t p = static_cast<uint32_t>(i * kMsdwThreads + tid);
      key[i] = __builtin_nontemporal_load(base + (p < static_cast<uint32_t>(nrows) ? p : static_cast<uint32_t>(nrows - 1)));
    }
  } else {
    const uint32_t* __restrict__ base = reinterpret_cast<const uint32_t*>(a.src_keys) + row0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const uint32_t p = static_cast<uint32_t>(i * kMsdwThreads + tid);
      key[i] = __builtin_nontemporal_load(base + (p < static_cast<uint32_t>(nrows) ? p : static_cast<uint32_t>(nrows - 1)));
    }
  }
}

// PREFETCH: the next tile's keys are requested before this tile's words move through LDS and out (one workgroup of
// 1024 threads is all a CU holds of this kernel — 156 KB of LDS — so nothing else would overlap its phases; workgroup
// barriers do not wait for global loads on gfx950, the loads stay in flight across them)
template <int RPT, bool PREFETCH>
__global__ __launch_bounds__(kMsdwThreads) void msdw_scatter1wc_kernel(MsdwArgs a) {
  static_assert(RPT % kMsdwRows == 0 && RPT % 2 == 0 && RPT * kMsdwThreads < 0xFFFF - kMsdwTile, "rounds; packed 16-bit places");
  __shared__ MsdwWcLds lds;
  constexpr int kTile = RPT * kMsdwThreads;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nb = 1 << a.b1;
  const int dshift = 64 - a.b1;
  const uint32_t dmask = static_cast<uint32_t>(nb - 1);
  uint64_t* __restrict__ out = reinterpret_cast<uint64_t*>(a.rec_x);
  if (tid < nb) lds.left[tid] = 0;
  uint32_t my_rows = 0;   // rows of bin `tid` this workgroup has seen
  const int64_t ntiles = (a.n + kTile - 1) / kTile;
  bool abandoned = false;
  uint64_t key[RPT];
  uint64_t nxt[PREFETCH ? RPT : 1];
  if (static_cast<int64_t>(blockIdx.x) >= ntiles) return;   // (more workgroups than tiles: nothing left over, no rows seen)
  if constexpr (PREFETCH) {
    const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kTile;
    msdw_wc_load<RPT>(a, row0, static_cast<int>(a.n - row0 < kTile ? a.n - row0 : kTile), key);
  }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // an earlier tile already found a bucket without room: the level will be repeated, do not finish this attempt
    if (a.sample_shift > 0) {
      if (tid == 0) lds.stop = __atomic_load_n(&a.flags[0], __ATOMIC_RELAXED) & 4u;
      __syncthreads();
      abandoned = lds.stop != 0;   // workgroup-uniform
      if (abandoned) break;
    }
    const int64_t row0 = tile * kTile;
    const int nrows = static_cast<int>(a.n - row0 < kTile ? a.n - row0 : kTile);
    if (tid < nb) lds.cnt[tid] = 0;
    if constexpr (!PREFETCH) msdw_wc_load<RPT>(a, row0, nrows, key);
#pragma unroll
    for (int i = 0; i < RPT; ++i) key[i] = key_from_bits(key[i], a.raw);
    __syncthreads();
    uint32_t pos2[RPT / 2];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const uint32_t d = static_cast<uint32_t>((key[i] << a.kshift) >> dshift) & dmask;
      const uint32_t pr = (i * kMsdwThreads + tid < nrows) ? atomicAdd(&lds.cnt[d], 1u) : 0xFFFFu;
      pos2[i / 2] = (i & 1) ? (pos2[i / 2] | (pr << 16)) : pr;
      if (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    // one thread per bin: its place in the sorted tile, what leaves as whole lines, where those go
    const uint32_t c = tid < nb ? lds.cnt[tid] : 0u;
    const uint32_t incl = wave_inclusive_scan_u32(c);
    if (lane == 63) lds.wave_tot[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - c;
    for (int k = 0; k < wave; ++k) pre += lds.wave_tot[k];
    const uint32_t old = tid < nb ? lds.left[tid] : 0u;
    const uint32_t tot = old + c;
    uint32_t cc[kMsdwMaxBins2 / kMsdwThreads] = {};
    uint32_t run_base[kMsdwMaxBins2 / kMsdwThreads];
    cc[0] = tot & ~15u;
    msdw_reserve_runs(a.cursor1, cc, 1, nb, run_base);
    if (tid < nb) {
      uint32_t base = run_base[0];
      if (cc[0] != 0 && (base + cc[0] > a.l1_end[tid] || base + cc[0] < base)) {
        base = 0xFFFFFFFFu;
        atomicOr(&a.flags[0], 4u);
      }
      MsdwWcBin b;
      b.start = pre;
      b.oldleft = old;
      b.full = cc[0];
      b.gbase = base;
      lds.info[tid] = b;
      lds.left[tid] = tot & 15u;
      my_rows += c;
    }
    if constexpr (PREFETCH) {
      // (unconditional: past the workgroup's last tile it requests this tile again — `nxt` is never read uninitialised,
      //  which the compiler may otherwise take as licence to run the loads of a tile that does not exist)
      const int64_t tn = tile + gridDim.x < ntiles ? tile + gridDim.x : tile;
      const int64_t rn = tn * kTile;
      msdw_wc_load<RPT>(a, rn, static_cast<int>(a.n - rn < kTile ? a.n - rn : kTile), nxt);
    }
    __syncthreads();
    // the left-overs of bins that now fill lines leave first (their slots take this tile's left-overs below)
    for (int slot = tid; slot < nb * 16; slot += kMsdwThreads) {
      const int d = slot >> 4, j = slot & 15;
      const MsdwWcBin b = lds.info[d];
      if (b.full != 0 && static_cast<uint32_t>(j) < b.oldleft && b.gbase != 0xFFFFFFFFu) out[b.gbase + static_cast<uint32_t>(j)] = lds.line[d][j];
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const uint32_t d = static_cast<uint32_t>((key[i] << a.kshift) >> dshift) & dmask;
      const uint32_t pr = (pos2[i / 2] >> ((i & 1) * 16)) & 0xFFFFu;
      const uint32_t at = pr != 0xFFFFu ? pr + lds.info[d].start : 0xFFFFu;
      pos2[i / 2] = (i & 1) ? ((pos2[i / 2] & 0xFFFFu) | (at << 16)) : ((pos2[i / 2] & 0xFFFF0000u) | at);
      if (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    for (int r = 0; r < RPT / kMsdwRows; ++r) {
      const uint32_t lo = static_cast<uint32_t>(r) * kMsdwTile;
      if (static_cast<int>(lo) >= nrows) break;   // workgroup-uniform
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const uint32_t q = ((pos2[i / 2] >> ((i & 1) * 16)) & 0xFFFFu) - lo;   // (no row: 0xFFFF - lo is never inside the buffer)
        if (q < static_cast<uint32_t>(kMsdwTile)) {
          lds.words[q] = msdw_word(key[i], static_cast<uint32_t>(row0) + static_cast<uint32_t>(i * kMsdwThreads + tid), a.kshift, a.b1);
          lds.bin[q] = static_cast<uint16_t>(static_cast<uint32_t>((key[i] << a.kshift) >> dshift) & dmask);
        }
      }
      __syncthreads();
      const int cnt = nrows - static_cast<int>(lo) < kMsdwTile ? nrows - static_cast<int>(lo) : kMsdwTile;
      for (int p = tid; p < cnt; p += kMsdwThreads) {
        const uint32_t d = lds.bin[p];
        const MsdwWcBin b = lds.info[d];
        const uint32_t sp = b.oldleft + (lo + static_cast<uint32_t>(p) - b.start);   // place in (left-over ++ run)
        if (sp < b.full) {
          if (b.gbase != 0xFFFFFFFFu) out[b.gbase + sp] = lds.words[p];
        } else {
          lds.line[d][sp - b.full] = lds.words[p];
        }
      }
      __syncthreads();
    }
    if constexpr (PREFETCH) {
#pragma unroll
      for (int i = 0; i < RPT; ++i) key[i] = nxt[i];
    }
  }
  if (abandoned) return;
  // what is still left over: one line per bin, filled up with pads
  {
    __syncthreads();
    const uint32_t old = tid < nb ? lds.left[tid] : 0u;
    uint32_t cc[kMsdwMaxBins2 / kMsdwThreads] = {};
    uint32_t run_base[kMsdwMaxBins2 / kMsdwThreads];
    cc[0] = old != 0 ? 16u : 0u;
    msdw_reserve_runs(a.cursor1, cc, 1, nb, run_base);
    if (tid < nb) {
      uint32_t base = run_base[0];
      if (cc[0] != 0 && (base + 16u > a.l1_end[tid] || base + 16u < base)) {
        base = 0xFFFFFFFFu;
        atomicOr(&a.flags[0], 4u);
      }
      lds.info[tid].gbase = base;
      if (my_rows != 0) atomicAdd(&a.l1_count[tid], my_rows);
    }
    __syncthreads();
    for (int slot = tid; slot < nb * 16; slot += kMsdwThreads) {
      const int d = slot >> 4, j = slot & 15;
      const uint32_t l = lds.left[d];
      const uint32_t gb = lds.info[d].gbase;
      if (l != 0 && gb != 0xFFFFFFFFu) out[gb + static_cast<uint32_t>(j)] = static_cast<uint32_t>(j) < l ? lds.line[d][j] : kMsdwPad;
    }
  }
}

// Level 1 of the rec8 form, write-combined by APPENDING (round 6; the form the group-by's lines plan found,
// csrc/groupby_lines.h, scripts/micro/wc_lines_bench.hip).  Round 5's kernel above ranks a tile, stages it through LDS and
// still splits the line a bin's left-over shares with the tile's first words into two partial stores (19.06 GB written
// for 16 GB of words, 3.3 TB/s: profiles/sort_traffic.json); with whole lines only the memory side alone takes 12 B read +
// lines written to 814 frontiers at 5.1 TB/s (profiles/r06_c_*).  Here a row takes its slot in its bin's ONE line with a
// returning LDS atomic on the bin's fill counter (all of a batch's atomics issued before the first dependent store), the
// row that takes slot 15 queues the bin, and after a barrier 8 lanes copy each queued line out, 16 bytes each: every
// global store is a whole 128-byte line.  A line's place comes from the workgroup's chunk of kMsdwWcK lines of the
// bucket — one returning global atomic per chunk on a cursor that has a 128-byte line to itself, issued when the chunk's
// last line leaves and picked up before the next flush.  A row that finds its line full is carried into the next batch
// (two per thread); a thread with more falls back to append / flush rounds.  At its end a workgroup fills its partial
// lines and the rest of its chunks with kMsdwPad words and adds their number to l1_count[] (rows = span - pads).
// A bucket that outgrows its room writes into its neighbour's — msdw_scan0b_kernel sees the cursor past the room's end
// (flags bit 4) and the level is repeated with exact counts; only the end of the buffer is never passed.
constexpr uint32_t kMsdwWcSkip = 0xFFFFFFFFu;
constexpr uint32_t kMsdwWcNever = 0xFFFFFFFEu;

// KT: the key type when the host knows it at compile time (0 .. 5 of the C ABI's ARX_KEY_*; the transform of a key is then
// straight-line code and the loads have one width), -1: a.raw at run time.  Round 6, after profiles/r06_j_*: the kernel's
// waves issue for a third of their time and the four of a SIMD share one VALU and one scalar unit (52 % / 44 % busy), so the
// instructions of a batch are part of its duration — the per-element switch over the key type, the 64-bit row arithmetic
// and the end-of-span checks in EVERY batch were a sixth of them (only a workgroup's last batch can be short).
template <int R, int KT, bool LISTQ = false>
__global__ __launch_bounds__(kMsdwThreads) void msdw_scatter1wc2_kernel(MsdwArgs a) {
  __shared__ uint64_t vals[kMsdwMaxBins * 16];
  // state = next line (absolute) << 1 | the workgroup holds that line (rooms and chunks start at multiples of kMsdwWcK lines)
  __shared__ uint32_t fill[kMsdwMaxBins], state[kMsdwMaxBins];
  __shared__ uint16_t wlist[kMsdwMaxBins];   // per wave: which of its 64 bins have a full line (flush_phase)
  __shared__ uint16_t list[LISTQ ? 2 : 1][LISTQ ? kMsdwMaxBins : 1];   // LISTQ (A/B only): the first form's queue of full lines
  __shared__ uint32_t nlist[2];
  __shared__ uint32_t again[2];
  const uint32_t K = static_cast<uint32_t>(a.wc_k);   // 1, 2 or 4
  const int tid = threadIdx.x, lane = tid & 63;
  const int nb = 1 << a.b1;
  const int dshift = 64 - a.b1;
  const uint32_t dmask = static_cast<uint32_t>(nb - 1);
  const uint32_t total_lines = static_cast<uint32_t>(a.capacity >> 4);
  uint64_t* __restrict__ out = reinterpret_cast<uint64_t*>(a.rec_x);
  if (tid < nb) {
    fill[tid] = 0;
    state[tid] = kMsdwWcNever;
  }
  if (tid < 2) {
    again[tid] = 0;
    nlist[tid] = 0;
  }
  __syncthreads();
  const int64_t lo = static_cast<int64_t>(blockIdx.x) * a.wc_rows_per_wg;
  const int64_t hi = lo + a.wc_rows_per_wg < a.n ? lo + a.wc_rows_per_wg : a.n;
  if (lo >= hi) return;
  const int xf = KT < 0 ? a.raw : ((a.raw & ~(7 << 4)) | (KT << 4));
  const bool wide_keys = KT < 0 ? key_type_is_64bit(a.raw) : (KT == 0 || KT == 1 || KT == 4);   // workgroup-uniform
  const uint32_t span = static_cast<uint32_t>(hi - lo);   // (a workgroup's rows: far below 2^32)
  const uint64_t* __restrict__ src64 = a.src_keys + lo;
  const uint32_t* __restrict__ src32 = reinterpret_cast<const uint32_t*>(a.src_keys) + lo;
  constexpr uint32_t kBatch = static_cast<uint32_t>(R) * kMsdwThreads;
  uint64_t kc[R], kn[R];
  // the keys of the batch at `rel` (rows from the workgroup's first); FULL: the whole batch lies inside the span
  auto issue = [&](uint32_t rel, auto full) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      uint32_t r = rel + static_cast<uint32_t>(i * kMsdwThreads + tid);
      if constexpr (!decltype(full)::value) r = r < span ? r : span - 1;
      kn[i] = wide_keys ? __builtin_nontemporal_load(src64 + r) : static_cast<uint64_t>(__builtin_nontemporal_load(src32 + r));
    }
  };
  auto issue_at = [&](uint32_t rel) {
    if (rel + kBatch <= span) issue(rel, std::true_type{});
    else issue(rel, std::false_type{});
  };
  int cur = 0;
  // (the row that takes slot 15 used to queue its bin in a list — a wave-aggregated returning LDS atomic and a wait for it
  //  behind every row of a batch, since some lane of 64 nearly always takes a slot 15; the flush finds the full lines itself now)
  auto place = [&](uint32_t bin, uint32_t slot, uint64_t word) {
    vals[bin * 16 + slot] = word;
    if constexpr (LISTQ) {
      if (slot == 15) list[cur][atomicAdd(&nlist[cur], 1u)] = static_cast<uint16_t>(bin);
    }
  };
  auto next_line = [&](uint32_t bin) -> uint32_t {
    uint32_t s = state[bin];
    if ((s & 1u) == 0) s = (atomicAdd(&a.wc_cursor[bin * kMsdwWcCursorStride], K) << 1) | 1u;
    const uint32_t line = s >> 1;
    state[bin] = ((line + 1) << 1) | (((line + 1) & (K - 1)) != 0 ? 1u : 0u);
    return line;
  };
  auto store_piece = [&](uint32_t line, int sub, arx_u32x4 d) {
    if (line >= total_lines) {
      atomicOr(&a.flags[0], 4u);
      return;
    }
    *reinterpret_cast<arx_u32x4*>(out + static_cast<size_t>(line) * 16 + sub * 2) = d;
  };
  bool rf = false;   // the chunk this thread's bin (tid) waits for
  uint32_t rv = 0;
  auto flush_phase = [&]() -> uint32_t {
    if (rf) {
      state[tid] = (rv << 1) | 1u;
      rf = false;
    }
    __syncthreads();
    const uint32_t go = again[cur];
    if (tid == 0) again[cur ^ 1] = 0;
    // a wave flushes the full lines of ITS 64 bins (bin = thread): which ones, compacted through the wave's piece of wlist
    // (written and read by this wave only — its LDS operations execute in order), then 8 lanes a line, 16 bytes each
    const int sub = tid & 7;
    if constexpr (LISTQ) {
      const uint32_t nfl = nlist[cur];
      if (tid == 0) nlist[cur ^ 1] = 0;
      for (uint32_t g0 = 0; g0 < nfl; g0 += kMsdwThreads / 8) {   // (workgroup-uniform trip count: the shuffle below)
        const uint32_t g = g0 + (tid >> 3);
        const bool on = g < nfl;
        const uint32_t bin = on ? list[cur][g] : 0u;
        uint32_t line = 0;
        if (on && sub == 0) line = next_line(bin);
        line = __shfl(line, lane & ~7, 64);
        if (on) {
          store_piece(line, sub, *reinterpret_cast<const arx_u32x4*>(&vals[bin * 16 + sub * 2]));
          if (sub == 0) fill[bin] = 0;
        }
      }
    } else {
      const bool full = tid < nb && fill[tid] >= 16u;
      const uint64_t fmask = __ballot(full);
      const uint32_t nf = static_cast<uint32_t>(__popcll(fmask));   // (wave-uniform)
      if (full) wlist[(tid & ~63) + __popcll(fmask & ((uint64_t(1) << lane) - 1))] = static_cast<uint16_t>(tid);
      __builtin_amdgcn_wave_barrier();
      for (uint32_t g0 = 0; g0 < nf; g0 += 8) {
        const uint32_t g = g0 + static_cast<uint32_t>(lane >> 3);
        const bool on = g < nf;
        const uint32_t bin = on ? wlist[(tid & ~63) + g] : 0u;
        uint32_t line = 0;
        if (on && sub == 0) line = next_line(bin);
        line = __shfl(line, lane & ~7, 64);
        if (on) {
          store_piece(line, sub, *reinterpret_cast<const arx_u32x4*>(&vals[bin * 16 + sub * 2]));
          if (sub == 0) fill[bin] = 0;
        }
      }
    }
    __syncthreads();
    cur ^= 1;
    if (tid < nb) {
      const uint32_t s = state[tid];
      if ((s & 1u) == 0 && s != kMsdwWcNever) {
        rv = atomicAdd(&a.wc_cursor[tid * kMsdwWcCursorStride], K);
        rf = true;
      }
    }
    return go;
  };
  uint64_t pw0 = 0, pw1 = 0;   // carried rows: their words and bins
  uint32_t pb0 = 0, pb1 = 0, np = 0;
  bool gave_up = false;
  const uint32_t row0 = static_cast<uint32_t>(lo);   // (row ids are 32 bits in this form)
  // One batch of R rows per thread; FULL: every row of it lies inside the span.  Returns false when the workgroup gives up.
  auto batch = [&](uint32_t rel, auto full) -> bool {
#pragma unroll
    for (int i = 0; i < R; ++i) kc[i] = kn[i];
    if (rel + kBatch < span) issue_at(rel + kBatch);
    uint32_t bn[R], sl[R];
    uint64_t wd[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const uint64_t key = key_from_bits(kc[i], xf);
      bn[i] = static_cast<uint32_t>((key << a.kshift) >> dshift) & dmask;
      wd[i] = msdw_word(key, row0 + rel + static_cast<uint32_t>(i * kMsdwThreads + tid), a.kshift, a.b1);
    }
    // (a branch-free form of this phase — dummy bins for what does not apply, one branch for the queueing — measured no
    //  faster, 24.8 against 24.3 ms end to end, profiles/r06_j_*)
    uint32_t cs0 = kMsdwWcSkip, cs1 = kMsdwWcSkip;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if constexpr (decltype(full)::value) sl[i] = atomicAdd(&fill[bn[i]], 1u);
      else sl[i] = rel + static_cast<uint32_t>(i * kMsdwThreads + tid) < span ? atomicAdd(&fill[bn[i]], 1u) : kMsdwWcSkip;
    }
    if (np > 0) cs0 = atomicAdd(&fill[pb0], 1u);
    if (np > 1) cs1 = atomicAdd(&fill[pb1], 1u);
    uint32_t pend = 0, cpend = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (sl[i] < 16u) place(bn[i], sl[i], wd[i]);
      else if (sl[i] != kMsdwWcSkip) pend |= 1u << i;
    }
    if (cs0 < 16u) place(pb0, cs0, pw0);
    else if (cs0 != kMsdwWcSkip) cpend |= 1u;
    if (cs1 < 16u) place(pb1, cs1, pw1);
    else if (cs1 != kMsdwWcSkip) cpend |= 2u;
    if (__builtin_popcount(pend) + __builtin_popcount(cpend) > 2) again[cur] = 1;
    uint32_t go = flush_phase();
    int rounds = 0;
    while (go) {   // (workgroup-uniform) some thread holds more than two rows: rounds until nobody holds any
      uint32_t still = 0, cstill = 0;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        if ((pend >> i) & 1u) {
          const uint32_t s = atomicAdd(&fill[bn[i]], 1u);
          if (s < 16u) place(bn[i], s, wd[i]);
          else still |= 1u << i;
        }
      }
      if (cpend & 1u) {
        const uint32_t s = atomicAdd(&fill[pb0], 1u);
        if (s < 16u) place(pb0, s, pw0);
        else cstill |= 1u;
      }
      if (cpend & 2u) {
        const uint32_t s = atomicAdd(&fill[pb1], 1u);
        if (s < 16u) place(pb1, s, pw1);
        else cstill |= 2u;
      }
      pend = still;
      cpend = cstill;
      if (pend | cpend) again[cur] = 1;
      go = flush_phase();
      if (++rounds > 64) return false;   // one digit takes most of the rows: 16 per round would take forever — flags bit 64,
    }                                    // the level is repeated by the tile-at-a-time kernel
    uint64_t nw0 = 0, nw1 = 0;
    uint32_t nb0 = 0, nb1 = 0, c = 0;
    auto push = [&](uint64_t w, uint32_t b) {
      if (c == 0) {
        nw0 = w;
        nb0 = b;
      } else {
        nw1 = w;
        nb1 = b;
      }
      ++c;
    };
    if (cpend & 1u) push(pw0, pb0);
    if (cpend & 2u) push(pw1, pb1);
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if ((pend >> i) & 1u) push(wd[i], bn[i]);
    }
    pw0 = nw0;
    pb0 = nb0;
    pw1 = nw1;
    pb1 = nb1;
    np = c < 2 ? c : 2;
    return true;
  };
  issue_at(0);
  for (uint32_t rel = 0; rel < span && !gave_up; rel += kBatch) {
    if (rel + kBatch <= span) gave_up = !batch(rel, std::true_type{});
    else gave_up = !batch(rel, std::false_type{});
  }
  if (gave_up) {   // (workgroup-uniform)
    if (tid == 0) atomicOr(&a.flags[0], 64u);
    return;
  }
  for (int rounds = 0;; ++rounds) {   // drain the carried rows
    uint64_t nw0 = 0, nw1 = 0;
    uint32_t nb0 = 0, nb1 = 0, c = 0;
    auto retry = [&](uint64_t w, uint32_t b) {
      const uint32_t s = atomicAdd(&fill[b], 1u);
      if (s < 16u) {
        place(b, s, w);
        return;
      }
      if (c == 0) {
        nw0 = w;
        nb0 = b;
      } else {
        nw1 = w;
        nb1 = b;
      }
      ++c;
    };
    if (np > 0) retry(pw0, pb0);
    if (np > 1) retry(pw1, pb1);
    pw0 = nw0;
    pb0 = nb0;
    pw1 = nw1;
    pb1 = nb1;
    np = c;
    if (np != 0) again[cur] = 1;
    if (!flush_phase()) break;
    if (rounds > 4096) {
      if (tid == 0) atomicOr(&a.flags[0], 64u);
      return;
    }
  }
  if (rf) state[tid] = (rv << 1) | 1u;
  __syncthreads();
  // the workgroup's partial lines filled up with pads, then whole lines of pads to the end of its last chunk of every bucket
  const int sub = tid & 7;
  for (int b0 = 0; b0 < nb; b0 += kMsdwThreads / 8) {   // (workgroup-uniform trip count: the shuffles below)
    const int b = b0 + (tid >> 3);
    const bool on = b < nb;
    const uint32_t f = on ? fill[b] : 0u;
    uint32_t line = 0;
    if (on && f != 0 && sub == 0) line = next_line(static_cast<uint32_t>(b));
    line = __shfl(line, lane & ~7, 64);
    if (on && f != 0) {
      arx_u32x4 d = *reinterpret_cast<const arx_u32x4*>(&vals[b * 16 + sub * 2]);
      if (static_cast<uint32_t>(sub * 2) >= f) {
        d[0] = 0xFFFFFFFFu;
        d[1] = 0xFFFFFFFFu;
      }
      if (static_cast<uint32_t>(sub * 2 + 1) >= f) {
        d[2] = 0xFFFFFFFFu;
        d[3] = 0xFFFFFFFFu;
      }
      store_piece(line, sub, d);
    }
    uint32_t s = kMsdwWcNever;
    if (on && sub == 0) s = state[b];
    s = __shfl(s, lane & ~7, 64);
    if ((s & 1u) != 0) {   // the rest of the workgroup's last chunk
      const uint32_t first = s >> 1, left = K - (first & (K - 1));
      for (uint32_t l = 0; l < left; ++l) {
        arx_u32x4 z = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        store_piece(first + l, sub, z);
      }
    }
  }
  __syncthreads();
  if (tid < nb) {   // (thread per bucket: consecutive addresses, one request per wave)
    const uint32_t f = fill[tid], s = state[tid];
    uint32_t pads = f != 0 ? 16u - f : 0u;
    if ((s & 1u) != 0) pads += 16u * (K - ((s >> 1) & (K - 1)));
    if (pads != 0) atomicAdd(&a.l1_count[tid], pads);
  }
}

// index of the last entry of start[0..nb] (nb + 1 entries, non-decreasing, start[0] == 0) that is <= g, computed by
// the first wave and returned to every thread through *slot
__device__ __forceinline__ uint32_t msdw_owner(const uint32_t* __restrict__ start, int nb, uint32_t g, uint32_t* slot) {
  if (threadIdx.x < 64) {
    uint32_t below = 0;
    for (int p = threadIdx.x; p < nb; p += 64) below += (start[p] <= g) ? 1u : 0u;
    below = wave_reduce_sum_u32(below);
    if (threadIdx.x == 0) *slot = below - 1;
  }
  __syncthreads();
  return *slot;
}

// Exact form, W4: counts of the next b2 bits inside level-1 buckets; work unit = <= kMsdwUnit rows of one bucket
__global__ __launch_bounds__(kMsdThreads) void msdw_hist1_kernel(MsdwArgs a) {
  __shared__ uint32_t h[kMsdwMaxBins2];
  __shared__ uint32_t part_s;
  const int tid = threadIdx.x;
  const int nb1 = 1 << a.b1;
  const uint32_t u = blockIdx.x;
  if (u >= a.unit_start[nb1]) return;  // over-provisioned grid
  const int nb2 = 1 << a.b2;
  for (int i = tid; i < nb2; i += kMsdThreads) h[i] = 0;
  const uint32_t p = msdw_owner(a.unit_start, nb1, u, &part_s);
  const int64_t begin = static_cast<int64_t>(a.l1_start[p]) + static_cast<int64_t>(u - a.unit_start[p]) * kMsdwUnit;
  const int64_t bucket_end = a.l1_end[p];
  const int64_t end = begin + kMsdwUnit < bucket_end ? begin + kMsdwUnit : bucket_end;
  const int shift = 64 - a.bits;
  const uint32_t mask = static_cast<uint32_t>(nb2 - 1);
  constexpr int U = 8;
  int64_t r = begin + tid;
  if (a.rec8) {   // (workgroup-uniform) words: the level-2 digit is their top b2 bits
    const uint64_t* __restrict__ words = reinterpret_cast<const uint64_t*>(a.rec_x);
    for (; r < end; r += kMsdThreads) {
      const uint64_t wd = words[r];
      if (wd != kMsdwPad) atomicAdd(&h[static_cast<uint32_t>(wd >> (64 - a.b2)) & mask], 1u);
    }
  }
  for (; r + (U - 1) * kMsdThreads < end; r += U * kMsdThreads) {
    uint64_t kk[U];
#pragma unroll
    for (int q = 0; q < U; ++q) kk[q] = msd_rec_key(a.rec_x[r + q * kMsdThreads]);
#pragma unroll
    for (int q = 0; q < U; ++q) atomicAdd(&h[static_cast<uint32_t>((kk[q] << a.kshift) >> shift) & mask], 1u);
  }
  for (; r < end; r += kMsdThreads) {
    atomicAdd(&h[static_cast<uint32_t>((msd_rec_key(a.rec_x[r]) << a.kshift) >> shift) & mask], 1u);
  }
  __syncthreads();
  uint32_t* dst = a.count2 + (static_cast<size_t>(p) << a.b2);
  for (int i = tid; i < nb2; i += kMsdThreads) {
    const uint32_t c = h[i];
    if (c != 0) atomicAdd(&dst[i], c);
  }
}

// One workgroup (1024 threads) per level-1 bucket: final positions of its 2^b2 level-2 buckets.
// Exact form (W5, before level 2): from the level-2 histogram; the cursors start there too (rec_y is compact).
// gap2 (after level 2): from the rows that arrived in every room; count2[] becomes part_in (the room's first record).
__global__ __launch_bounds__(1024) void msdw_scan1_kernel(MsdwArgs a) {
  __shared__ uint32_t wt[16];
  __shared__ uint32_t wmax[16];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const uint32_t p = blockIdx.x;
  const int nb2 = 1 << a.b2;
  const size_t base = static_cast<size_t>(p) << a.b2;
  const int per = (nb2 + 1023) / 1024;   // consecutive level-2 buckets per thread (1 up to 1024 of them)
  uint32_t cc[kMsdwMaxBins2 / 1024];
  uint32_t c = 0, mx = 0;
#pragma unroll
  for (int k = 0; k < kMsdwMaxBins2 / 1024; ++k) {
    const int d = tid * per + k;
    cc[k] = 0;
    if (k < per && d < nb2) {
      if (a.gap2) {
        const uint32_t room = a.room2[p];
        const uint32_t lo = a.y_base[p] + static_cast<uint32_t>(d) * room;
        const uint32_t cur = a.cursor2[base + d];
        if (cur > lo + room || cur < lo) {
          atomicOr(&a.flags[0], 16u);
          cc[k] = room;
        } else {
          cc[k] = cur - lo;
        }
        a.count2[base + d] = lo;
      } else {
        cc[k] = a.count2[base + d];
      }
    }
    c += cc[k];
    mx = mx > cc[k] ? mx : cc[k];
  }
  const uint32_t incl = wave_inclusive_scan_u32(c);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = __shfl_xor(mx, d, 64);
    mx = mx > o ? mx : o;
  }
  if (lane == 63) wt[wave] = incl;
  if (lane == 0) wmax[wave] = mx;
  __syncthreads();
  uint32_t pre = a.l1_out[p] + incl - c;
  uint32_t total = 0;
  for (int k = 0; k < 16; ++k) {
    if (k < wave) pre += wt[k];
    total += wt[k];
  }
#pragma unroll
  for (int k = 0; k < kMsdwMaxBins2 / 1024; ++k) {
    const int d = tid * per + k;
    if (k < per && d < nb2) {
      a.part_start[base + d] = pre;
      if (!a.gap2) {
        a.cursor2[base + d] = pre;
        a.count2[base + d] = pre;   // part_in of the bucket finish: rec_y is compact, record = final position
      }
      pre += cc[k];
    }
  }
  if (tid == 0) {
    uint32_t m = 0;
    for (int k = 0; k < 16; ++k) m = m > wmax[k] ? m : wmax[k];
    atomicMax(&a.flags[1], m);
    if (p + 1 == gridDim.x) a.part_start[base + nb2] = a.l1_out[p + 1];
    if (a.gap2 && total != a.l1_out[p + 1] - a.l1_out[p]) atomicOr(&a.flags[0], 16u);   // rows dropped by an overflowing tile
  }
}

// gap2: cursors of the level-2 rooms
__global__ __launch_bounds__(1024) void msdw_init2_kernel(MsdwArgs a) {
  const uint32_t p = blockIdx.x;
  const int nb2 = 1 << a.b2;
  for (int d = threadIdx.x; d < nb2; d += 1024) {
    a.cursor2[(static_cast<size_t>(p) << a.b2) + d] = a.y_base[p] + static_cast<uint32_t>(d) * a.room2[p];
  }
}

// W6: level 2 inside the level-1 buckets (tile map: l2_tile_start)
template <bool GAP, int RPT, bool REC8 = false>
__global__ __launch_bounds__(kMsdwThreads) void msdw_scatter2_kernel(MsdwArgs a) {
  constexpr int kTile = RPT * kMsdwThreads;
  __shared__ MsdwScatterLds lds;
  const int nb1 = 1 << a.b1;
  // a.xcd_map: XCD x takes a contiguous eighth of the tiles, i.e. whole level-1 buckets — the (tile, digit) runs of a
  // bucket then meet in one L2 instead of eight
  const uint32_t g = a.xcd_map ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  if (g >= a.l2_tile_start[nb1]) return;  // over-provisioned grid
  const uint32_t p = msdw_owner(a.l2_tile_start, nb1, g, &lds.part);
  const int64_t lo = a.l1_start[p];
  const int64_t hi = a.l1_end[p];
  const int64_t row0 = lo + static_cast<int64_t>(g - a.l2_tile_start[p]) * kTile;
  const int nrows = static_cast<int>(hi - row0 < kTile ? hi - row0 : kTile);
  // (rec8: records are words whose top b2 bits are the level-2 digit; rows are numbered in records either way)
  msdw_scatter_any_tile<REC8 ? 3 : 2, GAP ? 2 : 0, RPT, REC8>(a, lds, nullptr, nullptr, a.rec_x, row0, nrows, 1 << a.b2,
                                                              REC8 ? 64 - a.b2 : 64 - a.bits,
                                                              a.cursor2 + (static_cast<size_t>(p) << a.b2), nullptr,
                                                              GAP ? a.y_base[p] : 0u, GAP ? a.room2[p] : 0u, 16u, a.rec_y);
}

// Level 2 of the rec8 form in a workgroup SMALL enough for several to share a CU: words only (no row-id array), a staging
// buffer of STAGE words, <= 2048 bins — 56 KB of LDS with 4096 words — and T threads.  The one-per-CU workgroups of
// msdw_scatter2_kernel (147 KB: the LDS layout of the 12-byte records) run their phases — load, rank, scan, place, write —
// one after the other with the memory pipes idle in between; two or three independent workgroups per CU overlap them.
constexpr int kMsdwL2wBins = 2048;
template <int STAGE>
struct __attribute__((aligned(16))) MsdwL2wLds {
  uint64_t words[STAGE];
  uint32_t cnt[kMsdwL2wBins];
  uint32_t start[kMsdwL2wBins];
  uint32_t gbase[kMsdwL2wBins];
  uint32_t wave_tot[16];
  uint32_t part;
};

template <bool GAP, int T, int RPT, int STAGE>
__global__ __launch_bounds__(T) void msdw_scatter2w_kernel(MsdwArgs a) {
  constexpr int kTile = RPT * T;
  constexpr int K = kMsdwL2wBins / T;   // bins per thread of the scan (at most)
  static_assert(kTile % STAGE == 0 && kTile < 0xFFFF && RPT % 2 == 0 && K >= 1 && K <= kMsdwMaxBins2 / kMsdwThreads, "shape");
  __shared__ MsdwL2wLds<STAGE> lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nb1 = 1 << a.b1;
  const uint32_t g = a.xcd_map ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  if (g >= a.l2_tile_start[nb1]) return;  // over-provisioned grid
  const uint32_t p = msdw_owner(a.l2_tile_start, nb1, g, &lds.part);
  const int64_t lo_b = a.l1_start[p];
  const int64_t hi_b = a.l1_end[p];
  const int64_t row0 = lo_b + static_cast<int64_t>(g - a.l2_tile_start[p]) * kTile;
  const int nrows = static_cast<int>(hi_b - row0 < kTile ? hi_b - row0 : kTile);
  const int nb = 1 << a.b2;
  const int dsh = 64 - a.b2;
  uint32_t* __restrict__ gcursor = a.cursor2 + (static_cast<size_t>(p) << a.b2);
  const uint64_t* __restrict__ in = reinterpret_cast<const uint64_t*>(a.rec_x) + row0;
  uint64_t* __restrict__ out = reinterpret_cast<uint64_t*>(a.rec_y);
  for (int b = tid; b < nb; b += T) lds.cnt[b] = 0;
  uint64_t wd[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {   // unconditional (clamped) loads: all in flight together
    const uint32_t q = static_cast<uint32_t>(i * T + tid);
    wd[i] = in[q < static_cast<uint32_t>(nrows) ? q : static_cast<uint32_t>(nrows - 1)];
  }
  __syncthreads();
  uint32_t pos2[RPT / 2];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const uint32_t d = static_cast<uint32_t>(wd[i] >> dsh);
    const uint32_t pr = (i * T + tid < nrows && wd[i] != kMsdwPad) ? atomicAdd(&lds.cnt[d], 1u) : 0xFFFFu;
    pos2[i / 2] = (i & 1) ? (pos2[i / 2] | (pr << 16)) : pr;
    if (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  const int per = (nb + T - 1) / T;
  uint32_t cc[kMsdwMaxBins2 / kMsdwThreads];
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < kMsdwMaxBins2 / kMsdwThreads; ++k) {
    const int b = tid * per + k;
    cc[k] = (k < per && b < nb) ? lds.cnt[b] : 0u;
    mine += cc[k];
  }
  const uint32_t incl = wave_inclusive_scan_u32(mine);
  if (lane == 63) lds.wave_tot[wave] = incl;
  __syncthreads();
  uint32_t pre = incl - mine;
  for (int k = 0; k < wave; ++k) pre += lds.wave_tot[k];
  uint32_t run_base[kMsdwMaxBins2 / kMsdwThreads];
  msdw_reserve_runs(gcursor, cc, per, nb, run_base);
  const uint32_t room_base = GAP ? a.y_base[p] : 0u, room = GAP ? a.room2[p] : 0u;
#pragma unroll
  for (int k = 0; k < kMsdwMaxBins2 / kMsdwThreads; ++k) {
    const int b = tid * per + k;
    if (k < per && b < nb) {
      const uint32_t c = cc[k];
      lds.start[b] = pre;
      uint32_t base = run_base[k];
      if constexpr (GAP) {
        const uint32_t end = room_base + (static_cast<uint32_t>(b) + 1u) * room;
        if (c != 0 && (base + c > end || base + c < base)) {
          base = 0xFFFFFFFFu;
          atomicOr(&a.flags[0], 16u);
        }
      }
      lds.gbase[b] = base;
      pre += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const uint32_t d = static_cast<uint32_t>(wd[i] >> dsh);
    const uint32_t pr = (pos2[i / 2] >> ((i & 1) * 16)) & 0xFFFFu;
    const uint32_t at = pr != 0xFFFFu ? pr + lds.start[d] : 0xFFFFu;
    pos2[i / 2] = (i & 1) ? ((pos2[i / 2] & 0xFFFFu) | (at << 16)) : ((pos2[i / 2] & 0xFFFF0000u) | at);
    if (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);
  }
  const int nplaced = static_cast<int>(lds.start[nb - 1] + lds.cnt[nb - 1]);   // (pads take no place)
  for (int r = 0; r < kTile / STAGE; ++r) {
    const uint32_t lo = static_cast<uint32_t>(r) * STAGE;
    if (static_cast<int>(lo) >= nplaced) break;   // workgroup-uniform
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const uint32_t q = ((pos2[i / 2] >> ((i & 1) * 16)) & 0xFFFFu) - lo;   // (no row: 0xFFFF - lo is never inside the buffer)
      if (q < static_cast<uint32_t>(STAGE)) lds.words[q] = wd[i];
    }
    __syncthreads();
    const int cnt = nplaced - static_cast<int>(lo) < STAGE ? nplaced - static_cast<int>(lo) : STAGE;
    for (int q = tid; q < cnt; q += T) {
      const uint64_t w = lds.words[q];
      const uint32_t d = static_cast<uint32_t>(w >> dsh);
      const uint32_t gb = lds.gbase[d];
      if (GAP && gb == 0xFFFFFFFFu) continue;
      out[gb + (lo + static_cast<uint32_t>(q) - lds.start[d])] = w;
    }
    __syncthreads();
  }
}

// overflowed: 0 sorted; 1 the keys are too skewed for this form (a bucket does not fit LDS) — try the next form;
// 2 a level-2 room overflowed (gap2 only): call again with gap2 = 0 (the source may have been overwritten when it
// shares memory with rec_y).
// 3 (rec8 only) too many rows tied in the 32 key bits of their words: call again with rec8 = 0 (the source is the
// caller's column, untouched).
static int run_msd_sort_wide_form(const uint64_t* src_keys, const uint32_t* src_idx, int raw, int64_t n, MsdRec* rec_x,
                                  MsdRec* rec_y, int64_t capacity, uint8_t* tables, uint64_t* out_final, int gap2, int kshift,
                                  int rec8, hipStream_t st, int* overflowed, const MsdRec* src_rec = nullptr) {
  *overflowed = 0;
  if (n == 0) return ARX_OK;
  int lg = 0;
  while ((int64_t(1) << lg) < n) ++lg;   // ceil(log2 n)
  MsdwArgs a{};
  a.src_keys = src_keys;
  a.src_idx = src_idx;
  a.src_rec = src_rec;
  a.raw = raw;
  a.n = n;
  a.kshift = kshift;
  a.xcd_map = (g_sort_xcd_map & 1) != 0;
  a.xcd_map1 = (g_sort_xcd_map & 4) != 0;
  // 1024 < average bucket <= 2048 rows (up to 2^31 rows): the 256-thread finish holds them, four or five per CU.
  // 2e9 rows: 2^20 buckets 31.5 ms, 2^19 buckets (512-thread finish) 32.9 ms (profiles/r03_k_sort_ab.txt)
  a.bits = std::max(2, std::min(std::min(lg - 11, kMsdwMaxBits), 64 - kshift));
  if (g_sort_msd_wide_bits > 0) a.bits = std::max(2, std::min(int(g_sort_msd_wide_bits), 64 - kshift));
  // level 2 takes up to b2max bits (<= 4096 bins, default 1024): level 1 scatters over the whole array, where fewer
  // bins and longer runs pay; level 2 works inside a bucket whose short runs meet in one L2 (xcd_contiguous) — but a
  // 4096-bin level 2 loses more there than level 1 gains (profiles/r02_ah)
  a.b2 = int(g_sort_msd_wide_b2max) > 0 ? std::min(std::min(int(g_sort_msd_wide_b2max), 12), a.bits - 1)
                                   : a.bits - a.bits / 2;   // 0: the even split
  a.b1 = a.bits - a.b2;
  if (a.b1 > 10) {   // (level 1 has at most 1024 bins)
    a.b1 = 10;
    a.b2 = a.bits - a.b1;
  }
  const bool roomy = capacity < (int64_t(1) << 32);   // record positions are 32-bit
  a.capacity = roomy ? capacity : n;
  uint32_t* t = reinterpret_cast<uint32_t*>(tables);
  const size_t big = (size_t(1) << kMsdwMaxBits) + 64;
  const size_t small = kMsdwMaxBins + 64;
  a.count2 = t;
  a.part_start = t + big;
  a.cursor2 = t + 2 * big;
  a.l1_count = t + 3 * big;
  a.l1_start = a.l1_count + small;
  a.cursor1 = a.l1_start + small;
  a.l2_tile_start = a.cursor1 + small;
  a.unit_start = a.l2_tile_start + small;
  a.flags = a.unit_start + small;
  a.l1_end = a.flags + small;
  a.l1_out = a.l1_end + small;
  a.room2 = a.l1_out + small;
  a.y_base = a.room2 + small;
  a.wc_cursor = a.y_base + small;
  a.rec_x = rec_x;
  a.rec_y = rec_y;
  a.gap2 = (gap2 != 0 && roomy) ? 1 : 0;
  a.rec8 = (rec8 != 0 && raw != 0 && src_idx == nullptr && src_rec == nullptr) ? 1 : 0;   // (ties go back to the column: row id = position)
  const int nb1 = 1 << a.b1;
  const size_t nparts = size_t(1) << a.bits;
  // (each knob is read ONCE: the tile sizes, the grids and the template dispatch below must agree even if arx_set_option runs
  //  beside this call — ADVICE r3)
  const int rpt1 = g_sort_msd_wide_rpt1, rpt2 = g_sort_msd_wide_rpt2;
  a.tile1 = rpt1 * kMsdwThreads;
  a.tile2 = rpt2 * kMsdwThreads;
  const int l2w = (a.rec8 && (1 << a.b2) <= kMsdwL2wBins) ? int(g_sort_msd_wide_l2w) : 0;
  if (l2w == 1 || l2w == 2 || l2w == 3) a.tile2 = 8192;
  const unsigned grid0 = static_cast<unsigned>(ceil_div(n, kMsdwTile));   // the sampled histogram reads 8192-row chunks
  const unsigned grid1 = static_cast<unsigned>(ceil_div(n, a.tile1));
  // rec8: level 1 write-combined by persistent workgroups when its bins fit their LDS lines and the pads (one line per
  // bin and workgroup at most) fit half of the buffers' slack
  // Level-1 bucket sizes: estimated from 1 tile in 2^shift (the buckets then get room to spare and level 2 reads
  // what actually arrived), or — shift 0, and whenever an estimate turned out too small — counted exactly.
  unsigned int max_room = 0;
  int sample_shift = roomy ? g_sort_msd_wide_sample_shift : 0;
  while (sample_shift > 0 && (static_cast<int64_t>(grid0) >> sample_shift) < 8) --sample_shift;   // too few chunks to sample
  // (the append form keeps ONE line per bin: with few bins a batch of 4096 rows brings a bin many times its 16 slots and the
  //  batch needs round after round — below 256 bins the rank-and-stage kernel of round 5)
  a.wc_form = (g_sort_msd_wide_wc_form == 1 || nb1 < kMsdwWc2MinBins) ? 1 : 2;
  // (form 2: contiguous shares of >= 2^17 rows per workgroup keep the chunk tails small; its pads — two chunks per bucket and
  //  workgroup at most, + the rooms rounded up to chunks — must fit a third of the buffers' slack: the sampled rooms take up
  //  to n / 32 + n / 8 of the n / 4)
  int64_t wc_groups = std::min<int64_t>(int(g_sort_msd_wide_wc), grid1);
  int64_t wc_slack = int64_t(16) * wc_groups * nb1;
  a.wc_k = 0;
  if (a.wc_form == 2) {
    const int64_t groups2 = std::max<int64_t>(1, std::min<int64_t>(int(g_sort_msd_wide_wc), n / std::max<int64_t>(1, int64_t(g_sort_msd_wide_wc_min_rows))));
    for (int k : {kMsdwWcK, 2, 1}) {
      const int64_t slack2 = int64_t(16) * k * 2 * groups2 * nb1 + int64_t(16) * k * nb1;
      if ((sample_shift > 0 ? 3 : 1) * slack2 <= a.capacity - n) {
        a.wc_k = k;
        wc_groups = groups2;
        wc_slack = slack2;
        break;
      }
    }
    if (a.wc_k == 0) a.wc_form = 1;   // (its pads do not fit this input's slack: round 5's kernel pads one line per bin and workgroup)
  }
  a.wc1 = (a.rec8 && roomy && int(g_sort_msd_wide_wc) > 0 && wc_groups > 0 &&
           (a.wc_form == 2 ? nb1 <= kMsdwMaxBins : nb1 <= kMsdwWcBins && 2 * wc_slack <= a.capacity - n)) ? static_cast<int>(wc_groups) : 0;
  const int wc_r = g_sort_msd_wide_wc_rows == 8 ? 8 : g_sort_msd_wide_wc_rows == 6 ? 6 : 4;   // (read once: the grid's shares and the kernel must agree)
  if (a.wc1 > 0 && a.wc_form == 2) {
    const int64_t batch = int64_t(wc_r) * kMsdwThreads;
    a.wc_rows_per_wg = ceil_div(ceil_div(n, a.wc1), batch) * batch;
  }
  for (;;) {
    a.sample_shift = sample_shift;
    unsigned nch;
    if (sample_shift > 0) {
      a.chunk_rows = kMsdwTile;
      nch = static_cast<unsigned>(ceil_div(static_cast<int64_t>(grid0), int64_t(1) << sample_shift));
    } else {
      const int64_t ntiles0 = ceil_div(n, kMsdTile);
      a.chunk_rows = std::max<int64_t>(1, ceil_div(ntiles0, kMsdMaxChunks)) * kMsdTile;
      nch = static_cast<unsigned>(ceil_div(n, a.chunk_rows));
    }
    ARX_HIP(hipMemsetAsync(a.l1_count, 0, static_cast<size_t>(nb1) * 4, st));
    ARX_HIP(hipMemsetAsync(a.flags, 0, kMsdwFlagWords * 4, st));
    if (raw) {
      hipLaunchKernelGGL((msdw_hist0_kernel<true>), dim3(nch), dim3(kMsdThreads), 0, st, a);
    } else {
      hipLaunchKernelGGL((msdw_hist0_kernel<false>), dim3(nch), dim3(kMsdThreads), 0, st, a);
    }
    ARX_CHECK_LAUNCH("msdw_hist0_kernel");
    hipLaunchKernelGGL(msdw_scan0_kernel, dim3(1), dim3(1024), 0, st, a);
    ARX_CHECK_LAUNCH("msdw_scan0_kernel");
#define ARX_MSDW_SCATTER1(RAW, OUT8)                                                                                      \
  switch (rpt1) {                                                                                  \
    case 24: hipLaunchKernelGGL((msdw_scatter1_kernel<RAW, 24, OUT8>), dim3(grid1), dim3(kMsdwThreads), 0, st, a); break; \
    case 16: hipLaunchKernelGGL((msdw_scatter1_kernel<RAW, 16, OUT8>), dim3(grid1), dim3(kMsdwThreads), 0, st, a); break; \
    default: hipLaunchKernelGGL((msdw_scatter1_kernel<RAW, 8, OUT8>), dim3(grid1), dim3(kMsdwThreads), 0, st, a); break;  \
  }
    if (a.wc1 > 0 && a.wc_form == 2) {
      const int kt = (a.raw >> 4) & 7;
      if (wc_r == 8 && kt == 0) {
        hipLaunchKernelGGL((msdw_scatter1wc2_kernel<8, 0>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else if (wc_r == 6 && kt == 0) {
        hipLaunchKernelGGL((msdw_scatter1wc2_kernel<6, 0>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else if (wc_r == 8) {
        hipLaunchKernelGGL((msdw_scatter1wc2_kernel<8, -1>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else if (wc_r == 6) {
        hipLaunchKernelGGL((msdw_scatter1wc2_kernel<6, -1>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else if (kt == 0 && g_sort_msd_wide_wc_typed == 2) {   // (A/B: the queue of full lines)
        hipLaunchKernelGGL((msdw_scatter1wc2_kernel<4, 0, true>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else if (kt == 0 && g_sort_msd_wide_wc_typed) {
        hipLaunchKernelGGL((msdw_scatter1wc2_kernel<4, 0>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else if (kt == 1 && g_sort_msd_wide_wc_typed) {
        hipLaunchKernelGGL((msdw_scatter1wc2_kernel<4, 1>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else if (kt == 4 && g_sort_msd_wide_wc_typed) {
        hipLaunchKernelGGL((msdw_scatter1wc2_kernel<4, 4>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else if (kt == 2 && g_sort_msd_wide_wc_typed) {
        hipLaunchKernelGGL((msdw_scatter1wc2_kernel<4, 2>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else if (kt == 3 && g_sort_msd_wide_wc_typed) {
        hipLaunchKernelGGL((msdw_scatter1wc2_kernel<4, 3>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else if (kt == 5 && g_sort_msd_wide_wc_typed) {
        hipLaunchKernelGGL((msdw_scatter1wc2_kernel<4, 5>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else {
        hipLaunchKernelGGL((msdw_scatter1wc2_kernel<4, -1>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      }
    } else if (a.wc1 > 0) {
      // tiles of 16 rows per thread with the next tile's keys prefetched (sort_msd_wide_wc_prefetch), or rpt1 rows without
      const int wc_rpt = g_sort_msd_wide_wc_prefetch ? (rpt1 >= 16 ? 16 : 8) : rpt1;
      if (g_sort_msd_wide_wc_prefetch && wc_rpt == 16) {
        hipLaunchKernelGGL((msdw_scatter1wc_kernel<16, true>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else if (g_sort_msd_wide_wc_prefetch) {
        hipLaunchKernelGGL((msdw_scatter1wc_kernel<8, true>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else if (wc_rpt == 24) {
        hipLaunchKernelGGL((msdw_scatter1wc_kernel<24, false>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else if (wc_rpt == 16) {
        hipLaunchKernelGGL((msdw_scatter1wc_kernel<16, false>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      } else {
        hipLaunchKernelGGL((msdw_scatter1wc_kernel<8, false>), dim3(a.wc1), dim3(kMsdwThreads), 0, st, a);
      }
    } else if (a.src_rec != nullptr) {
      switch (rpt1) {
        case 24: hipLaunchKernelGGL((msdw_scatter1r_kernel<24>), dim3(grid1), dim3(kMsdwThreads), 0, st, a); break;
        case 16: hipLaunchKernelGGL((msdw_scatter1r_kernel<16>), dim3(grid1), dim3(kMsdwThreads), 0, st, a); break;
        default: hipLaunchKernelGGL((msdw_scatter1r_kernel<8>), dim3(grid1), dim3(kMsdwThreads), 0, st, a); break;
      }
    } else if (a.rec8) {
      ARX_MSDW_SCATTER1(true, true)
    } else if (raw) {
      ARX_MSDW_SCATTER1(true, false)
    } else {
      ARX_MSDW_SCATTER1(false, false)
    }
#undef ARX_MSDW_SCATTER1
    ARX_CHECK_LAUNCH("msdw_scatter1_kernel");
    hipLaunchKernelGGL(msdw_scan0b_kernel, dim3(1), dim3(1024), 0, st, a);
    ARX_CHECK_LAUNCH("msdw_scan0b_kernel");
    if (sample_shift == 0 && a.gap2 == 0 && !(a.wc1 > 0 && a.wc_form == 2)) break;   // exact counts all the way: nothing to look at yet
    unsigned int fl4[4] = {0, 0, 0, 0};
    ARX_HIP(hipMemcpyAsync(fl4, a.flags, 16, hipMemcpyDeviceToHost, st));
    ARX_HIP(hipStreamSynchronize(st));
    const unsigned int fl = fl4[0];
    max_room = fl4[3];
    if ((fl & 64u) != 0) {   // the append form gave up on a hot digit (rounds without end): the tile-at-a-time level 1 instead
      a.wc1 = 0;
      g_sort_wide_wc_given_up.fetch_add(1, std::memory_order_relaxed);
      continue;
    }
    if ((fl & 4u) != 0) {   // a level-1 bucket outgrew its room; the source is untouched (level 2 has not run)
      if (sample_shift == 0) {
        set_error("array_sort_indices: internal error (exact level-1 histogram disagrees with the scatter)");
        return ARX_INVALID;
      }
      if (g_sort_msd_wide_sample_strict) {
        set_error("array_sort_indices: sampled level-1 histogram underestimated a bucket (sort_msd_wide_sample_strict)");
        return ARX_INVALID;
      }
      sample_shift = 0;
      continue;
    }
    if ((fl & 8u) != 0) {   // fixed level-2 rooms are not possible (a room beyond LDS, or no space): count exactly
      a.gap2 = 0;
      ARX_HIP(hipMemsetAsync(a.flags, 0, 4, st));
    }
    break;
  }
  // (write-combined level 1: the buckets hold up to one padded line per bin and workgroup beside their rows)
  const int64_t spanned = n + (a.wc1 > 0 ? wc_slack : 0);
  const unsigned grid2 = static_cast<unsigned>(ceil_div(spanned, a.tile2)) + static_cast<unsigned>(nb1);
#define ARX_MSDW_SCATTER2_(GAP, REC8)                                                                                      \
  switch (rpt2) {                                                                                  \
    case 24: hipLaunchKernelGGL((msdw_scatter2_kernel<GAP, 24, REC8>), dim3(grid2), dim3(kMsdwThreads), 0, st, a); break; \
    case 16: hipLaunchKernelGGL((msdw_scatter2_kernel<GAP, 16, REC8>), dim3(grid2), dim3(kMsdwThreads), 0, st, a); break; \
    default: hipLaunchKernelGGL((msdw_scatter2_kernel<GAP, 8, REC8>), dim3(grid2), dim3(kMsdwThreads), 0, st, a); break;  \
  }
#define ARX_MSDW_SCATTER2(GAP) \
  if (l2w == 1) {              \
    hipLaunchKernelGGL((msdw_scatter2w_kernel<GAP, 512, 16, 4096>), dim3(grid2), dim3(512), 0, st, a); \
  } else if (l2w == 2) {       \
    hipLaunchKernelGGL((msdw_scatter2w_kernel<GAP, 512, 16, 2048>), dim3(grid2), dim3(512), 0, st, a); \
  } else if (l2w == 3) {       \
    hipLaunchKernelGGL((msdw_scatter2w_kernel<GAP, 1024, 8, 4096>), dim3(grid2), dim3(1024), 0, st, a); \
  } else if (a.rec8) {         \
    ARX_MSDW_SCATTER2_(GAP, true) \
  } else {                     \
    ARX_MSDW_SCATTER2_(GAP, false) \
  }
  unsigned int max_part = 0;
  if (a.gap2) {
    hipLaunchKernelGGL(msdw_init2_kernel, dim3(static_cast<unsigned>(nb1)), dim3(1024), 0, st, a);
    ARX_CHECK_LAUNCH("msdw_init2_kernel");
    ARX_MSDW_SCATTER2(true)
    ARX_CHECK_LAUNCH("msdw_scatter2_kernel");
    hipLaunchKernelGGL(msdw_scan1_kernel, dim3(static_cast<unsigned>(nb1)), dim3(1024), 0, st, a);
    ARX_CHECK_LAUNCH("msdw_scan1_kernel");
    max_part = max_room != 0 ? max_room : kBktCapSmall;   // rooms never exceed it
  } else {
    ARX_HIP(hipMemsetAsync(a.count2, 0, nparts * 4, st));
    const unsigned units = static_cast<unsigned>(ceil_div(spanned, kMsdwUnit) + nb1);
    hipLaunchKernelGGL(msdw_hist1_kernel, dim3(units), dim3(kMsdThreads), 0, st, a);
    ARX_CHECK_LAUNCH("msdw_hist1_kernel");
    hipLaunchKernelGGL(msdw_scan1_kernel, dim3(static_cast<unsigned>(nb1)), dim3(1024), 0, st, a);
    ARX_CHECK_LAUNCH("msdw_scan1_kernel");
    unsigned int fl[2] = {0, 0};
    ARX_HIP(hipMemcpyAsync(fl, a.flags, 8, hipMemcpyDeviceToHost, st));
    ARX_HIP(hipStreamSynchronize(st));
    if ((fl[0] & 4u) != 0) {
      set_error("array_sort_indices: internal error (exact level-1 histogram disagrees with the scatter)");
      return ARX_INVALID;
    }
    max_part = fl[1];
    if (max_part > static_cast<unsigned int>(kBktCap)) {   // skewed keys: a bucket would not fit LDS
      *overflowed = 1;
      return ARX_OK;
    }
    ARX_MSDW_SCATTER2(false)
    ARX_CHECK_LAUNCH("msdw_scatter2_kernel");
  }
#undef ARX_MSDW_SCATTER2
#undef ARX_MSDW_SCATTER2_
  MsdArgs f{};
  f.n = n;
  f.bits = a.bits;
  f.kshift = kshift;
  f.xcd_map = g_sort_xcd_map & 2;
  f.part_start = a.part_start;
  f.part_in = a.count2;
  f.overflow = a.flags;
  f.out_final = out_final;
  const bool small_bkt = max_part <= static_cast<unsigned int>(kBktCapSmall);
  const bool tiny_bkt = g_sort_msd_tiny_bucket != 0 && max_part <= static_cast<unsigned int>(kBktCapTiny);
  const int cpt_bits = g_sort_msd_bucket_cpt == 8 && (tiny_bkt || small_bkt) ? 1 : 0;   // twice the sub-buckets
  f.b3 = std::max(0, std::min(std::min(lg - (g_sort_msd_final_rows_log2 - 1) - a.bits,
                                       (tiny_bkt ? 10 : small_bkt ? 11 : 12) + cpt_bits),
                              64 - kshift - a.bits));
  const uint64_t* recs = reinterpret_cast<const uint64_t*>(rec_y);
  const bool tiny9 = tiny_bkt && g_sort_msd_tiny_bucket >= 2 && max_part <= static_cast<unsigned int>(kBktCapTiny9);
  if (a.rec8) {
    f.src_keys = src_keys;
    f.raw = raw;
    f.b1 = a.b1;
    f.b2 = a.b2;
    f.tie_shift = g_sort_msd_wide_rec8_tie_shift;
#define ARX_BUCKET2W(T, CPT, R) \
  hipLaunchKernelGGL((msd_bucket2w_kernel<T, CPT, R>), dim3(static_cast<unsigned>(nparts)), dim3(T), 0, st, f, recs)
    if (tiny9 && cpt_bits) {
      ARX_BUCKET2W(kBktThreadsTiny, 8, kBktRowsTiny9);
    } else if (tiny9) {
      ARX_BUCKET2W(kBktThreadsTiny, 4, kBktRowsTiny9);
    } else if (tiny_bkt && cpt_bits) {
      ARX_BUCKET2W(kBktThreadsTiny, 8, kBktRows);
    } else if (tiny_bkt) {
      ARX_BUCKET2W(kBktThreadsTiny, 4, kBktRows);
    } else if (small_bkt && cpt_bits) {
      ARX_BUCKET2W(kBktThreadsSmall, 8, kBktRows);
    } else if (small_bkt) {
      ARX_BUCKET2W(kBktThreadsSmall, 4, kBktRows);
    } else {
      ARX_BUCKET2W(kBktThreads, 4, kBktRows);
    }
#undef ARX_BUCKET2W
  } else if (tiny9 && cpt_bits) {
    hipLaunchKernelGGL((msd_bucket2_kernel<false, kBktThreadsTiny, true, 8, kBktRowsTiny9>), dim3(static_cast<unsigned>(nparts)),
                       dim3(kBktThreadsTiny), 0, st, f, recs, nullptr);
  } else if (tiny9) {
    hipLaunchKernelGGL((msd_bucket2_kernel<false, kBktThreadsTiny, true, 4, kBktRowsTiny9>), dim3(static_cast<unsigned>(nparts)),
                       dim3(kBktThreadsTiny), 0, st, f, recs, nullptr);
  } else if (tiny_bkt && cpt_bits) {
    hipLaunchKernelGGL((msd_bucket2_kernel<false, kBktThreadsTiny, true, 8>), dim3(static_cast<unsigned>(nparts)),
                       dim3(kBktThreadsTiny), 0, st, f, recs, nullptr);
  } else if (tiny_bkt) {
    hipLaunchKernelGGL((msd_bucket2_kernel<false, kBktThreadsTiny, true>), dim3(static_cast<unsigned>(nparts)),
                       dim3(kBktThreadsTiny), 0, st, f, recs, nullptr);
  } else if (small_bkt && cpt_bits) {
    hipLaunchKernelGGL((msd_bucket2_kernel<false, kBktThreadsSmall, true, 8>), dim3(static_cast<unsigned>(nparts)),
                       dim3(kBktThreadsSmall), 0, st, f, recs, nullptr);
  } else if (small_bkt) {
    hipLaunchKernelGGL((msd_bucket2_kernel<false, kBktThreadsSmall, true>), dim3(static_cast<unsigned>(nparts)),
                       dim3(kBktThreadsSmall), 0, st, f, recs, nullptr);
  } else {
    hipLaunchKernelGGL((msd_bucket2_kernel<false, kBktThreads, true>), dim3(static_cast<unsigned>(nparts)),
                       dim3(kBktThreads), 0, st, f, recs, nullptr);
  }
  ARX_CHECK_LAUNCH("msd_bucket2_kernel");
  static_assert(kMsdwFlagWords <= kMsdwMaxBins + 64, "the flag block is one `small` table");
  std::vector<unsigned int> flw(a.rec8 ? kMsdwFlagWords : 1, 0u);
  ARX_HIP(hipMemcpyAsync(flw.data(), a.flags, flw.size() * 4, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  const unsigned int flag = flw[0];
  g_sort_wide_runs.fetch_add(1, std::memory_order_relaxed);
  if (a.rec8) {
    int64_t ties = 0;
    for (int k = 0; k < 32; ++k) ties += flw[kMsdwTieSlot0 + 32 * k];
    g_sort_wide_rec8_runs.fetch_add(1, std::memory_order_relaxed);
    g_sort_wide_rec8_ties.fetch_add(ties, std::memory_order_relaxed);
    if (a.wc1 > 0) g_sort_wide_wc_runs.fetch_add(1, std::memory_order_relaxed);
  }
  if ((flag & 32u) != 0 && (flag & ~32u) == 0) {
    g_sort_wide_rec8_given_up.fetch_add(1, std::memory_order_relaxed);
    *overflowed = 3;
  } else if ((flag & 16u) != 0) {
    if (g_sort_msd_wide_sample_strict) {
      set_error("array_sort_indices: a level-2 room underestimated its bucket (sort_msd_wide_sample_strict)");
      return ARX_INVALID;
    }
    *overflowed = 2;
  } else {
    *overflowed = flag != 0;
  }
  return ARX_OK;
}

// *rec8: in, whether to try the word form; out, cleared once an attempt gave up (the caller's later attempts — counted
// level-2 buckets after a room overflow — then start with full records)
static int run_msd_sort_wide(const uint64_t* src_keys, const uint32_t* src_idx, int raw, int64_t n, MsdRec* rec_x,
                             MsdRec* rec_y, int64_t capacity, uint8_t* tables, uint64_t* out_final, int gap2, int kshift,
                             hipStream_t st, int* overflowed, int* rec8, const MsdRec* src_rec = nullptr) {
  int rc = run_msd_sort_wide_form(src_keys, src_idx, raw, n, rec_x, rec_y, capacity, tables, out_final, gap2, kshift, *rec8, st,
                                  overflowed, src_rec);
  if (rc == ARX_OK && *overflowed == 3) {
    *rec8 = 0;
    rc = run_msd_sort_wide_form(src_keys, src_idx, raw, n, rec_x, rec_y, capacity, tables, out_final, gap2, kshift, 0, st,
                                overflowed, src_rec);
  }
  return rc;
}

// Inputs beyond ~2^28 rows: one extra unstable level on the top b0 bits cuts the array into
// 2^b0 segments of ~2^27 rows (32 B/row), then every segment runs the pipeline above on the bits
// below (kshift = b0).  Synchronous.
static int run_msd_sort_segmented(const uint64_t* src_keys, const uint32_t* src_idx, int raw, int64_t n,
                                  uint64_t* keys_p, uint32_t* idx_p, uint64_t* keys_q, uint32_t* idx_q,
                                  uint8_t* tables, uint64_t* out_final, hipStream_t st, int* overflowed, int kshift = 0) {
  int b0 = 1;
  while ((n >> b0) > std::min<int64_t>(int64_t(g_sort_msd_segment_rows), int64_t(1) << 27) && b0 < 7) ++b0;
  b0 = std::max(b0, std::min(int(g_sort_msd_seg_min_bits), 7));
  MsdArgs a{};
  a.src_keys = src_keys;
  a.src_idx = src_idx;
  a.raw = raw;
  a.n = n;
  a.kshift = kshift;
  a.bits = b0;
  a.b1 = b0;
  a.b2 = 0;
  a.b3 = 0;
  const int64_t ntiles = ceil_div(n, kMsdTile);
  const int64_t chunk_tiles = std::max<int64_t>(1, ceil_div(ntiles, kMsdMaxChunks));
  a.chunk_rows = chunk_tiles * kMsdTile;
  a.nchunks = ceil_div(ntiles, chunk_tiles);
  const size_t np = kMsdTableWords;
  uint32_t* t = reinterpret_cast<uint32_t*>(tables);
  a.part_count = t;
  a.part_start = t + np;
  a.cursor2 = t + 2 * np;
  a.hist1 = t + 3 * np;
  a.l1_start = a.hist1 + size_t(128) * kMsdMaxChunks;
  a.l2_tile_start = a.l1_start + 192;
  a.overflow = a.l2_tile_start + 192;
  a.keys_x = keys_p;
  a.idx_x = idx_p;
  const int nseg = 1 << b0;
  ARX_HIP(hipMemsetAsync(a.part_count, 0, static_cast<size_t>(nseg) * 4, st));
  const unsigned nch = static_cast<unsigned>(a.nchunks);
  if (raw) {
    hipLaunchKernelGGL((msd_hist_kernel<true>), dim3(nch), dim3(kMsdThreads), 0, st, a);
  } else {
    hipLaunchKernelGGL((msd_hist_kernel<false>), dim3(nch), dim3(kMsdThreads), 0, st, a);
  }
  hipLaunchKernelGGL(msd_scan_a_kernel, dim3(1), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(msd_scan_b_kernel, dim3(1u << a.b1), dim3(64), 0, st, a);
  if (raw) {
    hipLaunchKernelGGL((msd_scatter1_kernel<true>), dim3(nch), dim3(kMsdThreads), 0, st, a);
  } else {
    hipLaunchKernelGGL((msd_scatter1_kernel<false>), dim3(nch), dim3(kMsdThreads), 0, st, a);
  }
  ARX_CHECK_LAUNCH("msd level 0");
  uint32_t seg_start[129];
  ARX_HIP(hipMemcpyAsync(seg_start, a.part_start, static_cast<size_t>(nseg + 1) * 4, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  *overflowed = 0;
  for (int sgm = 0; sgm < nseg; ++sgm) {
    const int64_t lo = seg_start[sgm];
    const int64_t m = static_cast<int64_t>(seg_start[sgm + 1]) - lo;
    int ovf = 0;
    // the segment's rows live in (keys_p, idx_p): levels ping-pong p -> q -> p -> q
    const int rc = run_msd_sort(keys_p + lo, idx_p + lo, 0, m, keys_q + lo, idx_q + lo, keys_p + lo, idx_p + lo,
                                tables, out_final + lo, st, &ovf, kshift + b0);
    if (rc != ARX_OK) return rc;
    if (ovf) {
      *overflowed = 1;
      return ARX_OK;
    }
  }
  return ARX_OK;
}

}  // namespace arx

using namespace arx;

extern "C" {

size_t arx_sort_indices_workspace_bytes(int64_t length) {
  if (length < 0) length = 0;
  return make_plan(length).total;
}

int arx_sort_indices(const ArxSpan* values, int key_type, int order, int null_placement, void* ws,
                     size_t ws_bytes, uint64_t* out_indices, void* stream) {
  if (values == nullptr) {
    set_error("values is NULL");
    return ARX_INVALID;
  }
  if (key_type < ARX_KEY_UINT64 || key_type > ARX_KEY_FLOAT32) {
    set_error("arx_sort_indices: unsupported key type %d", key_type);
    return ARX_NOT_IMPLEMENTED;
  }
  const int64_t len = values->length;
  if (len < 0 || values->offset < 0) {
    set_error("negative length/offset");
    return ARX_INVALID;
  }
  if (len == 0) return ARX_OK;
  if (len > static_cast<int64_t>(UINT32_MAX)) {
    set_error("arx_sort_indices: more than UINT32_MAX rows is not implemented");
    return ARX_NOT_IMPLEMENTED;
  }
  if (order != ARX_SORT_ASCENDING && order != ARX_SORT_DESCENDING) {
    set_error("bad sort order %d", order);
    return ARX_INVALID;
  }
  if (null_placement != ARX_NULLS_AT_START && null_placement != ARX_NULLS_AT_END) {
    set_error("bad null placement %d", null_placement);
    return ARX_INVALID;
  }
  if (values->data == nullptr || out_indices == nullptr || ws == nullptr) {
    set_error("values/out/ws buffer is NULL");
    return ARX_INVALID;
  }
  SortPlan plan = make_plan(len);
  if (ws_bytes < plan.total) {
    set_error("sort workspace too small: %zu < %zu", ws_bytes, plan.total);
    return ARX_INVALID;
  }
  if ((reinterpret_cast<uint64_t>(ws) & 255) != 0) {
    set_error("sort workspace must be 256-byte aligned");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  uint8_t* w = static_cast<uint8_t*>(ws);
  uint64_t* keys_a = reinterpret_cast<uint64_t*>(w + plan.off_keys_a);
  uint64_t* keys_b = reinterpret_cast<uint64_t*>(w + plan.off_keys_b);
  uint32_t* idx_a = reinterpret_cast<uint32_t*>(w + plan.off_idx_a);
  uint32_t* idx_b = reinterpret_cast<uint32_t*>(w + plan.off_idx_b);
  // the same memory as arrays of 12-byte records (wide form): (keys, idx) pairs are adjacent in the plan
  MsdRec* rec_a = reinterpret_cast<MsdRec*>(w + plan.off_keys_a);
  MsdRec* rec_b = reinterpret_cast<MsdRec*>(w + plan.off_keys_b);
  const int64_t rec_cap = std::max<int64_t>(len, 1) + msdw_slack_rows(std::max<int64_t>(len, 1));
  uint32_t* hist = reinterpret_cast<uint32_t*>(w + plan.off_hist);
  uint32_t* totals = reinterpret_cast<uint32_t*>(w + plan.off_totals);
  uint32_t* rows = reinterpret_cast<uint32_t*>(w + plan.off_rows);
  uint64_t* sortable_bits = reinterpret_cast<uint64_t*>(w + plan.off_float_bits);
  uint64_t* nan_bits = sortable_bits + ceil_div(len, 64);
  void* sel_ws = w + plan.off_sel_ws;
  const size_t sel_ws_bytes = plan.total - plan.off_sel_ws;

  const bool is_float = key_type == ARX_KEY_FLOAT64 || key_type == ARX_KEY_FLOAT32;
  const int key_width = (key_type == ARX_KEY_UINT32 || key_type == ARX_KEY_INT32 || key_type == ARX_KEY_FLOAT32) ? 4 : 8;
  const uint8_t* vals = static_cast<const uint8_t*>(values->data) + values->offset * key_width;
  const bool has_nulls = values->null_count != 0 && values->validity != nullptr;
  const int xf = make_xf(key_type, order == ARX_SORT_DESCENDING);

  // ---- null-like partition (PartitionNulls, stable): nulls, and for floats NaNs, keep row order
  // and sit at the end (values, NaNs, nulls) or the start (nulls, NaNs, values)
  int64_t n_null = 0, n_nan = 0;
  if (is_float) {
    const Bits vb = make_bits(has_nulls ? values->validity : nullptr, values->offset, len);
    const unsigned g = static_cast<unsigned>(std::min<int64_t>(ceil_div(ceil_div(len, 64), kWavesPerBlock), 2048));
    hipLaunchKernelGGL(sort_float_classify_kernel, dim3(g), dim3(kBlock), 0, st, vals,
                       key_type == ARX_KEY_FLOAT32 ? 1 : 0, vb, len, sortable_bits, nan_bits);
    ARX_CHECK_LAUNCH("sort_float_classify_kernel");
  }
  if (has_nulls) {
    const int rc = selection_bit_positions(values->validity, values->offset, len, /*invert=*/true, sel_ws,
                                           sel_ws_bytes, rows, &n_null, st);
    if (rc != ARX_OK) return rc;
    if (n_null > 0) {
      uint64_t* null_dst = null_placement == ARX_NULLS_AT_START ? out_indices : out_indices + (len - n_null);
      const unsigned g = static_cast<unsigned>(std::min<int64_t>(ceil_div(n_null, kBlock), 2048));
      hipLaunchKernelGGL(widen_kernel, dim3(g), dim3(kBlock), 0, st, rows, n_null, null_dst);
      ARX_CHECK_LAUNCH("widen_kernel");
    }
  }
  if (is_float) {
    const int rc = selection_bit_positions(nan_bits, 0, len, /*invert=*/false, sel_ws, sel_ws_bytes, rows, &n_nan, st);
    if (rc != ARX_OK) return rc;
    if (n_nan > 0) {
      uint64_t* nan_dst = null_placement == ARX_NULLS_AT_START ? out_indices + n_null
                                                               : out_indices + (len - n_null - n_nan);
      const unsigned g = static_cast<unsigned>(std::min<int64_t>(ceil_div(n_nan, kBlock), 2048));
      hipLaunchKernelGGL(widen_kernel, dim3(g), dim3(kBlock), 0, st, rows, n_nan, nan_dst);
      ARX_CHECK_LAUNCH("widen_kernel");
    }
  }
  const int64_t n_valid = len - n_null - n_nan;
  if (n_valid == 0) return ARX_OK;
  const uint32_t* valid_rows = nullptr;
  if (n_valid < len) {
    int64_t got = 0;
    const int rc = is_float ? selection_bit_positions(sortable_bits, 0, len, false, sel_ws, sel_ws_bytes, rows, &got, st)
                            : selection_bit_positions(values->validity, values->offset, len, false, sel_ws,
                                                      sel_ws_bytes, rows, &got, st);
    if (rc != ARX_OK) return rc;
    valid_rows = rows;
  }
  uint64_t* final_dst = null_placement == ARX_NULLS_AT_START ? out_indices + (len - n_valid) : out_indices;

  // ---- large inputs: MSD-hybrid path; falls through to the LSD passes if a bucket overflowed
  // auto: from 4M rows up; beyond 2^27 rows an extra level on the top bits cuts ~2^27-row segments
  const bool try_msd = g_sort_msd != 0 && n_valid < (int64_t(1) << 32) - kMsdTile &&
                       (g_sort_msd == 1 ? n_valid >= 256 : n_valid >= g_sort_msd_min_rows);
  const bool segmented = n_valid > g_sort_msd_segment_rows;
  const unsigned gprep = static_cast<unsigned>(std::min<int64_t>(ceil_div(n_valid, kBlock), 2048));
  if (try_msd && g_sort_msd_sampled != 2) {
    uint8_t* tables = w + plan.off_msd;
    int overflowed = 0;
    int rc;
    const bool wide = segmented && g_sort_msd_wide != 0 && key_width == 8;
    int ks = 0;   // leading key bits every row shares: the MSD digits start below them
    if (valid_rows == nullptr) {
      const uint64_t* src = reinterpret_cast<const uint64_t*>(vals);
      if (key_width == 8) {
        const int prc = sort_shared_prefix_bits(src, xf, n_valid, reinterpret_cast<unsigned long long*>(tables), st, &ks);
        if (prc != ARX_OK) return prc;
      }
      overflowed = 1;
      rc = ARX_OK;
      if (wide) {
        int rec8 = g_sort_msd_wide_rec8;
        rc = run_msd_sort_wide(src, nullptr, xf, n_valid, rec_a, rec_b, rec_cap, tables, final_dst, g_sort_msd_wide_gap2, ks, st, &overflowed, &rec8);
        if (rc == ARX_OK && overflowed == 2) {
          rc = run_msd_sort_wide(src, nullptr, xf, n_valid, rec_a, rec_b, rec_cap, tables, final_dst, 0, ks, st, &overflowed, &rec8);
        }
      }
      if (rc == ARX_OK && overflowed) {
        rc = segmented ? run_msd_sort_segmented(src, nullptr, xf, n_valid, keys_a, idx_a, keys_b, idx_b, tables,
                                                final_dst, st, &overflowed, ks)
                       : run_msd_sort(src, nullptr, xf, n_valid, keys_a, idx_a, keys_b, idx_b, tables, final_dst, st,
                                      &overflowed, ks);
      }
    } else {
      hipLaunchKernelGGL(sort_prep_kernel, dim3(gprep), dim3(kBlock), 0, st, vals, valid_rows, n_valid, xf, 0,
                         keys_a, idx_a);
      ARX_CHECK_LAUNCH("sort_prep_kernel");
      if (key_width == 8) {
        const int prc = sort_shared_prefix_bits(keys_a, 0, n_valid, reinterpret_cast<unsigned long long*>(tables), st, &ks);
        if (prc != ARX_OK) return prc;
      }
      overflowed = 1;
      rc = ARX_OK;
      // the wide form reads its source twice (level 1, then the histogram reads level-1 output): x = b, y = a is
      // safe because level 2 only starts after level 1 has consumed the source
      if (wide) {
        int rec8 = 0;   // (a prepped source: row ids are not positions in the column)
        rc = run_msd_sort_wide(keys_a, idx_a, 0, n_valid, rec_b, rec_a, rec_cap, tables, final_dst, g_sort_msd_wide_gap2, ks, st, &overflowed, &rec8);
        if (rc == ARX_OK && overflowed == 2) {   // the level-2 records went over the prepped source: rebuild it
          hipLaunchKernelGGL(sort_prep_kernel, dim3(gprep), dim3(kBlock), 0, st, vals, valid_rows, n_valid, xf, 0,
                             keys_a, idx_a);
          ARX_CHECK_LAUNCH("sort_prep_kernel");
          rc = run_msd_sort_wide(keys_a, idx_a, 0, n_valid, rec_b, rec_a, rec_cap, tables, final_dst, 0, ks, st, &overflowed, &rec8);
        }
      }
      if (rc == ARX_OK && overflowed) {
        if (wide) {   // a failed wide attempt may have overwritten the prepped source: rebuild it
          hipLaunchKernelGGL(sort_prep_kernel, dim3(gprep), dim3(kBlock), 0, st, vals, valid_rows, n_valid, xf, 0,
                             keys_a, idx_a);
          ARX_CHECK_LAUNCH("sort_prep_kernel");
        }
        rc = segmented ? run_msd_sort_segmented(keys_a, idx_a, 0, n_valid, keys_b, idx_b, keys_a, idx_a, tables,
                                                final_dst, st, &overflowed, ks)
                       : run_msd_sort(keys_a, idx_a, 0, n_valid, keys_b, idx_b, keys_a, idx_a, tables, final_dst, st,
                                      &overflowed, ks);
      }
    }
    if (rc != ARX_OK) return rc;
    if (!overflowed) return ARX_OK;
  }
  // skewed top bits: bucket boundaries from a sorted sample instead of equal-width prefixes
  // (sort_msd_sampled = 2 forces this form first, for the tests)
  // (32-bit keys: their 4 LSD passes measured faster than this form, 5.2 vs 7.4 ms at 2^27 rows)
  if ((try_msd || g_sort_msd_sampled == 2) && g_sort_msd_sampled != 0 && (key_width == 8 || g_sort_msd_sampled == 2)) {
    uint8_t* tables = w + plan.off_msd;
    int overflowed = 0;
    int rc = ARX_OK;
    if (n_valid >= (g_sort_msd_sampled == 2 ? 1024 : (int64_t(1) << 18)) && n_valid <= (int64_t(3) << 26)) {
      if (valid_rows == nullptr) {
        rc = run_msd_sort_sampled(reinterpret_cast<const uint64_t*>(vals), nullptr, xf, n_valid, keys_a, idx_a,
                                  keys_b, idx_b, tables, hist, totals, final_dst, st, &overflowed);
      } else {
        hipLaunchKernelGGL(sort_prep_kernel, dim3(gprep), dim3(kBlock), 0, st, vals, valid_rows, n_valid, xf, 0,
                           keys_a, idx_a);
        ARX_CHECK_LAUNCH("sort_prep_kernel");
        // source = (keys_a, idx_a); x = (keys_b, idx_b) also hosts the sample; y may reuse the source
        rc = run_msd_sort_sampled(keys_a, idx_a, 0, n_valid, keys_b, idx_b, keys_a, idx_a, tables, hist, totals,
                                  final_dst, st, &overflowed);
      }
      if (rc != ARX_OK) return rc;
      if (!overflowed) return ARX_OK;
    }
  }

  // ---- LSD passes over (key, row id) pairs.  With no null-likes the first pass reads the caller's
  // column directly (transform applied on load, row id = position); otherwise a prep pass gathers
  // the sortable rows.  32-bit keys live in the top half of the transformed key: 4 passes.
  plan = make_plan(n_valid);
  const int raw_first = (valid_rows == nullptr && g_sort_fuse_prep) ? xf : 0;
  if (!raw_first) {
    hipLaunchKernelGGL(sort_prep_kernel, dim3(gprep), dim3(kBlock), 0, st, vals, valid_rows, n_valid, xf, 0,
                       keys_a, idx_a);
    ARX_CHECK_LAUNCH("sort_prep_kernel");
  }
  const int64_t chunk_keys = plan.chunk_tiles * kSortTile;
  const unsigned nch = static_cast<unsigned>(plan.nchunks);
  const uint64_t* kin = raw_first ? reinterpret_cast<const uint64_t*>(vals) : keys_a;
  uint64_t* kout = keys_b;
  const uint32_t* iin = idx_a;
  uint32_t* iout = idx_b;
  const int first_pass = key_width == 4 ? 4 : 0;
  for (int pass = first_pass; pass < 8; ++pass) {
    const int shift = pass * 8;
    const int raw = pass == first_pass ? raw_first : 0;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nch), dim3(kBlock), 0, st, kin, n_valid, shift,
                       chunk_keys, plan.nchunks, hist, raw);
    ARX_CHECK_LAUNCH("radix_hist_kernel");
    hipLaunchKernelGGL(radix_digit_totals_kernel, dim3(kDigits), dim3(64), 0, st, hist, plan.nchunks,
                       totals);
    ARX_CHECK_LAUNCH("radix_digit_totals_kernel");
    hipLaunchKernelGGL(radix_scan_kernel, dim3(kDigits), dim3(64), 0, st, hist, plan.nchunks, totals);
    ARX_CHECK_LAUNCH("radix_scan_kernel");
    if (raw) {
      hipLaunchKernelGGL((radix_scatter_kernel<true>), dim3(nch), dim3(kBlock), 0, st, kin, iin, n_valid, shift,
                         plan.chunk_tiles, plan.nchunks, hist, kout, iout, final_dst, pass == 7 ? 1 : 0, raw);
    } else {
      hipLaunchKernelGGL((radix_scatter_kernel<false>), dim3(nch), dim3(kBlock), 0, st, kin, iin, n_valid,
                         shift, plan.chunk_tiles, plan.nchunks, hist, kout, iout, final_dst, pass == 7 ? 1 : 0,
                         0);
    }
    ARX_CHECK_LAUNCH("radix_scatter_kernel");
    // ping-pong between the two scratch pairs (the caller's column is only ever read)
    uint64_t* knext = (kout == keys_b) ? keys_a : keys_b;
    uint32_t* inext = (iout == idx_b) ? idx_a : idx_b;
    kin = kout;
    iin = iout;
    kout = knext;
    iout = inext;
  }
  return ARX_OK;
}

int arx_sort_indices_64(const ArxSpan* values, int is_signed, int order, int null_placement,
                        void* ws, size_t ws_bytes, uint64_t* out_indices, void* stream) {
  return arx_sort_indices(values, is_signed ? ARX_KEY_INT64 : ARX_KEY_UINT64, order, null_placement, ws,
                          ws_bytes, out_indices, stream);
}

static int sort_window_ok(const ArxSortKeyWindow* window, uint64_t* base, int* shift) {
  *base = 0;
  *shift = 0;
  if (window == nullptr) return ARX_OK;
  if (window->shift < 0 || window->shift > 63) {
    set_error("sort key window: shift %d outside [0, 63]", window->shift);
    return ARX_INVALID;
  }
  *base = window->key_min;
  *shift = window->shift;
  return ARX_OK;
}

int arx_sort_key_range(const ArxSpan* values, int is_signed, int order, uint64_t* out_range, void* stream) {
  return arx_sort_key_range_sampled(values, is_signed, order, 0, out_range, stream);
}

int arx_sort_key_range_sampled(const ArxSpan* values, int is_signed, int order, int sample_shift, uint64_t* out_range, void* stream) {
  if (values == nullptr || out_range == nullptr || sample_shift < 0 || sample_shift > 8) {
    set_error("bad arguments to arx_sort_key_range");
    return ARX_INVALID;
  }
  const int64_t n = values->length;
  if (n == 0) return ARX_OK;
  if (values->data == nullptr) {
    set_error("values buffer is NULL");
    return ARX_INVALID;
  }
  const uint64_t* vals = static_cast<const uint64_t*>(values->data) + values->offset;
  const Bits vb = make_bits(values->null_count != 0 ? values->validity : nullptr, values->offset, n);
  const unsigned g = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(n >> sample_shift, kBlock * 16), 2048)));
  hipLaunchKernelGGL(sort_key_range_kernel, dim3(g), dim3(kBlock), 0, as_stream(stream), vals, vb, n, is_signed,
                     order == ARX_SORT_DESCENDING, reinterpret_cast<unsigned long long*>(out_range), sample_shift);
  ARX_CHECK_LAUNCH("sort_key_range_kernel");
  return ARX_OK;
}

int arx_sort_key_histogram(const ArxSpan* values, int is_signed, int order, int bits,
                           uint64_t* out_hist, void* stream) {
  return arx_sort_key_histogram_window(values, is_signed, order, bits, nullptr, out_hist, stream);
}

int arx_sort_key_histogram_window(const ArxSpan* values, int is_signed, int order, int bits,
                                  const ArxSortKeyWindow* window, uint64_t* out_hist, void* stream) {
  return arx_sort_key_histogram_window_sampled(values, is_signed, order, bits, window, 0, out_hist, stream);
}

int arx_sort_key_histogram_window_sampled(const ArxSpan* values, int is_signed, int order, int bits,
                                          const ArxSortKeyWindow* window, int sample_shift, uint64_t* out_hist, void* stream) {
  if (values == nullptr || out_hist == nullptr || bits < 1 || bits > 12 || sample_shift < 0 || sample_shift > 8) {
    set_error("bad arguments to arx_sort_key_histogram (bits in [1,12])");
    return ARX_INVALID;
  }
  uint64_t base;
  int shift;
  if (sort_window_ok(window, &base, &shift) != ARX_OK) return ARX_INVALID;
  const int64_t n = values->length;
  if (n == 0) return ARX_OK;
  if (values->data == nullptr) {
    set_error("values buffer is NULL");
    return ARX_INVALID;
  }
  const uint64_t* vals = static_cast<const uint64_t*>(values->data) + values->offset;
  const Bits vb = make_bits(values->null_count != 0 ? values->validity : nullptr, values->offset, n);
  const unsigned g = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(n >> sample_shift, kBlock * 16), 2048)));
  hipLaunchKernelGGL(sort_key_hist_kernel, dim3(g), dim3(kBlock), 0, as_stream(stream), vals, vb, n,
                     is_signed, order == ARX_SORT_DESCENDING, bits, base, shift,
                     reinterpret_cast<unsigned long long*>(out_hist), sample_shift);
  ARX_CHECK_LAUNCH("sort_key_hist_kernel");
  return ARX_OK;
}

// Shared by the two forms below: the non-null rows in destination-major order (STABLE inside a destination)
// as row ids in *rows_out (workspace memory), per-destination counts (device int64) in out_counts.
static int sort_partition_rows(const ArxSpan* values, int is_signed, int order, int bits,
                               const ArxSortKeyWindow* window,
                               const uint32_t* splitter_bins, int num_parts, void* ws, size_t ws_bytes,
                               int64_t* out_counts, int64_t* out_num_valid, hipStream_t st,
                               const uint32_t** rows_out, SortPlan* plan_out) {
  if (values == nullptr || num_parts < 1 || num_parts > kDigits || bits < 1 || bits > 12 ||
      out_counts == nullptr || out_num_valid == nullptr || (num_parts > 1 && splitter_bins == nullptr)) {
    set_error("bad arguments to the sort partition");
    return ARX_INVALID;
  }
  uint64_t base;
  int shift;
  if (sort_window_ok(window, &base, &shift) != ARX_OK) return ARX_INVALID;
  const int64_t len = values->length;
  if (len > static_cast<int64_t>(UINT32_MAX)) {
    set_error("sort partition: more than UINT32_MAX rows per shard is not implemented");
    return ARX_NOT_IMPLEMENTED;
  }
  ARX_HIP(hipMemsetAsync(out_counts, 0, static_cast<size_t>(num_parts) * 8, st));
  *out_num_valid = 0;
  *rows_out = nullptr;
  if (len == 0) return ARX_OK;
  SortPlan plan = make_plan(len);
  *plan_out = plan;
  if (ws == nullptr || ws_bytes < plan.total || (reinterpret_cast<uint64_t>(ws) & 255) != 0 ||
      values->data == nullptr) {
    set_error("sort partition: NULL buffer or workspace too small / unaligned");
    return ARX_INVALID;
  }
  uint8_t* w = static_cast<uint8_t*>(ws);
  uint64_t* keys_a = reinterpret_cast<uint64_t*>(w + plan.off_keys_a);
  uint64_t* keys_b = reinterpret_cast<uint64_t*>(w + plan.off_keys_b);
  uint32_t* idx_a = reinterpret_cast<uint32_t*>(w + plan.off_idx_a);
  uint32_t* idx_b = reinterpret_cast<uint32_t*>(w + plan.off_idx_b);
  uint32_t* hist = reinterpret_cast<uint32_t*>(w + plan.off_hist);
  uint32_t* totals = reinterpret_cast<uint32_t*>(w + plan.off_totals);
  uint32_t* split = reinterpret_cast<uint32_t*>(w + plan.off_split);
  uint32_t* rows = reinterpret_cast<uint32_t*>(w + plan.off_rows);
  void* sel_ws = w + plan.off_sel_ws;
  const size_t sel_ws_bytes = plan.total - plan.off_sel_ws;
  const uint64_t* vals = static_cast<const uint64_t*>(values->data) + values->offset;
  const bool has_nulls = values->null_count != 0 && values->validity != nullptr;
  int64_t n_valid = len;
  const uint32_t* valid_rows = nullptr;
  if (has_nulls) {
    int64_t got = 0;
    const int rc = selection_bit_positions(values->validity, values->offset, len, /*invert=*/false, sel_ws,
                                           sel_ws_bytes, rows, &got, st);
    if (rc != ARX_OK) return rc;
    n_valid = got;
    if (n_valid < len) valid_rows = rows;
  }
  *out_num_valid = n_valid;
  if (n_valid == 0) return ARX_OK;
  if (num_parts > 1) {
    ARX_HIP(hipMemcpyAsync(split, splitter_bins, static_cast<size_t>(num_parts - 1) * 4,
                           hipMemcpyHostToDevice, st));
    ARX_HIP(hipStreamSynchronize(st));  // the caller's host array may go away after we return
  }
  plan = make_plan(n_valid);
  const unsigned g = static_cast<unsigned>(std::min<int64_t>(ceil_div(n_valid, kBlock), 2048));
  hipLaunchKernelGGL(sort_dest_prep_kernel, dim3(g), dim3(kBlock), 0, st, vals, valid_rows, n_valid,
                     is_signed, order == ARX_SORT_DESCENDING, bits, base, shift, split, num_parts - 1, keys_a, idx_a);
  ARX_CHECK_LAUNCH("sort_dest_prep_kernel");
  // one stable radix pass on the destination digit carries the row ids into destination-major order
  const int64_t chunk_keys = plan.chunk_tiles * kSortTile;
  const unsigned nch = static_cast<unsigned>(plan.nchunks);
  hipLaunchKernelGGL(radix_hist_kernel, dim3(nch), dim3(kBlock), 0, st, keys_a, n_valid, 0, chunk_keys,
                     plan.nchunks, hist, 0);
  hipLaunchKernelGGL(radix_digit_totals_kernel, dim3(kDigits), dim3(64), 0, st, hist, plan.nchunks, totals);
  hipLaunchKernelGGL(widen_counts_kernel, dim3(1), dim3(kDigits), 0, st, totals, num_parts, out_counts);
  hipLaunchKernelGGL(radix_scan_kernel, dim3(kDigits), dim3(64), 0, st, hist, plan.nchunks, totals);
  hipLaunchKernelGGL((radix_scatter_kernel<false>), dim3(nch), dim3(kBlock), 0, st, keys_a, idx_a, n_valid, 0,
                     plan.chunk_tiles, plan.nchunks, hist, keys_b, idx_b, static_cast<uint64_t*>(nullptr), 0, 0);
  ARX_CHECK_LAUNCH("radix partition pass");
  *rows_out = idx_b;
  return ARX_OK;
}

int arx_sort_partition_by_bins(const ArxSpan* values, int is_signed, int order, int bits,
                               const uint32_t* splitter_bins, int num_parts, void* ws, size_t ws_bytes,
                               uint64_t* out_keys, uint32_t* out_rows, int64_t* out_counts,
                               int64_t* out_num_valid, void* stream) {
  hipStream_t st = as_stream(stream);
  const uint32_t* rows = nullptr;
  SortPlan plan{};
  const int rc = sort_partition_rows(values, is_signed, order, bits, nullptr, splitter_bins, num_parts, ws, ws_bytes,
                                     out_counts, out_num_valid, st, &rows, &plan);
  if (rc != ARX_OK || rows == nullptr) return rc;
  if (out_keys == nullptr || out_rows == nullptr) {
    set_error("arx_sort_partition_by_bins: NULL output buffer");
    return ARX_INVALID;
  }
  const int64_t n_valid = *out_num_valid;
  const uint64_t* vals = static_cast<const uint64_t*>(values->data) + values->offset;
  const unsigned g = static_cast<unsigned>(std::min<int64_t>(ceil_div(n_valid, kBlock), 2048));
  // transformed keys and row ids in destination-major, row-order-preserving order
  hipLaunchKernelGGL(sort_prep_kernel, dim3(g), dim3(kBlock), 0, st, static_cast<const void*>(vals), rows, n_valid,
                     make_xf(is_signed ? ARX_KEY_INT64 : ARX_KEY_UINT64, order == ARX_SORT_DESCENDING), 0, out_keys,
                     out_rows);
  ARX_CHECK_LAUNCH("sort_prep_kernel");
  return ARX_OK;
}

int arx_sort_partition_records(const ArxSpan* values, int is_signed, int order, int null_placement, int bits,
                               const uint32_t* splitter_bins, int num_parts, void* ws, size_t ws_bytes,
                               ArxSortRecord* out_records, int64_t* out_counts, int64_t* out_num_valid,
                               void* stream) {
  return arx_sort_partition_records_window(values, is_signed, order, null_placement, bits, nullptr, splitter_bins,
                                           num_parts, ws, ws_bytes, out_records, out_counts, out_num_valid, stream);
}

int arx_sort_partition_records_window(const ArxSpan* values, int is_signed, int order, int null_placement, int bits,
                                      const ArxSortKeyWindow* window, const uint32_t* splitter_bins, int num_parts,
                                      void* ws, size_t ws_bytes, ArxSortRecord* out_records, int64_t* out_counts,
                                      int64_t* out_num_valid, void* stream) {
  if (null_placement != ARX_NULLS_AT_START && null_placement != ARX_NULLS_AT_END) {
    set_error("bad null placement %d", null_placement);
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  const uint32_t* rows = nullptr;
  SortPlan plan{};
  const int rc = sort_partition_rows(values, is_signed, order, bits, window, splitter_bins, num_parts, ws, ws_bytes,
                                     out_counts, out_num_valid, st, &rows, &plan);
  if (rc != ARX_OK) return rc;
  const int64_t len = values->length;
  if (len == 0) return ARX_OK;
  if (out_records == nullptr) {
    set_error("arx_sort_partition_records: NULL output buffer");
    return ARX_INVALID;
  }
  const int64_t n_valid = *out_num_valid;
  const int64_t n_null = len - n_valid;
  const uint64_t* vals = static_cast<const uint64_t*>(values->data) + values->offset;
  ArxSortRecord* valid_dst = null_placement == ARX_NULLS_AT_START ? out_records + n_null : out_records;
  ArxSortRecord* null_dst = null_placement == ARX_NULLS_AT_START ? out_records : out_records + n_valid;
  if (n_valid > 0) {
    const unsigned g = static_cast<unsigned>(std::min<int64_t>(ceil_div(n_valid, kBlock), 2048));
    hipLaunchKernelGGL(sort_pack_records_kernel, dim3(g), dim3(kBlock), 0, st, vals, rows, n_valid, is_signed,
                       order == ARX_SORT_DESCENDING ? 1 : 0, 0, valid_dst);
    ARX_CHECK_LAUNCH("sort_pack_records_kernel");
  }
  if (n_null > 0) {
    // null rows keep row order (PartitionNullsOnly) and travel as row numbers only
    uint8_t* w = static_cast<uint8_t*>(ws);
    uint32_t* nrows = reinterpret_cast<uint32_t*>(w + plan.off_rows);
    void* sel_ws = w + plan.off_sel_ws;
    int64_t got = 0;
    const int rc2 = selection_bit_positions(values->validity, values->offset, len, /*invert=*/true, sel_ws,
                                            plan.total - plan.off_sel_ws, nrows, &got, st);
    if (rc2 != ARX_OK) return rc2;
    const unsigned g = static_cast<unsigned>(std::min<int64_t>(ceil_div(n_null, kBlock), 2048));
    hipLaunchKernelGGL(sort_pack_records_kernel, dim3(g), dim3(kBlock), 0, st, vals, nrows, n_null, 0, 0, 1, null_dst);
    ARX_CHECK_LAUNCH("sort_pack_records_kernel");
  }
  return ARX_OK;
}

int arx_sort_unpack_records(const ArxSortRecord* records, int64_t num_records, const int64_t* block_meta,
                            int num_blocks, int nulls_first, uint64_t* out_keys, int64_t* out_rows,
                            int64_t* out_null_rows, void* stream) {
  if (num_records < 0 || num_blocks < 1 || num_blocks > kSortMaxBlocks || block_meta == nullptr) {
    set_error("bad arguments to arx_sort_unpack_records");
    return ARX_INVALID;
  }
  if (num_records == 0) return ARX_OK;
  if (records == nullptr || out_keys == nullptr || out_rows == nullptr || out_null_rows == nullptr) {
    set_error("arx_sort_unpack_records: NULL buffer");
    return ARX_INVALID;
  }
  const unsigned g = static_cast<unsigned>(std::min<int64_t>(ceil_div(num_records, kBlock), 2048));
  hipLaunchKernelGGL(sort_unpack_records_kernel, dim3(g), dim3(kBlock), 0, as_stream(stream), records, num_records,
                     block_meta, num_blocks, nulls_first, out_keys, out_rows, out_null_rows);
  ARX_CHECK_LAUNCH("sort_unpack_records_kernel");
  return ARX_OK;
}

int arx_sort_partition_records_global(const ArxSpan* values, int is_signed, int order, int bits, const ArxSortKeyWindow* window,
                                      const uint32_t* splitter_bins, int num_parts, uint32_t row_base, void* ws, size_t ws_bytes,
                                      ArxSortRecord* out_records, int64_t* out_counts, void* stream) {
  if (values == nullptr || num_parts < 1 || num_parts > kSortPartMax || bits < 1 || bits > 12 || out_counts == nullptr ||
      (num_parts > 1 && splitter_bins == nullptr)) {
    set_error("bad arguments to arx_sort_partition_records_global (1 to %d destinations)", kSortPartMax);
    return ARX_INVALID;
  }
  if (values->null_count != 0 && values->validity != nullptr) {
    set_error("arx_sort_partition_records_global: shards with nulls go through arx_sort_partition_records_window");
    return ARX_NOT_IMPLEMENTED;
  }
  uint64_t base;
  int shift;
  if (sort_window_ok(window, &base, &shift) != ARX_OK) return ARX_INVALID;
  const int64_t len = values->length;
  if (len < 0 || static_cast<int64_t>(row_base) + len > (int64_t(1) << 32)) {
    set_error("arx_sort_partition_records_global: global row numbers must fit 32 bits");
    return ARX_NOT_IMPLEMENTED;
  }
  hipStream_t st = as_stream(stream);
  ARX_HIP(hipMemsetAsync(out_counts, 0, static_cast<size_t>(num_parts) * 8, st));
  if (len == 0) return ARX_OK;
  if (values->data == nullptr || out_records == nullptr || ws == nullptr || ws_bytes < 1024 || (reinterpret_cast<uint64_t>(ws) & 7) != 0) {
    set_error("arx_sort_partition_records_global: NULL buffer or no scratch (1 KB, 8-byte aligned)");
    return ARX_INVALID;
  }
  uint32_t* split = static_cast<uint32_t*>(ws);                                                      // [64]
  unsigned long long* cursor = reinterpret_cast<unsigned long long*>(static_cast<uint8_t*>(ws) + 256);   // [64]
  if (num_parts > 1) {
    ARX_HIP(hipMemcpyAsync(split, splitter_bins, static_cast<size_t>(num_parts - 1) * 4, hipMemcpyHostToDevice, st));
    ARX_HIP(hipStreamSynchronize(st));   // the caller's host array may go away after we return
  }
  const uint64_t* vals = static_cast<const uint64_t*>(values->data) + values->offset;
  const int desc = order == ARX_SORT_DESCENDING ? 1 : 0;
  const unsigned gc = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(len, kSortPartThreads * 8), 1024)));
  hipLaunchKernelGGL(sort_part_count_kernel, dim3(gc), dim3(kSortPartThreads), 0, st, vals, len, is_signed, desc, bits, base, shift,
                     split, num_parts, reinterpret_cast<unsigned long long*>(out_counts));
  ARX_CHECK_LAUNCH("sort_part_count_kernel");
  hipLaunchKernelGGL(sort_part_scan_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<const unsigned long long*>(out_counts), num_parts, cursor);
  ARX_CHECK_LAUNCH("sort_part_scan_kernel");
  const unsigned gs = static_cast<unsigned>(ceil_div(len, kSortPartThreads * kSortPartRows));
  hipLaunchKernelGGL(sort_part_scatter_kernel, dim3(gs), dim3(kSortPartThreads), 0, st, vals, len, is_signed, desc, bits, base, shift,
                     split, num_parts, row_base, cursor, out_records);
  ARX_CHECK_LAUNCH("sort_part_scatter_kernel");
  return ARX_OK;
}

// records by (key, row) ascending -> their rows, widened to 64 bits.  The flow of arx_sort_indices over a PREPPED source
// (keys + row ids beside the column): wide form > segmented / hybrid MSD > sampled splitters > LSD passes.
int arx_sort_records(const ArxSortRecord* records, int64_t num_records, void* ws, size_t ws_bytes, uint64_t* out_rows, void* stream) {
  if (num_records < 0) {
    set_error("arx_sort_records: negative length");
    return ARX_INVALID;
  }
  if (num_records == 0) return ARX_OK;
  if (num_records > static_cast<int64_t>(UINT32_MAX)) {
    set_error("arx_sort_records: more than UINT32_MAX records is not implemented");
    return ARX_NOT_IMPLEMENTED;
  }
  SortPlan plan = make_plan(num_records);
  if (records == nullptr || out_rows == nullptr || ws == nullptr || ws_bytes < plan.total || (reinterpret_cast<uint64_t>(ws) & 255) != 0) {
    set_error("arx_sort_records: NULL buffer, or the workspace is too small (arx_sort_indices_workspace_bytes) / not 256-byte aligned");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  const int64_t n = num_records;
  uint8_t* w = static_cast<uint8_t*>(ws);
  uint64_t* keys_a = reinterpret_cast<uint64_t*>(w + plan.off_keys_a);
  uint64_t* keys_b = reinterpret_cast<uint64_t*>(w + plan.off_keys_b);
  uint32_t* idx_a = reinterpret_cast<uint32_t*>(w + plan.off_idx_a);
  uint32_t* idx_b = reinterpret_cast<uint32_t*>(w + plan.off_idx_b);
  MsdRec* rec_a = reinterpret_cast<MsdRec*>(w + plan.off_keys_a);
  MsdRec* rec_b = reinterpret_cast<MsdRec*>(w + plan.off_keys_b);
  const int64_t rec_cap = n + msdw_slack_rows(n);
  uint32_t* hist = reinterpret_cast<uint32_t*>(w + plan.off_hist);
  uint32_t* totals = reinterpret_cast<uint32_t*>(w + plan.off_totals);
  uint8_t* tables = w + plan.off_msd;
  const unsigned gprep = static_cast<unsigned>(std::min<int64_t>(ceil_div(n, kBlock), 2048));
  auto prep = [&]() -> int {   // (a failed attempt may have written over the prepped source: the records are untouched)
    hipLaunchKernelGGL(sort_split_records_kernel, dim3(gprep), dim3(kBlock), 0, st, records, n, keys_a, idx_a);
    ARX_CHECK_LAUNCH("sort_split_records_kernel");
    return ARX_OK;
  };
  int rc = ARX_OK;
  const bool try_msd = g_sort_msd != 0 && n < (int64_t(1) << 32) - kMsdTile && (g_sort_msd == 1 ? n >= 256 : n >= g_sort_msd_min_rows);
  const bool segmented = n > g_sort_msd_segment_rows;
  const bool wide = try_msd && g_sort_msd_sampled != 2 && segmented && g_sort_msd_wide != 0;
  // the wide form reads the records where they lie (level 1 takes 12-byte records as level 2 does): no split into a key
  // and a row array first — every other form starts from the split arrays (knob sort_records_in_place = 0: the split always)
  const MsdRec* in_place = wide && g_sort_records_in_place ? reinterpret_cast<const MsdRec*>(records) : nullptr;
  if (in_place == nullptr) {
    rc = prep();
    if (rc != ARX_OK) return rc;
  }
  if (try_msd && g_sort_msd_sampled != 2) {
    int ks = 0, overflowed = 1;
    rc = sort_shared_prefix_bits(in_place != nullptr ? nullptr : keys_a, 0, n, reinterpret_cast<unsigned long long*>(tables), st, &ks, in_place);
    if (rc != ARX_OK) return rc;
    if (wide) {
      int rec8 = 0;   // (row ids are not positions in a column: full records)
      const uint64_t* wk = in_place != nullptr ? nullptr : keys_a;
      const uint32_t* wi = in_place != nullptr ? nullptr : idx_a;
      rc = run_msd_sort_wide(wk, wi, 0, n, rec_b, rec_a, rec_cap, tables, out_rows, g_sort_msd_wide_gap2, ks, st, &overflowed, &rec8, in_place);
      if (rc == ARX_OK && overflowed == 2) {
        if (in_place == nullptr) rc = prep();
        if (rc == ARX_OK) rc = run_msd_sort_wide(wk, wi, 0, n, rec_b, rec_a, rec_cap, tables, out_rows, 0, ks, st, &overflowed, &rec8, in_place);
      }
      if (rc == ARX_OK && overflowed) rc = prep();
    }
    if (rc == ARX_OK && overflowed) {
      rc = segmented ? run_msd_sort_segmented(keys_a, idx_a, 0, n, keys_b, idx_b, keys_a, idx_a, tables, out_rows, st, &overflowed, ks)
                     : run_msd_sort(keys_a, idx_a, 0, n, keys_b, idx_b, keys_a, idx_a, tables, out_rows, st, &overflowed, ks);
    }
    if (rc != ARX_OK) return rc;
    if (!overflowed) return ARX_OK;
    rc = prep();
    if (rc != ARX_OK) return rc;
  }
  if ((try_msd || g_sort_msd_sampled == 2) && g_sort_msd_sampled != 0 && n >= (g_sort_msd_sampled == 2 ? 1024 : (int64_t(1) << 18)) &&
      n <= (int64_t(3) << 26)) {
    int overflowed = 0;
    rc = run_msd_sort_sampled(keys_a, idx_a, 0, n, keys_b, idx_b, keys_a, idx_a, tables, hist, totals, out_rows, st, &overflowed);
    if (rc != ARX_OK) return rc;
    if (!overflowed) return ARX_OK;
    rc = prep();
    if (rc != ARX_OK) return rc;
  }
  // LSD passes.  They are stable on the KEY alone and the records arrive in no particular order, so the rows are put in row
  // order first: four passes over {row, position}, one gather of the records by the positions, then the eight key passes.
  const int64_t chunk_keys = plan.chunk_tiles * kSortTile;
  const unsigned nch = static_cast<unsigned>(plan.nchunks);
  const uint64_t* kin = keys_a;
  uint64_t* kout = keys_b;
  const uint32_t* iin = idx_a;
  uint32_t* iout = idx_b;
  hipLaunchKernelGGL(sort_records_rows_kernel, dim3(gprep), dim3(kBlock), 0, st, records, n, keys_a, idx_a);
  ARX_CHECK_LAUNCH("sort_records_rows_kernel");
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nch), dim3(kBlock), 0, st, kin, n, pass * 8, chunk_keys, plan.nchunks, hist, 0);
    hipLaunchKernelGGL(radix_digit_totals_kernel, dim3(kDigits), dim3(64), 0, st, hist, plan.nchunks, totals);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(kDigits), dim3(64), 0, st, hist, plan.nchunks, totals);
    hipLaunchKernelGGL((radix_scatter_kernel<false>), dim3(nch), dim3(kBlock), 0, st, kin, iin, n, pass * 8, plan.chunk_tiles, plan.nchunks,
                       hist, kout, iout, static_cast<uint64_t*>(nullptr), 0, 0);
    ARX_CHECK_LAUNCH("radix pass over the rows");
    uint64_t* knext = (kout == keys_b) ? keys_a : keys_b;
    uint32_t* inext = (iout == idx_b) ? idx_a : idx_b;
    kin = kout;
    iin = iout;
    kout = knext;
    iout = inext;
  }
  // (four passes: the positions in row order are back in idx_a) -> the records in row order into (keys_b, idx_b)
  hipLaunchKernelGGL(sort_records_gather_kernel, dim3(gprep), dim3(kBlock), 0, st, records, idx_a, n, keys_b, idx_b);
  ARX_CHECK_LAUNCH("sort_records_gather_kernel");
  kin = keys_b;
  iin = idx_b;
  kout = keys_a;
  iout = idx_a;
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = pass * 8;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nch), dim3(kBlock), 0, st, kin, n, shift, chunk_keys, plan.nchunks, hist, 0);
    ARX_CHECK_LAUNCH("radix_hist_kernel");
    hipLaunchKernelGGL(radix_digit_totals_kernel, dim3(kDigits), dim3(64), 0, st, hist, plan.nchunks, totals);
    ARX_CHECK_LAUNCH("radix_digit_totals_kernel");
    hipLaunchKernelGGL(radix_scan_kernel, dim3(kDigits), dim3(64), 0, st, hist, plan.nchunks, totals);
    ARX_CHECK_LAUNCH("radix_scan_kernel");
    hipLaunchKernelGGL((radix_scatter_kernel<false>), dim3(nch), dim3(kBlock), 0, st, kin, iin, n, shift, plan.chunk_tiles, plan.nchunks,
                       hist, kout, iout, out_rows, pass == 7 ? 1 : 0, 0);
    ARX_CHECK_LAUNCH("radix_scatter_kernel");
    uint64_t* knext = (kout == keys_b) ? keys_a : keys_b;
    uint32_t* inext = (iout == idx_b) ? idx_a : idx_b;
    kin = kout;
    iin = iout;
    kout = knext;
    iout = inext;
  }
  return ARX_OK;
}

int arx_bitmap_to_indices(const void* bits, int64_t bit_offset, int64_t length, int invert, void* ws,
                          size_t ws_bytes, uint32_t* out_indices, int64_t* out_count, void* stream) {
  if (out_count == nullptr || length < 0 || bit_offset < 0) {
    set_error("bad arguments to arx_bitmap_to_indices");
    return ARX_INVALID;
  }
  *out_count = 0;
  if (length == 0) return ARX_OK;
  if (bits == nullptr || ws == nullptr || out_indices == nullptr) {
    set_error("NULL buffer passed to arx_bitmap_to_indices");
    return ARX_INVALID;
  }
  if (length > static_cast<int64_t>(UINT32_MAX)) {
    set_error("arx_bitmap_to_indices: more than UINT32_MAX bits is not implemented");
    return ARX_NOT_IMPLEMENTED;
  }
  return selection_bit_positions(bits, bit_offset, length, invert != 0, ws, ws_bytes, out_indices,
                                 out_count, as_stream(stream));
}

}  // extern "C"
