// rank / rank_quantile from a sorted order — the second half of RankMetaFunction (cpp/src/arrow/compute/kernels/
// vector_rank.cc:36-245): the reference sorts the indices (the array sorter, nulls and NaNs partitioned to one end),
// marks every index whose value equals the one before it (MarkDuplicates, :40-72 — all NaNs are one run, all nulls are
// one run) and walks the marked order once per tiebreaker (OrdinalRanker::CreateRankings, :203-263; quantile ranks
// BaseQuantileRanker, :163-196).  Here the sorted order comes from arx_sort_indices and the walk is three kernels over
// tiles of 2048 positions:
//   rank_mark_kernel   gathers the keys in sorted order, one bit per position = "a run starts here", plus per tile the
//                      number of run starts and the first / last position that has one
//   rank_scan_kernel   one workgroup over the tiles: run starts before each tile (dense ranks), the last run start
//                      before it and the first one after it
//   rank_emit_kernel   per position its run's start and end -> the rank, scattered to out[sorted[position]]
// Algorithmic bytes per row: 8 (sorted index) + 8 (key gather: a 128-byte line for 8 wanted bytes, as every take) +
// 8 (sorted index again) + 8 (rank scatter, same remark) — HBM-bound on the two random accesses.
#include "arx_common.h"
#include <math.h>

#include <string.h>

#include <algorithm>

namespace arx {

constexpr int kRankThreads = 256;
constexpr int kRankPer = 8;                           // positions per thread: one byte of run-start flags
constexpr int kRankTile = kRankThreads * kRankPer;    // 2048
constexpr int kRankScanThreads = 1024;
constexpr int kRankScanPer = 16;                      // tiles per thread and round of the scan

struct RankArgs {
  const void* values;     // the column's data buffer
  Bits valid;             // its validity (NULL: all valid), logical bit i = row i
  int64_t offset;         // of the column's first row in `values`
  int key_type;           // ARX_KEY_*
  const uint64_t* sorted; // n rows: the column's sorted order
  int64_t n;
  int64_t tiles;
  int tiebreaker;         // ARX_RANK_*
  uint8_t* flags;         // tiles * 256 bytes: bit j of byte t = a run starts at position 8 t + j
  uint32_t* tile_count;   // run starts in the tile
  int64_t* tile_first;    // first position of the tile with a run start (n: none)
  int64_t* tile_last;     // last one (-1: none)
  uint64_t* dense_base;   // run starts before the tile
  int64_t* carry_start;   // last run start before the tile (tile 0: 0)
  int64_t* carry_end;     // first run start after the tile (n: none)
  void* out;              // uint64[n] ranks, or double[n] quantile ranks
};

struct RankKey {
  uint64_t bits;
  uint32_t cls;   // 0 value, 1 NaN, 2 null
};

__device__ __forceinline__ RankKey rank_key_of(const RankArgs& a, uint64_t row) {
  RankKey k{0, 0};
  if (a.valid.base != nullptr) {
    const uint64_t bit = static_cast<uint64_t>(a.valid.shift) + row;
    if (((a.valid.base[bit >> 6] >> (bit & 63)) & 1ull) == 0) {
      k.cls = 2;
      return k;
    }
  }
  const int64_t at = a.offset + static_cast<int64_t>(row);
  switch (a.key_type) {
    case ARX_KEY_UINT64:
    case ARX_KEY_INT64: k.bits = static_cast<const uint64_t*>(a.values)[at]; break;
    case ARX_KEY_UINT32:
    case ARX_KEY_INT32: k.bits = static_cast<const uint32_t*>(a.values)[at]; break;
    case ARX_KEY_FLOAT64: {
      const uint64_t b = static_cast<const uint64_t*>(a.values)[at];
      if ((b << 1) > 0xFFE0000000000000ull) k.cls = 1;   // NaN: every NaN is the same value to the ranker
      else k.bits = (b << 1) == 0 ? 0 : b;               // -0.0 == 0.0
      break;
    }
    default: {
      const uint32_t b = static_cast<const uint32_t*>(a.values)[at];
      if ((b << 1) > 0xFF000000u) k.cls = 1;
      else k.bits = (b << 1) == 0 ? 0u : b;
      break;
    }
  }
  return k;
}

__global__ __launch_bounds__(kRankThreads) void rank_mark_kernel(RankArgs a) {
  __shared__ uint64_t last_bits[kRankThreads];
  __shared__ uint32_t last_cls[kRankThreads];
  __shared__ uint32_t w_count[kRankThreads / 64];
  __shared__ int64_t w_first[kRankThreads / 64], w_last[kRankThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tile = blockIdx.x;
  const int64_t p0 = tile * kRankTile + static_cast<int64_t>(tid) * kRankPer;
  uint64_t rows[kRankPer];
#pragma unroll
  for (int j = 0; j < kRankPer; ++j) rows[j] = p0 + j < a.n ? a.sorted[p0 + j] : 0;
  RankKey k[kRankPer];
#pragma unroll
  for (int j = 0; j < kRankPer; ++j) k[j] = p0 + j < a.n ? rank_key_of(a, rows[j]) : RankKey{0, 3};
  last_bits[tid] = k[kRankPer - 1].bits;
  last_cls[tid] = k[kRankPer - 1].cls;
  __syncthreads();
  RankKey prev{0, 4};   // (no position before the first: a class no row has)
  if (tid > 0) {
    prev.bits = last_bits[tid - 1];
    prev.cls = last_cls[tid - 1];
  } else if (p0 > 0) {
    prev = rank_key_of(a, a.sorted[p0 - 1]);
  }
  uint32_t byte = 0;
#pragma unroll
  for (int j = 0; j < kRankPer; ++j) {
    const bool starts = p0 + j < a.n && !(k[j].cls == prev.cls && k[j].bits == prev.bits);
    byte |= (starts ? 1u : 0u) << j;
    prev = k[j];
  }
  a.flags[tile * kRankThreads + tid] = static_cast<uint8_t>(byte);
  uint32_t cnt = static_cast<uint32_t>(__builtin_popcount(byte));
  int64_t first = byte != 0 ? p0 + __builtin_ctz(byte) : a.n;
  int64_t last = byte != 0 ? p0 + (31 - __builtin_clz(byte)) : -1;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    cnt += __shfl_xor(cnt, d, 64);
    const int64_t f = static_cast<int64_t>(shfl_u64(static_cast<uint64_t>(first), lane ^ d));
    const int64_t l = static_cast<int64_t>(shfl_u64(static_cast<uint64_t>(last), lane ^ d));
    first = f < first ? f : first;
    last = l > last ? l : last;
  }
  if (lane == 0) {
    w_count[wave] = cnt;
    w_first[wave] = first;
    w_last[wave] = last;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kRankThreads / 64; ++w) {
      cnt += w_count[w];
      first = w_first[w] < first ? w_first[w] : first;
      last = w_last[w] > last ? w_last[w] : last;
    }
    a.tile_count[tile] = cnt;
    a.tile_first[tile] = first;
    a.tile_last[tile] = last;
  }
}

// One workgroup: forward over the tiles (run starts before each, the last run start before each), then backward (the
// first run start after each).  A thread takes kRankScanPer consecutive tiles per round.
__global__ __launch_bounds__(kRankScanThreads) void rank_scan_kernel(RankArgs a) {
  __shared__ uint64_t w_sum[kRankScanThreads / 64];
  __shared__ int64_t w_ext[kRankScanThreads / 64];
  __shared__ uint64_t carry_sum;
  __shared__ int64_t carry_ext;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int64_t kRound = int64_t(kRankScanThreads) * kRankScanPer;
  if (tid == 0) {
    carry_sum = 0;
    carry_ext = 0;   // (position 0 starts a run whenever there is a row)
  }
  __syncthreads();
  for (int64_t base = 0; base < a.tiles; base += kRound) {
    const int64_t t0 = base + static_cast<int64_t>(tid) * kRankScanPer;
    uint64_t sum = 0;
    int64_t ext = -1;
    for (int j = 0; j < kRankScanPer; ++j) {
      if (t0 + j < a.tiles) {
        sum += a.tile_count[t0 + j];
        const int64_t l = a.tile_last[t0 + j];
        ext = l > ext ? l : ext;
      }
    }
    uint64_t isum = sum;
    int64_t iext = ext;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t s = shfl_u64(isum, lane - d < 0 ? lane : lane - d);
      const int64_t e = static_cast<int64_t>(shfl_u64(static_cast<uint64_t>(iext), lane - d < 0 ? lane : lane - d));
      if (lane >= d) {
        isum += s;
        iext = e > iext ? e : iext;
      }
    }
    if (lane == 63) {
      w_sum[wave] = isum;
      w_ext[wave] = iext;
    }
    __syncthreads();
    uint64_t before = carry_sum + (isum - sum);
    int64_t last_before = carry_ext;
    {   // exclusive over the lanes of this wave, then the waves before it
      const int64_t e = static_cast<int64_t>(shfl_u64(static_cast<uint64_t>(iext), lane == 0 ? 0 : lane - 1));
      if (lane > 0) last_before = e > last_before ? e : last_before;
    }
    for (int w = 0; w < wave; ++w) {
      before += w_sum[w];
      last_before = w_ext[w] > last_before ? w_ext[w] : last_before;
    }
    for (int j = 0; j < kRankScanPer; ++j) {
      if (t0 + j < a.tiles) {
        a.dense_base[t0 + j] = before;
        a.carry_start[t0 + j] = last_before;
        before += a.tile_count[t0 + j];
        const int64_t l = a.tile_last[t0 + j];
        last_before = l > last_before ? l : last_before;
      }
    }
    __syncthreads();
    if (tid == kRankScanThreads - 1) {
      carry_sum = before;
      carry_ext = last_before;
    }
    __syncthreads();
  }
  if (tid == 0) carry_ext = a.n;
  __syncthreads();
  const int64_t rounds = (a.tiles + kRound - 1) / kRound;
  for (int64_t r = rounds - 1; r >= 0; --r) {
    const int64_t t0 = r * kRound + static_cast<int64_t>(tid) * kRankScanPer;
    int64_t ext = a.n;
    for (int j = 0; j < kRankScanPer; ++j) {
      if (t0 + j < a.tiles) {
        const int64_t f = a.tile_first[t0 + j];
        ext = f < ext ? f : ext;
      }
    }
    int64_t iext = ext;   // inclusive suffix minimum over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int64_t e = static_cast<int64_t>(shfl_u64(static_cast<uint64_t>(iext), lane + d > 63 ? lane : lane + d));
      if (lane + d <= 63) iext = e < iext ? e : iext;
    }
    if (lane == 0) w_ext[wave] = iext;
    __syncthreads();
    int64_t first_after = carry_ext;
    {
      const int64_t e = static_cast<int64_t>(shfl_u64(static_cast<uint64_t>(iext), lane == 63 ? 63 : lane + 1));
      if (lane < 63) first_after = e < first_after ? e : first_after;
    }
    for (int w = wave + 1; w < kRankScanThreads / 64; ++w) first_after = w_ext[w] < first_after ? w_ext[w] : first_after;
    for (int j = kRankScanPer - 1; j >= 0; --j) {
      if (t0 + j < a.tiles) {
        a.carry_end[t0 + j] = first_after;
        const int64_t f = a.tile_first[t0 + j];
        first_after = f < first_after ? f : first_after;
      }
    }
    __syncthreads();
    if (tid == 0) carry_ext = first_after;
    __syncthreads();
  }
}

// NormalRanker::TransformValue (vector_rank.cc:204-209) = arrow::internal::NormalPPF (util/math_internal.cc:26-137): Wichura's
// Algorithm AS 241 (PPND16; Applied Statistics 37(3), 1988 — the published coefficients).  A rational function of
// 0.180625 - q^2 for |q = p - 1/2| < 0.425, of sqrt(-log(min(p, 1 - p))) - 1.6 (or - 5) in the tails; Horner form in the
// reference's order of operations with contraction OFF — the reference build has no fused multiply-add, so every product
// and every sum rounds as it does there: the centre comes out bit for bit, the tails within the two libraries' log().
__device__ __forceinline__ double ppf_horner(const double (&c)[8], double r) {
#pragma clang fp contract(off)
  double acc = c[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) acc = acc * r + c[k];
  return acc;
}

__device__ __attribute__((noinline)) double normal_ppf(double p) {
#pragma clang fp contract(off)
  constexpr double A[8] = {2.5090809287301226727e3, 3.3430575583588128105e4, 6.7265770927008700853e4, 4.5921953931549871457e4,
                           1.3731693765509461125e4, 1.9715909503065514427e3, 1.3314166789178437745e2, 3.3871328727963666080e0};
  constexpr double B[8] = {5.2264952788528545610e3, 2.8729085735721942674e4, 3.9307895800092710610e4, 2.1213794301586595867e4,
                           5.3941960214247511077e3, 6.8718700749205790830e2, 4.2313330701600911252e1, 1.0};
  constexpr double C[8] = {7.74545014278341407640e-4, 2.27238449892691845833e-2, 2.41780725177450611770e-1, 1.27045825245236838258e0,
                           3.64784832476320460504e0, 5.76949722146069140550e0, 4.63033784615654529590e0, 1.42343711074968357734e0};
  constexpr double D[8] = {1.05075007164441684324e-9, 5.47593808499534494600e-4, 1.51986665636164571966e-2, 1.48103976427480074590e-1,
                           6.89767334985100004550e-1, 1.67638483018380384940e0, 2.05319162663775882187e0, 1.0};
  constexpr double E[8] = {2.01033439929228813265e-7, 2.71155556874348757815e-5, 1.24266094738807843860e-3, 2.65321895265761230930e-2,
                           2.96560571828504891230e-1, 1.78482653991729133580e0, 5.46378491116411436990e0, 6.65790464350110377720e0};
  constexpr double F[8] = {2.04426310338993978564e-15, 1.42151175831644588870e-7, 1.84631831751005468180e-5, 7.86869131145613259100e-4,
                           1.48753612908506148525e-2, 1.36929880922735805310e-1, 5.99832206555887937690e-1, 1.0};
  if (p == 0.0) return -HUGE_VAL;
  if (p == 1.0) return HUGE_VAL;
  const double q = p - 0.5;
  if (fabs(q) < 0.425) {
    const double r = 0.180625 - q * q;
    return q * ppf_horner(A, r) / ppf_horner(B, r);
  }
  double r = sqrt(-log(q < 0.0 ? p : 1.0 - p));
  if (r < 5.0) {
    r -= 1.6;
    r = ppf_horner(C, r) / ppf_horner(D, r);
  } else {
    r -= 5.0;
    r = ppf_horner(E, r) / ppf_horner(F, r);
  }
  return copysign(r, q);
}


__global__ __launch_bounds__(kRankThreads) void rank_emit_kernel(RankArgs a) {
  __shared__ uint32_t w_count[kRankThreads / 64];
  __shared__ int64_t w_first[kRankThreads / 64], w_last[kRankThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tile = blockIdx.x;
  const int64_t p0 = tile * kRankTile + static_cast<int64_t>(tid) * kRankPer;
  const uint32_t byte = a.flags[tile * kRankThreads + tid];
  const uint32_t cnt = static_cast<uint32_t>(__builtin_popcount(byte));
  const int64_t first = byte != 0 ? p0 + __builtin_ctz(byte) : a.n;
  const int64_t last = byte != 0 ? p0 + (31 - __builtin_clz(byte)) : -1;
  uint32_t icnt = cnt;
  int64_t ilast = last, ifirst = first;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t c = __shfl(icnt, lane - d < 0 ? lane : lane - d, 64);
    const int64_t l = static_cast<int64_t>(shfl_u64(static_cast<uint64_t>(ilast), lane - d < 0 ? lane : lane - d));
    const int64_t f = static_cast<int64_t>(shfl_u64(static_cast<uint64_t>(ifirst), lane + d > 63 ? lane : lane + d));
    if (lane >= d) {
      icnt += c;
      ilast = l > ilast ? l : ilast;
    }
    if (lane + d <= 63) ifirst = f < ifirst ? f : ifirst;
  }
  if (lane == 63) {
    w_count[wave] = icnt;
    w_last[wave] = ilast;
  }
  if (lane == 0) w_first[wave] = ifirst;
  __syncthreads();
  uint64_t dense = a.dense_base[tile] + (icnt - cnt);
  int64_t run_start = a.carry_start[tile];
  int64_t next_start = a.carry_end[tile];
  {
    const int64_t l = static_cast<int64_t>(shfl_u64(static_cast<uint64_t>(ilast), lane == 0 ? 0 : lane - 1));
    const int64_t f = static_cast<int64_t>(shfl_u64(static_cast<uint64_t>(ifirst), lane == 63 ? 63 : lane + 1));
    if (lane > 0) run_start = l > run_start ? l : run_start;
    if (lane < 63) next_start = f < next_start ? f : next_start;
  }
  for (int w = 0; w < wave; ++w) {
    dense += w_count[w];
    run_start = w_last[w] > run_start ? w_last[w] : run_start;
  }
  for (int w = wave + 1; w < kRankThreads / 64; ++w) next_start = w_first[w] < next_start ? w_first[w] : next_start;
  // run_start: the last run start before this thread's positions; next_start: the first one after them
  int64_t ends[kRankPer];
  {
    int64_t e = next_start;
#pragma unroll
    for (int j = kRankPer - 1; j >= 0; --j) {
      ends[j] = e;                                 // the first run start AFTER position p0 + j
      if ((byte >> j) & 1u) e = p0 + j;
    }
  }
  const double length = static_cast<double>(a.n);
#pragma unroll
  for (int j = 0; j < kRankPer; ++j) {
    const int64_t p = p0 + j;
    if (p >= a.n) break;
    if ((byte >> j) & 1u) {
      run_start = p;
      ++dense;
    }
    const uint64_t row = a.sorted[p];
    switch (a.tiebreaker) {
      case ARX_RANK_MIN: static_cast<uint64_t*>(a.out)[row] = static_cast<uint64_t>(run_start) + 1; break;
      case ARX_RANK_MAX: static_cast<uint64_t*>(a.out)[row] = static_cast<uint64_t>(ends[j]); break;
      case ARX_RANK_DENSE: static_cast<uint64_t*>(a.out)[row] = dense; break;
      default: {   // quantile: (rows below the run + half the run) / n  (BaseQuantileRanker, vector_rank.cc:183-186)
        const double freq = static_cast<double>(ends[j] - run_start);
        const double quantile = (static_cast<double>(run_start) + 0.5 * freq) / length;
        static_cast<double*>(a.out)[row] = a.tiebreaker == ARX_RANK_NORMAL ? normal_ppf(quantile) : quantile;
        break;
      }
    }
  }
}

// RankOptions::First: the position in the sorted order, nothing to mark
__global__ __launch_bounds__(kRankThreads) void rank_first_kernel(const uint64_t* __restrict__ sorted, int64_t n,
                                                                   uint64_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kRankThreads;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * kRankThreads + threadIdx.x; p < n; p += stride) {
    out[sorted[p]] = static_cast<uint64_t>(p) + 1;
  }
}

}  // namespace arx

using namespace arx;

extern "C" {

size_t arx_rank_workspace_bytes(int64_t length) {
  const size_t tiles = static_cast<size_t>((std::max<int64_t>(length, 1) + kRankTile - 1) / kRankTile);
  auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
  return align(tiles * kRankThreads) + align(tiles * 4) + 5 * align(tiles * 8) + 256;
}

int arx_rank(const ArxSpan* values, int key_type, const uint64_t* sorted_rows, int tiebreaker, void* ws, size_t ws_bytes,
             void* out, void* stream) {
  if (values == nullptr || (values->length > 0 && (sorted_rows == nullptr || out == nullptr || values->data == nullptr))) {
    set_error("arx_rank: null argument");
    return ARX_INVALID;
  }
  if (key_type < ARX_KEY_UINT64 || key_type > ARX_KEY_FLOAT32) {
    set_error("arx_rank: key type %d", key_type);
    return ARX_NOT_IMPLEMENTED;
  }
  if (tiebreaker < ARX_RANK_MIN || tiebreaker > ARX_RANK_NORMAL) {
    set_error("arx_rank: tiebreaker %d", tiebreaker);
    return ARX_INVALID;
  }
  const int64_t n = values->length;
  if (n == 0) return ARX_OK;
  hipStream_t st = as_stream(stream);
  if (tiebreaker == ARX_RANK_FIRST) {
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>((n + kRankThreads - 1) / kRankThreads, 256 * 32));
    hipLaunchKernelGGL(rank_first_kernel, dim3(grid), dim3(kRankThreads), 0, st, sorted_rows, n, static_cast<uint64_t*>(out));
    ARX_CHECK_LAUNCH("rank_first_kernel");
    return ARX_OK;
  }
  if (ws == nullptr || ws_bytes < arx_rank_workspace_bytes(n) || (reinterpret_cast<uintptr_t>(ws) & 255) != 0) {
    set_error("arx_rank: workspace of %zu bytes, 256-byte aligned, needed", arx_rank_workspace_bytes(n));
    return ARX_INVALID;
  }
  RankArgs a{};
  a.values = values->data;
  a.valid = make_bits(values->null_count == 0 ? nullptr : values->validity, values->offset, n);
  a.offset = values->offset;
  a.key_type = key_type;
  a.sorted = sorted_rows;
  a.n = n;
  a.tiles = (n + kRankTile - 1) / kRankTile;
  a.tiebreaker = tiebreaker;
  auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
  uint8_t* p = static_cast<uint8_t*>(ws);
  const size_t tiles = static_cast<size_t>(a.tiles);
  a.flags = p; p += align(tiles * kRankThreads);
  a.tile_count = reinterpret_cast<uint32_t*>(p); p += align(tiles * 4);
  a.tile_first = reinterpret_cast<int64_t*>(p); p += align(tiles * 8);
  a.tile_last = reinterpret_cast<int64_t*>(p); p += align(tiles * 8);
  a.dense_base = reinterpret_cast<uint64_t*>(p); p += align(tiles * 8);
  a.carry_start = reinterpret_cast<int64_t*>(p); p += align(tiles * 8);
  a.carry_end = reinterpret_cast<int64_t*>(p); p += align(tiles * 8);
  a.out = out;
  hipLaunchKernelGGL(rank_mark_kernel, dim3(static_cast<unsigned>(a.tiles)), dim3(kRankThreads), 0, st, a);
  ARX_CHECK_LAUNCH("rank_mark_kernel");
  hipLaunchKernelGGL(rank_scan_kernel, dim3(1), dim3(kRankScanThreads), 0, st, a);
  ARX_CHECK_LAUNCH("rank_scan_kernel");
  hipLaunchKernelGGL(rank_emit_kernel, dim3(static_cast<unsigned>(a.tiles)), dim3(kRankThreads), 0, st, a);
  ARX_CHECK_LAUNCH("rank_emit_kernel");
  return ARX_OK;
}

}  // extern "C"
