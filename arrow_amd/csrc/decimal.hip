// hash_sum / hash_min / hash_max of decimal128 values over dense group ids.
//
// What it restates (semantics only): GroupedReducingAggregator<Decimal128Type, GroupedSumImpl>
// (cpp/src/arrow/compute/kernels/hash_aggregate_numeric.cc:44-152,189-215): the accumulator is a Decimal128 per group
// (FindAccumulatorType, aggregate_internal.h:41-58), Reduce = u + v — BasicDecimal128::operator+= (util/basic_decimal.cc), i.e.
// two's-complement addition modulo 2^128 —, the output type the input widened to precision 38 (WidenDecimalToMaxPrecision,
// :165-167).  Addition modulo 2^128 is associative and commutative, so the order the rows arrive in does not matter and the
// sum can be kept with atomics: the low word is added first, the atomic's return value says whether THAT addition wrapped
// (every wrap of the low accumulator is seen by exactly one row), and the carry goes into the row's addend for the high word.
#include "arx_common.h"

#include <algorithm>
#include <vector>

namespace arx {

struct Dec128 {   // BasicDecimal128 in memory: little-endian words (util/basic_decimal.h)
  uint64_t lo;
  uint64_t hi;
};

__global__ __launch_bounds__(kBlock) void dec128_hash_sum_kernel(const Dec128* __restrict__ values, Bits vvalid, int is_scalar, Dec128 scalar,
                                                                 int scalar_valid, const uint32_t* __restrict__ gids, int64_t n,
                                                                 unsigned long long* __restrict__ lo, unsigned long long* __restrict__ hi,
                                                                 unsigned long long* __restrict__ counts, uint32_t* __restrict__ null_seen) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint32_t g = gids[i];
    const bool ok = is_scalar ? scalar_valid != 0 : (vvalid.base == nullptr || ((load_word(vvalid, i >> 6) >> (i & 63)) & 1ull));
    if (!ok) {
      if ((null_seen[g] & 1u) == 0) atomicOr(&null_seen[g], 1u);
      continue;
    }
    const Dec128 v = is_scalar ? scalar : values[i];
    const unsigned long long old = atomicAdd(&lo[g], static_cast<unsigned long long>(v.lo));
    const unsigned long long carry = (old + v.lo) < old ? 1ull : 0ull;
    atomicAdd(&hi[g], static_cast<unsigned long long>(v.hi) + carry);
    atomicAdd(&counts[g], 1ull);
  }
}

__global__ __launch_bounds__(kBlock) void dec128_hash_sum_merge_kernel(unsigned long long* __restrict__ lo, unsigned long long* __restrict__ hi,
                                                                       long long* __restrict__ counts, uint32_t* __restrict__ null_seen,
                                                                       const unsigned long long* __restrict__ other_lo,
                                                                       const unsigned long long* __restrict__ other_hi,
                                                                       const long long* __restrict__ other_counts,
                                                                       const uint32_t* __restrict__ other_null_seen,
                                                                       const uint32_t* __restrict__ mapping, int64_t m) {
  // Merge (:85-107): group g of the other state lands on mapping[g], each target at most once per call
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; g < m; g += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint32_t t = mapping[g];
    const unsigned long long a = lo[t], b = other_lo[g];
    lo[t] = a + b;
    hi[t] = hi[t] + other_hi[g] + ((a + b) < a ? 1ull : 0ull);
    counts[t] += other_counts[g];
    if (other_null_seen[g] & 1u) null_seen[t] |= 1u;
  }
}

__global__ __launch_bounds__(kBlock) void dec128_pack_kernel(const unsigned long long* __restrict__ lo, const unsigned long long* __restrict__ hi,
                                                             int64_t m, Dec128* __restrict__ out) {
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; g < m; g += static_cast<int64_t>(gridDim.x) * kBlock) {
    out[g] = Dec128{lo[g], hi[g]};
  }
}

// the two halves of every value as columns of their own: a decimal128 sort key is the pair (high word as int64, low word
// as uint64) — the order of BasicDecimal128::operator< (util/basic_decimal.h) — sorted as two keys of the existing sort
__global__ __launch_bounds__(kBlock) void dec128_split_kernel(const Dec128* __restrict__ in, int64_t m, unsigned long long* __restrict__ lo,
                                                              unsigned long long* __restrict__ hi) {
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; g < m; g += static_cast<int64_t>(gridDim.x) * kBlock) {
    const Dec128 v = in[g];
    lo[g] = v.lo;
    hi[g] = v.hi;
  }
}

// ---- hash_min / hash_max of decimal128 values: GroupedMinMaxImpl<Decimal128Type> (kernels/hash_aggregate.cc:330-419) — per
// group the smallest and the largest value (signed 128-bit order), has_values and has_nulls.  There is no 128-bit atomic
// min / max: the rows are stably sorted by group id (arx_sort_indices), which makes every group one run, and one owner per
// group folds its run into the group's state — a thread for a short run, a wave for a run of more than 1024 rows (min / max
// are associative and commutative: the lanes take the run's rows round-robin and meet in a shuffle reduction).
constexpr int64_t kDecLongRun = 1024;

__device__ __forceinline__ bool dec128_less(const Dec128& a, const Dec128& b) {
  const long long ah = static_cast<long long>(a.hi), bh = static_cast<long long>(b.hi);
  return ah < bh || (ah == bh && a.lo < b.lo);
}

struct DecFold {
  Dec128 mn, mx;
  bool any, saw_null;
};
__device__ __forceinline__ void dec_fold_row(DecFold& f, const Dec128& v, bool ok) {
  if (!ok) {
    f.saw_null = true;
    return;
  }
  if (!f.any || dec128_less(v, f.mn)) f.mn = v;
  if (!f.any || dec128_less(f.mx, v)) f.mx = v;
  f.any = true;
}
__device__ __forceinline__ void dec_fold_store(const DecFold& f, uint32_t g, Dec128* __restrict__ mins, Dec128* __restrict__ maxs,
                                               uint32_t* __restrict__ seen) {
  // seen: bit 0 = a null value hit the group, bit 1 = the group has a value (then mins / maxs hold its extrema so far)
  const uint32_t before = seen[g];
  if (f.any) {
    if ((before & 2u) == 0 || dec128_less(f.mn, mins[g])) mins[g] = f.mn;
    if ((before & 2u) == 0 || dec128_less(maxs[g], f.mx)) maxs[g] = f.mx;
  }
  seen[g] = before | (f.any ? 2u : 0u) | (f.saw_null ? 1u : 0u);
}

__global__ __launch_bounds__(kBlock) void dec128_minmax_walk_kernel(const Dec128* __restrict__ values, Bits vvalid,
                                                                    const uint32_t* __restrict__ gids, const uint64_t* __restrict__ perm,
                                                                    int64_t n, Dec128* __restrict__ mins, Dec128* __restrict__ maxs,
                                                                    uint32_t* __restrict__ seen, unsigned long long* __restrict__ long_runs) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint32_t g = gids[perm[i]];
    if (i > 0 && gids[perm[i - 1]] == g) continue;   // not the first row of its group's run
    if (i + kDecLongRun < n && gids[perm[i + kDecLongRun]] == g) {
      long_runs[1 + atomicAdd(&long_runs[0], 1ull)] = static_cast<unsigned long long>(i);
      continue;
    }
    DecFold f{Dec128{0, 0}, Dec128{0, 0}, false, false};
    for (int64_t j = i; j < n; ++j) {
      const uint64_t r = perm[j];
      if (gids[r] != g) break;
      const bool ok = vvalid.base == nullptr || ((load_word(vvalid, static_cast<int64_t>(r) >> 6) >> (r & 63)) & 1ull);
      dec_fold_row(f, values[r], ok);
    }
    dec_fold_store(f, g, mins, maxs, seen);
  }
}

__device__ __forceinline__ Dec128 dec_shfl_xor(const Dec128& v, int mask) {
  Dec128 r;
  r.lo = (static_cast<uint64_t>(__shfl_xor(static_cast<uint32_t>(v.lo >> 32), mask, 64)) << 32) | __shfl_xor(static_cast<uint32_t>(v.lo), mask, 64);
  r.hi = (static_cast<uint64_t>(__shfl_xor(static_cast<uint32_t>(v.hi >> 32), mask, 64)) << 32) | __shfl_xor(static_cast<uint32_t>(v.hi), mask, 64);
  return r;
}

__global__ __launch_bounds__(64) void dec128_minmax_walk_long_kernel(const Dec128* __restrict__ values, Bits vvalid,
                                                                     const uint32_t* __restrict__ gids, const uint64_t* __restrict__ perm,
                                                                     int64_t n, Dec128* __restrict__ mins, Dec128* __restrict__ maxs,
                                                                     uint32_t* __restrict__ seen, const unsigned long long* __restrict__ long_runs) {
  const int lane = threadIdx.x;
  const int64_t nruns = static_cast<int64_t>(long_runs[0]);
  for (int64_t e = blockIdx.x; e < nruns; e += gridDim.x) {
    const int64_t i = static_cast<int64_t>(long_runs[1 + e]);
    const uint32_t g = gids[perm[i]];
    DecFold f{Dec128{0, 0}, Dec128{0, 0}, false, false};
    for (int64_t j = i; j < n; j += 64) {
      const int64_t jj = j + lane < n ? j + lane : n - 1;
      const uint64_t r = perm[jj];
      const bool in_run = j + lane < n && gids[r] == g;
      if (in_run) {
        const bool ok = vvalid.base == nullptr || ((load_word(vvalid, static_cast<int64_t>(r) >> 6) >> (r & 63)) & 1ull);
        dec_fold_row(f, values[r], ok);
      }
      if (__ballot(in_run) != ~0ull) break;   // the run ended inside these 64 rows
    }
    // the lanes' partial extrema meet (a lane without a value carries any = false and loses every comparison)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const Dec128 omn = dec_shfl_xor(f.mn, m), omx = dec_shfl_xor(f.mx, m);
      const bool oany = __shfl_xor(f.any ? 1 : 0, m, 64) != 0;
      const bool onull = __shfl_xor(f.saw_null ? 1 : 0, m, 64) != 0;
      if (oany && (!f.any || dec128_less(omn, f.mn))) f.mn = omn;
      if (oany && (!f.any || dec128_less(f.mx, omx))) f.mx = omx;
      f.any = f.any || oany;
      f.saw_null = f.saw_null || onull;
    }
    if (lane == 0) dec_fold_store(f, g, mins, maxs, seen);
  }
}

__global__ __launch_bounds__(kBlock) void dec128_minmax_finalize_kernel(const uint32_t* __restrict__ seen, int64_t m, int skip_nulls,
                                                                        uint64_t* __restrict__ out_validity,
                                                                        unsigned long long* __restrict__ valid_count) {
  // Finalize (:401-419): valid = has_values && (skip_nulls || !has_nulls); one 64-group word per wave step
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwords = (m + 63) >> 6;
  unsigned long long total = 0;
  for (int64_t w = wave_g; w < nwords; w += nwaves) {
    const int64_t g = (w << 6) + lane;
    const uint32_t s = g < m ? seen[g] : 0u;
    const bool valid = (s & 2u) != 0 && (skip_nulls != 0 || (s & 1u) == 0);
    const uint64_t word = __ballot(valid);
    if (lane == 0) out_validity[w] = word;
    total += __popcll(word);
  }
  if (valid_count != nullptr && lane == 0 && total != 0) atomicAdd(valid_count, total);
}

// ---- sum / mean / min_max of a decimal128 COLUMN: SumImpl / MeanImpl / MinMaxImpl<Decimal128Type>
// (kernels/aggregate_basic.inc.cc:49-110,229-258,776-860) keep {sum modulo 2^128, count of valid values, min, max}; a batch is
// folded by every thread over its strided rows, the lanes of a wave meet in a shuffle reduction (all four are associative and
// commutative), and every wave leaves one 64-byte partial for the host to combine — a few thousand, whatever the length.
struct DecPartial {
  uint64_t sum_lo, sum_hi, count, any;
  Dec128 mn, mx;
};

__device__ __forceinline__ uint64_t dec_shfl_xor_u64(uint64_t v, int mask) {
  return (static_cast<uint64_t>(__shfl_xor(static_cast<uint32_t>(v >> 32), mask, 64)) << 32) | __shfl_xor(static_cast<uint32_t>(v), mask, 64);
}

__global__ __launch_bounds__(kBlock) void dec128_reduce_kernel(const Dec128* __restrict__ values, Bits vvalid, int64_t n,
                                                               DecPartial* __restrict__ partials) {
  const int lane = lane_id();
  DecFold f{Dec128{0, 0}, Dec128{0, 0}, false, false};
  uint64_t slo = 0, shi = 0, cnt = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const bool ok = vvalid.base == nullptr || ((load_word(vvalid, i >> 6) >> (i & 63)) & 1ull);
    if (!ok) continue;
    const Dec128 v = values[i];
    const uint64_t before = slo;
    slo += v.lo;
    shi += v.hi + (slo < before ? 1ull : 0ull);
    ++cnt;
    dec_fold_row(f, v, true);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const uint64_t olo = dec_shfl_xor_u64(slo, m), ohi = dec_shfl_xor_u64(shi, m);
    const uint64_t before = slo;
    slo += olo;
    shi += ohi + (slo < before ? 1ull : 0ull);
    cnt += dec_shfl_xor_u64(cnt, m);
    const Dec128 omn = dec_shfl_xor(f.mn, m), omx = dec_shfl_xor(f.mx, m);
    const bool oany = __shfl_xor(f.any ? 1 : 0, m, 64) != 0;
    if (oany && (!f.any || dec128_less(omn, f.mn))) f.mn = omn;
    if (oany && (!f.any || dec128_less(f.mx, omx))) f.mx = omx;
    f.any = f.any || oany;
  }
  if (lane == 0) {
    const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    partials[wave_g] = DecPartial{slo, shi, cnt, f.any ? 1ull : 0ull, f.mn, f.mx};
  }
}

constexpr int kDecReduceBlocks = 1024;

static inline unsigned dec_grid(int64_t n) {
  return static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((n + kBlock - 1) / kBlock, 256 * 16)));
}

extern "C" {

int arx_hash_sum_dec128_consume(const ArxSpan* values, int values_is_scalar, uint64_t scalar_lo, uint64_t scalar_hi,
                                const uint32_t* group_ids, int64_t length, uint64_t* sums_lo, uint64_t* sums_hi, int64_t* counts,
                                uint32_t* null_seen, void* stream) {
  if (values == nullptr || length < 0) {
    set_error("bad arguments to arx_hash_sum_dec128_consume");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  if (group_ids == nullptr || sums_lo == nullptr || sums_hi == nullptr || counts == nullptr || null_seen == nullptr ||
      (!values_is_scalar && values->data == nullptr)) {
    set_error("arx_hash_sum_dec128_consume: NULL buffer");
    return ARX_INVALID;
  }
  const bool has_nulls = !values_is_scalar && values->null_count != 0 && values->validity != nullptr;
  const Bits vvalid = has_nulls ? make_bits(values->validity, values->offset, length) : Bits{};
  const Dec128* v = values_is_scalar ? nullptr : static_cast<const Dec128*>(values->data) + values->offset;
  hipLaunchKernelGGL(dec128_hash_sum_kernel, dim3(dec_grid(length)), dim3(kBlock), 0, as_stream(stream), v, vvalid, values_is_scalar,
                     Dec128{scalar_lo, scalar_hi}, values->null_count == 0 ? 1 : 0, group_ids, length,
                     reinterpret_cast<unsigned long long*>(sums_lo), reinterpret_cast<unsigned long long*>(sums_hi),
                     reinterpret_cast<unsigned long long*>(counts), null_seen);
  ARX_CHECK_LAUNCH("dec128_hash_sum_kernel");
  return ARX_OK;
}

int arx_hash_sum_dec128_merge(uint64_t* sums_lo, uint64_t* sums_hi, int64_t* counts, uint32_t* null_seen, const uint64_t* other_lo,
                              const uint64_t* other_hi, const int64_t* other_counts, const uint32_t* other_null_seen,
                              const uint32_t* group_id_mapping, int64_t other_num_groups, void* stream) {
  if (other_num_groups < 0) {
    set_error("bad arguments to arx_hash_sum_dec128_merge");
    return ARX_INVALID;
  }
  if (other_num_groups == 0) return ARX_OK;
  if (sums_lo == nullptr || sums_hi == nullptr || counts == nullptr || null_seen == nullptr || other_lo == nullptr || other_hi == nullptr ||
      other_counts == nullptr || other_null_seen == nullptr || group_id_mapping == nullptr) {
    set_error("arx_hash_sum_dec128_merge: NULL buffer");
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(dec128_hash_sum_merge_kernel, dim3(dec_grid(other_num_groups)), dim3(kBlock), 0, as_stream(stream),
                     reinterpret_cast<unsigned long long*>(sums_lo), reinterpret_cast<unsigned long long*>(sums_hi),
                     reinterpret_cast<long long*>(counts), null_seen, reinterpret_cast<const unsigned long long*>(other_lo),
                     reinterpret_cast<const unsigned long long*>(other_hi), reinterpret_cast<const long long*>(other_counts), other_null_seen,
                     group_id_mapping, other_num_groups);
  ARX_CHECK_LAUNCH("dec128_hash_sum_merge_kernel");
  return ARX_OK;
}

int arx_dec128_pack(const uint64_t* lo, const uint64_t* hi, int64_t n, void* out_values, void* stream) {
  if (n < 0 || (n > 0 && (lo == nullptr || hi == nullptr || out_values == nullptr))) {
    set_error("bad arguments to arx_dec128_pack");
    return ARX_INVALID;
  }
  if (n == 0) return ARX_OK;
  hipLaunchKernelGGL(dec128_pack_kernel, dim3(dec_grid(n)), dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const unsigned long long*>(lo),
                     reinterpret_cast<const unsigned long long*>(hi), n, static_cast<Dec128*>(out_values));
  ARX_CHECK_LAUNCH("dec128_pack_kernel");
  return ARX_OK;
}

int arx_dec128_split(const void* values, int64_t n, uint64_t* out_lo, uint64_t* out_hi, void* stream) {
  if (n < 0 || (n > 0 && (values == nullptr || out_lo == nullptr || out_hi == nullptr))) {
    set_error("bad arguments to arx_dec128_split");
    return ARX_INVALID;
  }
  if (n == 0) return ARX_OK;
  hipLaunchKernelGGL(dec128_split_kernel, dim3(dec_grid(n)), dim3(kBlock), 0, as_stream(stream), static_cast<const Dec128*>(values), n,
                     reinterpret_cast<unsigned long long*>(out_lo), reinterpret_cast<unsigned long long*>(out_hi));
  ARX_CHECK_LAUNCH("dec128_split_kernel");
  return ARX_OK;
}

size_t arx_hash_minmax_dec128_workspace_bytes(int64_t length) {
  if (length <= 0) return 0;
  const size_t n = static_cast<size_t>(length);
  auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
  return al(arx_sort_indices_workspace_bytes(length)) + al(n * 8) + al((n / kDecLongRun + 2) * 8) + 512;
}

int arx_hash_minmax_dec128_consume(const ArxSpan* values, const uint32_t* group_ids, int64_t length, void* ws, size_t ws_bytes, void* mins,
                                   void* maxs, uint32_t* seen, void* stream) {
  if (values == nullptr || length < 0) {
    set_error("bad arguments to arx_hash_minmax_dec128_consume");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  if (values->data == nullptr || group_ids == nullptr || mins == nullptr || maxs == nullptr || seen == nullptr || ws == nullptr ||
      ws_bytes < arx_hash_minmax_dec128_workspace_bytes(length)) {
    set_error("arx_hash_minmax_dec128_consume: NULL buffer or a workspace below arx_hash_minmax_dec128_workspace_bytes");
    return ARX_INVALID;
  }
  auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
  const size_t n = static_cast<size_t>(length);
  uint8_t* p = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~uintptr_t(255));
  const size_t sort_bytes = arx_sort_indices_workspace_bytes(length);
  uint64_t* perm = reinterpret_cast<uint64_t*>(p + al(sort_bytes));
  unsigned long long* long_runs = reinterpret_cast<unsigned long long*>(p + al(sort_bytes) + al(n * 8));
  const ArxSpan keys{nullptr, group_ids, 0, length, 0};
  const int rc = arx_sort_indices(&keys, ARX_KEY_UINT32, ARX_SORT_ASCENDING, ARX_NULLS_AT_END, p, sort_bytes, perm, stream);
  if (rc != ARX_OK) return rc;
  hipStream_t st = as_stream(stream);
  ARX_HIP(hipMemsetAsync(long_runs, 0, 8, st));
  const bool has_nulls = values->null_count != 0 && values->validity != nullptr;
  const Bits vvalid = has_nulls ? make_bits(values->validity, values->offset, length) : Bits{};
  const Dec128* v = static_cast<const Dec128*>(values->data) + values->offset;
  hipLaunchKernelGGL(dec128_minmax_walk_kernel, dim3(dec_grid(length)), dim3(kBlock), 0, st, v, vvalid, group_ids, perm, length,
                     static_cast<Dec128*>(mins), static_cast<Dec128*>(maxs), seen, long_runs);
  ARX_CHECK_LAUNCH("dec128_minmax_walk_kernel");
  if (length > kDecLongRun) {
    hipLaunchKernelGGL(dec128_minmax_walk_long_kernel, dim3(256 * 8), dim3(64), 0, st, v, vvalid, group_ids, perm, length,
                       static_cast<Dec128*>(mins), static_cast<Dec128*>(maxs), seen, long_runs);
    ARX_CHECK_LAUNCH("dec128_minmax_walk_long_kernel");
  }
  return ARX_OK;
}

int arx_hash_minmax_dec128_finalize(const uint32_t* seen, int64_t num_groups, int skip_nulls, void* out_validity, int64_t* valid_count,
                                    void* stream) {
  if (num_groups < 0 || (num_groups > 0 && (seen == nullptr || out_validity == nullptr))) {
    set_error("bad arguments to arx_hash_minmax_dec128_finalize");
    return ARX_INVALID;
  }
  if (num_groups == 0) return ARX_OK;
  const int64_t nwords = (num_groups + 63) / 64;
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((nwords + kWavesPerBlock - 1) / kWavesPerBlock, 256 * 8)));
  hipLaunchKernelGGL(dec128_minmax_finalize_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), seen, num_groups, skip_nulls,
                     static_cast<uint64_t*>(out_validity), reinterpret_cast<unsigned long long*>(valid_count));
  ARX_CHECK_LAUNCH("dec128_minmax_finalize_kernel");
  return ARX_OK;
}

size_t arx_reduce_dec128_workspace_bytes(void) { return static_cast<size_t>(kDecReduceBlocks) * kWavesPerBlock * sizeof(DecPartial) + 256; }

int arx_reduce_dec128(const ArxSpan* values, void* ws, size_t ws_bytes, uint64_t* out8, void* stream) {
  if (values == nullptr || out8 == nullptr || values->length < 0) {
    set_error("bad arguments to arx_reduce_dec128");
    return ARX_INVALID;
  }
  for (int k = 0; k < 8; ++k) out8[k] = 0;
  const int64_t n = values->length;
  if (n == 0) return ARX_OK;
  if (values->data == nullptr || ws == nullptr || ws_bytes < arx_reduce_dec128_workspace_bytes()) {
    set_error("arx_reduce_dec128: NULL buffer or a workspace below arx_reduce_dec128_workspace_bytes");
    return ARX_INVALID;
  }
  DecPartial* partials = reinterpret_cast<DecPartial*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~uintptr_t(255));
  const int blocks = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>((n + kBlock - 1) / kBlock, kDecReduceBlocks)));
  const bool has_nulls = values->null_count != 0 && values->validity != nullptr;
  const Bits vvalid = has_nulls ? make_bits(values->validity, values->offset, n) : Bits{};
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(dec128_reduce_kernel, dim3(blocks), dim3(kBlock), 0, st, static_cast<const Dec128*>(values->data) + values->offset, vvalid, n,
                     partials);
  ARX_CHECK_LAUNCH("dec128_reduce_kernel");
  std::vector<DecPartial> host(static_cast<size_t>(blocks) * kWavesPerBlock);
  ARX_HIP(hipMemcpyAsync(host.data(), partials, host.size() * sizeof(DecPartial), hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  unsigned __int128 sum = 0;
  __int128 mn = 0, mx = 0;
  uint64_t count = 0;
  bool any = false;
  for (const DecPartial& p : host) {
    sum += (static_cast<unsigned __int128>(p.sum_hi) << 64) | p.sum_lo;
    count += p.count;
    if (p.any == 0) continue;
    const __int128 pmn = static_cast<__int128>((static_cast<unsigned __int128>(p.mn.hi) << 64) | p.mn.lo);
    const __int128 pmx = static_cast<__int128>((static_cast<unsigned __int128>(p.mx.hi) << 64) | p.mx.lo);
    if (!any || pmn < mn) mn = pmn;
    if (!any || pmx > mx) mx = pmx;
    any = true;
  }
  out8[0] = static_cast<uint64_t>(sum);
  out8[1] = static_cast<uint64_t>(sum >> 64);
  out8[2] = count;
  out8[3] = any ? 1 : 0;
  out8[4] = static_cast<uint64_t>(static_cast<unsigned __int128>(mn));
  out8[5] = static_cast<uint64_t>(static_cast<unsigned __int128>(mn) >> 64);
  out8[6] = static_cast<uint64_t>(static_cast<unsigned __int128>(mx));
  out8[7] = static_cast<uint64_t>(static_cast<unsigned __int128>(mx) >> 64);
  return ARX_OK;
}

}  // extern "C"

}  // namespace arx
