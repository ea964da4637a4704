// hash_sum of decimal128 values over dense group ids.
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

}  // extern "C"

}  // namespace arx
