// Group-by hash_sum(int64) BY int32 key on gfx950, as one fused device operator.
//
// What it restates (semantics only):
//   Grouper::Consume (GrouperFastImpl)          cpp/src/arrow/compute/row/grouper.cc:662-815
//   GroupedReducingAggregator<Int64,Sum>        cpp/src/arrow/compute/kernels/hash_aggregate_numeric.cc:44-187
//     Consume :70-83 (wrap-around add :283-287, counts++, null clears no_nulls)
//     Merge   :85-107  (sums add, counts add, no_nulls AND)
//     Finalize:130-152 + Finish :109-128 (null where count < min_count; !skip_nulls -> & no_nulls)
//   GroupByNode::Consume/Merge/Finalize         cpp/src/arrow/acero/groupby_aggregate_node.cc:210-337
// The reference's hash function and group-id order are not observable (its tests compare
// key-sorted output, acero/hash_aggregate_test.cc:262-280), so the table below uses its own.
//
// State in HBM (caller-owned, arx_groupby_state_bytes):
//   header(64 B) | tagged keys u64[C] (0 = empty, else 1<<32 | key) | sums i64[C+1] |
//   counts i64[C+1] | flags u32[C+1] (bit0 = a null value was seen)      slot C = the null key
// Insertion: Fibonacci hash -> linear probing with 64-bit CAS; accumulation: device-scope
// 64-bit atomic adds (integer sums are associative and commutative, so any order is bit-exact).
#include "arx_common.h"

#include <algorithm>

namespace arx {

struct GroupbyHeader {
  int64_t capacity;
  unsigned long long num_groups;     // occupied slots (incl. the null-key group)
  unsigned long long export_cursor;  // scratch for export
  unsigned int null_used;
  unsigned int overflow;             // table full: results invalid
  int64_t pad[4];
};
static_assert(sizeof(GroupbyHeader) == 64, "one line");

struct GroupbyView {
  GroupbyHeader* hdr;
  unsigned long long* keys;   // [C]
  unsigned long long* sums;   // [C+1]
  unsigned long long* counts; // [C+1]
  unsigned int* flags;        // [C+1]
  int64_t capacity;
  int lg;
};

static inline GroupbyView gb_view(void* state, int64_t capacity) {
  GroupbyView v;
  uint8_t* p = static_cast<uint8_t*>(state);
  v.hdr = reinterpret_cast<GroupbyHeader*>(p);
  v.keys = reinterpret_cast<unsigned long long*>(p + 64);
  v.sums = v.keys + capacity;
  v.counts = v.sums + capacity + 1;
  v.flags = reinterpret_cast<unsigned int*>(v.counts + capacity + 1);
  v.capacity = capacity;
  int lg = 0;
  while ((int64_t(1) << lg) < capacity) ++lg;
  v.lg = lg;
  return v;
}

__device__ __forceinline__ uint64_t gb_hash(int32_t key) {
  return static_cast<uint64_t>(static_cast<uint32_t>(key)) * 0x9E3779B185EBCA87ull;
}

// Slot of `key`, inserting it if absent.  Returns -1 if the table is full.
__device__ __forceinline__ int64_t gb_find_or_insert(const GroupbyView& v, int32_t key) {
  const unsigned long long tagged = (1ull << 32) | static_cast<uint32_t>(key);
  const uint64_t mask = static_cast<uint64_t>(v.capacity) - 1;
  uint64_t h = v.lg == 0 ? 0 : (gb_hash(key) >> (64 - v.lg));
  for (int64_t probes = 0; probes < v.capacity; ++probes) {
    unsigned long long cur = __hip_atomic_load(&v.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == tagged) return static_cast<int64_t>(h);
    if (cur == 0) {
      const unsigned long long old = atomicCAS(&v.keys[h], 0ull, tagged);
      if (old == 0) {
        atomicAdd(&v.hdr->num_groups, 1ull);
        return static_cast<int64_t>(h);
      }
      if (old == tagged) return static_cast<int64_t>(h);
    }
    h = (h + 1) & mask;
  }
  return -1;
}

__device__ __forceinline__ int64_t gb_null_slot(const GroupbyView& v) {
  if (atomicExch(&v.hdr->null_used, 1u) == 0u) atomicAdd(&v.hdr->num_groups, 1ull);
  return v.capacity;
}

__global__ __launch_bounds__(kBlock) void groupby_consume_kernel(GroupbyView v,
                                                                 const int32_t* __restrict__ keys,
                                                                 Bits kvalid,
                                                                 const int64_t* __restrict__ values,
                                                                 Bits vvalid, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool kv = (load_word(kvalid, i >> 6) >> (i & 63)) & 1ull;
    const bool vv = (load_word(vvalid, i >> 6) >> (i & 63)) & 1ull;
    const int64_t slot = kv ? gb_find_or_insert(v, keys[i]) : gb_null_slot(v);
    if (slot < 0) {
      atomicExch(&v.hdr->overflow, 1u);
      continue;
    }
    if (vv) {
      atomicAdd(&v.sums[slot], static_cast<unsigned long long>(values[i]));
      atomicAdd(&v.counts[slot], 1ull);
    } else {
      atomicOr(&v.flags[slot], 1u);
    }
  }
}

__global__ __launch_bounds__(kBlock) void groupby_merge_kernel(
    GroupbyView v, const int32_t* __restrict__ keys, const uint8_t* __restrict__ key_is_valid,
    const int64_t* __restrict__ sums, const int64_t* __restrict__ counts,
    const uint8_t* __restrict__ no_nulls, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool kv = key_is_valid == nullptr || key_is_valid[i] != 0;
    const int64_t slot = kv ? gb_find_or_insert(v, keys[i]) : gb_null_slot(v);
    if (slot < 0) {
      atomicExch(&v.hdr->overflow, 1u);
      continue;
    }
    atomicAdd(&v.sums[slot], static_cast<unsigned long long>(sums[i]));
    atomicAdd(&v.counts[slot], static_cast<unsigned long long>(counts[i]));
    if (no_nulls != nullptr && no_nulls[i] == 0) atomicOr(&v.flags[slot], 1u);
  }
}

__global__ __launch_bounds__(kBlock) void groupby_export_kernel(
    GroupbyView v, int32_t* __restrict__ out_keys, uint8_t* __restrict__ out_key_is_valid,
    int64_t* __restrict__ out_sums, int64_t* __restrict__ out_counts,
    uint8_t* __restrict__ out_no_nulls) {
  const int lane = lane_id();
  const int64_t nslots = v.capacity + 1;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t rounds = (nslots + stride - 1) / stride;
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t s = r * stride + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    bool occ = false;
    unsigned long long tagged = 0;
    if (s < v.capacity) {
      tagged = v.keys[s];
      occ = tagged != 0;
    } else if (s == v.capacity) {
      occ = v.hdr->null_used != 0;
    }
    const uint64_t bal = __ballot(occ);
    if (bal == 0) continue;  // wave-uniform
    unsigned long long base = 0;
    const int leader = __ffsll(static_cast<unsigned long long>(bal)) - 1;
    if (lane == leader) base = atomicAdd(&v.hdr->export_cursor, static_cast<unsigned long long>(__popcll(bal)));
    base = shfl_u64(base, leader);
    if (occ) {
      const int64_t pos = static_cast<int64_t>(base) + __popcll(bal & ((uint64_t(1) << lane) - 1));
      out_keys[pos] = s < v.capacity ? static_cast<int32_t>(static_cast<uint32_t>(tagged)) : 0;
      out_key_is_valid[pos] = s < v.capacity ? 1 : 0;
      out_sums[pos] = static_cast<int64_t>(v.sums[s]);
      out_counts[pos] = static_cast<int64_t>(v.counts[s]);
      out_no_nulls[pos] = (v.flags[s] & 1u) ? 0 : 1;
    }
  }
}

__global__ __launch_bounds__(kBlock) void groupby_finalize_kernel(const int64_t* __restrict__ counts,
                                                                  const uint8_t* __restrict__ no_nulls,
                                                                  int64_t n, int skip_nulls,
                                                                  uint32_t min_count,
                                                                  uint8_t* __restrict__ out_valid) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    bool ok = counts[i] >= static_cast<int64_t>(min_count);
    if (!skip_nulls) ok = ok && no_nulls[i] != 0;
    out_valid[i] = ok ? 1 : 0;
  }
}

// ---- partition of partial aggregates by destination rank (multi-GPU exchange)
__device__ __forceinline__ int gb_dest(int32_t key, bool valid, int num_parts) {
  if (!valid) return 0;
  const uint64_t h = static_cast<uint64_t>(static_cast<uint32_t>(key)) * 0xD6E8FEB86659FD93ull;
  return static_cast<int>((h >> 32) % static_cast<uint64_t>(num_parts));
}

__global__ __launch_bounds__(kBlock) void partition_count_kernel(const int32_t* __restrict__ keys,
                                                                 const uint8_t* __restrict__ key_is_valid,
                                                                 int64_t n, int num_parts,
                                                                 unsigned long long* part_counts) {
  const int lane = lane_id();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t rounds = (n + stride - 1) / stride;
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t i = r * stride + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int d = i < n ? gb_dest(keys[i], key_is_valid == nullptr || key_is_valid[i] != 0, num_parts) : -1;
    for (int p = 0; p < num_parts; ++p) {
      const uint64_t bal = __ballot(d == p);
      if (bal != 0 && lane == (__ffsll(static_cast<unsigned long long>(bal)) - 1)) {
        atomicAdd(&part_counts[p], static_cast<unsigned long long>(__popcll(bal)));
      }
    }
  }
}

__global__ void partition_offsets_kernel(const unsigned long long* part_counts, int num_parts,
                                         unsigned long long* cursors, int64_t* out_part_counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    unsigned long long run = 0;
    for (int p = 0; p < num_parts; ++p) {
      cursors[p] = run;
      out_part_counts[p] = static_cast<int64_t>(part_counts[p]);
      run += part_counts[p];
    }
  }
}

__global__ __launch_bounds__(kBlock) void partition_scatter_kernel(
    const int32_t* __restrict__ keys, const uint8_t* __restrict__ key_is_valid,
    const int64_t* __restrict__ sums, const int64_t* __restrict__ counts,
    const uint8_t* __restrict__ no_nulls, int64_t n, int num_parts, unsigned long long* cursors,
    int32_t* __restrict__ out_keys, uint8_t* __restrict__ out_key_is_valid,
    int64_t* __restrict__ out_sums, int64_t* __restrict__ out_counts,
    uint8_t* __restrict__ out_no_nulls) {
  const int lane = lane_id();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t rounds = (n + stride - 1) / stride;
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t i = r * stride + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool kv = i < n && (key_is_valid == nullptr || key_is_valid[i] != 0);
    const int d = i < n ? gb_dest(keys[i], kv, num_parts) : -1;
    int64_t pos = -1;
    for (int p = 0; p < num_parts; ++p) {
      const uint64_t bal = __ballot(d == p);
      if (bal == 0) continue;
      const int leader = __ffsll(static_cast<unsigned long long>(bal)) - 1;
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(&cursors[p], static_cast<unsigned long long>(__popcll(bal)));
      base = shfl_u64(base, leader);
      if (d == p) pos = static_cast<int64_t>(base) + __popcll(bal & ((uint64_t(1) << lane) - 1));
    }
    if (pos >= 0) {
      out_keys[pos] = keys[i];
      out_key_is_valid[pos] = kv ? 1 : 0;
      out_sums[pos] = sums[i];
      out_counts[pos] = counts[i];
      out_no_nulls[pos] = no_nulls == nullptr ? 1 : no_nulls[i];
    }
  }
}


// ---- HashAggregateKernel path: dense group ids come from the caller's Grouper
// (GroupedReducingAggregator<Int64Type,GroupedSumImpl>::Consume / Merge,
//  cpp/src/arrow/compute/kernels/hash_aggregate_numeric.cc:70-107; VisitGroupedValues,
//  hash_aggregate_internal.h:140-176).  State = three dense device arrays indexed by group id.
__global__ __launch_bounds__(kBlock) void hash_sum_dense_consume_kernel(
    const int64_t* __restrict__ values, int64_t scalar_value, int values_is_scalar, Bits vvalid,
    const uint32_t* __restrict__ group_ids, int64_t n, unsigned long long* __restrict__ sums,
    unsigned long long* __restrict__ counts, unsigned int* __restrict__ null_seen) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t g = group_ids[i];
    const bool ok = (load_word(vvalid, i >> 6) >> (i & 63)) & 1ull;
    if (ok) {
      const int64_t v = values_is_scalar ? scalar_value : values[i];
      atomicAdd(&sums[g], static_cast<unsigned long long>(v));
      atomicAdd(&counts[g], 1ull);
    } else {
      atomicOr(&null_seen[g], 1u);
    }
  }
}

__global__ __launch_bounds__(kBlock) void hash_sum_dense_merge_kernel(
    const int64_t* __restrict__ other_sums, const int64_t* __restrict__ other_counts,
    const uint32_t* __restrict__ other_null_seen, const uint32_t* __restrict__ mapping, int64_t n,
    unsigned long long* __restrict__ sums, unsigned long long* __restrict__ counts,
    unsigned int* __restrict__ null_seen) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t g = mapping[i];
    atomicAdd(&sums[g], static_cast<unsigned long long>(other_sums[i]));
    atomicAdd(&counts[g], static_cast<unsigned long long>(other_counts[i]));
    if (other_null_seen[i] & 1u) atomicOr(&null_seen[g], 1u);
  }
}

// One 64-bit validity word per lane-iteration: bit g = counts[g] >= min_count && (skip_nulls ||
// !null_seen[g])  (Finish + Finalize, hash_aggregate_numeric.cc:109-152).
__global__ __launch_bounds__(kBlock) void hash_sum_dense_finalize_kernel(
    const int64_t* __restrict__ counts, const uint32_t* __restrict__ null_seen, int64_t n,
    int skip_nulls, uint32_t min_count, uint64_t* __restrict__ out_bits,
    unsigned long long* __restrict__ valid_count) {
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwords = (n + 63) >> 6;
  uint64_t nvalid = 0;
  for (int64_t w = wave_g; w < nwords; w += nwaves) {
    const int64_t i = (w << 6) + lane;
    bool ok = false;
    if (i < n) {
      ok = counts[i] >= static_cast<int64_t>(min_count);
      if (!skip_nulls) ok = ok && (null_seen[i] & 1u) == 0;
    }
    const uint64_t bal = __ballot(ok);
    if (lane == 0) out_bits[w] = bal;
    nvalid += __popcll(bal);
  }
  if (valid_count != nullptr && lane == 0 && nvalid != 0) atomicAdd(valid_count, nvalid);
}

static inline unsigned gb_grid(int64_t n) {
  return static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, kBlock), 256 * 16)));
}

static int read_header(void* state, GroupbyHeader* h, hipStream_t st) {
  ARX_HIP(hipMemcpyAsync(h, state, sizeof(GroupbyHeader), hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  return ARX_OK;
}

int set_groupby_option(const char*, int64_t) { return 0; }

}  // namespace arx

using namespace arx;

extern "C" {

size_t arx_groupby_state_bytes(int64_t capacity) {
  if (capacity < 1) capacity = 1;
  const size_t c = static_cast<size_t>(capacity);
  return (64 + c * 8 + (c + 1) * 8 + (c + 1) * 8 + (c + 1) * 4 + 255) & ~size_t(255);
}

int arx_groupby_init(void* state, int64_t capacity, void* stream) {
  if (state == nullptr || capacity < 1 || (capacity & (capacity - 1)) != 0) {
    set_error("group-by capacity must be a power of two and state non-NULL");
    return ARX_INVALID;
  }
  if ((reinterpret_cast<uint64_t>(state) & 63) != 0) {
    set_error("group-by state must be 64-byte aligned");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  ARX_HIP(hipMemsetAsync(state, 0, arx_groupby_state_bytes(capacity), st));
  GroupbyHeader h{};
  h.capacity = capacity;
  ARX_HIP(hipMemcpyAsync(state, &h, sizeof(h), hipMemcpyHostToDevice, st));
  // the header struct lives on this stack frame: make sure the copy has consumed it
  ARX_HIP(hipStreamSynchronize(st));
  return ARX_OK;
}

static int state_capacity(void* state, int64_t* cap, hipStream_t st) {
  GroupbyHeader h{};
  const int rc = read_header(state, &h, st);
  if (rc != ARX_OK) return rc;
  if (h.capacity < 1 || (h.capacity & (h.capacity - 1)) != 0) {
    set_error("group-by state is not initialised");
    return ARX_INVALID;
  }
  *cap = h.capacity;
  return ARX_OK;
}

int arx_groupby_sum_i64_consume(void* state, int64_t capacity, const ArxSpan* keys_i32,
                                const ArxSpan* values_i64, void* stream) {
  if (state == nullptr || keys_i32 == nullptr || values_i64 == nullptr) {
    set_error("NULL argument to arx_groupby_sum_i64_consume");
    return ARX_INVALID;
  }
  if (keys_i32->length != values_i64->length) {
    set_error("Array arguments must all be the same length (keys %lld vs values %lld)",
              static_cast<long long>(keys_i32->length), static_cast<long long>(values_i64->length));
    return ARX_INVALID;
  }
  const int64_t n = keys_i32->length;
  if (n == 0) return ARX_OK;
  hipStream_t st = as_stream(stream);
  GroupbyView v = gb_view(state, capacity);
  const int32_t* k = static_cast<const int32_t*>(keys_i32->data) + keys_i32->offset;
  const int64_t* val = static_cast<const int64_t*>(values_i64->data) + values_i64->offset;
  const Bits kb = make_bits(keys_i32->null_count != 0 ? keys_i32->validity : nullptr,
                            keys_i32->offset, n);
  const Bits vb = make_bits(values_i64->null_count != 0 ? values_i64->validity : nullptr,
                            values_i64->offset, n);
  hipLaunchKernelGGL(groupby_consume_kernel, dim3(gb_grid(n)), dim3(kBlock), 0, st, v, k, kb, val, vb, n);
  ARX_CHECK_LAUNCH("groupby_consume_kernel");
  return ARX_OK;
}

int arx_groupby_sum_i64_merge(void* state, int64_t capacity, const int32_t* keys,
                              const uint8_t* key_is_valid, const int64_t* sums,
                              const int64_t* counts, const uint8_t* no_nulls, int64_t num_groups,
                              void* stream) {
  if (state == nullptr || num_groups < 0) {
    set_error("bad arguments to arx_groupby_sum_i64_merge");
    return ARX_INVALID;
  }
  if (num_groups == 0) return ARX_OK;
  if (keys == nullptr || sums == nullptr || counts == nullptr) {
    set_error("NULL partial-aggregate column");
    return ARX_INVALID;
  }
  GroupbyView v = gb_view(state, capacity);
  hipLaunchKernelGGL(groupby_merge_kernel, dim3(gb_grid(num_groups)), dim3(kBlock), 0,
                     as_stream(stream), v, keys, key_is_valid, sums, counts, no_nulls, num_groups);
  ARX_CHECK_LAUNCH("groupby_merge_kernel");
  return ARX_OK;
}

int arx_groupby_num_groups(void* state, int64_t* out_num_groups, void* stream) {
  if (state == nullptr || out_num_groups == nullptr) {
    set_error("NULL argument to arx_groupby_num_groups");
    return ARX_INVALID;
  }
  GroupbyHeader h{};
  const int rc = read_header(state, &h, as_stream(stream));
  if (rc != ARX_OK) return rc;
  if (h.overflow) {
    set_error("group-by hash table is full (capacity %lld): allocate a larger state",
              static_cast<long long>(h.capacity));
    return ARX_INVALID;
  }
  *out_num_groups = static_cast<int64_t>(h.num_groups);
  return ARX_OK;
}

int arx_groupby_sum_i64_export(void* state, int32_t* out_keys, uint8_t* out_key_is_valid,
                               int64_t* out_sums, int64_t* out_counts, uint8_t* out_no_nulls,
                               void* stream) {
  if (state == nullptr) {
    set_error("state is NULL");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  int64_t cap = 0;
  const int rc = state_capacity(state, &cap, st);
  if (rc != ARX_OK) return rc;
  GroupbyView v = gb_view(state, cap);
  ARX_HIP(hipMemsetAsync(&v.hdr->export_cursor, 0, sizeof(unsigned long long), st));
  hipLaunchKernelGGL(groupby_export_kernel, dim3(gb_grid(cap + 1)), dim3(kBlock), 0, st, v, out_keys,
                     out_key_is_valid, out_sums, out_counts, out_no_nulls);
  ARX_CHECK_LAUNCH("groupby_export_kernel");
  return ARX_OK;
}

int arx_groupby_sum_i64_finalize(const int64_t* counts, const uint8_t* no_nulls, int64_t num_groups,
                                 int skip_nulls, uint32_t min_count, uint8_t* out_valid,
                                 void* stream) {
  if (num_groups < 0) {
    set_error("negative num_groups");
    return ARX_INVALID;
  }
  if (num_groups == 0) return ARX_OK;
  if (counts == nullptr || out_valid == nullptr || (!skip_nulls && no_nulls == nullptr)) {
    set_error("NULL argument to arx_groupby_sum_i64_finalize");
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(groupby_finalize_kernel, dim3(gb_grid(num_groups)), dim3(kBlock), 0,
                     as_stream(stream), counts, no_nulls, num_groups, skip_nulls, min_count, out_valid);
  ARX_CHECK_LAUNCH("groupby_finalize_kernel");
  return ARX_OK;
}

int arx_hash_sum_i64_consume(const ArxSpan* values, int values_is_scalar, int64_t scalar_value,
                             const uint32_t* group_ids, int64_t length, int64_t* sums,
                             int64_t* counts, uint32_t* null_seen, void* stream) {
  if (values == nullptr || length < 0) {
    set_error("bad arguments to arx_hash_sum_i64_consume");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  if (group_ids == nullptr || sums == nullptr || counts == nullptr || null_seen == nullptr ||
      (!values_is_scalar && values->data == nullptr)) {
    set_error("NULL buffer passed to arx_hash_sum_i64_consume");
    return ARX_INVALID;
  }
  if (!values_is_scalar && values->length != length) {
    set_error("Array arguments must all be the same length (values %lld vs group ids %lld)",
              static_cast<long long>(values->length), static_cast<long long>(length));
    return ARX_INVALID;
  }
  const int64_t* v = values_is_scalar ? nullptr
                                      : static_cast<const int64_t*>(values->data) + values->offset;
  // a null scalar is described by null_count != 0 with a NULL bitmap: every row is null
  Bits vb = make_bits(values->null_count != 0 ? values->validity : nullptr, values->offset, length);
  int scalar_null = values_is_scalar && values->null_count != 0;
  hipStream_t st = as_stream(stream);
  if (scalar_null) {
    // all rows null: only the null_seen flags change; express as a zero-length validity
    vb.base = nullptr;
    vb.length = 0;  // load_word returns 0 for every word => every row reads as null
  }
  hipLaunchKernelGGL(hash_sum_dense_consume_kernel, dim3(gb_grid(length)), dim3(kBlock), 0, st, v,
                     scalar_value, values_is_scalar, vb, group_ids, length,
                     reinterpret_cast<unsigned long long*>(sums),
                     reinterpret_cast<unsigned long long*>(counts), null_seen);
  ARX_CHECK_LAUNCH("hash_sum_dense_consume_kernel");
  return ARX_OK;
}

int arx_hash_sum_i64_merge(int64_t* sums, int64_t* counts, uint32_t* null_seen,
                           const int64_t* other_sums, const int64_t* other_counts,
                           const uint32_t* other_null_seen, const uint32_t* group_id_mapping,
                           int64_t other_num_groups, void* stream) {
  if (other_num_groups < 0) {
    set_error("negative other_num_groups");
    return ARX_INVALID;
  }
  if (other_num_groups == 0) return ARX_OK;
  if (sums == nullptr || counts == nullptr || null_seen == nullptr || other_sums == nullptr ||
      other_counts == nullptr || other_null_seen == nullptr || group_id_mapping == nullptr) {
    set_error("NULL buffer passed to arx_hash_sum_i64_merge");
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(hash_sum_dense_merge_kernel, dim3(gb_grid(other_num_groups)), dim3(kBlock), 0,
                     as_stream(stream), other_sums, other_counts, other_null_seen, group_id_mapping,
                     other_num_groups, reinterpret_cast<unsigned long long*>(sums),
                     reinterpret_cast<unsigned long long*>(counts), null_seen);
  ARX_CHECK_LAUNCH("hash_sum_dense_merge_kernel");
  return ARX_OK;
}

int arx_hash_sum_i64_finalize(const int64_t* counts, const uint32_t* null_seen, int64_t num_groups,
                              int skip_nulls, uint32_t min_count, void* out_validity,
                              int64_t* valid_count, void* stream) {
  if (num_groups < 0) {
    set_error("negative num_groups");
    return ARX_INVALID;
  }
  if (num_groups == 0) return ARX_OK;
  if (counts == nullptr || null_seen == nullptr || out_validity == nullptr) {
    set_error("NULL buffer passed to arx_hash_sum_i64_finalize");
    return ARX_INVALID;
  }
  const int64_t nwords = ceil_div(num_groups, 64);
  const unsigned grid = static_cast<unsigned>(
      std::max<int64_t>(1, std::min<int64_t>(ceil_div(nwords, kWavesPerBlock), 2048)));
  hipLaunchKernelGGL(hash_sum_dense_finalize_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream),
                     counts, null_seen, num_groups, skip_nulls, min_count,
                     static_cast<uint64_t*>(out_validity),
                     reinterpret_cast<unsigned long long*>(valid_count));
  ARX_CHECK_LAUNCH("hash_sum_dense_finalize_kernel");
  return ARX_OK;
}

size_t arx_groupby_partition_workspace_bytes(int num_parts) {
  if (num_parts < 1) num_parts = 1;
  return static_cast<size_t>(num_parts) * 16 + 64;
}

int arx_groupby_partition(const int32_t* keys, const uint8_t* key_is_valid, const int64_t* sums,
                          const int64_t* counts, const uint8_t* no_nulls, int64_t num_groups,
                          int num_parts, void* ws, size_t ws_bytes, int32_t* out_keys,
                          uint8_t* out_key_is_valid, int64_t* out_sums, int64_t* out_counts,
                          uint8_t* out_no_nulls, int64_t* out_part_counts, void* stream) {
  if (num_parts < 1 || num_parts > 1024 || num_groups < 0 || ws == nullptr ||
      ws_bytes < arx_groupby_partition_workspace_bytes(num_parts) || out_part_counts == nullptr) {
    set_error("bad arguments to arx_groupby_partition");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  unsigned long long* part_counts = static_cast<unsigned long long*>(ws);
  unsigned long long* cursors = part_counts + num_parts;
  ARX_HIP(hipMemsetAsync(ws, 0, static_cast<size_t>(num_parts) * 16, st));
  if (num_groups > 0) {
    hipLaunchKernelGGL(partition_count_kernel, dim3(gb_grid(num_groups)), dim3(kBlock), 0, st, keys,
                       key_is_valid, num_groups, num_parts, part_counts);
    ARX_CHECK_LAUNCH("partition_count_kernel");
  }
  hipLaunchKernelGGL(partition_offsets_kernel, dim3(1), dim3(64), 0, st, part_counts, num_parts,
                     cursors, out_part_counts);
  ARX_CHECK_LAUNCH("partition_offsets_kernel");
  if (num_groups > 0) {
    hipLaunchKernelGGL(partition_scatter_kernel, dim3(gb_grid(num_groups)), dim3(kBlock), 0, st, keys,
                       key_is_valid, sums, counts, no_nulls, num_groups, num_parts, cursors, out_keys,
                       out_key_is_valid, out_sums, out_counts, out_no_nulls);
    ARX_CHECK_LAUNCH("partition_scatter_kernel");
  }
  return ARX_OK;
}

}  // extern "C"
