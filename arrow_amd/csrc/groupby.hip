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

#include <string.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <type_traits>
#include <vector>

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

// Adds this thread's count of newly inserted keys to the header: one atomic per wave (a per-key
// atomic on the single counter serialises ~1 ns apiece — 10 ms for 10M new groups).
__device__ __forceinline__ void gb_publish_new_groups(const GroupbyView& v, uint32_t mine) {
  const uint32_t tot = wave_reduce_sum_u32(mine);
  if (lane_id() == 0 && tot != 0) atomicAdd(&v.hdr->num_groups, static_cast<unsigned long long>(tot));
}

// Slot of `key`, inserting it if absent (*inserted += 1 then; the caller publishes the total with
// gb_publish_new_groups).  Returns -1 if the table is full.
__device__ __forceinline__ int64_t gb_find_or_insert(const GroupbyView& v, int32_t key,
                                                     uint32_t* inserted) {
  const unsigned long long tagged = (1ull << 32) | static_cast<uint32_t>(key);
  const uint64_t mask = static_cast<uint64_t>(v.capacity) - 1;
  uint64_t h = v.lg == 0 ? 0 : (gb_hash(key) >> (64 - v.lg));
  for (int64_t probes = 0; probes < v.capacity; ++probes) {
    unsigned long long cur = __hip_atomic_load(&v.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == tagged) return static_cast<int64_t>(h);
    if (cur == 0) {
      const unsigned long long old = atomicCAS(&v.keys[h], 0ull, tagged);
      if (old == 0) {
        *inserted += 1;
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
  uint32_t fresh = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool kv = (load_word(kvalid, i >> 6) >> (i & 63)) & 1ull;
    const bool vv = (load_word(vvalid, i >> 6) >> (i & 63)) & 1ull;
    const int64_t slot = kv ? gb_find_or_insert(v, keys[i], &fresh) : gb_null_slot(v);
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
  gb_publish_new_groups(v, fresh);
}

__global__ __launch_bounds__(kBlock) void groupby_merge_kernel(
    GroupbyView v, const int32_t* __restrict__ keys, const uint8_t* __restrict__ key_is_valid,
    const int64_t* __restrict__ sums, const int64_t* __restrict__ counts,
    const uint8_t* __restrict__ no_nulls, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  uint32_t fresh = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool kv = key_is_valid == nullptr || key_is_valid[i] != 0;
    const int64_t slot = kv ? gb_find_or_insert(v, keys[i], &fresh) : gb_null_slot(v);
    if (slot < 0) {
      atomicExch(&v.hdr->overflow, 1u);
      continue;
    }
    atomicAdd(&v.sums[slot], static_cast<unsigned long long>(sums[i]));
    atomicAdd(&v.counts[slot], static_cast<unsigned long long>(counts[i]));
    if (no_nulls != nullptr && no_nulls[i] == 0) atomicOr(&v.flags[slot], 1u);
  }
  gb_publish_new_groups(v, fresh);
}

// One workgroup per 4096 consecutive slots: count the occupied ones, reserve the output range
// with ONE atomic per workgroup, then write them out (order unspecified, as for the reference).
constexpr int kExportSlotsPerBlock = kBlock * 16;

__global__ __launch_bounds__(kBlock) void groupby_export_kernel(
    GroupbyView v, int32_t* __restrict__ out_keys, uint8_t* __restrict__ out_key_is_valid,
    int64_t* __restrict__ out_sums, int64_t* __restrict__ out_counts,
    uint8_t* __restrict__ out_no_nulls, const long long* __restrict__ mins,
    const long long* __restrict__ maxs, int64_t* __restrict__ out_mins, int64_t* __restrict__ out_maxs) {
  __shared__ uint32_t wave_tot[kWavesPerBlock];
  __shared__ unsigned long long base_s;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t s0 = static_cast<int64_t>(blockIdx.x) * kExportSlotsPerBlock;
  unsigned long long tagged[16];
  uint32_t mine = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int64_t sl = s0 + i * kBlock + tid;
    tagged[i] = 0;
    if (sl < v.capacity) {
      tagged[i] = v.keys[sl];
    } else if (sl == v.capacity) {
      tagged[i] = v.hdr->null_used != 0 ? 1ull : 0ull;  // any non-zero marker
    }
    mine += tagged[i] != 0 ? 1u : 0u;
  }
  const uint32_t incl = wave_inclusive_scan_u32(mine);
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t pre = incl - mine;
  uint32_t total = 0;
  for (int k = 0; k < kWavesPerBlock; ++k) {
    if (k < wave) pre += wave_tot[k];
    total += wave_tot[k];
  }
  if (total == 0) return;  // workgroup-uniform
  if (tid == 0) base_s = atomicAdd(&v.hdr->export_cursor, static_cast<unsigned long long>(total));
  __syncthreads();
  int64_t pos = static_cast<int64_t>(base_s) + pre;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (tagged[i] == 0) continue;
    const int64_t sl = s0 + i * kBlock + tid;
    const bool is_key = sl < v.capacity;
    out_keys[pos] = is_key ? static_cast<int32_t>(static_cast<uint32_t>(tagged[i])) : 0;
    out_key_is_valid[pos] = is_key ? 1 : 0;
    out_sums[pos] = static_cast<int64_t>(v.sums[sl]);
    out_counts[pos] = static_cast<int64_t>(v.counts[sl]);
    out_no_nulls[pos] = (v.flags[sl] & 1u) ? 0 : 1;
    if (out_mins != nullptr) {  // same position as the other columns: one export, aligned columns
      out_mins[pos] = mins[sl];
      out_maxs[pos] = maxs[sl];
    }
    ++pos;
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

// Both kernels give every workgroup a contiguous chunk of kPartChunk rows and keep the
// per-destination counters in LDS: one global atomic per (workgroup, destination) instead of one
// per (wave, destination) — with 10M partial aggregates and 8 destinations the latter is
// 1.25M atomics on 8 addresses (~5 ms each pass), the former 20K.
constexpr int kPartChunk = kBlock * 32;
constexpr int kPartMaxParts = 1024;

__global__ __launch_bounds__(kBlock) void partition_count_kernel(const int32_t* __restrict__ keys,
                                                                 const uint8_t* __restrict__ key_is_valid,
                                                                 int64_t n, int num_parts,
                                                                 unsigned long long* part_counts) {
  __shared__ uint32_t cnt[kPartMaxParts];
  for (int p = threadIdx.x; p < num_parts; p += kBlock) cnt[p] = 0;
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * kPartChunk;
  const int64_t end = begin + kPartChunk < n ? begin + kPartChunk : n;
  for (int64_t i = begin + threadIdx.x; i < end; i += kBlock) {
    const int d = gb_dest(keys[i], key_is_valid == nullptr || key_is_valid[i] != 0, num_parts);
    atomicAdd(&cnt[d], 1u);
  }
  __syncthreads();
  for (int p = threadIdx.x; p < num_parts; p += kBlock) {
    if (cnt[p] != 0) atomicAdd(&part_counts[p], static_cast<unsigned long long>(cnt[p]));
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
  __shared__ uint32_t cnt[kPartMaxParts];
  __shared__ unsigned long long base[kPartMaxParts];
  for (int p = threadIdx.x; p < num_parts; p += kBlock) cnt[p] = 0;
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * kPartChunk;
  const int64_t end = begin + kPartChunk < n ? begin + kPartChunk : n;
  // pass 1 over the chunk: how many rows go to each destination
  for (int64_t i = begin + threadIdx.x; i < end; i += kBlock) {
    const int d = gb_dest(keys[i], key_is_valid == nullptr || key_is_valid[i] != 0, num_parts);
    atomicAdd(&cnt[d], 1u);
  }
  __syncthreads();
  // reserve this workgroup's output ranges, then reuse cnt[] as the running cursor inside them
  for (int p = threadIdx.x; p < num_parts; p += kBlock) {
    base[p] = cnt[p] != 0 ? atomicAdd(&cursors[p], static_cast<unsigned long long>(cnt[p])) : 0ull;
    cnt[p] = 0;
  }
  __syncthreads();
  // pass 2 (the chunk is L2-resident): write the rows (order inside a destination is unspecified)
  for (int64_t i = begin + threadIdx.x; i < end; i += kBlock) {
    const bool kv = key_is_valid == nullptr || key_is_valid[i] != 0;
    const int d = gb_dest(keys[i], kv, num_parts);
    const int64_t pos = static_cast<int64_t>(base[d]) + atomicAdd(&cnt[d], 1u);
    out_keys[pos] = keys[i];
    out_key_is_valid[pos] = kv ? 1 : 0;
    out_sums[pos] = sums[i];
    out_counts[pos] = counts[i];
    out_no_nulls[pos] = no_nulls == nullptr ? 1 : no_nulls[i];
  }
}

// ---- the ROW-level exchange (SURVEY.md 8e "row-level radix exchange", the variant BASELINE.json's north star words):
// the rows themselves leave for the rank that owns their key — 16-byte records {key, flags, value}, flags bit 0 = key
// valid, bit 1 = value valid — instead of one partial aggregate per local group.  Better than the partials exchange
// only when nearly every row is its own group (G -> N / P); otherwise it moves N x 16 B instead of G x 24 B.
struct ArxRowRecordDev {
  int32_t key;
  uint32_t flags;
  int64_t value;
};
static_assert(sizeof(ArxRowRecordDev) == 16, "16-byte row records");

template <bool SCATTER>
__global__ __launch_bounds__(kBlock) void partition_rows_kernel(const int32_t* __restrict__ keys, Bits kvalid,
                                                                const int64_t* __restrict__ values, Bits vvalid, int64_t n,
                                                                int num_parts, unsigned long long* part_counts,
                                                                unsigned long long* cursors, ArxRowRecordDev* __restrict__ out) {
  __shared__ uint32_t cnt[kPartMaxParts];
  __shared__ unsigned long long base[kPartMaxParts];
  for (int p = threadIdx.x; p < num_parts; p += kBlock) cnt[p] = 0;
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * kPartChunk;
  const int64_t end = begin + kPartChunk < n ? begin + kPartChunk : n;
  for (int64_t i = begin + threadIdx.x; i < end; i += kBlock) {
    const bool kv = (load_word(kvalid, i >> 6) >> (i & 63)) & 1ull;
    atomicAdd(&cnt[gb_dest(keys[i], kv, num_parts)], 1u);
  }
  __syncthreads();
  if (!SCATTER) {
    for (int p = threadIdx.x; p < num_parts; p += kBlock) {
      if (cnt[p] != 0) atomicAdd(&part_counts[p], static_cast<unsigned long long>(cnt[p]));
    }
    return;
  }
  for (int p = threadIdx.x; p < num_parts; p += kBlock) {
    base[p] = cnt[p] != 0 ? atomicAdd(&cursors[p], static_cast<unsigned long long>(cnt[p])) : 0ull;
    cnt[p] = 0;
  }
  __syncthreads();
  for (int64_t i = begin + threadIdx.x; i < end; i += kBlock) {
    const bool kv = (load_word(kvalid, i >> 6) >> (i & 63)) & 1ull;
    const bool vv = (load_word(vvalid, i >> 6) >> (i & 63)) & 1ull;
    const int d = gb_dest(keys[i], kv, num_parts);
    const int64_t pos = static_cast<int64_t>(base[d]) + atomicAdd(&cnt[d], 1u);
    out[pos] = ArxRowRecordDev{kv ? keys[i] : 0, (kv ? 1u : 0u) | (vv ? 2u : 0u), vv ? values[i] : 0};
  }
}

// received row records -> key / value columns + their validity bitmaps (one ballot word per 64 rows)
__global__ __launch_bounds__(kBlock) void unpack_rows_kernel(const ArxRowRecordDev* __restrict__ recs, int64_t n,
                                                             int32_t* __restrict__ keys, int64_t* __restrict__ values,
                                                             uint64_t* __restrict__ kbits, uint64_t* __restrict__ vbits) {
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwords = (n + 63) >> 6;
  for (int64_t w = wave_g; w < nwords; w += nwaves) {
    const int64_t i = (w << 6) + lane;
    bool kv = false, vv = false;
    if (i < n) {
      const ArxRowRecordDev r = recs[i];
      keys[i] = r.key;
      values[i] = r.value;
      kv = (r.flags & 1u) != 0;
      vv = (r.flags & 2u) != 0;
    }
    const uint64_t kb = __ballot(kv), vb = __ballot(vv);
    if (lane == 0) {
      kbits[w] = kb;
      vbits[w] = vb;
    }
  }
}

// ---- export straight into the multi-GPU exchange layout: one 24-byte record per group, grouped by
// destination rank.  Two sweeps over the table slots (count per destination, then
// reserve-and-write per workgroup), so the dense column export and its re-read are never materialised.
__device__ __forceinline__ bool gb_slot_group(const GroupbyView& v, int64_t sl, int32_t* key, bool* key_valid) {
  if (sl < v.capacity) {
    const unsigned long long t = v.keys[sl];
    *key = static_cast<int32_t>(static_cast<uint32_t>(t));
    *key_valid = true;
    return t != 0;
  }
  *key = 0;
  *key_valid = false;
  return sl == v.capacity && v.hdr->null_used != 0;
}

__global__ __launch_bounds__(kBlock) void export_part_count_kernel(GroupbyView v, int num_parts,
                                                                   unsigned long long* part_counts) {
  __shared__ uint32_t cnt[kPartMaxParts];
  for (int p = threadIdx.x; p < num_parts; p += kBlock) cnt[p] = 0;
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * kPartChunk;
  for (int64_t sl = begin + threadIdx.x; sl < begin + kPartChunk && sl <= v.capacity; sl += kBlock) {
    int32_t key;
    bool kv;
    if (gb_slot_group(v, sl, &key, &kv)) atomicAdd(&cnt[gb_dest(key, kv, num_parts)], 1u);
  }
  __syncthreads();
  for (int p = threadIdx.x; p < num_parts; p += kBlock) {
    if (cnt[p] != 0) atomicAdd(&part_counts[p], static_cast<unsigned long long>(cnt[p]));
  }
}

__global__ __launch_bounds__(kBlock) void export_part_scatter_kernel(GroupbyView v, int num_parts,
                                                                     unsigned long long* cursors,
                                                                     ArxGroupPartial* __restrict__ out) {
  __shared__ uint32_t cnt[kPartMaxParts];
  __shared__ unsigned long long base[kPartMaxParts];
  for (int p = threadIdx.x; p < num_parts; p += kBlock) cnt[p] = 0;
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * kPartChunk;
  for (int64_t sl = begin + threadIdx.x; sl < begin + kPartChunk && sl <= v.capacity; sl += kBlock) {
    int32_t key;
    bool kv;
    if (gb_slot_group(v, sl, &key, &kv)) atomicAdd(&cnt[gb_dest(key, kv, num_parts)], 1u);
  }
  __syncthreads();
  for (int p = threadIdx.x; p < num_parts; p += kBlock) {
    base[p] = cnt[p] != 0 ? atomicAdd(&cursors[p], static_cast<unsigned long long>(cnt[p])) : 0ull;
    cnt[p] = 0;
  }
  __syncthreads();
  for (int64_t sl = begin + threadIdx.x; sl < begin + kPartChunk && sl <= v.capacity; sl += kBlock) {
    int32_t key;
    bool kv;
    if (!gb_slot_group(v, sl, &key, &kv)) continue;
    const int d = gb_dest(key, kv, num_parts);
    const int64_t pos = static_cast<int64_t>(base[d]) + atomicAdd(&cnt[d], 1u);
    ArxGroupPartial r;
    r.sum = static_cast<int64_t>(v.sums[sl]);
    r.count = static_cast<int64_t>(v.counts[sl]);
    r.key = key;
    r.key_is_valid = kv ? 1 : 0;
    r.no_nulls = (v.flags[sl] & 1u) ? 0 : 1;
    r.pad[0] = r.pad[1] = 0;
    out[pos] = r;
  }
}

__global__ __launch_bounds__(kBlock) void groupby_merge_records_kernel(GroupbyView v,
                                                                       const ArxGroupPartial* __restrict__ rec,
                                                                       int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  uint32_t fresh = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const ArxGroupPartial r = rec[i];
    const int64_t slot = r.key_is_valid ? gb_find_or_insert(v, r.key, &fresh) : gb_null_slot(v);
    if (slot < 0) {
      atomicExch(&v.hdr->overflow, 1u);
      continue;
    }
    atomicAdd(&v.sums[slot], static_cast<unsigned long long>(r.sum));
    atomicAdd(&v.counts[slot], static_cast<unsigned long long>(r.count));
    if (r.no_nulls == 0) atomicOr(&v.flags[slot], 1u);
  }
  gb_publish_new_groups(v, fresh);
}

// ---- hash_mean(int64): GroupedMeanImpl (kernels/hash_aggregate_numeric.cc:352-430) accumulates DOUBLES in row
// order (GroupedMeanAccType: every number type sums in double), which a parallel reduction cannot reproduce bit for
// bit in general.  It can whenever every partial sum of a group is exactly representable: then double addition is
// exact in any order and equals the int64 sum.  |partial sums| <= count * max(|min|, |max|), so with that product
// below 2^53 the mean is (double)sum / count exactly as the reference computes it (DoMean :397-400); a group that
// breaks the bound raises *inexact and the caller declines with the reason.
__global__ __launch_bounds__(kBlock) void groupby_mean_finalize_kernel(
    const int64_t* __restrict__ sums, const int64_t* __restrict__ counts, const int64_t* __restrict__ mins,
    const int64_t* __restrict__ maxs, const uint8_t* __restrict__ no_nulls, int64_t n, int skip_nulls,
    uint32_t min_count, double* __restrict__ out_means, uint8_t* __restrict__ out_valid, unsigned int* inexact) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  bool bad = false;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t c = counts[i];
    bool ok = c >= static_cast<int64_t>(min_count);
    // Finish :402-420: the mean is computed wherever count >= min_count (0 / 0 = NaN with min_count = 0), 0 elsewhere
    out_means[i] = ok ? static_cast<double>(sums[i]) / static_cast<double>(c) : 0.0;
    if (!skip_nulls) ok = ok && no_nulls[i] != 0;
    out_valid[i] = ok ? 1 : 0;
    if (c > 0) {
      const int64_t lo = mins[i], hi = maxs[i];
      const uint64_t alo = lo < 0 ? static_cast<uint64_t>(0) - static_cast<uint64_t>(lo) : static_cast<uint64_t>(lo);
      const uint64_t ahi = hi < 0 ? static_cast<uint64_t>(0) - static_cast<uint64_t>(hi) : static_cast<uint64_t>(hi);
      const uint64_t amax = alo > ahi ? alo : ahi;
      const uint64_t uc = static_cast<uint64_t>(c);
      // count * amax < 2^53 without a 128-bit product: both factors below 2^53 and amax < 2^53 / count
      if (uc >= (uint64_t(1) << 53) || amax >= (uint64_t(1) << 53) / uc + (((uint64_t(1) << 53) % uc) != 0 ? 1 : 0)) bad = true;
    }
  }
  if (__any(bad) && lane_id() == 0) atomicOr(inexact, 1u);
}

// ---- lookup: out[i] = the sum column of keys[i]'s group, -1 if the key is not in the table
// (read-only probe; what dictionary_encode uses to turn rows into dictionary indices once the
// groups' dense ids have been merged in as their "sums")
__global__ __launch_bounds__(kBlock) void groupby_lookup_kernel(GroupbyView v, const int32_t* __restrict__ keys,
                                                                Bits kvalid, int64_t n, int32_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const uint64_t mask = static_cast<uint64_t>(v.capacity) - 1;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool kv = (load_word(kvalid, i >> 6) >> (i & 63)) & 1ull;
    int64_t slot = -1;
    if (!kv) {
      slot = v.hdr->null_used != 0 ? v.capacity : -1;
    } else {
      const int32_t key = keys[i];
      const unsigned long long tagged = (1ull << 32) | static_cast<uint32_t>(key);
      uint64_t h = v.lg == 0 ? 0 : (gb_hash(key) >> (64 - v.lg));
      for (int64_t probes = 0; probes < v.capacity; ++probes) {
        const unsigned long long cur = v.keys[h];
        if (cur == tagged) {
          slot = static_cast<int64_t>(h);
          break;
        }
        if (cur == 0) break;
        h = (h + 1) & mask;
      }
    }
    out[i] = slot < 0 ? -1 : static_cast<int32_t>(v.sums[slot]);
  }
}

// ---- hash_min / hash_max on the same table (GroupedMinMaxImpl, kernels/hash_aggregate.cc:330-419):
// mins start at INT64_MAX, maxes at INT64_MIN (AntiExtrema, :349-350); a valid value folds into both,
// a null value sets the group's null flag (the same flag hash_sum keeps); the group exists either
// way.  Integer min/max are associative and commutative: device atomics in any order are bit-exact.
// "has_values" is not stored: a group saw a value iff min <= max.
__global__ __launch_bounds__(kBlock) void groupby_minmax_init_kernel(long long* __restrict__ mins,
                                                                     long long* __restrict__ maxs, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    mins[i] = INT64_MAX;
    maxs[i] = INT64_MIN;
  }
}

__global__ __launch_bounds__(kBlock) void groupby_minmax_consume_kernel(GroupbyView v, long long* __restrict__ mins,
                                                                        long long* __restrict__ maxs,
                                                                        const int32_t* __restrict__ keys, Bits kvalid,
                                                                        const int64_t* __restrict__ values,
                                                                        Bits vvalid, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  uint32_t fresh = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool kv = (load_word(kvalid, i >> 6) >> (i & 63)) & 1ull;
    const bool vv = (load_word(vvalid, i >> 6) >> (i & 63)) & 1ull;
    const int64_t slot = kv ? gb_find_or_insert(v, keys[i], &fresh) : gb_null_slot(v);
    if (slot < 0) {
      atomicExch(&v.hdr->overflow, 1u);
      continue;
    }
    if (vv) {
      const long long x = values[i];
      // a plain read first: once a group's extrema have settled most rows change nothing
      if (x < __hip_atomic_load(&mins[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&mins[slot], x);
      if (x > __hip_atomic_load(&maxs[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxs[slot], x);
    } else {
      atomicOr(&v.flags[slot], 1u);
    }
  }
  gb_publish_new_groups(v, fresh);
}

// Merge (hash_aggregate.cc:371-399): extrema of another state's groups fold into this one
__global__ __launch_bounds__(kBlock) void groupby_minmax_merge_kernel(
    GroupbyView v, long long* __restrict__ mins, long long* __restrict__ maxs,
    const int32_t* __restrict__ keys, const uint8_t* __restrict__ key_is_valid,
    const int64_t* __restrict__ other_mins, const int64_t* __restrict__ other_maxs,
    const uint8_t* __restrict__ no_nulls, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  uint32_t fresh = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool kv = key_is_valid == nullptr || key_is_valid[i] != 0;
    const int64_t slot = kv ? gb_find_or_insert(v, keys[i], &fresh) : gb_null_slot(v);
    if (slot < 0) {
      atomicExch(&v.hdr->overflow, 1u);
      continue;
    }
    atomicMin(&mins[slot], static_cast<long long>(other_mins[i]));
    atomicMax(&maxs[slot], static_cast<long long>(other_maxs[i]));
    if (no_nulls != nullptr && no_nulls[i] == 0) atomicOr(&v.flags[slot], 1u);
  }
  gb_publish_new_groups(v, fresh);
}

// valid = the group saw a value (and, unless skip_nulls, no null): Finalize, hash_aggregate.cc:401-410
// (min_count is not consulted by the reference's min/max)
__global__ __launch_bounds__(kBlock) void groupby_minmax_finalize_kernel(const int64_t* __restrict__ mins,
                                                                         const int64_t* __restrict__ maxs,
                                                                         const uint8_t* __restrict__ no_nulls,
                                                                         int64_t n, int skip_nulls,
                                                                         uint8_t* __restrict__ out_valid) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    bool ok = mins[i] <= maxs[i];
    if (!skip_nulls) ok = ok && no_nulls[i] != 0;
    out_valid[i] = ok ? 1 : 0;
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

// hash_mean / hash_count over the dense per-group state of the hash_sum vtable: means[g] = double(sums[g]) / counts[g]
// (one correctly rounded division), which is what GroupedMeanImpl's row-order double accumulation gives whenever every
// partial sum is an integer below 2^53 — true when counts[g] * abs_bound < 2^53 (abs_bound >= |value| of every row);
// a group where that does not hold sets *inexact.
__global__ __launch_bounds__(kBlock) void hash_mean_dense_finalize_kernel(
    const int64_t* __restrict__ sums, const int64_t* __restrict__ counts, int64_t n, uint64_t abs_bound,
    double* __restrict__ means, unsigned int* __restrict__ inexact) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    const int64_t c = counts[i];
    double m = 0.0;
    if (c > 0) {
      // c * abs_bound < 2^53  <=>  abs_bound <= (2^53 - 1) / c
      if (abs_bound > ((uint64_t(1) << 53) - 1) / static_cast<uint64_t>(c)) atomicOr(inexact, 1u);
      m = static_cast<double>(sums[i]) / static_cast<double>(c);
    }
    means[i] = m;
  }
}

// hash_min / hash_max / hash_min_max over dense group ids — GroupedMinMaxImpl (kernels/hash_aggregate.cc:330-419):
// per group a running minimum and maximum (initialised to the anti-extrema by Resize, :343-353), "saw a null".  A group
// that saw no value keeps min = INT64_MAX > max = INT64_MIN, which is how Finalize tells (has_values_).  A plain read
// comes first and the atomic only runs when it would change something — after a group's first few rows almost none does.
__global__ __launch_bounds__(kBlock) void hash_minmax_dense_fill_kernel(long long* __restrict__ mins, long long* __restrict__ maxs,
                                                                        int64_t first, int64_t count) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < count; i += stride) {
    mins[first + i] = INT64_MAX;
    maxs[first + i] = INT64_MIN;
  }
}

__global__ __launch_bounds__(kBlock) void hash_minmax_dense_consume_kernel(
    const int64_t* __restrict__ values, int64_t scalar_value, int values_is_scalar, Bits vvalid,
    const uint32_t* __restrict__ group_ids, int64_t n, long long* __restrict__ mins, long long* __restrict__ maxs,
    unsigned int* __restrict__ null_seen) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t g = group_ids[i];
    const bool ok = (load_word(vvalid, i >> 6) >> (i & 63)) & 1ull;
    if (ok) {
      const long long v = values_is_scalar ? scalar_value : values[i];
      if (v < __hip_atomic_load(&mins[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&mins[g], v);
      if (v > __hip_atomic_load(&maxs[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxs[g], v);
    } else if ((__hip_atomic_load(&null_seen[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) == 0) {
      atomicOr(&null_seen[g], 1u);
    }
  }
}

__global__ __launch_bounds__(kBlock) void hash_minmax_dense_merge_kernel(
    const int64_t* __restrict__ other_mins, const int64_t* __restrict__ other_maxs,
    const uint32_t* __restrict__ other_null_seen, const uint32_t* __restrict__ mapping, int64_t n,
    long long* __restrict__ mins, long long* __restrict__ maxs, unsigned int* __restrict__ null_seen) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t g = mapping[i];
    atomicMin(&mins[g], static_cast<long long>(other_mins[i]));
    atomicMax(&maxs[g], static_cast<long long>(other_maxs[i]));
    if (other_null_seen[i] & 3u) atomicOr(&null_seen[g], other_null_seen[i] & 3u);   // bit 0: saw a null; bit 1 (float states): saw a value
  }
}

// ---- float32 / float64 extrema (GroupedMinMaxImpl with MinMaxOp = fmin / fmax and NaN anti-extrema,
// kernels/hash_aggregate.cc:306-326): the state arrays hold the values' ORDER KEYS — the int64 whose signed order is the
// numeric order of the doubles (-0.0 just below +0.0, the one tie the reference leaves to row order) — so the integer
// atomics and the merge kernel above serve them unchanged.  fmin / fmax skip NaNs: a NaN row only marks "saw a value"
// (bit 1 of null_seen; mins <= maxs cannot say it for an all-NaN group), and Finalize turns the untouched anti-extrema
// INT64_MAX / INT64_MIN — themselves the keys of NaN patterns — into the canonical quiet NaN, which is what fmin(NaN, NaN) leaves.
template <typename T>
__global__ __launch_bounds__(kBlock) void hash_minmax_dense_consume_float_kernel(
    const T* __restrict__ values, double scalar_value, int values_is_scalar, Bits vvalid,
    const uint32_t* __restrict__ group_ids, int64_t n, long long* __restrict__ mins, long long* __restrict__ maxs,
    unsigned int* __restrict__ null_seen) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t g = group_ids[i];
    const bool ok = (load_word(vvalid, i >> 6) >> (i & 63)) & 1ull;
    const unsigned int seen = __hip_atomic_load(&null_seen[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ok) {
      const double d = values_is_scalar ? scalar_value : static_cast<double>(values[i]);   // (float -> double is exact and keeps the order)
      if ((seen & 2u) == 0) atomicOr(&null_seen[g], 2u);
      if (d == d) {
        const long long v = float_order_key(d);
        if (v < __hip_atomic_load(&mins[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&mins[g], v);
        if (v > __hip_atomic_load(&maxs[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxs[g], v);
      }
    } else if ((seen & 1u) == 0) {
      atomicOr(&null_seen[g], 1u);
    }
  }
}

// values and validity of the float extrema: bit g = group g saw a value && (skip_nulls || saw no null)
template <typename T>
__global__ __launch_bounds__(kBlock) void hash_minmax_dense_finalize_float_kernel(
    const int64_t* __restrict__ mins, const int64_t* __restrict__ maxs, const uint32_t* __restrict__ null_seen, int64_t n,
    int skip_nulls, T* __restrict__ out_mins, T* __restrict__ out_maxs, uint64_t* __restrict__ out_bits,
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
      const uint32_t seen = null_seen[i];
      ok = (seen & 2u) != 0 && (skip_nulls || (seen & 1u) == 0);
      if (out_mins != nullptr) out_mins[i] = static_cast<T>(float_from_order_key(mins[i]));
      if (out_maxs != nullptr) out_maxs[i] = static_cast<T>(float_from_order_key(maxs[i]));
    }
    const uint64_t bal = __ballot(ok);
    if (lane == 0) out_bits[w] = bal;
    nvalid += __popcll(bal);
  }
  if (valid_count != nullptr && lane == 0 && nvalid != 0) atomicAdd(valid_count, nvalid);
}

// bit g = group g saw a value && (skip_nulls || saw no null)   (Finalize, hash_aggregate.cc:401-419)
__global__ __launch_bounds__(kBlock) void hash_minmax_dense_finalize_kernel(
    const int64_t* __restrict__ mins, const int64_t* __restrict__ maxs, const uint32_t* __restrict__ null_seen, int64_t n,
    int skip_nulls, uint64_t* __restrict__ out_bits, unsigned long long* __restrict__ valid_count) {
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwords = (n + 63) >> 6;
  uint64_t nvalid = 0;
  for (int64_t w = wave_g; w < nwords; w += nwaves) {
    const int64_t i = (w << 6) + lane;
    bool ok = false;
    if (i < n) {
      ok = mins[i] <= maxs[i];
      if (!skip_nulls) ok = ok && (null_seen[i] & 1u) == 0;
    }
    const uint64_t bal = __ballot(ok);
    if (lane == 0) out_bits[w] = bal;
    nvalid += __popcll(bal);
  }
  if (valid_count != nullptr && lane == 0 && nvalid != 0) atomicAdd(valid_count, nvalid);
}

// hash_count over dense group ids — GroupedCountImpl::Consume (kernels/hash_aggregate.cc:107-212): counts[g] += 1 for
// every row of group g that is valid (ONLY_VALID), null (ONLY_NULL) or either (ALL); the value type does not matter.
__global__ __launch_bounds__(kBlock) void hash_count_dense_consume_kernel(Bits vvalid, int mode, const uint32_t* __restrict__ group_ids,
                                                                          int64_t n, unsigned long long* __restrict__ counts) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const bool ok = (load_word(vvalid, i >> 6) >> (i & 63)) & 1ull;
    if (mode == 2 || (mode == 0) == ok) atomicAdd(&counts[group_ids[i]], 1ull);
  }
}

__global__ __launch_bounds__(kBlock) void hash_count_dense_merge_kernel(const int64_t* __restrict__ other_counts,
                                                                        const uint32_t* __restrict__ mapping, int64_t n,
                                                                        unsigned long long* __restrict__ counts) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    atomicAdd(&counts[mapping[i]], static_cast<unsigned long long>(other_counts[i]));
  }
}

static inline unsigned gb_grid(int64_t n) {
  return static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, kBlock), 256 * 16)));
}

// =====================================================================================
// Radix-partitioned consume (the fast path of arx_groupby_sum_i64_consume).
//
// Device-scope atomics into the HBM table top out at ~12 Grows/s whatever the table size
// (profiles/: exp_groupby_atomics) — the atomic units, not HBM, are the bound.  So rows are first
// partitioned by the top `bits` bits of a bijective 32-bit hash of the key until one partition's
// groups fit an LDS table (4096 slots, 64 KiB), aggregated there with LDS atomics, and only the
// per-partition partial aggregates (one row per group) touch the HBM table:
//   K0 gbp_null_rows    rows with a null key or a null value take the slow path (HBM atomics)
//   K1 gbp_hist         per-chunk level-1 digit counts + global per-partition counts  (4 B/row)
//   K2 gbp_scan_a/_b    partition starts, level-1 chunk offsets, level-2 cursors, tile map
//   K3 gbp_scatter<1>   level 1: chunked, exact offsets (no atomics)            (12 + 12 B/row)
//   K4 gbp_scatter<2>   level 2 inside every level-1 partition, one returning atomic per
//                       (tile, digit) on a cursor only that partition's tiles touch (12 + 12 B/row)
//   K5 gbp_aggregate    one workgroup per partition: LDS open-addressing table keyed by the low
//                       hash bits (the hash is a bijection, so low bits + partition id = key),
//                       ds_add_u64 sums / ds_add_u32 counts, then a flush of <= 4096 partials
//                       into the HBM table                                           (12 B/row)
// Row order inside a partition is irrelevant (integer sums commute), so ranking uses LDS atomics.
// =====================================================================================
constexpr int kGbThreads = 512;
constexpr int kGbTile = 4096;             // rows per scatter tile (8 per thread)
constexpr int kGbRowsPerThread = kGbTile / kGbThreads;
constexpr int kGbMaxBins = 256;           // per level
constexpr int kGbMaxBits = 14;
constexpr int kGbMaxChunks = 4096;      // upper bound of level-1 chunks (sizes hist1)
constexpr int kGbSlots = 4096;            // LDS table slots per partition
// The wide one-level form (round 3): ONE flat scatter into up to 2048 bins + 8192-slot LDS tables.
constexpr int kGbWideThreads = 1024;
constexpr int kGbWideRpt = 24;            // rows a thread keeps in registers
constexpr int kGbWideTile = kGbWideThreads * kGbWideRpt;   // 24576 rows per workgroup
constexpr int kGbWideRounds = 3;          // the tile goes through the LDS reorder buffer in 3 rounds
constexpr int kGbWideChunk = kGbWideTile / kGbWideRounds;  // 8192 records = 96 KiB of LDS
constexpr int kGbWideMaxBits = 11;
constexpr int kGbWideSlots = 8192;        // LDS table slots per partition (128 KiB)
constexpr int kGbWideMaxGroups = 6144;    // groups a partition may hold (load factor 0.75)
constexpr int kGbMaxProbes = 256;         // LDS slots a row looks at before it gives up on the LDS table
constexpr uint32_t kGbHashMul = 0x9E3779B1u;     // odd => k -> k * M mod 2^32 is a bijection
constexpr uint32_t kGbHashInv = 0x0E8B2F51u;     // M * Minv == 1 mod 2^32
static_assert(static_cast<uint32_t>(kGbHashMul * kGbHashInv) == 1u, "hash inverse");

__device__ __forceinline__ uint32_t gbp_hash_keyed(int32_t key) {
  return static_cast<uint32_t>(key) * kGbHashMul;
}

struct GbpArgs {
  const int32_t* keys;     // element 0 of this slice
  const int64_t* values;
  Bits kvalid, vvalid;     // logical bitmaps of this slice (base == NULL: all valid)
  int64_t n;               // rows in this slice
  int bits, b1, b2;        // partition bits: total, level 1, level 2 (b2 == 0: one level)
  int64_t chunk_rows;      // level-1 chunk (a multiple of kGbTile)
  int64_t nchunks;
  uint32_t* part_count;    // [2^bits]
  uint32_t* part_start;    // [2^bits + 1]
  uint32_t* cursor2;       // [2^bits]
  uint32_t* cursor1;       // [2^b1] level-1 running cursors (global-cursor form)
  uint32_t* hist1;         // [2^b1 * nchunks], digit-major
  uint32_t* l1_start;      // [2^b1 + 1]
  uint32_t* l2_tile_start; // [2^b1 + 1]
  uint32_t* agg_unit_start; // [2^bits + 1]: aggregate work units (<= kGbAggChunk rows) before partition p
  int32_t* keys_a;
  int64_t* vals_a;
  int32_t* keys_b;
  int64_t* vals_b;
  // dense-id form (the HashAggregateKernel boundary: "keys" are the caller's Grouper ids 0..G-1, never null).  The
  // bijection is then id << dense_shl (the top bits of the id range pick the partition) and the partial aggregates
  // are flushed straight into the caller's three state arrays instead of the keyed HBM table.
  int dense;               // 0 = keyed table, 1 = dense ids
  int dense_shl;           // 32 - ceil(log2(num_groups))
  unsigned long long* dense_sums;
  unsigned long long* dense_counts;
  unsigned int* dense_null_seen;
  int wide;                // 1 = the wide one-level form: b1 == 0, b2 == bits <= 11, 12-byte records in `recs`
  uint32_t room;           // != 0: the wide form WITHOUT a histogram pass — partition p owns records [p * room, (p + 1) * room)
  uint32_t stripe;         // != 0 (rooms only): chunks of `stripe` records — record `off` of partition p lives at ((off / stripe) * 2^bits + p) * stripe + off % stripe, see gbp_rec_phys
  uint32_t* part_end;      // [2^bits] one past a partition's last record (part_start[p + 1] with exact counts)
  uint32_t* overflow;      // set by the wide scatter when a partition outgrows its room
  uint8_t* recs;           // [n] {key u32, value lo, value hi}
  int xcd_map;             // bit 0: level-2 scatter, bit 1: aggregate, bit 2: level-1 scatter — XCD-contiguous work numbering
  int agg_pipe;            // software-pipelined loads in the LDS aggregate kernel (A/B knob)
  uint32_t agg_chunk;      // rows per aggregate work unit (a power of two)
  // emit form (arx_groupby_sum_i64_consume_partials, the sharded group-by): a work unit's groups leave as ArxGroupPartial
  // records in the region of the rank that owns their key instead of going into the HBM table
  ArxGroupPartial* emit_records;        // NULL: the table
  unsigned long long* emit_cursor;      // [emit_parts] records written to every region so far (may pass emit_capacity: overflow)
  int emit_parts;                       // <= kGbEmitMaxParts
  int64_t emit_capacity;                // records per region
};
constexpr int kGbEmitMaxParts = 64;

__device__ __forceinline__ uint32_t gbp_hash(const GbpArgs& a, int32_t key) {
  return a.dense ? (static_cast<uint32_t>(key) << a.dense_shl) : gbp_hash_keyed(key);
}
__device__ __forceinline__ int32_t gbp_unhash(const GbpArgs& a, uint32_t h) {
  return a.dense ? static_cast<int32_t>(h >> a.dense_shl) : static_cast<int32_t>(h * kGbHashInv);
}

// STRIPED rooms.  The flat scatter appends to 2^bits rooms that lie room * 12 bytes (24 MB at 4e9 rows) apart: every
// (tile, bin) run is in another translation of the per-CU TLB — 31 % of the kernel's UTCL1 requests miss (8e8 misses
// for 4e9 rows; a streaming copy: 7e4), while its DRAM credit stalls per ms are a quarter of the copy's
// (profiles/r04_w_*): the pass waits for address translation, not for memory.  With stripes, chunk c (`stripe` records) of
// EVERY room lies in the same 2^bits * stripe records (6 MB at 256): distinct keys fill the rooms in step, so all frontiers of
// the moment share a few translations.  Logical record numbers (p * room + off) stay what cursors, counts and work
// units are kept in; only the two kernels that touch `recs` map them.
__device__ __forceinline__ int64_t gbp_rec_phys(const GbpArgs& a, uint32_t part, uint32_t logical) {
  if (a.stripe == 0) return logical;
  const uint32_t off = logical - part * a.room;
  const uint32_t chunk = off / a.stripe;
  return ((static_cast<int64_t>(chunk) << a.bits) + part) * a.stripe + (off - chunk * a.stripe);
}

template <bool HAS_NULLS>
__device__ __forceinline__ bool gbp_streamed(const GbpArgs& a, int64_t r) {
  if constexpr (!HAS_NULLS) {
    return true;
  } else {
    const uint64_t kv = load_word(a.kvalid, r >> 6);
    const uint64_t vv = load_word(a.vvalid, r >> 6);
    return ((kv & vv) >> (r & 63)) & 1ull;
  }
}

// ---- K0: rows that are not fully valid.  Null key => the null-key group (wave-reduced, one
// atomic per wave); valid key + null value => the group must exist and gets its null flag.
__global__ __launch_bounds__(kBlock) void gbp_null_rows_kernel(GroupbyView v, GbpArgs a) {
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwords = (a.n + 63) >> 6;
  unsigned long long nsum = 0, ncnt = 0;
  bool nflag = false, nseen = false;
  uint32_t fresh = 0;
  for (int64_t w = wave_g; w < nwords; w += nwaves) {
    const uint64_t kv = load_word(a.kvalid, w);
    const uint64_t vv = load_word(a.vvalid, w);
    const uint64_t in_range = load_word(Bits{nullptr, 0, a.n, 0}, w);
    const uint64_t bad = ~(kv & vv) & in_range;
    if (bad == 0) continue;  // wave-uniform
    if ((bad >> lane) & 1ull) {
      const int64_t r = (w << 6) + lane;
      const bool kok = (kv >> lane) & 1ull;
      const bool vok = (vv >> lane) & 1ull;
      if (kok) {  // => value is null
        if (a.dense) {
          atomicOr(&a.dense_null_seen[static_cast<uint32_t>(a.keys[r])], 1u);
        } else {
          const int64_t slot = gb_find_or_insert(v, a.keys[r], &fresh);
          if (slot < 0) atomicExch(&v.hdr->overflow, 1u);
          else atomicOr(&v.flags[slot], 1u);
        }
      } else {
        nseen = true;
        if (vok) {
          nsum += static_cast<unsigned long long>(a.values[r]);
          ncnt += 1;
        } else {
          nflag = true;
        }
      }
    }
  }
  if (a.dense) return;   // group ids are never null
  gb_publish_new_groups(v, fresh);
  const bool any_seen = __any(nseen);
  if (!any_seen) return;
  nsum = wave_reduce_sum_u64(nsum);
  ncnt = wave_reduce_sum_u64(ncnt);
  const bool any_flag = __any(nflag);
  if (lane == 0) {
    const int64_t slot = gb_null_slot(v);
    if (ncnt != 0) {
      atomicAdd(&v.sums[slot], nsum);
      atomicAdd(&v.counts[slot], ncnt);
    }
    if (any_flag) atomicOr(&v.flags[slot], 1u);
  }
}

// ---- K1: histogram.  One workgroup per level-1 chunk.
template <bool HAS_NULLS>
__global__ __launch_bounds__(kGbThreads) void gbp_hist_kernel(GbpArgs a) {
  __shared__ uint32_t h[1 << kGbMaxBits];
  const int tid = threadIdx.x;
  const int nparts = 1 << a.bits;
  for (int i = tid; i < nparts; i += kGbThreads) h[i] = 0;
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * a.chunk_rows;
  const int64_t end = begin + a.chunk_rows < a.n ? begin + a.chunk_rows : a.n;
  const int shift = 32 - a.bits;
  constexpr int U = 8;  // independent key loads in flight per thread
  int64_t r = begin + tid;
  for (; r + (U - 1) * kGbThreads < end; r += U * kGbThreads) {
    int32_t kk[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      kk[u] = a.keys[r + u * kGbThreads];
      ok[u] = gbp_streamed<HAS_NULLS>(a, r + u * kGbThreads);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ok[u]) atomicAdd(&h[gbp_hash(a, kk[u]) >> shift], 1u);
    }
  }
  for (; r < end; r += kGbThreads) {
    if (gbp_streamed<HAS_NULLS>(a, r)) atomicAdd(&h[gbp_hash(a, a.keys[r]) >> shift], 1u);
  }
  __syncthreads();
  for (int i = tid; i < nparts; i += kGbThreads) {
    const uint32_t c = h[i];
    if (c != 0) atomicAdd(&a.part_count[i], c);
  }
  if (a.wide) return;   // (the flat level takes its offsets from the partition cursors)
  const int nb1 = 1 << a.b1;
  const int per = 1 << a.b2;
  for (int d = tid; d < nb1; d += kGbThreads) {
    uint32_t sum = 0;
    for (int j = 0; j < per; ++j) sum += h[(d << a.b2) + j];
    a.hist1[static_cast<int64_t>(d) * a.nchunks + blockIdx.x] = sum;
  }
}

// ---- K2a: one workgroup.  part_start = exclusive scan of part_count; level-1 starts; level-2
// cursors; the level-2 tile map (how many tiles each level-1 partition needs).
__global__ __launch_bounds__(1024) void gbp_scan_a_kernel(GbpArgs a) {
  __shared__ uint32_t ps[(1 << kGbMaxBits) + 1];
  __shared__ uint32_t wave_tot[16];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nparts = 1 << a.bits;
  const int per = (nparts + 1023) / 1024;
  const int b = tid * per;
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
    if (i < nparts) {
      a.cursor2[i] = s0;
      a.part_end[i] = ps[i + 1];
    }
  }
  const int nb1 = 1 << a.b1;
  for (int d = tid; d <= nb1; d += 1024) {
    const uint32_t s1 = ps[d == nb1 ? nparts : (d << a.b2)];
    a.l1_start[d] = s1;
    if (d < nb1) a.cursor1[d] = s1;
  }
  // tile map of level 2 (nb1 <= 256 entries): exclusive scan of ceil(size / tile)
  __syncthreads();
  uint32_t tiles = 0;
  if (tid < nb1) {
    const uint32_t lo = ps[tid << a.b2];
    const uint32_t hi = ps[(tid + 1) == nb1 ? nparts : ((tid + 1) << a.b2)];
    tiles = (hi - lo + kGbTile - 1) / kGbTile;
  }
  const uint32_t tincl = wave_inclusive_scan_u32(tiles);
  __syncthreads();
  if (lane == 63) wave_tot[wave] = tincl;
  __syncthreads();
  uint32_t tprefix = tincl - tiles;
  for (int k = 0; k < wave; ++k) tprefix += wave_tot[k];
  if (tid < nb1) a.l2_tile_start[tid] = tprefix;
  if (tid == nb1 - 1) a.l2_tile_start[nb1] = tprefix + tiles;
  // aggregate work units: every partition is cut into pieces of <= kGbAggChunk rows
  __syncthreads();
  uint32_t usum = 0;
  for (int i = b; i < e; ++i) usum += (ps[i + 1] - ps[i] + a.agg_chunk - 1) / a.agg_chunk;
  const uint32_t uincl = wave_inclusive_scan_u32(usum);
  if (lane == 63) wave_tot[wave] = uincl;
  __syncthreads();
  uint32_t uprefix = uincl - usum;
  for (int k = 0; k < wave; ++k) uprefix += wave_tot[k];
  for (int i = b; i < e; ++i) {
    a.agg_unit_start[i] = uprefix;
    uprefix += (ps[i + 1] - ps[i] + a.agg_chunk - 1) / a.agg_chunk;
  }
  if (tid == 1023) a.agg_unit_start[nparts] = uprefix;
}

// ---- K2b: one workgroup per level-1 digit: hist1[d][*] -> exclusive offsets (+ l1_start[d]).
__global__ __launch_bounds__(1024) void gbp_scan_b_kernel(GbpArgs a) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  uint32_t* row = a.hist1 + static_cast<int64_t>(blockIdx.x) * a.nchunks;
  if (tid == 0) carry_s = a.l1_start[blockIdx.x];
  __syncthreads();
  for (int64_t base = 0; base < a.nchunks; base += 1024) {
    const int64_t i = base + tid;
    const uint32_t x = i < a.nchunks ? row[i] : 0u;
    const uint32_t incl = wave_inclusive_scan_u32(x);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t prefix = incl - x;
    for (int k = 0; k < wave; ++k) prefix += wave_tot[k];
    const uint32_t carry = carry_s;
    if (i < a.nchunks) row[i] = carry + prefix;
    __syncthreads();
    if (tid == 1023) carry_s = carry + prefix + x;
    __syncthreads();
  }
}

// ---- K3/K4: scatter one tile through LDS.
struct __attribute__((aligned(16))) GbpScatterLds {
  uint64_t vals[kGbTile];
  uint32_t keys[kGbTile];
  uint32_t cnt[kGbMaxBins];
  uint32_t start[kGbMaxBins];
  uint32_t gbase[kGbMaxBins];
  uint32_t cursor[kGbMaxBins];
  uint32_t wave_tot[kGbThreads / 64];
};

// LEVEL 1: input = the caller's arrays (validity applies), digit = top b1 bits, offsets from the
//          chunk cursors.   LEVEL 2: input = level-1 output, digit = next b2 bits, offsets from
//          cursor2 atomics.
template <int LEVEL, bool HAS_NULLS, bool GCUR = (LEVEL == 2)>
__device__ __forceinline__ void gbp_scatter_tile(const GbpArgs& a, GbpScatterLds& lds,
                                                 const int32_t* __restrict__ kin,
                                                 const int64_t* __restrict__ vin, int64_t row0,
                                                 int nrows, uint32_t part_hi,
                                                 int32_t* __restrict__ kout,
                                                 int64_t* __restrict__ vout) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nb = LEVEL == 1 ? (1 << a.b1) : (1 << a.b2);
  const int dshift = LEVEL == 1 ? (32 - a.b1) : (32 - a.bits);
  const uint32_t dmask = static_cast<uint32_t>(nb - 1);
  if (tid < nb) lds.cnt[tid] = 0;
  __syncthreads();

  int32_t key[kGbRowsPerThread];
  int64_t val[kGbRowsPerThread];
  int dig[kGbRowsPerThread];
  uint32_t rank[kGbRowsPerThread];
  // Loads are unconditional (rows past the tile's end re-read its last row, null rows are read and
  // dropped): a guarded load makes the compiler wait for each one before issuing the next.
#pragma unroll
  for (int i = 0; i < kGbRowsPerThread; ++i) {
    const int p = i * kGbThreads + tid;
    const int64_t r = row0 + (p < nrows ? p : nrows - 1);
    key[i] = kin[r];
    val[i] = vin[r];
  }
#pragma unroll
  for (int i = 0; i < kGbRowsPerThread; ++i) {
    const int p = i * kGbThreads + tid;
    bool ok = p < nrows;
    if constexpr (LEVEL == 1 && HAS_NULLS) ok = ok && gbp_streamed<true>(a, row0 + (p < nrows ? p : nrows - 1));
    dig[i] = ok ? static_cast<int>((gbp_hash(a, key[i]) >> dshift) & dmask) : -1;
  }
#pragma unroll
  for (int i = 0; i < kGbRowsPerThread; ++i) {
    rank[i] = 0;
    if (dig[i] >= 0) rank[i] = atomicAdd(&lds.cnt[dig[i]], 1u);
  }
  __syncthreads();

  // exclusive scan of the bin counts (nb <= 256 = 4 waves) + global run bases
  uint32_t c = 0;
  if (tid < nb) c = lds.cnt[tid];
  const uint32_t incl = wave_inclusive_scan_u32(c);
  if (lane == 63) lds.wave_tot[wave] = incl;
  __syncthreads();
  if (tid < nb) {
    uint32_t pre = incl - c;
    for (int k = 0; k < wave; ++k) pre += lds.wave_tot[k];
    lds.start[tid] = pre;
    if constexpr (!GCUR) {
      const uint32_t g = lds.cursor[tid];
      lds.gbase[tid] = g;
      lds.cursor[tid] = g + c;
    } else if constexpr (LEVEL == 1) {
      lds.gbase[tid] = c != 0 ? atomicAdd(&a.cursor1[tid], c) : 0u;
    } else {
      lds.gbase[tid] = c != 0 ? atomicAdd(&a.cursor2[(part_hi << a.b2) + tid], c) : 0u;
    }
  }
  __syncthreads();

#pragma unroll
  for (int i = 0; i < kGbRowsPerThread; ++i) {
    if (dig[i] >= 0) {
      const uint32_t pos = lds.start[dig[i]] + rank[i];
      lds.keys[pos] = static_cast<uint32_t>(key[i]);
      lds.vals[pos] = static_cast<uint64_t>(val[i]);
    }
  }
  __syncthreads();

  const int total = static_cast<int>(lds.start[nb - 1] + lds.cnt[nb - 1]);
  for (int p = tid; p < total; p += kGbThreads) {
    const uint32_t k = lds.keys[p];
    const uint32_t d = (gbp_hash(a, static_cast<int32_t>(k)) >> dshift) & dmask;
    const uint32_t dst = lds.gbase[d] + (static_cast<uint32_t>(p) - lds.start[d]);
    kout[dst] = static_cast<int32_t>(k);
    vout[dst] = static_cast<int64_t>(lds.vals[p]);
  }
  __syncthreads();
}

template <bool HAS_NULLS>
__global__ __launch_bounds__(kGbThreads, 6) void gbp_scatter1_kernel(GbpArgs a) {
  __shared__ GbpScatterLds lds;
  const int tid = threadIdx.x;
  const int nb = 1 << a.b1;
  if (tid < nb) lds.cursor[tid] = a.hist1[static_cast<int64_t>(tid) * a.nchunks + blockIdx.x];
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * a.chunk_rows;
  const int64_t end = begin + a.chunk_rows < a.n ? begin + a.chunk_rows : a.n;
  for (int64_t row0 = begin; row0 < end; row0 += kGbTile) {
    const int nrows = static_cast<int>(end - row0 < kGbTile ? end - row0 : kGbTile);
    gbp_scatter_tile<1, HAS_NULLS>(a, lds, a.keys, a.values, row0, nrows, 0, a.keys_a, a.vals_a);
  }
}

// Level 1 with global cursors (A/B knob groupby_l1_global): one tile per workgroup, one returning atomic per
// (tile, digit) — consecutive tiles extend the same few output runs instead of every chunk owning its own.
template <bool HAS_NULLS>
__global__ __launch_bounds__(kGbThreads, 6) void gbp_scatter1g_kernel(GbpArgs a) {
  __shared__ GbpScatterLds lds;
  const uint32_t tile = (a.xcd_map & 4) ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int64_t row0 = static_cast<int64_t>(tile) * kGbTile;
  const int nrows = static_cast<int>(a.n - row0 < kGbTile ? a.n - row0 : kGbTile);
  gbp_scatter_tile<1, HAS_NULLS, true>(a, lds, a.keys, a.values, row0, nrows, 0, a.keys_a, a.vals_a);
}

__global__ __launch_bounds__(kGbThreads, 6) void gbp_scatter2_kernel(GbpArgs a) {
  __shared__ GbpScatterLds lds;
  __shared__ uint32_t part_s;
  const int tid = threadIdx.x;
  const int nb1 = 1 << a.b1;
  // XCD x takes a contiguous eighth of the tiles (whole level-1 partitions): a partition's (tile, digit) runs meet in
  // one L2 instead of eight (sort: -18 % on the same kernel shape)
  const uint32_t g = (a.xcd_map & 1) ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  if (g >= a.l2_tile_start[nb1]) return;  // over-provisioned grid
  // which level-1 partition owns tile g: the last p with l2_tile_start[p] <= g
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
  const int64_t row0 = lo + static_cast<int64_t>(g - a.l2_tile_start[p]) * kGbTile;
  const int nrows = static_cast<int>(hi - row0 < kGbTile ? hi - row0 : kGbTile);
  gbp_scatter_tile<2, false>(a, lds, a.keys_a, a.vals_a, row0, nrows, p, a.keys_b, a.vals_b);
}

// ---- K1r / K2r: the wide form without a histogram pass.  A bijective multiplicative hash spreads distinct keys evenly,
// so partition sizes are binomial around n / 2^bits: every partition gets a ROOM of mean + 6 sigma + 64 records
// (gbp_room_for) instead of its counted size, the scatter appends at the room's cursor and flags a partition that
// outgrows its room (few hot keys: the slice is then redone with the counted plan, and the rest of the call stays on
// it).  Saves the 4 B/row histogram pass: 2.6 of 43 ms at 4e9 rows.
__global__ __launch_bounds__(1024) void gbp_rooms_init_kernel(GbpArgs a) {
  const int nparts = 1 << a.bits;
  for (int i = threadIdx.x; i <= nparts; i += 1024) {
    a.part_start[i] = static_cast<uint32_t>(i) * a.room;
    if (i < nparts) a.cursor2[i] = static_cast<uint32_t>(i) * a.room;
  }
  if (threadIdx.x == 0) *a.overflow = 0;
}

// after the scatter: records that arrived per partition, the ends of the partitions, the aggregate's work units
__global__ __launch_bounds__(1024) void gbp_rooms_scan_kernel(GbpArgs a) {
  __shared__ uint32_t wave_tot[16];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nparts = 1 << a.bits;
  const int per = (nparts + 1023) / 1024;
  const int b = tid * per;
  const int e = b + per < nparts ? b + per : nparts;
  uint32_t usum = 0;
  for (int i = b; i < e; ++i) {
    const uint32_t lo = static_cast<uint32_t>(i) * a.room;
    const uint32_t cur = a.cursor2[i];
    const uint32_t cnt = (cur - lo) > a.room ? a.room : cur - lo;   // (an overflowed partition: the call is redone)
    a.part_count[i] = cnt;
    a.part_end[i] = lo + cnt;
    usum += (cnt + a.agg_chunk - 1) / a.agg_chunk;
  }
  const uint32_t uincl = wave_inclusive_scan_u32(usum);
  if (lane == 63) wave_tot[wave] = uincl;
  __syncthreads();
  uint32_t uprefix = uincl - usum;
  for (int k = 0; k < wave; ++k) uprefix += wave_tot[k];
  for (int i = b; i < e; ++i) {
    a.agg_unit_start[i] = uprefix;
    uprefix += (a.part_count[i] + a.agg_chunk - 1) / a.agg_chunk;
  }
  if (tid == 1023) a.agg_unit_start[nparts] = uprefix;
}

// ---- K3w: the flat level of the wide form.  One workgroup = one tile of 24576 rows held in REGISTERS (24 per
// thread): ranked with LDS atomics over up to 2048 bins, then moved through a 8192-record LDS buffer in three rounds,
// so a (tile, bin) run is 12 records at 2048 bins — three times what an LDS-resident tile gives — and written as
// 12-byte {key, value} records (one output stream per bin instead of two).  2^30 rows into 2048 bins: 7.9 ms
// (3.3 TB/s of 24 B/row) against 5.2 + 4.4 ms for the two levels it replaces
// (scripts/micro/wide_scatter_bench.hip, profiles/r03_a_wide_scatter_register_staged_tiles.txt).
// Cache policy of the wide form's one-pass streams (A/B by rebuilding: scripts/build_gb_variants.sh): bit 0 the scatter's
// key / value loads, bit 1 the aggregate's record loads, bit 2 the scatter's record stores are non-temporal.
constexpr int kGbNt = 0;
constexpr int kGbAosWide = 1;       // the wide form's aggregate reads four records per thread with three 16-byte loads (0: one 12-byte load per record).  4e9 rows: 40.9 / 38.9 ms against 41.8 / 40.3 (two boxes' worth of interleaved runs, profiles/r03_x_*)
constexpr int kGbWideAggU = 12;     // records a thread of the wide form's aggregate keeps in flight (A/B by rebuilding: scripts/build_gb_variants.sh)
constexpr int kGbAosRaw = 1;        // the wide form's aggregate keeps raw quads in its pipeline registers, loads unconditional (0: the round-3 form, whose loads were waited for inside their branch)
constexpr int kGbAggQuadProbe = 0;  // the raw form reads a quad's four home slots together before it branches per row
constexpr int kGbAggGroupProbe = 1;  // the LDS tables are probed by aligned groups of four slots (one ds_read_b128 per step)
constexpr int kGbKeysFirst = 0;     // the flat level issues all key loads before the value loads: ranking starts while the values are in flight
constexpr int kGbPairAtomics = 0;   // the flat level's cursors: 1 = one 64-bit atomic per pair of bins, 0 = one per bin.  A/B at 4e9 rows: 41.2 / 45.2 ms paired, 40.4 / 40.4 single (profiles/r03_t_groupby_paired_cursor_atomics_ab.txt): the level is not bound by its atomics
template <typename T>
__device__ __forceinline__ T gb_load(const T* p, bool nt) {
  return nt ? __builtin_nontemporal_load(p) : *p;
}
struct __attribute__((packed, aligned(4))) GbpRec {
  uint32_t key, vlo, vhi;
};
static_assert(sizeof(GbpRec) == 12, "12-byte records");

struct __attribute__((aligned(16))) GbpWideScatterLds {
  uint64_t vals[kGbWideChunk];
  uint32_t keys[kGbWideChunk];
  uint32_t start[1 << kGbWideMaxBits];   // bin counts, then the bins' first positions inside the tile
  uint32_t gbase[1 << kGbWideMaxBits];   // where the tile's run of a bin starts in the output
  uint32_t wave_tot[kGbWideThreads / 64];
  uint32_t total;
};

// -DARX_GBP_PROFILE (scripts/build_gb_variants2.sh, never the product build): phase timestamps of a few workgroups of the
// flat level, read back through arx_debug_gbp_profile
#ifdef ARX_GBP_PROFILE
__device__ unsigned long long g_gbp_prof[64 * 8];
#define GBP_STAMP(k)                                                                                     \
  do {                                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x % 1009 == 7 && blockIdx.x / 1009 < 64) {                          \
      g_gbp_prof[(blockIdx.x / 1009) * 8 + (k)] = __builtin_amdgcn_s_memrealtime();                      \
    }                                                                                                    \
  } while (0)
#else
#define GBP_STAMP(k) do { } while (0)
#endif

template <bool HAS_NULLS>
__global__ __launch_bounds__(kGbWideThreads) void gbp_scatter_wide_kernel(GbpArgs a) {
  __shared__ GbpWideScatterLds lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nb = 1 << a.bits;
  const int dshift = 32 - a.bits;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kGbWideTile;
  if (row0 >= a.n) return;   // workgroup-uniform
  const int nrows = static_cast<int>(a.n - row0 < kGbWideTile ? a.n - row0 : kGbWideTile);
  for (int b = tid; b < nb; b += kGbWideThreads) lds.start[b] = 0;
  GBP_STAMP(0);
  uint32_t key[kGbWideRpt];
  int64_t val[kGbWideRpt];
  // unconditional loads (rows past the tile's end re-read its last row): all in flight together
  if constexpr (kGbKeysFirst != 0) {
#pragma unroll
    for (int i = 0; i < kGbWideRpt; ++i) {
      const int p = i * kGbWideThreads + tid;
      const int64_t r = row0 + (p < nrows ? p : nrows - 1);
      key[i] = static_cast<uint32_t>(gb_load(a.keys + r, (kGbNt & 1) != 0));
    }
#pragma unroll
    for (int i = 0; i < kGbWideRpt; ++i) {
      const int p = i * kGbWideThreads + tid;
      const int64_t r = row0 + (p < nrows ? p : nrows - 1);
      val[i] = gb_load(a.values + r, (kGbNt & 1) != 0);
    }
  } else {
#pragma unroll
    for (int i = 0; i < kGbWideRpt; ++i) {
      const int p = i * kGbWideThreads + tid;
      const int64_t r = row0 + (p < nrows ? p : nrows - 1);
      key[i] = static_cast<uint32_t>(gb_load(a.keys + r, (kGbNt & 1) != 0));
      val[i] = gb_load(a.values + r, (kGbNt & 1) != 0);
    }
  }
  __syncthreads();
  uint32_t pos[kGbWideRpt];
#pragma unroll
  for (int i = 0; i < kGbWideRpt; ++i) {
    const int p = i * kGbWideThreads + tid;
    bool ok = p < nrows;
    if constexpr (HAS_NULLS) ok = ok && gbp_streamed<true>(a, row0 + (p < nrows ? p : nrows - 1));   // null rows: K0
    pos[i] = ok ? atomicAdd(&lds.start[gbp_hash(a, static_cast<int32_t>(key[i])) >> dshift], 1u) : 0xFFFFFFFFu;
  }
  __syncthreads();
  GBP_STAMP(1);
  // exclusive scan of the bin counts: `per` consecutive bins per thread (1 or 2)
  const int per = (nb + kGbWideThreads - 1) / kGbWideThreads;
  uint32_t c[2] = {0, 0};
  uint32_t mine = 0;
  for (int k = 0; k < per; ++k) {
    const int b = tid * per + k;
    c[k] = b < nb ? lds.start[b] : 0u;
    mine += c[k];
  }
  const uint32_t incl = wave_inclusive_scan_u32(mine);
  if (lane == 63) lds.wave_tot[wave] = incl;
  __syncthreads();
  uint32_t pre = incl - mine;
  for (int k = 0; k < wave; ++k) pre += lds.wave_tot[k];
  // kGbPairAtomics: ONE returning 64-bit atomic per PAIR of adjacent bins (low half = the even bin's cursor; positions
  // stay below 2^32, so the low half never carries).  Global atomics run at 24 G/s chip-wide and one per (tile, bin) is
  // 3.3e8 of them for 4e9 rows — but halving them did not move the kernel: they overlap the tile's loads.
  uint32_t gb[2] = {0, 0};
  if constexpr (kGbPairAtomics == 0) {
    for (int k = 0; k < per; ++k) {
      const int b = tid * per + k;
      if (b < nb && c[k] != 0) gb[k] = atomicAdd(&a.cursor2[b], c[k]);
    }
  } else if (per == 1) {   // (workgroup-uniform) the pair's bins sit in neighbouring lanes
    const uint32_t cn = __shfl_xor(c[0], 1, 64);
    unsigned long long old = 0;
    if ((tid & 1) == 0 && tid < nb && (c[0] | cn) != 0) {
      old = atomicAdd(reinterpret_cast<unsigned long long*>(a.cursor2 + tid),
                      static_cast<unsigned long long>(c[0]) | (static_cast<unsigned long long>(cn) << 32));
    }
    const uint32_t from_even = __shfl_xor(static_cast<uint32_t>(old >> 32), 1, 64);
    gb[0] = (tid & 1) == 0 ? static_cast<uint32_t>(old) : from_even;
  } else if (tid * 2 < nb && (c[0] | c[1]) != 0) {
    const unsigned long long old =
        atomicAdd(reinterpret_cast<unsigned long long*>(a.cursor2 + tid * 2),
                  static_cast<unsigned long long>(c[0]) | (static_cast<unsigned long long>(c[1]) << 32));
    gb[0] = static_cast<uint32_t>(old);
    gb[1] = static_cast<uint32_t>(old >> 32);
  }
  for (int k = 0; k < per; ++k) {
    const int b = tid * per + k;
    if (b < nb) {
      lds.start[b] = pre;
      const uint32_t g = gb[k];
      lds.gbase[b] = g;
      if (a.room != 0 && c[k] != 0 && (g - static_cast<uint32_t>(b) * a.room) + c[k] > a.room) atomicOr(a.overflow, 1u);
    }
    pre += c[k];
  }
  if (tid == kGbWideThreads - 1) lds.total = pre;
  __syncthreads();
  GBP_STAMP(2);
#pragma unroll
  for (int i = 0; i < kGbWideRpt; ++i) {
    if (pos[i] != 0xFFFFFFFFu) pos[i] += lds.start[gbp_hash(a, static_cast<int32_t>(key[i])) >> dshift];
  }
  GBP_STAMP(3);
  const int total = static_cast<int>(lds.total);
  GbpRec* __restrict__ out = reinterpret_cast<GbpRec*>(a.recs);
  for (int r = 0; r < kGbWideRounds; ++r) {
    const uint32_t lo = static_cast<uint32_t>(r) * kGbWideChunk;
    if (static_cast<int>(lo) >= total) break;   // workgroup-uniform
#pragma unroll
    for (int i = 0; i < kGbWideRpt; ++i) {
      const uint32_t q = pos[i] - lo;   // (skipped rows: 0xFFFFFFFF - lo is never < the chunk)
      if (q < static_cast<uint32_t>(kGbWideChunk)) {
        lds.keys[q] = key[i];
        lds.vals[q] = static_cast<uint64_t>(val[i]);
      }
    }
    __syncthreads();
    if (r == 0) GBP_STAMP(4);
    const int cnt = total - static_cast<int>(lo) < kGbWideChunk ? total - static_cast<int>(lo) : kGbWideChunk;
    for (int p = tid; p < cnt; p += kGbWideThreads) {
      const uint32_t k = lds.keys[p];
      const uint32_t d = gbp_hash(a, static_cast<int32_t>(k)) >> dshift;
      const uint64_t v = lds.vals[p];
      GbpRec rec;
      rec.key = k;
      rec.vlo = static_cast<uint32_t>(v);
      rec.vhi = static_cast<uint32_t>(v >> 32);
      const uint32_t dst = lds.gbase[d] + (lo + static_cast<uint32_t>(p) - lds.start[d]);
      if (a.room == 0 || dst - d * a.room < a.room) {   // (never past a room: the slice is redone then)
        const int64_t at = gbp_rec_phys(a, d, dst);
        if constexpr ((kGbNt & 4) != 0) {
          uint32_t* o = reinterpret_cast<uint32_t*>(out + at);
          __builtin_nontemporal_store(rec.key, o);
          __builtin_nontemporal_store(rec.vlo, o + 1);
          __builtin_nontemporal_store(rec.vhi, o + 2);
        } else {
          out[at] = rec;
        }
      }
    }
    __syncthreads();
    if (r == 0) GBP_STAMP(5);
    if (r == 1) GBP_STAMP(6);
  }
  GBP_STAMP(7);
}

// ---- K5: LDS aggregation.  One workgroup per work unit = <= kGbAggChunk rows of ONE partition
// (DIRECT: of the caller's rows — the plan with bits == 0 used when all groups fit one table).
template <int SLOTS, int THREADS>
struct __attribute__((aligned(16))) GbpAggLds {
  unsigned long long sums[SLOTS];
  uint32_t tags[SLOTS];   // 0 = empty, else the low (32 - bits) bits of the key hash
  uint32_t cnts[SLOTS];
  unsigned long long zsum;   // the one key whose tag is 0 has its own accumulator
  uint32_t zcnt;
  uint32_t part, row_lo, row_hi;
  uint32_t emit_wave[THREADS / 64][kGbEmitMaxParts];   // emit form: the groups of every owner rank among a wave's slots, then where they start in the unit's run ...
  unsigned long long emit_base[kGbEmitMaxParts];       // ... and where the unit's run starts in the owner's region
};

// SLOTS / THREADS: 4096 / 512 (two workgroups per CU), or the wide form's 8192 / 1024 (one per CU);
// AOS: the rows are the wide scatter's 12-byte records (a.recs) instead of the keys / vals arrays.
template <bool DIRECT, bool HAS_NULLS, int SLOTS = kGbSlots, int THREADS = kGbThreads, bool AOS = false, int U = 4>
__global__ __launch_bounds__(THREADS) void gbp_aggregate_kernel(GroupbyView v, GbpArgs a,
                                                                const int32_t* __restrict__ keys,
                                                                const int64_t* __restrict__ vals) {
  constexpr int kLgSlots = SLOTS == 8192 ? 13 : 12;
  static_assert((1 << kLgSlots) == SLOTS, "table sizes: 4096 or 8192 slots");
  __shared__ GbpAggLds<SLOTS, THREADS> t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  int64_t lo, hi;
  uint32_t q = 0;
  if constexpr (DIRECT) {
    lo = static_cast<int64_t>(blockIdx.x) * a.agg_chunk;
    hi = lo + a.agg_chunk < a.n ? lo + a.agg_chunk : a.n;
  } else {
    // which partition owns work unit u: the last q with agg_unit_start[q] <= u.  Two rounds of a
    // 64-lane search (coarse stride, then inside the stride) by the first wave.
    const uint32_t u = (a.xcd_map & 2) ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
    const int nparts = 1 << a.bits;
    if (u >= a.agg_unit_start[nparts]) return;  // over-provisioned grid (workgroup-uniform)
    if (tid < 64) {
      const int stride = (nparts + 63) / 64;
      const int c = lane * stride < nparts ? lane * stride : nparts;
      const uint64_t le = __ballot(lane * stride < nparts && a.agg_unit_start[c] <= u);
      const int coarse = (63 - __builtin_clzll(le)) * stride;  // le != 0: agg_unit_start[0] == 0
      uint32_t below = 0;
      for (int j = lane; j < stride; j += 64) {
        const int idx = coarse + j;
        below += (idx < nparts && a.agg_unit_start[idx] <= u) ? 1u : 0u;
      }
      below = wave_reduce_sum_u32(below);
      if (lane == 0) {
        const uint32_t part = static_cast<uint32_t>(coarse) + below - 1;
        const uint32_t first = a.part_start[part] + (u - a.agg_unit_start[part]) * a.agg_chunk;
        const uint32_t end = a.part_end[part];
        t.part = part;
        t.row_lo = first;
        t.row_hi = (end - first) > a.agg_chunk ? first + a.agg_chunk : end;
      }
    }
  }
  for (int i = tid; i < SLOTS; i += THREADS) {
    t.sums[i] = 0;
    t.tags[i] = 0;
    t.cnts[i] = 0;
  }
  if (tid == 0) {
    t.zsum = 0;
    t.zcnt = 0;
  }
  for (int i = tid; i < (THREADS / 64) * kGbEmitMaxParts; i += THREADS) (&t.emit_wave[0][0])[i] = 0;
  __syncthreads();
  if constexpr (!DIRECT) {
    q = t.part;
    lo = t.row_lo;
    hi = t.row_hi;
  }
  const int low_bits = 32 - a.bits;
  const uint32_t low_mask = a.bits == 0 ? 0xFFFFFFFFu : ((1u << low_bits) - 1u);
  const int hshift = low_bits > kLgSlots ? low_bits - kLgSlots : 0;
  uint32_t fresh = 0;   // keys this thread inserted into the HBM table
  // U rows in flight per thread: 4 x 12 B x 1024 threads = 48 KB per CU is about what 6 TB/s x 2 us of latency asks of
  // 256 CUs — the wide form keeps 8
  const int64_t span = hi - lo;
  // AOS && kGbAosWide: a thread takes FOUR consecutive 12-byte records with three 16-byte loads (48 bytes; dword-aligned
  // is all a global load needs) instead of twelve dword loads spread over four strided rows
  constexpr bool WIDE = AOS && kGbAosWide != 0 && (U % 4 == 0);
  const int64_t nit = WIDE ? ((span + 4 * THREADS - 1) / (4 * THREADS)) * 4 : (span + THREADS - 1) / THREADS;
  // software pipeline: the loads of batch i+1 are issued before batch i goes through the LDS
  // table, so the HBM latency overlaps the (serial, atomic) LDS work of the same wave
  int32_t kbuf[U], knext[U];
  unsigned long long vbuf[U], vnext[U];
  bool okbuf[U], oknext[U];
  auto load_batch = [&](int64_t it0, int32_t* kb, unsigned long long* vb, bool* ob) {
    if constexpr (WIDE) {
#pragma unroll
      for (int g = 0; g < U / 4; ++g) {
        const int64_t r0 = lo + ((it0 / 4 + g) * THREADS + tid) * 4;
        const uint32_t* rp = reinterpret_cast<const uint32_t*>(reinterpret_cast<const GbpRec*>(a.recs) +
                                                               gbp_rec_phys(a, q, static_cast<uint32_t>(r0 < hi ? r0 : hi - 1)));
        uint32_t d[12];
        if (r0 + 4 <= hi) {   // (striped rooms: a quad never crosses a chunk, r0 - room start and the chunk are multiples of 4)
          typedef arx_u32x4 __attribute__((aligned(4))) RecQuad;   // (a record starts at any multiple of 12 bytes)
          const RecQuad* q = reinterpret_cast<const RecQuad*>(rp);
          const arx_u32x4 q0 = q[0], q1 = q[1], q2 = q[2];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            d[e] = q0[e];
            d[4 + e] = q1[e];
            d[8 + e] = q2[e];
          }
        } else {   // the partition's last few records: one by one, clamped
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t rc = r0 + j < hi ? r0 + j : hi - 1;
            const uint32_t* p1 = reinterpret_cast<const uint32_t*>(reinterpret_cast<const GbpRec*>(a.recs) + gbp_rec_phys(a, q, static_cast<uint32_t>(rc)));
            d[3 * j] = p1[0]; d[3 * j + 1] = p1[1]; d[3 * j + 2] = p1[2];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ob[g * 4 + j] = r0 + j < hi;
          kb[g * 4 + j] = static_cast<int32_t>(d[3 * j]);
          vb[g * 4 + j] = (static_cast<unsigned long long>(d[3 * j + 2]) << 32) | d[3 * j + 1];
        }
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = lo + (it0 + u) * THREADS + tid;
      ob[u] = rr < hi;
      if constexpr (DIRECT && HAS_NULLS) {
        if (ob[u]) ob[u] = gbp_streamed<true>(a, rr);  // rows with a null were handled by K0
      }
      const int64_t rc = rr < hi ? rr : hi - 1;  // clamped: always readable
      if constexpr (AOS) {
        const uint32_t* rp = reinterpret_cast<const uint32_t*>(reinterpret_cast<const GbpRec*>(a.recs) + gbp_rec_phys(a, q, static_cast<uint32_t>(rc)));
        const uint32_t rk = gb_load(rp, (kGbNt & 2) != 0), rlo = gb_load(rp + 1, (kGbNt & 2) != 0), rhi = gb_load(rp + 2, (kGbNt & 2) != 0);
        kb[u] = static_cast<int32_t>(rk);
        vb[u] = (static_cast<unsigned long long>(rhi) << 32) | rlo;
      } else {
        kb[u] = keys[rc];
        vb[u] = static_cast<unsigned long long>(vals[rc]);
      }
    }
  };
  const bool pipe = a.agg_pipe != 0;  // A/B knob groupby_agg_pipe (uniform)
  auto process_row = [&](const int32_t key, const unsigned long long val) {
    const uint32_t kp = gbp_hash(a, key);
    const uint32_t tag = kp & low_mask;
    if (tag == 0) {
      atomicAdd(&t.zsum, val);
      atomicAdd(&t.zcnt, 1u);
      return;
    }
    // (a key that is not settled within kGbMaxProbes slots of its home goes to the HBM table: a full LDS table then
    //  costs a bounded number of probes per row, and both paths add into the same group in the end)
    uint32_t h = (kp >> hshift) & (SLOTS - 1);
    int probes = 0;
    if constexpr (kGbAggGroupProbe != 0) {
      // Probing by aligned GROUPS of four slots, one 16-byte LDS read per step: a wave goes round this loop as often as
      // its unluckiest lane, and at the wide form's load (0.6) the longest of 64 single-slot probe sequences is 8 - 10 steps
      // — each a dependent LDS round trip — where the longest group sequence is 2 - 3.  A key lives in the first group of
      // its sequence that had a free slot when it arrived (slots are never freed, tags never change): a step finds the
      // tag among the four, or claims the group's first empty slot; a slot read as empty may have been taken since — the
      // CAS then returns the owner, which is this key (found) or another (next slot).
      uint32_t g = h >> 2;
      bool found = false;
      for (; probes < kGbMaxProbes / 4; ++probes) {
        const arx_u32x4 four = *reinterpret_cast<const arx_u32x4*>(&t.tags[g << 2]);
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
          if (found) continue;
          uint32_t cur = four[sidx];
          if (cur == 0) {
            cur = atomicCAS(&t.tags[(g << 2) + sidx], 0u, tag);
            if (cur == 0) cur = tag;
          }
          if (cur == tag) {
            h = (g << 2) + sidx;
            found = true;
          }
        }
        if (found) break;
        g = (g + 1) & (SLOTS / 4 - 1);
      }
      if (!found) probes = kGbMaxProbes;
    } else {
      for (; probes < kGbMaxProbes; ++probes) {
        uint32_t cur = t.tags[h];
        if (cur == 0) {
          cur = atomicCAS(&t.tags[h], 0u, tag);
          if (cur == 0) cur = tag;
        }
        if (cur == tag) break;
        h = (h + 1) & (SLOTS - 1);
      }
    }
    if (probes < kGbMaxProbes) {
      atomicAdd(&t.sums[h], val);
      atomicAdd(&t.cnts[h], 1u);
    } else if (a.dense) {  // more ids than the LDS table holds: straight to the caller's arrays, still exact
      atomicAdd(&a.dense_sums[static_cast<uint32_t>(key)], val);
      atomicAdd(&a.dense_counts[static_cast<uint32_t>(key)], 1ull);
    } else if (a.emit_records != nullptr) {  // emit form: the row itself leaves as a partial of one value
      const int o = gb_dest(key, true, a.emit_parts);
      const unsigned long long at = atomicAdd(&a.emit_cursor[o], 1ull);
      if (at < static_cast<unsigned long long>(a.emit_capacity)) {
        ArxGroupPartial r{};
        r.sum = static_cast<int64_t>(val);
        r.count = 1;
        r.key = key;
        r.key_is_valid = 1;
        r.no_nulls = 1;
        a.emit_records[static_cast<int64_t>(o) * a.emit_capacity + static_cast<int64_t>(at)] = r;
      }
    } else {  // more groups than the LDS table holds: slow path, still exact
      const int64_t slot = gb_find_or_insert(v, key, &fresh);
      if (slot < 0) {
        atomicExch(&v.hdr->overflow, 1u);
      } else {
        atomicAdd(&v.sums[slot], val);
        atomicAdd(&v.counts[slot], 1ull);
      }
    }
  };
  if constexpr (WIDE && kGbAosRaw != 0) {
    // The pipeline registers hold the RAW quads (three 16-byte loads = four records) and the loads are unconditional:
    // a thread whose four records reach past the partition's end reads the window [hi - 4, hi) instead and shifts it
    // when it consumes it.  (With the loads inside an `if (whole quad) ... else ...`, as the first form had them, the
    // compiler waits for them before the branches join — `s_waitcnt vmcnt(0)` right behind the loads, no pipeline at all:
    // the kernel ran at one memory round trip per quad, 12.3 ms for 48 GB, whatever the number of rows in flight.)
    constexpr int G = U / 4;
    typedef arx_u32x4 __attribute__((aligned(4))) RecQuad;   // (a record starts at any multiple of 12 bytes)
    const int64_t hi4 = hi >= 4 ? hi - 4 : 0;
    arx_u32x4 cur[G][3], nxt[G][3];
    auto issue = [&](int64_t it0, arx_u32x4 (*raw)[3]) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int64_t r0 = lo + ((it0 / 4 + g) * THREADS + tid) * 4;
        // striped rooms: the thread's own aligned quad — it never crosses a chunk, and what it reads past `hi` is still
        // inside the chunk (rooms are whole chunks); a thread with nothing left re-reads the unit's first quad
        const int64_t ws = a.stripe != 0 ? gbp_rec_phys(a, q, static_cast<uint32_t>(r0 < hi ? r0 : lo)) : (r0 + 4 <= hi ? r0 : hi4);
        const RecQuad* rq = reinterpret_cast<const RecQuad*>(reinterpret_cast<const GbpRec*>(a.recs) + ws);
        raw[g][0] = rq[0];
        raw[g][1] = rq[1];
        raw[g][2] = rq[2];
      }
    };
    auto consume = [&](int64_t it0, arx_u32x4 (*raw)[3]) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int64_t r0 = lo + ((it0 / 4 + g) * THREADS + tid) * 4;
        if (r0 >= hi) continue;
        uint32_t d[12];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          d[e] = raw[g][0][e];
          d[4 + e] = raw[g][1][e];
          d[8 + e] = raw[g][2][e];
        }
        const int shift = (a.stripe != 0 || r0 + 4 <= hi) ? 0 : static_cast<int>(r0 - hi4);   // 1 .. 3 behind the partition's last whole quad
        for (int k = 0; k < shift; ++k) {
#pragma unroll
          for (int e = 0; e < 9; ++e) d[e] = d[e + 3];
        }
        if constexpr (kGbAggQuadProbe != 0) {
          // the four home slots are read together (one LDS round trip for the quad instead of one per row): a row whose
          // home slot already carries its tag — most rows, once a partition's groups are in — goes straight to its two
          // adds; the others take the probing path from their home slot
          uint32_t home[4], tg[4], seen[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t kp = gbp_hash(a, static_cast<int32_t>(d[3 * j]));
            tg[j] = kp & low_mask;
            home[j] = (kp >> hshift) & (SLOTS - 1);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) seen[j] = t.tags[home[j]];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (r0 + j >= hi) continue;
            const unsigned long long val = (static_cast<unsigned long long>(d[3 * j + 2]) << 32) | d[3 * j + 1];
            if (tg[j] != 0 && seen[j] == tg[j]) {
              atomicAdd(&t.sums[home[j]], val);
              atomicAdd(&t.cnts[home[j]], 1u);
            } else {
              process_row(static_cast<int32_t>(d[3 * j]), val);
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (r0 + j < hi) process_row(static_cast<int32_t>(d[3 * j]), (static_cast<unsigned long long>(d[3 * j + 2]) << 32) | d[3 * j + 1]);
          }
        }
      }
    };
    if (pipe && nit > 0) issue(0, nxt);
    for (int64_t it0 = 0; it0 < nit; it0 += U) {
      if (!pipe) issue(it0, nxt);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        cur[g][0] = nxt[g][0];
        cur[g][1] = nxt[g][1];
        cur[g][2] = nxt[g][2];
      }
      if (pipe && it0 + U < nit) issue(it0 + U, nxt);
      consume(it0, cur);
    }
  } else {
    if (pipe && nit > 0) load_batch(0, knext, vnext, oknext);
    for (int64_t it0 = 0; it0 < nit; it0 += U) {
      if (!pipe) load_batch(it0, knext, vnext, oknext);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        kbuf[u] = knext[u];
        vbuf[u] = vnext[u];
        okbuf[u] = oknext[u];
      }
      if (pipe && it0 + U < nit) load_batch(it0 + U, knext, vnext, oknext);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (okbuf[u]) process_row(kbuf[u], vbuf[u]);
      }
    }
  }
  __syncthreads();
  const uint32_t hi_bits = a.bits == 0 ? 0u : (q << low_bits);
  if (a.emit_records != nullptr) {   // (kernel-uniform)
    // the unit's groups as records, in the region of the rank that owns each key (hash(key) % ranks, as
    // arx_groupby_export_partitioned assigns them): ranks inside the unit by LDS atomics, ONE global atomic per (unit,
    // owner) for the place in the region — no probe of the HBM table, no two atomics per group, no export pass afterwards
    // A region's records of one unit lie in SLOT order (the order of the keys' partition hashes; a fixed order whatever the
    // waves' timing — it does not make the receiver's merge faster: 1.30 ms for 1e7 records either way, profiles/r05_l_*).
    // Wave w owns the slots [w S, (w + 1) S), S = SLOTS / waves, 64 at a time: first every wave counts its groups per
    // owner, one scan over the waves gives every (wave, owner) its place, then the wave walks its slots again and ranks them.
    constexpr int kWaves = THREADS / 64;
    constexpr int kPer = SLOTS / kWaves / 64;      // passes of 64 slots per wave
    const int wave = tid >> 6;
    const uint64_t below = (uint64_t(1) << lane) - 1;
    int owner[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int i = wave * (SLOTS / kWaves) + j * 64 + lane;
      const uint32_t tag = t.tags[i];
      owner[j] = tag != 0 ? gb_dest(gbp_unhash(a, hi_bits | tag), true, a.emit_parts) : -1;
    }
    const int zowner = t.zcnt != 0 ? gb_dest(gbp_unhash(a, hi_bits), true, a.emit_parts) : -1;   // the key whose tag is 0: the last wave's
    auto walk = [&](bool place) {
#pragma unroll
      for (int j = 0; j < kPer; ++j) {
        uint32_t at = 0;
        uint64_t todo = __ballot(owner[j] >= 0);
        while (todo != 0) {
          const int leader = __builtin_ctzll(todo);
          const int o = __shfl(owner[j], leader);
          const uint64_t same = __ballot(owner[j] == o);
          uint32_t old = 0;
          if (lane == leader) old = atomicAdd(&t.emit_wave[wave][o], static_cast<uint32_t>(__builtin_popcountll(same)));
          old = __shfl(old, leader);
          if (owner[j] == o) at = old + static_cast<uint32_t>(__builtin_popcountll(same & below));
          todo &= ~same;
        }
        if (place && owner[j] >= 0) {
          const int i = wave * (SLOTS / kWaves) + j * 64 + lane;
          const unsigned long long to = t.emit_base[owner[j]] + at;
          if (to < static_cast<unsigned long long>(a.emit_capacity)) {   // (else: the host sees the cursor past the capacity)
            ArxGroupPartial r{};
            r.sum = static_cast<int64_t>(t.sums[i]);
            r.count = static_cast<int64_t>(t.cnts[i]);
            r.key = gbp_unhash(a, hi_bits | t.tags[i]);
            r.key_is_valid = 1;
            r.no_nulls = 1;
            a.emit_records[static_cast<int64_t>(owner[j]) * a.emit_capacity + static_cast<int64_t>(to)] = r;
          }
        }
      }
      if (wave == kWaves - 1 && lane == 0 && zowner >= 0) {
        const uint32_t at = atomicAdd(&t.emit_wave[wave][zowner], 1u);
        const unsigned long long to = t.emit_base[zowner] + at;
        if (place && to < static_cast<unsigned long long>(a.emit_capacity)) {
          ArxGroupPartial r{};
          r.sum = static_cast<int64_t>(t.zsum);
          r.count = static_cast<int64_t>(t.zcnt);
          r.key = gbp_unhash(a, hi_bits);
          r.key_is_valid = 1;
          r.no_nulls = 1;
          a.emit_records[static_cast<int64_t>(zowner) * a.emit_capacity + static_cast<int64_t>(to)] = r;
        }
      }
    };
    walk(false);      // t.emit_wave[w][o] (zero since the unit's start) = groups of owner o among wave w's slots
    __syncthreads();
    if (tid < a.emit_parts) {   // -> where wave w's groups of owner o start inside the unit's run; the run's place in the region
      uint32_t run = 0;
      for (int w = 0; w < kWaves; ++w) {
        const uint32_t c = t.emit_wave[w][tid];
        t.emit_wave[w][tid] = run;
        run += c;
      }
      t.emit_base[tid] = run != 0 ? atomicAdd(&a.emit_cursor[tid], static_cast<unsigned long long>(run)) : 0ull;
    }
    __syncthreads();
    walk(true);
    return;
  }
  for (int i = tid; i < SLOTS + 1; i += THREADS) {
    uint32_t tag;
    unsigned long long sum;
    uint32_t cnt;
    if (i < SLOTS) {
      tag = t.tags[i];
      if (tag == 0) continue;
      sum = t.sums[i];
      cnt = t.cnts[i];
    } else {
      if (t.zcnt == 0) continue;
      tag = 0;
      sum = t.zsum;
      cnt = t.zcnt;
    }
    const int32_t key = gbp_unhash(a, hi_bits | tag);
    if (a.dense) {
      atomicAdd(&a.dense_sums[static_cast<uint32_t>(key)], sum);
      atomicAdd(&a.dense_counts[static_cast<uint32_t>(key)], static_cast<unsigned long long>(cnt));
      continue;
    }
    const int64_t slot = gb_find_or_insert(v, key, &fresh);
    if (slot < 0) {
      atomicExch(&v.hdr->overflow, 1u);
      continue;
    }
    atomicAdd(&v.sums[slot], sum);
    atomicAdd(&v.counts[slot], static_cast<unsigned long long>(cnt));
  }
  if (!a.dense) gb_publish_new_groups(v, fresh);
}

// ---- plan: how a slice of rows is laid out in the caller's workspace
struct GbpPlan {
  int bits, b1, b2;
  int wide;                 // the wide one-level form (b1 == 0, b2 == bits)
  uint32_t room;            // wide form: records every partition's room holds (see gbp_room_for)
  uint32_t stripe;          // rooms laid out in stripes of `stripe` records per partition (gbp_rec_phys); 0 = room after room
  int64_t slice_rows, chunk_rows, nchunks;
  size_t off_keys_a, off_vals_a, off_keys_b, off_vals_b, off_part_count, off_part_start,
      off_cursor2, off_cursor1, off_hist1, off_l1_start, off_l2_tile_start, off_agg_unit_start, off_part_end, off_overflow, total;
};

static Knob<int> g_gbp_min_rows{1 << 17};  // below this the direct HBM-atomics kernel is used
static Knob<int> g_gbp_bits{-1};           // -1 = from the capacity hint
static Knob<int> g_gbp_agg_pipe{1};
static Knob<int> g_gbp_xcd_map{1};         // XCD-contiguous work numbering: bit 0 level-2 scatter (-2.6 ms at 4e9 rows), bit 1 aggregate (+5 ms: off), bit 2 level-1 scatter (no effect)
static Knob<int> g_gbp_b1{-1};             // level-1 bits override (-1: half of the partition bits)
static Knob<int> g_gbp_chunks{2048};       // level-1 chunks = workgroups of the hist / scatter1 kernels
static Knob<int> g_gbp_wide_max_bits{kGbWideMaxBits};   // bins of the flat level the planner may ask for (A/B knob groupby_wide_max_bits; tests lower it)
static Knob<int> g_gbp_room_min_mean{1 << 14};   // rooms only for partitions of at least this many rows on average (knob groupby_wide_room_min_mean; tests lower it)
static Knob<int> g_gbp_sketch{1};          // the group count from a HyperLogLog sketch of the first groupby_probe_rows keys (0: round 3's probe slice on the two-level plan; A/B knob groupby_sketch)
static Knob<int> g_gbp_stripe{0};          // rooms in stripes of this many records, a multiple of 4 (A/B knob groupby_stripe; 0 = off)
static Knob<int> g_gbp_wide_rooms{1};      // the wide form without its histogram pass: fixed rooms per partition (A/B knob groupby_wide_rooms)
static Knob<int> g_gbp_wide{1};            // the wide one-level form where the group estimate allows it (A/B knob groupby_wide)
static Knob<int> g_gbp_wide_agg_chunk{1 << 21};   // rows per aggregate work unit of the wide form (A/B knob groupby_wide_agg_chunk_rows)
static Knob<int64_t> g_gbp_probe_rows{int64_t(1) << 25};   // rows of the probe slice that measures the group count (A/B knob groupby_probe_rows).  2^25: every group count the wide plan takes (<= 12.6M) still repeats most of its keys inside the probe; 0.5 ms less than 2^26 at 4e9 rows, 2^24 no better (profiles/r03_x_*)
static Knob<int> g_gbp_l1_global{1};       // level 1 with global cursors (one tile per workgroup; +3 % at 4e9 rows) instead of chunked exact offsets

static int gbp_bits_for(int64_t capacity) {
  if (g_gbp_bits >= 0) return std::min(int(g_gbp_bits), kGbMaxBits);
  // capacity = slots of the HBM table ~ 2x the distinct keys expected; aim at <= 2048 groups
  // per 4096-slot LDS table.  bits == 0: everything fits one table, no partitioning at all.
  const int64_t groups = std::max<int64_t>(1, capacity / 2);
  int bits = 0;
  while (bits < kGbMaxBits && (groups >> bits) > 2048) ++bits;
  return bits;
}

constexpr int64_t kGbMaxSlice = int64_t(1) << 30;  // row positions inside a slice are 32-bit
static Knob<int64_t> g_gbp_max_slice{kGbMaxSlice};     // A/B knob groupby_max_slice_rows
constexpr int64_t kGbHardMaxSlice = (int64_t(1) << 32) - (int64_t(1) << 26);   // (room for the rooms' slack below 2^32 positions)
static Knob<int64_t> g_gbp_wide_max_slice{kGbHardMaxSlice};   // A/B knob groupby_wide_max_slice_rows
static Knob<int> g_gbp_agg_chunk{1 << 18};             // A/B knob groupby_agg_chunk_rows (2^16: every group of a partition is flushed 2-8x per slice; 2^18: +4 %)

// Partition bits of the wide form for `groups` distinct keys (<= kGbWideMaxGroups per 8192-slot LDS table), or -1 when
// the flat level would need more than 2048 bins.
static int gbp_wide_bits_for(int64_t groups) {
  int bits = 1;
  while (bits <= g_gbp_wide_max_bits && ((groups + (int64_t(1) << bits) - 1) >> bits) > kGbWideMaxGroups) ++bits;
  return bits <= g_gbp_wide_max_bits ? bits : -1;
}

// Records a partition's room holds when the wide form runs without a histogram: the mean + 6 sigma of the binomial a
// bijective hash of DISTINCT keys gives, + 64.  Only worth it when the slack is small (mean >= 2^14: <= 5 %).
static uint32_t gbp_room_for(int64_t rows, int bits, int64_t stripe) {
  const int64_t mean = (rows + (int64_t(1) << bits) - 1) >> bits;
  if (mean < g_gbp_room_min_mean) return 0;
  int64_t sd = 1;
  while (sd * sd < mean) ++sd;
  int64_t room = mean + 6 * sd + 64;
  if (stripe > 0) room = (room / stripe + 1) * stripe;   // whole chunks (the aggregate over-reads inside a chunk)
  return (room << bits) < (int64_t(1) << 32) ? static_cast<uint32_t>(room) : 0;
}

// groups_hint: distinct keys expected in the rows to come (< 0: unknown — the capacity is the only bound).  The
// two-level plan is chosen from the capacity alone; the wide plan replaces it when the hint (or, without one, the
// capacity bound) says its 2048 tables of 8192 slots are enough.
static GbpPlan gbp_plan(int64_t slice_rows, int64_t capacity, int64_t groups_hint = -1, int dense_idbits = 0,
                        bool allow_rooms = true) {
  GbpPlan p{};
  p.bits = gbp_bits_for(capacity);
  if (p.bits <= 8) {
    p.b1 = p.bits;
    p.b2 = 0;
  } else {
    // As few level-1 bits as the 256-bin level 2 allows: level 1 scatters over the whole slice, where long runs pay
    // (32 bins: ~128-record runs), level 2 works inside a partition whose short runs meet in one L2 (xcd_contiguous) —
    // 4e9 rows / 1e7 keys: 7 + 6 bits 61.6 ms, 5 + 8 bits 54.4 ms; 1e6 keys: 5 + 4 bits 60.8 ms, 3 + 6 bits 54.0 ms
    p.b1 = std::max(p.bits - 8, std::min(3, p.bits - 1));
    if (g_gbp_b1 > 0) p.b1 = std::max(p.bits - 8, std::min(int(g_gbp_b1), std::min(8, p.bits - 1)));
    p.b2 = p.bits - p.b1;
  }
  // the wide one-level plan replaces a two-level plan when its tables are known to be enough; groupby_wide = 2 forces
  // it (tests / A-B) with groupby_partition_bits as its bin count
  int wb = -1;
  if (g_gbp_wide == 2) {
    wb = int(g_gbp_bits) >= 1 ? std::min(int(g_gbp_bits), kGbWideMaxBits) : kGbWideMaxBits;
  } else if (g_gbp_wide && p.bits > 8 && g_gbp_bits < 0) {
    wb = gbp_wide_bits_for(groups_hint >= 0 ? groups_hint : std::max<int64_t>(1, capacity / 2));
    if (dense_idbits > 0) {
      // dense ids: id << (32 - idbits) is the bijection, a partition's ids are consecutive and map to distinct slots
      // of the table — what has to fit is the partition's slice of the id space
      wb = std::max(1, dense_idbits - 13);
      if (wb > g_gbp_wide_max_bits) wb = -1;
    }
  }
  if (wb > 0) {
    p.wide = 1;
    p.bits = wb;
    p.b1 = 0;
    p.b2 = wb;
  }
  p.slice_rows = slice_rows;
  const int64_t ntiles = ceil_div(std::max<int64_t>(slice_rows, 1), kGbTile);
  const int64_t chunk_tiles = std::max<int64_t>(1, ceil_div(ntiles, int(g_gbp_chunks)));
  p.chunk_rows = chunk_tiles * kGbTile;
  p.nchunks = ceil_div(ntiles, chunk_tiles);
  auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
  const size_t n = static_cast<size_t>(std::max<int64_t>(slice_rows, 1));
  const size_t nparts = size_t(1) << p.bits;
  const size_t nb1 = size_t(1) << p.b1;
  size_t o = 0;
  // rooms need keys that spread like distinct hashed keys: the keyed table, never the dense ids of the hash_sum vtable
  // (id << shift puts all rows of the low ids into the low partitions)
  if (p.wide && g_gbp_wide_rooms && allow_rooms && dense_idbits == 0) {
    const int stripe = g_gbp_stripe;
    p.room = gbp_room_for(slice_rows, p.bits, stripe);
    p.stripe = p.room != 0 ? static_cast<uint32_t>(stripe) : 0u;
  }
  if (p.wide) {
    const size_t recs = p.room ? (static_cast<size_t>(p.room) << p.bits) : n;
    p.off_keys_a = o; o = align(o + recs * 12);   // the 12-byte records
    p.off_vals_a = p.off_keys_b = p.off_vals_b = o;
  } else {
    p.off_keys_a = o; o = align(o + (p.bits ? n * 4 : 0));
    p.off_vals_a = o; o = align(o + (p.bits ? n * 8 : 0));
    p.off_keys_b = o; o = align(o + (p.b2 ? n * 4 : 0));
    p.off_vals_b = o; o = align(o + (p.b2 ? n * 8 : 0));
  }
  p.off_part_count = o; o = align(o + nparts * 4);
  p.off_part_start = o; o = align(o + (nparts + 1) * 4);
  p.off_cursor2 = o; o = align(o + nparts * 4);
  p.off_cursor1 = o; o = align(o + nb1 * 4);
  p.off_hist1 = o; o = align(o + nb1 * static_cast<size_t>(kGbMaxChunks) * 4);
  p.off_l1_start = o; o = align(o + (nb1 + 1) * 4);
  p.off_l2_tile_start = o; o = align(o + (nb1 + 1) * 4);
  p.off_agg_unit_start = o; o = align(o + (nparts + 1) * 4);
  p.off_part_end = o; o = align(o + nparts * 4);
  p.off_overflow = o; o = align(o + 4);
  p.total = o;
  return p;
}

// The workspace pointers and knobs of a slice's plan.
static void gbp_bind(GbpArgs& a, const GbpPlan& plan, uint8_t* w) {
  a.bits = plan.bits;
  a.b1 = plan.b1;
  a.b2 = plan.b2;
  a.wide = plan.wide;
  a.chunk_rows = plan.chunk_rows;
  a.nchunks = plan.nchunks;
  a.recs = w + plan.off_keys_a;
  a.keys_a = reinterpret_cast<int32_t*>(w + plan.off_keys_a);
  a.vals_a = reinterpret_cast<int64_t*>(w + plan.off_vals_a);
  a.keys_b = reinterpret_cast<int32_t*>(w + plan.off_keys_b);
  a.vals_b = reinterpret_cast<int64_t*>(w + plan.off_vals_b);
  a.part_count = reinterpret_cast<uint32_t*>(w + plan.off_part_count);
  a.part_start = reinterpret_cast<uint32_t*>(w + plan.off_part_start);
  a.cursor2 = reinterpret_cast<uint32_t*>(w + plan.off_cursor2);
  a.cursor1 = reinterpret_cast<uint32_t*>(w + plan.off_cursor1);
  a.hist1 = reinterpret_cast<uint32_t*>(w + plan.off_hist1);
  a.l1_start = reinterpret_cast<uint32_t*>(w + plan.off_l1_start);
  a.l2_tile_start = reinterpret_cast<uint32_t*>(w + plan.off_l2_tile_start);
  a.agg_unit_start = reinterpret_cast<uint32_t*>(w + plan.off_agg_unit_start);
  a.part_end = reinterpret_cast<uint32_t*>(w + plan.off_part_end);
  a.overflow = reinterpret_cast<uint32_t*>(w + plan.off_overflow);
  a.room = plan.room;
  a.stripe = plan.room != 0 ? plan.stripe : 0u;
  a.agg_pipe = g_gbp_agg_pipe;
  a.xcd_map = g_gbp_xcd_map;
  a.agg_chunk = static_cast<uint32_t>(plan.wide ? g_gbp_wide_agg_chunk : g_gbp_agg_chunk);
}

// Largest slice (multiple of the tile) whose plan fits `ws_bytes`; 0 if not even one chunk fits.
static int64_t gbp_slice_for(size_t ws_bytes, int64_t n, int64_t capacity, int64_t groups_hint = -1) {
  // The two-level plan works in slices of <= 2^30 rows (larger ones are slower: its local levels lose their locality);
  // the wide plan has one flat level whatever the slice, flushes every group once per slice and aggregate unit, and
  // needs half the row scratch: its slices go up to the 32-bit position limit (4e9 rows / 1e7 keys: 50.2 -> 48.6 ms).
  const GbpPlan probe = gbp_plan(kGbTile, capacity, groups_hint);
  int64_t hi = std::min<int64_t>(n, probe.wide ? int64_t(g_gbp_wide_max_slice) : int64_t(g_gbp_max_slice));
  if (gbp_plan(hi, capacity, groups_hint).total <= ws_bytes) return hi;
  const size_t fixed = probe.total;
  if (fixed > ws_bytes) return 0;
  if (probe.bits == 0) return hi;  // no row scratch at all
  const int two = probe.b2 && !probe.wide ? 2 : 1;
  int64_t rows = static_cast<int64_t>((ws_bytes - fixed) / (12 * two));
  rows = rows / kGbTile * kGbTile;
  while (rows > 0 && gbp_plan(rows, capacity, groups_hint).total > ws_bytes) rows -= kGbTile;
  return std::min(rows, hi);
}

// Which plan the slices of the partitioned consume ran (arx_get_counter): the plan is chosen from estimates, and a test
// or a bench that means to measure one plan must be able to see that it did.
static std::atomic<int64_t> g_gbp_slices_direct{0}, g_gbp_slices_one_level{0}, g_gbp_slices_two_level{0},
    g_gbp_slices_wide{0}, g_gbp_slices_probe{0}, g_gbp_slices_rooms{0}, g_gbp_rooms_overflows{0};
constexpr int kGbpRoomsOverflow = -1000;   // gbp_run_slice: a partition outgrew its room, nothing consumed yet

static int get_groupby_lines_counter(const char* name, int64_t* out);   // groupby_lines.h

int get_groupby_counter(const char* name, int64_t* out) {
  if (strcmp(name, "groupby_slices_direct") == 0) *out = g_gbp_slices_direct.load();
  else if (strcmp(name, "groupby_slices_one_level") == 0) *out = g_gbp_slices_one_level.load();
  else if (strcmp(name, "groupby_slices_two_level") == 0) *out = g_gbp_slices_two_level.load();
  else if (strcmp(name, "groupby_slices_wide") == 0) *out = g_gbp_slices_wide.load();
  else if (strcmp(name, "groupby_slices_probe") == 0) *out = g_gbp_slices_probe.load();
  else if (strcmp(name, "groupby_slices_rooms") == 0) *out = g_gbp_slices_rooms.load();
  else if (strcmp(name, "groupby_rooms_overflows") == 0) *out = g_gbp_rooms_overflows.load();
  else return get_groupby_lines_counter(name, out);
  return 1;
}

template <bool HAS_NULLS>
static int gbp_run_slice(const GroupbyView& v, GbpArgs a, const GbpPlan& plan, hipStream_t st, bool redo = false) {
  if (HAS_NULLS && !redo) {   // (a slice redone after a rooms overflow: its null rows are in the table already)
    hipLaunchKernelGGL(gbp_null_rows_kernel, dim3(gb_grid(a.n / 8 + 1)), dim3(kBlock), 0, st, v, a);
    ARX_CHECK_LAUNCH("gbp_null_rows_kernel");
  }
  (a.bits == 0 ? g_gbp_slices_direct : a.wide ? g_gbp_slices_wide : a.b2 > 0 ? g_gbp_slices_two_level : g_gbp_slices_one_level)
      .fetch_add(1, std::memory_order_relaxed);
  if (a.wide && a.room != 0) g_gbp_slices_rooms.fetch_add(1, std::memory_order_relaxed);
  if (a.bits == 0) {
    const unsigned units = static_cast<unsigned>(ceil_div(a.n, a.agg_chunk));
    hipLaunchKernelGGL((gbp_aggregate_kernel<true, HAS_NULLS>), dim3(units), dim3(kGbThreads), 0, st, v, a,
                       a.keys, a.values);
    ARX_CHECK_LAUNCH("gbp_aggregate_kernel");
    return ARX_OK;
  }
  const unsigned nch = static_cast<unsigned>(a.nchunks);
  const int nparts = 1 << a.bits;
  if (a.wide && a.room != 0) {
    // no histogram: rooms -> scatter -> (the host looks at the overflow flag) -> ends + work units -> aggregate
    hipLaunchKernelGGL(gbp_rooms_init_kernel, dim3(1), dim3(1024), 0, st, a);
    ARX_CHECK_LAUNCH("gbp_rooms_init_kernel");
    hipLaunchKernelGGL((gbp_scatter_wide_kernel<HAS_NULLS>), dim3(static_cast<unsigned>(ceil_div(a.n, kGbWideTile))),
                       dim3(kGbWideThreads), 0, st, a);
    ARX_CHECK_LAUNCH("gbp_scatter_wide_kernel");
    uint32_t overflow = 0;
    ARX_HIP(hipMemcpyAsync(&overflow, a.overflow, 4, hipMemcpyDeviceToHost, st));
    ARX_HIP(hipStreamSynchronize(st));
    if (overflow != 0) {
      g_gbp_rooms_overflows.fetch_add(1, std::memory_order_relaxed);
      return kGbpRoomsOverflow;   // nothing has touched the table yet: the caller redoes the slice with counted partitions
    }
    hipLaunchKernelGGL(gbp_rooms_scan_kernel, dim3(1), dim3(1024), 0, st, a);
    ARX_CHECK_LAUNCH("gbp_rooms_scan_kernel");
    const unsigned wunits = static_cast<unsigned>(ceil_div(a.n, a.agg_chunk) + nparts);
    hipLaunchKernelGGL((gbp_aggregate_kernel<false, false, kGbWideSlots, kGbWideThreads, true, kGbWideAggU>), dim3(wunits),
                       dim3(kGbWideThreads), 0, st, v, a, nullptr, nullptr);
    ARX_CHECK_LAUNCH("gbp_aggregate_kernel (wide, rooms)");
    return ARX_OK;
  }
  ARX_HIP(hipMemsetAsync(a.part_count, 0, static_cast<size_t>(nparts) * 4, st));
  hipLaunchKernelGGL((gbp_hist_kernel<HAS_NULLS>), dim3(nch), dim3(kGbThreads), 0, st, a);
  ARX_CHECK_LAUNCH("gbp_hist_kernel");
  hipLaunchKernelGGL(gbp_scan_a_kernel, dim3(1), dim3(1024), 0, st, a);
  ARX_CHECK_LAUNCH("gbp_scan_a_kernel");
  if (a.wide) {
    hipLaunchKernelGGL((gbp_scatter_wide_kernel<HAS_NULLS>), dim3(static_cast<unsigned>(ceil_div(a.n, kGbWideTile))),
                       dim3(kGbWideThreads), 0, st, a);
    ARX_CHECK_LAUNCH("gbp_scatter_wide_kernel");
    const unsigned wunits = static_cast<unsigned>(ceil_div(a.n, a.agg_chunk) + nparts);
    hipLaunchKernelGGL((gbp_aggregate_kernel<false, false, kGbWideSlots, kGbWideThreads, true, kGbWideAggU>), dim3(wunits),
                       dim3(kGbWideThreads), 0, st, v, a, nullptr, nullptr);
    ARX_CHECK_LAUNCH("gbp_aggregate_kernel (wide)");
    return ARX_OK;
  }
  hipLaunchKernelGGL(gbp_scan_b_kernel, dim3(1u << a.b1), dim3(1024), 0, st, a);
  ARX_CHECK_LAUNCH("gbp_scan_b_kernel");
  if (g_gbp_l1_global) {
    hipLaunchKernelGGL((gbp_scatter1g_kernel<HAS_NULLS>), dim3(static_cast<unsigned>(ceil_div(a.n, kGbTile))),
                       dim3(kGbThreads), 0, st, a);
  } else {
    hipLaunchKernelGGL((gbp_scatter1_kernel<HAS_NULLS>), dim3(nch), dim3(kGbThreads), 0, st, a);
  }
  ARX_CHECK_LAUNCH("gbp_scatter1_kernel");
  const int32_t* fk = a.keys_a;
  const int64_t* fv = a.vals_a;
  if (a.b2 > 0) {
    const unsigned grid2 = static_cast<unsigned>(ceil_div(a.n, kGbTile) + (int64_t(1) << a.b1));
    hipLaunchKernelGGL(gbp_scatter2_kernel, dim3(grid2), dim3(kGbThreads), 0, st, a);
    ARX_CHECK_LAUNCH("gbp_scatter2_kernel");
    fk = a.keys_b;
    fv = a.vals_b;
  }
  const unsigned units = static_cast<unsigned>(ceil_div(a.n, a.agg_chunk) + nparts);
  hipLaunchKernelGGL((gbp_aggregate_kernel<false, false>), dim3(units), dim3(kGbThreads), 0, st, v, a, fk, fv);
  ARX_CHECK_LAUNCH("gbp_aggregate_kernel");
  return ARX_OK;
}

// ---- how many DISTINCT keys?  A HyperLogLog sketch of a prefix of the key column (round 4): 2^14 registers, per
// workgroup in LDS (ds_max), folded into the global registers by the few lanes that raise one.  The capacity only bounds
// the group count from above; round 3 measured it by running the first 2^25 rows as a "probe slice" on the safe two-level
// plan (2.4 ms whatever n: a third of one rank's time at N / 8 rows) — the sketch reads those keys once (128 MB, ~0.05 ms)
// and aggregates nothing, so the prefix goes through the plan the estimate selects like every other row.
constexpr int kHllBits = 14;
constexpr int kHllRegs = 1 << kHllBits;

__global__ __launch_bounds__(1024) void gbp_hll_kernel(const int32_t* __restrict__ keys, int64_t n, unsigned int* __restrict__ regs) {
  __shared__ unsigned int local[kHllRegs];
  for (int i = threadIdx.x; i < kHllRegs; i += 1024) local[i] = 0;
  __syncthreads();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 1024;
  auto fold = [&](int32_t key) {
    uint64_t z = static_cast<uint64_t>(static_cast<uint32_t>(key)) + 0x9E3779B97F4A7C15ull;   // splitmix64 of the key
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const unsigned int idx = static_cast<unsigned int>(z >> (64 - kHllBits));
    const unsigned int rank = static_cast<unsigned int>(__builtin_clzll((z << kHllBits) | (uint64_t(1) << (kHllBits - 1)))) + 1u;
    if (local[idx] < rank) atomicMax(&local[idx], rank);
  };
  // eight keys in flight per thread (one per iteration left the kernel waiting for its own load: 134 MB in 155 us)
  int64_t i = static_cast<int64_t>(blockIdx.x) * 1024 + threadIdx.x;
  for (; i + 7 * stride < n; i += 8 * stride) {
    int32_t k[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) k[u] = keys[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) fold(k[u]);
  }
  for (; i < n; i += stride) fold(keys[i]);
  __syncthreads();
  for (int i = threadIdx.x; i < kHllRegs; i += 1024) {
    const unsigned int r = local[i];
    if (r != 0 && regs[i] < r) atomicMax(&regs[i], r);
  }
}

// the sketch's estimate of the distinct keys among keys[0, n) (regs: kHllRegs x 4 device bytes, synchronous)
static int gbp_estimate_distinct(const int32_t* keys, int64_t n, unsigned int* regs, int64_t* out, hipStream_t st) {
  ARX_HIP(hipMemsetAsync(regs, 0, sizeof(unsigned int) * kHllRegs, st));
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, 1024 * 64), 1024)));
  hipLaunchKernelGGL(gbp_hll_kernel, dim3(grid), dim3(1024), 0, st, keys, n, regs);
  ARX_CHECK_LAUNCH("gbp_hll_kernel");
  std::vector<unsigned int> h(kHllRegs);
  ARX_HIP(hipMemcpyAsync(h.data(), regs, sizeof(unsigned int) * kHllRegs, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  double sum = 0;
  int zeros = 0;
  for (unsigned int r : h) {
    sum += std::ldexp(1.0, -static_cast<int>(r));
    zeros += r == 0;
  }
  const double m = kHllRegs;
  double e = 0.7213 / (1.0 + 1.079 / m) * m * m / sum;
  if (e <= 2.5 * m && zeros != 0) e = m * std::log(m / zeros);   // linear counting for small cardinalities
  *out = static_cast<int64_t>(e);
  return ARX_OK;
}

static int read_header(void* state, GroupbyHeader* h, hipStream_t st) {
  ARX_HIP(hipMemcpyAsync(h, state, sizeof(GroupbyHeader), hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  return ARX_OK;
}

static int set_groupby_lines_option(const char* name, int64_t value);   // groupby_lines.h

int set_groupby_option(const char* name, int64_t value) {
  if (set_groupby_lines_option(name, value)) return 1;
  if (strcmp(name, "groupby_partition_min_rows") == 0) {
    g_gbp_min_rows = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(value, INT32_MAX)));
    return 1;
  }
  if (strcmp(name, "groupby_chunks") == 0) {
    g_gbp_chunks = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(value, kGbMaxChunks)));
    return 1;
  }
  if (strcmp(name, "groupby_max_slice_rows") == 0) {
    g_gbp_max_slice = std::max<int64_t>(kGbTile, std::min<int64_t>(value, kGbHardMaxSlice)) / kGbTile * kGbTile;
    return 1;
  }
  if (strcmp(name, "groupby_agg_chunk_rows") == 0) {
    int lg = 12;
    while (lg < 24 && (int64_t(1) << lg) < value) ++lg;
    g_gbp_agg_chunk = 1 << lg;
    return 1;
  }
  if (strcmp(name, "groupby_wide") == 0) {
    g_gbp_wide = value < 0 ? 0 : static_cast<int>(std::min<int64_t>(value, 2));
    return 1;
  }
  if (strcmp(name, "groupby_wide_agg_chunk_rows") == 0) {
    int lg = 14;
    while (lg < 26 && (int64_t(1) << lg) < value) ++lg;
    g_gbp_wide_agg_chunk = 1 << lg;
    return 1;
  }
  if (strcmp(name, "groupby_wide_room_min_mean") == 0) {
    g_gbp_room_min_mean = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(value, INT32_MAX)));
    return 1;
  }
  if (strcmp(name, "groupby_sketch") == 0) {
    g_gbp_sketch = value != 0;
    return 1;
  }
  if (strcmp(name, "groupby_stripe") == 0) {
    g_gbp_stripe = value <= 0 ? 0 : static_cast<int>(std::max<int64_t>(4, std::min<int64_t>(value & ~int64_t(3), 1 << 20)));
    return 1;
  }
  if (strcmp(name, "groupby_wide_rooms") == 0) {
    g_gbp_wide_rooms = value != 0;
    return 1;
  }
  if (strcmp(name, "groupby_wide_max_slice_rows") == 0) {
    g_gbp_wide_max_slice = std::max<int64_t>(kGbTile, std::min<int64_t>(value, kGbHardMaxSlice)) / kGbTile * kGbTile;
    return 1;
  }
  if (strcmp(name, "groupby_wide_max_bits") == 0) {
    g_gbp_wide_max_bits = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(value, kGbWideMaxBits)));
    return 1;
  }
  if (strcmp(name, "groupby_probe_rows") == 0) {
    g_gbp_probe_rows = std::max<int64_t>(kGbTile, value);
    return 1;
  }
  if (strcmp(name, "groupby_l1_global") == 0) {
    g_gbp_l1_global = value != 0;
    return 1;
  }
  if (strcmp(name, "groupby_b1") == 0) {
    g_gbp_b1 = static_cast<int>(value);
    return 1;
  }
  if (strcmp(name, "groupby_xcd_map") == 0) {
    g_gbp_xcd_map = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(value, 7)));
    return 1;
  }
  if (strcmp(name, "groupby_agg_pipe") == 0) {
    g_gbp_agg_pipe = value != 0;
    return 1;
  }
  if (strcmp(name, "groupby_partition_bits") == 0) {
    g_gbp_bits = static_cast<int>(value);
    return 1;
  }
  return 0;
}

// hash_any / hash_all Finalize (GroupedBooleanAggregator::Finalize, kernels/hash_aggregate.cc:1321-1350) from three dense
// counts per group — valid rows, null rows, valid rows whose value is true: a wave packs 64 groups per ballot.
//   value    any: a true was seen          all: no false was seen
//   valid    counts >= min_count, and (unless skip_nulls) no null seen OR the value is already decided by what was seen
//            (any: a true; all: a false) — AdjustForMinCount's BitmapOr / BitmapOrNot (:1376-1398)
__global__ __launch_bounds__(256) void hash_bool_finalize_kernel(const long long* __restrict__ n_valid,
                                                                const long long* __restrict__ n_null,
                                                                const long long* __restrict__ n_true, int64_t g, int is_all,
                                                                int skip_nulls, uint32_t min_count,
                                                                uint64_t* __restrict__ out_values,
                                                                uint64_t* __restrict__ out_validity,
                                                                unsigned long long* __restrict__ valid_count) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (g + 63) / 64;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * 256) >> 6;
  unsigned long long mine = 0;
  for (int64_t w = wave; w < nwords; w += nwaves) {
    const int64_t i = w * 64 + lane;
    bool value = false, valid = false;
    if (i < g) {
      const long long nv = n_valid[i], nn = n_null[i], nt = n_true[i];
      value = is_all ? (nt == nv) : (nt > 0);
      valid = nv >= static_cast<long long>(min_count);
      if (!skip_nulls) valid = valid && (nn == 0 || (is_all ? !value : value));
    }
    const uint64_t vw = __ballot(value), ow = __ballot(valid);
    if (lane == 0) {
      out_values[w] = vw;
      out_validity[w] = ow;
      mine += static_cast<unsigned long long>(__popcll(ow));
    }
  }
  __shared__ unsigned long long part[4];
  if (lane == 0) part[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (valid_count != nullptr && threadIdx.x == 0) {
    const unsigned long long total = part[0] + part[1] + part[2] + part[3];
    if (total != 0) atomicAdd(valid_count, total);
  }
}

// Smallest and largest int32 key of a column (slots of null keys included: they only widen the range).  A caller
// that must size a table for "as many groups as rows" gets a much tighter bound from max - min + 1 when the keys are
// ids or codes — the usual case for an int32 key column.
__global__ __launch_bounds__(256) void groupby_key_range_kernel(const int32_t* __restrict__ keys, int64_t n,
                                                               int32_t* __restrict__ out_min_max) {
  int32_t lo = INT32_MAX, hi = INT32_MIN;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
    const int32_t k = keys[i];
    lo = k < lo ? k : lo;
    hi = k > hi ? k : hi;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int32_t ol = __shfl_xor(lo, d, 64), oh = __shfl_xor(hi, d, 64);
    lo = ol < lo ? ol : lo;
    hi = oh > hi ? oh : hi;
  }
  // one atomic pair per workgroup (atomics on one address serialise behind the L2)
  __shared__ int32_t wlo[4], whi[4];
  if ((threadIdx.x & 63) == 0) {
    wlo[threadIdx.x >> 6] = lo;
    whi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int wv = 1; wv < 4; ++wv) {
      lo = wlo[wv] < lo ? wlo[wv] : lo;
      hi = whi[wv] > hi ? whi[wv] : hi;
    }
    atomicMin(&out_min_max[0], lo);
    atomicMax(&out_min_max[1], hi);
  }
}

#include "groupby_lines.h"

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

size_t arx_groupby_consume_workspace_bytes(int64_t length, int64_t capacity) {
  if (length < g_gbp_min_rows || length <= 0) return 0;
  // the plan the capacity alone selects, in its slices — and, where a measured group count may select the wide plan
  // later (arx_groupby_sum_i64_consume's probe slice), room for that plan's larger slices
  size_t need = gbp_plan(std::min<int64_t>(length, gbp_plan(kGbTile, capacity).wide ? int64_t(g_gbp_wide_max_slice) : int64_t(g_gbp_max_slice)),
                         capacity).total;
  if (g_gbp_wide && g_gbp_bits < 0 && !gbp_plan(kGbTile, capacity).wide && gbp_plan(kGbTile, capacity).b2 > 0) {
    // (the rooms' slack grows with the partition count: size for the fewest and for the most partitions a hint can select,
    //  so that ONE slice takes all the rows whatever the estimate — round 3 sized for the fewest only, which fit because
    //  the probe slice had taken 2^25 rows first)
    const int64_t few = int64_t(kGbWideMaxGroups) << 1;
    const int64_t many = int64_t(kGbWideMaxGroups) << std::min<int>(int(g_gbp_wide_max_bits), kGbWideMaxBits);
    const int64_t rows = std::min<int64_t>(length, int64_t(g_gbp_wide_max_slice));
    need = std::max({need, gbp_plan(rows, capacity, few).total, gbp_plan(rows, capacity, many).total});
  }
  return std::max(need, length >= g_gbl_min_rows ? gbl_workspace_bytes(length, true) : size_t(0));
}

struct GbpEmit {   // where the emit form writes (arx_groupby_sum_i64_consume_partials)
  ArxGroupPartial* records;
  unsigned long long* cursor;
  int parts;
  int64_t capacity;
};
static int gb_sum_i64_consume(void* state, int64_t capacity, const ArxSpan* keys_i32, const ArxSpan* values_i64, void* ws, size_t ws_bytes,
                              void* stream, const GbpEmit* emit);

int arx_groupby_sum_i64_consume(void* state, int64_t capacity, const ArxSpan* keys_i32,
                                const ArxSpan* values_i64, void* ws, size_t ws_bytes,
                                void* stream) {
  return gb_sum_i64_consume(state, capacity, keys_i32, values_i64, ws, ws_bytes, stream, nullptr);
}

int64_t arx_groupby_partials_capacity(int64_t num_rows, int64_t capacity, int num_parts) {
  if (num_parts < 1) return 0;
  // a work unit of the aggregate writes every group it met once.  A partition's first unit: the groups, at most
  // capacity / 2, once per slice; every further unit of a partition (one per 2^18 rows or more) at most the 4097 groups
  // its LDS table holds: num_rows / 64.  hash(key) % num_parts spreads them evenly (+ 25 %, + 65536 for a few groups that
  // all land on one rank).  Rows that find no place in an LDS table leave as records of their own; a shard with many of
  // those overflows its regions and goes through the table (ARX_CAPACITY_ERROR).
  const int64_t slices = std::max<int64_t>(1, (num_rows + kGbHardMaxSlice - 1) / kGbHardMaxSlice);
  const int64_t bound = std::min<int64_t>(num_rows, slices * std::max<int64_t>(capacity / 2, 1) + num_rows / 64);
  return bound / num_parts + bound / num_parts / 4 + 65536;
}

int arx_groupby_sum_i64_consume_partials(void* state, int64_t capacity, const ArxSpan* keys_i32, const ArxSpan* values_i64, void* ws,
                                         size_t ws_bytes, int num_parts, ArxGroupPartial* out_records, int64_t records_per_part,
                                         int64_t* out_part_counts, void* stream) {
  if (keys_i32 == nullptr || values_i64 == nullptr || out_records == nullptr || out_part_counts == nullptr) {
    set_error("NULL argument to arx_groupby_sum_i64_consume_partials");
    return ARX_INVALID;
  }
  if (num_parts < 1 || num_parts > kGbEmitMaxParts || records_per_part < 1) {
    set_error("arx_groupby_sum_i64_consume_partials: 1 to %d parts, a positive region size", kGbEmitMaxParts);
    return ARX_INVALID;
  }
  if ((keys_i32->null_count != 0 && keys_i32->validity != nullptr) || (values_i64->null_count != 0 && values_i64->validity != nullptr)) {
    set_error("arx_groupby_sum_i64_consume_partials: rows with nulls go through the table (arx_groupby_sum_i64_consume + export)");
    return ARX_NOT_IMPLEMENTED;
  }
  hipStream_t st = as_stream(stream);
  // the cursors live in out_part_counts (device int64[num_parts]): zeroed here, read back at the end
  ARX_HIP(hipMemsetAsync(out_part_counts, 0, static_cast<size_t>(num_parts) * 8, st));
  GbpEmit emit{out_records, reinterpret_cast<unsigned long long*>(out_part_counts), num_parts, records_per_part};
  const int rc = gb_sum_i64_consume(state, capacity, keys_i32, values_i64, ws, ws_bytes, stream, &emit);
  if (rc != ARX_OK) return rc;
  int64_t counts[kGbEmitMaxParts];
  ARX_HIP(hipMemcpyAsync(counts, out_part_counts, static_cast<size_t>(num_parts) * 8, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  for (int p = 0; p < num_parts; ++p) {
    if (counts[p] > records_per_part) {
      set_error("arx_groupby_sum_i64_consume_partials: region %d holds %lld records, %lld arrived (use the table path)", p,
                static_cast<long long>(records_per_part), static_cast<long long>(counts[p]));
      return ARX_CAPACITY_ERROR;
    }
  }
  return ARX_OK;
}

static int gb_sum_i64_consume(void* state, int64_t capacity, const ArxSpan* keys_i32, const ArxSpan* values_i64, void* ws, size_t ws_bytes,
                              void* stream, const GbpEmit* emit) {
  if ((state == nullptr && emit == nullptr) || keys_i32 == nullptr || values_i64 == nullptr) {
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
  // (the emit form never touches the table — no null rows, no row or group that goes to it —: its view is empty)
  GroupbyView v = emit != nullptr ? GroupbyView{} : gb_view(state, capacity);
  const int32_t* k = static_cast<const int32_t*>(keys_i32->data) + keys_i32->offset;
  const int64_t* val = static_cast<const int64_t*>(values_i64->data) + values_i64->offset;
  const void* kbm = keys_i32->null_count != 0 ? keys_i32->validity : nullptr;
  const void* vbm = values_i64->null_count != 0 ? values_i64->validity : nullptr;

  // ---- partitioned path: needs scratch; slices bound the scratch and keep positions 32-bit
  const int64_t slice = (ws != nullptr && n >= g_gbp_min_rows && (reinterpret_cast<uint64_t>(ws) & 255) == 0)
                            ? gbp_slice_for(ws_bytes, n, capacity)
                            : 0;
  if (slice >= kGbTile && slice >= std::min<int64_t>(n, 1 << 16)) {
    uint8_t* w = static_cast<uint8_t*>(ws);
    // Round 6: keys from a narrow range (ids, codes) take the LINES plan — partitions are slices of the key range, every
    // store of the scatter a whole 128-byte line, direct-indexed LDS tables (groupby_lines.h).  It declines (a range too
    // wide, a room that overflowed, a hot key) before anything touched the table; the plans below take the rows then.
    if (g_gbl && emit == nullptr && n >= g_gbl_min_rows) {
      const Bits kb = make_bits(kbm, keys_i32->offset, n), vb = make_bits(vbm, values_i64->offset, n);
      const int rc = (kbm != nullptr || vbm != nullptr) ? gbl_try<true>(v, k, val, kb, vb, n, w, ws_bytes, st)
                                                          : gbl_try<false>(v, k, val, kb, vb, n, w, ws_bytes, st);
      if (rc != kGblDeclined) return rc;
    }
    // The capacity only bounds the number of groups from above (a power of two > 2 G: G is anywhere in its upper half).
    // When that bound asks for the two-level plan but the wide one-level plan is within reach, the first 2^26 rows run
    // as a probe slice on the safe plan; the distinct keys the table holds after it — if the probe saw most of its
    // keys more than once — are the estimate the remaining slices are planned with.  A wrong estimate costs time,
    // never exactness (rows that find no room in an LDS table go to the HBM table).
    const GbpPlan unhinted = gbp_plan(slice, capacity);
    const int64_t probe_rows = std::max<int64_t>(kGbTile, int64_t(g_gbp_probe_rows) / kGbTile * kGbTile);
    const bool sketch = g_gbp_sketch != 0 && g_gbp_wide && g_gbp_bits < 0 && unhinted.b2 > 0 && !unhinted.wide && n >= 2 * probe_rows;
    const bool probe = emit == nullptr && !sketch && g_gbp_wide && g_gbp_bits < 0 && unhinted.b2 > 0 && !unhinted.wide && n >= 4 * probe_rows;   // (the probe reads the table's group count: the emit form leaves the table empty)
    int64_t groups_hint = -1;
    bool rooms_ok = true;
    unsigned long long groups_before = 0;
    if (probe) {
      GroupbyHeader h0{};
      const int rc0 = read_header(state, &h0, st);
      if (rc0 != ARX_OK) return rc0;
      groups_before = h0.num_groups;
    }
    if (sketch) {
      // (round 4) the first probe_rows keys through a HyperLogLog sketch instead of through the two-level plan: the same
      // question — did most of these keys repeat, and how many are there —, nothing aggregated, one 64 KB read-back.  The
      // registers sit at the top of the caller's scratch, which no slice touches before the first plan is bound.
      int64_t distinct = 0;
      const int rc0 = gbp_estimate_distinct(k, probe_rows, reinterpret_cast<unsigned int*>(w), &distinct, st);
      if (rc0 != ARX_OK) return rc0;
      g_gbp_slices_probe.fetch_add(1, std::memory_order_relaxed);
      if (distinct * 2 < probe_rows) groups_hint = distinct + distinct / 16 + 1024;
    }
    for (int64_t r0 = 0; r0 < n;) {
      const bool probing = probe && r0 == 0;
      int64_t m = std::min(probing ? probe_rows : slice, n - r0);
      if (!probing && groups_hint >= 0) {   // the plan the estimate selects may take larger slices from the same scratch
        const int64_t hinted = gbp_slice_for(ws_bytes, n - r0, capacity, groups_hint);
        if (hinted >= kGbTile) m = std::min(hinted, n - r0);
      }
      GbpPlan plan = gbp_plan(m, capacity, groups_hint, 0, rooms_ok);
      GbpArgs a{};
      a.dense = 0;
      a.keys = k + r0;
      a.values = val + r0;
      a.kvalid = make_bits(kbm, keys_i32->offset + r0, m);
      a.vvalid = make_bits(vbm, values_i64->offset + r0, m);
      a.n = m;
      gbp_bind(a, plan, w);
      if (emit != nullptr) {
        a.emit_records = emit->records;
        a.emit_cursor = emit->cursor;
        a.emit_parts = emit->parts;
        a.emit_capacity = emit->capacity;
      }
      int rc = (kbm != nullptr || vbm != nullptr) ? gbp_run_slice<true>(v, a, plan, st)
                                                  : gbp_run_slice<false>(v, a, plan, st);
      if (rc == kGbpRoomsOverflow) {
        // a partition outgrew its room (hot keys): this slice again with counted partitions, and the rest of the call too
        rooms_ok = false;
        plan = gbp_plan(m, capacity, groups_hint, 0, false);
        gbp_bind(a, plan, w);
        rc = (kbm != nullptr || vbm != nullptr) ? gbp_run_slice<true>(v, a, plan, st, true)
                                                : gbp_run_slice<false>(v, a, plan, st, true);
      }
      if (rc != ARX_OK) return rc;
      if (probing) {
        g_gbp_slices_probe.fetch_add(1, std::memory_order_relaxed);
        GroupbyHeader h1{};
        const int rc1 = read_header(state, &h1, st);
        if (rc1 != ARX_OK) return rc1;
        const unsigned long long fresh = h1.num_groups - groups_before;
        if (fresh * 2 < static_cast<unsigned long long>(m)) {   // most keys of the probe repeated: the count is near G
          groups_hint = static_cast<int64_t>(h1.num_groups + h1.num_groups / 16 + 1024);
        }
      }
      r0 += m;
    }
    return ARX_OK;
  }

  if (emit != nullptr) {
    set_error("arx_groupby_sum_i64_consume_partials: a batch this small (or without scratch) goes through the table");
    return ARX_NOT_IMPLEMENTED;
  }
  // ---- direct path (small batches / no scratch): every row goes to the HBM table
  const Bits kb = make_bits(kbm, keys_i32->offset, n);
  const Bits vb = make_bits(vbm, values_i64->offset, n);
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

int arx_hash_bool_finalize(const int64_t* n_valid, const int64_t* n_null, const int64_t* n_true, int64_t num_groups, int is_all,
                           int skip_nulls, uint32_t min_count, void* out_values, void* out_validity, int64_t* valid_count,
                           void* stream) {
  if (num_groups < 0) {
    set_error("bad arguments to arx_hash_bool_finalize");
    return ARX_INVALID;
  }
  if (num_groups == 0) return ARX_OK;
  if (n_valid == nullptr || n_null == nullptr || n_true == nullptr || out_values == nullptr || out_validity == nullptr) {
    set_error("NULL buffer passed to arx_hash_bool_finalize");
    return ARX_INVALID;
  }
  const unsigned grid = static_cast<unsigned>(std::min<int64_t>(ceil_div(num_groups, 256), 1024));
  hipLaunchKernelGGL(hash_bool_finalize_kernel, dim3(grid), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const long long*>(n_valid), reinterpret_cast<const long long*>(n_null),
                     reinterpret_cast<const long long*>(n_true), num_groups, is_all, skip_nulls, min_count,
                     static_cast<uint64_t*>(out_values), static_cast<uint64_t*>(out_validity),
                     reinterpret_cast<unsigned long long*>(valid_count));
  ARX_CHECK_LAUNCH("hash_bool_finalize_kernel");
  return ARX_OK;
}

int arx_groupby_key_range_i32(const ArxSpan* keys, int32_t* out_min_max, void* stream) {
  if (keys == nullptr || out_min_max == nullptr || keys->length < 0 || (keys->length > 0 && keys->data == nullptr)) {
    set_error("bad arguments to arx_groupby_key_range_i32");
    return ARX_INVALID;
  }
  if (keys->length == 0) return ARX_OK;
  const int32_t* k = static_cast<const int32_t*>(keys->data) + keys->offset;
  const unsigned grid = static_cast<unsigned>(std::min<int64_t>(ceil_div(keys->length, int64_t(256) * 16), 256 * 8));
  hipLaunchKernelGGL(groupby_key_range_kernel, dim3(std::max(grid, 1u)), dim3(256), 0, as_stream(stream), k, keys->length,
                     out_min_max);
  ARX_CHECK_LAUNCH("groupby_key_range_kernel");
  return ARX_OK;
}

// ---- the range-partitioned state (groupby_lines.h)
int arx_groupby_range_plan(int64_t max_rows, int32_t key_min, int32_t key_max, ArxRangePlan* out) {
  if (out == nullptr || max_rows < 0 || key_min > key_max) {
    set_error("bad arguments to arx_groupby_range_plan");
    return ARX_INVALID;
  }
  int width = 0, wshift = 0, bins = 0;
  GblPlan plan{};
  if (!gbl_partitions(key_min, key_max, &width, &wshift, &bins) ||
      !gbl_plan(std::max<int64_t>(max_rows, 1), key_min, width, wshift, bins, false, &plan)) {
    set_error("arx_groupby_range_plan: keys in [%d, %d] need more than %d partitions of %d keys, or fewer than %d of 8 "
              "(or too many rows for one pass): use the table operator", key_min, key_max, kGblMaxBins, kGblMaxWidth, kGblMinBins);
    return ARX_NOT_IMPLEMENTED;
  }
  out->key_min = key_min;
  out->width = width;
  out->partitions = bins;
  out->reserved = 0;
  out->slots = static_cast<int64_t>(bins) * width;
  out->state_bytes = static_cast<uint64_t>(out->slots) * 16;
  out->workspace_bytes = plan.total + 256;
  return ARX_OK;
}

int arx_groupby_key_range_sampled_i32(const ArxSpan* keys, int64_t sample_rows, int32_t* out_min_max, void* stream) {
  if (keys == nullptr || out_min_max == nullptr || keys->length < 0 || (keys->length > 0 && keys->data == nullptr)) {
    set_error("bad arguments to arx_groupby_key_range_sampled_i32");
    return ARX_INVALID;
  }
  if (keys->length == 0) return ARX_OK;
  return gbl_sample_range(static_cast<const int32_t*>(keys->data) + keys->offset, keys->length, std::max<int64_t>(sample_rows, 64),
                          out_min_max, as_stream(stream));
}

int arx_groupby_range_sum_i64_consume(void* state, const ArxRangePlan* plan, const ArxSpan* keys_i32, const ArxSpan* values_i64,
                                      void* ws, size_t ws_bytes, void* stream) {
  if (state == nullptr || plan == nullptr || keys_i32 == nullptr || values_i64 == nullptr) {
    set_error("NULL argument to arx_groupby_range_sum_i64_consume");
    return ARX_INVALID;
  }
  if (keys_i32->length != values_i64->length) {
    set_error("Array arguments must all be the same length (keys %lld vs values %lld)", static_cast<long long>(keys_i32->length),
              static_cast<long long>(values_i64->length));
    return ARX_INVALID;
  }
  const int64_t n = keys_i32->length;
  if (n == 0) return ARX_OK;
  if ((keys_i32->null_count != 0 && keys_i32->validity != nullptr) || (values_i64->null_count != 0 && values_i64->validity != nullptr)) {
    set_error("arx_groupby_range_sum_i64_consume: rows with nulls go through the table operator (arx_groupby_sum_i64_consume)");
    return ARX_NOT_IMPLEMENTED;
  }
  const bool pow2 = plan->width >= 8 && plan->width <= 8192 && (plan->width & (plan->width - 1)) == 0;
  if ((!pow2 && plan->width != kGblMaxWidth) || plan->partitions < 1 || plan->partitions > kGblMaxBins ||
      plan->slots != static_cast<int64_t>(plan->partitions) * plan->width || (reinterpret_cast<uint64_t>(state) & 15) != 0) {
    set_error("arx_groupby_range_sum_i64_consume: not a plan of arx_groupby_range_plan (or a state that is not 16-byte aligned)");
    return ARX_INVALID;
  }
  int wshift = 0;
  if (pow2) {
    while ((1 << wshift) < plan->width) ++wshift;
  }
  uint8_t* w = reinterpret_cast<uint8_t*>((reinterpret_cast<uint64_t>(ws) + 255) & ~uint64_t(255));
  const size_t lost = ws == nullptr ? 0 : static_cast<size_t>(w - static_cast<uint8_t*>(ws));
  GblPlan gp{};
  if (ws == nullptr || ws_bytes < lost || !gbl_plan(n, plan->key_min, plan->width, wshift, plan->partitions, false, &gp) ||
      gp.total > ws_bytes - lost) {
    set_error("arx_groupby_range_sum_i64_consume: %lld rows need %llu bytes of scratch", static_cast<long long>(n),
              static_cast<unsigned long long>(gp.total + 256));
    return ARX_CAPACITY_ERROR;
  }
  hipStream_t st = as_stream(stream);
  GblArgs a{};
  a.keys = static_cast<const int32_t*>(keys_i32->data) + keys_i32->offset;
  a.values = static_cast<const int64_t*>(values_i64->data) + values_i64->offset;
  a.kvalid = make_bits(nullptr, 0, n);
  a.vvalid = make_bits(nullptr, 0, n);
  a.n = n;
  gbl_bind(a, gp, w);
  a.dense = static_cast<uint8_t*>(state);
  uint32_t flags[4] = {0, 0, 0, 0};
  const int prc = gbl_partition<false>(a, flags, st);
  if (prc != ARX_OK) return prc;
  if (flags[0] != 0 || flags[1] != 0 || flags[2] != 0) {
    g_gbl_fallbacks.fetch_add(1, std::memory_order_relaxed);
    set_error("arx_groupby_range_sum_i64_consume: %s — nothing consumed, use the table operator for these rows",
              flags[2] != 0 ? "keys outside the plan's range" : flags[1] != 0 ? "a hot key (rounds without end)" : "a partition outgrew its room");
    return ARX_CAPACITY_ERROR;
  }
  g_gbl_slices.fetch_add(1, std::memory_order_relaxed);
  return gbl_aggregate(a, st);
}

int arx_groupby_range_merge(void* state, const void* others, int32_t width, int64_t num_partitions, int num_others,
                            int64_t others_stride_bytes, void* stream) {
  if (state == nullptr || others == nullptr || width < 1 || num_partitions < 0 || num_others < 0 || (others_stride_bytes & 7) != 0) {
    set_error("bad arguments to arx_groupby_range_merge");
    return ARX_INVALID;
  }
  const int64_t words = num_partitions * width * 2;
  if (words == 0 || num_others == 0) return ARX_OK;
  hipLaunchKernelGGL(gbl_merge_kernel, dim3(gb_grid(words)), dim3(kBlock), 0, as_stream(stream), static_cast<unsigned long long*>(state),
                     static_cast<const unsigned long long*>(others), words, num_others, others_stride_bytes / 8);
  ARX_CHECK_LAUNCH("gbl_merge_kernel");
  return ARX_OK;
}

size_t arx_groupby_range_finalize_workspace_bytes(int64_t slots) {
  return static_cast<size_t>(ceil_div(std::max<int64_t>(slots, 1), kGblTile)) * 8 + 256;
}

int arx_groupby_range_finalize(const void* partitions, int32_t first_key, int32_t width, int64_t num_partitions, uint32_t min_count,
                               void* ws, size_t ws_bytes, int32_t* out_keys, int64_t* out_sums, int64_t* out_counts,
                               uint8_t* out_valid, int64_t* out_num_groups, void* stream) {
  if (width < 1 || num_partitions < 0 || out_num_groups == nullptr) {
    set_error("bad arguments to arx_groupby_range_finalize");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  const int64_t slots = num_partitions * width;
  if (slots == 0) {
    ARX_HIP(hipMemsetAsync(out_num_groups, 0, 8, st));
    return ARX_OK;
  }
  if (partitions == nullptr || out_keys == nullptr || out_sums == nullptr || out_valid == nullptr || ws == nullptr ||
      ws_bytes < arx_groupby_range_finalize_workspace_bytes(slots)) {
    set_error("arx_groupby_range_finalize: NULL buffer or too little scratch");
    return ARX_INVALID;
  }
  unsigned long long* tiles = reinterpret_cast<unsigned long long*>((reinterpret_cast<uint64_t>(ws) + 7) & ~uint64_t(7));
  const int64_t ntiles = ceil_div(slots, kGblTile);
  const uint8_t* d = static_cast<const uint8_t*>(partitions);
  hipLaunchKernelGGL(gbl_finalize_count_kernel, dim3(static_cast<unsigned>(ntiles)), dim3(kBlock), 0, st, d, width, slots, tiles);
  ARX_CHECK_LAUNCH("gbl_finalize_count_kernel");
  hipLaunchKernelGGL(gbl_finalize_scan_kernel, dim3(1), dim3(1024), 0, st, tiles, ntiles, out_num_groups);
  ARX_CHECK_LAUNCH("gbl_finalize_scan_kernel");
  hipLaunchKernelGGL(gbl_finalize_emit_kernel, dim3(static_cast<unsigned>(ntiles)), dim3(kBlock), 0, st, d, width, slots, first_key,
                     static_cast<unsigned long long>(min_count), tiles, out_keys, out_sums, out_counts, out_valid);
  ARX_CHECK_LAUNCH("gbl_finalize_emit_kernel");
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
  return arx_groupby_export(state, nullptr, out_keys, out_key_is_valid, out_sums, out_counts, out_no_nulls,
                            nullptr, nullptr, stream);
}

int arx_groupby_export(void* state, const void* minmax, int32_t* out_keys, uint8_t* out_key_is_valid,
                       int64_t* out_sums, int64_t* out_counts, uint8_t* out_no_nulls, int64_t* out_mins,
                       int64_t* out_maxs, void* stream) {
  if (state == nullptr) {
    set_error("state is NULL");
    return ARX_INVALID;
  }
  if ((out_mins != nullptr || out_maxs != nullptr) && (minmax == nullptr || out_mins == nullptr || out_maxs == nullptr)) {
    set_error("arx_groupby_export: out_mins/out_maxs need the minmax buffer and each other");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  int64_t cap = 0;
  const int rc = state_capacity(state, &cap, st);
  if (rc != ARX_OK) return rc;
  GroupbyView v = gb_view(state, cap);
  ARX_HIP(hipMemsetAsync(&v.hdr->export_cursor, 0, sizeof(unsigned long long), st));
  const unsigned egrid = static_cast<unsigned>(ceil_div(cap + 1, kExportSlotsPerBlock));
  const long long* mins = static_cast<const long long*>(minmax);
  const long long* maxs = mins == nullptr ? nullptr : mins + cap + 1;
  hipLaunchKernelGGL(groupby_export_kernel, dim3(egrid), dim3(kBlock), 0, st, v, out_keys,
                     out_key_is_valid, out_sums, out_counts, out_no_nulls, mins, maxs, out_mins, out_maxs);
  ARX_CHECK_LAUNCH("groupby_export_kernel");
  return ARX_OK;
}

int arx_groupby_lookup_i32(void* state, int64_t capacity, const ArxSpan* keys_i32, int32_t* out, void* stream) {
  if (state == nullptr || keys_i32 == nullptr || capacity < 1 || (capacity & (capacity - 1)) != 0) {
    set_error("bad arguments to arx_groupby_lookup_i32");
    return ARX_INVALID;
  }
  const int64_t n = keys_i32->length;
  if (n == 0) return ARX_OK;
  if (keys_i32->data == nullptr || out == nullptr) {
    set_error("NULL buffer passed to arx_groupby_lookup_i32");
    return ARX_INVALID;
  }
  GroupbyView v = gb_view(state, capacity);
  const Bits kb = make_bits(keys_i32->null_count != 0 ? keys_i32->validity : nullptr, keys_i32->offset, n);
  hipLaunchKernelGGL(groupby_lookup_kernel, dim3(gb_grid(n)), dim3(kBlock), 0, as_stream(stream), v,
                     static_cast<const int32_t*>(keys_i32->data) + keys_i32->offset, kb, n, out);
  ARX_CHECK_LAUNCH("groupby_lookup_kernel");
  return ARX_OK;
}

size_t arx_groupby_minmax_bytes(int64_t capacity) {
  if (capacity < 1) capacity = 1;
  return (static_cast<size_t>(capacity + 1) * 16 + 255) & ~size_t(255);
}

int arx_groupby_minmax_init(void* minmax, int64_t capacity, void* stream) {
  if (minmax == nullptr || capacity < 1) {
    set_error("bad arguments to arx_groupby_minmax_init");
    return ARX_INVALID;
  }
  long long* mins = static_cast<long long*>(minmax);
  hipLaunchKernelGGL(groupby_minmax_init_kernel, dim3(gb_grid(capacity + 1)), dim3(kBlock), 0, as_stream(stream),
                     mins, mins + capacity + 1, capacity + 1);
  ARX_CHECK_LAUNCH("groupby_minmax_init_kernel");
  return ARX_OK;
}

int arx_groupby_minmax_i64_consume(void* state, void* minmax, int64_t capacity, const ArxSpan* keys_i32,
                                   const ArxSpan* values_i64, void* stream) {
  if (state == nullptr || minmax == nullptr || keys_i32 == nullptr || values_i64 == nullptr) {
    set_error("NULL argument to arx_groupby_minmax_i64_consume");
    return ARX_INVALID;
  }
  if (capacity < 1 || (capacity & (capacity - 1)) != 0) {
    set_error("group-by capacity must be a power of two");
    return ARX_INVALID;
  }
  if (keys_i32->length != values_i64->length) {
    set_error("Array arguments must all be the same length (keys %lld vs values %lld)",
              static_cast<long long>(keys_i32->length), static_cast<long long>(values_i64->length));
    return ARX_INVALID;
  }
  const int64_t n = keys_i32->length;
  if (n == 0) return ARX_OK;
  if (keys_i32->data == nullptr || values_i64->data == nullptr) {
    set_error("NULL data buffer passed to arx_groupby_minmax_i64_consume");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  GroupbyView v = gb_view(state, capacity);
  long long* mins = static_cast<long long*>(minmax);
  const Bits kb = make_bits(keys_i32->null_count != 0 ? keys_i32->validity : nullptr, keys_i32->offset, n);
  const Bits vb = make_bits(values_i64->null_count != 0 ? values_i64->validity : nullptr, values_i64->offset, n);
  hipLaunchKernelGGL(groupby_minmax_consume_kernel, dim3(gb_grid(n)), dim3(kBlock), 0, st, v, mins,
                     mins + capacity + 1, static_cast<const int32_t*>(keys_i32->data) + keys_i32->offset, kb,
                     static_cast<const int64_t*>(values_i64->data) + values_i64->offset, vb, n);
  ARX_CHECK_LAUNCH("groupby_minmax_consume_kernel");
  GroupbyHeader h;
  const int rc = read_header(state, &h, st);
  if (rc != ARX_OK) return rc;
  if (h.overflow != 0) {
    set_error("group-by table overflow: more distinct keys than capacity %lld", static_cast<long long>(capacity));
    return ARX_INVALID;
  }
  return ARX_OK;
}

int arx_groupby_minmax_merge(void* state, void* minmax, int64_t capacity, const int32_t* keys,
                             const uint8_t* key_is_valid, const int64_t* mins, const int64_t* maxs,
                             const uint8_t* no_nulls, int64_t num_groups, void* stream) {
  if (state == nullptr || minmax == nullptr || num_groups < 0) {
    set_error("bad arguments to arx_groupby_minmax_merge");
    return ARX_INVALID;
  }
  if (num_groups == 0) return ARX_OK;
  if (keys == nullptr || mins == nullptr || maxs == nullptr) {
    set_error("NULL partial-aggregate column");
    return ARX_INVALID;
  }
  GroupbyView v = gb_view(state, capacity);
  long long* m = static_cast<long long*>(minmax);
  hipLaunchKernelGGL(groupby_minmax_merge_kernel, dim3(gb_grid(num_groups)), dim3(kBlock), 0, as_stream(stream),
                     v, m, m + capacity + 1, keys, key_is_valid, mins, maxs, no_nulls, num_groups);
  ARX_CHECK_LAUNCH("groupby_minmax_merge_kernel");
  return ARX_OK;
}

int arx_groupby_minmax_finalize(const int64_t* mins, const int64_t* maxs, const uint8_t* no_nulls,
                                int64_t num_groups, int skip_nulls, uint8_t* out_valid, void* stream) {
  if (num_groups < 0) {
    set_error("negative num_groups");
    return ARX_INVALID;
  }
  if (num_groups == 0) return ARX_OK;
  if (mins == nullptr || maxs == nullptr || out_valid == nullptr || (!skip_nulls && no_nulls == nullptr)) {
    set_error("NULL argument to arx_groupby_minmax_finalize");
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(groupby_minmax_finalize_kernel, dim3(gb_grid(num_groups)), dim3(kBlock), 0, as_stream(stream),
                     mins, maxs, no_nulls, num_groups, skip_nulls, out_valid);
  ARX_CHECK_LAUNCH("groupby_minmax_finalize_kernel");
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

int arx_hash_minmax_i64_fill(int64_t* mins, int64_t* maxs, int64_t first_group, int64_t num_new_groups, void* stream) {
  if (num_new_groups < 0 || first_group < 0 || (num_new_groups > 0 && (mins == nullptr || maxs == nullptr))) {
    set_error("bad arguments to arx_hash_minmax_i64_fill");
    return ARX_INVALID;
  }
  if (num_new_groups == 0) return ARX_OK;
  hipLaunchKernelGGL(hash_minmax_dense_fill_kernel, dim3(gb_grid(num_new_groups)), dim3(kBlock), 0, as_stream(stream),
                     reinterpret_cast<long long*>(mins), reinterpret_cast<long long*>(maxs), first_group, num_new_groups);
  ARX_CHECK_LAUNCH("hash_minmax_dense_fill_kernel");
  return ARX_OK;
}

// values / scalar conventions of arx_hash_sum_i64_consume
static int dense_values(const char* what, const ArxSpan* values, int values_is_scalar, const uint32_t* group_ids,
                        int64_t length, const int64_t** v, Bits* vb) {
  if (values == nullptr || length < 0) {
    set_error("bad arguments to %s", what);
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  if (group_ids == nullptr || (!values_is_scalar && values->data == nullptr)) {
    set_error("NULL buffer passed to %s", what);
    return ARX_INVALID;
  }
  if (!values_is_scalar && values->length != length) {
    set_error("Array arguments must all be the same length (values %lld vs group ids %lld)",
              static_cast<long long>(values->length), static_cast<long long>(length));
    return ARX_INVALID;
  }
  *v = values_is_scalar ? nullptr : static_cast<const int64_t*>(values->data) + values->offset;
  *vb = make_bits(values->null_count != 0 ? values->validity : nullptr, values->offset, length);
  if (values_is_scalar && values->null_count != 0) {   // a null scalar: every row reads as null
    vb->base = nullptr;
    vb->length = 0;
  }
  return ARX_OK;
}

int arx_hash_minmax_i64_consume(const ArxSpan* values, int values_is_scalar, int64_t scalar_value, const uint32_t* group_ids,
                                int64_t length, int64_t* mins, int64_t* maxs, uint32_t* null_seen, void* stream) {
  const int64_t* v = nullptr;
  Bits vb{};
  const int rc = dense_values("arx_hash_minmax_i64_consume", values, values_is_scalar, group_ids, length, &v, &vb);
  if (rc != ARX_OK || length == 0) return rc;
  if (mins == nullptr || maxs == nullptr || null_seen == nullptr) {
    set_error("NULL state passed to arx_hash_minmax_i64_consume");
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(hash_minmax_dense_consume_kernel, dim3(gb_grid(length)), dim3(kBlock), 0, as_stream(stream), v,
                     scalar_value, values_is_scalar, vb, group_ids, length, reinterpret_cast<long long*>(mins),
                     reinterpret_cast<long long*>(maxs), null_seen);
  ARX_CHECK_LAUNCH("hash_minmax_dense_consume_kernel");
  return ARX_OK;
}

int arx_hash_minmax_i64_merge(int64_t* mins, int64_t* maxs, uint32_t* null_seen, const int64_t* other_mins,
                              const int64_t* other_maxs, const uint32_t* other_null_seen, const uint32_t* group_id_mapping,
                              int64_t other_num_groups, void* stream) {
  if (other_num_groups < 0) {
    set_error("bad arguments to arx_hash_minmax_i64_merge");
    return ARX_INVALID;
  }
  if (other_num_groups == 0) return ARX_OK;
  if (mins == nullptr || maxs == nullptr || null_seen == nullptr || other_mins == nullptr || other_maxs == nullptr ||
      other_null_seen == nullptr || group_id_mapping == nullptr) {
    set_error("NULL buffer passed to arx_hash_minmax_i64_merge");
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(hash_minmax_dense_merge_kernel, dim3(gb_grid(other_num_groups)), dim3(kBlock), 0, as_stream(stream),
                     other_mins, other_maxs, other_null_seen, group_id_mapping, other_num_groups,
                     reinterpret_cast<long long*>(mins), reinterpret_cast<long long*>(maxs), null_seen);
  ARX_CHECK_LAUNCH("hash_minmax_dense_merge_kernel");
  return ARX_OK;
}

int arx_hash_minmax_i64_finalize(const int64_t* mins, const int64_t* maxs, const uint32_t* null_seen, int64_t num_groups,
                                 int skip_nulls, void* out_validity, int64_t* valid_count, void* stream) {
  if (num_groups < 0) {
    set_error("bad arguments to arx_hash_minmax_i64_finalize");
    return ARX_INVALID;
  }
  if (num_groups == 0) return ARX_OK;
  if (mins == nullptr || maxs == nullptr || null_seen == nullptr || out_validity == nullptr) {
    set_error("NULL buffer passed to arx_hash_minmax_i64_finalize");
    return ARX_INVALID;
  }
  const int64_t nwords = ceil_div(num_groups, 64);
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(nwords, kWavesPerBlock), 256 * 8)));
  hipLaunchKernelGGL(hash_minmax_dense_finalize_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), mins, maxs, null_seen,
                     num_groups, skip_nulls, static_cast<uint64_t*>(out_validity),
                     reinterpret_cast<unsigned long long*>(valid_count));
  ARX_CHECK_LAUNCH("hash_minmax_dense_finalize_kernel");
  return ARX_OK;
}

#ifdef ARX_GBP_PROFILE
// the phase timestamps of the flat level's sampled workgroups, 64 x 8 x u64, 100 MHz
int arx_debug_gbp_profile(void* dst_host) {
  void* sym = nullptr;
  ARX_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_gbp_prof)));
  ARX_HIP(hipMemcpy(dst_host, sym, sizeof(unsigned long long) * 64 * 8, hipMemcpyDeviceToHost));
  return ARX_OK;
}
#endif

int arx_hash_minmax_float_consume(const ArxSpan* values, int num_type, int values_is_scalar, double scalar_value,
                                  const uint32_t* group_ids, int64_t length, int64_t* mins, int64_t* maxs, uint32_t* null_seen,
                                  void* stream) {
  if (num_type != ARX_NUM_FLOAT32 && num_type != ARX_NUM_FLOAT64) {
    set_error("arx_hash_minmax_float_consume: num_type %d is not float32 / float64", num_type);
    return ARX_INVALID;
  }
  const int64_t* v = nullptr;   // (dense_values offsets 8-byte elements: redone below for the element type)
  Bits vb{};
  const int rc = dense_values("arx_hash_minmax_float_consume", values, values_is_scalar, group_ids, length, &v, &vb);
  if (rc != ARX_OK || length == 0) return rc;
  if (mins == nullptr || maxs == nullptr || null_seen == nullptr) {
    set_error("NULL state passed to arx_hash_minmax_float_consume");
    return ARX_INVALID;
  }
  if (num_type == ARX_NUM_FLOAT64) {
    const double* d = values_is_scalar ? nullptr : static_cast<const double*>(values->data) + values->offset;
    hipLaunchKernelGGL(hash_minmax_dense_consume_float_kernel<double>, dim3(gb_grid(length)), dim3(kBlock), 0, as_stream(stream),
                       d, scalar_value, values_is_scalar, vb, group_ids, length, reinterpret_cast<long long*>(mins),
                       reinterpret_cast<long long*>(maxs), null_seen);
  } else {
    const float* f = values_is_scalar ? nullptr : static_cast<const float*>(values->data) + values->offset;
    hipLaunchKernelGGL(hash_minmax_dense_consume_float_kernel<float>, dim3(gb_grid(length)), dim3(kBlock), 0, as_stream(stream),
                       f, scalar_value, values_is_scalar, vb, group_ids, length, reinterpret_cast<long long*>(mins),
                       reinterpret_cast<long long*>(maxs), null_seen);
  }
  ARX_CHECK_LAUNCH("hash_minmax_dense_consume_float_kernel");
  return ARX_OK;
}

int arx_hash_minmax_float_finalize(const int64_t* mins, const int64_t* maxs, const uint32_t* null_seen, int64_t num_groups,
                                   int skip_nulls, int num_type, void* out_mins, void* out_maxs, void* out_validity,
                                   int64_t* valid_count, void* stream) {
  if (num_groups < 0 || (num_type != ARX_NUM_FLOAT32 && num_type != ARX_NUM_FLOAT64)) {
    set_error("bad arguments to arx_hash_minmax_float_finalize");
    return ARX_INVALID;
  }
  if (num_groups == 0) return ARX_OK;
  if (mins == nullptr || maxs == nullptr || null_seen == nullptr || out_validity == nullptr) {
    set_error("NULL buffer passed to arx_hash_minmax_float_finalize");
    return ARX_INVALID;
  }
  const int64_t nwords = ceil_div(num_groups, 64);
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(nwords, kWavesPerBlock), 256 * 8)));
  if (num_type == ARX_NUM_FLOAT64) {
    hipLaunchKernelGGL(hash_minmax_dense_finalize_float_kernel<double>, dim3(grid), dim3(kBlock), 0, as_stream(stream), mins, maxs,
                       null_seen, num_groups, skip_nulls, static_cast<double*>(out_mins), static_cast<double*>(out_maxs),
                       static_cast<uint64_t*>(out_validity), reinterpret_cast<unsigned long long*>(valid_count));
  } else {
    hipLaunchKernelGGL(hash_minmax_dense_finalize_float_kernel<float>, dim3(grid), dim3(kBlock), 0, as_stream(stream), mins, maxs,
                       null_seen, num_groups, skip_nulls, static_cast<float*>(out_mins), static_cast<float*>(out_maxs),
                       static_cast<uint64_t*>(out_validity), reinterpret_cast<unsigned long long*>(valid_count));
  }
  ARX_CHECK_LAUNCH("hash_minmax_dense_finalize_float_kernel");
  return ARX_OK;
}

int arx_hash_count_consume(const void* values_validity, int64_t values_offset, int64_t values_null_count, int mode,
                           const uint32_t* group_ids, int64_t length, int64_t* counts, void* stream) {
  if (length < 0 || mode < 0 || mode > 2) {
    set_error("bad arguments to arx_hash_count_consume");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  if (group_ids == nullptr || counts == nullptr) {
    set_error("NULL buffer passed to arx_hash_count_consume");
    return ARX_INVALID;
  }
  // values_null_count != 0 with a NULL bitmap = a null broadcast scalar: every row is null
  Bits vb = make_bits(values_null_count != 0 ? values_validity : nullptr, values_offset, length);
  if (values_null_count != 0 && values_validity == nullptr) {
    vb.base = nullptr;
    vb.length = 0;
  }
  hipLaunchKernelGGL(hash_count_dense_consume_kernel, dim3(gb_grid(length)), dim3(kBlock), 0, as_stream(stream), vb, mode,
                     group_ids, length, reinterpret_cast<unsigned long long*>(counts));
  ARX_CHECK_LAUNCH("hash_count_dense_consume_kernel");
  return ARX_OK;
}

int arx_hash_count_merge(int64_t* counts, const int64_t* other_counts, const uint32_t* group_id_mapping,
                         int64_t other_num_groups, void* stream) {
  if (other_num_groups < 0) {
    set_error("bad arguments to arx_hash_count_merge");
    return ARX_INVALID;
  }
  if (other_num_groups == 0) return ARX_OK;
  if (counts == nullptr || other_counts == nullptr || group_id_mapping == nullptr) {
    set_error("NULL buffer passed to arx_hash_count_merge");
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(hash_count_dense_merge_kernel, dim3(gb_grid(other_num_groups)), dim3(kBlock), 0, as_stream(stream),
                     other_counts, group_id_mapping, other_num_groups, reinterpret_cast<unsigned long long*>(counts));
  ARX_CHECK_LAUNCH("hash_count_dense_merge_kernel");
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

int arx_hash_mean_i64_finalize(const int64_t* sums, const int64_t* counts, int64_t num_groups, uint64_t abs_bound,
                               double* out_means, uint32_t* inexact, void* stream) {
  if (num_groups < 0) {
    set_error("negative num_groups");
    return ARX_INVALID;
  }
  if (num_groups == 0) return ARX_OK;
  if (sums == nullptr || counts == nullptr || out_means == nullptr || inexact == nullptr) {
    set_error("NULL buffer passed to arx_hash_mean_i64_finalize");
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(hash_mean_dense_finalize_kernel, dim3(gb_grid(num_groups)), dim3(kBlock), 0, as_stream(stream), sums,
                     counts, num_groups, abs_bound, out_means, inexact);
  ARX_CHECK_LAUNCH("hash_mean_dense_finalize_kernel");
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
    hipLaunchKernelGGL(partition_count_kernel, dim3(static_cast<unsigned>(ceil_div(num_groups, kPartChunk))),
                       dim3(kBlock), 0, st, keys, key_is_valid, num_groups, num_parts, part_counts);
    ARX_CHECK_LAUNCH("partition_count_kernel");
  }
  hipLaunchKernelGGL(partition_offsets_kernel, dim3(1), dim3(64), 0, st, part_counts, num_parts,
                     cursors, out_part_counts);
  ARX_CHECK_LAUNCH("partition_offsets_kernel");
  if (num_groups > 0) {
    hipLaunchKernelGGL(partition_scatter_kernel, dim3(static_cast<unsigned>(ceil_div(num_groups, kPartChunk))),
                       dim3(kBlock), 0, st, keys, key_is_valid, sums, counts, no_nulls, num_groups, num_parts, cursors, out_keys,
                       out_key_is_valid, out_sums, out_counts, out_no_nulls);
    ARX_CHECK_LAUNCH("partition_scatter_kernel");
  }
  return ARX_OK;
}

int arx_groupby_partition_rows(const ArxSpan* keys_i32, const ArxSpan* values_i64, int num_parts, void* ws, size_t ws_bytes,
                               ArxRowRecord* out_records, int64_t* out_part_counts, void* stream) {
  if (keys_i32 == nullptr || values_i64 == nullptr || num_parts < 1 || num_parts > kPartMaxParts || ws == nullptr ||
      ws_bytes < arx_groupby_partition_workspace_bytes(num_parts) || out_part_counts == nullptr ||
      keys_i32->length != values_i64->length) {
    set_error("bad arguments to arx_groupby_partition_rows");
    return ARX_INVALID;
  }
  const int64_t n = keys_i32->length;
  hipStream_t st = as_stream(stream);
  unsigned long long* part_counts = static_cast<unsigned long long*>(ws);
  unsigned long long* cursors = part_counts + num_parts;
  ARX_HIP(hipMemsetAsync(ws, 0, static_cast<size_t>(num_parts) * 16, st));
  const int32_t* k = static_cast<const int32_t*>(keys_i32->data) + keys_i32->offset;
  const int64_t* v = static_cast<const int64_t*>(values_i64->data) + values_i64->offset;
  const Bits kb = make_bits(keys_i32->null_count != 0 ? keys_i32->validity : nullptr, keys_i32->offset, n);
  const Bits vb = make_bits(values_i64->null_count != 0 ? values_i64->validity : nullptr, values_i64->offset, n);
  ArxRowRecordDev* out = reinterpret_cast<ArxRowRecordDev*>(out_records);
  const unsigned grid = static_cast<unsigned>(ceil_div(std::max<int64_t>(n, 1), kPartChunk));
  if (n > 0) {
    if (out == nullptr || k == nullptr || v == nullptr) {
      set_error("NULL buffer passed to arx_groupby_partition_rows");
      return ARX_INVALID;
    }
    hipLaunchKernelGGL((partition_rows_kernel<false>), dim3(grid), dim3(kBlock), 0, st, k, kb, v, vb, n, num_parts, part_counts,
                       cursors, out);
    ARX_CHECK_LAUNCH("partition_rows_kernel<count>");
  }
  hipLaunchKernelGGL(partition_offsets_kernel, dim3(1), dim3(64), 0, st, part_counts, num_parts, cursors, out_part_counts);
  ARX_CHECK_LAUNCH("partition_offsets_kernel");
  if (n > 0) {
    hipLaunchKernelGGL((partition_rows_kernel<true>), dim3(grid), dim3(kBlock), 0, st, k, kb, v, vb, n, num_parts, part_counts,
                       cursors, out);
    ARX_CHECK_LAUNCH("partition_rows_kernel<scatter>");
  }
  return ARX_OK;
}

int arx_groupby_unpack_rows(const ArxRowRecord* records, int64_t num_records, int32_t* out_keys, int64_t* out_values,
                            void* out_key_validity, void* out_value_validity, void* stream) {
  if (num_records < 0 || (num_records > 0 && (records == nullptr || out_keys == nullptr || out_values == nullptr ||
                                              out_key_validity == nullptr || out_value_validity == nullptr))) {
    set_error("bad arguments to arx_groupby_unpack_rows");
    return ARX_INVALID;
  }
  if (num_records == 0) return ARX_OK;
  const int64_t nwords = ceil_div(num_records, 64);
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(nwords, kWavesPerBlock), 256 * 32)));
  hipLaunchKernelGGL(unpack_rows_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream),
                     reinterpret_cast<const ArxRowRecordDev*>(records), num_records, out_keys, out_values,
                     static_cast<uint64_t*>(out_key_validity), static_cast<uint64_t*>(out_value_validity));
  ARX_CHECK_LAUNCH("unpack_rows_kernel");
  return ARX_OK;
}

int arx_groupby_export_partitioned(void* state, int num_parts, void* ws, size_t ws_bytes,
                                   ArxGroupPartial* out_records, int64_t* out_part_counts, void* stream) {
  if (state == nullptr || num_parts < 1 || num_parts > kPartMaxParts || ws == nullptr ||
      ws_bytes < arx_groupby_partition_workspace_bytes(num_parts) || out_part_counts == nullptr ||
      out_records == nullptr) {
    set_error("bad arguments to arx_groupby_export_partitioned");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  GroupbyHeader h;
  const int rc = read_header(state, &h, st);
  if (rc != ARX_OK) return rc;
  if (h.overflow) {
    set_error("group-by table overflowed (capacity %lld)", static_cast<long long>(h.capacity));
    return ARX_INVALID;
  }
  const GroupbyView v = gb_view(state, h.capacity);
  unsigned long long* part_counts = static_cast<unsigned long long*>(ws);
  unsigned long long* cursors = part_counts + num_parts;
  ARX_HIP(hipMemsetAsync(ws, 0, static_cast<size_t>(num_parts) * 16, st));
  const unsigned grid = static_cast<unsigned>(ceil_div(h.capacity + 1, kPartChunk));
  hipLaunchKernelGGL(export_part_count_kernel, dim3(grid), dim3(kBlock), 0, st, v, num_parts, part_counts);
  ARX_CHECK_LAUNCH("export_part_count_kernel");
  hipLaunchKernelGGL(partition_offsets_kernel, dim3(1), dim3(64), 0, st, part_counts, num_parts, cursors,
                     out_part_counts);
  ARX_CHECK_LAUNCH("partition_offsets_kernel");
  hipLaunchKernelGGL(export_part_scatter_kernel, dim3(grid), dim3(kBlock), 0, st, v, num_parts, cursors, out_records);
  ARX_CHECK_LAUNCH("export_part_scatter_kernel");
  return ARX_OK;
}

int arx_groupby_sum_i64_merge_records(void* state, int64_t capacity, const ArxGroupPartial* records,
                                      int64_t num_records, void* stream) {
  if (state == nullptr || num_records < 0 || (num_records > 0 && records == nullptr)) {
    set_error("bad arguments to arx_groupby_sum_i64_merge_records");
    return ARX_INVALID;
  }
  if (num_records == 0) return ARX_OK;
  const GroupbyView v = gb_view(state, capacity);
  hipLaunchKernelGGL(groupby_merge_records_kernel, dim3(gb_grid(num_records)), dim3(kBlock), 0, as_stream(stream),
                     v, records, num_records);
  ARX_CHECK_LAUNCH("groupby_merge_records_kernel");
  return ARX_OK;
}

int arx_groupby_mean_i64_finalize(const int64_t* sums, const int64_t* counts, const int64_t* mins, const int64_t* maxs,
                                  const uint8_t* no_nulls, int64_t num_groups, int skip_nulls, uint32_t min_count,
                                  double* out_means, uint8_t* out_valid, uint32_t* out_inexact, void* stream) {
  if (num_groups < 0 || out_inexact == nullptr) {
    set_error("bad arguments to arx_groupby_mean_i64_finalize");
    return ARX_INVALID;
  }
  if (num_groups == 0) return ARX_OK;
  if (sums == nullptr || counts == nullptr || mins == nullptr || maxs == nullptr || out_means == nullptr ||
      out_valid == nullptr || (!skip_nulls && no_nulls == nullptr)) {
    set_error("NULL buffer passed to arx_groupby_mean_i64_finalize");
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(groupby_mean_finalize_kernel, dim3(gb_grid(num_groups)), dim3(kBlock), 0, as_stream(stream), sums,
                     counts, mins, maxs, no_nulls, num_groups, skip_nulls, min_count, out_means, out_valid, out_inexact);
  ARX_CHECK_LAUNCH("groupby_mean_finalize_kernel");
  return ARX_OK;
}

// ---- hash_sum(int64, uint32 group id) with scratch: the radix-partitioned LDS aggregation keyed on the group id
static int dense_id_bits(int64_t num_groups) {
  int b = 1;
  while (b < 32 && (int64_t(1) << b) < num_groups) ++b;
  return b;
}

size_t arx_hash_sum_consume_workspace_bytes(int64_t length, int64_t num_groups) {
  if (length <= 0 || num_groups <= 0) return 0;
  const int64_t cap = int64_t(2) << dense_id_bits(num_groups);
  return gbp_plan(std::min<int64_t>(length, int64_t(g_gbp_max_slice)), cap).total;
}

int arx_hash_sum_i64_consume_ws(const ArxSpan* values, int values_is_scalar, int64_t scalar_value,
                                const uint32_t* group_ids, int64_t length, int64_t num_groups, int64_t* sums,
                                int64_t* counts, uint32_t* null_seen, void* ws, size_t ws_bytes, void* stream) {
  // broadcast scalars, small batches and calls without (enough) scratch keep the per-row device atomics
  const bool partitioned = !values_is_scalar && values != nullptr && length >= g_gbp_min_rows && num_groups > 0 &&
                           num_groups <= (int64_t(1) << 31) && ws != nullptr &&
                           (reinterpret_cast<uint64_t>(ws) & 255) == 0;
  const int64_t cap = partitioned ? (int64_t(2) << dense_id_bits(num_groups)) : 0;
  const int64_t slice = partitioned ? gbp_slice_for(ws_bytes, length, cap) : 0;
  if (!partitioned || slice < kGbTile || slice < std::min<int64_t>(length, 1 << 16)) {
    return arx_hash_sum_i64_consume(values, values_is_scalar, scalar_value, group_ids, length, sums, counts, null_seen,
                                    stream);
  }
  if (group_ids == nullptr || sums == nullptr || counts == nullptr || null_seen == nullptr || values->data == nullptr) {
    set_error("NULL buffer passed to arx_hash_sum_i64_consume_ws");
    return ARX_INVALID;
  }
  if (values->length != length) {
    set_error("Array arguments must all be the same length (values %lld vs group ids %lld)",
              static_cast<long long>(values->length), static_cast<long long>(length));
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  const int64_t* val = static_cast<const int64_t*>(values->data) + values->offset;
  const void* vbm = values->null_count != 0 ? values->validity : nullptr;
  const GroupbyView none{};   // the keyed table is not touched in the dense form
  uint8_t* w = static_cast<uint8_t*>(ws);
  for (int64_t r0 = 0; r0 < length; r0 += slice) {
    const int64_t m = std::min(slice, length - r0);
    const GbpPlan plan = gbp_plan(m, cap, num_groups, dense_id_bits(num_groups));
    GbpArgs a{};
    a.dense = 1;
    a.dense_shl = 32 - dense_id_bits(num_groups);
    a.dense_sums = reinterpret_cast<unsigned long long*>(sums);
    a.dense_counts = reinterpret_cast<unsigned long long*>(counts);
    a.dense_null_seen = null_seen;
    a.keys = reinterpret_cast<const int32_t*>(group_ids) + r0;
    a.values = val + r0;
    a.kvalid = make_bits(nullptr, 0, m);
    a.vvalid = make_bits(vbm, values->offset + r0, m);
    a.n = m;
    gbp_bind(a, plan, w);
    const int rc = vbm != nullptr ? gbp_run_slice<true>(none, a, plan, st) : gbp_run_slice<false>(none, a, plan, st);
    if (rc != ARX_OK) return rc;
  }
  return ARX_OK;
}

}  // extern "C"
