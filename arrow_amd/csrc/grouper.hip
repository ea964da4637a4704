// Grouper for key rows of up to 16 bytes (one or several fixed-width key columns): what arrow::compute::Grouper does
// (cpp/src/arrow/compute/row/grouper.h:104-137; GrouperFastImpl::ConsumeImpl, row/grouper.cc:695-815 — encode the key
// columns of a batch as rows, map every row through a hash table, hand out dense group ids in order of first
// appearance, keep the unique rows for GetUniques :835-880), for batches that live in HBM.
//
// The table is open addressing over 32-byte slots {meta, key word 0, key word 1}.  A row claims an empty slot with one
// 64-bit CAS on `meta`, writes its key, then publishes {state, null mask, group id} with a second store; rows that run
// into a claimed-but-unpublished slot are not allowed to wait for it (lanes of one wave cannot wait for each other) —
// they go to a pending list that the next launch drains, by which time every publication of the previous launch is
// visible.  Rows of groups that already exist only read.
//
// Group ids are in order of first appearance (GrouperImpl's order): the k-th distinct key row in ROW ORDER gets id k.  Concurrent insertion hands out
// provisional ids in arrival order; every new group records its smallest row (atomicMin), the set of those rows is a
// bitmap whose ascending positions (the filter machinery's bit -> row-number compaction) are the ranks, and the new
// groups and the batch's ids are renumbered once.  Batches that add no group skip all of that.
#include "arx_common.h"

#include <algorithm>

namespace arx {

size_t selection_workspace_bytes(int64_t length);
int selection_bit_positions(const void* bitmap, int64_t bit_offset, int64_t length, bool invert, void* ws,
                            size_t ws_bytes, uint32_t* out, int64_t* out_count_host, hipStream_t st);

constexpr int kGrouperMaxKeys = 8;
constexpr unsigned long long kSlotEmpty = 0, kSlotClaimed = 1, kSlotPublished = 3;

struct GrouperHeader {
  int64_t max_groups;
  int64_t slots;
  unsigned long long num_groups;
  unsigned int overflow;    // more than max_groups distinct key rows
  unsigned int pending;     // rows deferred by the current launch
  int64_t pad[4];
};
static_assert(sizeof(GrouperHeader) == 64, "one line");

struct __attribute__((aligned(32))) GrouperSlot {
  unsigned long long meta;  // bits 0-1 state, 8-15 null mask, 32-63 group id
  unsigned long long k0, k1;
  unsigned long long unused;
};

struct GrouperView {
  GrouperHeader* hdr;
  GrouperSlot* slots;
  unsigned long long* uniq_k0;   // [max_groups]
  unsigned long long* uniq_k1;   // [max_groups]
  unsigned int* uniq_mask;       // [max_groups]
  unsigned int* slot_of;         // [max_groups] slot of every group
  int64_t nslots;
  int lg;
};

static inline int64_t grouper_slots_for(int64_t max_groups) {
  int64_t s = 1024;
  while (s < 2 * max_groups) s <<= 1;
  return s;
}

static inline GrouperView grouper_view(void* state, int64_t max_groups) {
  GrouperView v;
  uint8_t* p = static_cast<uint8_t*>(state);
  v.hdr = reinterpret_cast<GrouperHeader*>(p);
  v.nslots = grouper_slots_for(max_groups);
  v.slots = reinterpret_cast<GrouperSlot*>(p + 64);
  v.uniq_k0 = reinterpret_cast<unsigned long long*>(p + 64 + v.nslots * sizeof(GrouperSlot));
  v.uniq_k1 = v.uniq_k0 + max_groups;
  v.uniq_mask = reinterpret_cast<unsigned int*>(v.uniq_k1 + max_groups);
  v.slot_of = v.uniq_mask + max_groups;
  int lg = 0;
  while ((int64_t(1) << lg) < v.nslots) ++lg;
  v.lg = lg;
  return v;
}

struct GrouperCols {
  const uint8_t* data[kGrouperMaxKeys];
  Bits valid[kGrouperMaxKeys];
  int64_t offset[kGrouperMaxKeys];
  int width[kGrouperMaxKeys];      // 1, 2, 4, 8
  int byte_pos[kGrouperMaxKeys];   // position of the column inside the 16-byte key row
  int num_keys;
};

struct GrouperKey {
  unsigned long long k0, k1;
  unsigned int mask;
};

// the key row of `row`: columns little-endian back to back, a null column contributes zeros and its bit in mask
__device__ __forceinline__ GrouperKey grouper_load_key(const GrouperCols& c, int64_t row) {
  GrouperKey k{0, 0, 0};
  for (int j = 0; j < c.num_keys; ++j) {
    bool valid = true;
    if (c.valid[j].base != nullptr) {
      const uint64_t w = load_word(c.valid[j], row >> 6);
      valid = ((w >> (row & 63)) & 1) != 0;
    }
    if (!valid) {
      k.mask |= 1u << j;
      continue;
    }
    const uint8_t* p = c.data[j] + (c.offset[j] + row) * c.width[j];
    unsigned long long v;
    switch (c.width[j]) {
      case 1: v = *p; break;
      case 2: v = *reinterpret_cast<const uint16_t*>(p); break;
      case 4: v = *reinterpret_cast<const uint32_t*>(p); break;
      default: v = *reinterpret_cast<const unsigned long long*>(p); break;
    }
    const int pos = c.byte_pos[j];
    if (pos < 8) {
      k.k0 |= v << (8 * pos);
      if (pos + c.width[j] > 8) k.k1 |= v >> (8 * (8 - pos));   // a column that straddles the two key words (pos > 0 here)
    } else {
      k.k1 |= v << (8 * (pos - 8));
    }
  }
  return k;
}

__device__ __forceinline__ unsigned long long grouper_hash(const GrouperKey& k) {
  unsigned long long z = k.k0 * 0x9E3779B97F4A7C15ull + (k.k1 ^ (static_cast<unsigned long long>(k.mask) << 56));
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z ^= k.k1 * 0xD6E8FEB86659FD93ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct GrouperArgs {
  GrouperCols cols;
  int64_t n;                    // rows of the batch
  const uint32_t* todo;         // NULL: rows 0..n_todo; else the pending rows of the previous launch
  int64_t n_todo;
  uint32_t* next;               // pending rows for the next launch
  uint32_t* out_ids;
  uint32_t* first_row;          // [new groups of this batch] smallest row, by provisional id - base
  unsigned long long base;      // groups before this batch
  int64_t max_new;              // entries of first_row
  int insert;                   // 0: Lookup
  uint8_t* found_bits;          // Lookup: validity bitmap of out_ids (caller-zeroed, atomicOr per word) or NULL
};

__global__ __launch_bounds__(kBlock) void grouper_probe_kernel(GrouperView v, GrouperArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const uint64_t smask = static_cast<uint64_t>(v.nslots - 1);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.n_todo; i += stride) {
    const int64_t row = a.todo ? static_cast<int64_t>(a.todo[i]) : i;
    const GrouperKey key = grouper_load_key(a.cols, row);
    uint64_t s = grouper_hash(key) >> (64 - v.lg);
    for (int64_t probes = 0; probes < v.nslots; ++probes, s = (s + 1) & smask) {
      GrouperSlot* slot = v.slots + s;
      // agent-scope acquire: pairs with the winner's fence + publishing store below; the key words are then read with
      // agent-scope loads of their own (no device-wide fence per probing row)
      unsigned long long meta = __hip_atomic_load(&slot->meta, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if (meta == kSlotEmpty) {
        if (!a.insert) {
          a.out_ids[row] = 0;   // Lookup: unseen key -> null
          break;
        }
        meta = atomicCAS(&slot->meta, kSlotEmpty, kSlotClaimed);
        if (meta == kSlotEmpty) {   // ours
          const unsigned long long id = atomicAdd(&v.hdr->num_groups, 1ull);
          if (id >= static_cast<unsigned long long>(v.hdr->max_groups)) {
            atomicOr(&v.hdr->overflow, 1u);
            a.out_ids[row] = 0;
            break;   // the slot stays claimed: the call fails as a whole
          }
          slot->k0 = key.k0;
          slot->k1 = key.k1;
          v.slot_of[id] = static_cast<unsigned int>(s);
          atomicMin(&a.first_row[id - a.base], static_cast<uint32_t>(row));
          __threadfence();
          __hip_atomic_store(&slot->meta, kSlotPublished | (static_cast<unsigned long long>(key.mask) << 8) | (id << 32),
                             __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          a.out_ids[row] = static_cast<uint32_t>(id);
          break;
        }
      }
      if ((meta & 3) == kSlotClaimed) {   // somebody is writing this slot: look again in the next launch
        a.next[atomicAdd(&v.hdr->pending, 1u)] = static_cast<uint32_t>(row);
        break;
      }
      if (__hip_atomic_load(&slot->k0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key.k0 &&
          __hip_atomic_load(&slot->k1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key.k1 && ((meta >> 8) & 0xFF) == key.mask) {
        const unsigned long long id = meta >> 32;
        if (a.insert) {
          // (the group's first row so far is read before it is lowered: with a few hot keys every row used to send an
          //  atomicMin to one of a handful of addresses — 6.7e7 rows on 100 keys: 66 ms of serialised atomics; a stale,
          //  larger value read here only costs the atomic it would have cost anyway)
          if (id >= a.base && static_cast<uint32_t>(row) < __hip_atomic_load(&a.first_row[id - a.base], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            atomicMin(&a.first_row[id - a.base], static_cast<uint32_t>(row));
          }
        } else if (a.found_bits != nullptr) {
          atomicOr(reinterpret_cast<unsigned long long*>(a.found_bits) + (row >> 6), 1ull << (row & 63));
        }
        a.out_ids[row] = static_cast<uint32_t>(id);
        break;
      }
    }
  }
}

// first rows of the new groups -> bitmap over the batch's rows
__global__ __launch_bounds__(kBlock) void grouper_mark_first_kernel(const uint32_t* __restrict__ first_row, int64_t m,
                                                                    unsigned long long* __restrict__ bits) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (j >= m) return;
  const uint32_t r = first_row[j];
  atomicOr(&bits[r >> 6], 1ull << (r & 63));
}

// positions[k] = k-th first row (ascending): the group that row belongs to gets final id base + k
__global__ __launch_bounds__(kBlock) void grouper_rank_kernel(GrouperView v, const uint32_t* __restrict__ positions,
                                                              int64_t m, const uint32_t* __restrict__ ids,
                                                              unsigned long long base, uint32_t* __restrict__ rank) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (k >= m) return;
  const uint32_t prov = ids[positions[k]];
  rank[prov - base] = static_cast<uint32_t>(k);
}

// new groups: final id into the slot, key row into the uniques (slot_of is rebuilt in final-id order through tmp)
__global__ __launch_bounds__(kBlock) void grouper_renumber_groups_kernel(GrouperView v, int64_t m, unsigned long long base,
                                                                         const uint32_t* __restrict__ rank,
                                                                         uint32_t* __restrict__ slot_tmp) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (j >= m) return;
  const unsigned int s = v.slot_of[base + j];
  const unsigned long long id = base + rank[j];
  GrouperSlot* slot = v.slots + s;
  const unsigned long long meta = slot->meta;
  slot->meta = (meta & 0xFFFFFFFFull) | (id << 32);
  v.uniq_k0[id] = slot->k0;
  v.uniq_k1[id] = slot->k1;
  v.uniq_mask[id] = static_cast<unsigned int>((meta >> 8) & 0xFF);
  slot_tmp[rank[j]] = s;
}

__global__ __launch_bounds__(kBlock) void grouper_copy_u32_kernel(const uint32_t* __restrict__ src, int64_t m,
                                                                  unsigned int* __restrict__ dst) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (j < m) dst[j] = src[j];
}

__global__ __launch_bounds__(kBlock) void grouper_renumber_rows_kernel(uint32_t* __restrict__ ids, int64_t n,
                                                                       unsigned long long base,
                                                                       const uint32_t* __restrict__ rank) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    const uint32_t id = ids[i];
    if (id >= base) ids[i] = static_cast<uint32_t>(base + rank[id - base]);
  }
}

// GetUniques: column j of the unique key rows
__global__ __launch_bounds__(kBlock) void grouper_uniques_kernel(GrouperView v, int64_t g, int col, int width, int byte_pos,
                                                                 uint8_t* __restrict__ out,
                                                                 unsigned long long* __restrict__ out_valid,
                                                                 unsigned long long* __restrict__ null_count) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const bool in = i < g;
  bool valid = false;
  if (in) {
    valid = ((v.uniq_mask[i] >> col) & 1) == 0;
    unsigned long long w;
    if (byte_pos < 8) {
      w = v.uniq_k0[i] >> (8 * byte_pos);
      if (byte_pos + width > 8) w |= v.uniq_k1[i] << (8 * (8 - byte_pos));   // a column that straddles the two key words
    } else {
      w = v.uniq_k1[i] >> (8 * (byte_pos - 8));
    }
    uint8_t* p = out + i * width;
    switch (width) {
      case 1: *p = static_cast<uint8_t>(w); break;
      case 2: *reinterpret_cast<uint16_t*>(p) = static_cast<uint16_t>(w); break;
      case 4: *reinterpret_cast<uint32_t*>(p) = static_cast<uint32_t>(w); break;
      default: *reinterpret_cast<unsigned long long*>(p) = w; break;
    }
  }
  const unsigned long long b = __ballot(valid);
  if ((threadIdx.x & 63) == 0 && (i - (i & 63)) < g) {
    out_valid[i >> 6] = b;
    const int64_t rows = g - i < 64 ? g - i : 64;
    const unsigned int nulls = static_cast<unsigned int>(rows) - static_cast<unsigned int>(__popcll(b));
    if (nulls) atomicAdd(null_count, static_cast<unsigned long long>(nulls));
  }
}

// ids above `pivot` move down by one: DictionaryEncode with null_encoding = MASK drops the null's entry from the
// dictionary (DictEncodeAction, kernels/vector_hash.cc:173-270), so the groups that appeared after it renumber
__global__ __launch_bounds__(kBlock) void ids_skip_group_kernel(uint32_t* __restrict__ ids, int64_t n, uint32_t pivot) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    const uint32_t id = ids[i];
    if (id > pivot) ids[i] = id - 1;
  }
}

static int grouper_cols(const ArxSpan* cols, const int32_t* widths, int num_keys, GrouperCols* out, int64_t* length) {
  if (num_keys < 1 || num_keys > kGrouperMaxKeys) {
    set_error("Grouper: 1 to %d key columns (got %d)", kGrouperMaxKeys, num_keys);
    return ARX_NOT_IMPLEMENTED;
  }
  int pos = 0;
  int64_t n = cols[0].length;
  out->num_keys = num_keys;
  for (int j = 0; j < num_keys; ++j) {
    const int w = widths[j];
    if (w != 1 && w != 2 && w != 4 && w != 8) {
      set_error("Grouper: key column %d has byte width %d (1, 2, 4 or 8)", j, w);
      return ARX_NOT_IMPLEMENTED;
    }
    if (cols[j].length != n) {
      set_error("Grouper: key columns must all be the same length");
      return ARX_INVALID;
    }
    if (n > 0 && cols[j].data == nullptr) {
      set_error("Grouper: key column %d has no data", j);
      return ARX_INVALID;
    }
    out->data[j] = static_cast<const uint8_t*>(cols[j].data);
    out->offset[j] = cols[j].offset;
    const bool has_nulls = cols[j].validity != nullptr && cols[j].null_count != 0;
    out->valid[j] = make_bits(has_nulls ? cols[j].validity : nullptr, cols[j].offset, n);
    out->width[j] = w;
    out->byte_pos[j] = pos;
    pos += w;
  }
  if (pos > 16) {
    set_error("Grouper: the key columns take %d bytes per row; the device table holds rows of up to 16 bytes", pos);
    return ARX_NOT_IMPLEMENTED;
  }
  *length = n;
  return ARX_OK;
}

static inline unsigned grouper_grid(int64_t n) {
  return static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((n + kBlock - 1) / kBlock, 16384)));
}

struct GrouperWs {
  size_t off_first, off_rank, off_pos, off_bits, off_pend_a, off_pend_b, off_sel, total;
};

static GrouperWs grouper_ws(int64_t n) {
  auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
  const size_t rows = static_cast<size_t>(std::max<int64_t>(n, 1));
  GrouperWs w{};
  size_t o = 0;
  w.off_first = o; o = align(o + rows * 4);
  w.off_rank = o; o = align(o + rows * 4);
  w.off_pos = o; o = align(o + rows * 4);
  w.off_bits = o; o = align(o + (rows / 64 + 2) * 8);
  w.off_pend_a = o; o = align(o + rows * 4);
  w.off_pend_b = o; o = align(o + rows * 4);
  w.off_sel = o; o = align(o + selection_workspace_bytes(n));
  w.total = o;
  return w;
}

}  // namespace arx

using namespace arx;

// ---------------------------------------------------------------- var-width key columns
// A utf8 / binary key column enters the chain of tables as FIXED-WIDTH virtual columns: its length (uint32; a null is
// the length 0xFFFFFFFF, which no int32-offset string has — null stays a key value of its own, distinct from "") and
// then 12 bytes of the string per table level, zero-padded past the end, as one uint64 and one uint32 column beside
// the previous level's 4-byte id.  Equal rows agree in every virtual column; rows that differ differ in the length or
// in some chunk (the reference compares length + bytes of its encoded row the same way, row/grouper.cc:559-611 with
// the var-length part of RowTableEncoder).
__global__ __launch_bounds__(kBlock) void binary_key_lengths_kernel(Bits valid, const int32_t* __restrict__ offsets,
                                                                    int64_t n, uint32_t* __restrict__ out_len,
                                                                    unsigned int* __restrict__ max_len) {
  unsigned int mine = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const bool ok = (load_word(valid, i >> 6) >> (i & 63)) & 1;
    const unsigned int len = ok ? static_cast<unsigned int>(offsets[i + 1] - offsets[i]) : 0xFFFFFFFFu;
    out_len[i] = len;
    if (ok && len > mine) mine = len;
  }
  if (mine != 0) atomicMax(max_len, mine);
}

// utf8 / binary SORT keys: bytes [chunk_pos, chunk_pos + 8) of every string as ONE big-endian uint64 (byte 0 in the top
// byte), zero-padded past the string's end — the unsigned order of these words, chunk after chunk, then the length, is the
// bytewise lexicographic order the reference compares strings in (std::string_view::compare: memcmp of the common prefix,
// then the shorter first — kernels/vector_sort_internal.h, CompareTypeValues)
__global__ __launch_bounds__(kBlock) void binary_sort_chunk_kernel(Bits valid, const int32_t* __restrict__ offsets,
                                                                   const uint8_t* __restrict__ data, int64_t n, int64_t chunk_pos,
                                                                   uint64_t* __restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const bool ok = (load_word(valid, i >> 6) >> (i & 63)) & 1;
    const int64_t start = offsets[i];
    const int64_t len = ok ? offsets[i + 1] - start : 0;
    uint64_t k = 0;
    const int64_t have = len - chunk_pos;
    if (have > 0) {
      const uint8_t* p = data + start + chunk_pos;
      const int m = have < 8 ? static_cast<int>(have) : 8;
      for (int b = 0; b < 8; ++b) {
        if (b < m) k |= static_cast<uint64_t>(p[b]) << (8 * (7 - b));
      }
    }
    out[i] = k;
  }
}

__global__ __launch_bounds__(kBlock) void binary_key_chunk_kernel(Bits valid, const int32_t* __restrict__ offsets,
                                                                  const uint8_t* __restrict__ data, int64_t n,
                                                                  int64_t chunk_pos, uint64_t* __restrict__ out_lo,
                                                                  uint32_t* __restrict__ out_hi) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const bool ok = (load_word(valid, i >> 6) >> (i & 63)) & 1;
    const int64_t start = offsets[i];
    const int64_t len = ok ? offsets[i + 1] - start : 0;
    uint64_t lo = 0;
    uint32_t hi = 0;
    const int64_t have = len - chunk_pos;   // bytes of this string at and after the chunk
    if (have > 0) {
      const uint8_t* p = data + start + chunk_pos;
      const int m = have < 12 ? static_cast<int>(have) : 12;
      for (int b = 0; b < 8; ++b) {
        if (b < m) lo |= static_cast<uint64_t>(p[b]) << (8 * b);
      }
      for (int b = 8; b < 12; ++b) {
        if (b < m) hi |= static_cast<uint32_t>(p[b]) << (8 * (b - 8));
      }
    }
    out_lo[i] = lo;
    out_hi[i] = hi;
  }
}

// ---- var-width keys in ONE pass (round 4): instead of ceil(longest / 12) table levels, a string enters the tables as
// its length and a 64-bit hash of its bytes (12 bytes, one level whatever the lengths); the groups are then VERIFIED —
// every row's bytes against the bytes of its group's first row — and only a batch in which two different strings of
// one length share a hash (about d^2 / 2^65 for d distinct strings) is grouped again by the exact chunk columns above.
// The reference hashes the encoded var-length row once as well and compares rows on a hash match
// (row/grouper.cc:695-815, key_map / key_compare).  Bytes are read as the aligned 8-byte words that hold them: a word
// is loaded only if it holds a byte of the string, so nothing outside the value buffer's words is touched.
// bytes [pos, pos + 8) of the string data[start, start + len), zero past its end (pos < len, pos a multiple of 8): the one
// or two aligned words that hold them, the second only if it holds a byte of the string
__device__ __forceinline__ uint64_t string_word(const uint8_t* data, int64_t start, int64_t len, int64_t pos) {
  const uint64_t addr = reinterpret_cast<uint64_t>(data) + static_cast<uint64_t>(start + pos);
  const uint64_t* wp = reinterpret_cast<const uint64_t*>(addr & ~uint64_t(7));
  const int sh = static_cast<int>(addr & 7) * 8;
  uint64_t w = wp[0] >> sh;
  const int64_t rem = len - pos;
  if (sh != 0 && (8 - (sh >> 3)) < rem) w |= wp[1] << (64 - sh);
  if (rem < 8) w &= (uint64_t(1) << (8 * rem)) - 1;
  return w;
}

// The hash is a XOR over the string's 8-byte words of a mix of (word, word number), finished with the length: the words
// can be folded in any order, so a lane folds a short string alone and a whole wave folds a long one 512 bytes a step
// (coalesced) — one kernel for columns of any mix of lengths, cost proportional to the bytes.
__device__ __forceinline__ uint64_t string_word_mix(uint64_t w, int64_t k) {
  uint64_t x = (w ^ (static_cast<uint64_t>(k + 1) * 0x9E3779B97F4A7C15ull)) * 0x9FB21C651E98DF25ull;
  x ^= x >> 32;
  return x * 0xD6E8FEB86659FD93ull;
}
__device__ __forceinline__ uint64_t string_hash_finish(uint64_t acc, int64_t len) {
  uint64_t h = acc ^ (static_cast<uint64_t>(len) * 0xC2B2AE3D27D4EB4Full) ^ 0x9E3779B97F4A7C15ull;
  h = (h ^ (h >> 33)) * 0xFF51AFD7ED558CCDull;
  h = (h ^ (h >> 33)) * 0xC4CEB9FE1A85EC53ull;
  return h ^ (h >> 33);
}
constexpr int64_t kStringLaneBytes = 64;   // longer strings are folded by the whole wave

__device__ __forceinline__ uint64_t wave_xor_u64(uint64_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v ^= shfl_u64(v, lane_id() ^ d);
  return v;
}

__global__ __launch_bounds__(kBlock) void binary_key_hash_kernel(Bits valid, const int32_t* __restrict__ offsets,
                                                                 const uint8_t* __restrict__ data, int64_t n, uint64_t keep_mask,
                                                                 uint64_t* __restrict__ out_hash) {
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  for (int64_t base = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6)) * 64; base < n; base += nwaves * 64) {
    const int64_t i = base + lane;
    const bool ok = i < n && ((load_word(valid, i >> 6) >> (i & 63)) & 1);
    const int64_t start = ok ? offsets[i] : 0;
    const int64_t len = ok ? offsets[i + 1] - start : 0;
    uint64_t acc = 0;
    if (ok && len <= kStringLaneBytes) {
      for (int64_t pos = 0; pos < len; pos += 8) acc ^= string_word_mix(string_word(data, start, len, pos), pos >> 3);
    }
    uint64_t longs = __ballot(ok && len > kStringLaneBytes);
    while (longs != 0) {
      const int src = __ffsll(static_cast<unsigned long long>(longs)) - 1;
      longs &= longs - 1;
      const int64_t s0 = shfl_u64(static_cast<uint64_t>(start), src);
      const int64_t sl = shfl_u64(static_cast<uint64_t>(len), src);
      uint64_t part = 0;
      for (int64_t pos = static_cast<int64_t>(lane) * 8; pos < sl; pos += 512) part ^= string_word_mix(string_word(data, s0, sl, pos), pos >> 3);
      part = wave_xor_u64(part);
      if (lane == src) acc = part;
    }
    if (i < n) out_hash[i] = ok ? (string_hash_finish(acc, len) & keep_mask) : 0;
  }
}

// mismatches += rows whose bytes differ from the bytes of the first row of their group (the validity and the length are
// compared too, although the length is a key column of its own)
__global__ __launch_bounds__(kBlock) void binary_key_verify_kernel(Bits valid, const int32_t* __restrict__ offsets,
                                                                   const uint8_t* __restrict__ data, int64_t n,
                                                                   const uint32_t* __restrict__ ids, const uint32_t* __restrict__ first_rows,
                                                                   unsigned long long* __restrict__ mismatches) {
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  unsigned int bad = 0;
  for (int64_t base = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6)) * 64; base < n; base += nwaves * 64) {
    const int64_t i = base + lane;
    const int64_t f = i < n ? static_cast<int64_t>(first_rows[ids[i]]) : 0;
    const bool check = i < n && f != i;
    const bool ok = check && ((load_word(valid, i >> 6) >> (i & 63)) & 1);
    const bool fok = check && ((load_word(valid, f >> 6) >> (f & 63)) & 1);
    const int64_t a0 = ok ? offsets[i] : 0, b0 = fok ? offsets[f] : 0;
    const int64_t len = ok ? offsets[i + 1] - a0 : 0;
    const int64_t flen = fok ? offsets[f + 1] - b0 : 0;
    bool differs = check && (ok != fok || len != flen);
    const bool bytes = check && ok && !differs;
    if (bytes && len <= kStringLaneBytes) {
      for (int64_t pos = 0; pos < len && !differs; pos += 8) differs = string_word(data, a0, len, pos) != string_word(data, b0, len, pos);
    }
    uint64_t longs = __ballot(bytes && len > kStringLaneBytes);
    while (longs != 0) {
      const int src = __ffsll(static_cast<unsigned long long>(longs)) - 1;
      longs &= longs - 1;
      const int64_t sa = shfl_u64(static_cast<uint64_t>(a0), src), sb = shfl_u64(static_cast<uint64_t>(b0), src);
      const int64_t sl = shfl_u64(static_cast<uint64_t>(len), src);
      bool d = false;
      for (int64_t pos = static_cast<int64_t>(lane) * 8; pos < sl && !d; pos += 512) d = string_word(data, sa, sl, pos) != string_word(data, sb, sl, pos);
      const bool any = __ballot(d) != 0;
      if (lane == src) differs = any;
    }
    bad += differs ? 1u : 0u;
  }
  const unsigned int total = wave_reduce_sum_u32(bad);
  if (lane == 0 && total != 0) atomicAdd(mismatches, static_cast<unsigned long long>(total));
}

// first_rows[g] = the smallest row whose group id is g (caller fills first_rows with 0xFF bytes)
// The smallest (last = 0) or largest (last = 1) row of every group among the rows whose validity bit is set: what
// GroupedFirstLastImpl / GroupedOneImpl keep per group is the first / last NON-NULL value in row order
// (kernels/hash_aggregate.cc:775-808, :1575-1590) — here the row number of it, the value is one take away.  Largest rows
// are kept as 0xFFFFFFFE - row so that both ends are one atomicMin and 0xFFFFFFFF stays "no row".
__global__ __launch_bounds__(kBlock) void group_edge_rows_kernel(const uint32_t* __restrict__ ids, Bits valid, int64_t n,
                                                                 int64_t num_groups, int last, unsigned int* __restrict__ rows) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint32_t g = ids[i];
    if (g >= num_groups) continue;
    if (valid.base != nullptr && ((load_word(valid, i >> 6) >> (i & 63)) & 1ull) == 0) continue;
    const unsigned int mine = last ? 0xFFFFFFFEu - static_cast<unsigned int>(i) : static_cast<unsigned int>(i);
    if (rows[g] > mine) atomicMin(&rows[g], mine);   // (read first: most rows of a group lose)
  }
}

// rows[] back to row numbers, groups without a row to row 0 + a cleared bit of has_row (a validity bitmap for the take)
__global__ __launch_bounds__(kBlock) void group_edge_rows_finish_kernel(unsigned int* __restrict__ rows, int64_t num_groups, int last,
                                                                        uint64_t* __restrict__ has_row) {
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; g < (num_groups + 63) / 64 * 64;
       g += static_cast<int64_t>(gridDim.x) * kBlock) {
    bool has = false;
    if (g < num_groups) {
      const unsigned int r = rows[g];
      has = r != 0xFFFFFFFFu;
      rows[g] = !has ? 0u : (last ? 0xFFFFFFFEu - r : r);
    }
    const uint64_t word = __ballot(has);
    if ((threadIdx.x & 63) == 0) has_row[g >> 6] = word;
  }
}

__global__ __launch_bounds__(kBlock) void group_first_rows_kernel(const uint32_t* __restrict__ ids, int64_t n,
                                                                  int64_t num_groups, unsigned int* __restrict__ first) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint32_t g = ids[i];
    // rows of one group mostly lose: read before the atomic so that only candidates reach the L2's atomic units
    if (g < num_groups && first[g] > static_cast<unsigned int>(i)) atomicMin(&first[g], static_cast<unsigned int>(i));
  }
}

extern "C" {

size_t arx_grouper_state_bytes(int64_t max_groups) {
  if (max_groups < 1) max_groups = 1;
  const int64_t s = grouper_slots_for(max_groups);
  return 64 + static_cast<size_t>(s) * sizeof(GrouperSlot) + static_cast<size_t>(max_groups) * (8 + 8 + 4 + 4) + 256;
}

int arx_grouper_init(void* state, int64_t max_groups, void* stream) {
  if (state == nullptr || max_groups < 1 || max_groups >= (int64_t(1) << 31)) {
    set_error("Grouper: state is NULL or max_groups out of range");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  GrouperView v = grouper_view(state, max_groups);
  ARX_HIP(hipMemsetAsync(state, 0, 64 + static_cast<size_t>(v.nslots) * sizeof(GrouperSlot), st));
  GrouperHeader h{};
  h.max_groups = max_groups;
  h.slots = v.nslots;
  ARX_HIP(hipMemcpyAsync(state, &h, sizeof(h), hipMemcpyHostToDevice, st));
  ARX_HIP(hipStreamSynchronize(st));
  return ARX_OK;
}

size_t arx_grouper_consume_workspace_bytes(int64_t length) { return grouper_ws(length).total; }

static int grouper_run(void* state, int64_t max_groups, const ArxSpan* key_columns, const int32_t* key_byte_widths,
                       int num_keys, void* ws, size_t ws_bytes, uint32_t* out_group_ids, uint8_t* out_found_bits,
                       int insert, void* stream) {
  if (state == nullptr || key_columns == nullptr || key_byte_widths == nullptr) {
    set_error("Grouper: NULL argument");
    return ARX_INVALID;
  }
  GrouperArgs a{};
  int64_t n = 0;
  const int rc = grouper_cols(key_columns, key_byte_widths, num_keys, &a.cols, &n);
  if (rc != ARX_OK) return rc;
  if (n == 0) return ARX_OK;
  if (n >= (int64_t(1) << 32) - 64) {
    set_error("Grouper: at most 2^32 - 65 rows per batch");
    return ARX_INVALID;
  }
  if (out_group_ids == nullptr) {
    set_error("Grouper: out_group_ids is NULL");
    return ARX_INVALID;
  }
  const GrouperWs plan = grouper_ws(n);
  if (ws == nullptr || ws_bytes < plan.total || (reinterpret_cast<uint64_t>(ws) & 255) != 0) {
    set_error("Grouper: workspace too small or not 256-byte aligned (%zu needed)", plan.total);
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  GrouperView v = grouper_view(state, max_groups);
  GrouperHeader h{};
  ARX_HIP(hipMemcpyAsync(&h, v.hdr, sizeof(h), hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  if (h.max_groups != max_groups) {
    set_error("Grouper: state was initialised for %lld groups, called with %lld", static_cast<long long>(h.max_groups),
              static_cast<long long>(max_groups));
    return ARX_INVALID;
  }
  if (h.overflow) {
    set_error("Grouper: the table overflowed in an earlier call");
    return ARX_INVALID;
  }
  uint8_t* w = static_cast<uint8_t*>(ws);
  uint32_t* first_row = reinterpret_cast<uint32_t*>(w + plan.off_first);
  uint32_t* rank = reinterpret_cast<uint32_t*>(w + plan.off_rank);
  uint32_t* positions = reinterpret_cast<uint32_t*>(w + plan.off_pos);
  unsigned long long* bits = reinterpret_cast<unsigned long long*>(w + plan.off_bits);
  uint32_t* pend[2] = {reinterpret_cast<uint32_t*>(w + plan.off_pend_a), reinterpret_cast<uint32_t*>(w + plan.off_pend_b)};
  const unsigned long long base = h.num_groups;
  const int64_t max_new = std::min<int64_t>(n, max_groups - static_cast<int64_t>(base));
  a.n = n;
  a.out_ids = out_group_ids;
  a.first_row = first_row;
  a.base = base;
  a.max_new = max_new;
  a.insert = insert;
  a.found_bits = out_found_bits;
  if (insert && max_new > 0) ARX_HIP(hipMemsetAsync(first_row, 0xFF, static_cast<size_t>(max_new) * 4, st));
  if (!insert && out_found_bits != nullptr) {
    ARX_HIP(hipMemsetAsync(out_found_bits, 0, static_cast<size_t>((n + 63) / 64) * 8, st));
  }
  // probe; rows that met a slot in the middle of being written are looked at again by the next launch
  const uint32_t* todo = nullptr;
  int64_t n_todo = n;
  for (int round = 0; n_todo > 0; ++round) {
    a.todo = todo;
    a.n_todo = n_todo;
    a.next = pend[round & 1];
    hipLaunchKernelGGL(grouper_probe_kernel, dim3(grouper_grid(n_todo)), dim3(kBlock), 0, st, v, a);
    ARX_CHECK_LAUNCH("grouper_probe_kernel");
    ARX_HIP(hipMemcpyAsync(&h, v.hdr, sizeof(h), hipMemcpyDeviceToHost, st));
    ARX_HIP(hipStreamSynchronize(st));
    if (h.overflow) {
      set_error("Grouper: more than %lld distinct key rows", static_cast<long long>(max_groups));
      return ARX_INVALID;
    }
    n_todo = h.pending;
    todo = pend[round & 1];
    if (n_todo > 0) ARX_HIP(hipMemsetAsync(&v.hdr->pending, 0, 4, st));
    if (round > 64) {
      set_error("Grouper: internal error (pending rows do not drain)");
      return ARX_INVALID;
    }
  }
  const int64_t m = static_cast<int64_t>(h.num_groups - base);
  if (!insert || m == 0) return ARX_OK;
  // new groups in order of first appearance
  ARX_HIP(hipMemsetAsync(bits, 0, static_cast<size_t>(n / 64 + 2) * 8, st));
  const unsigned gm = static_cast<unsigned>((m + kBlock - 1) / kBlock);   // one thread per new group
  hipLaunchKernelGGL(grouper_mark_first_kernel, dim3(gm), dim3(kBlock), 0, st, first_row, m, bits);
  ARX_CHECK_LAUNCH("grouper_mark_first_kernel");
  int64_t got = 0;
  const int rc2 = selection_bit_positions(bits, 0, n, false, w + plan.off_sel, ws_bytes - plan.off_sel, positions, &got, st);
  if (rc2 != ARX_OK) return rc2;
  if (got != m) {
    set_error("Grouper: internal error (%lld first rows for %lld new groups)", static_cast<long long>(got),
              static_cast<long long>(m));
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(grouper_rank_kernel, dim3(gm), dim3(kBlock), 0, st, v, positions, m, out_group_ids, base, rank);
  ARX_CHECK_LAUNCH("grouper_rank_kernel");
  uint32_t* slot_tmp = first_row;   // (first rows are no longer needed)
  hipLaunchKernelGGL(grouper_renumber_groups_kernel, dim3(gm), dim3(kBlock), 0, st, v, m, base, rank, slot_tmp);
  ARX_CHECK_LAUNCH("grouper_renumber_groups_kernel");
  hipLaunchKernelGGL(grouper_copy_u32_kernel, dim3(gm), dim3(kBlock), 0, st, slot_tmp, m, v.slot_of + base);
  ARX_CHECK_LAUNCH("grouper_copy_u32_kernel");
  hipLaunchKernelGGL(grouper_renumber_rows_kernel, dim3(grouper_grid(n)), dim3(kBlock), 0, st, out_group_ids, n, base, rank);
  ARX_CHECK_LAUNCH("grouper_renumber_rows_kernel");
  ARX_HIP(hipStreamSynchronize(st));
  return ARX_OK;
}

int arx_grouper_consume(void* state, int64_t max_groups, const ArxSpan* key_columns, const int32_t* key_byte_widths,
                        int num_keys, void* ws, size_t ws_bytes, uint32_t* out_group_ids, void* stream) {
  return grouper_run(state, max_groups, key_columns, key_byte_widths, num_keys, ws, ws_bytes, out_group_ids, nullptr, 1,
                     stream);
}

int arx_grouper_lookup(void* state, int64_t max_groups, const ArxSpan* key_columns, const int32_t* key_byte_widths,
                       int num_keys, void* ws, size_t ws_bytes, uint32_t* out_group_ids, uint8_t* out_validity,
                       void* stream) {
  if (out_validity == nullptr || (reinterpret_cast<uint64_t>(out_validity) & 7) != 0) {
    set_error("Grouper: out_validity must be an 8-byte aligned bitmap");
    return ARX_INVALID;
  }
  return grouper_run(state, max_groups, key_columns, key_byte_widths, num_keys, ws, ws_bytes, out_group_ids, out_validity, 0,
                     stream);
}

int arx_group_ids_skip_group(uint32_t* group_ids, int64_t length, uint32_t skipped_id, void* stream) {
  if (length < 0 || (length > 0 && group_ids == nullptr)) {
    set_error("bad arguments to arx_group_ids_skip_group");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  hipLaunchKernelGGL(ids_skip_group_kernel, dim3(grouper_grid(length)), dim3(kBlock), 0, as_stream(stream), group_ids, length,
                     skipped_id);
  ARX_CHECK_LAUNCH("ids_skip_group_kernel");
  return ARX_OK;
}

int arx_grouper_num_groups(void* state, int64_t* out_num_groups, void* stream) {
  if (state == nullptr || out_num_groups == nullptr) {
    set_error("Grouper: NULL argument");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  GrouperHeader h{};
  ARX_HIP(hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  *out_num_groups = static_cast<int64_t>(h.num_groups);
  return ARX_OK;
}

int arx_grouper_get_uniques(void* state, int64_t max_groups, const int32_t* key_byte_widths, int num_keys, int key_index,
                            void* out_values, uint8_t* out_validity, int64_t* out_null_count, void* stream) {
  if (state == nullptr || key_byte_widths == nullptr || out_null_count == nullptr) {
    set_error("Grouper: NULL argument");
    return ARX_INVALID;
  }
  if (num_keys < 1 || num_keys > kGrouperMaxKeys || key_index < 0 || key_index >= num_keys) {
    set_error("Grouper: key_index out of range");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  GrouperView v = grouper_view(state, max_groups);
  GrouperHeader h{};
  ARX_HIP(hipMemcpyAsync(&h, v.hdr, sizeof(h), hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  const int64_t g = static_cast<int64_t>(h.num_groups);
  *out_null_count = 0;
  if (g == 0) return ARX_OK;
  if (out_values == nullptr || out_validity == nullptr || (reinterpret_cast<uint64_t>(out_validity) & 7) != 0) {
    set_error("Grouper: out_values / out_validity (8-byte aligned bitmap of ceil(g / 64) words) are required");
    return ARX_INVALID;
  }
  int pos = 0;
  for (int j = 0; j < key_index; ++j) pos += key_byte_widths[j];
  unsigned long long* counter = reinterpret_cast<unsigned long long*>(&v.hdr->pad[1]);   // scratch word of the header
  ARX_HIP(hipMemsetAsync(counter, 0, 8, st));
  const unsigned grid = static_cast<unsigned>(((g + 63) / 64 * 64 + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(grouper_uniques_kernel, dim3(grid), dim3(kBlock), 0, st, v, g, key_index, key_byte_widths[key_index],
                     pos, static_cast<uint8_t*>(out_values), reinterpret_cast<unsigned long long*>(out_validity), counter);
  ARX_CHECK_LAUNCH("grouper_uniques_kernel");
  unsigned long long nulls = 0;
  ARX_HIP(hipMemcpyAsync(&nulls, counter, 8, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  *out_null_count = static_cast<int64_t>(nulls);
  return ARX_OK;
}


int arx_binary_key_lengths(const ArxBinarySpan* values, uint32_t* out_lengths, int64_t* out_max_length, void* ws,
                           void* stream) {
  if (values == nullptr || out_max_length == nullptr) {
    set_error("binary key lengths: NULL argument");
    return ARX_INVALID;
  }
  *out_max_length = 0;
  const int64_t n = values->length;
  if (n <= 0) return ARX_OK;
  if (values->offsets == nullptr || out_lengths == nullptr || ws == nullptr) {
    set_error("binary key lengths: NULL buffer");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  ARX_HIP(hipMemsetAsync(ws, 0, 8, st));
  const Bits valid = make_bits(values->null_count == 0 ? nullptr : values->validity, values->offset, n);
  hipLaunchKernelGGL(binary_key_lengths_kernel, dim3(grouper_grid(n)), dim3(kBlock), 0, st, valid,
                     values->offsets + values->offset, n, out_lengths, static_cast<unsigned int*>(ws));
  ARX_CHECK_LAUNCH("binary_key_lengths_kernel");
  unsigned int m = 0;
  ARX_HIP(hipMemcpyAsync(&m, ws, 4, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  *out_max_length = m;
  return ARX_OK;
}

int arx_binary_sort_chunk(const ArxBinarySpan* values, int64_t chunk_index, uint64_t* out_keys, void* stream) {
  if (values == nullptr || chunk_index < 0) {
    set_error("binary sort chunk: NULL values or negative chunk");
    return ARX_INVALID;
  }
  const int64_t n = values->length;
  if (n <= 0) return ARX_OK;
  if (values->offsets == nullptr || out_keys == nullptr) {
    set_error("binary sort chunk: NULL buffer");
    return ARX_INVALID;
  }
  const Bits valid = make_bits(values->null_count == 0 ? nullptr : values->validity, values->offset, n);
  hipLaunchKernelGGL(binary_sort_chunk_kernel, dim3(grouper_grid(n)), dim3(kBlock), 0, as_stream(stream), valid,
                     values->offsets + values->offset, static_cast<const uint8_t*>(values->data), n, chunk_index * 8, out_keys);
  ARX_CHECK_LAUNCH("binary_sort_chunk_kernel");
  return ARX_OK;
}

int arx_binary_key_chunk(const ArxBinarySpan* values, int64_t chunk_index, uint64_t* out_lo, uint32_t* out_hi,
                         void* stream) {
  if (values == nullptr || chunk_index < 0) {
    set_error("binary key chunk: NULL values or negative chunk");
    return ARX_INVALID;
  }
  const int64_t n = values->length;
  if (n <= 0) return ARX_OK;
  if (values->offsets == nullptr || out_lo == nullptr || out_hi == nullptr) {
    set_error("binary key chunk: NULL buffer");
    return ARX_INVALID;
  }
  const Bits valid = make_bits(values->null_count == 0 ? nullptr : values->validity, values->offset, n);
  hipLaunchKernelGGL(binary_key_chunk_kernel, dim3(grouper_grid(n)), dim3(kBlock), 0, as_stream(stream), valid,
                     values->offsets + values->offset, static_cast<const uint8_t*>(values->data), n, chunk_index * 12,
                     out_lo, out_hi);
  ARX_CHECK_LAUNCH("binary_key_chunk_kernel");
  return ARX_OK;
}

int arx_binary_key_hash(const ArxBinarySpan* values, int hash_bits, uint64_t* out_hash, void* stream) {
  if (values == nullptr || hash_bits < 1 || hash_bits > 64) {
    set_error("binary key hash: NULL values or hash_bits outside 1 .. 64");
    return ARX_INVALID;
  }
  const int64_t n = values->length;
  if (n <= 0) return ARX_OK;
  if (values->offsets == nullptr || out_hash == nullptr) {
    set_error("binary key hash: NULL buffer");
    return ARX_INVALID;
  }
  const Bits valid = make_bits(values->null_count == 0 ? nullptr : values->validity, values->offset, n);
  const uint64_t keep = hash_bits >= 64 ? ~uint64_t(0) : ((uint64_t(1) << hash_bits) - 1);
  hipLaunchKernelGGL(binary_key_hash_kernel, dim3(grouper_grid(n)), dim3(kBlock), 0, as_stream(stream), valid,
                     values->offsets + values->offset, static_cast<const uint8_t*>(values->data), n, keep, out_hash);
  ARX_CHECK_LAUNCH("binary_key_hash_kernel");
  return ARX_OK;
}

int arx_binary_key_verify(const ArxBinarySpan* values, const uint32_t* group_ids, const uint32_t* first_rows,
                          int64_t* out_mismatches, void* ws, void* stream) {
  if (values == nullptr || out_mismatches == nullptr || ws == nullptr) {
    set_error("binary key verify: NULL argument");
    return ARX_INVALID;
  }
  *out_mismatches = 0;
  const int64_t n = values->length;
  if (n <= 0) return ARX_OK;
  if (values->offsets == nullptr || group_ids == nullptr || first_rows == nullptr) {
    set_error("binary key verify: NULL buffer");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  ARX_HIP(hipMemsetAsync(ws, 0, 8, st));
  const Bits valid = make_bits(values->null_count == 0 ? nullptr : values->validity, values->offset, n);
  hipLaunchKernelGGL(binary_key_verify_kernel, dim3(grouper_grid(n)), dim3(kBlock), 0, st, valid, values->offsets + values->offset,
                     static_cast<const uint8_t*>(values->data), n, group_ids, first_rows, static_cast<unsigned long long*>(ws));
  ARX_CHECK_LAUNCH("binary_key_verify_kernel");
  ARX_HIP(hipMemcpyAsync(out_mismatches, ws, 8, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  return ARX_OK;
}

int arx_group_edge_rows(const uint32_t* group_ids, const void* validity, int64_t validity_offset, int64_t length,
                        int64_t num_groups, int last, uint32_t* out_rows, void* out_has_row, void* stream) {
  if (num_groups <= 0) return ARX_OK;
  if (out_rows == nullptr || out_has_row == nullptr || (length > 0 && group_ids == nullptr)) {
    set_error("group edge rows: NULL buffer");
    return ARX_INVALID;
  }
  if (length >= (int64_t(1) << 32) - 2) {
    set_error("group edge rows: row numbers are uint32");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  ARX_HIP(hipMemsetAsync(out_rows, 0xFF, static_cast<size_t>(num_groups) * 4, st));
  if (length > 0) {
    const Bits vb = make_bits(validity, validity_offset, length);
    hipLaunchKernelGGL(group_edge_rows_kernel, dim3(grouper_grid(length)), dim3(kBlock), 0, st, group_ids, vb, length, num_groups,
                       last != 0 ? 1 : 0, out_rows);
    ARX_CHECK_LAUNCH("group_edge_rows_kernel");
  }
  hipLaunchKernelGGL(group_edge_rows_finish_kernel, dim3(grouper_grid((num_groups + 63) / 64 * 64)), dim3(kBlock), 0, st, out_rows,
                     num_groups, last != 0 ? 1 : 0, static_cast<uint64_t*>(out_has_row));
  ARX_CHECK_LAUNCH("group_edge_rows_finish_kernel");
  return ARX_OK;
}

int arx_group_first_rows(const uint32_t* group_ids, int64_t length, int64_t num_groups, uint32_t* out_first_rows,
                         void* stream) {
  if (num_groups <= 0) return ARX_OK;
  if (out_first_rows == nullptr || (length > 0 && group_ids == nullptr)) {
    set_error("group first rows: NULL buffer");
    return ARX_INVALID;
  }
  if (length >= (int64_t(1) << 32) - 1) {
    set_error("group first rows: row numbers are uint32");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  ARX_HIP(hipMemsetAsync(out_first_rows, 0xFF, static_cast<size_t>(num_groups) * 4, st));
  if (length <= 0) return ARX_OK;
  hipLaunchKernelGGL(group_first_rows_kernel, dim3(grouper_grid(length)), dim3(kBlock), 0, st, group_ids, length,
                     num_groups, out_first_rows);
  ARX_CHECK_LAUNCH("group_first_rows_kernel");
  return ARX_OK;
}

}  // extern "C"
