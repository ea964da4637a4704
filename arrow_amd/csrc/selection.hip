// Selection kernels for gfx950: filter (stream compaction), mask -> indices, take (gather),
// index bounds check.
//
// What they restate (semantics only; the structure is GPU-first):
//   GetFilterOutputSize        cpp/src/arrow/compute/kernels/vector_selection_filter_internal.cc:62-114
//   PrimitiveFilterImpl::Exec  same file :238-372  (WriteValue/WriteNull :376-418)
//   GetTakeIndicesFromBitmap   cpp/src/arrow/compute/kernels/vector_selection_take_internal.cc:62-168
//   FixedWidthTakeImpl / Gather  same file :339-380, gather_internal.h:60-165
//   CheckIndexBounds           cpp/src/arrow/util/int_util.cc:530-587
//
// Data layout in HBM: Arrow columnar buffers as-is (values, LSB-first validity bitmap,
// logical offset).  Decomposition:
//   wave tile  = 4096 rows = 64 lanes x one 64-bit mask word per lane
//   tile group = 64 wave tiles (count kernel workgroup; its 4 waves take 16 tiles each)
//   K1 count_kernel   : per-tile popcount of the emit mask -> tile_counts[], group totals
//   K2 scan_kernel    : exclusive scan of the group totals (single workgroup) + grand total
//   K3 compact_kernel : one wave per tile; rank by popcount-prefix inside the mask word,
//                       stage emitted elements in a per-wave LDS ring and flush them as
//                       2 KiB, 2 KiB-aligned runs of 16 B/lane stores; output validity by
//                       per-lane software PEXT + LDS ds_or, boundary words via atomicOr.
// No workgroup barrier is used in K3: waves are independent.
#include "arx_common.h"

#include <string.h>

#include <algorithm>
#include <type_traits>

namespace arx {

// ------------------------------------------------------------------ workspace
struct FilterWsHeader {
  int64_t total;     // number of emitted rows (written by scan_kernel)
  int64_t ntiles;
  int64_t ngroups;
  int64_t length;
  int64_t valid;     // emitted rows with a valid output slot (scan_kernel, when the count ran with the values' validity)
  int64_t pad[3];
};
static_assert(sizeof(FilterWsHeader) == 64, "header is one 64-byte line");

struct FilterWsView {
  FilterWsHeader* hdr;
  int64_t* group_excl;    // [ngroups]
  int64_t* group_total;   // [ngroups]
  uint32_t* tile_counts;  // [ngroups * 64]
  int64_t* group_valid;   // [ngroups] emitted rows whose OUTPUT slot is valid (arx_filter_count_nulls)
};

static inline int64_t num_tiles(int64_t length) { return ceil_div(length, kTileRows); }
static inline int64_t num_groups(int64_t length) {
  return ceil_div(num_tiles(length), kTilesPerGroup);
}

static inline FilterWsView ws_view(void* ws, int64_t length) {
  FilterWsView v;
  const int64_t ng = num_groups(length);
  uint8_t* p = static_cast<uint8_t*>(ws);
  v.hdr = reinterpret_cast<FilterWsHeader*>(p);
  v.group_excl = reinterpret_cast<int64_t*>(p + 64);
  v.group_total = v.group_excl + ng;
  v.tile_counts = reinterpret_cast<uint32_t*>(v.group_total + ng);
  v.group_valid = reinterpret_cast<int64_t*>(v.tile_counts + ng * kTilesPerGroup);
  return v;
}

// The emit mask of logical word w:  DROP: mask & mask_valid;  EMIT_NULL: mask | ~mask_valid
// (NextAndWord / NextOrNotWord in GetBitmapFilterOutputSize).
// `invert` selects the clear bits instead (used for the null partition of sort_indices).
__device__ __forceinline__ uint64_t emit_word(const Bits& mask, const Bits& mvalid, int64_t w,
                                              bool emit_null, bool invert,
                                              uint64_t* mask_valid_out) {
  uint64_t m = load_word(mask, w);
  const uint64_t mv = load_word(mvalid, w);  // all ones (within length) if no validity
  *mask_valid_out = mv;
  if (emit_null || invert) {
    // rows past `length` must not be emitted: bound by the mask bitmap's own length mask
    const uint64_t in_range = load_word(Bits{nullptr, 0, mask.length, 0}, w);
    if (invert) m = ~m & in_range;
    if (emit_null) return (m | ~mv) & in_range;
  }
  return m & mv;
}

// ------------------------------------------------------------------ K1: count
// VALID: also counts the emitted rows whose OUTPUT slot is valid — mask slot valid (a null mask slot emits a null under
// EMIT_NULL) and value valid — so the output's null count comes back with its length, in the one read-back the caller
// needs anyway for the allocation (the reference leaves null_count unknown and counts lazily on the host,
// vector_selection_filter_internal.cc:462-467; a device array's bitmap cannot be counted there).
template <bool VALID>
__global__ __launch_bounds__(kBlock) void count_kernel(Bits mask, Bits mvalid, Bits vvalid, int emit_null,
                                                       int invert, int64_t ntiles,
                                                       uint32_t* tile_counts,
                                                       int64_t* group_total, int64_t* group_valid) {
  __shared__ uint32_t wave_sums[kWavesPerBlock];
  __shared__ uint32_t wave_valid[kWavesPerBlock];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t group = blockIdx.x;
  const int64_t t0 = group * kTilesPerGroup + wave * 16;
  uint32_t mine = 0;
  uint32_t kv = 0;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int64_t t = t0 + i;
    uint32_t k = 0;
    if (t < ntiles) {
      uint64_t mv;
      const uint64_t e = emit_word(mask, mvalid, t * 64 + lane, emit_null != 0, invert != 0, &mv);
      k = __popcll(e);
      if constexpr (VALID) kv += __popcll(e & mv & load_word(vvalid, t * 64 + lane));
    }
    const uint32_t s = wave_reduce_sum_u32(k);
    if (lane == i) mine = s;
  }
  if (lane < 16) tile_counts[t0 + lane] = mine;  // tile_counts is padded to ngroups*64
  const uint32_t wsum = wave_reduce_sum_u32(lane < 16 ? mine : 0u);
  if (lane == 0) wave_sums[wave] = wsum;
  if constexpr (VALID) {
    const uint32_t vsum = wave_reduce_sum_u32(kv);
    if (lane == 0) wave_valid[wave] = vsum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t s = 0, v = 0;
    for (int i = 0; i < kWavesPerBlock; ++i) {
      s += wave_sums[i];
      if constexpr (VALID) v += wave_valid[i];
    }
    group_total[group] = s;
    if constexpr (VALID) group_valid[group] = v;
  }
}

// ------------------------------------------------------------------ K2: scan of group totals
__global__ __launch_bounds__(1024) void scan_kernel(const int64_t* group_total, int64_t ngroups,
                                                    int64_t* group_excl, FilterWsHeader* hdr,
                                                    const int64_t* group_valid = nullptr) {
  __shared__ int64_t wave_tot[16];
  __shared__ int64_t carry_s;
  __shared__ unsigned long long valid_s;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  if (tid == 0) valid_s = 0;
  __syncthreads();
  if (group_valid != nullptr) {   // (kernel-uniform) the sum of the groups' valid counts
    int64_t mine = 0;
    for (int64_t i = tid; i < ngroups; i += 1024) mine += group_valid[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if (lane == 0 && mine != 0) atomicAdd(&valid_s, static_cast<unsigned long long>(mine));
    __syncthreads();
  }
  for (int64_t base = 0; base < ngroups; base += 1024) {
    const int64_t i = base + tid;
    const int64_t v = i < ngroups ? group_total[i] : 0;
    // inclusive scan inside the wave (64-bit)
    int64_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int64_t n = __shfl_up(x, d, 64);
      if (lane >= d) x += n;
    }
    if (lane == 63) wave_tot[wave] = x;
    __syncthreads();
    int64_t wave_prefix = 0;
    for (int k = 0; k < wave; ++k) wave_prefix += wave_tot[k];
    const int64_t carry = carry_s;
    if (i < ngroups) group_excl[i] = carry + wave_prefix + x - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wave_prefix + x;
    __syncthreads();
  }
  if (tid == 0) {
    hdr->total = carry_s;
    hdr->valid = static_cast<int64_t>(valid_s);   // (the loop above ends with a barrier: every wave's add has landed)
  }
}

// ------------------------------------------------------------------ K3: compaction
struct CompactArgs {
  const uint8_t* values;  // pre-offset: element 0 of the logical array (unused for IOTA)
  Bits mask;
  Bits mvalid;
  Bits vvalid;
  int64_t length;
  int64_t ntiles;
  const uint32_t* tile_counts;
  const int64_t* group_excl;
  uint8_t* out_data;
  uint64_t* out_validity;  // may be NULL
  int emit_null;
  int invert;            // select clear mask bits (sort's null partition)
  int values_aligned16;  // values pointer is 16-byte aligned -> one dwordx4 load per lane
};

constexpr int kFlushBytes = 2048;  // flush unit, also the global alignment of full flushes

template <int RING>
struct __attribute__((aligned(16))) CompactLds {
  uint8_t ring[kWavesPerBlock][RING];
  uint64_t vbits[kWavesPerBlock][72];  // 4096 bits + 63 bits of misalignment -> 65 words (+pad)
};

template <int W>
struct ElemT;
template <> struct ElemT<1> { using type = uint8_t; };
template <> struct ElemT<2> { using type = uint16_t; };
template <> struct ElemT<4> { using type = uint32_t; };
template <> struct ElemT<8> { using type = uint64_t; };
template <> struct ElemT<16> { using type = uint4; };
// 32-byte values (decimal256, fixed_size_binary(32); PrimitiveFilterExec / FixedWidthTakeExec's widest case,
// vector_selection_filter_internal.cc:479-508): two 16-byte halves, moved by the gather forms only
struct Elem32 { uint4 lo, hi; };
template <> struct ElemT<32> { using type = Elem32; };

template <int W>
__device__ __forceinline__ typename ElemT<W>::type zero_elem() {
  if constexpr (W == 32) {
    return Elem32{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  } else if constexpr (W == 16) {
    return make_uint4(0, 0, 0, 0);
  } else {
    return 0;
  }
}

// Cache policy of the selection kernels' one-pass streams (compile-time: the buffer-load policy is an immediate).
// bit 0: value / index loads non-temporal, bit 1: output stores non-temporal, bit 2: take's gathers non-temporal.
// A/B per build: scripts/gpu_r03_sel_nt_ab.sh -> profiles/r03_b_selection_nt_ab.txt.
constexpr int kSelNt = 3;

template <int W>
__device__ __forceinline__ typename ElemT<W>::type sel_load(const typename ElemT<W>::type* p, bool nt) {
  using E = typename ElemT<W>::type;
  if constexpr (W == 32) {
    return *p;
  } else if constexpr (W == 16) {
    return nt ? nt_load16(p) : *p;
  } else {
    return nt ? nt_load<E>(p) : *p;
  }
}
template <int W>
__device__ __forceinline__ void sel_store(typename ElemT<W>::type* p, typename ElemT<W>::type v) {
  using E = typename ElemT<W>::type;
  if constexpr (W == 32) {
    *p = v;
  } else if constexpr (W == 16) {
    if constexpr ((kSelNt & 2) != 0) nt_store16(p, v); else *p = v;
  } else {
    if constexpr ((kSelNt & 2) != 0) nt_store<E>(p, v); else *p = v;
  }
}

__device__ __forceinline__ void wave_lds_sync() {
  // LDS traffic of one wave is processed in order; this only pins the compiler's ordering.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Flush logical ring bytes [F, F + 2048) to global; only bytes in [lo, hi) are owned by
// this wave.  Fully owned 16-byte granules go out as dwordx4 stores, the others element-wise.
template <int W, int RING>
__device__ __forceinline__ void flush_region(const uint8_t* ring, uint8_t* gbase, int64_t F,
                                             int64_t lo, int64_t hi, int lane) {
  using E = typename ElemT<W>::type;
  constexpr int R = 16 / W;
  const int ring_off = static_cast<int>(F & (RING - 1));
#pragma unroll
  for (int h = 0; h < kFlushBytes / (16 * kWave); ++h) {
    const int g = h * kWave + lane;  // granule within the region
    const int64_t gs = F + g * 16;
    const int64_t ge = gs + 16;
    if (ge <= lo || gs >= hi) continue;
    const uint8_t* src = ring + ring_off + g * 16;
    uint8_t* dst = gbase + gs;
    if (gs >= lo && ge <= hi) {
      sel_store<16>(reinterpret_cast<uint4*>(dst), *reinterpret_cast<const uint4*>(src));
    } else {
#pragma unroll
      for (int e = 0; e < R; ++e) {
        const int64_t es = gs + e * W;
        if (es >= lo && es + W <= hi) {
          *reinterpret_cast<E*>(dst + e * W) = *reinterpret_cast<const E*>(src + e * W);
        }
      }
    }
  }
}

// One batch of B wave-iterations: which rows of this lane's granules are emitted, their output
// ranks, and the (in-flight) granule loads.
template <int W, int B>
struct Batch {
  using E = typename ElemT<W>::type;
  static constexpr int R = 16 / W;
  uint32_t bits[B];
  uint32_t zbits[B];
  uint32_t rank[B];
  E v[B][R];
};

// Word `widx_base + q` of a per-lane 64-bit value, where q = lane / (64 / R) differs between lane
// groups: R wave-uniform v_readlane pairs + selects instead of a ds_bpermute round trip.
template <int R, bool EMIT>
__device__ __forceinline__ void select_words(uint64_t Ew, uint32_t p, uint64_t Zw,
                                             int widx_base, int lane, uint64_t* wE, uint32_t* wp,
                                             uint64_t* wZ) {
  constexpr int kLanesPerWord = 64 / R;
  const int q_mine = lane / kLanesPerWord;
  uint64_t e = 0, z = 0;
  uint32_t pp = 0;
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int src = widx_base + q;  // wave-uniform
    const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(Ew), src);
    const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(Ew >> 32), src);
    const uint32_t pq = __builtin_amdgcn_readlane(p, src);
    uint64_t zq = 0;
    if constexpr (EMIT) {
      const uint32_t zlo = __builtin_amdgcn_readlane(static_cast<uint32_t>(Zw), src);
      const uint32_t zhi = __builtin_amdgcn_readlane(static_cast<uint32_t>(Zw >> 32), src);
      zq = (static_cast<uint64_t>(zhi) << 32) | zlo;
    }
    if (q == q_mine) {
      e = (static_cast<uint64_t>(hi) << 32) | lo;
      pp = pq;
      z = zq;
    }
  }
  *wE = e;
  *wp = pp;
  *wZ = z;
}

struct TileCtx {
  uint64_t Ew, Zw;
  uint32_t p;
  int lane;
  int64_t tile_row0;
  const uint8_t* values;
  __amdgpu_buffer_rsrc_t rsrc;  // the 4096 values of this tile (aligned path)
};

// Issue phase: ranks + granule loads of iterations [j0, j0 + B)
template <int W, bool IOTA, bool EMIT, bool ALIGNED, int B>
__device__ __forceinline__ void issue_batch(const TileCtx& c, int j0, Batch<W, B>& b) {
  using E = typename ElemT<W>::type;
  constexpr int R = 16 / W;
  constexpr int kRowsPerIter = kWave * R;
#pragma unroll
  for (int i = 0; i < B; ++i) {
    const int j = j0 + i;
    const int row_in_tile = j * kRowsPerIter + c.lane * R;
    const int bpos = row_in_tile & 63;
    uint64_t wE, wZ;
    uint32_t wp;
    select_words<R, EMIT>(c.Ew, c.p, c.Zw, j * R, c.lane, &wE, &wp, &wZ);
    b.bits[i] = static_cast<uint32_t>(wE >> bpos) & ((1u << R) - 1u);
    b.zbits[i] = static_cast<uint32_t>(wZ >> bpos) & ((1u << R) - 1u);
    b.rank[i] = wp + __popcll(wE & low_mask64(bpos));
    if constexpr (IOTA) {
#pragma unroll
      for (int r = 0; r < R; ++r) b.v[i][r] = static_cast<E>(c.tile_row0 + row_in_tile + r);
    } else {
      const int64_t row = c.tile_row0 + row_in_tile;
      const uint8_t* src = c.values + row * static_cast<int64_t>(W);
      if constexpr (ALIGNED) {
        // One 16-byte buffer load per lane.  Lanes with no emitted row get an out-of-range offset:
        // the hardware returns 0 without a memory access, so 128-byte lines holding no selected
        // row are never fetched from HBM, and there is no branch to wait at.
        const uint32_t voff = b.bits[i] != 0 ? static_cast<uint32_t>(row_in_tile * W) : kBufferSkip;
        const uint4 q = (kSelNt & 1) ? buffer_load_b128_nt(c.rsrc, voff) : buffer_load_b128(c.rsrc, voff);
        if constexpr (W == 16) {
          b.v[i][0] = q;
        } else {
          const E* qe = reinterpret_cast<const E*>(&q);
#pragma unroll
          for (int r = 0; r < R; ++r) b.v[i][r] = qe[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          b.v[i][r] = zero_elem<W>();
          if ((b.bits[i] >> r) & 1u) {
            if constexpr (W == 16) {
              const uint64_t* s64 = reinterpret_cast<const uint64_t*>(src + r * W);
              const uint64_t lo64 = s64[0], hi64 = s64[1];
              b.v[i][r] = make_uint4(static_cast<uint32_t>(lo64), static_cast<uint32_t>(lo64 >> 32),
                                     static_cast<uint32_t>(hi64), static_cast<uint32_t>(hi64 >> 32));
            } else {
              b.v[i][r] = *reinterpret_cast<const E*>(src + r * W);
            }
          }
        }
      }
    }
  }
}

// Consume phase: emitted elements go to the LDS ring at their output rank
template <int W, int B, int RING>
__device__ __forceinline__ void consume_batch(const Batch<W, B>& b, uint8_t* ring, int64_t a0) {
  using E = typename ElemT<W>::type;
  constexpr int R = 16 / W;
#pragma unroll
  for (int i = 0; i < B; ++i) {
    if (b.bits[i] != 0) {
      uint32_t rk = b.rank[i];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if ((b.bits[i] >> r) & 1u) {
          E e = b.v[i][r];
          if ((b.zbits[i] >> r) & 1u) e = zero_elem<W>();
          const int ro = static_cast<int>((a0 + static_cast<int64_t>(rk) * W) & (RING - 1));
          *reinterpret_cast<E*>(ring + ro) = e;
          ++rk;
        }
      }
    }
  }
}

// W       : element width in bytes
// IOTA    : emit row numbers instead of loaded values (GetTakeIndices)
// EMIT    : FilterOptions::EMIT_NULL with a nullable mask (null mask slots emit zero-filled nulls)
// ALIGNED : the values pointer is 16-byte aligned (one buffer_load_dwordx4 per lane)
// B       : wave-iterations per batch (loads of a batch are issued back to back)
// PIPE    : software pipeline - the loads of batch k+1 are in flight while batch k is consumed
template <int W, bool IOTA, bool EMIT, bool ALIGNED, int B, bool PIPE>
__global__ __launch_bounds__(kBlock) void compact_kernel(CompactArgs a) {
  using E = typename ElemT<W>::type;
  constexpr int R = 16 / W;                // rows per lane per iteration (one 16-byte granule)
  constexpr int kRowsPerIter = kWave * R;  // rows per wave iteration
  constexpr int kIters = kTileRows / kRowsPerIter;
  constexpr int kWordsPerIter = kRowsPerIter / 64;  // == R
  constexpr int kBatches = kIters / B;
  constexpr int RING = (B <= 2) ? 4096 : 8192;  // pending < 2 KiB + B KiB
  static_assert(kIters % B == 0, "batch must divide the iteration count");
  static_assert(!PIPE || kBatches % 2 == 0, "pipelined form walks the batches in pairs");

  __shared__ CompactLds<RING> lds;

  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  if (t >= a.ntiles) return;  // wave-uniform; no workgroup barrier below

  // ---- per-lane mask words of this tile
  const int64_t w = t * 64 + lane;
  uint64_t mv;
  const uint64_t Ew = emit_word(a.mask, a.mvalid, w, EMIT, a.invert != 0, &mv);
  const uint32_t k = __popcll(Ew);
  const uint32_t incl = wave_inclusive_scan_u32(k);
  const uint32_t p = incl - k;  // emitted rows in this tile before this lane's word
  const uint32_t total = __shfl(incl, 63, 64);
  if (total == 0) return;

  // ---- output offset of this tile: group prefix + counts of earlier tiles in the group
  const int64_t grp = t >> 6;
  const int tin = static_cast<int>(t & 63);
  const uint32_t cprev = lane < tin ? a.tile_counts[grp * kTilesPerGroup + lane] : 0u;
  const int64_t off = a.group_excl[grp] + wave_reduce_sum_u32(cprev);

  // ---- LDS ring: logical byte x of the ring <-> global byte (gbase + x); gbase is 2 KiB aligned
  uint8_t* ring = lds.ring[wave];
  const int64_t a0 =
      static_cast<int64_t>((reinterpret_cast<uint64_t>(a.out_data) + static_cast<uint64_t>(off) * W) &
                           (kFlushBytes - 1));
  uint8_t* gbase = a.out_data + (off * static_cast<int64_t>(W) - a0);  // stays a global pointer
  int64_t flushed = 0;

  TileCtx c;
  c.Ew = Ew;
  c.Zw = EMIT ? (Ew & ~mv) : 0;  // rows emitted only because the mask slot is null: zero-filled
  c.p = p;
  c.lane = lane;
  c.tile_row0 = t * kTileRows;
  c.values = a.values;
  c.rsrc = make_rsrc(IOTA ? nullptr : a.values + c.tile_row0 * static_cast<int64_t>(W),
                     static_cast<uint32_t>(kTileRows * W));

  auto flush_check = [&](int next_batch) {
    // wave-uniform: elements emitted through the batches consumed so far
    const uint32_t cum = next_batch < kBatches
                             ? __builtin_amdgcn_readlane(p, next_batch * B * kWordsPerIter)
                             : total;
    const int64_t written = a0 + static_cast<int64_t>(cum) * W;
    if (written - flushed >= kFlushBytes) {
      wave_lds_sync();
      while (written - flushed >= kFlushBytes) {
        flush_region<W, RING>(ring, gbase, flushed, a0, written, lane);
        flushed += kFlushBytes;
      }
      wave_lds_sync();
    }
  };

  // GetTakeIndices at low selectivity: when the whole tile's output fits the ring, every lane
  // just walks the set bits of its own mask word (a handful of iterations) instead of the
  // 4096-row sweep below.
  bool tile_done = false;
  if constexpr (IOTA) {
    if (a0 + static_cast<int64_t>(total) * W <= RING) {  // wave-uniform
      uint64_t m = Ew;
      uint32_t rk = p;
      while (m != 0) {
        const int bpos = __ffsll(static_cast<unsigned long long>(m)) - 1;
        m &= m - 1;
        E val = static_cast<E>(c.tile_row0 + lane * 64 + bpos);
        if constexpr (EMIT) {
          if ((c.Zw >> bpos) & 1ull) val = 0;
        }
        *reinterpret_cast<E*>(ring + a0 + static_cast<int64_t>(rk) * W) = val;
        ++rk;
      }
      tile_done = true;
    }
  }

  if (tile_done) {
    // nothing else to stage
  } else if constexpr (PIPE) {
    // two batches of loads in flight; the last pair is peeled so that the loop body has no
    // conditional issue (the vmcnt bookkeeping stays exact: 2B loads outstanding at each wait)
    Batch<W, B> b0, b1;
    issue_batch<W, IOTA, EMIT, ALIGNED, B>(c, 0, b0);
    int jb = 0;
    for (; jb + 2 < kBatches; jb += 2) {
      issue_batch<W, IOTA, EMIT, ALIGNED, B>(c, (jb + 1) * B, b1);
      consume_batch<W, B, RING>(b0, ring, a0);
      flush_check(jb + 1);
      issue_batch<W, IOTA, EMIT, ALIGNED, B>(c, (jb + 2) * B, b0);
      consume_batch<W, B, RING>(b1, ring, a0);
      flush_check(jb + 2);
    }
    issue_batch<W, IOTA, EMIT, ALIGNED, B>(c, (jb + 1) * B, b1);
    consume_batch<W, B, RING>(b0, ring, a0);
    flush_check(jb + 1);
    consume_batch<W, B, RING>(b1, ring, a0);
    flush_check(jb + 2);
  } else {
    for (int jb = 0; jb < kBatches; ++jb) {
      Batch<W, B> b0;
      issue_batch<W, IOTA, EMIT, ALIGNED, B>(c, jb * B, b0);
      consume_batch<W, B, RING>(b0, ring, a0);
      flush_check(jb + 1);
    }
  }

  // ---- tail flush
  {
    const int64_t written = a0 + static_cast<int64_t>(total) * W;
    wave_lds_sync();
    while (flushed < written) {
      flush_region<W, RING>(ring, gbase, flushed, a0, written, lane);
      flushed += kFlushBytes;
    }
  }

  // ---- output validity: bits of (values_valid & mask_valid) at emitted rows, compacted
  if (a.out_validity != nullptr) {
    uint64_t* vb = lds.vbits[wave];
    vb[lane] = 0;
    if (lane < 8) vb[64 + lane] = 0;
    wave_lds_sync();

    const uint64_t vv = load_word(a.vvalid, w);
    const uint64_t Vs = vv & mv;  // DROP: emitted rows have mv = 1; EMIT_NULL: null mask -> null
    uint64_t cbits;
    if (__any((Ew & ~Vs) != 0)) {
      cbits = pext64(Vs, Ew);
    } else {
      cbits = low_mask64(static_cast<int>(k));
    }
    const int64_t obit = (off & 63) + p;  // bit position inside the tile-local LDS bitmap
    const int word = static_cast<int>(obit >> 6);
    const int sh = static_cast<int>(obit & 63);
    if (k != 0) {
      atomicOr(reinterpret_cast<unsigned long long*>(&vb[word]),
               static_cast<unsigned long long>(cbits << sh));
      if (sh != 0 && (sh + static_cast<int>(k)) > 64) {
        atomicOr(reinterpret_cast<unsigned long long*>(&vb[word + 1]),
                 static_cast<unsigned long long>(cbits >> (64 - sh)));
      }
    }
    wave_lds_sync();

    const int64_t gw0 = off >> 6;  // first global word touched
    const int64_t bit_lo = off;    // owned global bit range [bit_lo, bit_hi)
    const int64_t bit_hi = off + total;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = h * 64 + lane;
      if (j <= 64) {
        const int64_t wlo = (gw0 + j) << 6;
        const int64_t whi = wlo + 64;
        if (whi > bit_lo && wlo < bit_hi) {
          const uint64_t val = vb[j];
          if (wlo >= bit_lo && whi <= bit_hi) {
            a.out_validity[gw0 + j] = val;  // fully owned word
          } else if (val != 0) {
            atomicOr(reinterpret_cast<unsigned long long*>(&a.out_validity[gw0 + j]),
                     static_cast<unsigned long long>(val));  // shared with a neighbouring tile
          }
        }
      }
    }
  }
}


// ------------------------------------------------------------------ K3': sparse compaction
// The form used when few rows are selected (S <= N/4): instead of sweeping all 4096 rows of the
// tile, every lane walks the set bits of its own mask word and appends the row numbers to a
// per-wave LDS list at their output rank; the wave then gathers 64 selected rows per step —
// lane j loads values[row_j] (8 bytes) and stores out[off + s + j], a fully coalesced 512-byte
// run — so the instruction count scales with the rows emitted, not the rows scanned.  HBM
// traffic is the same as K3 (128-byte lines without a selected row are never touched).
// Steps are aligned to 64-bit words of the OUTPUT bitmap, so the output validity of a step is
// one __ballot: full words are stored, the tile's first/last partial words are OR-ed in.
constexpr int kSparseWindow = 1024;  // emitted rows staged per round (a tile emits ~410 at 10 %)
constexpr int kSparseU = 2;          // gather steps (64 rows each) in flight per wave: 2 is 3 % faster than 4 at 10 % selectivity, equal at 25 / 50 % (fewer registers, more waves; profiles/r03_q_filter_gather_steps_in_flight_ab.jsonl)

struct __attribute__((aligned(16))) SparseLds {
  uint16_t sel[kWavesPerBlock][kSparseWindow];  // row-in-tile of the s-th emitted row of the window
  uint64_t vs[kWavesPerBlock][64];              // values_valid & mask_valid words of the tile
  uint64_t zw[kWavesPerBlock][64];              // rows emitted only because the mask slot is null
};

// IOTA: emit row numbers instead of gathered values (GetTakeIndices)
template <int W, bool EMIT, bool IOTA>
__global__ __launch_bounds__(kBlock) void compact_sparse_kernel(CompactArgs a) {
  using E = typename ElemT<W>::type;
  constexpr int U = kSparseU;  // gather steps in flight
  __shared__ SparseLds lds;

  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  if (t >= a.ntiles) return;  // wave-uniform; no workgroup barrier below

  const int64_t w = t * 64 + lane;
  uint64_t mv;
  const uint64_t Ew = emit_word(a.mask, a.mvalid, w, EMIT, a.invert != 0, &mv);
  const uint32_t k = __popcll(Ew);
  const uint32_t incl = wave_inclusive_scan_u32(k);
  const uint32_t p = incl - k;
  const int total = static_cast<int>(__shfl(incl, 63, 64));
  if (total == 0) return;

  const int64_t grp = t >> 6;
  const int tin = static_cast<int>(t & 63);
  const uint32_t cprev = lane < tin ? a.tile_counts[grp * kTilesPerGroup + lane] : 0u;
  const int64_t off = a.group_excl[grp] + wave_reduce_sum_u32(cprev);

  uint16_t* sel = lds.sel[wave];
  const bool want_validity = a.out_validity != nullptr;
  if (want_validity) lds.vs[wave][lane] = load_word(a.vvalid, w) & mv;
  if constexpr (EMIT) lds.zw[wave][lane] = Ew & ~mv;

  const E* __restrict__ values = IOTA ? nullptr : reinterpret_cast<const E*>(a.values) + t * kTileRows;
  E* __restrict__ out = reinterpret_cast<E*>(a.out_data);
  const int head = static_cast<int>(off & 63);  // the first step starts `head` bits into a word
  const int nsteps = (head + total + 63) >> 6;
  const int64_t word0 = off >> 6;
  constexpr int kStepsPerWindow = kSparseWindow / 64;

  // Emitted rows are staged window by window (kSparseWindow rows = kStepsPerWindow output words);
  // window r covers the steps [r * 16, r * 16 + 16), i.e. the emitted rows
  // [r * 1024 - head, (r + 1) * 1024 - head).  One window is enough up to 25 % selectivity.
  uint64_t m = Ew;   // bits of this lane's word not yet listed
  uint32_t rk = p;   // output rank (inside the tile) of the next unlisted bit
  for (int step0 = 0; step0 < nsteps; step0 += kStepsPerWindow) {
    const int win_lo = step0 * 64 - head;          // first emitted row of the window (may be < 0)
    const int win_hi = win_lo + kSparseWindow;     // one past the last
    // ---- 1. list the rows of this window
    while (m != 0 && static_cast<int>(rk) < win_hi) {
      const int bpos = __ffsll(static_cast<unsigned long long>(m)) - 1;
      m &= m - 1;
      sel[static_cast<int>(rk) - win_lo] = static_cast<uint16_t>(lane * 64 + bpos);
      ++rk;
    }
    wave_lds_sync();
    // ---- 2. gather, 64 emitted rows per step, steps aligned to output bitmap words
    const int step_end = (step0 + kStepsPerWindow) < nsteps ? (step0 + kStepsPerWindow) : nsteps;
    const int first_listed = win_lo < 0 ? 0 : win_lo;  // a row the wave lists anyway
    for (int i0 = step0; i0 < step_end; i0 += U) {
      int s[U];
      bool act[U];
      int r[U];
      E v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s[u] = (i0 + u) * 64 + lane - head;  // emitted-row number inside the tile
        act[u] = s[u] >= 0 && s[u] < total && (i0 + u) < step_end;
        r[u] = sel[(act[u] ? s[u] : first_listed) - win_lo];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (IOTA) {
          if constexpr (W <= 8) v[u] = static_cast<E>(t * kTileRows + r[u]);
        } else {
          v[u] = sel_load<W>(values + r[u], (kSelNt & 1) != 0);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if ((i0 + u) >= step_end) break;  // wave-uniform
        E e = v[u];
        if constexpr (EMIT) {
          if ((lds.zw[wave][r[u] >> 6] >> (r[u] & 63)) & 1ull) e = zero_elem<W>();
        }
        if (act[u]) sel_store<W>(out + off + s[u], e);
        if (want_validity) {
          const bool vbit = act[u] && ((lds.vs[wave][r[u] >> 6] >> (r[u] & 63)) & 1ull);
          const uint64_t bal = __ballot(vbit);
          const uint64_t owned = __ballot(act[u]);
          if (lane == 0) {
            uint64_t* dst = a.out_validity + word0 + i0 + u;
            if (owned == ~uint64_t(0)) {
              *dst = bal;  // the whole output word belongs to this tile
            } else if (bal != 0) {
              atomicOr(reinterpret_cast<unsigned long long*>(dst), static_cast<unsigned long long>(bal));
            }
          }
        }
      }
    }
    wave_lds_sync();  // the next window overwrites the list
  }
}

// ------------------------------------------------------------------ expand (the inverse of a DROP filter)
// out[r] = mask[r] ? dense[rank(r)] : 0, where rank(r) = number of set mask bits before r: what a
// Parquet reader does when it spreads the non-null values of an optional column over their slots
// (DefLevelsToBitmap + the "spaced" decode, cpp/src/parquet/level_conversion.cc, decoder.cc).
// Same tiling and the same count / scan workspace as the compaction; one wave per 4096-row tile,
// word-by-word so that stores are dense and the gathers from `dense` monotonic.
template <int W>
__global__ __launch_bounds__(kBlock) void expand_kernel(CompactArgs a) {
  using E = typename ElemT<W>::type;
  __shared__ uint64_t words[kWavesPerBlock][64];
  __shared__ uint32_t before[kWavesPerBlock][64];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  if (t >= a.ntiles) return;  // wave-uniform; no workgroup barrier below
  uint64_t mv;
  const uint64_t mw = emit_word(a.mask, a.mvalid, t * 64 + lane, false, false, &mv);
  const uint32_t k = __popcll(mw);
  const uint32_t incl = wave_inclusive_scan_u32(k);
  words[wave][lane] = mw;
  before[wave][lane] = incl - k;
  const int64_t grp = t >> 6;
  const int tin = static_cast<int>(t & 63);
  const uint32_t cprev = lane < tin ? a.tile_counts[grp * kTilesPerGroup + lane] : 0u;
  const int64_t off = a.group_excl[grp] + wave_reduce_sum_u32(cprev);
  wave_lds_sync();
  const E* __restrict__ dense = reinterpret_cast<const E*>(a.values);
  E* __restrict__ out = reinterpret_cast<E*>(a.out_data);
  const int64_t row0 = t * kTileRows;
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll 4
  for (int j = 0; j < 64; ++j) {
    const int64_t row = row0 + j * 64 + lane;
    if (row >= a.length) break;  // lanes drop out only past the end of the array
    const uint64_t wj = words[wave][j];
    E e = zero_elem<W>();
    if ((wj >> lane) & 1ull) e = dense[off + before[wave][j] + __popcll(wj & below)];
    out[row] = e;
  }
}

// ------------------------------------------------------------------ run-end-encoded boolean masks
// array_filter also registers run_end_encoded<boolean> filters (vector_selection_filter_internal.cc:1090, 1115-;
// VisitPlainxREEFilterOutputSegments, vector_selection_internal.cc:79-153: every run contributes its value's
// (valid, selected) pair to all its rows).  The device form expands the runs into the plain mask layout — one bitmap
// of selection bits, one of validity bits — and hands it to the ordinary compaction: same output by construction.
// One lane per 64-row output word: binary search for the run holding the word's first row, then a walk over the runs.
template <typename R>
__global__ __launch_bounds__(kBlock) void ree_bool_expand_kernel(const R* __restrict__ run_ends, int64_t num_runs,
                                                                 Bits vsel, Bits vvalid, int64_t logical_offset,
                                                                 int64_t length, uint64_t* __restrict__ out_bits,
                                                                 uint64_t* __restrict__ out_valid) {
  const int64_t nwords = (length + 63) >> 6;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < nwords; w += stride) {
    const int64_t row0 = w << 6;
    const int64_t rows = length - row0 < 64 ? length - row0 : 64;
    const int64_t p0 = logical_offset + row0;          // physical position of the word's first row
    int64_t lo = 0, hi = num_runs;                      // first run with run_end > p0
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (static_cast<int64_t>(run_ends[mid]) > p0) hi = mid; else lo = mid + 1;
    }
    uint64_t bits = 0, valid = 0;
    int64_t done = 0;
    for (int64_t r = lo; r < num_runs && done < rows; ++r) {
      const int64_t end = static_cast<int64_t>(run_ends[r]) - p0;     // exclusive, relative to the word
      const int64_t upto = end < rows ? end : rows;
      if (upto > done) {
        const uint64_t span = (upto - done >= 64 ? ~uint64_t(0) : ((uint64_t(1) << (upto - done)) - 1)) << done;
        if ((load_word(vsel, r >> 6) >> (r & 63)) & 1ull) bits |= span;
        if ((load_word(vvalid, r >> 6) >> (r & 63)) & 1ull) valid |= span;
        done = upto;
      }
    }
    out_bits[w] = bits;
    if (out_valid != nullptr) out_valid[w] = valid;
  }
}

// ------------------------------------------------------------------ take
struct TakeArgs {
  const uint8_t* values;   // pre-offset to element 0
  Bits vvalid_unused;      // (kept for symmetry; source validity is probed per bit below)
  const uint8_t* src_valid_bytes;  // source validity bitmap bytes or NULL
  int64_t src_valid_offset;        // bit offset of element 0 in src_valid_bytes
  const uint8_t* indices;          // pre-offset to element 0
  Bits ivalid;                     // index validity (logical)
  int64_t length;                  // number of indices
  uint8_t* out_data;
  uint64_t* out_validity;          // may be NULL
  unsigned long long* valid_count; // may be NULL
};

// Every load below is unconditional: lanes that must not gather (row past the end, null index)
// read a harmless in-bounds address instead — their own output slot — and the result is
// discarded.  With no branches the compiler issues the U index loads, then the U value gathers
// and the U source-validity probes back to back: 2 dependent round trips per 64*U rows (the
// value of a null source slot is loaded and dropped — it is addressable memory).
template <int W, typename IdxT, bool HAS_IV, bool HAS_SV>
__global__ __launch_bounds__(kBlock) void take_kernel(TakeArgs a) {
  using E = typename ElemT<W>::type;
  constexpr int U = 8;  // independent gathers in flight per lane
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock +
                         __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nchunks = (a.length + 64 * U - 1) / (64 * U);
  const int64_t last = a.length - 1;
  const IdxT* __restrict__ indices = reinterpret_cast<const IdxT*>(a.indices);
  const E* __restrict__ values = reinterpret_cast<const E*>(a.values);
  E* __restrict__ out = reinterpret_cast<E*>(a.out_data);
  uint64_t nvalid = 0;
  for (int64_t c = wave_g; c < nchunks; c += nwaves) {
    const int64_t base = c * (64 * U);
    int64_t pos[U];
    uint64_t idx[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = base + u * 64 + lane;
      pos[u] = p <= last ? p : last;  // clamped: always a readable/writable slot
      idx[u] = static_cast<uint64_t>((kSelNt & 1) ? nt_load<IdxT>(indices + pos[u]) : indices[pos[u]]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok[u] = (base + u * 64 + lane) <= last;
      if constexpr (HAS_IV) {
        const uint64_t wbits = load_word_nb(a.ivalid, (base >> 6) + u);
        ok[u] = ok[u] && ((wbits >> lane) & 1ull);
      }
    }
    E val[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const E* src = ok[u] ? (values + idx[u]) : (out + pos[u]);
      val[u] = sel_load<W>(src, (kSelNt & 4) != 0);
    }
    if constexpr (HAS_SV) {
      uint8_t vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t bit = ok[u] ? static_cast<uint64_t>(a.src_valid_offset) + idx[u] : 0;
        vb[u] = a.src_valid_bytes[bit >> 3];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t bit = static_cast<uint64_t>(a.src_valid_offset) + idx[u];
        ok[u] = ok[u] && ((vb[u] >> (bit & 7)) & 1);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = base + u * 64 + lane;
      if (p <= last) sel_store<W>(out + p, ok[u] ? val[u] : zero_elem<W>());
      const uint64_t vbal = __ballot(ok[u]);
      nvalid += __popcll(vbal);
      if (a.out_validity != nullptr && lane == 0 && (base + u * 64) <= last) {
        a.out_validity[(base >> 6) + u] = vbal;
      }
    }
  }
  if (a.valid_count != nullptr && lane == 0 && nvalid != 0) atomicAdd(a.valid_count, nvalid);
}

// Take of ROWS of row_bytes contiguous bytes — a fixed_size_list whose nested values are fixed-width and free of nulls, or
// a fixed_size_binary of any width: what FSLTakeExec hands to FixedWidthTakeExec (vector_selection_internal.cc:991-1003;
// util::IsFixedWidthLike) for ANY byte width, where the element kernels above stop at 32.  A wave owns 64 output rows: one
// index (+ validity probe) per lane, then the rows' units of UB bytes (the largest power of two <= 16 that divides
// row_bytes) are copied FLATTENED — lane t of step k copies unit (k * 64 + t) of the 64 * units-per-row units, taking its
// row's source index from that row's lane — so the stores are one contiguous stream whatever the row size and a short row
// (12 bytes) keeps 64 lanes busy; a null row is zero-filled (WriteZero, gather_internal.h:114-153).
template <int UB> struct RowUnit;
template <> struct RowUnit<1> { typedef uint8_t type; };
template <> struct RowUnit<2> { typedef uint16_t type; };
template <> struct RowUnit<4> { typedef uint32_t type; };
template <> struct RowUnit<8> { typedef uint64_t type; };
template <> struct RowUnit<16> { typedef arx_u32x4 type; };

template <int UB, typename IdxT, bool HAS_IV, bool HAS_SV>
__global__ __launch_bounds__(kBlock) void take_rows_kernel(TakeArgs a, uint32_t units_per_row) {
  using E = typename RowUnit<UB>::type;
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nchunks = (a.length + 63) >> 6;
  const int64_t last = a.length - 1;
  const IdxT* __restrict__ indices = reinterpret_cast<const IdxT*>(a.indices);
  const E* __restrict__ values = reinterpret_cast<const E*>(a.values);
  E* __restrict__ out = reinterpret_cast<E*>(a.out_data);
  uint64_t nvalid = 0;
  for (int64_t c = wave_g; c < nchunks; c += nwaves) {
    const int64_t base = c << 6;
    const int64_t p = base + lane <= last ? base + lane : last;
    uint64_t idx = static_cast<uint64_t>(indices[p]);
    bool ok = base + lane <= last;
    if constexpr (HAS_IV) ok = ok && ((load_word_nb(a.ivalid, base >> 6) >> lane) & 1ull);
    if constexpr (HAS_SV) {
      const uint64_t bit = ok ? static_cast<uint64_t>(a.src_valid_offset) + idx : 0;
      ok = ok && ((a.src_valid_bytes[bit >> 3] >> (bit & 7)) & 1);
    }
    if (!ok) idx = 0;
    const uint64_t vbal = __ballot(ok);
    nvalid += __popcll(vbal);
    if (a.out_validity != nullptr && lane == 0) a.out_validity[base >> 6] = vbal;
    const int rows_here = static_cast<int>(last - base + 1 < 64 ? last - base + 1 : 64);
    const uint32_t total = static_cast<uint32_t>(rows_here) * units_per_row;
    const uint32_t idx_lo = static_cast<uint32_t>(idx), idx_hi = static_cast<uint32_t>(idx >> 32);
    for (uint32_t t = lane; t < ((total + 63u) & ~63u); t += 64) {   // (wave-uniform trip count: the shuffles below are whole)
      const uint32_t tt = t < total ? t : total - 1;
      const uint32_t row = tt / units_per_row;
      const uint32_t j = tt - row * units_per_row;
      const uint64_t src = (static_cast<uint64_t>(__shfl(idx_hi, static_cast<int>(row), 64)) << 32) | __shfl(idx_lo, static_cast<int>(row), 64);
      const bool row_ok = (vbal >> row) & 1ull;
      // (a null / out-of-range row reads nothing: the values may be an EMPTY child with no buffer at all — ADVICE r4)
      E v = E{};
      if (row_ok) v = values[src * units_per_row + j];
      if (t < total) out[(static_cast<uint64_t>(base) + row) * units_per_row + j] = v;
    }
  }
  if (a.valid_count != nullptr && lane == 0 && nvalid != 0) atomicAdd(a.valid_count, nvalid);
}

// Take of SEVERAL fixed-width columns by the same indices in one launch (TakeRAR, vector_selection_take_internal.cc:
// 619-633, runs TakeAAA per column: every column re-reads the indices and their validity; here a wave reads them once
// per 256 rows and gathers column after column).  Same per-column result as take_kernel.
constexpr int kTakeMaxCols = 16;
struct TakeColsArgs {
  const uint8_t* values[kTakeMaxCols];          // pre-offset to element 0
  const uint8_t* src_valid_bytes[kTakeMaxCols]; // source validity bitmap bytes or NULL
  int64_t src_valid_offset[kTakeMaxCols];
  uint8_t* out_data[kTakeMaxCols];
  uint64_t* out_validity[kTakeMaxCols];         // may be NULL (then neither the column nor the indices have nulls)
  int width[kTakeMaxCols];
  int num_cols;
  const uint8_t* indices;                       // pre-offset to element 0
  Bits ivalid;
  int64_t length;
  unsigned long long* valid_counts;             // [num_cols] or NULL
};

template <int W, int U>
__device__ __forceinline__ uint64_t take_one_column(const TakeColsArgs& a, int c, const int64_t (&pos)[U],
                                                    const uint64_t (&idx)[U], const bool (&ok0)[U], int64_t base,
                                                    int64_t last, int lane) {
  using E = typename ElemT<W>::type;
  const E* __restrict__ values = reinterpret_cast<const E*>(a.values[c]);
  E* __restrict__ out = reinterpret_cast<E*>(a.out_data[c]);
  const uint8_t* __restrict__ svb = a.src_valid_bytes[c];
  const uint64_t svo = static_cast<uint64_t>(a.src_valid_offset[c]);
  E val[U];
  bool ok[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const E* src = ok0[u] ? (values + idx[u]) : (out + pos[u]);   // (a harmless in-bounds address when not gathering)
    val[u] = *src;
    ok[u] = ok0[u];
  }
  if (svb != nullptr) {   // wave-uniform
    uint8_t vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) vb[u] = svb[(ok0[u] ? svo + idx[u] : 0) >> 3];
#pragma unroll
    for (int u = 0; u < U; ++u) ok[u] = ok[u] && ((vb[u] >> ((svo + idx[u]) & 7)) & 1);
  }
  uint64_t nvalid = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t p = base + u * 64 + lane;
    if (p <= last) out[p] = ok[u] ? val[u] : zero_elem<W>();
    const uint64_t vbal = __ballot(ok[u]);
    nvalid += __popcll(vbal);
    if (a.out_validity[c] != nullptr && lane == 0 && (base + u * 64) <= last) a.out_validity[c][(base >> 6) + u] = vbal;
  }
  return nvalid;
}

template <typename IdxT, bool HAS_IV>
__global__ __launch_bounds__(kBlock) void take_columns_kernel(TakeColsArgs a) {
  constexpr int U = 4;
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock +
                         __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nchunks = (a.length + 64 * U - 1) / (64 * U);
  const int64_t last = a.length - 1;
  const IdxT* __restrict__ indices = reinterpret_cast<const IdxT*>(a.indices);
  uint64_t nvalid[kTakeMaxCols];
  for (int c = 0; c < kTakeMaxCols; ++c) nvalid[c] = 0;
  for (int64_t ch = wave_g; ch < nchunks; ch += nwaves) {
    const int64_t base = ch * (64 * U);
    int64_t pos[U];
    uint64_t idx[U];
    bool ok0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = base + u * 64 + lane;
      pos[u] = p <= last ? p : last;
      idx[u] = static_cast<uint64_t>(indices[pos[u]]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok0[u] = (base + u * 64 + lane) <= last;
      if constexpr (HAS_IV) {
        const uint64_t wbits = load_word_nb(a.ivalid, (base >> 6) + u);
        ok0[u] = ok0[u] && ((wbits >> lane) & 1ull);
      }
    }
    for (int c = 0; c < a.num_cols; ++c) {   // wave-uniform
      uint64_t nv;
      switch (a.width[c]) {
        case 1: nv = take_one_column<1, U>(a, c, pos, idx, ok0, base, last, lane); break;
        case 2: nv = take_one_column<2, U>(a, c, pos, idx, ok0, base, last, lane); break;
        case 4: nv = take_one_column<4, U>(a, c, pos, idx, ok0, base, last, lane); break;
        case 8: nv = take_one_column<8, U>(a, c, pos, idx, ok0, base, last, lane); break;
        case 16: nv = take_one_column<16, U>(a, c, pos, idx, ok0, base, last, lane); break;
        default: nv = take_one_column<32, U>(a, c, pos, idx, ok0, base, last, lane); break;
      }
      nvalid[c] += nv;
    }
  }
  if (a.valid_counts != nullptr && lane == 0) {
    for (int c = 0; c < a.num_cols; ++c) {
      if (nvalid[c] != 0) atomicAdd(a.valid_counts + c, nvalid[c]);
    }
  }
}

// Take on BOOLEAN values (bit-packed): out bit i = value bit idx[i], 0 for a null slot
// (Gather</*kValueWidthInBits=*/1>, gather_internal.h; the bit twin of take_kernel).  One output word
// per wave step: the value bits and the validity are both packed by ballot.
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void take_bits_kernel(TakeArgs a, int64_t value_bit_offset) {
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock +
                         __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwords = (a.length + 63) >> 6;
  const int64_t last = a.length - 1;
  const IdxT* __restrict__ indices = reinterpret_cast<const IdxT*>(a.indices);
  uint64_t* __restrict__ out = reinterpret_cast<uint64_t*>(a.out_data);
  uint64_t nvalid = 0;
  for (int64_t w = wave_g; w < nwords; w += nwaves) {
    const int64_t p = (w << 6) + lane;
    const int64_t pc = p <= last ? p : last;
    const uint64_t idx = static_cast<uint64_t>(indices[pc]);
    bool ok = p <= last && ((load_word(a.ivalid, w) >> lane) & 1ull);
    if (ok && a.src_valid_bytes != nullptr) {
      const uint64_t vb = static_cast<uint64_t>(a.src_valid_offset) + idx;
      ok = (a.src_valid_bytes[vb >> 3] >> (vb & 7)) & 1;
    }
    bool bit = false;
    if (ok) {
      const uint64_t b = static_cast<uint64_t>(value_bit_offset) + idx;
      bit = (a.values[b >> 3] >> (b & 7)) & 1;
    }
    const uint64_t data_bal = __ballot(bit);
    const uint64_t valid_bal = __ballot(ok);
    nvalid += __popcll(valid_bal);
    if (lane == 0) {
      out[w] = data_bal;
      if (a.out_validity != nullptr) a.out_validity[w] = valid_bal;
    }
  }
  if (a.valid_count != nullptr && lane == 0 && nvalid != 0) atomicAdd(a.valid_count, nvalid);
}

// ------------------------------------------------------------------ bounds check
struct BoundsWs {
  unsigned long long first_bad_pos;  // min position of an offending index, ~0 if none
  long long bad_value_signed;
  unsigned long long bad_value_unsigned;
  int64_t pad[5];
};

template <typename IdxT>
__global__ __launch_bounds__(kBlock) void bounds_kernel(const uint8_t* indices, Bits ivalid,
                                                        int64_t length, uint64_t upper,
                                                        BoundsWs* ws) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  unsigned long long bad = ~0ull;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < length;
       i += stride) {
    const IdxT v = reinterpret_cast<const IdxT*>(indices)[i];
    const bool valid = (load_word(ivalid, i >> 6) >> (i & 63)) & 1ull;
    bool oob;
    if constexpr (std::is_signed<IdxT>::value) {
      oob = v < 0 || static_cast<uint64_t>(v) >= upper;
    } else {
      oob = static_cast<uint64_t>(v) >= upper;
    }
    if (valid && oob) {
      bad = static_cast<unsigned long long>(i);
      break;  // positions increase along the grid-stride loop
    }
  }
  if (bad != ~0ull) atomicMin(&ws->first_bad_pos, bad);
}

template <typename IdxT>
__global__ void bounds_fetch_kernel(const uint8_t* indices, BoundsWs* ws) {
  const unsigned long long pos = ws->first_bad_pos;
  if (pos == ~0ull) return;
  const IdxT v = reinterpret_cast<const IdxT*>(indices)[pos];
  ws->bad_value_signed = static_cast<long long>(v);
  ws->bad_value_unsigned = static_cast<unsigned long long>(v);
}

// ------------------------------------------------------------------ host side
static int check_mask(const ArxSpan* mask, int null_selection) {
  if (mask == nullptr) {
    set_error("filter mask is NULL");
    return ARX_INVALID;
  }
  if (mask->length < 0 || mask->offset < 0) {
    set_error("negative length/offset");
    return ARX_INVALID;
  }
  if (mask->length > 0 && mask->data == nullptr) {
    set_error("filter mask has no data buffer");
    return ARX_INVALID;
  }
  if (null_selection != ARX_FILTER_DROP && null_selection != ARX_FILTER_EMIT_NULL) {
    set_error("bad null_selection %d", null_selection);
    return ARX_INVALID;
  }
  return ARX_OK;
}

// filter.MayHaveNulls() (cpp/src/arrow/array/data.h): null_count != 0 && validity != NULL
static inline const void* effective_validity(const ArxSpan* s) {
  return (s->null_count != 0) ? s->validity : nullptr;
}

static int launch_count(const ArxSpan* mask, int null_selection, void* ws, size_t ws_bytes,
                        hipStream_t st, int invert = 0, const ArxSpan* values = nullptr) {
  const int rc = check_mask(mask, null_selection);
  if (rc != ARX_OK) return rc;
  const size_t need = arx_filter_workspace_bytes(mask->length);
  if (ws == nullptr || ws_bytes < need) {
    set_error("filter workspace too small: %zu < %zu", ws_bytes, need);
    return ARX_INVALID;
  }
  if ((reinterpret_cast<uint64_t>(ws) & 63) != 0) {
    set_error("filter workspace must be 64-byte aligned");
    return ARX_INVALID;
  }
  FilterWsView v = ws_view(ws, mask->length);
  const int64_t nt = num_tiles(mask->length);
  const int64_t ng = num_groups(mask->length);
  FilterWsHeader h{};
  h.total = 0;
  h.ntiles = nt;
  h.ngroups = ng;
  h.length = mask->length;
  ARX_HIP(hipMemcpyAsync(v.hdr, &h, sizeof(h), hipMemcpyHostToDevice, st));
  if (mask->length == 0) return ARX_OK;
  const Bits mb = make_bits(mask->data, mask->offset, mask->length);
  const Bits mvb = make_bits(effective_validity(mask), mask->offset, mask->length);
  if (values != nullptr) {
    const Bits vvb = make_bits(effective_validity(values), values->offset, mask->length);
    hipLaunchKernelGGL((count_kernel<true>), dim3(static_cast<unsigned>(ng)), dim3(kBlock), 0, st, mb, mvb, vvb,
                       null_selection, invert, nt, v.tile_counts, v.group_total, v.group_valid);
    ARX_CHECK_LAUNCH("count_kernel");
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, v.group_total, ng, v.group_excl, v.hdr,
                       static_cast<const int64_t*>(v.group_valid));
  } else {
    hipLaunchKernelGGL((count_kernel<false>), dim3(static_cast<unsigned>(ng)), dim3(kBlock), 0, st, mb, mvb, mvb,
                       null_selection, invert, nt, v.tile_counts, v.group_total, static_cast<int64_t*>(nullptr));
    ARX_CHECK_LAUNCH("count_kernel");
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, v.group_total, ng, v.group_excl, v.hdr,
                       static_cast<const int64_t*>(nullptr));
  }
  ARX_CHECK_LAUNCH("scan_kernel");
  return ARX_OK;
}

// Tuning knobs (arx_set_option): filter_batch in {1,4}, filter_pipe in {0,1}.  Results never change.
static Knob<int> g_filter_batch{4};
static Knob<int> g_filter_pipe{1};
static Knob<int> g_filter_sparse{-1};  // -1 = auto (see launch_compact), 0 = never, 1 = always

template <int W, bool IOTA, bool EMIT, bool ALIGNED>
static void launch_compact_e(const CompactArgs& a, unsigned grid, hipStream_t st) {
  constexpr int kIters = kTileRows / (kWave * (16 / W));
  constexpr int kBig = kIters >= 8 ? 4 : 2;  // batch size of the pipelined form (even #batches)
  if (ALIGNED && g_filter_batch >= 4 && g_filter_pipe) {
    hipLaunchKernelGGL((compact_kernel<W, IOTA, EMIT, ALIGNED, kBig, true>), dim3(grid),
                       dim3(kBlock), 0, st, a);
  } else {
    // the plain form: one iteration at a time (also the only form of the unaligned path)
    hipLaunchKernelGGL((compact_kernel<W, IOTA, EMIT, ALIGNED, 1, false>), dim3(grid), dim3(kBlock),
                       0, st, a);
  }
}

template <int W, bool IOTA>
static void launch_compact_w(const CompactArgs& a, unsigned grid, hipStream_t st) {
  const bool aligned = IOTA || a.values_aligned16 != 0;
  if (a.emit_null) {
    if (aligned) launch_compact_e<W, IOTA, true, true>(a, grid, st);
    else if constexpr (!IOTA) launch_compact_e<W, IOTA, true, false>(a, grid, st);
  } else {
    if (aligned) launch_compact_e<W, IOTA, false, true>(a, grid, st);
    else if constexpr (!IOTA) launch_compact_e<W, IOTA, false, false>(a, grid, st);
  }
}

template <int W, bool IOTA = false>
static void launch_sparse_w(const CompactArgs& a, unsigned grid, hipStream_t st) {
  if (a.emit_null) {
    hipLaunchKernelGGL((compact_sparse_kernel<W, true, IOTA>), dim3(grid), dim3(kBlock), 0, st, a);
  } else {
    hipLaunchKernelGGL((compact_sparse_kernel<W, false, IOTA>), dim3(grid), dim3(kBlock), 0, st, a);
  }
}

// out_length < 0: unknown -> the sweeping form
static int launch_compact(bool iota, int W, const CompactArgs& a, hipStream_t st,
                          int64_t out_length = -1) {
  const unsigned grid = static_cast<unsigned>(ceil_div(a.ntiles, kWavesPerBlock));
  // Measured on MI355X (profiles/r01_c_filter_selectivity_sweep.txt): for 8- and 16-byte values the
  // gather form wins at EVERY selectivity (1.94 vs 2.46 ms at 25 %, 2.38 vs 2.84 ms at 50 %, equal
  // at 100 %); narrower values keep the sweeping form above 25 % (its 16-byte granules carry
  // 4-16 rows per lane, the gather form would issue 1-4 byte accesses).
  const bool sparse = !iota && !a.invert &&
                      (g_filter_sparse == 1 ||
                       (g_filter_sparse < 0 &&
                        (W >= 8 || (out_length >= 0 && out_length * 4 <= a.length))));
  // GetTakeIndices: the gather form (row numbers instead of gathered values) unless the bitmap is
  // being inverted for the sort's null partition
  if (iota && !a.invert && g_filter_sparse != 0 && (W == 2 || W == 4 || W == 8)) {
    if (W == 2) launch_sparse_w<2, true>(a, grid, st);
    else if (W == 4) launch_sparse_w<4, true>(a, grid, st);
    else launch_sparse_w<8, true>(a, grid, st);
    ARX_CHECK_LAUNCH("compact_sparse_kernel<IOTA>");
    return ARX_OK;
  }
  if (sparse) {
    switch (W) {
      case 1: launch_sparse_w<1>(a, grid, st); break;
      case 2: launch_sparse_w<2>(a, grid, st); break;
      case 4: launch_sparse_w<4>(a, grid, st); break;
      case 8: launch_sparse_w<8>(a, grid, st); break;
      case 16: launch_sparse_w<16>(a, grid, st); break;
      case 32: launch_sparse_w<32>(a, grid, st); break;
      default:
        set_error("unsupported byte width %d for the gfx950 filter", W);
        return ARX_NOT_IMPLEMENTED;
    }
    ARX_CHECK_LAUNCH("compact_sparse_kernel");
    return ARX_OK;
  }
  if (iota) {
    switch (W) {
      case 2: launch_compact_w<2, true>(a, grid, st); break;
      case 4: launch_compact_w<4, true>(a, grid, st); break;
      default:
        set_error("unsupported index width %d", W);
        return ARX_NOT_IMPLEMENTED;
    }
  } else {
    switch (W) {
      case 1: launch_compact_w<1, false>(a, grid, st); break;
      case 2: launch_compact_w<2, false>(a, grid, st); break;
      case 4: launch_compact_w<4, false>(a, grid, st); break;
      case 8: launch_compact_w<8, false>(a, grid, st); break;
      case 16: launch_compact_w<16, false>(a, grid, st); break;
      default:
        // PrimitiveFilterExec also handles 1-bit and "any width" values; those stay on the CPU
        set_error("unsupported byte width %d for the gfx950 filter", W);
        return ARX_NOT_IMPLEMENTED;
    }
  }
  ARX_CHECK_LAUNCH("compact_kernel");
  return ARX_OK;
}

int set_selection_option(const char* name, int64_t value) {
  if (strcmp(name, "filter_batch") == 0) {
    g_filter_batch = value >= 4 ? 4 : 1;
    return 1;
  }
  if (strcmp(name, "filter_pipe") == 0) {
    g_filter_pipe = value != 0;
    return 1;
  }
  if (strcmp(name, "filter_sparse") == 0) {
    g_filter_sparse = value < 0 ? -1 : (value != 0);
    return 1;
  }
  return 0;
}

// Zero the ceil(S/64) words of an output bitmap (tile-boundary words are OR-ed in).
static int zero_out_validity(void* out_validity, int64_t out_length, hipStream_t st) {
  if (out_validity == nullptr) return ARX_OK;
  if ((reinterpret_cast<uint64_t>(out_validity) & 7) != 0) {
    set_error("out_validity must be 8-byte aligned");
    return ARX_INVALID;
  }
  if (out_length < 0) {
    set_error("out_length (from arx_filter_count) is required when out_validity is given");
    return ARX_INVALID;
  }
  if (out_length > 0) {
    ARX_HIP(hipMemsetAsync(out_validity, 0, static_cast<size_t>(ceil_div(out_length, 64)) * 8, st));
  }
  return ARX_OK;
}

size_t selection_workspace_bytes(int64_t length) { return arx_filter_workspace_bytes(length); }

// Ascending row numbers (uint32) of the set (invert: clear) bits of bitmap[bit_offset, +length).
// Synchronous: *out_count_host receives the number of rows written.
int selection_bit_positions(const void* bitmap, int64_t bit_offset, int64_t length, bool invert,
                            void* ws, size_t ws_bytes, uint32_t* out, int64_t* out_count_host,
                            hipStream_t st) {
  ArxSpan m{};
  m.validity = nullptr;
  m.data = bitmap;
  m.offset = bit_offset;
  m.length = length;
  m.null_count = 0;
  int rc = launch_count(&m, ARX_FILTER_DROP, ws, ws_bytes, st, invert ? 1 : 0);
  if (rc != ARX_OK) return rc;
  int64_t total = 0;
  ARX_HIP(hipMemcpyAsync(&total, ws, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  *out_count_host = total;
  if (total == 0) return ARX_OK;
  FilterWsView v = ws_view(ws, length);
  CompactArgs a{};
  a.values = nullptr;
  a.mask = make_bits(bitmap, bit_offset, length);
  a.mvalid = make_bits(nullptr, 0, length);
  a.vvalid = make_bits(nullptr, 0, length);
  a.length = length;
  a.ntiles = num_tiles(length);
  a.tile_counts = v.tile_counts;
  a.group_excl = v.group_excl;
  a.out_data = reinterpret_cast<uint8_t*>(out);
  a.out_validity = nullptr;
  a.emit_null = 0;
  a.invert = invert ? 1 : 0;
  a.values_aligned16 = 1;
  return launch_compact(true, 4, a, st);
}

}  // namespace arx

using namespace arx;

extern "C" {

size_t arx_filter_workspace_bytes(int64_t length) {
  if (length < 0) length = 0;
  const int64_t ng = num_groups(length);
  return static_cast<size_t>(64 + ng * 16 + ng * kTilesPerGroup * 4 + ng * 8 + 64);
}

int arx_filter_count_async(const ArxSpan* mask, int null_selection, void* ws, size_t ws_bytes,
                           void* stream) {
  return launch_count(mask, null_selection, ws, ws_bytes, as_stream(stream));
}

int arx_filter_count_nulls(const ArxSpan* values, const ArxSpan* mask, int null_selection, void* ws, size_t ws_bytes,
                           int64_t* out_length, int64_t* out_null_count, void* stream) {
  if (out_length == nullptr || out_null_count == nullptr || values == nullptr) {
    set_error("values / out_length / out_null_count is NULL");
    return ARX_INVALID;
  }
  if (mask != nullptr && values->length != mask->length) {
    set_error("filter values and mask differ in length (%lld vs %lld)", static_cast<long long>(values->length),
              static_cast<long long>(mask->length));
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  const int rc = launch_count(mask, null_selection, ws, ws_bytes, st, 0, values);
  if (rc != ARX_OK) return rc;
  int64_t hdr[5] = {0, 0, 0, 0, 0};   // FilterWsHeader: total, ntiles, ngroups, length, valid
  ARX_HIP(hipMemcpyAsync(hdr, ws, sizeof(hdr), hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  *out_length = hdr[0];
  *out_null_count = mask->length == 0 ? 0 : hdr[0] - hdr[4];
  return ARX_OK;
}

int arx_filter_count(const ArxSpan* mask, int null_selection, void* ws, size_t ws_bytes,
                     int64_t* out_length, void* stream) {
  if (out_length == nullptr) {
    set_error("out_length is NULL");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  const int rc = launch_count(mask, null_selection, ws, ws_bytes, st);
  if (rc != ARX_OK) return rc;
  int64_t total = 0;
  ARX_HIP(hipMemcpyAsync(&total, ws, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  *out_length = total;
  return ARX_OK;
}

int arx_filter_exec(const ArxSpan* values, int byte_width, const ArxSpan* mask, int null_selection,
                    const void* ws, int64_t out_length, void* out_data, void* out_validity,
                    void* stream) {
  int rc = check_mask(mask, null_selection);
  if (rc != ARX_OK) return rc;
  if (values == nullptr || ws == nullptr) {
    set_error("values/ws is NULL");
    return ARX_INVALID;
  }
  // ExecSpanIterator::Init rejects mismatched lengths (cpp/src/arrow/compute/exec.cc:349-355)
  if (values->length != mask->length) {
    set_error("Array arguments must all be the same length (values %lld vs filter %lld)",
              static_cast<long long>(values->length), static_cast<long long>(mask->length));
    return ARX_INVALID;
  }
  if (mask->length == 0) return ARX_OK;
  if (values->data == nullptr || out_data == nullptr) {
    set_error("values/out data buffer is NULL");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  FilterWsView v = ws_view(const_cast<void*>(ws), mask->length);
  CompactArgs a{};
  a.values = static_cast<const uint8_t*>(values->data) + values->offset * byte_width;
  a.mask = make_bits(mask->data, mask->offset, mask->length);
  a.mvalid = make_bits(effective_validity(mask), mask->offset, mask->length);
  a.vvalid = make_bits(effective_validity(values), values->offset, values->length);
  a.length = mask->length;
  a.ntiles = num_tiles(mask->length);
  a.tile_counts = v.tile_counts;
  a.group_excl = v.group_excl;
  a.out_data = static_cast<uint8_t*>(out_data);
  a.out_validity = static_cast<uint64_t*>(out_validity);
  a.emit_null = null_selection == ARX_FILTER_EMIT_NULL && a.mvalid.base != nullptr;
  a.values_aligned16 = (reinterpret_cast<uint64_t>(a.values) & 15) == 0;
  if ((reinterpret_cast<uint64_t>(out_data) % byte_width) != 0) {
    set_error("out_data is not aligned to the element width");
    return ARX_INVALID;
  }
  rc = zero_out_validity(out_validity, out_length, st);
  if (rc != ARX_OK) return rc;
  return launch_compact(false, byte_width, a, st, out_length);
}

int arx_expand_by_mask(const void* dense, int byte_width, const ArxSpan* mask, const void* ws, void* out_data,
                       void* stream) {
  int rc = check_mask(mask, ARX_FILTER_DROP);
  if (rc != ARX_OK) return rc;
  if (ws == nullptr) {
    set_error("ws is NULL (run arx_filter_count on the mask first)");
    return ARX_INVALID;
  }
  if (mask->length == 0) return ARX_OK;
  if (out_data == nullptr || (reinterpret_cast<uint64_t>(out_data) % byte_width) != 0 ||
      (dense != nullptr && (reinterpret_cast<uint64_t>(dense) % byte_width) != 0)) {
    set_error("expand: NULL or misaligned buffer");
    return ARX_INVALID;
  }
  FilterWsView v = ws_view(const_cast<void*>(ws), mask->length);
  CompactArgs a{};
  a.values = static_cast<const uint8_t*>(dense);
  a.mask = make_bits(mask->data, mask->offset, mask->length);
  a.mvalid = make_bits(effective_validity(mask), mask->offset, mask->length);
  a.length = mask->length;
  a.ntiles = num_tiles(mask->length);
  a.tile_counts = v.tile_counts;
  a.group_excl = v.group_excl;
  a.out_data = static_cast<uint8_t*>(out_data);
  const unsigned grid = static_cast<unsigned>(ceil_div(a.ntiles, kWavesPerBlock));
  hipStream_t st = as_stream(stream);
  switch (byte_width) {
    case 1: hipLaunchKernelGGL((expand_kernel<1>), dim3(grid), dim3(kBlock), 0, st, a); break;
    case 2: hipLaunchKernelGGL((expand_kernel<2>), dim3(grid), dim3(kBlock), 0, st, a); break;
    case 4: hipLaunchKernelGGL((expand_kernel<4>), dim3(grid), dim3(kBlock), 0, st, a); break;
    case 8: hipLaunchKernelGGL((expand_kernel<8>), dim3(grid), dim3(kBlock), 0, st, a); break;
    case 16: hipLaunchKernelGGL((expand_kernel<16>), dim3(grid), dim3(kBlock), 0, st, a); break;
    default:
      set_error("unsupported byte width %d for expand", byte_width);
      return ARX_NOT_IMPLEMENTED;
  }
  ARX_CHECK_LAUNCH("expand_kernel");
  return ARX_OK;
}

int arx_mask_to_indices(const ArxSpan* mask, int null_selection, const void* ws, int64_t out_length,
                        int index_width, void* out_indices, void* out_validity, void* stream) {
  int rc = check_mask(mask, null_selection);
  if (rc != ARX_OK) return rc;
  if (ws == nullptr) {
    set_error("ws is NULL");
    return ARX_INVALID;
  }
  if (index_width != 2 && index_width != 4 && index_width != 8) {
    set_error("index_width must be 2, 4 or 8");
    return ARX_INVALID;
  }
  // 8: the uint64 row numbers indices_nonzero returns (DoNonZero, kernels/vector_selection.cc:228-300), written by the
  // gather form directly instead of widened from uint32 in a second pass
  if (index_width == 8 && (g_filter_sparse == 0 || null_selection == ARX_FILTER_EMIT_NULL)) {
    set_error("index_width 8 is served by the gather form with DROP only");
    return ARX_NOT_IMPLEMENTED;
  }
  // GetTakeIndicesFromBitmap: uint16 up to 65535 rows, uint32 up to UINT32_MAX, else
  // NotImplemented (vector_selection_take_internal.cc:258-272)
  if (mask->length > static_cast<int64_t>(UINT32_MAX)) {
    set_error("Filter length exceeds UINT32_MAX, consider a different strategy for selecting elements");
    return ARX_NOT_IMPLEMENTED;
  }
  if (index_width == 2 && mask->length > 65535) {
    set_error("index_width 2 requires length <= 65535");
    return ARX_INVALID;
  }
  if (mask->length == 0) return ARX_OK;
  if (out_indices == nullptr) {
    set_error("out_indices is NULL");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  FilterWsView v = ws_view(const_cast<void*>(ws), mask->length);
  CompactArgs a{};
  a.values = nullptr;
  a.mask = make_bits(mask->data, mask->offset, mask->length);
  a.mvalid = make_bits(effective_validity(mask), mask->offset, mask->length);
  a.vvalid = make_bits(nullptr, 0, mask->length);
  a.length = mask->length;
  a.ntiles = num_tiles(mask->length);
  a.tile_counts = v.tile_counts;
  a.group_excl = v.group_excl;
  a.out_data = static_cast<uint8_t*>(out_indices);
  a.out_validity = static_cast<uint64_t*>(out_validity);
  a.emit_null = null_selection == ARX_FILTER_EMIT_NULL && a.mvalid.base != nullptr;
  a.values_aligned16 = 1;
  const bool emit = a.emit_null != 0;
  if (emit && out_validity == nullptr) {
    set_error("EMIT_NULL with a nullable mask needs out_validity");
    return ARX_INVALID;
  }
  rc = zero_out_validity(out_validity, out_length, st);
  if (rc != ARX_OK) return rc;
  return launch_compact(true, index_width, a, st);
}

size_t arx_take_workspace_bytes(void) { return sizeof(BoundsWs); }

int arx_check_index_bounds(const ArxSpan* indices, int index_type, uint64_t upper_limit, void* ws,
                           size_t ws_bytes, void* stream) {
  if (indices == nullptr || ws == nullptr || ws_bytes < sizeof(BoundsWs)) {
    set_error("bad arguments to arx_check_index_bounds");
    return ARX_INVALID;
  }
  if (indices->length == 0) return ARX_OK;
  hipStream_t st = as_stream(stream);
  static const int widths[8] = {1, 1, 2, 2, 4, 4, 8, 8};
  if (index_type < 0 || index_type > 7) {
    set_error("Invalid index type for boundschecking");
    return ARX_INVALID;
  }
  const int iw = widths[index_type];
  const uint8_t* idx = static_cast<const uint8_t*>(indices->data) + indices->offset * iw;
  const Bits iv = make_bits(effective_validity(indices), indices->offset, indices->length);
  BoundsWs* bws = static_cast<BoundsWs*>(ws);
  BoundsWs init{};
  init.first_bad_pos = ~0ull;
  ARX_HIP(hipMemcpyAsync(bws, &init, sizeof(init), hipMemcpyHostToDevice, st));
  const int64_t blocks = std::min<int64_t>(ceil_div(indices->length, kBlock), 256 * 8);
  const dim3 grid(static_cast<unsigned>(blocks)), block(kBlock);
#define ARX_BOUNDS(T)                                                                        \
  hipLaunchKernelGGL((bounds_kernel<T>), grid, block, 0, st, idx, iv, indices->length,       \
                     upper_limit, bws);                                                      \
  hipLaunchKernelGGL((bounds_fetch_kernel<T>), dim3(1), dim3(1), 0, st, idx, bws)
  switch (index_type) {
    case ARX_UINT8: ARX_BOUNDS(uint8_t); break;
    case ARX_INT8: ARX_BOUNDS(int8_t); break;
    case ARX_UINT16: ARX_BOUNDS(uint16_t); break;
    case ARX_INT16: ARX_BOUNDS(int16_t); break;
    case ARX_UINT32: ARX_BOUNDS(uint32_t); break;
    case ARX_INT32: ARX_BOUNDS(int32_t); break;
    case ARX_UINT64: ARX_BOUNDS(uint64_t); break;
    default: ARX_BOUNDS(int64_t); break;
  }
#undef ARX_BOUNDS
  ARX_CHECK_LAUNCH("bounds_kernel");
  BoundsWs res{};
  ARX_HIP(hipMemcpyAsync(&res, bws, sizeof(res), hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  if (res.first_bad_pos != ~0ull) {
    const bool is_signed = (index_type & 1) != 0;
    if (is_signed) {
      set_error("Index %lld out of bounds", res.bad_value_signed);
    } else {
      set_error("Index %llu out of bounds", res.bad_value_unsigned);
    }
    return ARX_INDEX_ERROR;
  }
  return ARX_OK;
}

int arx_take(const ArxSpan* values, int byte_width, const ArxSpan* indices, int index_type,
             void* out_data, void* out_validity, int64_t* valid_count, void* stream) {
  if (values == nullptr || indices == nullptr) {
    set_error("values/indices is NULL");
    return ARX_INVALID;
  }
  if (index_type < 0 || index_type > 7) {
    set_error("Unsupported index type %d for take", index_type);
    return ARX_NOT_IMPLEMENTED;
  }
  if (indices->length == 0) return ARX_OK;
  if (out_data == nullptr || indices->data == nullptr) {
    set_error("indices/out data buffer is NULL");
    return ARX_INVALID;
  }
  static const int widths[8] = {1, 1, 2, 2, 4, 4, 8, 8};
  const int iw = widths[index_type];
  hipStream_t st = as_stream(stream);
  TakeArgs a{};
  a.values = static_cast<const uint8_t*>(values->data) + values->offset * byte_width;
  a.src_valid_bytes = static_cast<const uint8_t*>(effective_validity(values));
  a.src_valid_offset = values->offset;
  a.indices = static_cast<const uint8_t*>(indices->data) + indices->offset * iw;
  a.ivalid = make_bits(effective_validity(indices), indices->offset, indices->length);
  a.length = indices->length;
  a.out_data = static_cast<uint8_t*>(out_data);
  a.out_validity = static_cast<uint64_t*>(out_validity);
  a.valid_count = reinterpret_cast<unsigned long long*>(valid_count);
  const bool needs_validity = a.src_valid_bytes != nullptr || a.ivalid.base != nullptr;
  if (needs_validity && out_validity == nullptr) {
    set_error("take: inputs may have nulls but out_validity is NULL");
    return ARX_INVALID;
  }
  const int64_t nchunks = ceil_div(indices->length, 512);
  // a persistent grid: the gathers are latency-bound (index load -> dependent gather), and a one-shot grid — the form
  // that lifts the streaming kernels of scalar.hip — is 1.7x SLOWER here (2.63 vs 1.58 ms at 1e8 monotonic indices,
  // 19.4 vs 6.1 ms for the 4-column form; profiles/r03_b_selection_nt_ab.txt)
  const int64_t blocks = std::min<int64_t>(ceil_div(nchunks, kWavesPerBlock), 256 * 32);
  const dim3 grid(static_cast<unsigned>(blocks)), block(kBlock);
  const bool has_iv = a.ivalid.base != nullptr;
  const bool has_sv = a.src_valid_bytes != nullptr;
#define ARX_TAKE_V(WW, IT)                                                                     \
  do {                                                                                         \
    if (has_iv && has_sv) hipLaunchKernelGGL((take_kernel<WW, IT, true, true>), grid, block, 0, st, a);        \
    else if (has_iv) hipLaunchKernelGGL((take_kernel<WW, IT, true, false>), grid, block, 0, st, a);            \
    else if (has_sv) hipLaunchKernelGGL((take_kernel<WW, IT, false, true>), grid, block, 0, st, a);            \
    else hipLaunchKernelGGL((take_kernel<WW, IT, false, false>), grid, block, 0, st, a);                       \
  } while (0)
#define ARX_TAKE_W(WW)                      \
  switch (iw) {                             \
    case 1: ARX_TAKE_V(WW, uint8_t); break; \
    case 2: ARX_TAKE_V(WW, uint16_t); break;\
    case 4: ARX_TAKE_V(WW, uint32_t); break;\
    default: ARX_TAKE_V(WW, uint64_t); break;\
  }
  switch (byte_width) {
    case 1: ARX_TAKE_W(1); break;
    case 2: ARX_TAKE_W(2); break;
    case 4: ARX_TAKE_W(4); break;
    case 8: ARX_TAKE_W(8); break;
    case 16: ARX_TAKE_W(16); break;
    case 32: ARX_TAKE_W(32); break;
    default:
      set_error("Unsupported primitive type for take: byte width %d", byte_width);
      return ARX_NOT_IMPLEMENTED;
  }
#undef ARX_TAKE_W
#undef ARX_TAKE_V
  ARX_CHECK_LAUNCH("take_kernel");
  return ARX_OK;
}

int arx_take_rows(const ArxSpan* values, int64_t row_bytes, const ArxSpan* indices, int index_type, void* out_data,
                  void* out_validity, int64_t* valid_count, void* stream) {
  if (values == nullptr || indices == nullptr) {
    set_error("values/indices is NULL");
    return ARX_INVALID;
  }
  if (index_type < 0 || index_type > 7) {
    set_error("Unsupported index type %d for take", index_type);
    return ARX_NOT_IMPLEMENTED;
  }
  if (row_bytes <= 0 || row_bytes > (int64_t(1) << 24)) {   // (64 rows x units-per-row is counted in 32 bits)
    set_error("arx_take_rows: rows of %lld bytes (1 .. 2^24)", static_cast<long long>(row_bytes));
    return row_bytes < 0 ? ARX_INVALID : ARX_NOT_IMPLEMENTED;
  }
  if (indices->length == 0) return ARX_OK;
  if (out_data == nullptr || indices->data == nullptr) {
    set_error("indices/out data buffer is NULL");
    return ARX_INVALID;
  }
  static const int widths[8] = {1, 1, 2, 2, 4, 4, 8, 8};
  const int iw = widths[index_type];
  TakeArgs a{};
  a.values = static_cast<const uint8_t*>(values->data) + values->offset * row_bytes;
  a.src_valid_bytes = static_cast<const uint8_t*>(effective_validity(values));
  a.src_valid_offset = values->offset;
  a.indices = static_cast<const uint8_t*>(indices->data) + indices->offset * iw;
  a.ivalid = make_bits(effective_validity(indices), indices->offset, indices->length);
  a.length = indices->length;
  a.out_data = static_cast<uint8_t*>(out_data);
  a.out_validity = static_cast<uint64_t*>(out_validity);
  a.valid_count = reinterpret_cast<unsigned long long*>(valid_count);
  const bool has_iv = a.ivalid.base != nullptr, has_sv = a.src_valid_bytes != nullptr;
  if ((has_iv || has_sv) && out_validity == nullptr) {
    set_error("take: inputs may have nulls but out_validity is NULL");
    return ARX_INVALID;
  }
  // the copy unit: the largest power of two <= 16 that divides the row and both base addresses
  int ub = 16;
  while (ub > 1 && (row_bytes % ub != 0 || reinterpret_cast<uintptr_t>(a.values) % ub != 0 || reinterpret_cast<uintptr_t>(out_data) % ub != 0)) ub >>= 1;
  const uint32_t upr = static_cast<uint32_t>(row_bytes / ub);
  const int64_t nchunks = ceil_div(indices->length, 64);
  const int64_t blocks = std::min<int64_t>(ceil_div(nchunks, kWavesPerBlock), 256 * 32);
  const dim3 grid(static_cast<unsigned>(blocks)), block(kBlock);
  hipStream_t st = as_stream(stream);
#define ARX_ROWS_V(UB, IT)                                                                                        \
  do {                                                                                                            \
    if (has_iv && has_sv) hipLaunchKernelGGL((take_rows_kernel<UB, IT, true, true>), grid, block, 0, st, a, upr);  \
    else if (has_iv) hipLaunchKernelGGL((take_rows_kernel<UB, IT, true, false>), grid, block, 0, st, a, upr);      \
    else if (has_sv) hipLaunchKernelGGL((take_rows_kernel<UB, IT, false, true>), grid, block, 0, st, a, upr);      \
    else hipLaunchKernelGGL((take_rows_kernel<UB, IT, false, false>), grid, block, 0, st, a, upr);                 \
  } while (0)
#define ARX_ROWS_U(UB)                        \
  switch (iw) {                               \
    case 1: ARX_ROWS_V(UB, uint8_t); break;   \
    case 2: ARX_ROWS_V(UB, uint16_t); break;  \
    case 4: ARX_ROWS_V(UB, uint32_t); break;  \
    default: ARX_ROWS_V(UB, uint64_t); break; \
  }
  switch (ub) {
    case 16: ARX_ROWS_U(16); break;
    case 8: ARX_ROWS_U(8); break;
    case 4: ARX_ROWS_U(4); break;
    case 2: ARX_ROWS_U(2); break;
    default: ARX_ROWS_U(1); break;
  }
#undef ARX_ROWS_U
#undef ARX_ROWS_V
  ARX_CHECK_LAUNCH("take_rows_kernel");
  return ARX_OK;
}

int arx_take_columns(const ArxSpan* columns, const int32_t* byte_widths, int num_columns, const ArxSpan* indices,
                     int index_type, void* const* out_data, void* const* out_validity, int64_t* valid_counts,
                     void* stream) {
  if (columns == nullptr || byte_widths == nullptr || indices == nullptr || out_data == nullptr || out_validity == nullptr) {
    set_error("take_columns: NULL argument");
    return ARX_INVALID;
  }
  if (num_columns < 1 || num_columns > kTakeMaxCols) {
    set_error("take_columns: 1 to %d columns per call (got %d)", kTakeMaxCols, num_columns);
    return ARX_INVALID;
  }
  if (index_type < 0 || index_type > 7) {
    set_error("Unsupported index type %d for take", index_type);
    return ARX_NOT_IMPLEMENTED;
  }
  if (indices->length == 0) return ARX_OK;
  if (indices->data == nullptr) {
    set_error("indices data buffer is NULL");
    return ARX_INVALID;
  }
  static const int widths[8] = {1, 1, 2, 2, 4, 4, 8, 8};
  const int iw = widths[index_type];
  TakeColsArgs a{};
  a.num_cols = num_columns;
  a.indices = static_cast<const uint8_t*>(indices->data) + indices->offset * iw;
  a.ivalid = make_bits(effective_validity(indices), indices->offset, indices->length);
  a.length = indices->length;
  a.valid_counts = reinterpret_cast<unsigned long long*>(valid_counts);
  for (int c = 0; c < num_columns; ++c) {
    const int w = byte_widths[c];
    if (w != 1 && w != 2 && w != 4 && w != 8 && w != 16 && w != 32) {
      set_error("Unsupported primitive type for take: byte width %d", w);
      return ARX_NOT_IMPLEMENTED;
    }
    if (out_data[c] == nullptr) {
      set_error("take_columns: out data buffer of column %d is NULL", c);
      return ARX_INVALID;
    }
    a.width[c] = w;
    a.values[c] = static_cast<const uint8_t*>(columns[c].data) + columns[c].offset * w;
    a.src_valid_bytes[c] = static_cast<const uint8_t*>(effective_validity(&columns[c]));
    a.src_valid_offset[c] = columns[c].offset;
    a.out_data[c] = static_cast<uint8_t*>(out_data[c]);
    a.out_validity[c] = static_cast<uint64_t*>(out_validity[c]);
    if ((a.src_valid_bytes[c] != nullptr || a.ivalid.base != nullptr) && out_validity[c] == nullptr) {
      set_error("take: inputs may have nulls but out_validity is NULL");
      return ARX_INVALID;
    }
  }
  hipStream_t st = as_stream(stream);
  const int64_t nchunks = ceil_div(indices->length, 256);
  const int64_t blocks = std::min<int64_t>(ceil_div(nchunks, kWavesPerBlock), 256 * 32);
  const dim3 grid(static_cast<unsigned>(blocks)), block(kBlock);
  const bool has_iv = a.ivalid.base != nullptr;
#define ARX_TAKEC(IT)                                                                                   \
  do {                                                                                                  \
    if (has_iv) hipLaunchKernelGGL((take_columns_kernel<IT, true>), grid, block, 0, st, a);             \
    else hipLaunchKernelGGL((take_columns_kernel<IT, false>), grid, block, 0, st, a);                   \
  } while (0)
  switch (iw) {
    case 1: ARX_TAKEC(uint8_t); break;
    case 2: ARX_TAKEC(uint16_t); break;
    case 4: ARX_TAKEC(uint32_t); break;
    default: ARX_TAKEC(uint64_t); break;
  }
#undef ARX_TAKEC
  ARX_CHECK_LAUNCH("take_columns_kernel");
  return ARX_OK;
}

int arx_take_bits(const ArxSpan* values, const ArxSpan* indices, int index_type, void* out_bits, void* out_validity,
                  int64_t* valid_count, void* stream) {
  if (values == nullptr || indices == nullptr) {
    set_error("values/indices is NULL");
    return ARX_INVALID;
  }
  if (index_type < 0 || index_type > 7) {
    set_error("Unsupported index type %d for take", index_type);
    return ARX_NOT_IMPLEMENTED;
  }
  if (indices->length == 0) return ARX_OK;
  if (out_bits == nullptr || indices->data == nullptr || values->data == nullptr) {
    set_error("values/indices/out buffer is NULL");
    return ARX_INVALID;
  }
  static const int widths[8] = {1, 1, 2, 2, 4, 4, 8, 8};
  const int iw = widths[index_type];
  TakeArgs a{};
  a.values = static_cast<const uint8_t*>(values->data);
  a.src_valid_bytes = static_cast<const uint8_t*>(effective_validity(values));
  a.src_valid_offset = values->offset;
  a.indices = static_cast<const uint8_t*>(indices->data) + indices->offset * iw;
  a.ivalid = make_bits(effective_validity(indices), indices->offset, indices->length);
  a.length = indices->length;
  a.out_data = static_cast<uint8_t*>(out_bits);
  a.out_validity = static_cast<uint64_t*>(out_validity);
  a.valid_count = reinterpret_cast<unsigned long long*>(valid_count);
  if ((a.src_valid_bytes != nullptr || a.ivalid.base != nullptr) && out_validity == nullptr) {
    set_error("take: inputs may have nulls but out_validity is NULL");
    return ARX_INVALID;
  }
  const int64_t nwords = ceil_div(indices->length, 64);
  const dim3 grid(static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(nwords, kWavesPerBlock), 256 * 32))));
  const dim3 block(kBlock);
  hipStream_t st = as_stream(stream);
  switch (iw) {
    case 1: hipLaunchKernelGGL((take_bits_kernel<uint8_t>), grid, block, 0, st, a, values->offset); break;
    case 2: hipLaunchKernelGGL((take_bits_kernel<uint16_t>), grid, block, 0, st, a, values->offset); break;
    case 4: hipLaunchKernelGGL((take_bits_kernel<uint32_t>), grid, block, 0, st, a, values->offset); break;
    default: hipLaunchKernelGGL((take_bits_kernel<uint64_t>), grid, block, 0, st, a, values->offset); break;
  }
  ARX_CHECK_LAUNCH("take_bits_kernel");
  return ARX_OK;
}

int arx_ree_bool_expand(const void* run_ends, int run_end_width, int64_t num_runs, const ArxSpan* values,
                        int64_t logical_offset, int64_t length, void* out_bits, void* out_validity, void* stream) {
  if (length < 0 || num_runs < 0 || logical_offset < 0 || values == nullptr ||
      (run_end_width != 2 && run_end_width != 4 && run_end_width != 8)) {
    set_error("bad arguments to arx_ree_bool_expand");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  if (run_ends == nullptr || values->data == nullptr || out_bits == nullptr || num_runs == 0 ||
      values->length < num_runs) {
    set_error("arx_ree_bool_expand: NULL buffer or fewer run values than run ends");
    return ARX_INVALID;
  }
  const Bits vsel = make_bits(values->data, values->offset, num_runs);
  const Bits vvalid = make_bits(values->null_count != 0 ? values->validity : nullptr, values->offset, num_runs);
  const int64_t nwords = ceil_div(length, 64);
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(nwords, kBlock), 2048)));
  hipStream_t st = as_stream(stream);
  uint64_t* ob = static_cast<uint64_t*>(out_bits);
  uint64_t* ov = static_cast<uint64_t*>(out_validity);
  if (run_end_width == 2) {
    hipLaunchKernelGGL((ree_bool_expand_kernel<int16_t>), dim3(grid), dim3(kBlock), 0, st, static_cast<const int16_t*>(run_ends),
                       num_runs, vsel, vvalid, logical_offset, length, ob, ov);
  } else if (run_end_width == 4) {
    hipLaunchKernelGGL((ree_bool_expand_kernel<int32_t>), dim3(grid), dim3(kBlock), 0, st, static_cast<const int32_t*>(run_ends),
                       num_runs, vsel, vvalid, logical_offset, length, ob, ov);
  } else {
    hipLaunchKernelGGL((ree_bool_expand_kernel<int64_t>), dim3(grid), dim3(kBlock), 0, st, static_cast<const int64_t*>(run_ends),
                       num_runs, vsel, vvalid, logical_offset, length, ob, ov);
  }
  ARX_CHECK_LAUNCH("ree_bool_expand_kernel");
  return ARX_OK;
}

}  // extern "C"
