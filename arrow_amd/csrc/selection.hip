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
  int64_t pad[4];
};
static_assert(sizeof(FilterWsHeader) == 64, "header is one 64-byte line");

struct FilterWsView {
  FilterWsHeader* hdr;
  int64_t* group_excl;    // [ngroups]
  int64_t* group_total;   // [ngroups]
  uint32_t* tile_counts;  // [ngroups * 64]
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
__global__ __launch_bounds__(kBlock) void count_kernel(Bits mask, Bits mvalid, int emit_null,
                                                       int invert, int64_t ntiles,
                                                       uint32_t* tile_counts,
                                                       int64_t* group_total) {
  __shared__ uint32_t wave_sums[kWavesPerBlock];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t group = blockIdx.x;
  const int64_t t0 = group * kTilesPerGroup + wave * 16;
  uint32_t mine = 0;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int64_t t = t0 + i;
    uint32_t k = 0;
    if (t < ntiles) {
      uint64_t mv;
      k = __popcll(emit_word(mask, mvalid, t * 64 + lane, emit_null != 0, invert != 0, &mv));
    }
    const uint32_t s = wave_reduce_sum_u32(k);
    if (lane == i) mine = s;
  }
  if (lane < 16) tile_counts[t0 + lane] = mine;  // tile_counts is padded to ngroups*64
  const uint32_t wsum = wave_reduce_sum_u32(lane < 16 ? mine : 0u);
  if (lane == 0) wave_sums[wave] = wsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t s = 0;
    for (int i = 0; i < kWavesPerBlock; ++i) s += wave_sums[i];
    group_total[group] = s;
  }
}

// ------------------------------------------------------------------ K2: scan of group totals
__global__ __launch_bounds__(1024) void scan_kernel(const int64_t* group_total, int64_t ngroups,
                                                    int64_t* group_excl, FilterWsHeader* hdr) {
  __shared__ int64_t wave_tot[16];
  __shared__ int64_t carry_s;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
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
  if (tid == 0) hdr->total = carry_s;
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

template <int W>
__device__ __forceinline__ typename ElemT<W>::type zero_elem() {
  if constexpr (W == 16) {
    return make_uint4(0, 0, 0, 0);
  } else {
    return 0;
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
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
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

// W     : element width in bytes
// IOTA  : emit row numbers instead of loaded values (GetTakeIndices)
// B     : iterations whose loads are issued back to back before any is consumed
// DENSE : load every granule (pure stream) instead of only granules holding an emitted row
template <int W, bool IOTA, int B, bool DENSE>
__global__ __launch_bounds__(kBlock) void compact_kernel(CompactArgs a) {
  using E = typename ElemT<W>::type;
  constexpr int R = 16 / W;                // rows per lane per iteration (one 16-byte granule)
  constexpr int kRowsPerIter = kWave * R;  // rows per wave iteration
  constexpr int kIters = kTileRows / kRowsPerIter;
  constexpr int kWordsPerIter = kRowsPerIter / 64;  // == R
  constexpr int RING = (B <= 2) ? 4096 : 8192;      // pending < 2 KiB + B KiB
  static_assert(kIters % B == 0, "batch must divide the iteration count");

  __shared__ CompactLds<RING> lds;

  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  if (t >= a.ntiles) return;  // wave-uniform; no workgroup barrier below
  const bool emit = a.emit_null != 0;

  // ---- per-lane mask words of this tile
  const int64_t w = t * 64 + lane;
  uint64_t mv;
  const uint64_t Ew = emit_word(a.mask, a.mvalid, w, emit, a.invert != 0, &mv);
  const uint32_t k = __popcll(Ew);
  const uint32_t incl = wave_inclusive_scan_u32(k);
  const uint32_t p = incl - k;  // emitted rows in this tile before this lane's word
  const uint32_t total = __shfl(incl, 63, 64);
  if (total == 0) return;

  // rows emitted only because the mask slot is null: zero-filled (WriteNull)
  const uint64_t Zw = emit ? (Ew & ~mv) : 0;

  // ---- output offset of this tile: group prefix + counts of earlier tiles in the group
  const int64_t grp = t >> 6;
  const int tin = static_cast<int>(t & 63);
  const uint32_t cprev = lane < tin ? a.tile_counts[grp * kTilesPerGroup + lane] : 0u;
  const int64_t off = a.group_excl[grp] + wave_reduce_sum_u32(cprev);

  // ---- LDS ring: logical byte x of the ring <-> global byte (gbase + x)
  uint8_t* ring = lds.ring[wave];
  const uint64_t G0 = reinterpret_cast<uint64_t>(a.out_data) + static_cast<uint64_t>(off) * W;
  const int64_t a0 = static_cast<int64_t>(G0 & (kFlushBytes - 1));
  uint8_t* gbase = reinterpret_cast<uint8_t*>(G0 - a0);
  int64_t flushed = 0;

  const int64_t tile_row0 = t * kTileRows;

  for (int j0 = 0; j0 < kIters; j0 += B) {
    uint32_t bits[B], zbits[B], rank[B];
    E v[B][R];
    // -- issue phase: ranks + loads of B iterations
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int row_in_tile = (j0 + b) * kRowsPerIter + lane * R;
      const int widx = row_in_tile >> 6;
      const int bpos = row_in_tile & 63;
      const uint64_t wE = shfl_u64(Ew, widx);
      const uint32_t wp = __shfl(p, widx, 64);
      bits[b] = static_cast<uint32_t>(wE >> bpos) & ((1u << R) - 1u);
      zbits[b] = 0;
      if (emit) zbits[b] = static_cast<uint32_t>(shfl_u64(Zw, widx) >> bpos) & ((1u << R) - 1u);
      rank[b] = wp + __popcll(wE & low_mask64(bpos));
      if constexpr (IOTA) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[b][r] = static_cast<E>(tile_row0 + row_in_tile + r);
      } else {
        const int64_t row = tile_row0 + row_in_tile;
        const uint8_t* src = a.values + row * static_cast<int64_t>(W);
        if (a.values_aligned16) {
          // one 16-byte load per lane.  Sparse mode: lanes with no emitted row issue nothing,
          // so untouched 64-byte sectors are never fetched from HBM.
          const bool do_load = DENSE ? (row < a.length) : (bits[b] != 0);
          uint4 q = make_uint4(0, 0, 0, 0);
          if (do_load) q = *reinterpret_cast<const uint4*>(src);
          if constexpr (W == 16) {
            v[b][0] = q;
          } else {
            const E* qe = reinterpret_cast<const E*>(&q);
#pragma unroll
            for (int r = 0; r < R; ++r) v[b][r] = qe[r];
          }
        } else {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            v[b][r] = zero_elem<W>();
            if ((bits[b] >> r) & 1u) {
              if constexpr (W == 16) {
                const uint64_t* s64 = reinterpret_cast<const uint64_t*>(src + r * W);
                const uint64_t lo64 = s64[0], hi64 = s64[1];
                v[b][r] = make_uint4(static_cast<uint32_t>(lo64), static_cast<uint32_t>(lo64 >> 32),
                                     static_cast<uint32_t>(hi64), static_cast<uint32_t>(hi64 >> 32));
              } else {
                v[b][r] = *reinterpret_cast<const E*>(src + r * W);
              }
            }
          }
        }
      }
    }
    // -- consume phase: emitted elements go to the ring at their output rank
#pragma unroll
    for (int b = 0; b < B; ++b) {
      if (bits[b] != 0) {
        uint32_t rk = rank[b];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if ((bits[b] >> r) & 1u) {
            E e = v[b][r];
            if ((zbits[b] >> r) & 1u) e = zero_elem<W>();
            const int ro = static_cast<int>((a0 + static_cast<int64_t>(rk) * W) & (RING - 1));
            *reinterpret_cast<E*>(ring + ro) = e;
            ++rk;
          }
        }
      }
    }

    // ---- wave-uniform flush decision: elements emitted through iteration j0 + B - 1
    const uint32_t cum = (j0 + B < kIters) ? __shfl(p, (j0 + B) * kWordsPerIter, 64) : total;
    const int64_t written = a0 + static_cast<int64_t>(cum) * W;
    if (written - flushed >= kFlushBytes) {
      wave_lds_sync();
      while (written - flushed >= kFlushBytes) {
        flush_region<W, RING>(ring, gbase, flushed, a0, written, lane);
        flushed += kFlushBytes;
      }
      wave_lds_sync();
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
    uint64_t c;
    if (__any((Ew & ~Vs) != 0)) {
      c = pext64(Vs, Ew);
    } else {
      c = low_mask64(static_cast<int>(k));
    }
    const int64_t obit = (off & 63) + p;  // bit position inside the tile-local LDS bitmap
    const int word = static_cast<int>(obit >> 6);
    const int sh = static_cast<int>(obit & 63);
    if (k != 0) {
      atomicOr(reinterpret_cast<unsigned long long*>(&vb[word]),
               static_cast<unsigned long long>(c << sh));
      if (sh != 0 && (sh + static_cast<int>(k)) > 64) {
        atomicOr(reinterpret_cast<unsigned long long*>(&vb[word + 1]),
                 static_cast<unsigned long long>(c >> (64 - sh)));
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

template <typename IdxT>
__device__ __forceinline__ uint64_t load_index(const uint8_t* p, int64_t i) {
  return static_cast<uint64_t>(reinterpret_cast<const IdxT*>(p)[i]);
}

template <int W, typename IdxT>
__global__ __launch_bounds__(kBlock) void take_kernel(TakeArgs a) {
  using E = typename ElemT<W>::type;
  constexpr int U = 4;  // independent gathers in flight per lane
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nchunks = (a.length + 64 * U - 1) / (64 * U);
  uint64_t nvalid = 0;
  for (int64_t c = wave_g; c < nchunks; c += nwaves) {
    const int64_t base = c * (64 * U);
    uint64_t idx[U];
    bool ok[U];
    E val[U];
    // index validity words for the U x 64 rows of this chunk (wave-uniform addresses)
    uint64_t ivw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) ivw[u] = load_word(a.ivalid, (base >> 6) + u);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pos = base + u * 64 + lane;
      const bool in = pos < a.length;
      idx[u] = in ? load_index<IdxT>(a.indices, pos) : 0;
      ok[u] = in && ((ivw[u] >> lane) & 1ull);
    }
    if (a.src_valid_bytes != nullptr) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ok[u]) {
          const uint64_t bit = static_cast<uint64_t>(a.src_valid_offset) + idx[u];
          ok[u] = (a.src_valid_bytes[bit >> 3] >> (bit & 7)) & 1;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ok[u]) {
        if constexpr (W == 16) {
          const uint64_t* s64 = reinterpret_cast<const uint64_t*>(a.values + idx[u] * 16);
          const uint64_t lo64 = s64[0], hi64 = s64[1];
          val[u] = make_uint4(static_cast<uint32_t>(lo64), static_cast<uint32_t>(lo64 >> 32),
                              static_cast<uint32_t>(hi64), static_cast<uint32_t>(hi64 >> 32));
        } else {
          val[u] = reinterpret_cast<const E*>(a.values)[idx[u]];
        }
      } else {
        val[u] = zero_elem<W>();
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pos = base + u * 64 + lane;
      if (pos < a.length) reinterpret_cast<E*>(a.out_data)[pos] = val[u];
      const uint64_t vb = __ballot(ok[u]);
      nvalid += __popcll(vb);
      if (a.out_validity != nullptr && lane == 0 && (base + u * 64) < a.length) {
        a.out_validity[(base >> 6) + u] = vb;
      }
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
                        hipStream_t st, int invert = 0) {
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
  hipLaunchKernelGGL(count_kernel, dim3(static_cast<unsigned>(ng)), dim3(kBlock), 0, st, mb, mvb,
                     null_selection, invert, nt, v.tile_counts, v.group_total);
  ARX_CHECK_LAUNCH("count_kernel");
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, v.group_total, ng, v.group_excl,
                     v.hdr);
  ARX_CHECK_LAUNCH("scan_kernel");
  return ARX_OK;
}

// Tuning knobs (arx_set_option): filter_batch in {1,4}, filter_dense in {0,1}.
static int g_filter_batch = 4;
static int g_filter_dense = 0;

template <int W, bool IOTA>
static void launch_compact_w(const CompactArgs& a, unsigned grid, hipStream_t st) {
  const int batch = (kTileRows / (kWave * (16 / W))) % 4 == 0 ? g_filter_batch : 1;
  if constexpr (IOTA) {
    if (batch >= 4) {
      hipLaunchKernelGGL((compact_kernel<W, true, 4, false>), dim3(grid), dim3(kBlock), 0, st, a);
    } else {
      hipLaunchKernelGGL((compact_kernel<W, true, 1, false>), dim3(grid), dim3(kBlock), 0, st, a);
    }
  } else {
    if (batch >= 4) {
      if (g_filter_dense) {
        hipLaunchKernelGGL((compact_kernel<W, false, 4, true>), dim3(grid), dim3(kBlock), 0, st, a);
      } else {
        hipLaunchKernelGGL((compact_kernel<W, false, 4, false>), dim3(grid), dim3(kBlock), 0, st, a);
      }
    } else {
      if (g_filter_dense) {
        hipLaunchKernelGGL((compact_kernel<W, false, 1, true>), dim3(grid), dim3(kBlock), 0, st, a);
      } else {
        hipLaunchKernelGGL((compact_kernel<W, false, 1, false>), dim3(grid), dim3(kBlock), 0, st, a);
      }
    }
  }
}

static int launch_compact(bool iota, int W, const CompactArgs& a, hipStream_t st) {
  const unsigned grid = static_cast<unsigned>(ceil_div(a.ntiles, kWavesPerBlock));
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
  if (strcmp(name, "filter_dense") == 0) {
    g_filter_dense = value != 0;
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
  return static_cast<size_t>(64 + ng * 16 + ng * kTilesPerGroup * 4 + 64);
}

int arx_filter_count_async(const ArxSpan* mask, int null_selection, void* ws, size_t ws_bytes,
                           void* stream) {
  return launch_count(mask, null_selection, ws, ws_bytes, as_stream(stream));
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
  return launch_compact(false, byte_width, a, st);
}

int arx_mask_to_indices(const ArxSpan* mask, int null_selection, const void* ws, int64_t out_length,
                        int index_width, void* out_indices, void* out_validity, void* stream) {
  int rc = check_mask(mask, null_selection);
  if (rc != ARX_OK) return rc;
  if (ws == nullptr) {
    set_error("ws is NULL");
    return ARX_INVALID;
  }
  if (index_width != 2 && index_width != 4) {
    set_error("index_width must be 2 or 4");
    return ARX_INVALID;
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
  const int64_t nchunks = ceil_div(indices->length, 256);
  const int64_t blocks = std::min<int64_t>(ceil_div(nchunks, kWavesPerBlock), 256 * 32);
  const dim3 grid(static_cast<unsigned>(blocks)), block(kBlock);
#define ARX_TAKE_W(WW)                                                                        \
  switch (iw) {                                                                               \
    case 1: hipLaunchKernelGGL((take_kernel<WW, uint8_t>), grid, block, 0, st, a); break;     \
    case 2: hipLaunchKernelGGL((take_kernel<WW, uint16_t>), grid, block, 0, st, a); break;    \
    case 4: hipLaunchKernelGGL((take_kernel<WW, uint32_t>), grid, block, 0, st, a); break;    \
    default: hipLaunchKernelGGL((take_kernel<WW, uint64_t>), grid, block, 0, st, a); break;   \
  }
  switch (byte_width) {
    case 1: ARX_TAKE_W(1); break;
    case 2: ARX_TAKE_W(2); break;
    case 4: ARX_TAKE_W(4); break;
    case 8: ARX_TAKE_W(8); break;
    case 16: ARX_TAKE_W(16); break;
    default:
      set_error("Unsupported primitive type for take: byte width %d", byte_width);
      return ARX_NOT_IMPLEMENTED;
  }
#undef ARX_TAKE_W
  ARX_CHECK_LAUNCH("take_kernel");
  return ARX_OK;
}

}  // extern "C"
