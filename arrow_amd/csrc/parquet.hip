// Parquet page decode on gfx950: the RLE / bit-packed hybrid that carries definition levels and
// dictionary indices.
//
// What it restates (semantics only):
//   RleBitPackedDecoder             cpp/src/arrow/util/rle_encoding_internal.h:40-90,462-
//     encoded-block := run*;  run := varint(count << 1) value   (repeated run)
//                                  | varint(groups << 1 | 1) groups*bit_width bytes (literal run,
//                                    8 values per group, LSB-first bit packing)
//   LevelDecoder::SetData           cpp/src/parquet/column_reader.cc:128-172 (levels: 4-byte length + runs)
//   DictDecoderImpl::SetData        cpp/src/parquet/decoder.cc (1 byte bit width + runs)
// The run headers are variable-length and sequential: the caller walks them once on the host (a
// few bytes per run, the page bytes are on the host anyway after decompression) and hands over a
// run table; here every output value finds its run by binary search and reads its bits directly,
// so the decode itself is embarrassingly parallel and reads the page bytes from HBM once.
#include "arx_common.h"
#include <string.h>

#include <algorithm>

namespace arx {

__device__ __forceinline__ uint32_t rle_value_at(const uint8_t* __restrict__ bytes, uint64_t nbytes,
                                                 const ArxRleRun* __restrict__ runs, int64_t nruns, int bit_width,
                                                 int64_t i) {
  // the last run whose first output index is <= i
  int64_t lo = 0, hi = nruns;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (static_cast<int64_t>(runs[mid].out_start) <= i) lo = mid; else hi = mid;
  }
  const ArxRleRun r = runs[lo];
  if ((r.kind & 0xFFu) == 0) return static_cast<uint32_t>(r.payload);
  // a run may carry its own bit width in bits 8.. of `kind`: the pages of one column chunk are decoded by
  // one launch, and a growing dictionary gives later pages wider indices
  if ((r.kind >> 8) != 0) bit_width = static_cast<int>(r.kind >> 8);
  const uint64_t bit = static_cast<uint64_t>(i - r.out_start) * static_cast<uint64_t>(bit_width);
  const uint64_t b0 = r.payload + (bit >> 3);
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {  // (bit & 7) + bit_width <= 7 + 32 bits = 5 bytes
    const uint64_t b = b0 + k;
    acc |= static_cast<uint64_t>(b < nbytes ? bytes[b] : 0) << (8 * k);
  }
  const uint64_t mask = bit_width >= 32 ? 0xFFFFFFFFull : ((1ull << bit_width) - 1ull);
  return static_cast<uint32_t>((acc >> (bit & 7)) & mask);
}

// BITS: emit bit i = (value == equals) as an LSB-first bitmap (definition levels -> validity)
template <bool BITS>
__global__ __launch_bounds__(kBlock) void rle_decode_kernel(const uint8_t* __restrict__ bytes, uint64_t nbytes,
                                                            const ArxRleRun* __restrict__ runs, int64_t nruns,
                                                            int bit_width, int64_t n, uint32_t equals,
                                                            uint32_t* __restrict__ out, uint64_t* __restrict__ out_bits) {
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwords = (n + 63) >> 6;
  for (int64_t w = wave_g; w < nwords; w += nwaves) {  // wave-uniform trip count: the ballot below is whole
    const int64_t i = (w << 6) + lane;
    uint32_t v = 0;
    if (i < n) v = rle_value_at(bytes, nbytes, runs, nruns, bit_width, i);
    if constexpr (BITS) {
      const uint64_t bal = __ballot(i < n && v == equals);
      if (lane == 0) out_bits[w] = bal;
    } else {
      if (i < n) out[i] = v;
    }
  }
}

static int rle_check(const void* bytes, const ArxRleRun* runs, int64_t nruns, int bit_width, int64_t n) {
  if (n < 0 || nruns < 0 || bit_width < 0 || bit_width > 32) {
    set_error("bad arguments to the RLE decode (bit_width %d)", bit_width);
    return ARX_INVALID;
  }
  if (n > 0 && (runs == nullptr || nruns == 0)) {
    set_error("RLE decode: %lld values but no runs", static_cast<long long>(n));
    return ARX_INVALID;
  }
  (void)bytes;
  return ARX_OK;
}


// ---------------------------------------------------------------- DELTA_BINARY_PACKED
// DeltaBitPackDecoder (cpp/src/parquet/decoder.cc; format: Encodings.md "Delta encoding"):
//   header  := varint(block size) varint(miniblocks per block) varint(total values) zigzag(first value)
//   block   := zigzag(min delta) bit_width[miniblocks] miniblock*
//   value_i := value_{i-1} + min_delta(block of i) + unpack(miniblock of i)            (wrap-around)
// The headers are walked once on the host (arx_delta_scan_miniblocks: one table entry per miniblock); every
// miniblock holds the same number of deltas, so value i finds its entry by division.  The recurrence is a
// prefix sum: per 4096-value tile the deltas are unpacked and summed (pass 1), the tile totals are scanned by
// one block, then every tile is unpacked again, scanned with its carry and written (pass 2) — the packed page
// is read twice, nothing intermediate per value is stored.
constexpr int kDeltaTile = 4096;

// The summand of position i: the page's first value at 0, then min_delta + the unpacked delta.
struct DeltaSource {
  const uint64_t* __restrict__ words;
  const ArxDeltaMiniblock* __restrict__ mbs;
  int64_t values_per_miniblock;
  long long first_value;
  __device__ __forceinline__ long long at(int64_t i) const {
    if (i == 0) return first_value;
    const int64_t j = i - 1;
    const ArxDeltaMiniblock mb = mbs[j / values_per_miniblock];
    if (mb.bit_width == 0) return static_cast<long long>(mb.min_delta);
    const uint64_t bit = mb.bit_start + static_cast<uint64_t>(j % values_per_miniblock) * mb.bit_width;
    const uint64_t w = bit >> 6;
    const int s = static_cast<int>(bit & 63);
    uint64_t v = words[w] >> s;
    if (s + static_cast<int>(mb.bit_width) > 64) v |= words[w + 1] << (64 - s);
    if (mb.bit_width < 64) v &= (uint64_t(1) << mb.bit_width) - 1;
    return static_cast<long long>(static_cast<uint64_t>(mb.min_delta) + v);
  }
};
// lengths -> offsets (DELTA_LENGTH_BYTE_ARRAY; offsets[0] = base, offsets[i] = offsets[i-1] + length[i-1]): the
// same recurrence with the lengths as the deltas
struct LengthSource {
  const int32_t* __restrict__ lengths;
  long long base;
  __device__ __forceinline__ long long at(int64_t i) const { return i == 0 ? base : lengths[i - 1]; }
};

template <typename Source>
__global__ __launch_bounds__(kBlock) void delta_tile_sums_kernel(Source src, int64_t n, long long* __restrict__ tile_sums) {
  __shared__ long long wave_sum[kWavesPerBlock];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kDeltaTile;
  unsigned long long acc = 0;
  for (int k = threadIdx.x; k < kDeltaTile; k += kBlock) {
    const int64_t i = base + k;
    if (i < n) acc += static_cast<unsigned long long>(src.at(i));
  }
  acc = wave_reduce_sum_u64(acc);
  if (lane_id() == 0) wave_sum[threadIdx.x >> 6] = static_cast<long long>(acc);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) t += static_cast<unsigned long long>(wave_sum[w]);
    tile_sums[blockIdx.x] = static_cast<long long>(t);
  }
}

// in-place exclusive scan of the tile totals by one block (same shape as bin_scan_blocks_kernel)
__global__ __launch_bounds__(1024) void delta_scan_tiles_kernel(long long* sums, int64_t ntiles) {
  __shared__ long long wave_tot[16];
  __shared__ long long carry_s;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < ntiles; base += 1024) {
    const int64_t i = base + tid;
    const unsigned long long v = i < ntiles ? static_cast<unsigned long long>(sums[i]) : 0;
    unsigned long long x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long nb = __shfl_up(x, d, 64);
      if (lane >= d) x += nb;
    }
    if (lane == 63) wave_tot[wave] = static_cast<long long>(x);
    __syncthreads();
    unsigned long long pre = 0;
    for (int k = 0; k < wave; ++k) pre += static_cast<unsigned long long>(wave_tot[k]);
    const unsigned long long carry = static_cast<unsigned long long>(carry_s);
    if (i < ntiles) sums[i] = static_cast<long long>(carry + pre + x - v);
    __syncthreads();
    if (tid == 1023) carry_s = static_cast<long long>(carry + pre + x);
    __syncthreads();
  }
}

template <typename OutT, typename Source>
__global__ __launch_bounds__(kBlock) void delta_write_kernel(Source src, int64_t n, const long long* __restrict__ tile_carry,
                                                             OutT* __restrict__ out) {
  __shared__ long long wave_tot[kWavesPerBlock];
  __shared__ long long carry_s;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kDeltaTile;
  if (threadIdx.x == 0) carry_s = tile_carry[blockIdx.x];
  __syncthreads();
  for (int k0 = 0; k0 < kDeltaTile; k0 += kBlock) {      // block-uniform trip count
    const int64_t i = base + k0 + threadIdx.x;
    const unsigned long long v = i < n ? static_cast<unsigned long long>(src.at(i)) : 0;
    unsigned long long x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long nb = __shfl_up(x, d, 64);
      if (lane >= d) x += nb;
    }
    if (lane == 63) wave_tot[wave] = static_cast<long long>(x);
    __syncthreads();
    unsigned long long pre = 0;
    for (int w = 0; w < wave; ++w) pre += static_cast<unsigned long long>(wave_tot[w]);
    const unsigned long long carry = static_cast<unsigned long long>(carry_s);
    if (i < n) out[i] = static_cast<OutT>(carry + pre + x);
    __syncthreads();
    if (threadIdx.x == kBlock - 1) carry_s = static_cast<long long>(carry + pre + x);
    __syncthreads();
  }
}

// All DELTA_BINARY_PACKED pages of a column chunk in ONE launch sequence: every page restarts the recurrence at its own
// first value, so tiles never span pages (page p owns tiles [first_tile, first_tile + ceil(count / 4096))) and a tile's
// carry is the difference of two entries of the plain (unsegmented) scan of the tile totals: excl[tile] -
// excl[first tile of its page] — the page's first value is the summand of its position 0.
__device__ __forceinline__ int64_t delta_page_of(const ArxDeltaPage* __restrict__ pages, int64_t num_pages, int64_t tile) {
  int64_t lo = 0, hi = num_pages - 1;     // last page with first_tile <= tile
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (pages[mid].first_tile <= tile) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ DeltaSource delta_page_source(const uint64_t* words, const ArxDeltaMiniblock* mbs, const ArxDeltaPage& pg) {
  return DeltaSource{words, mbs + pg.first_miniblock, pg.values_per_miniblock, static_cast<long long>(pg.first_value)};
}

__global__ __launch_bounds__(kBlock) void delta_pages_tile_sums_kernel(const uint64_t* __restrict__ words,
                                                                       const ArxDeltaMiniblock* __restrict__ mbs,
                                                                       const ArxDeltaPage* __restrict__ pages, int64_t num_pages,
                                                                       long long* __restrict__ tile_sums) {
  __shared__ long long wave_sum[kWavesPerBlock];
  const ArxDeltaPage pg = pages[delta_page_of(pages, num_pages, blockIdx.x)];
  const DeltaSource src = delta_page_source(words, mbs, pg);
  const int64_t base = (static_cast<int64_t>(blockIdx.x) - pg.first_tile) * kDeltaTile;
  unsigned long long acc = 0;
  for (int k = threadIdx.x; k < kDeltaTile; k += kBlock) {
    const int64_t i = base + k;
    if (i < pg.num_values) acc += static_cast<unsigned long long>(src.at(i));
  }
  acc = wave_reduce_sum_u64(acc);
  if (lane_id() == 0) wave_sum[threadIdx.x >> 6] = static_cast<long long>(acc);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) t += static_cast<unsigned long long>(wave_sum[w]);
    tile_sums[blockIdx.x] = static_cast<long long>(t);
  }
}

template <typename OutT>
__global__ __launch_bounds__(kBlock) void delta_pages_write_kernel(const uint64_t* __restrict__ words,
                                                                   const ArxDeltaMiniblock* __restrict__ mbs,
                                                                   const ArxDeltaPage* __restrict__ pages, int64_t num_pages,
                                                                   const long long* __restrict__ tile_excl, OutT* __restrict__ out) {
  __shared__ long long wave_tot[kWavesPerBlock];
  __shared__ long long carry_s;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const ArxDeltaPage pg = pages[delta_page_of(pages, num_pages, blockIdx.x)];
  const DeltaSource src = delta_page_source(words, mbs, pg);
  const int64_t base = (static_cast<int64_t>(blockIdx.x) - pg.first_tile) * kDeltaTile;
  OutT* __restrict__ dst = out + pg.out_start;
  if (threadIdx.x == 0) {
    carry_s = static_cast<long long>(static_cast<unsigned long long>(tile_excl[blockIdx.x]) -
                                     static_cast<unsigned long long>(tile_excl[pg.first_tile]));
  }
  __syncthreads();
  for (int k0 = 0; k0 < kDeltaTile; k0 += kBlock) {      // block-uniform trip count
    const int64_t i = base + k0 + threadIdx.x;
    const unsigned long long v = i < pg.num_values ? static_cast<unsigned long long>(src.at(i)) : 0;
    unsigned long long x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long nb = __shfl_up(x, d, 64);
      if (lane >= d) x += nb;
    }
    if (lane == 63) wave_tot[wave] = static_cast<long long>(x);
    __syncthreads();
    unsigned long long pre = 0;
    for (int w = 0; w < wave; ++w) pre += static_cast<unsigned long long>(wave_tot[w]);
    const unsigned long long carry = static_cast<unsigned long long>(carry_s);
    if (i < pg.num_values) dst[i] = static_cast<OutT>(carry + pre + x);
    __syncthreads();
    if (threadIdx.x == kBlock - 1) carry_s = static_cast<long long>(carry + pre + x);
    __syncthreads();
  }
}

// ---------------------------------------------------------------- BYTE_STREAM_SPLIT
// ByteStreamSplitDecoder (cpp/src/parquet/decoder.cc; arrow/util/byte_stream_split_internal.h): a page of n
// W-byte values is stored as W streams of n bytes (stream k = byte k of every value).  One value per lane:
// W coalesced byte loads (consecutive lanes read consecutive bytes of a stream), one W-byte store.
template <int W>
__global__ __launch_bounds__(kBlock) void byte_stream_split_kernel(const uint8_t* __restrict__ in, int64_t n,
                                                                  uint8_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint64_t v = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) v |= static_cast<uint64_t>(in[k * n + i]) << (8 * k);
    if constexpr (W == 8) {
      reinterpret_cast<uint64_t*>(out)[i] = v;
    } else if constexpr (W == 4) {
      reinterpret_cast<uint32_t*>(out)[i] = static_cast<uint32_t>(v);
    } else {
      reinterpret_cast<uint16_t*>(out)[i] = static_cast<uint16_t>(v);
    }
  }
}

// A/B knob snappy_lds: 1 = input window + recent output in LDS, 0 = every byte through global memory, -1 (default) =
// by the page count: the LDS form is 8-25 % faster on pages of >= 160 KB (profiles/r02_aq) but its 70 KB of LDS per
// workgroup halves the waves per CU, which costs 40-65 % when there are thousands of small pages to overlap
static Knob<int> g_snappy_lds{-1};
int set_parquet_option(const char* name, int64_t value) {
  if (strcmp(name, "snappy_lds") == 0) {
    g_snappy_lds = value < 0 ? -1 : (value != 0);
    return 1;
  }
  return 0;
}

// ------------------------------------------------------------------ Snappy page decompression
// What SnappyCodec::Decompress -> snappy::RawUncompress does per page (cpp/src/arrow/util/compression_snappy.cc:42-62;
// format: github.com/google/snappy format_description.txt, the bundled third-party codec is not in this tree): a varint
// with the uncompressed length, then elements — literal (tag & 3 == 0: length - 1 in the tag's upper six bits, or in the
// next 1..4 bytes when that field is 60..63) or copy (01: 3-bit length - 4, 11-bit offset; 10 / 11: 6-bit length - 1,
// 16- / 32-bit offset).  The element stream is sequential, so ONE WAVE owns a page: every lane follows the same
// (uniform) tag parse, the bytes of an element are moved by the 64 lanes together — a copy whose offset is shorter
// than its length repeats the last `offset` bytes, i.e. byte j comes from out[op - offset + j % offset], which makes
// even the self-overlapping case lane-parallel.  A copy that reads bytes this wave stored since the last fence waits for
// them first.  Pages are independent: the grid is one wave per page (4 per workgroup).
// status per page: 0 ok, 1 length mismatch / bad preamble, 2 element runs past the input or the output, 3 bad offset.
// A literal of `len` bytes moved by the 64 lanes.  Long literals (incompressible pages are made of 64 KB ones) go 16
// bytes per lane and four such loads in flight: the destination is brought to a 16-byte boundary first (bytewise, the
// same few bytes for every lane count), the source is read wherever it is (global memory takes unaligned dwordx4 loads).
__device__ __forceinline__ void snappy_copy_literal(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t len,
                                                    int lane) {
  if (len < 512) {
    for (uint32_t j = lane; j < len; j += 64) dst[j] = src[j];
    return;
  }
  const uint32_t head = static_cast<uint32_t>((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15);
  if (static_cast<uint32_t>(lane) < head) dst[lane] = src[lane];
  const uint32_t body = (len - head) & ~15u;   // whole 16-byte units
  uint8_t* d = dst + head;
  const uint8_t* s = src + head;
  uint32_t j = static_cast<uint32_t>(lane) * 16;
  for (; j + 3 * 1024 < body; j += 4 * 1024) {
    const uint4 a = load16_unaligned(s + j);
    const uint4 b = load16_unaligned(s + j + 1024);
    const uint4 c = load16_unaligned(s + j + 2048);
    const uint4 e = load16_unaligned(s + j + 3072);
    *reinterpret_cast<uint4*>(d + j) = a;
    *reinterpret_cast<uint4*>(d + j + 1024) = b;
    *reinterpret_cast<uint4*>(d + j + 2048) = c;
    *reinterpret_cast<uint4*>(d + j + 3072) = e;
  }
  for (; j < body; j += 1024) *reinterpret_cast<uint4*>(d + j) = load16_unaligned(s + j);
  const uint32_t tail = len - head - body;     // < 16
  if (static_cast<uint32_t>(lane) < tail) d[body + lane] = s[body + lane];
}

__global__ __launch_bounds__(kBlock) void snappy_decode_kernel(const uint8_t* __restrict__ src, const ArxSnappyPage* __restrict__ pages,
                                                               int64_t npages, uint8_t* dst, uint32_t* __restrict__ status) {
  const int lane = lane_id();
  const int64_t pg = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (pg >= npages) return;  // wave-uniform
  const ArxSnappyPage p = pages[pg];
  const uint8_t* in = src + p.src_offset;
  uint8_t* out = dst + p.dst_offset;
  const uint32_t n_in = p.src_size;
  uint32_t ip = 0, ulen = 0, err = 0;
  // preamble: varint32
  {
    int shift = 0;
    for (;;) {
      if (ip >= n_in || shift > 28) { err = 1; break; }
      const uint32_t c = in[ip++];
      ulen |= (c & 0x7Fu) << shift;
      if ((c & 0x80u) == 0) break;
      shift += 7;
    }
    if (!err && ulen != p.dst_size) err = 1;
  }
  uint32_t op = 0, pending = 0;
  while (!err && ip < n_in) {
    // tag + the up to four bytes that may follow it, fetched together (clamped: speculative, bounds are checked below)
    uint32_t b[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) b[k] = in[(ip + k) < n_in ? (ip + k) : (n_in - 1)];
    const uint32_t tag = b[0];
    uint32_t len, off = 0, adv;
    if ((tag & 3u) == 0) {
      len = (tag >> 2) + 1;
      adv = 1;
      if (len > 60) {
        const uint32_t nb = len - 60;
        uint32_t v = 0;
        for (uint32_t k = 0; k < nb; ++k) v |= b[1 + k] << (8 * k);
        len = v + 1;
        adv = 1 + nb;
        if (v == 0xFFFFFFFFu) { err = 2; break; }
      }
      if (ip + adv > n_in || len > n_in - (ip + adv) || len > ulen - op) { err = 2; break; }
      ip += adv;
      snappy_copy_literal(out + op, in + ip, len, lane);
      ip += len;
    } else {
      if ((tag & 3u) == 1) {
        len = 4 + ((tag >> 2) & 7u);
        off = ((tag >> 5) << 8) | b[1];
        adv = 2;
      } else if ((tag & 3u) == 2) {
        len = 1 + (tag >> 2);
        off = b[1] | (b[2] << 8);
        adv = 3;
      } else {
        len = 1 + (tag >> 2);
        off = b[1] | (b[2] << 8) | (b[3] << 16) | (b[4] << 24);
        adv = 5;
      }
      if (ip + adv > n_in || len > ulen - op) { err = 2; break; }
      if (off == 0 || off > op) { err = 3; break; }
      ip += adv;
      // the source range [op - off, op - off + min(len, off)) must not hold bytes still in flight
      if (static_cast<int64_t>(off) - static_cast<int64_t>(len < off ? len : off) < static_cast<int64_t>(pending)) {
        __threadfence_block();
        pending = 0;
      }
      const uint8_t* from = out + (op - off);
      if (off >= len) {
        for (uint32_t j = lane; j < len; j += 64) out[op + j] = from[j];
      } else {
        for (uint32_t j = lane; j < len; j += 64) out[op + j] = from[j % off];
      }
    }
    op += len;
    pending += len;
    // lanes move on to the next element together (a scheduling barrier on the GPU, where the wave runs in lockstep
    // anyway; the rendezvous the SIMT emulator needs before a later copy reads these bytes)
    __builtin_amdgcn_wave_barrier();
  }
  if (!err && op != ulen) err = 2;
  if (lane == 0) status[pg] = err;
}

// The same decoder with the two dependent global loads of every element taken out of its critical path: the next
// kSnWin bytes of INPUT sit in LDS (tag bytes and short literals are LDS reads; refilled by the 64 lanes, 16 bytes
// each) and so do the last kSnRing bytes of OUTPUT (a copy whose source lies there — offsets are mostly small — reads
// LDS and needs no fence for the wave's own stores in flight).  Every output byte still goes to global memory; the
// ring is a write-through cache of it.  Long literals keep the 16-bytes-per-lane global copy and just mark the ring as
// not holding them (ring_lo); copies that reach behind the ring take the global path of the kernel above.
constexpr uint32_t kSnWin = 1024;
constexpr uint32_t kSnRing = 16384;
struct __attribute__((aligned(16))) SnappyLds {
  uint8_t win[kWavesPerBlock][kSnWin + 16];
  uint8_t ring[kWavesPerBlock][kSnRing];
};

__global__ __launch_bounds__(kBlock) void snappy_decode_lds_kernel(const uint8_t* __restrict__ src, const ArxSnappyPage* __restrict__ pages,
                                                                   int64_t npages, uint8_t* dst, uint32_t* __restrict__ status) {
  __shared__ SnappyLds lds;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t pg = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  if (pg >= npages) return;  // wave-uniform (no workgroup barrier below)
  const ArxSnappyPage p = pages[pg];
  const uint8_t* in = src + p.src_offset;
  uint8_t* out = dst + p.dst_offset;
  uint8_t* win = lds.win[wave];
  uint8_t* ring = lds.ring[wave];
  const uint32_t n_in = p.src_size;
  uint32_t ip = 0, ulen = 0, err = 0;
  uint32_t wbase = 0, wlen = 0;   // the window holds input bytes [wbase, wbase + wlen)
  // refill so that the window starts at `at` (called with the lanes together; `at` < n_in)
  auto refill = [&](uint32_t at) {
    __builtin_amdgcn_wave_barrier();   // (every lane is done reading the old window)
    wbase = at;
    wlen = n_in - at < kSnWin ? n_in - at : kSnWin;
    const uint32_t k = static_cast<uint32_t>(lane) * 16;
    if (k + 16 <= wlen) {
      *reinterpret_cast<uint4*>(win + k) = load16_unaligned(in + at + k);
    } else {
      for (uint32_t b = k; b < wlen; ++b) win[b] = in[at + b];
    }
    __builtin_amdgcn_wave_barrier();
  };
  if (n_in > 0) refill(0);
  {  // preamble: varint32
    int shift = 0;
    for (;;) {
      if (ip >= n_in || ip >= wlen || shift > 28) { err = 1; break; }
      const uint32_t c = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(win[ip++]));
      ulen |= (c & 0x7Fu) << shift;
      if ((c & 0x80u) == 0) break;
      shift += 7;
    }
    if (!err && ulen != p.dst_size) err = 1;
  }
  uint32_t op = 0, pending = 0, ring_lo = 0;   // output bytes [ring_lo, op) younger than kSnRing are in the ring
  while (!err && ip < n_in) {
    if (ip + 5 > wbase + wlen && wbase + wlen < n_in) refill(ip);   // the tag and what may follow it
    const uint8_t* w = win + (ip - wbase);
    uint32_t b[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) b[k] = w[k];   // (past the input's end: stale window bytes, rejected by the checks below)
    // the bytes are the same in every lane: from here on the parse is scalar (SGPRs, SALU, scalar branches)
#pragma unroll
    for (int k = 0; k < 5; ++k) b[k] = __builtin_amdgcn_readfirstlane(b[k]);
    const uint32_t tag = b[0];
    uint32_t len, off = 0, adv;
    if ((tag & 3u) == 0) {
      len = (tag >> 2) + 1;
      adv = 1;
      if (len > 60) {
        const uint32_t nb = len - 60;
        uint32_t v = 0;
        for (uint32_t k = 0; k < nb; ++k) v |= b[1 + k] << (8 * k);
        len = v + 1;
        adv = 1 + nb;
        if (v == 0xFFFFFFFFu) { err = 2; break; }
      }
      if (ip + adv > n_in || len > n_in - (ip + adv) || len > ulen - op) { err = 2; break; }
      ip += adv;
      if (len >= 2048) {   // long literal: global to global, 16 bytes per lane; the ring does not get it
        snappy_copy_literal(out + op, in + ip, len, lane);
        ip += len;
        op += len;
        pending += len;
        ring_lo = op;
      } else {             // through the window, piece by piece
        uint32_t left = len;
        while (left > 0) {
          if (ip >= wbase + wlen) refill(ip);
          const uint32_t piece = left < wbase + wlen - ip ? left : wbase + wlen - ip;
          const uint8_t* from = win + (ip - wbase);
          for (uint32_t j = lane; j < piece; j += 64) {
            const uint8_t c = from[j];
            out[op + j] = c;
            ring[(op + j) & (kSnRing - 1)] = c;
          }
          ip += piece;
          op += piece;
          pending += piece;
          left -= piece;
        }
      }
    } else {
      if ((tag & 3u) == 1) {
        len = 4 + ((tag >> 2) & 7u);
        off = ((tag >> 5) << 8) | b[1];
        adv = 2;
      } else if ((tag & 3u) == 2) {
        len = 1 + (tag >> 2);
        off = b[1] | (b[2] << 8);
        adv = 3;
      } else {
        len = 1 + (tag >> 2);
        off = b[1] | (b[2] << 8) | (b[3] << 16) | (b[4] << 24);
        adv = 5;
      }
      if (ip + adv > n_in || len > ulen - op) { err = 2; break; }
      if (off == 0 || off > op) { err = 3; break; }
      ip += adv;
      // the source is in the ring (len <= 64: one step per lane) — and at least 64 slots away from the slots this element
      // overwrites (positions op.. map onto op - kSnRing..): on the GPU every lane reads before any lane writes, but a
      // copy must not depend on that
      if (off + 64 <= kSnRing && op - off >= ring_lo) {
        const uint32_t j = lane;
        if (j < len) {
          const uint8_t c = ring[(op - off + (off >= len ? j : j % off)) & (kSnRing - 1)];
          out[op + j] = c;
          ring[(op + j) & (kSnRing - 1)] = c;
        }
      } else {
        if (static_cast<int64_t>(off) - static_cast<int64_t>(len < off ? len : off) < static_cast<int64_t>(pending)) {
          __threadfence_block();
          pending = 0;
        }
        const uint8_t* from = out + (op - off);
        for (uint32_t j = lane; j < len; j += 64) {
          const uint8_t c = from[off >= len ? j : j % off];
          out[op + j] = c;
          ring[(op + j) & (kSnRing - 1)] = c;
        }
      }
      op += len;
      pending += len;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (!err && op != ulen) err = 2;
  if (lane == 0) status[pg] = err;
}

// ------------------------------------------------------------------ LZ4 (the buffers of a compressed IPC record batch)
// What Lz4FrameCodec::Decompress -> LZ4F_decompress does per buffer (cpp/src/arrow/util/compression_lz4.cc; the frame
// and block formats are lz4's published lz4_Frame_format.md / lz4_Block_format.md, the bundled codec is not in this
// tree).  The host walks the frame (magic, descriptor, block sizes: a few bytes per 64 KB..4 MB block) and hands over
// one STREAM per buffer = its blocks in order; blocks of a frame may be linked (a match reaches into the previous
// blocks' output), so one wave decodes a stream's blocks one after the other into the same output, and streams —
// the buffers of all columns — are independent.  A block: sequences of {token (literal length : match length - 4, 15 =
// more length bytes follow), literals, 2-byte offset, match}; the last sequence ends after its literals.
// status per stream: 0 ok, 1 output length mismatch, 2 a sequence runs past the block or the output, 3 bad offset.
__global__ __launch_bounds__(kBlock) void lz4_decode_kernel(const uint8_t* __restrict__ src, const ArxLz4Stream* __restrict__ streams,
                                                            const ArxLz4Block* __restrict__ blocks, int64_t nstreams, uint8_t* dst,
                                                            uint32_t* __restrict__ status) {
  const int lane = lane_id();
  const int64_t sid = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (sid >= nstreams) return;  // wave-uniform
  const ArxLz4Stream stream = streams[sid];
  uint8_t* out = dst + stream.dst_offset;
  const uint64_t ulen = stream.dst_size;
  uint64_t op = 0, pending = 0;
  uint32_t err = 0;
  for (uint32_t bi = 0; bi < stream.num_blocks && !err; ++bi) {
    const ArxLz4Block blk = blocks[stream.first_block + bi];
    const uint8_t* in = src + blk.src_offset;
    const uint32_t n_in = blk.src_size;
    if (blk.stored) {   // an incompressible block is kept as it is
      if (n_in > ulen - op) { err = 2; break; }
      snappy_copy_literal(out + op, in, n_in, lane);
      op += n_in;
      pending += n_in;
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    uint32_t ip = 0;
    while (ip < n_in) {
      const uint32_t token = in[ip++];
      // literals
      uint64_t len = token >> 4;
      if (len == 15) {
        for (;;) {
          if (ip >= n_in) { err = 2; break; }
          const uint32_t c = in[ip++];
          len += c;
          if (c != 255) break;
        }
        if (err) break;
      }
      if (len > n_in - ip || len > ulen - op) { err = 2; break; }
      if (len > 0) {
        snappy_copy_literal(out + op, in + ip, static_cast<uint32_t>(len), lane);
        ip += static_cast<uint32_t>(len);
        op += len;
        pending += len;
        __builtin_amdgcn_wave_barrier();   // (the match below may read these bytes)
      }
      if (ip >= n_in) break;   // the block's last sequence has no match
      // match
      if (ip + 2 > n_in) { err = 2; break; }
      const uint32_t off = in[ip] | (static_cast<uint32_t>(in[ip + 1]) << 8);
      ip += 2;
      uint64_t mlen = (token & 15u);
      if (mlen == 15) {
        for (;;) {
          if (ip >= n_in) { err = 2; break; }
          const uint32_t c = in[ip++];
          mlen += c;
          if (c != 255) break;
        }
        if (err) break;
      }
      mlen += 4;
      if (off == 0 || off > op) { err = 3; break; }
      if (mlen > ulen - op) { err = 2; break; }
      // the source range must not hold bytes this wave stored since the last fence
      if (static_cast<int64_t>(off) - static_cast<int64_t>(mlen < off ? mlen : off) < static_cast<int64_t>(pending)) {
        __threadfence_block();
        pending = 0;
      }
      const uint8_t* from = out + (op - off);
      if (off >= mlen) {
        for (uint64_t j = lane; j < mlen; j += 64) out[op + j] = from[j];
      } else {
        for (uint64_t j = lane; j < mlen; j += 64) out[op + j] = from[j % off];
      }
      op += mlen;
      pending += mlen;
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (!err && op != ulen) err = 1;
  if (lane == 0) status[sid] = err;
}

// ---- definition levels of a flat optional column (bit width 1) -> validity bits, run headers walked on the device.
// One wave per page.  Every lane parses the same header bytes (uniform loads), then the 64 lanes share the run's bits:
// a repeated run of ones sets a bit range, a bit-packed run ORs its payload bytes in at the page's bit position (pages
// start at any row, so neighbours share words: atomicOr into a zeroed bitmap).  The checks are those of the host
// walker (arx_rle_scan_runs): a header or payload that runs past the block marks the page corrupt.
__global__ __launch_bounds__(kBlock) void rle_levels_bitmap_kernel(const uint8_t* __restrict__ bytes,
                                                                   const ArxLevelPage* __restrict__ pages, int64_t num_pages,
                                                                   unsigned long long* __restrict__ out_bits,
                                                                   uint32_t* __restrict__ ones, uint32_t* __restrict__ status) {
  const int64_t pg = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (pg >= num_pages) return;   // wave-uniform
  const int lane = lane_id();
  const ArxLevelPage page = pages[pg];
  const uint8_t* p = bytes + page.byte_start;
  const uint64_t nbytes = page.nbytes;
  const int64_t num_values = page.num_values;
  uint64_t pos = 0;
  int64_t done = 0;
  uint32_t my_ones = 0, bad = 0;
  while (done < num_values) {
    uint64_t h = 0;
    int shift = 0;
    for (;;) {   // varint header
      if (pos >= nbytes || shift > 56) {
        bad = 1;
        break;
      }
      const uint8_t c = p[pos++];
      h |= static_cast<uint64_t>(c & 0x7F) << shift;
      if ((c & 0x80) == 0) break;
      shift += 7;
    }
    if (bad) break;
    const uint64_t bit0 = page.row_start + static_cast<uint64_t>(done);   // where this run's first value lands
    if (h & 1) {   // bit-packed: (h >> 1) groups of 8 one-bit values = that many bytes
      const uint64_t groups = h >> 1;
      if (groups == 0 || groups > (uint64_t(1) << 40)) {
        bad = 1;
        break;
      }
      const int64_t count = static_cast<int64_t>(groups) * 8;
      const int64_t needed = count < num_values - done ? count : num_values - done;
      const uint64_t need_bytes = (static_cast<uint64_t>(needed) + 7) / 8;
      const bool reaches_end = done + count >= num_values;
      if ((reaches_end ? need_bytes : groups) > nbytes - pos) {
        bad = 1;
        break;
      }
      // lane l takes bytes [8 l, 8 l + 8) of every 512-byte stretch
      for (uint64_t b0 = static_cast<uint64_t>(lane) * 8; b0 < need_bytes; b0 += 512) {
        uint64_t w = 0;
        const uint64_t nb = need_bytes - b0 < 8 ? need_bytes - b0 : 8;
        for (uint64_t k = 0; k < nb; ++k) w |= static_cast<uint64_t>(p[pos + b0 + k]) << (8 * k);
        const int64_t bits_here = needed - static_cast<int64_t>(b0 * 8) < 64 ? needed - static_cast<int64_t>(b0 * 8) : 64;
        if (bits_here < 64) w &= (uint64_t(1) << bits_here) - 1;
        if (w != 0) {
          my_ones += static_cast<uint32_t>(__popcll(w));
          const uint64_t at = bit0 + b0 * 8;
          const int sh = static_cast<int>(at & 63);
          atomicOr(&out_bits[at >> 6], static_cast<unsigned long long>(w << sh));
          if (sh != 0 && (w >> (64 - sh)) != 0) atomicOr(&out_bits[(at >> 6) + 1], static_cast<unsigned long long>(w >> (64 - sh)));
        }
      }
      pos += groups;
      done += count;
    } else {   // repeated: (h >> 1) copies of one byte-sized value
      const int64_t count = static_cast<int64_t>(h >> 1);
      if (count == 0 || pos + 1 > nbytes) {
        bad = 1;
        break;
      }
      const uint8_t value = p[pos];
      pos += 1;
      const int64_t take = count < num_values - done ? count : num_values - done;
      if (value == 1) {
        if (lane == 0) my_ones += static_cast<uint32_t>(take);
        const uint64_t first = bit0, last = bit0 + static_cast<uint64_t>(take) - 1;   // inclusive
        for (uint64_t wd = (first >> 6) + lane; wd <= (last >> 6); wd += 64) {
          uint64_t m = ~uint64_t(0);
          if (wd == (first >> 6)) m &= ~uint64_t(0) << (first & 63);
          if (wd == (last >> 6)) m &= ~uint64_t(0) >> (63 - (last & 63));
          atomicOr(&out_bits[wd], static_cast<unsigned long long>(m));
        }
      } else if (value > 1) {   // a level above the column's maximum
        bad = 1;
        break;
      }
      done += count;
    }
  }
  const uint32_t total = wave_reduce_sum_u32(my_ones);
  if (lane == 0) {
    ones[pg] = total;
    status[pg] = bad;
  }
}


// ---- DELTA_BYTE_ARRAY (DeltaByteArrayDecoderImpl, cpp/src/parquet/decoder.cc:1974-2204): value i = the first
// prefix[i] bytes of value i - 1 ++ suffix i; the prefix lengths are a DELTA_BINARY_PACKED stream, the suffixes a
// DELTA_LENGTH_BYTE_ARRAY block (both decoded by the kernels above), and every page starts from the empty string
// (SetData: last_value_.clear(), :2017).  The recurrence runs over whole values, so a page is ONE wave's sequential
// walk — pages are the parallel axis, as they are the reference's unit of decoding — with the lanes as the byte axis:
// the current value's first kDbaWindow bytes live in LDS (a value keeps its prefix there and overwrites the rest
// with its suffix), the suffix bytes of 64 values at a time are staged through LDS with one round trip, and every
// output byte is written once.
constexpr int kDbaWindow = 8192;   // bytes of the current value kept in LDS (longer prefixes are re-read from the output)
constexpr int kDbaStage = 8192;    // suffix bytes of a group of 64 values staged in LDS

__device__ __forceinline__ int64_t dba_page_of(const int64_t* __restrict__ page_first, int64_t npages, int64_t i) {
  int64_t lo = 0, hi = npages;   // the last page whose first value is <= i
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (page_first[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// out_len[i] = prefix[i] + suffix_len[i], and the checks of :2096-2107 / BuildBufferInternal :2039: status bit 0 =
// negative prefix length, bit 1 = prefix longer than the previous value (the first value of a page: longer than ""),
// bit 2 = negative suffix length, bit 3 = a value past the int32 offsets
__global__ __launch_bounds__(kBlock) void dba_lengths_kernel(const int32_t* __restrict__ prefix, const int32_t* __restrict__ suffix_len,
                                                             int64_t n, const int64_t* __restrict__ page_first, int64_t npages,
                                                             int32_t* __restrict__ out_len, unsigned long long* __restrict__ state) {
  uint32_t bad = 0;
  unsigned long long bytes = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int32_t p = prefix[i], s = suffix_len[i];
    if (p < 0) bad |= 1u;
    if (s < 0) bad |= 4u;
    const bool first = page_first[dba_page_of(page_first, npages, i)] == i;
    const int64_t before = first ? 0 : static_cast<int64_t>(prefix[i - 1]) + suffix_len[i - 1];
    if (p > before) bad |= 2u;
    const int64_t len = static_cast<int64_t>(p) + s;
    if (len > INT32_MAX) bad |= 8u;
    const int32_t kept = static_cast<int32_t>(len < 0 ? 0 : (len > INT32_MAX ? INT32_MAX : len));
    out_len[i] = kept;
    bytes += static_cast<unsigned long long>(kept);
  }
  if (bad != 0) atomicOr(&state[0], static_cast<unsigned long long>(bad));
  bytes = wave_reduce_sum_u64(bytes);
  if (lane_id() == 0 && bytes != 0) atomicAdd(&state[1], bytes);
}

struct DbaLds {
  uint8_t cur[kDbaWindow];
  uint32_t stage[kDbaStage / 4 + 2];
};

__global__ __launch_bounds__(64) void dba_expand_kernel(const int32_t* __restrict__ prefix, const int32_t* __restrict__ suffix_off,
                                                        const uint8_t* __restrict__ suffix, int64_t suffix_size,
                                                        const int32_t* __restrict__ out_off, int32_t out_base,
                                                        const int64_t* __restrict__ page_first,
                                                        const int64_t* __restrict__ page_suffix_first, unsigned long long* __restrict__ state,
                                                        uint8_t* out) {
  __shared__ DbaLds lds;
  const int lane = threadIdx.x;
  const int64_t v0 = page_first[blockIdx.x], v1 = page_first[blockIdx.x + 1];
  if (v0 >= v1) return;
  // the page's suffix lengths must add up to the suffix bytes it carries (else its values would read their neighbours')
  if (page_suffix_first != nullptr &&
      (suffix_off[v0] != page_suffix_first[blockIdx.x] || suffix_off[v1] != page_suffix_first[blockIdx.x + 1])) {
    if (lane == 0) atomicOr(&state[0], 16ull);
    return;
  }
  const uint8_t* stage_bytes = reinterpret_cast<const uint8_t*>(lds.stage);
  int64_t prev_out = 0;   // where the previous value starts in `out` (prefixes past the LDS window are read there)
  for (int64_t g = v0; g < v1; g += 64) {
    const int64_t i = g + lane < v1 ? g + lane : v1 - 1;
    const int32_t my_p = prefix[i], my_so = suffix_off[i], my_se = suffix_off[i + 1], my_oo = out_off[i];
    const int cnt = static_cast<int>(v1 - g < 64 ? v1 - g : 64);
    // the group's suffix bytes are one contiguous range of the stream: staged with aligned dword loads when they fit
    const int64_t s_lo = __shfl(my_so, 0, 64), s_hi = __shfl(my_se, cnt - 1, 64);
    const int64_t a_lo = s_lo & ~int64_t(3);
    const bool staged = s_hi - a_lo <= kDbaStage && ((s_hi + 3) & ~int64_t(3)) <= ((suffix_size + 3) & ~int64_t(3));
    __syncthreads();   // (one wave: the previous group's reads of the stage come before it is overwritten)
    if (staged) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(suffix + a_lo);
      const int nd = static_cast<int>((s_hi - a_lo + 3) >> 2);
      for (int k = lane; k < nd; k += 64) lds.stage[k] = src[k];
    }
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      const int32_t p = __shfl(my_p, j, 64);
      const int64_t so = __shfl(my_so, j, 64);
      const int32_t sl = __shfl(my_se, j, 64) - static_cast<int32_t>(so);
      const int64_t oo = static_cast<int64_t>(__shfl(my_oo, j, 64)) - out_base;
      // (a negative length is corrupt data the lengths kernel has flagged — callers stop there; a value is never written
      //  before its own start all the same)
      if (p < 0 || sl < 0 || so < 0 || so + sl > suffix_size) continue;
      // (a) the part of the prefix past the LDS window: from the previous value's bytes in the output (rare: > 8 KB)
      if (p > kDbaWindow) {
        __threadfence();
        for (int b = kDbaWindow + lane; b < p; b += 64) out[oo + b] = out[prev_out + b];
      }
      // (b) the prefix inside the window, as the previous value left it
      const int pw = p < kDbaWindow ? p : kDbaWindow;
      for (int b = lane; b < pw; b += 64) out[oo + b] = lds.cur[b];
      // (c) the suffix: to the output and, where it falls inside the window, over the tail of the previous value
      for (int b = lane; b < sl; b += 64) {
        const uint8_t byte = staged ? stage_bytes[so - a_lo + b] : suffix[so + b];
        const int q = p + b;
        out[oo + q] = byte;
        if (q < kDbaWindow) lds.cur[q] = byte;
      }
      prev_out = oo;
      __syncthreads();   // the next value reads what this one wrote into the window
    }
  }
}

// ---------------------------------------------------------------------------
// Repeated columns: DefRepLevelsToList (cpp/src/parquet/level_conversion.cc:40-124) and DefLevelsToBitmap over the
// decoded level arrays.  The reference walks the levels once, one at a time; here every level slot classifies itself —
//   kept   = def >= repeated_ancestor_def_level && rep <= rep_level     (level_conversion.cc:52-55)
//   start  = kept && rep <  rep_level   (a new list entry: one offset, one validity bit)
//   elem   = kept && (rep == rep_level || def >= def_level)   (the entry's offset grows by one, :57-66 and :86-92)
// — and two running counts (starts, elems) carried by a tile scan give entry j its offset (the elems before its
// start slot) and its bit (def >= def_level - 1, :98-106).  Starts in the high and elems in the low half of one
// 64-bit word: one scan for both (both are below 2^31 by the entry check).
constexpr int kListTile = 4096;

struct ListLevels {
  const uint32_t* def;
  const uint32_t* rep;
  uint32_t def_level, rep_level, ancestor;
  __device__ __forceinline__ unsigned long long at(int64_t i, uint32_t* d_out) const {
    const uint32_t d = def[i];
    const uint32_t r = rep != nullptr ? rep[i] : 0u;
    *d_out = d;
    if (d < ancestor || r > rep_level) return 0ull;
    if (r == rep_level) return 1ull;                               // a continuation of the current list
    return (1ull << 32) | (d >= def_level ? 1ull : 0ull);           // a start (with or without a first element)
  }
};

__global__ __launch_bounds__(kBlock) void list_tile_sums_kernel(ListLevels lv, int64_t n, long long* __restrict__ tile_sums) {
  __shared__ long long wave_sum[kWavesPerBlock];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kListTile;
  unsigned long long acc = 0;
  for (int k = threadIdx.x; k < kListTile; k += kBlock) {
    const int64_t i = base + k;
    uint32_t d;
    if (i < n) acc += lv.at(i, &d);
  }
  acc = wave_reduce_sum_u64(acc);
  if (lane_id() == 0) wave_sum[threadIdx.x >> 6] = static_cast<long long>(acc);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) t += static_cast<unsigned long long>(wave_sum[w]);
    tile_sums[blockIdx.x] = static_cast<long long>(t);
  }
}

// counts (device): [0] entries, [1] elements, [2] null entries, [3] 1 if more than max_entries entries
__global__ __launch_bounds__(kBlock) void list_write_kernel(ListLevels lv, int64_t n, const long long* __restrict__ tile_carry,
                                                            int64_t max_entries, int32_t* __restrict__ offsets,
                                                            uint32_t* __restrict__ valid_bits,
                                                            unsigned long long* __restrict__ counts) {
  __shared__ long long wave_tot[kWavesPerBlock];
  __shared__ long long carry_s;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kListTile;
  if (threadIdx.x == 0) carry_s = tile_carry[blockIdx.x];
  __syncthreads();
  uint32_t nulls = 0;
  for (int k0 = 0; k0 < kListTile; k0 += kBlock) {      // block-uniform trip count
    const int64_t i = base + k0 + threadIdx.x;
    uint32_t d = 0;
    const unsigned long long v = i < n ? lv.at(i, &d) : 0ull;
    const unsigned long long x = wave_inclusive_scan_u64(v);
    if (lane == 63) wave_tot[wave] = static_cast<long long>(x);
    __syncthreads();
    unsigned long long pre = 0;
    for (int w = 0; w < wave; ++w) pre += static_cast<unsigned long long>(wave_tot[w]);
    const unsigned long long carry = static_cast<unsigned long long>(carry_s);
    const unsigned long long excl = carry + pre + x - v;
    if (v >> 32) {
      const int64_t j = static_cast<int64_t>(excl >> 32);
      if (j < max_entries) {
        offsets[j] = static_cast<int32_t>(excl & 0xFFFFFFFFull);
        if (valid_bits != nullptr) {
          if (d + 1 >= lv.def_level) atomicOr(&valid_bits[j >> 5], 1u << (j & 31));
          else ++nulls;
        }
      }
    }
    if (i == n - 1) {
      const unsigned long long tot = excl + v;
      const int64_t entries = static_cast<int64_t>(tot >> 32);
      counts[0] = static_cast<unsigned long long>(entries);
      counts[1] = tot & 0xFFFFFFFFull;
      if (entries > max_entries) counts[3] = 1;
      else offsets[entries] = static_cast<int32_t>(tot & 0xFFFFFFFFull);
    }
    __syncthreads();
    if (threadIdx.x == kBlock - 1) carry_s = static_cast<long long>(carry + pre + x);
    __syncthreads();
  }
  nulls = wave_reduce_sum_u32(nulls);
  if (lane == 0 && nulls != 0) atomicAdd(&counts[2], static_cast<unsigned long long>(nulls));
}

// bit i = levels[i] >= threshold (every slot keeps its place: DefLevelsToBitmap for a column without a repeated
// ancestor, level_conversion_inc.h, and the "slot exists" mask of one with); ones (device, may be NULL) += set bits
__global__ __launch_bounds__(kBlock) void levels_ge_bitmap_kernel(const uint32_t* __restrict__ levels, int64_t n, uint32_t threshold,
                                                                  uint64_t* __restrict__ out_bits,
                                                                  unsigned long long* __restrict__ ones) {
  const int lane = lane_id();
  const int64_t nwords = (n + 63) >> 6;
  const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  unsigned long long local = 0;
  for (int64_t w = wave0; w < nwords; w += stride) {
    const int64_t i = w * 64 + lane;
    const bool bit = i < n && levels[i] >= threshold;
    const uint64_t word = __ballot(bit);
    if (lane == 0) {
      out_bits[w] = word;
      local += static_cast<unsigned long long>(__popcll(word));
    }
  }
  if (ones != nullptr && lane == 0 && local != 0) atomicAdd(ones, local);
}

// ---------------------------------------------------------------------------
// GZIP / zlib page decompression on the device — GZipCodec::Decompress (cpp/src/arrow/util/compression_zlib.cc:88-180:
// inflateInit2 with window bits 15 | 32 = header auto-detection).  zlib is a bundled third-party dependency
// (cpp/thirdparty/versions.txt:122 pins 1.3.1), not vendored in the tree: the formats are restated from their published
// specifications — RFC 1952 (gzip member: 10-byte header, optional FEXTRA / FNAME / FCOMMENT / FHCRC, deflate stream,
// CRC-32 + ISIZE), RFC 1950 (zlib: CMF / FLG, deflate stream, Adler-32) and RFC 1951 (deflate: stored, fixed-Huffman and
// dynamic-Huffman blocks; canonical codes packed most-significant bit first into a least-significant-bit-first stream;
// length / distance symbols with extra bits; a 32 KB window).  Checksums are skipped, ISIZE is checked.
//
// One wave per page, as for Snappy: the symbol stream is sequential, so all 64 lanes decode it in lockstep (uniform control
// flow, every lane holds the same bit buffer) out of an input window staged in LDS.  The last 32 KB of OUTPUT — deflate's
// whole window: every possible match source — live in an LDS ring as well (a write-through copy: every byte also goes to
// global memory), so a match never reads global memory back and no store has to be waited for; a literal is one byte by
// lane 0, a match is copied by the 64 lanes (read, rendezvous, write: source and destination may share ring slots when the
// distance is within 258 bytes of the ring size).  Per wave (= per workgroup of 64 threads; 39 KB, four to a CU) in LDS: the
// ring, the input window, a 10-bit lookup table for the literal / length code and a 9-bit one for the distance code (entry
// = symbol << 4 | code length; the bit-reversed code is the index, every longer code falls back to the canonical walk
// over count[] / symbol[]), the code lengths being read.
constexpr uint32_t kInfWin = 2048;          // bytes of input staged per wave
constexpr uint32_t kInfRing = 32768;        // deflate's window (RFC 1951: distances up to 32768)
constexpr int kInfLitBits = 10;
constexpr int kInfDistBits = 9;
constexpr int kInfThreads = 64;
struct __attribute__((aligned(16))) InflateLds {
  uint8_t ring[kInfRing];
  uint8_t win[kInfWin + 16];
  uint16_t lit_lut[1 << kInfLitBits];
  uint16_t dist_lut[1 << kInfDistBits];
  uint16_t lit_sym[288];     // symbols sorted by (code length, symbol): the canonical walk's table
  uint16_t dist_sym[32];
  uint16_t lit_count[16];    // codes per length
  uint16_t dist_count[16];
  uint8_t lengths[320];      // code lengths of the block being set up (288 + 32)
  uint8_t staged[320];       // ... as the code-length code delivers them (literal / length, then distance)
};

struct InflateBits {
  const uint8_t* in;        // the page's compressed bytes (global)
  uint8_t* win;             // this wave's LDS window: win[k] = in[win_base + k]
  uint32_t n_in, win_base, ip;   // ip: next input byte not yet in the bit buffer
  uint64_t buf;             // bit buffer, next bit = bit 0
  int cnt;                  // valid bits in buf
  int lane;
  uint32_t overrun;         // bits requested past the end of the input
};

// the window covers [win_base, win_base + kInfWin); refilled by the 64 lanes (16 bytes each, twice) when ip leaves it
__device__ __forceinline__ void inflate_stage(InflateBits& b) {
  b.win_base = b.ip;
  __builtin_amdgcn_wave_barrier();
  for (uint32_t k = static_cast<uint32_t>(b.lane) * 16; k < kInfWin + 16; k += 64 * 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t at = b.win_base + k + j;
      b.win[k + j] = at < b.n_in ? b.in[at] : 0;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void inflate_refill(InflateBits& b) {
  while (b.cnt <= 32) {   // four bytes at a time (the window holds 16 bytes beyond kInfWin, zeros beyond the input)
    if (b.ip - b.win_base + 4 > kInfWin) inflate_stage(b);
    // two ALIGNED 32-bit LDS reads and a shift (four byte reads get merged into one unaligned ds_read_b32, which this
    // target's LDS does not serve: it returns the aligned word)
    const uint32_t off = b.ip - b.win_base;
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(b.win);
    const uint64_t pair = static_cast<uint64_t>(w32[off >> 2]) | (static_cast<uint64_t>(w32[(off >> 2) + 1]) << 32);
    const uint64_t four = (pair >> ((off & 3u) * 8u)) & 0xFFFFFFFFull;
    b.buf |= four << b.cnt;
    b.cnt += 32;
    b.ip += 4;
  }
}

__device__ __forceinline__ uint32_t inflate_take(InflateBits& b, int n) {   // n <= 32
  if (b.cnt < n) inflate_refill(b);
  const uint32_t v = static_cast<uint32_t>(b.buf & ((1ull << n) - 1ull));
  b.buf >>= n;
  b.cnt -= n;
  return v;
}

// bytes the decoder has really consumed: what went into the bit buffer minus the whole bytes still in it
__device__ __forceinline__ uint32_t inflate_consumed(const InflateBits& b) { return b.ip - static_cast<uint32_t>(b.cnt >> 3); }

// canonical walk (any code length up to 15) over the next bits; returns the symbol or 0xFFFF for an unassigned code
__device__ __forceinline__ uint32_t inflate_walk(InflateBits& b, const uint16_t* count, const uint16_t* symbol) {
  if (b.cnt < 15) inflate_refill(b);
  uint32_t code = 0, first = 0, index = 0;
  uint64_t w = b.buf;
  for (int len = 1; len <= 15; ++len) {
    code |= static_cast<uint32_t>(w & 1u);
    w >>= 1;
    const uint32_t c = count[len];
    if (code < first + c) {
      b.buf >>= len;
      b.cnt -= len;
      return symbol[index + (code - first)];
    }
    index += c;
    first = (first + c) << 1;
    code <<= 1;
  }
  return 0xFFFFu;
}

// Builds count[] / symbol[] and the lookup table of one code from lengths[0, n).  Returns 0, or 4 for an over-subscribed
// code (an incomplete code is accepted, as zlib accepts the single-code distance tables real encoders emit; its unassigned
// codes fail when they are met).  Wave-uniform; lane 0 writes the serial tables, all lanes fill the lookup table.
__device__ __forceinline__ uint32_t inflate_build(const uint8_t* lengths, int n, uint16_t* count, uint16_t* symbol, uint16_t* lut,
                                                   int lut_bits, int lane) {
  __builtin_amdgcn_wave_barrier();
  uint32_t cnt[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) cnt[l] = 0;
  for (int s = 0; s < n; ++s) ++cnt[lengths[s]];
  int left = 1;
  for (int l = 1; l <= 15; ++l) {
    left = (left << 1) - static_cast<int>(cnt[l]);
    if (left < 0) return 4;
  }
  uint32_t offs[16], next_code[16];
  offs[1] = 0;
  next_code[0] = 0;
  next_code[1] = 0;
  for (int l = 1; l < 15; ++l) {
    offs[l + 1] = offs[l] + cnt[l];
    next_code[l + 1] = (next_code[l] + cnt[l]) << 1;
  }
  if (lane == 0) {
    count[0] = 0;
    for (int l = 1; l <= 15; ++l) count[l] = static_cast<uint16_t>(cnt[l]);
  }
  for (uint32_t k = lane; k < (1u << lut_bits); k += 64) lut[k] = 0;
  __builtin_amdgcn_wave_barrier();
  // serial: a symbol's code is next_code[its length] + the number of earlier symbols of that length
  for (int s = 0; s < n; ++s) {
    const int l = lengths[s];
    if (l == 0) continue;
    const uint32_t code = next_code[l]++;
    const uint32_t at = offs[l]++;
    if (lane == 0) symbol[at] = static_cast<uint16_t>(s);
    if (l <= lut_bits) {
      uint32_t rev = 0;
      for (int k = 0; k < l; ++k) rev |= ((code >> k) & 1u) << (l - 1 - k);
      const uint16_t entry = static_cast<uint16_t>((s << 4) | l);
      for (uint32_t k = rev + (static_cast<uint32_t>(lane) << l); k < (1u << lut_bits); k += 64u << l) lut[k] = entry;
    }
  }
  __builtin_amdgcn_wave_barrier();
  return 0;
}

__device__ __forceinline__ uint32_t inflate_symbol(InflateBits& b, const uint16_t* lut, int lut_bits, const uint16_t* count,
                                                   const uint16_t* symbol) {
  if (b.cnt < 15) inflate_refill(b);
  const uint16_t e = lut[b.buf & ((1u << lut_bits) - 1u)];
  const int l = e & 15;
  if (l != 0) {
    b.buf >>= l;
    b.cnt -= l;
    return e >> 4;
  }
  return inflate_walk(b, count, symbol);
}

// status: 0 ok, 1 bad gzip / zlib header or the stream does not produce dst_size bytes, 2 the stream runs past its block or
// the output, 3 a distance reaches before the output, 4 an invalid Huffman code (over-subscribed table, unassigned code,
// bad block type or stored-block length)
__global__ __launch_bounds__(kInfThreads) void inflate_pages_kernel(const uint8_t* __restrict__ src, const ArxSnappyPage* __restrict__ pages,
                                                               int64_t npages, uint8_t* dst, uint32_t* __restrict__ status) {
  __shared__ InflateLds lds;
  const int lane = threadIdx.x;
  const int64_t pg = blockIdx.x;
  if (pg >= npages) return;
  const ArxSnappyPage p = pages[pg];
  const uint8_t* in = src + p.src_offset;
  uint8_t* out = dst + p.dst_offset;
  const uint32_t n_in = p.src_size, ulen = p.dst_size;
  uint32_t err = 0, start = 0, trailer = 0;
  // ---- container header (auto-detection, as inflateInit2 with 15 | 32)
  if (n_in >= 18 && in[0] == 0x1F && in[1] == 0x8B) {          // gzip member (RFC 1952)
    const uint32_t flg = in[3];
    if (in[2] != 8 || (flg & 0xE0u) != 0) err = 1;
    uint32_t at = 10;
    if (!err && (flg & 4u)) {                                  // FEXTRA: 2-byte length + that many bytes
      if (at + 2 > n_in) err = 1; else at += 2 + (in[at] | (static_cast<uint32_t>(in[at + 1]) << 8));
    }
    for (uint32_t bit = 8; bit <= 16 && !err; bit <<= 1) {     // FNAME, FCOMMENT: zero-terminated
      if (flg & bit) {
        while (at < n_in && in[at] != 0) ++at;
        ++at;
      }
    }
    if (!err && (flg & 2u)) at += 2;                           // FHCRC
    if (at + 8 > n_in) err = 1;
    start = at;
    trailer = 8;
    if (!err) {
      const uint32_t isize = in[n_in - 4] | (static_cast<uint32_t>(in[n_in - 3]) << 8) | (static_cast<uint32_t>(in[n_in - 2]) << 16) |
                             (static_cast<uint32_t>(in[n_in - 1]) << 24);
      if (isize != ulen) err = 1;
    }
  } else if (n_in >= 6 && (in[0] & 0x0Fu) == 8 && ((static_cast<uint32_t>(in[0]) << 8) | in[1]) % 31u == 0 && (in[1] & 0x20u) == 0) {
    start = 2;                                                 // zlib stream (RFC 1950), no preset dictionary
    trailer = 4;
  } else {
    err = 1;
  }
  // (n_in - trailer unconditionally: trailer is 0 on the error paths.  `err ? 0 : n_in - trailer` compiled to s_add_i32 +
  //  s_cselect on the add's own SCC with this toolchain — always 0: an empty window, every page "corrupt")
  InflateBits b{in, lds.win, n_in - trailer, start, start, 0, 0, lane, 0};
  uint16_t* lit_lut = lds.lit_lut;
  uint16_t* dist_lut = lds.dist_lut;
  uint8_t* lengths = lds.lengths;
  uint8_t* staged = lds.staged;
  uint32_t op = 0;
  uint8_t* ring = lds.ring;
  if (!err) inflate_stage(b);
  bool last = false;
  while (!err && !last) {
    last = inflate_take(b, 1) != 0;
    const uint32_t type = inflate_take(b, 2);
    if (type == 0) {
      // stored: to the next byte boundary, LEN, ~LEN, LEN bytes
      const int drop = b.cnt & 7;
      b.buf >>= drop;
      b.cnt -= drop;
      const uint32_t len = inflate_take(b, 16), nlen = inflate_take(b, 16);
      if ((len ^ 0xFFFFu) != nlen) { err = 4; break; }
      const uint32_t from = inflate_consumed(b);
      if (from > b.n_in || len > b.n_in - from || len > ulen - op) { err = 2; break; }
      for (uint32_t j = lane; j < len; j += 64) {
        const uint8_t v = in[from + j];
        out[op + j] = v;
        ring[(op + j) & (kInfRing - 1)] = v;     // (a block longer than the ring overwrites itself in order: len <= 65535)
      }
      op += len;
      __builtin_amdgcn_wave_barrier();
      b.ip = from + len;       // the bit buffer is dropped: the next block starts at a byte boundary
      b.buf = 0;
      b.cnt = 0;
      inflate_stage(b);
      continue;
    }
    if (type == 3) { err = 4; break; }
    int nlit = 288, ndist = 30;
    if (type == 1) {
      // fixed code (RFC 1951 3.2.6): 8 bits for 0-143, 9 for 144-255, 7 for 256-279, 8 for 280-287; 5 bits for every distance
      __builtin_amdgcn_wave_barrier();
      for (int s = lane; s < 288; s += 64) lengths[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
      for (int s = lane; s < 32; s += 64) lengths[288 + s] = 5;
      ndist = 32;
      __builtin_amdgcn_wave_barrier();
    } else {
      nlit = static_cast<int>(inflate_take(b, 5)) + 257;
      ndist = static_cast<int>(inflate_take(b, 5)) + 1;
      const int ncode = static_cast<int>(inflate_take(b, 4)) + 4;
      if (nlit > 286 || ndist > 30) { err = 4; break; }
      // the code-length code: 3 bits each, in the order of 3.2.7
      const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      __builtin_amdgcn_wave_barrier();
      if (lane < 19) lengths[lane] = 0;
      __builtin_amdgcn_wave_barrier();
      for (int k = 0; k < ncode; ++k) {
        const uint32_t l = inflate_take(b, 3);
        if (lane == 0) lengths[order[k]] = static_cast<uint8_t>(l);
      }
      __builtin_amdgcn_wave_barrier();
      // (its tables live where the distance code's will: both are rebuilt below)
      err = inflate_build(lengths, 19, lds.dist_count, lds.dist_sym, dist_lut, 7, lane);
      if (err) break;
      int at = 0;
      uint32_t prev = 0;
      while (at < nlit + ndist && !err) {
        const uint32_t sym = inflate_symbol(b, dist_lut, 7, lds.dist_count, lds.dist_sym);
        uint32_t value = 0, repeat = 1;
        if (sym < 16) {
          value = sym;
        } else if (sym == 16) {
          if (at == 0) { err = 4; break; }
          value = prev;
          repeat = 3 + inflate_take(b, 2);
        } else if (sym == 17) {
          repeat = 3 + inflate_take(b, 3);
        } else if (sym == 18) {
          repeat = 11 + inflate_take(b, 7);
        } else {
          err = 4;
          break;
        }
        if (at + static_cast<int>(repeat) > nlit + ndist) { err = 4; break; }
        for (uint32_t r = 0; r < repeat; ++r, ++at) {
          if (lane == 0) staged[at] = static_cast<uint8_t>(value);
        }
        prev = value;
      }
      if (err) break;
      __builtin_amdgcn_wave_barrier();
      // lengths[0, nlit) literal / length, lengths[288, 288 + ndist) distance; the end-of-block symbol needs a code
      for (int s = lane; s < 288; s += 64) lengths[s] = s < nlit ? staged[s] : 0;
      for (int s = lane; s < 32; s += 64) lengths[288 + s] = s < ndist ? staged[nlit + s] : 0;
      __builtin_amdgcn_wave_barrier();
      if (lengths[256] == 0) { err = 4; break; }
      nlit = 288;
      ndist = 32;
    }
    err = inflate_build(lengths, nlit, lds.lit_count, lds.lit_sym, lit_lut, kInfLitBits, lane);
    if (err) break;
    err = inflate_build(lengths + 288, ndist, lds.dist_count, lds.dist_sym, dist_lut, kInfDistBits, lane);
    if (err) break;
    // ---- the block's symbols
    for (;;) {
      const uint32_t sym = inflate_symbol(b, lit_lut, kInfLitBits, lds.lit_count, lds.lit_sym);
      if (sym < 256) {
        if (op >= ulen) { err = 2; break; }
        if (lane == 0) {
          out[op] = static_cast<uint8_t>(sym);
          ring[op & (kInfRing - 1)] = static_cast<uint8_t>(sym);
        }
        ++op;
        continue;
      }
      if (sym == 256) break;
      if (sym > 285) { err = 4; break; }
      // length: base + extra bits (3.2.5)
      const uint32_t ls = sym - 257;
      uint32_t len, lextra;
      if (ls < 8) { len = 3 + ls; lextra = 0; }
      else if (ls == 28) { len = 258; lextra = 0; }
      else { lextra = (ls >> 2) - 1; len = 3 + ((4u + (ls & 3u)) << lextra); }
      if (lextra) len += inflate_take(b, static_cast<int>(lextra));
      const uint32_t ds = inflate_symbol(b, dist_lut, kInfDistBits, lds.dist_count, lds.dist_sym);
      if (ds > 29) { err = 4; break; }
      uint32_t dist, dextra;
      if (ds < 4) { dist = 1 + ds; dextra = 0; }
      else { dextra = (ds >> 1) - 1; dist = 1 + ((2u + (ds & 1u)) << dextra); }
      if (dextra) dist += inflate_take(b, static_cast<int>(dextra));
      if (dist > op) { err = 3; break; }
      if (len > ulen - op) { err = 2; break; }
      // the source is in the ring, whatever the distance; every lane past its earlier ring stores before any lane reads
      // (lockstep on the GPU; the SIMT emulator runs lanes in any order between rendezvous points)
      __builtin_amdgcn_wave_barrier();
      const uint32_t from = op - dist;
      for (uint32_t j0 = 0; j0 < len; j0 += 64) {
        const uint32_t j = j0 + static_cast<uint32_t>(lane);
        const bool mine = j < len;
        const uint8_t v = mine ? ring[(from + (dist >= len ? j : j % dist)) & (kInfRing - 1)] : 0;
        __builtin_amdgcn_wave_barrier();           // (a destination slot may be another lane's source slot: dist near the ring size)
        if (mine) {
          ring[(op + j) & (kInfRing - 1)] = v;
          out[op + j] = v;
        }
        __builtin_amdgcn_wave_barrier();
      }
      op += len;
    }
  }
  if (!err && (op != ulen || inflate_consumed(b) > b.n_in)) err = op != ulen ? 1 : 2;
  if (lane == 0) status[pg] = err;
}

}  // namespace arx

using namespace arx;

extern "C" {

// Host-side walk over the run headers (no device work): the only sequential part of the hybrid.
// `equals`: the value *ones counts (bit width 1: popcounts of the literal bytes; wider levels: value by value)
static int rle_scan_runs_impl(const void* data, size_t nbytes, int bit_width, int64_t num_values, uint32_t out_base,
                              uint64_t byte_base, ArxRleRun* runs, int64_t max_runs, int64_t* num_runs, int64_t* ones,
                              uint32_t equals) {
  if (num_values < 0 || bit_width < 0 || bit_width > 32 || num_runs == nullptr || (num_values > 0 && data == nullptr)) {
    set_error("bad arguments to arx_rle_scan_runs");
    return ARX_INVALID;
  }
  const uint8_t* p = static_cast<const uint8_t*>(data);
  const int vbytes = (bit_width + 7) / 8;
  size_t pos = 0;
  int64_t done = 0, nr = 0, count_ones = 0;
  while (done < num_values) {
    uint64_t h = 0;
    int shift = 0;
    for (;;) {
      if (pos >= nbytes || shift > 56) {
        set_error("Parquet: RLE block ended before all its values (corrupt data page?)");
        return ARX_INVALID;
      }
      const uint8_t c = p[pos++];
      h |= static_cast<uint64_t>(c & 0x7F) << shift;
      if ((c & 0x80) == 0) break;
      shift += 7;
    }
    if (runs != nullptr && nr >= max_runs) {
      set_error("arx_rle_scan_runs: more than %lld runs", static_cast<long long>(max_runs));
      return ARX_INVALID;
    }
    if (h & 1) {  // literal run: (h >> 1) groups of 8 bit-packed values
      // the group count is bounded before it is multiplied (a 2^62-group header must not wrap the byte count), and
      // only the run that reaches num_values may be cut short — by exactly the bytes its needed values do not use
      // (RleBitPackedDecoder reads no further either, rle_encoding_internal.h)
      if ((h >> 1) == 0 || (h >> 1) > (uint64_t(1) << 40)) {
        set_error("Parquet: literal run exceeds the RLE block (corrupt data page?)");
        return ARX_INVALID;
      }
      const int64_t groups = static_cast<int64_t>(h >> 1);
      const int64_t count = groups * 8;
      const size_t lbytes = static_cast<size_t>(groups) * bit_width;
      const int64_t needed = std::min<int64_t>(count, num_values - done);
      const size_t need_bytes = (static_cast<size_t>(needed) * bit_width + 7) / 8;
      const bool reaches_end = done + count >= num_values;
      if (pos > nbytes || (reaches_end ? need_bytes : lbytes) > nbytes - pos) {
        set_error("Parquet: literal run exceeds the RLE block (corrupt data page?)");
        return ARX_INVALID;
      }
      if (runs != nullptr) runs[nr] = ArxRleRun{static_cast<uint32_t>(out_base + done), 1u, byte_base + pos};
      if (ones != nullptr && bit_width == 1 && equals == 1) {
        const int64_t take = std::min<int64_t>(count, num_values - done);
        for (int64_t b = 0; b < take; b += 8) {
          uint8_t byte = (pos + (b >> 3)) < nbytes ? p[pos + (b >> 3)] : 0;
          if (take - b < 8) byte &= static_cast<uint8_t>((1u << (take - b)) - 1u);
          count_ones += __builtin_popcount(byte);
        }
      } else if (ones != nullptr && bit_width >= 1 && bit_width <= 16) {
        const int64_t take = std::min<int64_t>(count, num_values - done);
        const uint32_t vmask = (1u << bit_width) - 1u;
        for (int64_t k = 0; k < take; ++k) {
          const uint64_t bit = static_cast<uint64_t>(k) * bit_width;
          uint32_t acc = 0;
          for (int b = 0; b < 3; ++b) {   // (bit & 7) + 16 bits <= 3 bytes
            const size_t at = pos + (bit >> 3) + b;
            acc |= static_cast<uint32_t>(at < nbytes ? p[at] : 0) << (8 * b);
          }
          if (((acc >> (bit & 7)) & vmask) == equals) ++count_ones;
        }
      }
      pos += lbytes;
      done += count;
    } else {  // repeated run
      const int64_t count = static_cast<int64_t>(h >> 1);
      if (count == 0 || pos + vbytes > nbytes) {
        set_error("Parquet: bad repeated run (corrupt data page?)");
        return ARX_INVALID;
      }
      uint64_t value = 0;
      for (int k = 0; k < vbytes; ++k) value |= static_cast<uint64_t>(p[pos + k]) << (8 * k);
      pos += vbytes;
      if (runs != nullptr) runs[nr] = ArxRleRun{static_cast<uint32_t>(out_base + done), 0u, value};
      if (ones != nullptr && bit_width <= 16 && value == equals) count_ones += std::min<int64_t>(count, num_values - done);
      done += count;
    }
    ++nr;
  }
  *num_runs = nr;
  if (ones != nullptr) *ones = count_ones;
  return ARX_OK;
}

int arx_rle_scan_runs(const void* data, size_t nbytes, int bit_width, int64_t num_values, uint32_t out_base,
                      uint64_t byte_base, ArxRleRun* runs, int64_t max_runs, int64_t* num_runs, int64_t* ones) {
  if (bit_width != 1 && ones != nullptr) *ones = 0;
  return rle_scan_runs_impl(data, nbytes, bit_width, num_values, out_base, byte_base, runs, max_runs, num_runs,
                            bit_width == 1 ? ones : nullptr, 1u);
}

int arx_rle_scan_runs_equals(const void* data, size_t nbytes, int bit_width, int64_t num_values, uint32_t equals,
                             uint32_t out_base, uint64_t byte_base, ArxRleRun* runs, int64_t max_runs, int64_t* num_runs,
                             int64_t* count) {
  if (bit_width > 16 && count != nullptr) {
    set_error("arx_rle_scan_runs_equals counts levels of at most 16 bits");
    return ARX_INVALID;
  }
  if (bit_width == 0 && count != nullptr) {   // a block of zero-width values: every value is 0
    const int rc = rle_scan_runs_impl(data, nbytes, bit_width, num_values, out_base, byte_base, runs, max_runs, num_runs,
                                      nullptr, 0u);
    if (rc == ARX_OK) *count = equals == 0 ? num_values : 0;
    return rc;
  }
  return rle_scan_runs_impl(data, nbytes, bit_width, num_values, out_base, byte_base, runs, max_runs, num_runs, count,
                            equals);
}

size_t arx_levels_to_list_workspace_bytes(int64_t num_levels) {
  return static_cast<size_t>(ceil_div(std::max<int64_t>(num_levels, 1), kListTile)) * sizeof(long long) + 64;
}

int arx_def_rep_levels_to_list(const uint32_t* def_levels, const uint32_t* rep_levels, int64_t num_levels, int def_level,
                               int rep_level, int repeated_ancestor_def_level, int64_t max_entries, int32_t* offsets,
                               void* valid_bits, uint64_t* counts, void* ws, size_t ws_bytes, void* stream) {
  if (num_levels < 0 || num_levels > INT32_MAX || max_entries < 0 || offsets == nullptr || counts == nullptr ||
      def_level < 1 || rep_level < 0 || repeated_ancestor_def_level < 0 || (num_levels > 0 && def_levels == nullptr) ||
      (rep_level > 0 && num_levels > 0 && rep_levels == nullptr)) {
    set_error("bad arguments to arx_def_rep_levels_to_list");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  ARX_HIP(hipMemsetAsync(counts, 0, 4 * sizeof(uint64_t), st));
  ARX_HIP(hipMemsetAsync(offsets, 0, sizeof(int32_t), st));          // offsets[0] (and the whole answer for no level)
  if (num_levels == 0) return ARX_OK;
  const int64_t ntiles = ceil_div(num_levels, kListTile);
  if (ws == nullptr || ws_bytes < static_cast<size_t>(ntiles) * sizeof(long long)) {
    set_error("arx_def_rep_levels_to_list: workspace too small");
    return ARX_INVALID;
  }
  long long* sums = static_cast<long long*>(ws);
  const ListLevels lv{def_levels, rep_levels, static_cast<uint32_t>(def_level), static_cast<uint32_t>(rep_level),
                      static_cast<uint32_t>(repeated_ancestor_def_level)};
  hipLaunchKernelGGL(list_tile_sums_kernel, dim3(static_cast<unsigned>(ntiles)), dim3(kBlock), 0, st, lv, num_levels, sums);
  ARX_CHECK_LAUNCH("list_tile_sums_kernel");
  hipLaunchKernelGGL(delta_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, sums, ntiles);
  ARX_CHECK_LAUNCH("delta_scan_tiles_kernel");
  hipLaunchKernelGGL(list_write_kernel, dim3(static_cast<unsigned>(ntiles)), dim3(kBlock), 0, st, lv, num_levels,
                     static_cast<const long long*>(sums), max_entries, offsets, static_cast<uint32_t*>(valid_bits),
                     reinterpret_cast<unsigned long long*>(counts));
  ARX_CHECK_LAUNCH("list_write_kernel");
  return ARX_OK;
}

int arx_levels_ge_bitmap(const uint32_t* levels, int64_t num_levels, uint32_t threshold, void* out_bits, uint64_t* ones,
                         void* stream) {
  if (num_levels < 0 || (num_levels > 0 && (levels == nullptr || out_bits == nullptr))) {
    set_error("bad arguments to arx_levels_ge_bitmap");
    return ARX_INVALID;
  }
  if (num_levels == 0) return ARX_OK;
  const int64_t nwords = ceil_div(num_levels, 64);
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(nwords, kWavesPerBlock), 256 * 16)));
  hipLaunchKernelGGL(levels_ge_bitmap_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), levels, num_levels, threshold,
                     static_cast<uint64_t*>(out_bits), reinterpret_cast<unsigned long long*>(ones));
  ARX_CHECK_LAUNCH("levels_ge_bitmap_kernel");
  return ARX_OK;
}

// Host-side walk over PLAIN BYTE_ARRAY values (4-byte little-endian length + bytes each,
// PlainByteArrayDecoder, cpp/src/parquet/decoder.cc): writes 2 * count + 1 offsets that describe the
// block as alternating {length prefix, value} entries, so that a var-width take of the odd entries
// (arx_binary_take_*) compacts the values on the device.
int arx_plain_byte_array_offsets(const void* data, size_t nbytes, int64_t count, int32_t base, int32_t* out_offsets) {
  if (count < 0 || out_offsets == nullptr || (count > 0 && data == nullptr)) {
    set_error("bad arguments to arx_plain_byte_array_offsets");
    return ARX_INVALID;
  }
  const uint8_t* p = static_cast<const uint8_t*>(data);
  size_t pos = 0;
  out_offsets[0] = base;
  for (int64_t i = 0; i < count; ++i) {
    if (pos + 4 > nbytes) {
      set_error("Parquet: byte-array value %lld starts past the page (corrupt page?)", static_cast<long long>(i));
      return ARX_INVALID;
    }
    const uint32_t len = static_cast<uint32_t>(p[pos]) | (static_cast<uint32_t>(p[pos + 1]) << 8) |
                         (static_cast<uint32_t>(p[pos + 2]) << 16) | (static_cast<uint32_t>(p[pos + 3]) << 24);
    if (len > nbytes - pos - 4 || static_cast<uint64_t>(base) + pos + 4 + len > 2147483647ull) {
      set_error("Parquet: byte-array value %lld runs past the page or the int32 offset range", static_cast<long long>(i));
      return ARX_INVALID;
    }
    out_offsets[2 * i + 1] = base + static_cast<int32_t>(pos + 4);
    pos += 4 + len;
    out_offsets[2 * i + 2] = base + static_cast<int32_t>(pos);
  }
  return ARX_OK;
}

// Host-side walk over the headers of one DELTA_BINARY_PACKED block sequence (no device work).  Fills one
// ArxDeltaMiniblock per miniblock that holds values; `byte_base` = where `data` sits in the buffer the device
// will see (its bit positions are absolute in that buffer).  Returns the header fields and the bytes consumed
// (DELTA_BYTE_ARRAY-style callers continue after them).
int arx_delta_scan_miniblocks(const void* data, size_t nbytes, uint64_t byte_base, ArxDeltaMiniblock* out,
                              int64_t max_miniblocks, int64_t* num_miniblocks, int64_t* values_per_miniblock,
                              int64_t* total_values, int64_t* first_value, size_t* bytes_consumed) {
  if (data == nullptr || num_miniblocks == nullptr || values_per_miniblock == nullptr || total_values == nullptr ||
      first_value == nullptr) {
    set_error("bad arguments to arx_delta_scan_miniblocks");
    return ARX_INVALID;
  }
  const uint8_t* p = static_cast<const uint8_t*>(data);
  size_t pos = 0;
  bool bad = false;
  auto varint = [&]() -> uint64_t {
    uint64_t v = 0;
    for (int shift = 0; shift < 70; shift += 7) {
      if (pos >= nbytes) { bad = true; return 0; }
      const uint8_t c = p[pos++];
      v |= static_cast<uint64_t>(c & 0x7F) << shift;
      if ((c & 0x80) == 0) return v;
    }
    bad = true;
    return 0;
  };
  auto zigzag = [&]() -> int64_t {
    const uint64_t u = varint();
    return static_cast<int64_t>((u >> 1) ^ (~(u & 1) + 1));
  };
  const uint64_t block_size = varint();
  const uint64_t per_block = varint();
  const uint64_t total = varint();
  const int64_t first = zigzag();
  // DeltaBitPackDecoder reads the three header fields as uint32 (decoder.cc); a block of more than 2^20 values per
  // miniblock is no writer's output, and bounding it keeps every byte count below far from wrapping
  if (bad || per_block == 0 || block_size == 0 || block_size > 0xFFFFFFFFull || per_block > 0xFFFFFFFFull ||
      total > 0xFFFFFFFFull || block_size % 128 != 0 || block_size % per_block != 0 ||
      (block_size / per_block) % 32 != 0 || block_size / per_block > (uint64_t(1) << 20)) {
    set_error("Parquet: bad DELTA_BINARY_PACKED header (corrupt data page?)");
    return ARX_INVALID;
  }
  const int64_t vpm = static_cast<int64_t>(block_size / per_block);
  int64_t remaining = static_cast<int64_t>(total) - 1, nmb = 0;
  while (remaining > 0) {
    const int64_t min_delta = zigzag();
    if (bad || pos + per_block > nbytes) {
      set_error("Parquet: DELTA_BINARY_PACKED block header runs past the page (corrupt data page?)");
      return ARX_INVALID;
    }
    const uint8_t* widths = p + pos;
    pos += per_block;
    for (uint64_t m = 0; m < per_block && remaining > 0; ++m) {
      const uint32_t bw = widths[m];
      const size_t mbytes = static_cast<size_t>(vpm) * bw / 8;   // vpm <= 2^20, bw <= 64: no wrap
      // the last miniblock need not be padded to its full size: only the deltas it still holds must be there
      // (the reference decodes no more than it needs)
      const size_t need = remaining >= vpm ? mbytes : (static_cast<size_t>(remaining) * bw + 7) / 8;
      if (bw > 64 || pos > nbytes || need > nbytes - pos) {
        set_error("Parquet: DELTA_BINARY_PACKED miniblock runs past the page (corrupt data page?)");
        return ARX_INVALID;
      }
      if (out != nullptr) {
        if (nmb >= max_miniblocks) {
          set_error("arx_delta_scan_miniblocks: more than %lld miniblocks", static_cast<long long>(max_miniblocks));
          return ARX_INVALID;
        }
        out[nmb] = ArxDeltaMiniblock{(byte_base + pos) * 8, min_delta, bw, 0u};
      }
      ++nmb;
      pos += std::min(mbytes, nbytes - pos);
      remaining -= vpm;
    }
  }
  *num_miniblocks = nmb;
  *values_per_miniblock = vpm;
  *total_values = static_cast<int64_t>(total);
  *first_value = first;
  if (bytes_consumed != nullptr) *bytes_consumed = pos;
  return ARX_OK;
}

size_t arx_delta_decode_workspace_bytes(int64_t num_values) {
  return (static_cast<size_t>(ceil_div(std::max<int64_t>(num_values, 1), kDeltaTile)) * 8 + 63) & ~static_cast<size_t>(63);
}

// bytes: the page buffer on the device, 8-byte aligned and readable 8 bytes past its last miniblock.
int arx_delta_decode(const void* bytes, const ArxDeltaMiniblock* miniblocks, int64_t num_miniblocks,
                     int64_t values_per_miniblock, int64_t first_value, int64_t num_values, int out_byte_width,
                     void* ws, size_t ws_bytes, void* out, void* stream) {
  if (num_values < 0 || num_miniblocks < 0 || (out_byte_width != 4 && out_byte_width != 8) ||
      (reinterpret_cast<uintptr_t>(bytes) & 7) != 0) {
    set_error("bad arguments to arx_delta_decode");
    return ARX_INVALID;
  }
  if (num_values == 0) return ARX_OK;
  if (out == nullptr || ws == nullptr || ws_bytes < arx_delta_decode_workspace_bytes(num_values) ||
      (num_values > 1 && (bytes == nullptr || miniblocks == nullptr || values_per_miniblock <= 0 ||
                          num_miniblocks * values_per_miniblock < num_values - 1))) {
    set_error("arx_delta_decode: %lld values need their miniblocks, an output and %zu workspace bytes",
              static_cast<long long>(num_values), arx_delta_decode_workspace_bytes(num_values));
    return ARX_INVALID;
  }
  const int64_t ntiles = ceil_div(num_values, kDeltaTile);
  const DeltaSource src{static_cast<const uint64_t*>(bytes), miniblocks, values_per_miniblock, static_cast<long long>(first_value)};
  long long* sums = static_cast<long long*>(ws);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL((delta_tile_sums_kernel<DeltaSource>), dim3(static_cast<unsigned>(ntiles)), dim3(kBlock), 0, st, src,
                     num_values, sums);
  ARX_CHECK_LAUNCH("delta_tile_sums_kernel");
  hipLaunchKernelGGL(delta_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, sums, ntiles);
  ARX_CHECK_LAUNCH("delta_scan_tiles_kernel");
  if (out_byte_width == 8) {
    hipLaunchKernelGGL((delta_write_kernel<long long, DeltaSource>), dim3(static_cast<unsigned>(ntiles)), dim3(kBlock), 0, st,
                       src, num_values, sums, static_cast<long long*>(out));
  } else {
    hipLaunchKernelGGL((delta_write_kernel<int32_t, DeltaSource>), dim3(static_cast<unsigned>(ntiles)), dim3(kBlock), 0, st,
                       src, num_values, sums, static_cast<int32_t*>(out));
  }
  ARX_CHECK_LAUNCH("delta_write_kernel");
  return ARX_OK;
}

// The pages of a whole column chunk: `pages` (device) describes them, page p covering tiles [first_tile, next page's
// first_tile) of 4096 values; total_tiles = their sum.  ws: arx_delta_decode_workspace_bytes(total_tiles * 4096).
int arx_delta_decode_pages(const void* bytes, const ArxDeltaMiniblock* miniblocks, const ArxDeltaPage* pages, int64_t num_pages,
                           int64_t total_tiles, int out_byte_width, void* ws, size_t ws_bytes, void* out, void* stream) {
  if (num_pages < 0 || total_tiles < 0 || (out_byte_width != 4 && out_byte_width != 8) ||
      (reinterpret_cast<uintptr_t>(bytes) & 7) != 0) {
    set_error("bad arguments to arx_delta_decode_pages");
    return ARX_INVALID;
  }
  if (num_pages == 0 || total_tiles == 0) return ARX_OK;
  if (pages == nullptr || out == nullptr || ws == nullptr || total_tiles > (int64_t(1) << 31) - 1 ||
      ws_bytes < arx_delta_decode_workspace_bytes(total_tiles * kDeltaTile)) {
    set_error("arx_delta_decode_pages: %lld tiles need a page table, an output and %zu workspace bytes",
              static_cast<long long>(total_tiles), arx_delta_decode_workspace_bytes(total_tiles * kDeltaTile));
    return ARX_INVALID;
  }
  const uint64_t* words = static_cast<const uint64_t*>(bytes);
  long long* sums = static_cast<long long*>(ws);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(delta_pages_tile_sums_kernel, dim3(static_cast<unsigned>(total_tiles)), dim3(kBlock), 0, st, words,
                     miniblocks, pages, num_pages, sums);
  ARX_CHECK_LAUNCH("delta_pages_tile_sums_kernel");
  hipLaunchKernelGGL(delta_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, sums, total_tiles);
  ARX_CHECK_LAUNCH("delta_scan_tiles_kernel");
  if (out_byte_width == 8) {
    hipLaunchKernelGGL((delta_pages_write_kernel<long long>), dim3(static_cast<unsigned>(total_tiles)), dim3(kBlock), 0, st,
                       words, miniblocks, pages, num_pages, sums, static_cast<long long*>(out));
  } else {
    hipLaunchKernelGGL((delta_pages_write_kernel<int32_t>), dim3(static_cast<unsigned>(total_tiles)), dim3(kBlock), 0, st, words,
                       miniblocks, pages, num_pages, sums, static_cast<int32_t*>(out));
  }
  ARX_CHECK_LAUNCH("delta_pages_write_kernel");
  return ARX_OK;
}

// lengths[0..n) -> out[0..n]: out[0] = base, out[i] = base + lengths[0] + ... + lengths[i-1] (int32 offsets of a
// utf8 / binary array).  ws: arx_delta_decode_workspace_bytes(n + 1).
int arx_lengths_to_offsets_i32(const int32_t* lengths, int64_t n, int32_t base, int32_t* out, void* ws, size_t ws_bytes,
                               void* stream) {
  if (n < 0 || out == nullptr || (n > 0 && lengths == nullptr) || ws == nullptr ||
      ws_bytes < arx_delta_decode_workspace_bytes(n + 1)) {
    set_error("bad arguments to arx_lengths_to_offsets_i32 (%lld lengths, %zu workspace bytes needed)",
              static_cast<long long>(n), arx_delta_decode_workspace_bytes(std::max<int64_t>(n, 0) + 1));
    return ARX_INVALID;
  }
  const int64_t count = n + 1;
  const int64_t ntiles = ceil_div(count, kDeltaTile);
  const LengthSource src{lengths, static_cast<long long>(base)};
  long long* sums = static_cast<long long*>(ws);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL((delta_tile_sums_kernel<LengthSource>), dim3(static_cast<unsigned>(ntiles)), dim3(kBlock), 0, st, src,
                     count, sums);
  ARX_CHECK_LAUNCH("delta_tile_sums_kernel");
  hipLaunchKernelGGL(delta_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, sums, ntiles);
  ARX_CHECK_LAUNCH("delta_scan_tiles_kernel");
  hipLaunchKernelGGL((delta_write_kernel<int32_t, LengthSource>), dim3(static_cast<unsigned>(ntiles)), dim3(kBlock), 0, st, src,
                     count, sums, out);
  ARX_CHECK_LAUNCH("delta_write_kernel");
  return ARX_OK;
}

int arx_delta_byte_array_lengths(const int32_t* prefix, const int32_t* suffix_len, int64_t n, const int64_t* page_first,
                                 int64_t num_pages, int32_t* out_len, uint64_t* state, void* stream) {
  if (n < 0 || num_pages < 0 || state == nullptr || (n > 0 && (prefix == nullptr || suffix_len == nullptr || out_len == nullptr ||
                                                                 page_first == nullptr || num_pages == 0))) {
    set_error("bad arguments to arx_delta_byte_array_lengths (%lld values, %lld pages)", static_cast<long long>(n),
              static_cast<long long>(num_pages));
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  ARX_HIP(hipMemsetAsync(state, 0, 16, st));
  if (n == 0) return ARX_OK;
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, kBlock), 256 * 8)));
  hipLaunchKernelGGL(dba_lengths_kernel, dim3(grid), dim3(kBlock), 0, st, prefix, suffix_len, n, page_first, num_pages, out_len,
                     reinterpret_cast<unsigned long long*>(state));
  ARX_CHECK_LAUNCH("dba_lengths_kernel");
  return ARX_OK;
}

int arx_delta_byte_array_expand(const int32_t* prefix, const int32_t* suffix_offsets, const void* suffix_bytes, int64_t suffix_size,
                                const int32_t* out_offsets, int32_t out_base, const int64_t* page_first,
                                const int64_t* page_suffix_first, int64_t num_pages, void* out_data, uint64_t* state, void* stream) {
  if (num_pages < 0 || suffix_size < 0 || (num_pages > 0 && (prefix == nullptr || suffix_offsets == nullptr || out_offsets == nullptr ||
                                                               page_first == nullptr || (page_suffix_first != nullptr && state == nullptr)))) {
    set_error("bad arguments to arx_delta_byte_array_expand (%lld pages)", static_cast<long long>(num_pages));
    return ARX_INVALID;
  }
  if (num_pages == 0) return ARX_OK;
  if ((reinterpret_cast<uintptr_t>(suffix_bytes) & 3) != 0) {
    set_error("arx_delta_byte_array_expand: the suffix bytes must start on a 4-byte boundary");
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(dba_expand_kernel, dim3(static_cast<unsigned>(num_pages)), dim3(64), 0, as_stream(stream), prefix, suffix_offsets,
                     static_cast<const uint8_t*>(suffix_bytes), suffix_size, out_offsets, out_base, page_first, page_suffix_first,
                     reinterpret_cast<unsigned long long*>(state), static_cast<uint8_t*>(out_data));
  ARX_CHECK_LAUNCH("dba_expand_kernel");
  return ARX_OK;
}

int arx_byte_stream_split_decode(const void* in, int64_t num_values, int byte_width, void* out, void* stream) {
  if (num_values < 0 || (byte_width != 2 && byte_width != 4 && byte_width != 8)) {
    set_error("arx_byte_stream_split_decode: %lld values of width %d", static_cast<long long>(num_values), byte_width);
    return num_values < 0 ? ARX_INVALID : ARX_NOT_IMPLEMENTED;
  }
  if (num_values == 0) return ARX_OK;
  if (in == nullptr || out == nullptr || (reinterpret_cast<uintptr_t>(out) % byte_width) != 0) {
    set_error("arx_byte_stream_split_decode: NULL or misaligned buffer");
    return ARX_INVALID;
  }
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(num_values, kBlock), 256 * 8)));
  const uint8_t* src = static_cast<const uint8_t*>(in);
  uint8_t* dst = static_cast<uint8_t*>(out);
  hipStream_t st = as_stream(stream);
  if (byte_width == 8) {
    hipLaunchKernelGGL((byte_stream_split_kernel<8>), dim3(grid), dim3(kBlock), 0, st, src, num_values, dst);
  } else if (byte_width == 4) {
    hipLaunchKernelGGL((byte_stream_split_kernel<4>), dim3(grid), dim3(kBlock), 0, st, src, num_values, dst);
  } else {
    hipLaunchKernelGGL((byte_stream_split_kernel<2>), dim3(grid), dim3(kBlock), 0, st, src, num_values, dst);
  }
  ARX_CHECK_LAUNCH("byte_stream_split_kernel");
  return ARX_OK;
}

int arx_rle_decode_u32(const void* bytes, size_t nbytes, const ArxRleRun* runs, int64_t nruns, int bit_width,
                       int64_t num_values, uint32_t* out, void* stream) {
  const int rc = rle_check(bytes, runs, nruns, bit_width, num_values);
  if (rc != ARX_OK) return rc;
  if (num_values == 0) return ARX_OK;
  if (out == nullptr) {
    set_error("RLE decode: out is NULL");
    return ARX_INVALID;
  }
  const int64_t nwords = ceil_div(num_values, 64);
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(nwords, kWavesPerBlock), 256 * 16)));
  hipLaunchKernelGGL((rle_decode_kernel<false>), dim3(grid), dim3(kBlock), 0, as_stream(stream),
                     static_cast<const uint8_t*>(bytes), static_cast<uint64_t>(nbytes), runs, nruns, bit_width,
                     num_values, 0u, out, static_cast<uint64_t*>(nullptr));
  ARX_CHECK_LAUNCH("rle_decode_kernel");
  return ARX_OK;
}

int arx_rle_decode_equals_bitmap(const void* bytes, size_t nbytes, const ArxRleRun* runs, int64_t nruns,
                                 int bit_width, int64_t num_values, uint32_t equals, void* out_bits, void* stream) {
  const int rc = rle_check(bytes, runs, nruns, bit_width, num_values);
  if (rc != ARX_OK) return rc;
  if (num_values == 0) return ARX_OK;
  if (out_bits == nullptr) {
    set_error("RLE decode: out_bits is NULL");
    return ARX_INVALID;
  }
  const int64_t nwords = ceil_div(num_values, 64);
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(nwords, kWavesPerBlock), 256 * 16)));
  hipLaunchKernelGGL((rle_decode_kernel<true>), dim3(grid), dim3(kBlock), 0, as_stream(stream),
                     static_cast<const uint8_t*>(bytes), static_cast<uint64_t>(nbytes), runs, nruns, bit_width,
                     num_values, equals, static_cast<uint32_t*>(nullptr), static_cast<uint64_t*>(out_bits));
  ARX_CHECK_LAUNCH("rle_decode_kernel");
  return ARX_OK;
}

// HOST: the blocks of one LZ4 frame (lz4_Frame_format.md): magic 0x184D2204, FLG (version 01, block independence,
// block checksum, content size, content checksum, dictionary id), BD, optional 8-byte content size, optional 4-byte
// dictionary id, header checksum byte; then {4-byte size (top bit: stored), data, optional 4-byte block checksum}
// until a zero size (EndMark), then the optional content checksum.  Checksums are skipped, not verified.
int arx_lz4_frame_scan(const void* data, size_t nbytes, uint64_t byte_base, ArxLz4Block* blocks, int64_t max_blocks,
                       int64_t* num_blocks, uint64_t* content_size) {
  if (data == nullptr || num_blocks == nullptr) {
    set_error("bad arguments to arx_lz4_frame_scan");
    return ARX_INVALID;
  }
  const uint8_t* p = static_cast<const uint8_t*>(data);
  auto le32 = [&](size_t at) {
    return static_cast<uint32_t>(p[at]) | (static_cast<uint32_t>(p[at + 1]) << 8) | (static_cast<uint32_t>(p[at + 2]) << 16) |
           (static_cast<uint32_t>(p[at + 3]) << 24);
  };
  if (nbytes < 7 || le32(0) != 0x184D2204u) {
    set_error("Lz4 compressed input contains less than one frame");   // (the reference's text for a body that is no frame)
    return ARX_INVALID;
  }
  const uint8_t flg = p[4];
  if ((flg >> 6) != 1) {
    set_error("LZ4 frame: unsupported version %d", flg >> 6);
    return ARX_NOT_IMPLEMENTED;
  }
  if (flg & 1) {
    set_error("LZ4 frame: dictionaries are not supported");
    return ARX_NOT_IMPLEMENTED;
  }
  const bool block_checksum = (flg >> 4) & 1, has_size = (flg >> 3) & 1;
  size_t pos = 6;
  uint64_t csize = 0;
  if (has_size) {
    if (pos + 8 > nbytes) {
      set_error("LZ4 frame: truncated header");
      return ARX_INVALID;
    }
    for (int k = 0; k < 8; ++k) csize |= static_cast<uint64_t>(p[pos + k]) << (8 * k);
    pos += 8;
  }
  pos += 1;   // header checksum
  int64_t nb = 0;
  for (;;) {
    if (pos + 4 > nbytes) {
      set_error("LZ4 frame: truncated before its end mark");
      return ARX_INVALID;
    }
    const uint32_t word = le32(pos);
    pos += 4;
    if (word == 0) break;   // EndMark
    const uint32_t size = word & 0x7FFFFFFFu;
    if (size > nbytes - pos) {
      set_error("LZ4 frame: a block runs past the buffer");
      return ARX_INVALID;
    }
    if (blocks != nullptr) {
      if (nb >= max_blocks) {
        set_error("arx_lz4_frame_scan: more than %lld blocks", static_cast<long long>(max_blocks));
        return ARX_INVALID;
      }
      blocks[nb] = ArxLz4Block{byte_base + pos, size, word >> 31};
    }
    ++nb;
    pos += size;
    if (block_checksum) pos += 4;
  }
  *num_blocks = nb;
  if (content_size != nullptr) *content_size = csize;
  return ARX_OK;
}

int arx_lz4_decompress_streams(const void* compressed, const ArxLz4Stream* streams, const ArxLz4Block* blocks,
                               int64_t num_streams, void* out, uint32_t* status, void* stream) {
  if (num_streams < 0) {
    set_error("bad arguments to arx_lz4_decompress_streams");
    return ARX_INVALID;
  }
  if (num_streams == 0) return ARX_OK;
  if (compressed == nullptr || streams == nullptr || blocks == nullptr || out == nullptr || status == nullptr) {
    set_error("NULL buffer passed to arx_lz4_decompress_streams");
    return ARX_INVALID;
  }
  const unsigned grid = static_cast<unsigned>(ceil_div(num_streams, kWavesPerBlock));
  hipLaunchKernelGGL(lz4_decode_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), static_cast<const uint8_t*>(compressed),
                     streams, blocks, num_streams, static_cast<uint8_t*>(out), status);
  ARX_CHECK_LAUNCH("lz4_decode_kernel");
  return ARX_OK;
}

int arx_rle_levels_to_bitmap(const void* bytes, const ArxLevelPage* pages, int64_t num_pages, void* out_bits,
                             uint32_t* ones, uint32_t* status, void* stream) {
  if (num_pages < 0) {
    set_error("bad arguments to arx_rle_levels_to_bitmap");
    return ARX_INVALID;
  }
  if (num_pages == 0) return ARX_OK;
  if (bytes == nullptr || pages == nullptr || out_bits == nullptr || ones == nullptr || status == nullptr ||
      (reinterpret_cast<uintptr_t>(out_bits) & 7) != 0) {
    set_error("arx_rle_levels_to_bitmap: NULL buffer or out_bits not 8-byte aligned");
    return ARX_INVALID;
  }
  const unsigned grid = static_cast<unsigned>(ceil_div(num_pages, kWavesPerBlock));
  hipLaunchKernelGGL(rle_levels_bitmap_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream),
                     static_cast<const uint8_t*>(bytes), pages, num_pages, static_cast<unsigned long long*>(out_bits), ones,
                     status);
  ARX_CHECK_LAUNCH("rle_levels_bitmap_kernel");
  return ARX_OK;
}

int arx_snappy_decompress_pages(const void* compressed, const ArxSnappyPage* pages, int64_t num_pages, void* out,
                                uint32_t* status, void* stream) {
  if (num_pages < 0) {
    set_error("bad arguments to arx_snappy_decompress_pages");
    return ARX_INVALID;
  }
  if (num_pages == 0) return ARX_OK;
  if (compressed == nullptr || pages == nullptr || out == nullptr || status == nullptr) {
    set_error("NULL buffer passed to arx_snappy_decompress_pages");
    return ARX_INVALID;
  }
  const unsigned grid = static_cast<unsigned>(ceil_div(num_pages, kWavesPerBlock));
  if (g_snappy_lds == 1 || (g_snappy_lds < 0 && num_pages < 4096)) {
    hipLaunchKernelGGL(snappy_decode_lds_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream),
                       static_cast<const uint8_t*>(compressed), pages, num_pages, static_cast<uint8_t*>(out), status);
  } else {
    hipLaunchKernelGGL(snappy_decode_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream),
                       static_cast<const uint8_t*>(compressed), pages, num_pages, static_cast<uint8_t*>(out), status);
  }
  ARX_CHECK_LAUNCH("snappy_decode_kernel");
  return ARX_OK;
}

int arx_gzip_decompress_pages(const void* compressed, const ArxSnappyPage* pages, int64_t num_pages, void* out,
                              uint32_t* status, void* stream) {
  if (num_pages < 0 || (num_pages > 0 && (compressed == nullptr || pages == nullptr || out == nullptr || status == nullptr))) {
    set_error("bad arguments to arx_gzip_decompress_pages");
    return ARX_INVALID;
  }
  if (num_pages == 0) return ARX_OK;
  if (num_pages > INT32_MAX) {
    set_error("arx_gzip_decompress_pages: more than 2^31 pages");
    return ARX_INVALID;
  }
  const unsigned grid = static_cast<unsigned>(num_pages);      // one wave (a workgroup of 64 threads) per page
  hipLaunchKernelGGL(inflate_pages_kernel, dim3(grid), dim3(kInfThreads), 0, as_stream(stream), static_cast<const uint8_t*>(compressed),
                     pages, num_pages, static_cast<uint8_t*>(out), status);
  ARX_CHECK_LAUNCH("inflate_pages_kernel");
  return ARX_OK;
}

}  // extern "C"
