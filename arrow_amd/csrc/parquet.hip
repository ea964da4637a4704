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
  if (r.kind == 0) return static_cast<uint32_t>(r.payload);
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

}  // namespace arx

using namespace arx;

extern "C" {

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

}  // extern "C"
