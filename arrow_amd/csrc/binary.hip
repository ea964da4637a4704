// Take (and therefore Filter) on base-binary values — binary / utf8 with int32 offsets — for gfx950.
//
// What it restates (semantics only):
//   TakeExec for base binary / VarBinaryTakeImpl   cpp/src/arrow/compute/kernels/vector_selection_take_internal.cc
//   BinaryFilterImpl (filter == take of GetTakeIndices for these types)
//                                                  cpp/src/arrow/compute/kernels/vector_selection_filter_internal.cc:517-800
// Out slot i is valid iff index i is valid and the source value is valid; a valid slot appends
// the source bytes, a null slot appends nothing; out offsets start at 0.
//
// Data layout: Arrow's variable-size binary layout as is (validity bitmap, int32 offsets[n+1],
// data bytes; docs/source/format/Columnar.rst).  Three kernels + one copy:
//   bin_lengths   : length of every output slot (0 for null slots), output validity by ballot,
//                   per-workgroup byte totals (64-bit)
//   bin_scan_blocks: exclusive scan of the workgroup totals (one workgroup)
//   bin_offsets   : exclusive scan inside each workgroup's 4096 slots + its base -> out offsets
//   bin_copy      : output-centric, one workgroup per 16 KiB of output bytes (dense dword stores)
#include "arx_common.h"

#include <algorithm>

namespace arx {

constexpr int kBinRows = kBlock * 16;  // output slots per workgroup

struct BinTakeArgs {
  const void* offsets;      // element 0 of the logical values array: int32 (utf8 / binary) or int64 (large_utf8 / large_binary) entries
  const uint8_t* data;
  const uint8_t* src_valid_bytes;  // source validity bitmap bytes or NULL
  int64_t src_valid_offset;
  const uint8_t* indices;   // pre-offset to element 0
  int index_type;
  Bits ivalid;
  int64_t length;           // number of indices
};

__device__ __forceinline__ uint64_t bin_load_index(const uint8_t* p, int type, int64_t i) {
  switch (type) {
    case ARX_UINT8: return p[i];
    case ARX_INT8: return static_cast<uint8_t>(reinterpret_cast<const int8_t*>(p)[i]);
    case ARX_UINT16: return reinterpret_cast<const uint16_t*>(p)[i];
    case ARX_INT16: return static_cast<uint16_t>(reinterpret_cast<const int16_t*>(p)[i]);
    case ARX_UINT32: return reinterpret_cast<const uint32_t*>(p)[i];
    case ARX_INT32: return static_cast<uint32_t>(reinterpret_cast<const int32_t*>(p)[i]);
    default: return reinterpret_cast<const uint64_t*>(p)[i];
  }
}

// slot i: is it valid, and which source row
__device__ __forceinline__ bool bin_slot(const BinTakeArgs& a, int64_t i, uint64_t* idx) {
  bool ok = (load_word(a.ivalid, i >> 6) >> (i & 63)) & 1ull;
  *idx = 0;
  if (ok) {
    *idx = bin_load_index(a.indices, a.index_type, i);
    if (a.src_valid_bytes != nullptr) {
      const uint64_t bit = static_cast<uint64_t>(a.src_valid_offset) + *idx;
      ok = (a.src_valid_bytes[bit >> 3] >> (bit & 7)) & 1;
    }
  }
  return ok;
}

// O: the offsets' type — int32_t, or int64_t for the large_ types (round 4: same kernels, wider offsets)
template <typename O>
__global__ __launch_bounds__(kBlock) void bin_lengths_kernel(BinTakeArgs a, O* __restrict__ lens,
                                                             O* __restrict__ src_start,
                                                             uint64_t* __restrict__ out_validity,
                                                             unsigned long long* __restrict__ valid_count,
                                                             long long* __restrict__ block_sums) {
  __shared__ long long wave_sum[kWavesPerBlock];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kBinRows;
  long long mine = 0;
  uint64_t nvalid = 0;
  // three passes over 16 slots held in registers so that the 16 index loads, then the 16 pairs of
  // dependent offset loads, are all in flight together
  bool ok[16];
  uint64_t idx[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int64_t i = base + it * kBlock + tid;
    ok[it] = i < a.length && ((load_word(a.ivalid, i >> 6) >> (i & 63)) & 1ull);
    idx[it] = ok[it] ? bin_load_index(a.indices, a.index_type, i) : 0;
  }
  if (a.src_valid_bytes != nullptr) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const uint64_t bit = static_cast<uint64_t>(a.src_valid_offset) + idx[it];
      ok[it] = ok[it] && ((a.src_valid_bytes[bit >> 3] >> (bit & 7)) & 1);
    }
  }
  O start[16], end[16];
  const O* __restrict__ offs = static_cast<const O*>(a.offsets);
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    start[it] = ok[it] ? offs[idx[it]] : 0;
    end[it] = ok[it] ? offs[idx[it] + 1] : 0;
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int64_t i = base + it * kBlock + tid;
    const O len = end[it] - start[it];
    if (i < a.length) {
      lens[i] = len;
      src_start[i] = start[it];
    }
    mine += len;
    const uint64_t bal = __ballot(ok[it]);
    nvalid += __popcll(bal);
    if (out_validity != nullptr && lane == 0 && (i - lane) < a.length) out_validity[(i - lane) >> 6] = bal;
  }
  // workgroup total (64-bit)
  long long s = mine;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
  if (lane == 0) wave_sum[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    long long t = 0;
    for (int k = 0; k < kWavesPerBlock; ++k) t += wave_sum[k];
    block_sums[blockIdx.x] = t;
  }
  if (valid_count != nullptr && lane == 0 && nvalid != 0) atomicAdd(valid_count, static_cast<unsigned long long>(nvalid));
}

// in-place exclusive scan of nblocks 64-bit totals; total -> sums[nblocks]
__global__ __launch_bounds__(1024) void bin_scan_blocks_kernel(long long* sums, int64_t nblocks) {
  __shared__ long long wave_tot[16];
  __shared__ long long carry_s;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < nblocks; base += 1024) {
    const int64_t i = base + tid;
    const long long v = i < nblocks ? sums[i] : 0;
    long long x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const long long n = __shfl_up(x, d, 64);
      if (lane >= d) x += n;
    }
    if (lane == 63) wave_tot[wave] = x;
    __syncthreads();
    long long pre = 0;
    for (int k = 0; k < wave; ++k) pre += wave_tot[k];
    const long long carry = carry_s;
    if (i < nblocks) sums[i] = carry + pre + x - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + pre + x;
    __syncthreads();
  }
  if (tid == 0) sums[nblocks] = carry_s;
}

// lens (in out_offsets) -> exclusive offsets, per workgroup of 4096 slots
template <typename O>
__global__ __launch_bounds__(kBlock) void bin_offsets_kernel(O* __restrict__ out_offsets, int64_t length,
                                                             const long long* __restrict__ block_base) {
  __shared__ uint64_t wave_tot[kWavesPerBlock];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kBinRows;
  // thread t owns 16 CONSECUTIVE slots so that the scan order is the slot order
  uint64_t v[16];
  uint64_t mine = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t i = base + tid * 16 + k;
    v[k] = i < length ? static_cast<uint64_t>(out_offsets[i]) : 0u;
    mine += v[k];
  }
  const uint64_t incl = wave_inclusive_scan_u64(mine);
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint64_t pre = incl - mine;
  for (int k = 0; k < wave; ++k) pre += wave_tot[k];
  long long run = block_base[blockIdx.x] + pre;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t i = base + tid * 16 + k;
    if (i < length) out_offsets[i] = static_cast<O>(run);
    run += v[k];
  }
  if (blockIdx.x == gridDim.x - 1 && tid == kBlock - 1) out_offsets[length] = static_cast<O>(block_base[gridDim.x]);
}

// Output-centric copy: one workgroup per kBinChunk output bytes.  It finds the rows overlapping
// its byte range by binary search in out_offsets, stages their {out offset, source start} through
// LDS in batches, and every lane then produces one output dword per step: the row of its first
// byte by binary search in LDS, the following bytes by walking forward (empty rows are skipped).
// Stores are dense dwords; loads are byte loads that consecutive lanes issue to consecutive
// source bytes of the same value.
constexpr int kBinChunk = 16384;   // output bytes per workgroup
constexpr int kBinBatch = 2048;    // rows staged per pass

// (SHIFT: the entries are list offsets in ELEMENTS of 2^SHIFT bytes — a list whose values are fixed-width and free of nulls
//  is a binary array with offsets scaled by the element width; 0 for utf8 / binary.  Positions below are bytes.)
template <typename O>
__device__ __forceinline__ int64_t bin_row_of(const O* __restrict__ out_offsets, int64_t m, int64_t pos, int shift) {
  // largest r in [0, m) with out_offsets[r] <= pos  (pos < out_offsets[m])
  int64_t lo = 0, hi = m;  // invariant: out_offsets[lo] <= pos < out_offsets[hi]
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if ((static_cast<int64_t>(out_offsets[mid]) << shift) <= pos) lo = mid; else hi = mid;
  }
  return lo;
}

template <typename O>
__global__ __launch_bounds__(kBlock) void bin_copy_kernel(const uint8_t* __restrict__ data,
                                                          const O* __restrict__ src_start,
                                                          const O* __restrict__ out_offsets, int64_t m,
                                                          int64_t total, uint8_t* __restrict__ out_data, int shift) {
  __shared__ int64_t s_out[kBinBatch + 1];
  __shared__ int64_t s_src[kBinBatch];
  const int tid = threadIdx.x;
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * kBinChunk;
  const int64_t b1 = b0 + kBinChunk < total ? b0 + kBinChunk : total;
  const int64_t rlo = bin_row_of<O>(out_offsets, m, b0, shift);
  const int64_t rhi = bin_row_of<O>(out_offsets, m, b1 - 1, shift);
  for (int64_t rb = rlo; rb <= rhi; rb += kBinBatch) {
    const int nb = static_cast<int>(rhi + 1 - rb < kBinBatch ? rhi + 1 - rb : kBinBatch);
    __syncthreads();
    for (int k = tid; k <= nb; k += kBlock) s_out[k] = static_cast<int64_t>(out_offsets[rb + k]) << shift;
    for (int k = tid; k < nb; k += kBlock) s_src[k] = static_cast<int64_t>(src_start[rb + k]) << shift;
    __syncthreads();
    const int64_t lo = s_out[0] > b0 ? s_out[0] : b0;
    const int64_t hi = s_out[nb] < b1 ? s_out[nb] : b1;
    if (hi <= lo) continue;
    const int64_t q0 = lo - static_cast<int64_t>((reinterpret_cast<uint64_t>(out_data) + lo) & 3);
    for (int64_t q = q0 + 4 * tid; q < hi; q += 4 * kBlock) {
      const int64_t first = q > lo ? q : lo;
      // row of `first` inside the batch
      int l = 0, h = nb;
      while (h - l > 1) {
        const int mid = (l + h) >> 1;
        if (s_out[mid] <= first) l = mid; else h = mid;
      }
      int r = l;
      int64_t r_out = s_out[r], r_end = s_out[r + 1], r_src = s_src[r];
      uint32_t word = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t pos = q + j;
        if (pos < lo || pos >= hi) continue;
        while (pos >= r_end) {
          ++r;
          r_out = r_end;
          r_end = s_out[r + 1];
          r_src = s_src[r];
        }
        word |= static_cast<uint32_t>(data[r_src + (pos - r_out)]) << (8 * j);
      }
      if (q >= lo && q + 4 <= hi) {
        *reinterpret_cast<uint32_t*>(out_data + q) = word;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (q + j >= lo && q + j < hi) out_data[q + j] = static_cast<uint8_t>(word >> (8 * j));
      }
    }
  }
}

static int make_args(const ArxBinarySpan* values, const ArxSpan* indices, int index_type, BinTakeArgs* a, int offset_width = 4) {
  if (values == nullptr || indices == nullptr) {
    set_error("values/indices is NULL");
    return ARX_INVALID;
  }
  if (index_type < 0 || index_type > 7) {
    set_error("Unsupported index type %d for take", index_type);
    return ARX_NOT_IMPLEMENTED;
  }
  if (indices->length > 0 && (indices->data == nullptr || values->offsets == nullptr)) {
    set_error("indices data / values offsets buffer is NULL");
    return ARX_INVALID;
  }
  static const int widths[8] = {1, 1, 2, 2, 4, 4, 8, 8};
  a->offsets = reinterpret_cast<const uint8_t*>(values->offsets) + values->offset * offset_width;
  a->data = static_cast<const uint8_t*>(values->data);
  a->src_valid_bytes = values->null_count != 0 ? static_cast<const uint8_t*>(values->validity) : nullptr;
  a->src_valid_offset = values->offset;
  a->indices = static_cast<const uint8_t*>(indices->data) + indices->offset * widths[index_type];
  a->index_type = index_type;
  a->ivalid = make_bits(indices->null_count != 0 ? indices->validity : nullptr, indices->offset, indices->length);
  a->length = indices->length;
  return ARX_OK;
}

}  // namespace arx

using namespace arx;

// workspace: [workgroup byte totals, 64-bit][source start of every output slot, one offset each]
static size_t bin_sums_bytes(int64_t num_indices) {
  return (static_cast<size_t>(ceil_div(num_indices, kBinRows) + 2) * 8 + 63) & ~static_cast<size_t>(63);
}
static size_t bin_workspace_bytes(int64_t num_indices, int offset_width) {
  if (num_indices < 0) num_indices = 0;
  return bin_sums_bytes(num_indices) + static_cast<size_t>(num_indices) * offset_width + 64;
}

template <typename O>
static int binary_take_offsets(const ArxBinarySpan* values, const ArxSpan* indices, int index_type, void* ws, size_t ws_bytes,
                               O* out_offsets, void* out_validity, int64_t* valid_count, int64_t* out_total_bytes, void* stream) {
  BinTakeArgs a{};
  const int rc = make_args(values, indices, index_type, &a, static_cast<int>(sizeof(O)));
  if (rc != ARX_OK) return rc;
  if (out_offsets == nullptr || out_total_bytes == nullptr) {
    set_error("out_offsets / out_total_bytes is NULL");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  *out_total_bytes = 0;
  if (a.length == 0) {
    ARX_HIP(hipMemsetAsync(out_offsets, 0, sizeof(O), st));
    return ARX_OK;
  }
  if (ws == nullptr || ws_bytes < bin_workspace_bytes(a.length, static_cast<int>(sizeof(O))) || (reinterpret_cast<uint64_t>(ws) & 7) != 0) {
    set_error("binary take workspace too small / unaligned");
    return ARX_INVALID;
  }
  const bool needs_validity = a.src_valid_bytes != nullptr || a.ivalid.base != nullptr;
  if (needs_validity && out_validity == nullptr) {
    set_error("take: inputs may have nulls but out_validity is NULL");
    return ARX_INVALID;
  }
  const int64_t nblocks = ceil_div(a.length, kBinRows);
  long long* sums = static_cast<long long*>(ws);
  O* src_start = reinterpret_cast<O*>(static_cast<uint8_t*>(ws) + bin_sums_bytes(a.length));
  hipLaunchKernelGGL(bin_lengths_kernel<O>, dim3(static_cast<unsigned>(nblocks)), dim3(kBlock), 0, st, a, out_offsets,
                     src_start, static_cast<uint64_t*>(out_validity), reinterpret_cast<unsigned long long*>(valid_count), sums);
  ARX_CHECK_LAUNCH("bin_lengths_kernel");
  hipLaunchKernelGGL(bin_scan_blocks_kernel, dim3(1), dim3(1024), 0, st, sums, nblocks);
  ARX_CHECK_LAUNCH("bin_scan_blocks_kernel");
  long long total = 0;
  ARX_HIP(hipMemcpyAsync(&total, sums + nblocks, 8, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  if (sizeof(O) == 4 && total > 2147483647LL) {
    // the reference's offset builder overflows the same way (int32 offsets)
    set_error("offset overflow while taking from a binary array: %lld bytes do not fit int32 offsets", total);
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(bin_offsets_kernel<O>, dim3(static_cast<unsigned>(nblocks)), dim3(kBlock), 0, st, out_offsets,
                     a.length, sums);
  ARX_CHECK_LAUNCH("bin_offsets_kernel");
  *out_total_bytes = total;
  return ARX_OK;
}

// shift: see bin_copy_kernel — total_bytes stays what the offsets count (elements for a list), the copy moves it << shift
template <typename O>
static int binary_take_data(const ArxBinarySpan* values, int64_t num_indices, const void* ws, size_t ws_bytes,
                            const O* out_offsets, int64_t total_bytes, void* out_data, void* stream, int shift = 0) {
  if (values == nullptr) {
    set_error("values is NULL");
    return ARX_INVALID;
  }
  if (num_indices <= 0 || total_bytes <= 0) return ARX_OK;
  if (out_offsets == nullptr || out_data == nullptr || values->data == nullptr || ws == nullptr ||
      ws_bytes < bin_workspace_bytes(num_indices, static_cast<int>(sizeof(O)))) {
    set_error("binary take: NULL buffer or workspace too small (pass the workspace of the offsets call)");
    return ARX_INVALID;
  }
  if (sizeof(O) == 4 && total_bytes > 2147483647LL) {
    set_error("binary take: total_bytes does not fit int32 offsets");
    return ARX_INVALID;
  }
  const O* src_start = reinterpret_cast<const O*>(static_cast<const uint8_t*>(ws) + bin_sums_bytes(num_indices));
  const int64_t copy_bytes = total_bytes << shift;
  const int64_t chunks = ceil_div(copy_bytes, kBinChunk);
  if (chunks > 0x7FFFFFFFll) {
    set_error("binary take: %lld output bytes are more than one launch takes", static_cast<long long>(total_bytes));
    return ARX_NOT_IMPLEMENTED;
  }
  hipLaunchKernelGGL(bin_copy_kernel<O>, dim3(static_cast<unsigned>(chunks)), dim3(kBlock), 0, as_stream(stream),
                     static_cast<const uint8_t*>(values->data), src_start, out_offsets, num_indices, copy_bytes,
                     static_cast<uint8_t*>(out_data), shift);
  ARX_CHECK_LAUNCH("bin_copy_kernel");
  return ARX_OK;
}

extern "C" {

size_t arx_binary_take_workspace_bytes(int64_t num_indices) { return bin_workspace_bytes(num_indices, 4); }
size_t arx_large_binary_take_workspace_bytes(int64_t num_indices) { return bin_workspace_bytes(num_indices, 8); }

int arx_binary_take_offsets(const ArxBinarySpan* values, const ArxSpan* indices, int index_type, void* ws,
                            size_t ws_bytes, int32_t* out_offsets, void* out_validity, int64_t* valid_count,
                            int64_t* out_total_bytes, void* stream) {
  return binary_take_offsets<int32_t>(values, indices, index_type, ws, ws_bytes, out_offsets, out_validity, valid_count,
                                      out_total_bytes, stream);
}

int arx_binary_take_data(const ArxBinarySpan* values, int64_t num_indices, const void* ws, size_t ws_bytes,
                         const int32_t* out_offsets, int64_t total_bytes, void* out_data, void* stream) {
  return binary_take_data<int32_t>(values, num_indices, ws, ws_bytes, out_offsets, total_bytes, out_data, stream);
}

// large_utf8 / large_binary: values->offsets points at int64 entries
int arx_large_binary_take_offsets(const ArxBinarySpan* values, const ArxSpan* indices, int index_type, void* ws,
                                  size_t ws_bytes, int64_t* out_offsets, void* out_validity, int64_t* valid_count,
                                  int64_t* out_total_bytes, void* stream) {
  return binary_take_offsets<int64_t>(values, indices, index_type, ws, ws_bytes, out_offsets, out_validity, valid_count,
                                      out_total_bytes, stream);
}

int arx_large_binary_take_data(const ArxBinarySpan* values, int64_t num_indices, const void* ws, size_t ws_bytes,
                               const int64_t* out_offsets, int64_t total_bytes, void* out_data, void* stream) {
  return binary_take_data<int64_t>(values, num_indices, ws, ws_bytes, out_offsets, total_bytes, out_data, stream);
}

// list<T> / large_list<T> with fixed-width T (2^elem_shift bytes) whose values have no nulls: values->offsets are the list
// offsets (in elements), values->data element 0 of the nested values; the offsets call is arx_(large_)binary_take_offsets
// as it is (it counts elements), this one copies the elements
int arx_list_take_data(const ArxBinarySpan* values, int elem_shift, int64_t num_indices, const void* ws, size_t ws_bytes,
                       const int32_t* out_offsets, int64_t total_elements, void* out_data, void* stream) {
  if (elem_shift < 0 || elem_shift > 5) {
    set_error("arx_list_take_data: elements of 2^%d bytes", elem_shift);
    return ARX_INVALID;
  }
  return binary_take_data<int32_t>(values, num_indices, ws, ws_bytes, out_offsets, total_elements, out_data, stream, elem_shift);
}

int arx_large_list_take_data(const ArxBinarySpan* values, int elem_shift, int64_t num_indices, const void* ws, size_t ws_bytes,
                             const int64_t* out_offsets, int64_t total_elements, void* out_data, void* stream) {
  if (elem_shift < 0 || elem_shift > 5) {
    set_error("arx_large_list_take_data: elements of 2^%d bytes", elem_shift);
    return ARX_INVALID;
  }
  return binary_take_data<int64_t>(values, num_indices, ws, ws_bytes, out_offsets, total_elements, out_data, stream, elem_shift);
}

}  // extern "C"
