// coalesce(values, fill) for two operands of one fixed-width (or boolean) type — what fill_null(values, fill_value)
// calls — for gfx950.
//
// What it restates (semantics only):
//   CoalesceFunctor / ExecArrayCoalesce   cpp/src/arrow/compute/kernels/scalar_if_else.cc   (fixed-width types)
// out[i] = values[i] where values is valid, else fill[i] (an array) / the fill scalar; out is valid where either is
// (a null fill scalar leaves the validity of `values`).  One wave takes 64 rows: each lane selects its slot, the
// validity word of the 64 rows is the OR of the two input words; booleans are selected 64 at a time with bit operations.
// HBM: values + fill once, the result once (streaming).
#include "arx_common.h"

#include <algorithm>

namespace arx {

template <typename T>
__global__ __launch_bounds__(kBlock) void coalesce2_kernel(const T* __restrict__ a, Bits av, const T* __restrict__ b, T b_scalar,
                                                           Bits bv, int b_kind /*0 array, 1 valid scalar, 2 null scalar*/,
                                                           int64_t n, T* __restrict__ out, uint64_t* __restrict__ out_valid) {
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t nwords = (n + 63) >> 6;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6); w < nwords; w += nwaves) {
    const uint64_t va = load_word(av, w);
    const uint64_t vb = b_kind == 0 ? load_word(bv, w) : (b_kind == 1 ? load_word(Bits{nullptr, 0, n, 0}, w) : 0ull);
    const int64_t i = (w << 6) + lane;
    if (i < n) {
      const bool use_a = (va >> lane) & 1ull;
      T v = use_a ? a[i] : (b_kind == 0 ? b[i] : b_scalar);
      if (!use_a && !((vb >> lane) & 1ull)) v = T(0);   // (a null result slot: zeroed)
      out[i] = v;
    }
    if (lane == 0) out_valid[w] = va | vb;
  }
}

// booleans: 64 rows per lane
__global__ __launch_bounds__(kBlock) void coalesce2_bool_kernel(Bits a, Bits av, Bits b, int b_scalar, Bits bv, int b_kind, int64_t n,
                                                                uint64_t* __restrict__ out, uint64_t* __restrict__ out_valid) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t nwords = (n + 63) >> 6;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < nwords; w += stride) {
    const uint64_t va = load_word(av, w);
    const uint64_t all = load_word(Bits{nullptr, 0, n, 0}, w);
    const uint64_t vb = b_kind == 0 ? load_word(bv, w) : (b_kind == 1 ? all : 0ull);
    const uint64_t db = b_kind == 0 ? load_word(b, w) : (b_scalar ? all : 0ull);
    out[w] = (load_word(a, w) & va) | (db & vb & ~va);
    out_valid[w] = va | vb;
  }
}

extern "C" {

int arx_coalesce2(int byte_width, const ArxSpan* values, const ArxSpan* fill, const void* fill_scalar, int64_t length,
                  void* out_data, void* out_validity, void* stream) {
  if (values == nullptr || length < 0 || values->length != length || (fill != nullptr && fill->length != length) ||
      (byte_width != 0 && byte_width != 1 && byte_width != 2 && byte_width != 4 && byte_width != 8)) {
    set_error("bad arguments to arx_coalesce2");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  if (values->data == nullptr || out_data == nullptr || out_validity == nullptr || (fill != nullptr && fill->data == nullptr)) {
    set_error("NULL buffer passed to arx_coalesce2");
    return ARX_INVALID;
  }
  const int b_kind = fill != nullptr ? 0 : (fill_scalar != nullptr ? 1 : 2);
  const Bits av = make_bits(values->null_count != 0 ? values->validity : nullptr, values->offset, length);
  const Bits bv = fill != nullptr ? make_bits(fill->null_count != 0 ? fill->validity : nullptr, fill->offset, length) : Bits{nullptr, 0, length, 0};
  hipStream_t st = as_stream(stream);
  const int64_t nwords = ceil_div(length, 64);
  if (byte_width == 0) {
    const Bits a = make_bits(values->data, values->offset, length);
    const Bits b = fill != nullptr ? make_bits(fill->data, fill->offset, length) : Bits{nullptr, 0, length, 0};
    const int sc = fill_scalar != nullptr ? (*static_cast<const uint8_t*>(fill_scalar) != 0) : 0;
    const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(nwords, kBlock), 1 << 16)));
    hipLaunchKernelGGL(coalesce2_bool_kernel, dim3(grid), dim3(kBlock), 0, st, a, av, b, sc, bv, b_kind, length,
                       static_cast<uint64_t*>(out_data), static_cast<uint64_t*>(out_validity));
    ARX_CHECK_LAUNCH("coalesce2_bool_kernel");
    return ARX_OK;
  }
  const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(nwords, kWavesPerBlock * 4), 1 << 20)));
#define ARX_COALESCE_CASE(W, T)                                                                                              \
  case W: {                                                                                                                  \
    T sc = T(0);                                                                                                             \
    if (fill_scalar != nullptr) sc = *static_cast<const T*>(fill_scalar);                                                    \
    hipLaunchKernelGGL(coalesce2_kernel<T>, dim3(grid), dim3(kBlock), 0, st, static_cast<const T*>(values->data) + values->offset, \
                       av, fill != nullptr ? static_cast<const T*>(fill->data) + fill->offset : nullptr, sc, bv, b_kind, length, \
                       static_cast<T*>(out_data), static_cast<uint64_t*>(out_validity));                                     \
    break;                                                                                                                   \
  }
  switch (byte_width) {
    ARX_COALESCE_CASE(1, uint8_t)
    ARX_COALESCE_CASE(2, uint16_t)
    ARX_COALESCE_CASE(4, uint32_t)
    ARX_COALESCE_CASE(8, uint64_t)
  }
#undef ARX_COALESCE_CASE
  ARX_CHECK_LAUNCH("coalesce2_kernel");
  return ARX_OK;
}

}  // extern "C"

}  // namespace arx
