// Streaming element-wise kernels for gfx950: cast f64->f32, greater, add, and the validity
// plumbing (bitmap copy / and / popcount) the ScalarExecutor does on the host.
//
// What they restate:
//   CastPrimitive<FloatType,DoubleType>::Exec   cpp/src/arrow/compute/kernels/scalar_cast_internal.cc:41-53
//   ComparePrimitiveArrayArray/ArrayScalar/ScalarArray  cpp/src/arrow/compute/kernels/scalar_compare.cc:165-247
//   Greater::Call                               same file :58-64
//   Add::Call (wrap-around for ints)            cpp/src/arrow/compute/kernels/base_arithmetic_internal.h:45-80
//   PropagateNullsSpans / BitmapAnd / CopyBitmap cpp/src/arrow/compute/exec.cc:1222-1281, util/bitmap_ops.cc
//
// All of them are HBM-bound streams: every lane moves 16 bytes per load instruction
// (1 KiB per wave instruction), 4 independent loads in flight, grid-stride over a grid of
// 256 CUs x 8 workgroups.
#include "arx_common.h"

#include <limits>

#include <string.h>

#include <algorithm>
#include <type_traits>

namespace arx {

// One-shot grids: one block-iteration of work per workgroup.  A persistent grid of 256 x 8 workgroups striding over
// the arrays measured 4.8-5.6 TB/s on the cast / compare shapes, the same kernels launched with one workgroup per
// block-iteration 5.6-6.5 (blocks are dispatched in order, so the active ones cover a compact moving window of
// memory instead of 2048 streams a stride apart; scripts/micro/stream_bench.hip, profiles/r03_a_stream_bench.txt).
// The grid-stride loops stay: they run once, and keep the kernels correct for any grid.
static inline unsigned stream_grid(int64_t work_items_per_block_iter, int64_t n) {
  const int64_t blocks = ceil_div(n, work_items_per_block_iter);
  return static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(blocks, int64_t(1) << 30)));
}

// ------------------------------------------------------------------ cast f64 -> f32
// Per workgroup: 2 x (256 lanes x 2 doubles) = 1024 rows; 16-byte non-temporal loads, 8-byte non-temporal stores
// (4 doubles per lane -> one 16-byte store measured no faster, in either lane arrangement: stream_bench cast4s / cast4x).
template <bool ALIGNED>
__global__ __launch_bounds__(kBlock) void cast_f64_f32_kernel(const double* __restrict__ in,
                                                              int64_t n, float* __restrict__ out) {
  constexpr int U = 2;
  const int64_t rows_per_block = kBlock * 2 * U;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * rows_per_block;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * rows_per_block; base < n; base += stride) {
    double2 v[U];
    int64_t r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      r[u] = base + (u * kBlock + threadIdx.x) * 2;
      if (r[u] + 1 < n) {
        if constexpr (ALIGNED) {
          const uint4 q = nt_load16(in + r[u]);
          v[u] = __builtin_bit_cast(double2, q);
        } else {
          v[u].x = nt_load(in + r[u]);
          v[u].y = nt_load(in + r[u] + 1);
        }
      } else if (r[u] < n) {
        v[u].x = in[r[u]];
        v[u].y = 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r[u] + 1 < n) {
        float2 o;
        o.x = static_cast<float>(v[u].x);  // v_cvt_f32_f64: IEEE round-to-nearest-even
        o.y = static_cast<float>(v[u].y);
        if constexpr (ALIGNED) {
          nt_store8(out + r[u], __builtin_bit_cast(uint2, o));
        } else {
          nt_store(out + r[u], o.x);
          nt_store(out + r[u] + 1, o.y);
        }
      } else if (r[u] < n) {
        out[r[u]] = static_cast<float>(v[u].x);
      }
    }
  }
}

// ------------------------------------------------------------------ buffer copy
// Device-to-device copy of a byte range: 16 bytes per lane, one 4 KiB piece per workgroup, one-shot grid — the form
// that reaches the box's copy rate (6.2 TB/s read + written; hipMemcpyAsync d2d: 4.7; profiles/r03_a_stream_bench.txt).
// Head and tail bytes around the 16-byte aligned middle are moved by the first workgroup.
__global__ __launch_bounds__(kBlock) void buffer_copy_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                             int64_t head, int64_t n16, int64_t nbytes) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const uint4* __restrict__ s16 = reinterpret_cast<const uint4*>(src + head);
  uint4* __restrict__ d16 = reinterpret_cast<uint4*>(dst + head);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n16; i += stride) d16[i] = s16[i];
  if (blockIdx.x == 0) {
    const int64_t tail0 = head + n16 * 16;
    for (int64_t b = threadIdx.x; b < head; b += kBlock) dst[b] = src[b];
    for (int64_t b = tail0 + threadIdx.x; b < nbytes; b += kBlock) dst[b] = src[b];
  }
}

// ------------------------------------------------------------------ integer casts
// CastIntegerToInteger (kernels/scalar_cast_numeric.cc:46-54): unless allow_int_overflow, IntegersCanFit
// -> IntegersInRange (util/int_util.cc:594-665) rejects the first VALID slot (in row order) whose value
// does not fit the target; then every slot is converted with static_cast (nulls included).
// One pass: the cast is written regardless, the smallest offending row number goes to *first_bad.
template <typename OutT>
__global__ __launch_bounds__(kBlock) void cast_i64_kernel(const int64_t* __restrict__ in, Bits valid, int64_t n,
                                                          OutT* __restrict__ out, int check, int64_t lo, int64_t hi,
                                                          unsigned long long* __restrict__ first_bad) {
  constexpr int U = 4;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x * U;
  unsigned long long bad = ~0ull;
  for (int64_t base = (static_cast<int64_t>(blockIdx.x) * blockDim.x) * U + threadIdx.x; base < n; base += stride) {
    int64_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + static_cast<int64_t>(u) * blockDim.x;
      v[u] = in[i < n ? i : n - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + static_cast<int64_t>(u) * blockDim.x;
      if (i >= n) continue;
      out[i] = static_cast<OutT>(v[u]);
      if (check && (v[u] < lo || v[u] > hi)) {
        const bool ok = (load_word(valid, i >> 6) >> (i & 63)) & 1ull;
        if (ok && static_cast<unsigned long long>(i) < bad) bad = static_cast<unsigned long long>(i);
      }
    }
  }
  if (check && bad != ~0ull) atomicMin(first_bad, bad);
}

__global__ __launch_bounds__(kBlock) void cast_i32_i64_kernel(const int32_t* __restrict__ in, int64_t n,
                                                              int64_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    out[i] = static_cast<int64_t>(in[i]);
  }
}

// ------------------------------------------------------------------ numeric casts, all 10 x 10 pairs
// CastNumberToNumberUnsafe = static_cast on every slot (scalar_cast_internal.cc:41-53, 158-...), then/before the
// mode's check on VALID slots only; the smallest offending row goes to *first_bad (the reference reports the first
// offender in row order):
//   MODE 1  IntegersInRange(lo, hi) — CastIntegerToInteger / CastIntegerToFloating (scalar_cast_numeric.cc:46-54,
//           270-279; util/int_util.cc:594-665); bounds are values of the INPUT type
//   MODE 2  WasTruncated: static_cast<In>(out) != in — CastFloatingToInteger (:62-207): out-of-range values and
//           NaN fail it as well, like in the reference
template <typename InT, typename OutT, int MODE>
__global__ __launch_bounds__(kBlock) void cast_numeric_kernel(const InT* __restrict__ in, Bits valid, int64_t n,
                                                              OutT* __restrict__ out, InT lo, InT hi,
                                                              unsigned long long* __restrict__ first_bad) {
  constexpr int U = 4;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x * U;
  unsigned long long bad = ~0ull;
  for (int64_t base = (static_cast<int64_t>(blockIdx.x) * blockDim.x) * U + threadIdx.x; base < n; base += stride) {
    InT v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + static_cast<int64_t>(u) * blockDim.x;
      v[u] = in[i < n ? i : n - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + static_cast<int64_t>(u) * blockDim.x;
      if (i >= n) continue;
      const OutT o = static_cast<OutT>(v[u]);
      out[i] = o;
      bool fail = false;
      if constexpr (MODE == 1) fail = v[u] < lo || v[u] > hi;
      if constexpr (MODE == 2 && std::is_integral<OutT>::value) {
        // the hardware conversion saturates where the reference's is undefined (x86 yields INT_MIN): an explicit range
        // test keeps 2^31 / 2^63 / 2^64, which survive a saturated round trip, on the failing side as there
        constexpr double kLo = static_cast<double>(std::numeric_limits<OutT>::lowest());
        constexpr double kHi = 2.0 * static_cast<double>(OutT(1) << (std::numeric_limits<OutT>::digits - 1));
        const double dv = static_cast<double>(v[u]);
        fail = !(dv >= kLo && dv < kHi) || !(static_cast<InT>(o) == v[u]);
      }
      if (fail) {
        const bool ok = (load_word(valid, i >> 6) >> (i & 63)) & 1ull;
        if (ok && static_cast<unsigned long long>(i) < bad) bad = static_cast<unsigned long long>(i);
      }
    }
  }
  if (MODE != 0 && bad != ~0ull) atomicMin(first_bad, bad);
}

// ------------------------------------------------------------------ compare (greater)
// Each wave step covers 128 rows: lane l holds rows 2l, 2l+1 (one 16-byte load per operand).
// The two ballots (even rows / odd rows) are interleaved with scalar bit-spreads into two
// 64-bit output words, LSB-first like bit_util::PackBits (cpp/src/arrow/util/bit_util.h:270).
enum OperandKind { kArray = 0, kScalar = 1 };

template <typename T>
struct Pair {
  T x, y;
};

template <typename T, bool ALIGNED>
__device__ __forceinline__ Pair<T> load_pair(const T* p, int64_t r, int64_t n) {
  Pair<T> v;
  v.x = T(0);
  v.y = T(0);
  if (r + 1 < n) {
    if constexpr (ALIGNED) {
      const uint4 q = nt_load16(p + r);
      const T* t = reinterpret_cast<const T*>(&q);
      v.x = t[0];
      v.y = t[1];
    } else {
      v.x = p[r];
      v.y = p[r + 1];
    }
  } else if (r < n) {
    v.x = p[r];
  }
  return v;
}

// CMP: ARX_CMP_GREATER / GREATER_EQUAL / EQUAL / NOT_EQUAL (less / less_equal swap the operands at
// the launch site); IEEE semantics for floats: every ordered comparison with a NaN is false,
// not_equal is true (Equal / NotEqual / Greater / GreaterEqual, kernels/scalar_compare.cc:38-64)
template <int CMP, typename T>
__device__ __forceinline__ bool cmp_apply(T a, T b) {
  if constexpr (CMP == ARX_CMP_GREATER) return a > b;
  else if constexpr (CMP == ARX_CMP_GREATER_EQUAL) return a >= b;
  else if constexpr (CMP == ARX_CMP_EQUAL) return a == b;
  else return a != b;
}

template <typename T, int LK, int RK, bool ALIGNED, int CMP = ARX_CMP_GREATER>
__global__ __launch_bounds__(kBlock) void greater_kernel(const T* __restrict__ left, T lscalar,
                                                         const T* __restrict__ right, T rscalar,
                                                         int64_t n, uint64_t* __restrict__ out) {
  constexpr int U = 4;
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t rows_per_iter = 128 * U;
  const int64_t niter = (n + rows_per_iter - 1) / rows_per_iter;
  for (int64_t it = wave_g; it < niter; it += nwaves) {
    const int64_t base = it * rows_per_iter;
    Pair<T> l[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = base + u * 128 + lane * 2;
      if constexpr (LK == kArray) {
        l[u] = load_pair<T, ALIGNED>(left, row, n);
      } else {
        l[u].x = lscalar;
        l[u].y = lscalar;
      }
      if constexpr (RK == kArray) {
        r[u] = load_pair<T, ALIGNED>(right, row, n);
      } else {
        r[u].x = rscalar;
        r[u].y = rscalar;
      }
    }
    uint64_t words[2 * U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = base + u * 128 + lane * 2;
      const bool b0 = (row < n) && cmp_apply<CMP>(l[u].x, r[u].x);
      const bool b1 = (row + 1 < n) && cmp_apply<CMP>(l[u].y, r[u].y);
      const uint64_t e = __ballot(b0);
      const uint64_t o = __ballot(b1);
      words[2 * u] = spread32(e) | (spread32(o) << 1);
      words[2 * u + 1] = spread32(e >> 32) | (spread32(o >> 32) << 1);
    }
    // 2U output words per iteration: lanes 0..2U-1 store one each
    const int64_t w0 = base >> 6;
    const int64_t nwords = (n + 63) >> 6;
    uint64_t mine = 0;
#pragma unroll
    for (int k = 0; k < 2 * U; ++k) {
      if (lane == k) mine = words[k];
    }
    if (lane < 2 * U && (w0 + lane) < nwords) out[w0 + lane] = mine;
  }
}

// ------------------------------------------------------------------ add
template <typename T, bool ALIGNED, bool RSCALAR>
__global__ __launch_bounds__(kBlock) void add_kernel(const T* __restrict__ left,
                                                     const T* __restrict__ right, T rscalar, int64_t n,
                                                     T* __restrict__ out) {
  static_assert(sizeof(T) == 8, "64-bit element types");
  constexpr int U = 4;
  const int64_t rows_per_block = kBlock * 2 * U;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * rows_per_block;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * rows_per_block; base < n; base += stride) {
    Pair<T> l[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = base + (u * kBlock + threadIdx.x) * 2;
      l[u] = load_pair<T, ALIGNED>(left, row, n);
      if constexpr (RSCALAR) {
        r[u] = Pair<T>{rscalar, rscalar};
      } else {
        r[u] = load_pair<T, ALIGNED>(right, row, n);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = base + (u * kBlock + threadIdx.x) * 2;
      T sx, sy;
      if constexpr (std::is_integral<T>::value) {
        // unchecked add wraps (SafeSignedAdd, base_arithmetic_internal.h)
        sx = static_cast<T>(static_cast<uint64_t>(l[u].x) + static_cast<uint64_t>(r[u].x));
        sy = static_cast<T>(static_cast<uint64_t>(l[u].y) + static_cast<uint64_t>(r[u].y));
      } else {
        sx = l[u].x + r[u].x;
        sy = l[u].y + r[u].y;
      }
      if (row + 1 < n) {
        if constexpr (ALIGNED) {
          Pair<T> o{sx, sy};
          nt_store16(out + row, *reinterpret_cast<const uint4*>(&o));
        } else {
          out[row] = sx;
          out[row + 1] = sy;
        }
      } else if (row < n) {
        out[row] = sx;
      }
    }
  }
}

// ------------------------------------------------------------------ add / subtract / multiply (+ checked)
// Add / Subtract / Multiply and their *Checked forms (base_arithmetic_internal.h:45-120,290-364):
// unchecked integer results wrap (the reference computes them in unsigned arithmetic), the checked
// forms raise Status::Invalid("overflow") — but only for slots where BOTH operands are valid, since
// the reference's ScalarBinaryNotNull never visits a null slot.  Floats: plain IEEE arithmetic in
// both forms.  LK / RK: array or broadcast scalar, as for compare.
template <typename T, int OP, bool CHECKED, int LK, int RK>
__global__ __launch_bounds__(kBlock) void arith_kernel(const T* __restrict__ left, T lscalar,
                                                       const T* __restrict__ right, T rscalar, Bits lvalid,
                                                       Bits rvalid, int64_t n, T* __restrict__ out,
                                                       unsigned int* __restrict__ overflow) {
  constexpr int U = 4;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x * U;
  bool bad = false;
  for (int64_t base = (static_cast<int64_t>(blockIdx.x) * blockDim.x) * U + threadIdx.x; base < n; base += stride) {
    T l[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + static_cast<int64_t>(u) * blockDim.x;
      const int64_t ic = i < n ? i : n - 1;
      l[u] = LK == kArray ? left[ic] : lscalar;
      r[u] = RK == kArray ? right[ic] : rscalar;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + static_cast<int64_t>(u) * blockDim.x;
      if (i >= n) continue;
      T res;
      if constexpr (std::is_integral<T>::value) {
        bool ovf;
        if constexpr (OP == ARX_ARITH_ADD) ovf = __builtin_add_overflow(l[u], r[u], &res);
        else if constexpr (OP == ARX_ARITH_SUBTRACT) ovf = __builtin_sub_overflow(l[u], r[u], &res);
        else ovf = __builtin_mul_overflow(l[u], r[u], &res);  // res = the wrapped result in every case
        if constexpr (CHECKED) {
          if (ovf) {
            const bool both = ((load_word(lvalid, i >> 6) & load_word(rvalid, i >> 6)) >> (i & 63)) & 1ull;
            bad = bad || both;
          }
        }
      } else {
        if constexpr (OP == ARX_ARITH_ADD) res = l[u] + r[u];
        else if constexpr (OP == ARX_ARITH_SUBTRACT) res = l[u] - r[u];
        else res = l[u] * r[u];
      }
      out[i] = res;
    }
  }
  if constexpr (CHECKED) {
    if (bad) atomicOr(overflow, 1u);
  }
}


// ------------------------------------------------------------------ divide / divide_checked
// Divide / DivideChecked (base_arithmetic_internal.h:366-424), visited only where both operands are valid
// (ScalarBinaryNotNull).  int64: truncating division; a zero divisor is Status::Invalid("divide by zero") in BOTH
// forms; INT64_MIN / -1 yields 0 in the unchecked form and Status::Invalid("overflow") in the checked one.  double:
// IEEE division; the checked form fails on a zero divisor.  The reference overwrites its Status on every failing
// slot, so the LAST failing slot decides the message: errors[0] / errors[1] keep 1 + the largest failing row index
// of the overflow / zero-divisor kind.
template <typename T, bool CHECKED, int LK, int RK>
__global__ __launch_bounds__(kBlock) void divide_kernel(const T* __restrict__ left, T lscalar, const T* __restrict__ right,
                                                        T rscalar, Bits lvalid, Bits rvalid, int64_t n, T* __restrict__ out,
                                                        unsigned long long* __restrict__ errors) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  unsigned long long last_overflow = 0, last_zero = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const T l = LK == kArray ? left[i] : lscalar;
    const T r = RK == kArray ? right[i] : rscalar;
    T res = T(0);
    bool zero = false, overflow = false;
    if constexpr (std::is_integral<T>::value) {
      // DivideWithOverflowGeneric (util/int_util_overflow.h:124-138): a zero divisor, and for the SIGNED types min / -1
      bool min_by_minus_one = false;
      if constexpr (std::is_signed<T>::value) min_by_minus_one = l == std::numeric_limits<T>::min() && r == T(-1);
      if (r == 0) zero = true;
      else if (min_by_minus_one) overflow = true;    // (result 0 in the unchecked form)
      else res = static_cast<T>(l / r);
    } else {
      if (CHECKED && r == T(0)) zero = true;
      else res = l / r;
    }
    if (zero || (CHECKED && overflow)) {
      const bool both = ((load_word(lvalid, i >> 6) & load_word(rvalid, i >> 6)) >> (i & 63)) & 1ull;
      if (both) {
        if (zero) last_zero = static_cast<unsigned long long>(i) + 1;
        else last_overflow = static_cast<unsigned long long>(i) + 1;
      }
    }
    out[i] = res;
  }
  if (last_overflow != 0) atomicMax(errors, last_overflow);
  if (last_zero != 0) atomicMax(errors + 1, last_zero);
}

// ------------------------------------------------------------------ scalar aggregates over int64
// One pass produces what SumImpl / CountImpl / MinMaxImpl (kernels/aggregate_basic.inc.cc:49-110,
// 776-860) keep per column: the wrap-around sum of the valid values, their count, min and max.
// acc = {sum, count, min, max} (4 x int64, device, arx_reduce_i64_init), accumulated across calls
// like Consume / MergeFrom.  Per-thread partials -> wave shuffles -> 4 atomics per workgroup.
__global__ __launch_bounds__(kBlock) void reduce_i64_kernel(const int64_t* __restrict__ in, Bits valid, int64_t n,
                                                            unsigned long long* __restrict__ acc) {
  __shared__ unsigned long long s_sum[kWavesPerBlock], s_cnt[kWavesPerBlock];
  __shared__ long long s_min[kWavesPerBlock], s_max[kWavesPerBlock];
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  unsigned long long sum = 0, cnt = 0;
  long long mn = INT64_MAX, mx = INT64_MIN;
  // 4 unconditional 16-byte loads (2 rows each) in flight per lane; a null slot's value is loaded and
  // dropped.  `in` is 16-byte aligned here: the launcher peels an unaligned first row off.
  constexpr int U = 4;
  const int64_t npairs = n >> 1;
  const longlong2* __restrict__ in2 = reinterpret_cast<const longlong2*>(in);
  auto fold = [&](long long v, bool ok) {
    sum += ok ? static_cast<unsigned long long>(v) : 0ull;
    cnt += ok ? 1ull : 0ull;
    mn = (ok && v < mn) ? v : mn;
    mx = (ok && v > mx) ? v : mx;
  };
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; base < npairs; base += stride * U) {
    longlong2 v[U];
    uint32_t ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = base + u * stride;
      const int64_t pc = p < npairs ? p : npairs - 1;
      v[u] = in2[pc];
      const int64_t r = pc * 2;   // even: both bits sit in the same 64-bit word
      ok[u] = p < npairs ? static_cast<uint32_t>((load_word(valid, r >> 6) >> (r & 63)) & 3ull) : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      fold(v[u].x, ok[u] & 1u);
      fold(v[u].y, ok[u] & 2u);
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    fold(in[i], (load_word(valid, i >> 6) >> (i & 63)) & 1ull);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    sum += __shfl_xor(sum, d, 64);
    cnt += __shfl_xor(cnt, d, 64);
    const long long omn = __shfl_xor(mn, d, 64), omx = __shfl_xor(mx, d, 64);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
  }
  const int wave = threadIdx.x >> 6;
  if (lane_id() == 0) {
    s_sum[wave] = sum;
    s_cnt[wave] = cnt;
    s_min[wave] = mn;
    s_max[wave] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kWavesPerBlock; ++k) {
      sum += s_sum[k];
      cnt += s_cnt[k];
      mn = s_min[k] < mn ? s_min[k] : mn;
      mx = s_max[k] > mx ? s_max[k] : mx;
    }
    if (cnt != 0) {
      atomicAdd(&acc[0], sum);
      atomicAdd(&acc[1], cnt);
      atomicMin(reinterpret_cast<long long*>(&acc[2]), mn);
      atomicMax(reinterpret_cast<long long*>(&acc[3]), mx);
    }
  }
}

__global__ void reduce_i64_init_kernel(long long* acc) {
  acc[0] = 0;
  acc[1] = 0;
  acc[2] = INT64_MAX;
  acc[3] = INT64_MIN;
}

// ------------------------------------------------------------------ bitmaps
template <bool AND>
__global__ __launch_bounds__(kBlock) void bitmap_kernel(Bits a, Bits b, int64_t nwords,
                                                        uint64_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < nwords;
       w += stride) {
    uint64_t v = load_word(a, w);
    if constexpr (AND) v &= load_word(b, w);
    out[w] = v;
  }
}

// Bit-granular concatenation: OR `length` bits of `src` into `dst` starting at bit `dst_off` (what
// arrow::Concatenate does for validity / boolean data, array/concatenate.cc, when chunks are glued at
// arbitrary bit positions).  One destination word per lane-iteration: interior words are stored whole,
// the first and last word of the range are OR-ed (the destination is zero there, or holds the tail of
// the previous chunk written by an earlier launch on the same stream).
__global__ __launch_bounds__(kBlock) void bitmap_copy_at_kernel(Bits src, int64_t src_words, int64_t first_word,
                                                                int64_t dst_words, int shift,
                                                                uint64_t* __restrict__ dst) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; r < dst_words; r += stride) {
    uint64_t v = r < src_words ? load_word(src, r) << shift : 0;
    if (shift != 0 && r > 0) v |= load_word(src, r - 1) >> (64 - shift);
    if (r == 0 || r == dst_words - 1) {
      if (v != 0) atomicOr(reinterpret_cast<unsigned long long*>(dst + first_word + r), static_cast<unsigned long long>(v));
    } else {
      dst[first_word + r] = v;
    }
  }
}

// Offsets of one chunk of a utf8/binary array moved to their place in the concatenated array:
// out[i] = in[i] - in[0] + base for i in [0, n] (Concatenate's PutOffsets, array/concatenate.cc).
__global__ __launch_bounds__(kBlock) void rebase_offsets_kernel(const int32_t* __restrict__ in, int64_t n,
                                                                int32_t base, int32_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int32_t first = in[0];
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i <= n; i += stride) {
    out[i] = in[i] - first + base;
  }
}

// and_kleene / or_kleene / invert on boolean arrays, one 64-bit word per lane-iteration
// (KleeneAndOp / KleeneOrOp / InvertOp, kernels/scalar_boolean.cc:138-260): with
//   x_true = x_valid & x_data,  x_false = x_valid & ~x_data
//   and: data = l_true & r_true,  valid = l_false | r_false | (l_true & r_true)
//   or : data = l_true | r_true,  valid = l_true | r_true | (l_false & r_false)
// A NULL validity reads as all ones, padding bits past `length` come out zero (load_word).
template <int OP>
__global__ __launch_bounds__(kBlock) void kleene_kernel(Bits ld, Bits lv, Bits rd, Bits rv, int64_t nwords,
                                                        uint64_t* __restrict__ out_data,
                                                        uint64_t* __restrict__ out_valid) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < nwords; w += stride) {
    const uint64_t l_valid = load_word(lv, w), r_valid = load_word(rv, w);
    const uint64_t l_data = load_word(ld, w), r_data = load_word(rd, w);
    const uint64_t lt = l_valid & l_data, lf = l_valid & ~l_data;
    const uint64_t rt = r_valid & r_data, rf = r_valid & ~r_data;
    uint64_t data, valid;
    if constexpr (OP == ARX_AND_KLEENE) {
      data = lt & rt;
      valid = lf | rf | (lt & rt);
    } else {
      data = lt | rt;
      valid = lt | rt | (lf & rf);
    }
    out_data[w] = data;
    if (out_valid != nullptr) out_valid[w] = valid;
  }
}

__global__ __launch_bounds__(kBlock) void invert_kernel(Bits a, Bits ones, int64_t nwords,
                                                        uint64_t* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < nwords; w += stride) {
    out[w] = ~load_word(a, w) & load_word(ones, w);  // `ones` only carries the tail mask
  }
}

// One byte per row (0 = clear) -> LSB-first bitmap words: a wave packs 64 rows with one ballot (BytesToBits,
// util/bitmap_builders.cc, for per-group validity bytes that are already in HBM); set bits are added to *set_count.
__global__ __launch_bounds__(kBlock) void bytes_to_bitmap_kernel(const uint8_t* __restrict__ bytes, int64_t n,
                                                                 uint64_t* __restrict__ out,
                                                                 unsigned long long* __restrict__ set_count) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (n + 63) / 64;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) >> 6;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * kBlock) >> 6;
  unsigned long long mine = 0;
  for (int64_t w = wave; w < nwords; w += nwaves) {
    const int64_t i = w * 64 + lane;
    const uint64_t word = __ballot(i < n && bytes[i] != 0);
    if (lane == 0) {
      out[w] = word;
      mine += static_cast<unsigned long long>(__popcll(word));
    }
  }
  // one atomic per WORKGROUP and a capped grid: atomics on one address serialise behind the L2 at ~12 ns each — one per
  // wave of a one-shot grid was 1.9 ms for 10M rows, 40x the pass itself (profiles/r03_r_aggregate_rocm_result_phase.txt)
  __shared__ unsigned long long part[kBlock / 64];
  if (lane == 0) part[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (set_count != nullptr && threadIdx.x == 0) {
    unsigned long long total = 0;
    for (int wv = 0; wv < kBlock / 64; ++wv) total += part[wv];
    if (total != 0) atomicAdd(set_count, total);
  }
}

__global__ __launch_bounds__(kBlock) void popcount_kernel(Bits a, int64_t nwords,
                                                          unsigned long long* total) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  uint64_t c = 0;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < nwords;
       w += stride) {
    c += __popcll(load_word(a, w));
  }
  c = wave_reduce_sum_u64(c);
  // one atomic per workgroup of a capped grid (see bytes_to_bitmap_kernel: atomics on one address serialise, ~12 ns each)
  __shared__ unsigned long long part[kBlock / 64];
  if (lane_id() == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long sum = 0;
    for (int wv = 0; wv < kBlock / 64; ++wv) sum += part[wv];
    if (sum != 0) atomicAdd(total, sum);
  }
}

template <typename T, int LK, int RK, int CMP = ARX_CMP_GREATER>
static int launch_greater(const T* left, T ls, const T* right, T rs, int64_t n, uint64_t* out,
                          hipStream_t st) {
  if (n == 0) return ARX_OK;  // empty arrays may come with NULL buffers
  if (n < 0 || out == nullptr || (LK == kArray && left == nullptr) || (RK == kArray && right == nullptr)) {
    set_error("bad arguments to compare");
    return ARX_INVALID;
  }
  const bool aligned = (LK != kArray || (reinterpret_cast<uint64_t>(left) & 15) == 0) &&
                       (RK != kArray || (reinterpret_cast<uint64_t>(right) & 15) == 0);
  const unsigned grid = stream_grid(kWavesPerBlock * 128 * 4, n);
  if (aligned) {
    hipLaunchKernelGGL((greater_kernel<T, LK, RK, true, CMP>), dim3(grid), dim3(kBlock), 0, st, left, ls,
                       right, rs, n, out);
  } else {
    hipLaunchKernelGGL((greater_kernel<T, LK, RK, false, CMP>), dim3(grid), dim3(kBlock), 0, st, left,
                       ls, right, rs, n, out);
  }
  ARX_CHECK_LAUNCH("greater_kernel");
  return ARX_OK;
}

template <typename T, bool RSCALAR>
static int launch_add(const T* left, const T* right, T rscalar, int64_t n, T* out, hipStream_t st) {
  if (n < 0 || (n > 0 && (left == nullptr || (!RSCALAR && right == nullptr) || out == nullptr))) {
    set_error("bad arguments to add");
    return ARX_INVALID;
  }
  if (n == 0) return ARX_OK;
  const bool aligned = ((reinterpret_cast<uint64_t>(left) | (RSCALAR ? 0 : reinterpret_cast<uint64_t>(right)) |
                         reinterpret_cast<uint64_t>(out)) & 15) == 0;
  const unsigned grid = stream_grid(kBlock * 2 * 4, n);
  if (aligned) {
    hipLaunchKernelGGL((add_kernel<T, true, RSCALAR>), dim3(grid), dim3(kBlock), 0, st, left, right, rscalar, n, out);
  } else {
    hipLaunchKernelGGL((add_kernel<T, false, RSCALAR>), dim3(grid), dim3(kBlock), 0, st, left, right, rscalar, n, out);
  }
  ARX_CHECK_LAUNCH("add_kernel");
  return ARX_OK;
}

template <typename T, int OP, bool CHECKED>
static int arith_shapes(const T* left, T ls, const T* right, T rs, const Bits& lv, const Bits& rv, int64_t n,
                        T* out, unsigned int* overflow, hipStream_t st) {
  const unsigned grid = stream_grid(kBlock * 4, n);
#define ARX_ARITH_LAUNCH(LK, RK)                                                                                  \
  hipLaunchKernelGGL((arith_kernel<T, OP, CHECKED, LK, RK>), dim3(grid), dim3(kBlock), 0, st, left, ls, right, rs, \
                     lv, rv, n, out, overflow)
  if (left != nullptr && right != nullptr) ARX_ARITH_LAUNCH(kArray, kArray);
  else if (left != nullptr) ARX_ARITH_LAUNCH(kArray, kScalar);
  else if (right != nullptr) ARX_ARITH_LAUNCH(kScalar, kArray);
  else {
    set_error("arithmetic: at least one operand must be an array");
    return ARX_INVALID;
  }
#undef ARX_ARITH_LAUNCH
  ARX_CHECK_LAUNCH("arith_kernel");
  return ARX_OK;
}

template <typename T>
static int divide_any(const T* left, T ls, const void* lvalid, int64_t loff, const T* right, T rs, const void* rvalid,
                      int64_t roff, int64_t n, int checked, T* out, uint64_t* errors, hipStream_t st) {
  if (n < 0 || loff < 0 || roff < 0 || (n > 0 && (out == nullptr || errors == nullptr))) {
    set_error("bad arguments to divide");
    return ARX_INVALID;
  }
  if (n == 0) return ARX_OK;
  if (left == nullptr && right == nullptr) {
    set_error("divide: at least one operand must be an array");
    return ARX_INVALID;
  }
  const Bits lv = make_bits(left != nullptr ? lvalid : nullptr, loff, n);
  const Bits rv = make_bits(right != nullptr ? rvalid : nullptr, roff, n);
  const unsigned grid = stream_grid(kBlock, n);
  unsigned long long* err = reinterpret_cast<unsigned long long*>(errors);
#define ARX_DIVIDE_LAUNCH(CHECKED, LK, RK)                                                                              \
  hipLaunchKernelGGL((divide_kernel<T, CHECKED, LK, RK>), dim3(grid), dim3(kBlock), 0, st, left, ls, right, rs, lv, rv, \
                     n, out, err)
  if (checked) {
    if (left != nullptr && right != nullptr) ARX_DIVIDE_LAUNCH(true, kArray, kArray);
    else if (left != nullptr) ARX_DIVIDE_LAUNCH(true, kArray, kScalar);
    else ARX_DIVIDE_LAUNCH(true, kScalar, kArray);
  } else {
    if (left != nullptr && right != nullptr) ARX_DIVIDE_LAUNCH(false, kArray, kArray);
    else if (left != nullptr) ARX_DIVIDE_LAUNCH(false, kArray, kScalar);
    else ARX_DIVIDE_LAUNCH(false, kScalar, kArray);
  }
#undef ARX_DIVIDE_LAUNCH
  ARX_CHECK_LAUNCH("divide_kernel");
  return ARX_OK;
}

template <typename T, bool CHECKED>
static int arith_any(int op, const T* left, T ls, const T* right, T rs, const Bits& lv, const Bits& rv, int64_t n,
                     T* out, unsigned int* overflow, hipStream_t st) {
  if (n < 0 || (n > 0 && out == nullptr)) {
    set_error("bad arguments to arithmetic");
    return ARX_INVALID;
  }
  if (n == 0) return ARX_OK;
  switch (op) {
    case ARX_ARITH_ADD: return arith_shapes<T, ARX_ARITH_ADD, CHECKED>(left, ls, right, rs, lv, rv, n, out, overflow, st);
    case ARX_ARITH_SUBTRACT: return arith_shapes<T, ARX_ARITH_SUBTRACT, CHECKED>(left, ls, right, rs, lv, rv, n, out, overflow, st);
    case ARX_ARITH_MULTIPLY: return arith_shapes<T, ARX_ARITH_MULTIPLY, CHECKED>(left, ls, right, rs, lv, rv, n, out, overflow, st);
    default:
      set_error("unknown arithmetic op %d", op);
      return ARX_INVALID;
  }
}

// op in ARX_CMP_*; less / less_equal = greater / greater_equal with the operands swapped.
// left / right NULL = that side is the scalar.
template <typename T, int CMP>
static int compare_shapes(const T* left, T ls, const T* right, T rs, int64_t n, uint64_t* out, hipStream_t st) {
  if (left != nullptr && right != nullptr) return launch_greater<T, kArray, kArray, CMP>(left, ls, right, rs, n, out, st);
  if (left != nullptr) return launch_greater<T, kArray, kScalar, CMP>(left, ls, nullptr, rs, n, out, st);
  if (right != nullptr) return launch_greater<T, kScalar, kArray, CMP>(nullptr, ls, right, rs, n, out, st);
  set_error("compare: at least one operand must be an array");
  return ARX_INVALID;
}

template <typename T>
static int compare_any(int op, const T* left, T ls, const T* right, T rs, int64_t n, uint64_t* out, hipStream_t st) {
  switch (op) {
    case ARX_CMP_EQUAL: return compare_shapes<T, ARX_CMP_EQUAL>(left, ls, right, rs, n, out, st);
    case ARX_CMP_NOT_EQUAL: return compare_shapes<T, ARX_CMP_NOT_EQUAL>(left, ls, right, rs, n, out, st);
    case ARX_CMP_GREATER: return compare_shapes<T, ARX_CMP_GREATER>(left, ls, right, rs, n, out, st);
    case ARX_CMP_GREATER_EQUAL: return compare_shapes<T, ARX_CMP_GREATER_EQUAL>(left, ls, right, rs, n, out, st);
    case ARX_CMP_LESS: return compare_shapes<T, ARX_CMP_GREATER>(right, rs, left, ls, n, out, st);
    case ARX_CMP_LESS_EQUAL: return compare_shapes<T, ARX_CMP_GREATER_EQUAL>(right, rs, left, ls, n, out, st);
    default:
      set_error("unknown compare op %d", op);
      return ARX_INVALID;
  }
}

// ------------------------------------------------------------------ many small device-to-device copies in one launch
// Concatenate (array/concatenate.cc) of many small chunks — Acero hands operators 32K-row batches — costs a launch (or
// a hipMemcpy) per chunk and column when done copy by copy; here a device table of {src, dst, nbytes} drives one
// launch: `per_seg` workgroups per segment, each copying 64 KiB slices with 16-byte accesses (the destination side aligned, the source wherever it is).
__global__ __launch_bounds__(kBlock) void copy_segments_kernel(const ArxCopySeg* __restrict__ segs, int64_t nsegs, unsigned per_seg) {
  constexpr uint64_t kSlice = 64 * 1024;
  const int64_t sg = blockIdx.x / per_seg;
  const unsigned part = blockIdx.x % per_seg;
  if (sg >= nsegs) return;
  const ArxCopySeg seg = segs[sg];
  const uint8_t* __restrict__ src = static_cast<const uint8_t*>(seg.src);
  uint8_t* __restrict__ dst = static_cast<uint8_t*>(seg.dst);
  const bool wide = ((reinterpret_cast<uint64_t>(src) | reinterpret_cast<uint64_t>(dst)) & 15) == 0;
  for (uint64_t base = static_cast<uint64_t>(part) * kSlice; base < seg.nbytes; base += static_cast<uint64_t>(per_seg) * kSlice) {
    const uint64_t end = base + kSlice < seg.nbytes ? base + kSlice : seg.nbytes;
    if (wide) {
      const uint64_t end16 = base + ((end - base) & ~uint64_t(15));
      for (uint64_t o = base + threadIdx.x * 16ull; o < end16; o += kBlock * 16ull) {
        *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(src + o);
      }
      for (uint64_t o = end16 + threadIdx.x; o < end; o += kBlock) dst[o] = src[o];
    } else {   // any alignment: bytes up to the destination's 16-byte boundary, then aligned stores of unaligned loads
      const uint64_t to_boundary = (16 - (reinterpret_cast<uint64_t>(dst + base) & 15)) & 15;
      const uint64_t head = to_boundary < end - base ? to_boundary : end - base;
      if (threadIdx.x < head) dst[base + threadIdx.x] = src[base + threadIdx.x];
      const uint64_t b16 = base + head;
      const uint64_t end16 = b16 + ((end - b16) & ~uint64_t(15));
      for (uint64_t o = b16 + threadIdx.x * 16ull; o < end16; o += kBlock * 16ull) {
        *reinterpret_cast<uint4*>(dst + o) = load16_unaligned(src + o);
      }
      for (uint64_t o = end16 + threadIdx.x; o < end; o += kBlock) dst[o] = src[o];
    }
  }
}

// The validity half of the same concatenation: many bitmap ranges, each at its own bit offset, ORed into ZEROED
// destination bitmaps at their own bit positions (ranges meet inside words: atomicOr).  src == NULL: all ones (a chunk
// without a validity buffer).  One thread per 64 source bits.
__global__ __launch_bounds__(kBlock) void bitmap_copy_segments_kernel(const ArxBitSeg* __restrict__ segs, int64_t nsegs,
                                                                      unsigned per_seg) {
  const int64_t sg = blockIdx.x / per_seg;
  const unsigned part = blockIdx.x % per_seg;
  if (sg >= nsegs) return;
  const ArxBitSeg seg = segs[sg];
  const Bits src = make_bits_device(seg.src, seg.src_bit_offset, seg.nbits);
  unsigned long long* dst = static_cast<unsigned long long*>(seg.dst);
  const int64_t nwords = (seg.nbits + 63) >> 6;
  for (int64_t w = static_cast<int64_t>(part) * kBlock + threadIdx.x; w < nwords; w += static_cast<int64_t>(per_seg) * kBlock) {
    const uint64_t v = load_word(src, w);   // bits past nbits read as zero; a NULL bitmap reads as ones
    if (v == 0) continue;
    const uint64_t at = static_cast<uint64_t>(seg.dst_bit_offset) + (static_cast<uint64_t>(w) << 6);
    const int sh = static_cast<int>(at & 63);
    atomicOr(&dst[at >> 6], static_cast<unsigned long long>(v << sh));
    if (sh != 0 && (v >> (64 - sh)) != 0) atomicOr(&dst[(at >> 6) + 1], static_cast<unsigned long long>(v >> (64 - sh)));
  }
}

// ------------------------------------------------------------------ compare / arithmetic on every numeric type
// The comparison family and add / subtract / multiply (+ _checked) for the element types the 64-bit kernels above do
// not take (int8 ... uint32, uint64, float): the same Call bodies (scalar_compare.cc:38-64,
// base_arithmetic_internal.h:45-150,290-364) instantiated per type — unchecked integer results wrap in the type's
// width, the checked forms report an overflow of THAT type (only where both operands are valid).
// One lane per row and step, U steps in flight: a wave reads 64 * sizeof(T) contiguous bytes per load.
template <typename T, int LK, int RK, int CMP>
__global__ __launch_bounds__(kBlock) void compare_rows_kernel(const T* __restrict__ left, T lscalar,
                                                              const T* __restrict__ right, T rscalar, int64_t n,
                                                              uint64_t* __restrict__ out) {
  constexpr int U = 8;
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t wave_g = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwords = (n + 63) >> 6;
  const int64_t niter = (nwords + U - 1) / U;
  const int64_t last = n - 1;
  for (int64_t it = wave_g; it < niter; it += nwaves) {
    const int64_t w0 = it * U;
    T l[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = ((w0 + u) << 6) + lane;
      const int64_t rc = row <= last ? row : last;   // unconditional (clamped) loads
      l[u] = LK == kArray ? left[rc] : lscalar;
      r[u] = RK == kArray ? right[rc] : rscalar;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = ((w0 + u) << 6) + lane;
      const uint64_t bal = __ballot(row <= last && cmp_apply<CMP>(l[u], r[u]));
      if (lane == 0 && (w0 + u) < nwords) out[w0 + u] = bal;
    }
  }
}

template <typename T, int CMP>
static int compare_rows_shapes(const T* left, T ls, const T* right, T rs, int64_t n, uint64_t* out, hipStream_t st) {
  const unsigned grid = stream_grid(kWavesPerBlock * 64 * 8, n);
  if (left != nullptr && right != nullptr) {
    hipLaunchKernelGGL((compare_rows_kernel<T, kArray, kArray, CMP>), dim3(grid), dim3(kBlock), 0, st, left, ls, right, rs, n, out);
  } else if (left != nullptr) {
    hipLaunchKernelGGL((compare_rows_kernel<T, kArray, kScalar, CMP>), dim3(grid), dim3(kBlock), 0, st, left, ls, right, rs, n, out);
  } else if (right != nullptr) {
    hipLaunchKernelGGL((compare_rows_kernel<T, kScalar, kArray, CMP>), dim3(grid), dim3(kBlock), 0, st, left, ls, right, rs, n, out);
  } else {
    set_error("compare: at least one operand must be an array");
    return ARX_INVALID;
  }
  ARX_CHECK_LAUNCH("compare_rows_kernel");
  return ARX_OK;
}

template <typename T>
static int compare_rows_any(int op, const void* left, const void* lsp, const void* right, const void* rsp, int64_t n,
                            uint64_t* out, hipStream_t st) {
  const T* l = static_cast<const T*>(left);
  const T* r = static_cast<const T*>(right);
  T ls = T(0), rs = T(0);
  if (l == nullptr && lsp != nullptr) memcpy(&ls, lsp, sizeof(T));
  if (r == nullptr && rsp != nullptr) memcpy(&rs, rsp, sizeof(T));
  switch (op) {
    case ARX_CMP_EQUAL: return compare_rows_shapes<T, ARX_CMP_EQUAL>(l, ls, r, rs, n, out, st);
    case ARX_CMP_NOT_EQUAL: return compare_rows_shapes<T, ARX_CMP_NOT_EQUAL>(l, ls, r, rs, n, out, st);
    case ARX_CMP_GREATER: return compare_rows_shapes<T, ARX_CMP_GREATER>(l, ls, r, rs, n, out, st);
    case ARX_CMP_GREATER_EQUAL: return compare_rows_shapes<T, ARX_CMP_GREATER_EQUAL>(l, ls, r, rs, n, out, st);
    case ARX_CMP_LESS: return compare_rows_shapes<T, ARX_CMP_GREATER>(r, rs, l, ls, n, out, st);
    case ARX_CMP_LESS_EQUAL: return compare_rows_shapes<T, ARX_CMP_GREATER_EQUAL>(r, rs, l, ls, n, out, st);
    default:
      set_error("unknown compare op %d", op);
      return ARX_INVALID;
  }
}

template <typename T>
static int arith_numeric_any(int op, int checked, const void* left, const void* lsp, const Bits& lv, const void* right,
                             const void* rsp, const Bits& rv, int64_t n, void* out, unsigned int* overflow, hipStream_t st) {
  const T* l = static_cast<const T*>(left);
  const T* r = static_cast<const T*>(right);
  T ls = T(0), rs = T(0);
  if (l == nullptr && lsp != nullptr) memcpy(&ls, lsp, sizeof(T));
  if (r == nullptr && rsp != nullptr) memcpy(&rs, rsp, sizeof(T));
  if (checked && std::is_integral<T>::value) {
    return arith_any<T, true>(op, l, ls, r, rs, lv, rv, n, static_cast<T*>(out), overflow, st);
  }
  return arith_any<T, false>(op, l, ls, r, rs, lv, rv, n, static_cast<T*>(out), nullptr, st);
}

template <typename OutT>
static int cast_i64_checked(const char* what, const ArxSpan* values, int unchecked, int64_t lo, int64_t hi, void* ws,
                            size_t ws_bytes, OutT* out, void* stream) {
  if (values == nullptr || values->length < 0) {
    set_error("bad arguments to %s", what);
    return ARX_INVALID;
  }
  const int64_t n = values->length;
  if (n == 0) return ARX_OK;
  if (values->data == nullptr || out == nullptr || (!unchecked && (ws == nullptr || ws_bytes < 8))) {
    set_error("NULL buffer / workspace passed to %s", what);
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  const int64_t* in = static_cast<const int64_t*>(values->data) + values->offset;
  const Bits valid = make_bits(values->null_count != 0 ? values->validity : nullptr, values->offset, n);
  unsigned long long* first_bad = static_cast<unsigned long long*>(ws);
  if (!unchecked) ARX_HIP(hipMemsetAsync(first_bad, 0xFF, 8, st));
  hipLaunchKernelGGL((cast_i64_kernel<OutT>), dim3(stream_grid(kBlock * 4, n)), dim3(kBlock), 0, st, in, valid, n, out,
                     unchecked ? 0 : 1, lo, hi, first_bad);
  ARX_CHECK_LAUNCH("cast_i64_kernel");
  if (unchecked) return ARX_OK;
  unsigned long long bad = ~0ull;
  ARX_HIP(hipMemcpyAsync(&bad, first_bad, 8, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  if (bad != ~0ull) {
    long long v = 0;
    ARX_HIP(hipMemcpyAsync(&v, in + bad, 8, hipMemcpyDeviceToHost, st));
    ARX_HIP(hipStreamSynchronize(st));
    // the text of IntegersInRange's GetErrorMessage, util/int_util.cc:607-611
    set_error("Integer value %lld not in range: %lld to %lld", v, static_cast<long long>(lo), static_cast<long long>(hi));
    return ARX_INVALID;
  }
  return ARX_OK;
}

// ---- generic numeric cast (arx_cast_numeric)
template <typename T> struct NumName;
#define ARX_NUM_NAME(T, S) template <> struct NumName<T> { static const char* get() { return S; } }
ARX_NUM_NAME(int8_t, "int8"); ARX_NUM_NAME(uint8_t, "uint8"); ARX_NUM_NAME(int16_t, "int16"); ARX_NUM_NAME(uint16_t, "uint16");
ARX_NUM_NAME(int32_t, "int32"); ARX_NUM_NAME(uint32_t, "uint32"); ARX_NUM_NAME(int64_t, "int64"); ARX_NUM_NAME(uint64_t, "uint64");
ARX_NUM_NAME(float, "float"); ARX_NUM_NAME(double, "double");
#undef ARX_NUM_NAME

template <typename T>
static void format_num(char* buf, size_t cap, T v) {
  if constexpr (std::is_floating_point<T>::value) {
    snprintf(buf, cap, "%f", static_cast<double>(v));                       // std::to_string, as the reference's StringBuilder prints floats
  } else if constexpr (std::is_signed<T>::value) {
    snprintf(buf, cap, "%lld", static_cast<long long>(v));
  } else {
    snprintf(buf, cap, "%llu", static_cast<unsigned long long>(v));
  }
}

template <typename InT, typename OutT>
static int cast_numeric_pair(const ArxSpan* values, int allow_int_overflow, int allow_float_truncate, void* ws,
                             size_t ws_bytes, void* out_v, hipStream_t st) {
  const int64_t n = values->length;
  const InT* in = static_cast<const InT*>(values->data) + values->offset;
  OutT* out = static_cast<OutT*>(out_v);
  const Bits valid = make_bits(values->null_count != 0 ? values->validity : nullptr, values->offset, n);
  constexpr bool in_f = std::is_floating_point<InT>::value, out_f = std::is_floating_point<OutT>::value;
  int mode = 0;
  InT lo = std::numeric_limits<InT>::lowest(), hi = std::numeric_limits<InT>::max();
  if constexpr (!in_f && !out_f) {
    // GetSafeMinMax (util/int_util.cc:795-880): the part of the output range the input type can express
    if (!allow_int_overflow && !std::is_same<InT, OutT>::value) {
      constexpr bool in_s = std::is_signed<InT>::value, out_s = std::is_signed<OutT>::value;
      if (in_s == out_s) {
        if (sizeof(OutT) < sizeof(InT)) {
          lo = static_cast<InT>(std::numeric_limits<OutT>::lowest());
          hi = static_cast<InT>(std::numeric_limits<OutT>::max());
        }
      } else if (!in_s) {   // unsigned -> signed
        lo = 0;
        hi = sizeof(InT) < sizeof(OutT) ? std::numeric_limits<InT>::max() : static_cast<InT>(std::numeric_limits<OutT>::max());
      } else {              // signed -> unsigned
        lo = 0;
        hi = sizeof(InT) <= sizeof(OutT) ? std::numeric_limits<InT>::max() : static_cast<InT>(std::numeric_limits<OutT>::max());
      }
      if (lo != std::numeric_limits<InT>::lowest() || hi != std::numeric_limits<InT>::max()) mode = 1;
    }
  } else if constexpr (!in_f && out_f) {
    // CheckForIntegerToFloatingTruncation (scalar_cast_numeric.cc:229-268): 32/64-bit integers into float, 64-bit into double
    if (!allow_float_truncate) {
      const bool f32 = sizeof(OutT) == 4;
      if (sizeof(InT) == 8 || (sizeof(InT) == 4 && f32)) {
        const long long limit = f32 ? (1LL << 24) : (1LL << 53);
        lo = std::is_signed<InT>::value ? static_cast<InT>(-limit) : static_cast<InT>(0);
        hi = static_cast<InT>(limit);
        mode = 1;
      }
    }
  } else if constexpr (in_f && !out_f) {
    if (!allow_float_truncate) mode = 2;
  }
  unsigned long long* first_bad = static_cast<unsigned long long*>(ws);
  if (mode != 0) {
    if (ws == nullptr || ws_bytes < 8) {
      set_error("arx_cast_numeric: a checked cast needs >= 8 bytes of device workspace");
      return ARX_INVALID;
    }
    ARX_HIP(hipMemsetAsync(first_bad, 0xFF, 8, st));
  }
  const unsigned grid = stream_grid(kBlock * 4, n);
  if (mode == 0) {
    hipLaunchKernelGGL((cast_numeric_kernel<InT, OutT, 0>), dim3(grid), dim3(kBlock), 0, st, in, valid, n, out, lo, hi, first_bad);
  } else if (mode == 1) {
    hipLaunchKernelGGL((cast_numeric_kernel<InT, OutT, 1>), dim3(grid), dim3(kBlock), 0, st, in, valid, n, out, lo, hi, first_bad);
  } else {
    hipLaunchKernelGGL((cast_numeric_kernel<InT, OutT, 2>), dim3(grid), dim3(kBlock), 0, st, in, valid, n, out, lo, hi, first_bad);
  }
  ARX_CHECK_LAUNCH("cast_numeric_kernel");
  if (mode == 0) return ARX_OK;
  unsigned long long bad = ~0ull;
  ARX_HIP(hipMemcpyAsync(&bad, first_bad, 8, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  if (bad == ~0ull) return ARX_OK;
  InT v{};
  ARX_HIP(hipMemcpyAsync(&v, in + bad, sizeof(InT), hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  char sv[64], slo[64], shi[64];
  format_num(sv, sizeof(sv), v);
  if (mode == 1) {
    format_num(slo, sizeof(slo), lo);
    format_num(shi, sizeof(shi), hi);
    set_error("Integer value %s not in range: %s to %s", sv, slo, shi);      // util/int_util.cc:607-611
  } else {
    set_error("Float value %s was truncated converting to %s", sv, NumName<OutT>::get());   // scalar_cast_numeric.cc:96-99
  }
  return ARX_INVALID;
}

template <typename InT>
static int cast_numeric_from(int out_type, const ArxSpan* values, int aio, int aft, void* ws, size_t ws_bytes, void* out,
                             hipStream_t st) {
  switch (out_type) {
    case ARX_NUM_INT8: return cast_numeric_pair<InT, int8_t>(values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_UINT8: return cast_numeric_pair<InT, uint8_t>(values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_INT16: return cast_numeric_pair<InT, int16_t>(values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_UINT16: return cast_numeric_pair<InT, uint16_t>(values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_INT32: return cast_numeric_pair<InT, int32_t>(values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_UINT32: return cast_numeric_pair<InT, uint32_t>(values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_INT64: return cast_numeric_pair<InT, int64_t>(values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_UINT64: return cast_numeric_pair<InT, uint64_t>(values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_FLOAT32: return cast_numeric_pair<InT, float>(values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_FLOAT64: return cast_numeric_pair<InT, double>(values, aio, aft, ws, ws_bytes, out, st);
    default: set_error("arx_cast_numeric: unknown output type %d", out_type); return ARX_INVALID;
  }
}

}  // namespace arx

using namespace arx;

extern "C" {

int arx_cast_numeric(const ArxSpan* values, int in_type, int out_type, int allow_int_overflow, int allow_float_truncate,
                     void* ws, size_t ws_bytes, void* out, void* stream) {
  if (values == nullptr || values->length < 0 || values->offset < 0) {
    set_error("bad arguments to arx_cast_numeric");
    return ARX_INVALID;
  }
  if (values->length == 0) return ARX_OK;
  if (values->data == nullptr || out == nullptr) {
    set_error("NULL buffer passed to arx_cast_numeric");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  const int aio = allow_int_overflow, aft = allow_float_truncate;
  switch (in_type) {
    case ARX_NUM_INT8: return cast_numeric_from<int8_t>(out_type, values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_UINT8: return cast_numeric_from<uint8_t>(out_type, values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_INT16: return cast_numeric_from<int16_t>(out_type, values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_UINT16: return cast_numeric_from<uint16_t>(out_type, values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_INT32: return cast_numeric_from<int32_t>(out_type, values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_UINT32: return cast_numeric_from<uint32_t>(out_type, values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_INT64: return cast_numeric_from<int64_t>(out_type, values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_UINT64: return cast_numeric_from<uint64_t>(out_type, values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_FLOAT32: return cast_numeric_from<float>(out_type, values, aio, aft, ws, ws_bytes, out, st);
    case ARX_NUM_FLOAT64: return cast_numeric_from<double>(out_type, values, aio, aft, ws, ws_bytes, out, st);
    default: set_error("arx_cast_numeric: unknown input type %d", in_type); return ARX_INVALID;
  }
}

int arx_cast_f64_f32(const double* in, int64_t length, float* out, void* stream) {
  if (length < 0 || (length > 0 && (in == nullptr || out == nullptr))) {
    set_error("bad arguments to arx_cast_f64_f32");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  hipStream_t st = as_stream(stream);
  const bool aligned =
      (reinterpret_cast<uint64_t>(in) & 15) == 0 && (reinterpret_cast<uint64_t>(out) & 7) == 0;
  const unsigned grid = stream_grid(kBlock * 2 * 2, length);
  if (aligned) {
    hipLaunchKernelGGL((cast_f64_f32_kernel<true>), dim3(grid), dim3(kBlock), 0, st, in, length, out);
  } else {
    hipLaunchKernelGGL((cast_f64_f32_kernel<false>), dim3(grid), dim3(kBlock), 0, st, in, length, out);
  }
  ARX_CHECK_LAUNCH("cast_f64_f32_kernel");
  return ARX_OK;
}

int arx_cast_i64_i32(const ArxSpan* values, int allow_int_overflow, void* ws, size_t ws_bytes, int32_t* out,
                     void* stream) {
  return cast_i64_checked<int32_t>("arx_cast_i64_i32", values, allow_int_overflow, INT32_MIN, INT32_MAX, ws, ws_bytes,
                                   out, stream);
}

int arx_cast_i64_f64(const ArxSpan* values, int allow_float_truncate, void* ws, size_t ws_bytes, double* out,
                     void* stream) {
  // whole numbers are exact in a double up to 2^53 (FloatingIntegerBound<double>, scalar_cast_numeric.cc:213-215)
  return cast_i64_checked<double>("arx_cast_i64_f64", values, allow_float_truncate, -(int64_t(1) << 53),
                                  int64_t(1) << 53, ws, ws_bytes, out, stream);
}

int arx_cast_i32_i64(const int32_t* values, int64_t length, int64_t* out, void* stream) {
  if (length < 0 || (length > 0 && (values == nullptr || out == nullptr))) {
    set_error("bad arguments to arx_cast_i32_i64");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  hipLaunchKernelGGL(cast_i32_i64_kernel, dim3(stream_grid(kBlock, length)), dim3(kBlock), 0, as_stream(stream), values,
                     length, out);
  ARX_CHECK_LAUNCH("cast_i32_i64_kernel");
  return ARX_OK;
}

int arx_greater_f64(const double* left, const double* right, int64_t length, uint64_t* out_bits,
                    void* stream) {
  return launch_greater<double, kArray, kArray>(left, 0.0, right, 0.0, length, out_bits,
                                                as_stream(stream));
}
int arx_greater_f64_array_scalar(const double* left, double right, int64_t length,
                                 uint64_t* out_bits, void* stream) {
  return launch_greater<double, kArray, kScalar>(left, 0.0, nullptr, right, length, out_bits,
                                                 as_stream(stream));
}
int arx_greater_f64_scalar_array(double left, const double* right, int64_t length,
                                 uint64_t* out_bits, void* stream) {
  return launch_greater<double, kScalar, kArray>(nullptr, left, right, 0.0, length, out_bits,
                                                 as_stream(stream));
}
int arx_greater_i64(const int64_t* left, const int64_t* right, int64_t length, uint64_t* out_bits,
                    void* stream) {
  return launch_greater<int64_t, kArray, kArray>(left, 0, right, 0, length, out_bits,
                                                 as_stream(stream));
}

int arx_compare_f64(int op, const double* left, double left_scalar, const double* right, double right_scalar,
                    int64_t length, uint64_t* out_bits, void* stream) {
  return compare_any<double>(op, left, left_scalar, right, right_scalar, length, out_bits, as_stream(stream));
}
int arx_compare_i64(int op, const int64_t* left, int64_t left_scalar, const int64_t* right, int64_t right_scalar,
                    int64_t length, uint64_t* out_bits, void* stream) {
  return compare_any<int64_t>(op, left, left_scalar, right, right_scalar, length, out_bits, as_stream(stream));
}

int arx_arith_i64(int op, const int64_t* left, int64_t left_scalar, const int64_t* right, int64_t right_scalar,
                  int64_t length, int64_t* out, void* stream) {
  const Bits none = make_bits(nullptr, 0, length);
  return arith_any<int64_t, false>(op, left, left_scalar, right, right_scalar, none, none, length, out, nullptr,
                                   as_stream(stream));
}
int arx_arith_f64(int op, const double* left, double left_scalar, const double* right, double right_scalar,
                  int64_t length, double* out, void* stream) {
  const Bits none = make_bits(nullptr, 0, length);
  return arith_any<double, false>(op, left, left_scalar, right, right_scalar, none, none, length, out, nullptr,
                                  as_stream(stream));
}
int arx_arith_checked_i64(int op, const int64_t* left, int64_t left_scalar, const void* left_validity,
                          int64_t left_offset, const int64_t* right, int64_t right_scalar,
                          const void* right_validity, int64_t right_offset, int64_t length, int64_t* out,
                          unsigned int* overflow_flag, void* stream) {
  if (overflow_flag == nullptr) {
    set_error("arx_arith_checked_i64: overflow_flag is NULL");
    return ARX_INVALID;
  }
  const Bits lv = make_bits(left_validity, left_offset, length);
  const Bits rv = make_bits(right_validity, right_offset, length);
  return arith_any<int64_t, true>(op, left, left_scalar, right, right_scalar, lv, rv, length, out, overflow_flag,
                                  as_stream(stream));
}

int arx_copy_segments(const ArxCopySeg* segments, int64_t num_segments, uint64_t max_segment_bytes, void* stream) {
  if (num_segments < 0 || (num_segments > 0 && segments == nullptr)) {
    set_error("bad arguments to arx_copy_segments");
    return ARX_INVALID;
  }
  if (num_segments == 0 || max_segment_bytes == 0) return ARX_OK;
  constexpr uint64_t kSlice = 64 * 1024;   // bytes per workgroup
  const uint64_t per_seg = std::min<uint64_t>((max_segment_bytes + kSlice - 1) / kSlice, 4096);
  const uint64_t blocks = per_seg * static_cast<uint64_t>(num_segments);
  if (blocks > (uint64_t(1) << 31) - 1) {
    set_error("arx_copy_segments: too many segments for one launch (%lld)", static_cast<long long>(num_segments));
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(copy_segments_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, as_stream(stream), segments,
                     num_segments, static_cast<unsigned>(per_seg));
  ARX_CHECK_LAUNCH("copy_segments_kernel");
  return ARX_OK;
}

int arx_bitmap_copy_segments(const ArxBitSeg* segments, int64_t num_segments, int64_t max_segment_bits, void* stream) {
  if (num_segments < 0 || max_segment_bits < 0 || (num_segments > 0 && segments == nullptr)) {
    set_error("bad arguments to arx_bitmap_copy_segments");
    return ARX_INVALID;
  }
  if (num_segments == 0 || max_segment_bits == 0) return ARX_OK;
  const uint64_t words = (static_cast<uint64_t>(max_segment_bits) + 63) / 64;
  const uint64_t per_seg = std::min<uint64_t>((words + kBlock - 1) / kBlock, 4096);
  const uint64_t blocks = per_seg * static_cast<uint64_t>(num_segments);
  if (blocks > (uint64_t(1) << 31) - 1) {
    set_error("arx_bitmap_copy_segments: too many segments for one launch (%lld)", static_cast<long long>(num_segments));
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(bitmap_copy_segments_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, as_stream(stream),
                     segments, num_segments, static_cast<unsigned>(per_seg));
  ARX_CHECK_LAUNCH("bitmap_copy_segments_kernel");
  return ARX_OK;
}

int arx_compare_numeric(int op, int num_type, const void* left, const void* left_scalar, const void* right,
                        const void* right_scalar, int64_t length, uint64_t* out_bits, void* stream) {
  if (length == 0) return ARX_OK;
  if (length < 0 || out_bits == nullptr || (left == nullptr && left_scalar == nullptr) ||
      (right == nullptr && right_scalar == nullptr)) {
    set_error("bad arguments to compare");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  switch (num_type) {
    case ARX_NUM_INT8: return compare_rows_any<int8_t>(op, left, left_scalar, right, right_scalar, length, out_bits, st);
    case ARX_NUM_UINT8: return compare_rows_any<uint8_t>(op, left, left_scalar, right, right_scalar, length, out_bits, st);
    case ARX_NUM_INT16: return compare_rows_any<int16_t>(op, left, left_scalar, right, right_scalar, length, out_bits, st);
    case ARX_NUM_UINT16: return compare_rows_any<uint16_t>(op, left, left_scalar, right, right_scalar, length, out_bits, st);
    case ARX_NUM_INT32: return compare_rows_any<int32_t>(op, left, left_scalar, right, right_scalar, length, out_bits, st);
    case ARX_NUM_UINT32: return compare_rows_any<uint32_t>(op, left, left_scalar, right, right_scalar, length, out_bits, st);
    case ARX_NUM_INT64: return compare_rows_any<int64_t>(op, left, left_scalar, right, right_scalar, length, out_bits, st);
    case ARX_NUM_UINT64: return compare_rows_any<uint64_t>(op, left, left_scalar, right, right_scalar, length, out_bits, st);
    case ARX_NUM_FLOAT32: return compare_rows_any<float>(op, left, left_scalar, right, right_scalar, length, out_bits, st);
    case ARX_NUM_FLOAT64: return compare_rows_any<double>(op, left, left_scalar, right, right_scalar, length, out_bits, st);
    default:
      set_error("compare: unknown numeric type %d", num_type);
      return ARX_NOT_IMPLEMENTED;
  }
}

int arx_arith_numeric(int op, int checked, int num_type, const void* left, const void* left_scalar,
                      const void* left_validity, int64_t left_offset, const void* right, const void* right_scalar,
                      const void* right_validity, int64_t right_offset, int64_t length, void* out,
                      uint32_t* overflow_flag, void* stream) {
  if (length == 0) return ARX_OK;
  if (length < 0 || out == nullptr || (left == nullptr && left_scalar == nullptr) ||
      (right == nullptr && right_scalar == nullptr) || (left == nullptr && right == nullptr)) {
    set_error("bad arguments to arithmetic");
    return ARX_INVALID;
  }
  const bool is_float = num_type == ARX_NUM_FLOAT32 || num_type == ARX_NUM_FLOAT64;
  if (checked && !is_float && overflow_flag == nullptr) {
    set_error("arx_arith_numeric: overflow_flag is NULL");
    return ARX_INVALID;
  }
  hipStream_t st = as_stream(stream);
  const Bits lv = make_bits(left_validity, left_offset, length);
  const Bits rv = make_bits(right_validity, right_offset, length);
#define ARX_ARITH_T(T) return arith_numeric_any<T>(op, checked, left, left_scalar, lv, right, right_scalar, rv, length, out, overflow_flag, st)
  switch (num_type) {
    case ARX_NUM_INT8: ARX_ARITH_T(int8_t);
    case ARX_NUM_UINT8: ARX_ARITH_T(uint8_t);
    case ARX_NUM_INT16: ARX_ARITH_T(int16_t);
    case ARX_NUM_UINT16: ARX_ARITH_T(uint16_t);
    case ARX_NUM_INT32: ARX_ARITH_T(int32_t);
    case ARX_NUM_UINT32: ARX_ARITH_T(uint32_t);
    case ARX_NUM_INT64: ARX_ARITH_T(int64_t);
    case ARX_NUM_UINT64: ARX_ARITH_T(uint64_t);
    case ARX_NUM_FLOAT32: ARX_ARITH_T(float);
    case ARX_NUM_FLOAT64: ARX_ARITH_T(double);
    default:
      set_error("arithmetic: unknown numeric type %d", num_type);
      return ARX_NOT_IMPLEMENTED;
  }
#undef ARX_ARITH_T
}

int arx_divide_i64(const int64_t* left, int64_t left_scalar, const void* left_validity, int64_t left_offset,
                   const int64_t* right, int64_t right_scalar, const void* right_validity, int64_t right_offset,
                   int64_t length, int checked, int64_t* out, uint64_t* errors, void* stream) {
  return divide_any<int64_t>(left, left_scalar, left_validity, left_offset, right, right_scalar, right_validity,
                             right_offset, length, checked, out, errors, as_stream(stream));
}
int arx_divide_f64(const double* left, double left_scalar, const void* left_validity, int64_t left_offset,
                   const double* right, double right_scalar, const void* right_validity, int64_t right_offset,
                   int64_t length, int checked, double* out, uint64_t* errors, void* stream) {
  return divide_any<double>(left, left_scalar, left_validity, left_offset, right, right_scalar, right_validity,
                            right_offset, length, checked, out, errors, as_stream(stream));
}

int arx_divide_numeric(int checked, int num_type, const void* left, const void* left_scalar, const void* left_validity,
                       int64_t left_offset, const void* right, const void* right_scalar, const void* right_validity,
                       int64_t right_offset, int64_t length, void* out, uint64_t* errors, void* stream) {
  if ((left == nullptr && left_scalar == nullptr) || (right == nullptr && right_scalar == nullptr)) {
    set_error("divide: an operand is neither an array nor a scalar");
    return ARX_INVALID;
  }
#define ARX_DIVIDE_T(T)                                                                                                    \
  return divide_any<T>(static_cast<const T*>(left), left == nullptr ? *static_cast<const T*>(left_scalar) : T(0),          \
                       left_validity, left_offset, static_cast<const T*>(right),                                           \
                       right == nullptr ? *static_cast<const T*>(right_scalar) : T(0), right_validity, right_offset,        \
                       length, checked, static_cast<T*>(out), errors, as_stream(stream))
  switch (num_type) {
    case ARX_NUM_INT8: ARX_DIVIDE_T(int8_t);
    case ARX_NUM_UINT8: ARX_DIVIDE_T(uint8_t);
    case ARX_NUM_INT16: ARX_DIVIDE_T(int16_t);
    case ARX_NUM_UINT16: ARX_DIVIDE_T(uint16_t);
    case ARX_NUM_INT32: ARX_DIVIDE_T(int32_t);
    case ARX_NUM_UINT32: ARX_DIVIDE_T(uint32_t);
    case ARX_NUM_INT64: ARX_DIVIDE_T(int64_t);
    case ARX_NUM_UINT64: ARX_DIVIDE_T(uint64_t);
    case ARX_NUM_FLOAT32: ARX_DIVIDE_T(float);
    case ARX_NUM_FLOAT64: ARX_DIVIDE_T(double);
    default:
      set_error("divide: unknown numeric type %d", num_type);
      return ARX_NOT_IMPLEMENTED;
  }
#undef ARX_DIVIDE_T
}

int arx_add_i64(const int64_t* left, const int64_t* right, int64_t length, int64_t* out,
                void* stream) {
  return launch_add<int64_t, false>(left, right, 0, length, out, as_stream(stream));
}
int arx_add_f64(const double* left, const double* right, int64_t length, double* out,
                void* stream) {
  return launch_add<double, false>(left, right, 0.0, length, out, as_stream(stream));
}

int arx_add_i64_array_scalar(const int64_t* left, int64_t right, int64_t length, int64_t* out, void* stream) {
  return launch_add<int64_t, true>(left, nullptr, right, length, out, as_stream(stream));
}
int arx_add_f64_array_scalar(const double* left, double right, int64_t length, double* out, void* stream) {
  return launch_add<double, true>(left, nullptr, right, length, out, as_stream(stream));
}
int arx_greater_i64_array_scalar(const int64_t* left, int64_t right, int64_t length, uint64_t* out_bits,
                                 void* stream) {
  return launch_greater<int64_t, kArray, kScalar>(left, 0, nullptr, right, length, out_bits, as_stream(stream));
}
int arx_greater_i64_scalar_array(int64_t left, const int64_t* right, int64_t length, uint64_t* out_bits,
                                 void* stream) {
  return launch_greater<int64_t, kScalar, kArray>(nullptr, left, right, 0, length, out_bits, as_stream(stream));
}

int arx_buffer_copy(const void* src, void* dst, int64_t nbytes, void* stream) {
  if (nbytes < 0 || (nbytes > 0 && (src == nullptr || dst == nullptr))) {
    set_error("bad arguments to arx_buffer_copy");
    return ARX_INVALID;
  }
  if (nbytes == 0) return ARX_OK;
  const uint64_t sa = reinterpret_cast<uint64_t>(src), da = reinterpret_cast<uint64_t>(dst);
  int64_t head = 0, n16 = 0;
  if (((sa ^ da) & 15) == 0) {   // the two ranges share their 16-byte phase: an aligned middle exists
    head = std::min<int64_t>(nbytes, static_cast<int64_t>((16 - (sa & 15)) & 15));
    n16 = (nbytes - head) / 16;
  } else {
    head = nbytes;   // (byte loop; buffers from hipMalloc / the pool are 256-byte aligned, slices of them rarely differ in phase)
    if (nbytes > (1 << 20)) {
      ARX_HIP(hipMemcpyAsync(dst, src, static_cast<size_t>(nbytes), hipMemcpyDefault, as_stream(stream)));   // (either side may be mapped host memory)
      return ARX_OK;
    }
  }
  const unsigned grid = stream_grid(kBlock, n16);
  hipLaunchKernelGGL(buffer_copy_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), static_cast<const uint8_t*>(src),
                     static_cast<uint8_t*>(dst), head, n16, nbytes);
  ARX_CHECK_LAUNCH("buffer_copy_kernel");
  return ARX_OK;
}

int arx_bitmap_copy(const void* bits, int64_t bit_offset, int64_t length, void* out, void* stream) {
  if (length < 0 || bit_offset < 0 || (length > 0 && out == nullptr)) {
    set_error("bad arguments to arx_bitmap_copy");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  const Bits a = make_bits(bits, bit_offset, length);
  const int64_t nwords = ceil_div(length, 64);
  const unsigned grid = stream_grid(kBlock, nwords);
  hipLaunchKernelGGL((bitmap_kernel<false>), dim3(grid), dim3(kBlock), 0, as_stream(stream), a, a,
                     nwords, static_cast<uint64_t*>(out));
  ARX_CHECK_LAUNCH("bitmap_kernel");
  return ARX_OK;
}

int arx_bitmap_copy_at(const void* bits, int64_t bit_offset, int64_t length, void* out, int64_t out_bit_offset,
                       void* stream) {
  if (length < 0 || bit_offset < 0 || out_bit_offset < 0 || (length > 0 && out == nullptr)) {
    set_error("bad arguments to arx_bitmap_copy_at");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  const Bits a = make_bits(bits, bit_offset, length);
  const int64_t first_word = out_bit_offset / 64;
  const int64_t dst_words = (out_bit_offset + length - 1) / 64 - first_word + 1;
  const unsigned grid = stream_grid(kBlock, dst_words);
  hipLaunchKernelGGL(bitmap_copy_at_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), a, ceil_div(length, 64),
                     first_word, dst_words, static_cast<int>(out_bit_offset % 64), static_cast<uint64_t*>(out));
  ARX_CHECK_LAUNCH("bitmap_copy_at_kernel");
  return ARX_OK;
}

int arx_binary_rebase_offsets(const int32_t* offsets, int64_t length, int32_t base, int32_t* out, void* stream) {
  if (length < 0 || offsets == nullptr || out == nullptr) {
    set_error("bad arguments to arx_binary_rebase_offsets");
    return ARX_INVALID;
  }
  const unsigned grid = stream_grid(kBlock, length + 1);
  hipLaunchKernelGGL(rebase_offsets_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), offsets, length, base, out);
  ARX_CHECK_LAUNCH("rebase_offsets_kernel");
  return ARX_OK;
}

int arx_bitmap_and(const void* left, int64_t left_offset, const void* right, int64_t right_offset,
                   int64_t length, void* out, void* stream) {
  if (length < 0 || left_offset < 0 || right_offset < 0 || (length > 0 && out == nullptr)) {
    set_error("bad arguments to arx_bitmap_and");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  const Bits a = make_bits(left, left_offset, length);
  const Bits b = make_bits(right, right_offset, length);
  const int64_t nwords = ceil_div(length, 64);
  const unsigned grid = stream_grid(kBlock, nwords);
  hipLaunchKernelGGL((bitmap_kernel<true>), dim3(grid), dim3(kBlock), 0, as_stream(stream), a, b,
                     nwords, static_cast<uint64_t*>(out));
  ARX_CHECK_LAUNCH("bitmap_kernel");
  return ARX_OK;
}

int arx_reduce_i64_init(void* acc, void* stream) {
  if (acc == nullptr) {
    set_error("arx_reduce_i64_init: acc is NULL");
    return ARX_INVALID;
  }
  hipLaunchKernelGGL(reduce_i64_init_kernel, dim3(1), dim3(1), 0, as_stream(stream), static_cast<long long*>(acc));
  ARX_CHECK_LAUNCH("reduce_i64_init_kernel");
  return ARX_OK;
}

int arx_reduce_i64_consume(const ArxSpan* values, void* acc, void* stream) {
  if (values == nullptr || acc == nullptr || values->length < 0) {
    set_error("bad arguments to arx_reduce_i64_consume");
    return ARX_INVALID;
  }
  const int64_t n = values->length;
  if (n == 0) return ARX_OK;
  if (values->data == nullptr) {
    set_error("NULL data buffer passed to arx_reduce_i64_consume");
    return ARX_INVALID;
  }
  const void* vbits = values->null_count != 0 ? values->validity : nullptr;
  const int64_t* in = static_cast<const int64_t*>(values->data) + values->offset;
  // the kernel reads 16-byte pairs: peel a leading row that is only 8-byte aligned
  const int64_t head = (reinterpret_cast<uint64_t>(in) & 15) != 0 ? 1 : 0;
  if (head) {
    const Bits v1 = make_bits(vbits, values->offset, 1);
    hipLaunchKernelGGL(reduce_i64_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), in, v1, int64_t(1),
                       static_cast<unsigned long long*>(acc));
  }
  if (n - head > 0) {
    const Bits valid = make_bits(vbits, values->offset + head, n - head);
    hipLaunchKernelGGL(reduce_i64_kernel, dim3(stream_grid(kBlock * 8, n - head)), dim3(kBlock), 0, as_stream(stream),
                       in + head, valid, n - head, static_cast<unsigned long long*>(acc));
  }
  ARX_CHECK_LAUNCH("reduce_i64_kernel");
  return ARX_OK;
}

int arx_boolean_kleene(int op, const ArxSpan* left, const ArxSpan* right, void* out_data, void* out_validity,
                       void* stream) {
  if (left == nullptr || right == nullptr || (op != ARX_AND_KLEENE && op != ARX_OR_KLEENE)) {
    set_error("bad arguments to arx_boolean_kleene");
    return ARX_INVALID;
  }
  if (left->length != right->length) {
    set_error("Array arguments must all be the same length (%lld vs %lld)", static_cast<long long>(left->length),
              static_cast<long long>(right->length));
    return ARX_INVALID;
  }
  const int64_t n = left->length;
  if (n == 0) return ARX_OK;
  if (left->data == nullptr || right->data == nullptr || out_data == nullptr) {
    set_error("NULL buffer passed to arx_boolean_kleene");
    return ARX_INVALID;
  }
  const void* lvp = left->null_count != 0 ? left->validity : nullptr;
  const void* rvp = right->null_count != 0 ? right->validity : nullptr;
  if ((lvp != nullptr || rvp != nullptr) && out_validity == nullptr) {
    set_error("arx_boolean_kleene: inputs may have nulls but out_validity is NULL");
    return ARX_INVALID;
  }
  const Bits ld = make_bits(left->data, left->offset, n), lv = make_bits(lvp, left->offset, n);
  const Bits rd = make_bits(right->data, right->offset, n), rv = make_bits(rvp, right->offset, n);
  const int64_t nwords = ceil_div(n, 64);
  const unsigned grid = stream_grid(kBlock, nwords);
  if (op == ARX_AND_KLEENE) {
    hipLaunchKernelGGL((kleene_kernel<ARX_AND_KLEENE>), dim3(grid), dim3(kBlock), 0, as_stream(stream), ld, lv, rd,
                       rv, nwords, static_cast<uint64_t*>(out_data), static_cast<uint64_t*>(out_validity));
  } else {
    hipLaunchKernelGGL((kleene_kernel<ARX_OR_KLEENE>), dim3(grid), dim3(kBlock), 0, as_stream(stream), ld, lv, rd,
                       rv, nwords, static_cast<uint64_t*>(out_data), static_cast<uint64_t*>(out_validity));
  }
  ARX_CHECK_LAUNCH("kleene_kernel");
  return ARX_OK;
}

int arx_boolean_invert(const void* bits, int64_t bit_offset, int64_t length, void* out, void* stream) {
  if (length < 0 || bit_offset < 0 || (length > 0 && (bits == nullptr || out == nullptr))) {
    set_error("bad arguments to arx_boolean_invert");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  const Bits a = make_bits(bits, bit_offset, length);
  const Bits ones = make_bits(nullptr, 0, length);
  const int64_t nwords = ceil_div(length, 64);
  hipLaunchKernelGGL(invert_kernel, dim3(stream_grid(kBlock, nwords)), dim3(kBlock), 0, as_stream(stream), a, ones,
                     nwords, static_cast<uint64_t*>(out));
  ARX_CHECK_LAUNCH("invert_kernel");
  return ARX_OK;
}

int arx_bytes_to_bitmap(const uint8_t* bytes, int64_t length, void* out_bits, int64_t* set_count, void* stream) {
  if (length < 0 || (length > 0 && (bytes == nullptr || out_bits == nullptr))) {
    set_error("bad arguments to arx_bytes_to_bitmap");
    return ARX_INVALID;
  }
  if (length == 0) return ARX_OK;
  const int64_t nwords = ceil_div(length, 64);
  const unsigned grid = static_cast<unsigned>(std::min<int64_t>(ceil_div(nwords * 64, kBlock), 2048));
  hipLaunchKernelGGL(bytes_to_bitmap_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream),
                     bytes, length, static_cast<uint64_t*>(out_bits), reinterpret_cast<unsigned long long*>(set_count));
  ARX_CHECK_LAUNCH("bytes_to_bitmap_kernel");
  return ARX_OK;
}

int arx_bitmap_popcount(const void* bits, int64_t bit_offset, int64_t length, void* ws,
                        size_t ws_bytes, int64_t* out_count, void* stream) {
  if (length < 0 || bit_offset < 0 || out_count == nullptr || ws == nullptr || ws_bytes < 8) {
    set_error("bad arguments to arx_bitmap_popcount");
    return ARX_INVALID;
  }
  if (length == 0) {
    *out_count = 0;
    return ARX_OK;
  }
  hipStream_t st = as_stream(stream);
  const Bits a = make_bits(bits, bit_offset, length);
  const int64_t nwords = ceil_div(length, 64);
  ARX_HIP(hipMemsetAsync(ws, 0, 8, st));
  const unsigned grid = std::min(stream_grid(kBlock, nwords), 2048u);   // (grid-stride loop; few atomics)
  hipLaunchKernelGGL(popcount_kernel, dim3(grid), dim3(kBlock), 0, st, a, nwords,
                     static_cast<unsigned long long*>(ws));
  ARX_CHECK_LAUNCH("popcount_kernel");
  int64_t total = 0;
  ARX_HIP(hipMemcpyAsync(&total, ws, 8, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  *out_count = total;
  return ARX_OK;
}

}  // extern "C"
