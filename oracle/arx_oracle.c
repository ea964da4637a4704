/*
 * arx_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the arrow::compute kernels on the hot
 * path, used only as the checker in tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Nothing under arrow_amd/ may import, link or call this file.
 *
 * Pinned: every function here is checked against the reference's own build
 * (pyarrow 25.0.0 = libarrow.so.2500, see tests/test_oracle_pin.py and
 * tests/golden/make_golden.py) and against golden vectors transcribed from the
 * reference's unit tests (tests/golden/reference_vectors.json).
 *
 * Each function cites the reference source it follows (paths relative to
 * /root/reference/cpp/src/arrow).  The restatement is row-at-a-time: it states WHAT
 * the reference computes, byte for byte, not how the reference blocks its loops.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int get_bit(const uint8_t* bits, int64_t i) {
  return (bits[i >> 3] >> (i & 7)) & 1;
}
static inline void set_bit_to(uint8_t* bits, int64_t i, int v) {
  if (v) bits[i >> 3] |= (uint8_t)(1u << (i & 7));
  else bits[i >> 3] &= (uint8_t)~(1u << (i & 7));
}
/* validity == NULL means "all valid" (ArraySpan::MayHaveNulls, array/data.h) */
static inline int is_valid(const uint8_t* validity, int64_t off, int64_t i) {
  return validity == NULL ? 1 : get_bit(validity, off + i);
}

/* ---------------------------------------------------------------------------
 * GetBitmapFilterOutputSize — compute/kernels/vector_selection_filter_internal.cc:62-91
 * null_selection: 0 = DROP (count mask & valid), 1 = EMIT_NULL (count mask | !valid)
 * ------------------------------------------------------------------------- */
int64_t arxo_filter_output_size(const uint8_t* mask, const uint8_t* mask_valid, int64_t mask_off,
                                int64_t length, int null_selection) {
  int64_t n = 0;
  for (int64_t i = 0; i < length; ++i) {
    const int v = is_valid(mask_valid, mask_off, i);
    const int m = get_bit(mask, mask_off + i);
    if (null_selection == 1) n += (m || !v);
    else n += (m && v);
  }
  return n;
}

/* ---------------------------------------------------------------------------
 * PrimitiveFilterImpl<W>::Exec — vector_selection_filter_internal.cc:238-372
 *   selected & mask valid  -> copy the W source bytes (even if the value is null,
 *                             WriteMaybeNull -> WriteValue :267-272,376-385);
 *                             out validity bit = value validity
 *   mask null & EMIT_NULL  -> W zero bytes, out validity bit 0 (WriteNull :398-407)
 * out_valid may be NULL (no validity allocated, :472).  Returns the output length.
 * `values` points at the buffer start (values_off applied here).
 * ------------------------------------------------------------------------- */
int64_t arxo_filter(const uint8_t* values, int byte_width, const uint8_t* values_valid,
                    int64_t values_off, const uint8_t* mask, const uint8_t* mask_valid,
                    int64_t mask_off, int64_t length, int null_selection, uint8_t* out_data,
                    uint8_t* out_valid) {
  int64_t o = 0;
  for (int64_t i = 0; i < length; ++i) {
    const int mv = is_valid(mask_valid, mask_off, i);
    const int m = get_bit(mask, mask_off + i);
    if (mv && m) {
      memcpy(out_data + o * byte_width, values + (values_off + i) * byte_width, (size_t)byte_width);
      if (out_valid) set_bit_to(out_valid, o, is_valid(values_valid, values_off, i));
      ++o;
    } else if (!mv && null_selection == 1) {
      memset(out_data + o * byte_width, 0, (size_t)byte_width);
      if (out_valid) set_bit_to(out_valid, o, 0);
      ++o;
    }
  }
  return o;
}

/* ---------------------------------------------------------------------------
 * GetTakeIndicesFromBitmapImpl — compute/kernels/vector_selection_take_internal.cc:62-168
 * index_width 2 (uint16) or 4 (uint32).  EMIT_NULL + null mask slot -> null index
 * (value 0 from UnsafeAppendNull).  out_valid may be NULL.  Returns the length.
 * ------------------------------------------------------------------------- */
int64_t arxo_mask_to_indices(const uint8_t* mask, const uint8_t* mask_valid, int64_t mask_off,
                             int64_t length, int null_selection, int index_width, void* out,
                             uint8_t* out_valid) {
  int64_t o = 0;
  for (int64_t i = 0; i < length; ++i) {
    const int mv = is_valid(mask_valid, mask_off, i);
    const int m = get_bit(mask, mask_off + i);
    int emit = 0, valid = 1;
    uint32_t val = (uint32_t)i;
    if (mv && m) emit = 1;
    else if (!mv && null_selection == 1) { emit = 1; valid = 0; val = 0; }
    if (!emit) continue;
    if (index_width == 2) ((uint16_t*)out)[o] = (uint16_t)val;
    else ((uint32_t*)out)[o] = val;
    if (out_valid) set_bit_to(out_valid, o, valid);
    ++o;
  }
  return o;
}

/* index types: 0 u8, 1 i8, 2 u16, 3 i16, 4 u32, 5 i32, 6 u64, 7 i64 */
static inline int index_width_of(int t) { static const int w[8] = {1,1,2,2,4,4,8,8}; return w[t]; }
static inline int64_t load_index_signed(const void* p, int t, int64_t i) {
  switch (t) {
    case 0: return ((const uint8_t*)p)[i];
    case 1: return ((const int8_t*)p)[i];
    case 2: return ((const uint16_t*)p)[i];
    case 3: return ((const int16_t*)p)[i];
    case 4: return ((const uint32_t*)p)[i];
    case 5: return ((const int32_t*)p)[i];
    case 6: return (int64_t)((const uint64_t*)p)[i];
    default: return ((const int64_t*)p)[i];
  }
}

/* ---------------------------------------------------------------------------
 * CheckIndexBounds — util/int_util.cc:530-587.  Only valid (non-null) slots are
 * checked.  Returns 0 if all in bounds, else 1 and the FIRST offending index
 * (as it would be printed: "Index N out of bounds", :554).
 * `indices` is the buffer start (idx_off applied here).
 * ------------------------------------------------------------------------- */
int arxo_check_index_bounds(const void* indices, int index_type, const uint8_t* idx_valid,
                            int64_t idx_off, int64_t length, uint64_t upper_limit,
                            int64_t* bad_signed, uint64_t* bad_unsigned) {
  const uint8_t* base = (const uint8_t*)indices + idx_off * index_width_of(index_type);
  const int is_signed = index_type & 1;
  for (int64_t i = 0; i < length; ++i) {
    if (!is_valid(idx_valid, idx_off, i)) continue;
    const int64_t s = load_index_signed(base, index_type, i);
    int oob;
    if (is_signed) oob = (s < 0) || ((uint64_t)s >= upper_limit);
    else oob = ((uint64_t)s >= upper_limit);
    if (oob) {
      *bad_signed = s;
      *bad_unsigned = (uint64_t)s;
      if (index_type == 6) *bad_unsigned = ((const uint64_t*)base)[i];
      return 1;
    }
  }
  return 0;
}

/* ---------------------------------------------------------------------------
 * FixedWidthTakeImpl::Exec / Gather::ExecuteWithNulls —
 * compute/kernels/vector_selection_take_internal.cc:339-380, gather_internal.h:84-165.
 * out validity bit = index valid AND source value valid; every null output slot is
 * zero-filled (WriteZero).  out_valid may be NULL when neither side may have nulls.
 * Signed indices are read as unsigned after the bounds check (:383-400).
 * Returns valid_count (null_count = length - valid_count, :377).
 * ------------------------------------------------------------------------- */
int64_t arxo_take(const uint8_t* values, int byte_width, const uint8_t* values_valid,
                  int64_t values_off, const void* indices, int index_type,
                  const uint8_t* idx_valid, int64_t idx_off, int64_t length, uint8_t* out_data,
                  uint8_t* out_valid) {
  const uint8_t* ibase = (const uint8_t*)indices + idx_off * index_width_of(index_type);
  int64_t valid_count = 0;
  if (out_valid) memset(out_valid, 0, (size_t)((length + 7) / 8));
  for (int64_t i = 0; i < length; ++i) {
    int ok = is_valid(idx_valid, idx_off, i);
    uint64_t idx = 0;
    if (ok) {
      idx = (uint64_t)load_index_signed(ibase, index_type, i);
      if (index_type < 6) {
        /* reinterpret the narrow signed types as unsigned of the same width */
        const int w = index_width_of(index_type);
        if (w == 1) idx &= 0xffu; else if (w == 2) idx &= 0xffffu; else idx &= 0xffffffffu;
      }
      ok = is_valid(values_valid, values_off, (int64_t)idx);
    }
    if (ok) {
      memcpy(out_data + i * byte_width, values + (values_off + (int64_t)idx) * byte_width,
             (size_t)byte_width);
      if (out_valid) set_bit_to(out_valid, i, 1);
      ++valid_count;
    } else {
      memset(out_data + i * byte_width, 0, (size_t)byte_width);
    }
  }
  return valid_count;
}

/* ---------------------------------------------------------------------------
 * Take on base-binary values (binary / utf8, int32 offsets) — VarBinaryTakeImpl /
 * TakeExec for base binary (compute/kernels/vector_selection_take_internal.cc, Selection CRTP in
 * vector_selection_internal.cc): out slot i is valid iff index i is valid AND the source value is
 * valid; a valid slot appends the source bytes, a null slot appends nothing (its offsets are
 * equal), out offsets start at 0.  Total bytes must fit int32 (the reference raises on overflow).
 * The same rule is what Filter produces for base-binary values
 * (vector_selection_filter_internal.cc:517-800: BinaryFilterImpl): filter == take(GetTakeIndices).
 * `offsets` is the buffer start (values_off applied here).  out_offsets: length+1 entries.
 * Returns total bytes, or -1 on int32 overflow.
 * ------------------------------------------------------------------------- */
int64_t arxo_binary_take(const int32_t* offsets, const uint8_t* data, const uint8_t* values_valid,
                         int64_t values_off, const void* indices, int index_type,
                         const uint8_t* idx_valid, int64_t idx_off, int64_t length,
                         int32_t* out_offsets, uint8_t* out_data, uint8_t* out_valid,
                         int64_t* out_valid_count) {
  const uint8_t* ibase = (const uint8_t*)indices + idx_off * index_width_of(index_type);
  int64_t total = 0, valid_count = 0;
  if (out_valid) memset(out_valid, 0, (size_t)((length + 7) / 8));
  for (int64_t i = 0; i < length; ++i) {
    out_offsets[i] = (int32_t)total;
    int ok = is_valid(idx_valid, idx_off, i);
    uint64_t idx = 0;
    if (ok) {
      idx = (uint64_t)load_index_signed(ibase, index_type, i);
      if (index_type < 6) {
        const int w = index_width_of(index_type);
        if (w == 1) idx &= 0xffu; else if (w == 2) idx &= 0xffffu; else idx &= 0xffffffffu;
      }
      ok = is_valid(values_valid, values_off, (int64_t)idx);
    }
    if (ok) {
      const int32_t b = offsets[values_off + (int64_t)idx];
      const int32_t e = offsets[values_off + (int64_t)idx + 1];
      if (out_data) memcpy(out_data + total, data + b, (size_t)(e - b));
      total += e - b;
      if (total > 2147483647LL) return -1;
      if (out_valid) set_bit_to(out_valid, i, 1);
      ++valid_count;
    }
  }
  out_offsets[length] = (int32_t)total;
  if (out_valid_count) *out_valid_count = valid_count;
  return total;
}

/* ---------------------------------------------------------------------------
 * CastPrimitive<FloatType, DoubleType>::Exec — compute/kernels/scalar_cast_internal.cc:41-53
 * static_cast<float>(double) on every slot (IEEE RNE via cvtsd2ss on x86).
 * ------------------------------------------------------------------------- */
void arxo_cast_f64_f32(const double* in, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = (float)in[i];
}

/* ---------------------------------------------------------------------------
 * ComparePrimitiveArrayArray<DoubleType, Greater> — compute/kernels/scalar_compare.cc:165-190
 * (Greater::Call :58-64).  LSB-first packed bits; out must hold ceil(n/8) bytes,
 * zero-initialised by the caller like KernelContext::AllocateBitmap (compute/kernel.cc:52-60).
 * right_is_scalar / left_is_scalar: broadcast element 0 (ArrayScalar / ScalarArray :192-247).
 * ------------------------------------------------------------------------- */
void arxo_greater_f64(const double* left, int left_is_scalar, const double* right,
                      int right_is_scalar, int64_t n, uint8_t* out_bits) {
  for (int64_t i = 0; i < n; ++i) {
    const double l = left_is_scalar ? left[0] : left[i];
    const double r = right_is_scalar ? right[0] : right[i];
    set_bit_to(out_bits, i, l > r);
  }
}
void arxo_greater_i64(const int64_t* left, const int64_t* right, int64_t n, uint8_t* out_bits) {
  for (int64_t i = 0; i < n; ++i) set_bit_to(out_bits, i, left[i] > right[i]);
}

/* Add::Call — compute/kernels/base_arithmetic_internal.h:45-80 (unchecked: wraps). */
void arxo_add_i64(const int64_t* l, const int64_t* r, int64_t n, int64_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = (int64_t)((uint64_t)l[i] + (uint64_t)r[i]);
}
void arxo_add_f64(const double* l, const double* r, int64_t n, double* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = l[i] + r[i];
}

/* BitmapAnd / CopyBitmap with offsets — util/bitmap_ops.cc; out offset 0, zero padded. */
void arxo_bitmap_and(const uint8_t* a, int64_t a_off, const uint8_t* b, int64_t b_off, int64_t n,
                     uint8_t* out) {
  memset(out, 0, (size_t)(((n + 63) / 64) * 8));
  for (int64_t i = 0; i < n; ++i) set_bit_to(out, i, is_valid(a, a_off, i) && is_valid(b, b_off, i));
}
int64_t arxo_bitmap_popcount(const uint8_t* a, int64_t off, int64_t n) {
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i) c += get_bit(a, off + i);
  return c;
}

/* ---------------------------------------------------------------------------
 * ArraySortIndices<UInt64Type,*>::Exec — compute/kernels/vector_array_sort.cc:524-540
 *   iota, PartitionNullsOnly<StablePartitioner> (vector_sort_internal.h:225-293),
 *   then std::stable_sort with `lhs < rhs` (ascending) or `rhs < lhs` (descending)
 *   (ArrayCompareSorter :144-178).  The counting-sort branch (:277-362) is stable too
 *   and yields the identical permutation, so one restatement covers both.
 * order: 0 ascending, 1 descending.  null_placement: 0 at start, 1 at end.
 * Stable merge sort on (key, index) gives exactly std::stable_sort's result.
 * ------------------------------------------------------------------------- */
typedef struct { uint64_t key; uint64_t idx; } KeyIdx;

static int less_u64(uint64_t a, uint64_t b, int is_signed) {
  if (is_signed) return (int64_t)a < (int64_t)b;
  return a < b;
}

static void merge_sort(KeyIdx* a, KeyIdx* tmp, int64_t n, int is_signed, int descending) {
  if (n < 2) return;
  const int64_t h = n / 2;
  merge_sort(a, tmp, h, is_signed, descending);
  merge_sort(a + h, tmp, n - h, is_signed, descending);
  int64_t i = 0, j = h, k = 0;
  while (i < h && j < n) {
    /* take from the right run only if it is strictly "before" the left element */
    int right_first = descending ? less_u64(a[i].key, a[j].key, is_signed)
                                 : less_u64(a[j].key, a[i].key, is_signed);
    if (right_first) tmp[k++] = a[j++]; else tmp[k++] = a[i++];
  }
  while (i < h) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, (size_t)n * sizeof(KeyIdx));
}

int arxo_sort_indices_64(const uint64_t* values, const uint8_t* valid, int64_t off, int64_t n,
                         int is_signed, int order, int null_placement, uint64_t* out) {
  KeyIdx* a = (KeyIdx*)malloc((size_t)(n > 0 ? n : 1) * sizeof(KeyIdx));
  KeyIdx* tmp = (KeyIdx*)malloc((size_t)(n > 0 ? n : 1) * sizeof(KeyIdx));
  if (!a || !tmp) { free(a); free(tmp); return -1; }
  int64_t nn = 0, nnull = 0;
  for (int64_t i = 0; i < n; ++i) nnull += !is_valid(valid, off, i);
  const int64_t null_begin = null_placement == 0 ? 0 : n - nnull;
  const int64_t val_begin = null_placement == 0 ? nnull : 0;
  int64_t k = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (is_valid(valid, off, i)) { a[nn].key = values[off + i]; a[nn].idx = (uint64_t)i; ++nn; }
    else out[null_begin + k++] = (uint64_t)i;
  }
  merge_sort(a, tmp, nn, is_signed, order == 1);
  for (int64_t i = 0; i < nn; ++i) out[val_begin + i] = a[i].idx;
  free(a); free(tmp);
  return 0;
}

/* ---------------------------------------------------------------------------
 * array_sort_indices for the other fixed-width key types (uint32/int32/float64/float32) —
 * ArraySortIndices<*, T>::Exec (vector_array_sort.cc:524-540) with PartitionNulls
 * (vector_sort_internal.h:225-293): for floating point keys NaNs are "null-likes": they are
 * stably partitioned next to the nulls — values, NaNs, nulls (at_end) or nulls, NaNs, values
 * (at_start) — whatever the sort order; -0.0 and 0.0 compare equal (plain `<`), so they keep row
 * order.  key_type: 0 uint64, 1 int64, 2 uint32, 3 int32, 4 float64, 5 float32.
 * ------------------------------------------------------------------------- */
#include <math.h>

typedef struct { double f; uint64_t u; int64_t s; uint64_t idx; } TypedKey;

static int typed_less(const TypedKey* a, const TypedKey* b, int key_type) {
  switch (key_type) {
    case 0: case 2: return a->u < b->u;
    case 1: case 3: return a->s < b->s;
    default: return a->f < b->f;
  }
}

static void typed_merge_sort(TypedKey* a, TypedKey* tmp, int64_t n, int key_type, int descending) {
  if (n < 2) return;
  const int64_t h = n / 2;
  typed_merge_sort(a, tmp, h, key_type, descending);
  typed_merge_sort(a + h, tmp, n - h, key_type, descending);
  int64_t i = 0, j = h, k = 0;
  while (i < h && j < n) {
    const int right_first = descending ? typed_less(&a[i], &a[j], key_type) : typed_less(&a[j], &a[i], key_type);
    if (right_first) tmp[k++] = a[j++]; else tmp[k++] = a[i++];
  }
  while (i < h) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, (size_t)n * sizeof(TypedKey));
}

int arxo_sort_indices(const void* values, int key_type, const uint8_t* valid, int64_t off, int64_t n,
                      int order, int null_placement, uint64_t* out) {
  TypedKey* a = (TypedKey*)malloc((size_t)(n > 0 ? n : 1) * sizeof(TypedKey));
  TypedKey* tmp = (TypedKey*)malloc((size_t)(n > 0 ? n : 1) * sizeof(TypedKey));
  uint64_t* nans = (uint64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(uint64_t));
  uint64_t* nulls = (uint64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(uint64_t));
  if (!a || !tmp || !nans || !nulls) { free(a); free(tmp); free(nans); free(nulls); return -1; }
  int64_t nv = 0, nnan = 0, nnull = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (!is_valid(valid, off, i)) { nulls[nnull++] = (uint64_t)i; continue; }
    TypedKey t; t.f = 0; t.u = 0; t.s = 0; t.idx = (uint64_t)i;
    switch (key_type) {
      case 0: t.u = ((const uint64_t*)values)[off + i]; break;
      case 1: t.s = ((const int64_t*)values)[off + i]; break;
      case 2: t.u = ((const uint32_t*)values)[off + i]; break;
      case 3: t.s = ((const int32_t*)values)[off + i]; break;
      case 4: t.f = ((const double*)values)[off + i]; break;
      default: t.f = (double)((const float*)values)[off + i]; break;
    }
    if (key_type >= 4 && isnan(t.f)) { nans[nnan++] = (uint64_t)i; continue; }
    a[nv++] = t;
  }
  typed_merge_sort(a, tmp, nv, key_type, order == 1);
  int64_t o = 0;
  if (null_placement == 0) {   /* nulls, NaNs, values */
    for (int64_t i = 0; i < nnull; ++i) out[o++] = nulls[i];
    for (int64_t i = 0; i < nnan; ++i) out[o++] = nans[i];
    for (int64_t i = 0; i < nv; ++i) out[o++] = a[i].idx;
  } else {                     /* values, NaNs, nulls */
    for (int64_t i = 0; i < nv; ++i) out[o++] = a[i].idx;
    for (int64_t i = 0; i < nnan; ++i) out[o++] = nans[i];
    for (int64_t i = 0; i < nnull; ++i) out[o++] = nulls[i];
  }
  free(a); free(tmp); free(nans); free(nulls);
  return 0;
}

/* ---------------------------------------------------------------------------
 * Group-by hash_sum(int64) BY int32 — the composition
 *   Grouper::Consume (compute/row/grouper.cc:662-815): dense uint32 group ids in
 *     first-occurrence order; a null key is its own group (:448-458 of key_hash);
 *   GroupedReducingAggregator<Int64Type,GroupedSumImpl>::Consume
 *     (compute/kernels/hash_aggregate_numeric.cc:70-83, Reduce :283-287): wrap-around
 *     sum, counts++, null value clears no_nulls;
 *   Finalize (:130-152) + Finish (:109-128): null where count < min_count, and where
 *     any null was seen if !skip_nulls.
 * The hash function is not observable; an open-addressing table keyed by the int32
 * is used here.  Outputs are in first-occurrence order.  Returns num_groups (or -1).
 * out_* arrays must hold `length` entries (upper bound on groups).
 * ------------------------------------------------------------------------- */
int64_t arxo_groupby_sum_i64(const int32_t* keys, const uint8_t* key_valid, int64_t key_off,
                             const int64_t* values, const uint8_t* val_valid, int64_t val_off,
                             int64_t length, int skip_nulls, uint32_t min_count,
                             int32_t* out_keys, uint8_t* out_key_is_valid, int64_t* out_sums,
                             int64_t* out_counts, uint8_t* out_no_nulls, uint8_t* out_valid) {
  uint64_t cap = 16;
  while (cap < (uint64_t)length * 2 + 2) cap <<= 1;
  int64_t* slot_gid = (int64_t*)malloc(cap * sizeof(int64_t));
  if (!slot_gid) return -1;
  for (uint64_t i = 0; i < cap; ++i) slot_gid[i] = -1;
  int64_t null_gid = -1, ng = 0;
  for (int64_t i = 0; i < length; ++i) {
    int64_t g;
    if (!is_valid(key_valid, key_off, i)) {
      if (null_gid < 0) {
        null_gid = ng++;
        out_keys[null_gid] = 0; out_key_is_valid[null_gid] = 0;
        out_sums[null_gid] = 0; out_counts[null_gid] = 0; out_no_nulls[null_gid] = 1;
      }
      g = null_gid;
    } else {
      const int32_t key = keys[key_off + i];
      uint64_t h = ((uint64_t)(uint32_t)key * 0x9E3779B185EBCA87ull) >> 17;
      h &= cap - 1;
      for (;;) {
        if (slot_gid[h] < 0) {
          g = ng++;
          slot_gid[h] = g;
          out_keys[g] = key; out_key_is_valid[g] = 1;
          out_sums[g] = 0; out_counts[g] = 0; out_no_nulls[g] = 1;
          break;
        }
        if (out_keys[slot_gid[h]] == key && out_key_is_valid[slot_gid[h]]) { g = slot_gid[h]; break; }
        h = (h + 1) & (cap - 1);
      }
    }
    if (is_valid(val_valid, val_off, i)) {
      out_sums[g] = (int64_t)((uint64_t)out_sums[g] + (uint64_t)values[val_off + i]);
      out_counts[g] += 1;
    } else {
      out_no_nulls[g] = 0;
    }
  }
  for (int64_t g = 0; g < ng; ++g) {
    int v = out_counts[g] >= (int64_t)min_count;
    if (!skip_nulls) v = v && out_no_nulls[g];
    out_valid[g] = (uint8_t)v;
  }
  free(slot_gid);
  return ng;
}

/* ---------------------------------------------------------------------------
 * hash_sum(int64, uint32 group id) with dense ids — the HashAggregateKernel vtable
 * of GroupedReducingAggregator<Int64Type,GroupedSumImpl>
 * (compute/kernels/hash_aggregate_numeric.cc):
 *   Consume :70-83   (VisitGroupedValues, hash_aggregate_internal.h:140-176: a scalar
 *                     value argument is broadcast; a null scalar nulls every row)
 *   Merge   :85-107  (group_id_mapping[other_g] -> this group)
 *   Finalize:130-152 + Finish :109-128
 * State: sums/counts/no_nulls(one byte per group) of num_groups entries.
 * ------------------------------------------------------------------------- */
void arxo_hash_sum_i64_consume(const int64_t* values, const uint8_t* val_valid, int64_t val_off,
                               int values_is_scalar, int64_t scalar_value, int scalar_is_valid,
                               const uint32_t* group_ids, int64_t length, int64_t* sums,
                               int64_t* counts, uint8_t* no_nulls) {
  for (int64_t i = 0; i < length; ++i) {
    const uint32_t g = group_ids[i];
    int ok;
    int64_t v;
    if (values_is_scalar) { ok = scalar_is_valid; v = scalar_value; }
    else { ok = is_valid(val_valid, val_off, i); v = ok ? values[val_off + i] : 0; }
    if (ok) {
      sums[g] = (int64_t)((uint64_t)sums[g] + (uint64_t)v);
      counts[g] += 1;
    } else {
      no_nulls[g] = 0;
    }
  }
}

void arxo_hash_sum_i64_merge(int64_t* sums, int64_t* counts, uint8_t* no_nulls,
                             const int64_t* other_sums, const int64_t* other_counts,
                             const uint8_t* other_no_nulls, const uint32_t* mapping,
                             int64_t other_num_groups) {
  for (int64_t og = 0; og < other_num_groups; ++og) {
    const uint32_t g = mapping[og];
    counts[g] += other_counts[og];
    sums[g] = (int64_t)((uint64_t)sums[g] + (uint64_t)other_sums[og]);
    no_nulls[g] = (uint8_t)(no_nulls[g] && other_no_nulls[og]);
  }
}

/* out_valid: one byte per group.  Returns the null count. */
int64_t arxo_hash_sum_i64_finalize(const int64_t* counts, const uint8_t* no_nulls,
                                   int64_t num_groups, int skip_nulls, uint32_t min_count,
                                   uint8_t* out_valid) {
  int64_t nulls = 0;
  for (int64_t g = 0; g < num_groups; ++g) {
    int v = counts[g] >= (int64_t)min_count;
    if (!skip_nulls) v = v && no_nulls[g];
    out_valid[g] = (uint8_t)v;
    nulls += !v;
  }
  return nulls;
}
