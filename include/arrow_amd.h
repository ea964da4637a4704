/*
 * arrow_amd.h — the C ABI of libarrow_amd.so: MI355X (gfx950) execution of the
 * arrow::compute vectorized-kernel hot path (filter, take, cast, compare,
 * sort_indices, hash_sum group-by).
 *
 * This is the drop-in boundary.  Arrow's kernels are C++ function
 * pointers of type  ArrayKernelExec = Status(*)(KernelContext*, const ExecSpan&, ExecResult*)
 * (cpp/src/arrow/compute/kernel.h:556) registered with Function::AddKernel
 * (cpp/src/arrow/compute/function.h:316,346,378).  A registration shim (see
 * arrow_amd/csrc/arrow_plugin.cc and INTEGRATION.md) unpacks ExecSpan/ArraySpan
 * (cpp/src/arrow/array/data.h:525-553) into the plain structs below and calls
 * these entry points; nothing here mentions an Arrow C++ type, a torch type or a
 * HIP type (streams travel as void*).
 *
 * Conventions
 *  - All data pointers are DEVICE pointers (HBM) unless a parameter is documented
 *    as "host".  The layout of an array is Arrow's columnar format
 *    (docs/source/format/Columnar.rst): a validity bitmap (LSB-first, 1 = valid,
 *    may be NULL = all valid) plus a fixed-width values buffer, plus a logical
 *    element `offset` that applies to both — the same fields as
 *    struct ArrowArray in cpp/src/arrow/c/abi.h:44-65.
 *  - Bitmaps are read and written in aligned 64-bit words: every bitmap buffer
 *    must be addressable through the enclosing 8-byte-aligned words (true of any
 *    hipMalloc/Arrow allocation; Arrow pads buffers to 64 bytes,
 *    cpp/src/arrow/memory_pool.h).  Output bitmaps start at bit 0 and their
 *    padding bits up to the next 64-bit boundary are written as 0.
 *  - Every function returns ARX_OK (0) or a negative ArxStatus; the message is
 *    available from arx_last_error() (thread-local).  No exceptions cross the
 *    boundary.  Status codes mirror arrow::StatusCode (cpp/src/arrow/status.h:83-107).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All
 *    kernels are enqueued on it; functions documented as "synchronous" also wait
 *    for it.  No function allocates device memory: scratch space is caller
 *    provided (`*_workspace_bytes` + `ws`), so the calls are re-entrant and safe
 *    from several host threads with distinct workspaces/streams (the threading
 *    contract of Kernel::exec).
 */
#ifndef ARROW_AMD_H_
#define ARROW_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARX_ABI_VERSION 3

/* arrow::StatusCode twins (cpp/src/arrow/status.h:83-107). */
typedef enum ArxStatus {
  ARX_OK = 0,
  ARX_OUT_OF_MEMORY = -1,
  ARX_INVALID = -4,          /* Status::Invalid       */
  ARX_CAPACITY_ERROR = -6,   /* Status::CapacityError (a caller-sized output region was too small; nothing else was touched) */
  ARX_INDEX_ERROR = -7,      /* Status::IndexError    */
  ARX_NOT_IMPLEMENTED = -10, /* Status::NotImplemented*/
  ARX_DEVICE_ERROR = -100    /* a hip* call failed    */
} ArxStatus;

/* FilterOptions::NullSelectionBehavior (cpp/src/arrow/compute/api_vector.h:37-52). */
enum { ARX_FILTER_DROP = 0, ARX_FILTER_EMIT_NULL = 1 };
/* SortOrder / NullPlacement (cpp/src/arrow/compute/ordering.h). */
enum { ARX_SORT_ASCENDING = 0, ARX_SORT_DESCENDING = 1 };
enum { ARX_NULLS_AT_START = 0, ARX_NULLS_AT_END = 1 };

/* Integer index types accepted by take (match::Integer(),
 * cpp/src/arrow/compute/kernels/vector_selection_take_internal.cc:718). */
enum {
  ARX_UINT8 = 0, ARX_INT8 = 1, ARX_UINT16 = 2, ARX_INT16 = 3,
  ARX_UINT32 = 4, ARX_INT32 = 5, ARX_UINT64 = 6, ARX_INT64 = 7
};

/* One fixed-width (or boolean) array: the device twin of arrow::ArraySpan
 * (cpp/src/arrow/array/data.h:553) restricted to {validity, values}.
 * For boolean arrays `data` is a bitmap and `offset` is a bit offset into it. */
typedef struct ArxSpan {
  const void* validity; /* bitmap or NULL */
  const void* data;     /* values buffer (NOT pre-offset) */
  int64_t offset;       /* logical element offset into both buffers */
  int64_t length;       /* logical length */
  int64_t null_count;   /* exact, 0, or -1 = unknown (kUnknownNullCount) */
} ArxSpan;

const char* arx_last_error(void);
int arx_abi_version(void);
/* Number of HIP devices visible, or a negative ArxStatus. */
int arx_device_count(void);
/* Process-wide tuning knobs for A/B measurements ("filter_sparse", "groupby_partition_bits",
 * "sort_msd", ...; the list is in DESIGN.md 4.8).  Never changes results.  May be called while
 * other threads run kernels: every knob is a relaxed atomic, a call in flight uses the old or the
 * new value of each knob it reads (which of the two is unspecified).  Not part of the reference
 * interface. */
int arx_set_option(const char* name, int64_t value);
/* Process-wide diagnostic counters (monotonic): which plan the slices of the partitioned group-by consume ran —
 * "groupby_slices_direct" / "_one_level" / "_two_level" / "_wide" / "_probe" (DESIGN.md 4.6); which record form the wide
 * sorts ran with — "sort_wide_runs", "sort_wide_rec8_runs", "sort_wide_rec8_ties" (rows that read their full key),
 * "sort_wide_rec8_given_up", "sort_wide_wc_runs" (write-combined level 1).  -1 + arx_last_error() for an unknown name.  Not part of the reference
 * interface. */
int64_t arx_get_counter(const char* name);

/* ---------------------------------------------------------------------------
 * Filter  — replaces PrimitiveFilterExec / PrimitiveFilterImpl<W>::Exec
 * (cpp/src/arrow/compute/kernels/vector_selection_filter_internal.cc:445-510, 238-372)
 * and GetFilterOutputSize (same file :62-114).
 *
 * Two-phase, exactly like the reference: (1) count the output rows so the caller
 * can allocate, (2) compact.  Phase 1 leaves per-tile output offsets in `ws`,
 * which phase 2 (and arx_mask_to_indices) consume.
 * ------------------------------------------------------------------------- */

/* Bytes of scratch needed for a mask of `length` rows. */
size_t arx_filter_workspace_bytes(int64_t length);

/* Phase 1 (synchronous): *out_length (host) = number of emitted rows of `mask`
 * (boolean ArxSpan) under `null_selection`.  Fills `ws`. */
int arx_filter_count(const ArxSpan* mask, int null_selection, void* ws, size_t ws_bytes,
                     int64_t* out_length, void* stream);

/* Phase 1 + the output's null count in the same pass and the same read-back: *out_null_count = emitted rows whose
 * output slot is null (a null mask slot under EMIT_NULL, or a selected row whose value is null).  The reference leaves
 * the output's null_count unknown and counts the bitmap on first use (:462-467); a device-resident output cannot be
 * counted on the host, and a second kernel + read-back after phase 2 costs ~70 us of a 1.5 ms filter. */
int arx_filter_count_nulls(const ArxSpan* values, const ArxSpan* mask, int null_selection, void* ws, size_t ws_bytes,
                           int64_t* out_length, int64_t* out_null_count, void* stream);

/* Phase 1, asynchronous flavour: as above but the total stays on the device
 * (first 8 bytes of ws); use when the output capacity is already known. */
int arx_filter_count_async(const ArxSpan* mask, int null_selection, void* ws,
                           size_t ws_bytes, void* stream);

/* Phase 2 (asynchronous): out_data[0..S) = values at emitted rows (byte_width in
 * {1,2,4,8,16}); out_validity (may be NULL when neither input can be null:
 * the `allocate_validity` rule at :472) receives S bits.  A row emitted because
 * the mask slot is null (EMIT_NULL) is zero-filled and marked null (:398-407);
 * a selected row whose value is null keeps its source bytes (:267-272).
 * `ws` must come from arx_filter_count* on the same mask/null_selection;
 * `out_length` is the S that arx_filter_count returned (only needed when
 * out_validity != NULL: its ceil(S/64) words are cleared first). */
int arx_filter_exec(const ArxSpan* values, int byte_width, const ArxSpan* mask,
                    int null_selection, const void* ws, int64_t out_length, void* out_data,
                    void* out_validity, void* stream);

/* GetTakeIndices — replaces GetTakeIndicesFromBitmapImpl<UInt16/UInt32>
 * (cpp/src/arrow/compute/kernels/vector_selection_take_internal.cc:62-168,258-305).
 * index_width is 2 (length <= 65535) or 4; EMIT_NULL writes 0 + null for null
 * mask slots (out_validity required then, otherwise may be NULL).  index_width 8 (DROP only): the uint64 row numbers
 * indices_nonzero returns (DoNonZero, compute/kernels/vector_selection.cc:228-300) in one pass; ARX_NOT_IMPLEMENTED
 * where only the 2 / 4-byte forms apply (the caller widens then). */
int arx_mask_to_indices(const ArxSpan* mask, int null_selection, const void* ws,
                        int64_t out_length, int index_width, void* out_indices,
                        void* out_validity, void* stream);

/* ---------------------------------------------------------------------------
 * Take — replaces FixedWidthTakeExec / Gather<W,Idx,false>
 * (vector_selection_take_internal.cc:339-468, gather_internal.h:47-251) and
 * CheckIndexBounds (cpp/src/arrow/util/int_util.cc:530-587).
 * ------------------------------------------------------------------------- */
size_t arx_take_workspace_bytes(void);

/* Synchronous.  ARX_INDEX_ERROR + "Index N out of bounds" (N = the first
 * offending valid index, int_util.cc:554) if any non-null index is <0 or
 * >= upper_limit. */
int arx_check_index_bounds(const ArxSpan* indices, int index_type, uint64_t upper_limit,
                           void* ws, size_t ws_bytes, void* stream);

/* Asynchronous gather.  out_data[i] = values[indices[i]]; every null output slot
 * (null index or null source value) is zero-filled (WriteZero,
 * gather_internal.h:114-153).  out_validity may be NULL iff neither input has a
 * validity buffer.  If valid_count (device int64*, may be NULL) is given it is
 * incremented by the number of valid output rows (caller zeroes it). */
int arx_take(const ArxSpan* values, int byte_width, const ArxSpan* indices, int index_type,
             void* out_data, void* out_validity, int64_t* valid_count, void* stream);

/* Take of rows of `row_bytes` contiguous bytes (any width): what FSLTakeExec hands to FixedWidthTakeExec for a
 * fixed_size_list whose nested values are fixed-width and free of nulls (vector_selection_internal.cc:991-1003,
 * util::IsFixedWidthLike), and fixed_size_binary of a width the element kernels of arx_take do not have.  values->data =
 * row 0 of the nested values, values->offset / ->validity = the list array's; otherwise as arx_take.  Asynchronous. */
int arx_take_rows(const ArxSpan* values, int64_t row_bytes, const ArxSpan* indices, int index_type, void* out_data,
                  void* out_validity, int64_t* valid_count, void* stream);

/* The same gather for SEVERAL fixed-width columns by one index array in one launch — what TakeRAR / TakeTAT
 * (vector_selection_take_internal.cc:619-660: `take` of a RecordBatch / Table) get by running TakeAAA column after
 * column; here the indices and their validity are read once per row and every column is gathered behind them.
 * columns / byte_widths / out_data / out_validity: arrays of num_columns (1..16) entries (host memory; the buffers they
 * point to are device memory); out_validity[c] may be NULL iff neither column c nor the indices have a validity buffer;
 * valid_counts: device int64[num_columns] (caller-zeroed) or NULL.  Per column the result is arx_take's.  Bounds are
 * the caller's (arx_check_index_bounds against the shortest column, once).  Asynchronous. */
int arx_take_columns(const ArxSpan* columns, const int32_t* byte_widths, int num_columns, const ArxSpan* indices,
                     int index_type, void* const* out_data, void* const* out_validity, int64_t* valid_counts,
                     void* stream);
/* The same for BOOLEAN values (bit-packed data buffer, values->offset in bits): out bit i = value bit
 * idx[i], 0 for null slots (Gather with 1-bit values, gather_internal.h).  out_bits / out_validity:
 * ceil(M/64) 64-bit words, zero padded.  A filter on boolean values is arx_mask_to_indices + this. */
int arx_take_bits(const ArxSpan* values, const ArxSpan* indices, int index_type, void* out_bits, void* out_validity,
                  int64_t* valid_count, void* stream);

/* ---------------------------------------------------------------------------
 * Take / Filter on base-binary values (binary, utf8: int32 offsets) — replaces TakeExec for base
 * binary (cpp/src/arrow/compute/kernels/vector_selection_take_internal.cc) and, through
 * GetTakeIndices, BinaryFilterImpl (vector_selection_filter_internal.cc:517-800): out slot i is
 * valid iff the index and the source value are; a valid slot appends the source bytes, a null slot
 * nothing; out offsets start at 0.  Two steps because the data size is data dependent (the
 * reference grows a builder):
 *   arx_binary_take_offsets (synchronous): out_offsets[0..M] (device int32), out_validity,
 *     valid_count (device, caller-zeroed, may be NULL), *out_total_bytes (host); ARX_INVALID
 *     "offset overflow" if the bytes do not fit int32 offsets.
 *   arx_binary_take_data (asynchronous): the bytes, into out_data[0 .. total).  `ws` is the
 *     workspace arx_binary_take_offsets filled (it holds the source position of every slot) and
 *     must not be reused in between.
 * A filter is arx_mask_to_indices followed by these two.
 * ------------------------------------------------------------------------- */
typedef struct ArxBinarySpan {
  const void* validity;    /* bitmap or NULL */
  const int32_t* offsets;  /* offsets buffer (NOT pre-offset): length + offset + 1 entries */
  const void* data;        /* value bytes */
  int64_t offset;
  int64_t length;
  int64_t null_count;
} ArxBinarySpan;
size_t arx_binary_take_workspace_bytes(int64_t num_indices);
int arx_binary_take_offsets(const ArxBinarySpan* values, const ArxSpan* indices, int index_type, void* ws,
                            size_t ws_bytes, int32_t* out_offsets, void* out_validity, int64_t* valid_count,
                            int64_t* out_total_bytes, void* stream);
int arx_binary_take_data(const ArxBinarySpan* values, int64_t num_indices, const void* ws, size_t ws_bytes,
                         const int32_t* out_offsets, int64_t total_bytes, void* out_data, void* stream);
/* The same for large_utf8 / large_binary (int64 offsets, vector_selection_take_internal.cc / _filter_internal.cc register
 * the large types with the same VarBinary implementations): values->offsets points at int64_t entries, the output offsets
 * are int64_t, and there is no 2 GB limit. */
size_t arx_large_binary_take_workspace_bytes(int64_t num_indices);
int arx_large_binary_take_offsets(const ArxBinarySpan* values, const ArxSpan* indices, int index_type, void* ws,
                                  size_t ws_bytes, int64_t* out_offsets, void* out_validity, int64_t* valid_count,
                                  int64_t* out_total_bytes, void* stream);
int arx_large_binary_take_data(const ArxBinarySpan* values, int64_t num_indices, const void* ws, size_t ws_bytes,
                               const int64_t* out_offsets, int64_t total_bytes, void* out_data, void* stream);

/* list<T> / large_list<T> whose nested values are fixed-width (2^elem_shift bytes, elem_shift 0 .. 5) and free of nulls —
 * ListSelectionImpl (vector_selection_internal.cc:620-760; ListTakeExec :975, ListFilterExec :910): such a list is a binary
 * array whose offsets count elements.  values->offsets = the list offsets, values->data = element 0 of the nested
 * values (the child's own offset applied); arx_(large_)binary_take_offsets computes the output offsets unchanged —
 * total "bytes" is then the number of ELEMENTS —, these copy them.  Asynchronous. */
int arx_list_take_data(const ArxBinarySpan* values, int elem_shift, int64_t num_indices, const void* ws, size_t ws_bytes,
                       const int32_t* out_offsets, int64_t total_elements, void* out_data, void* stream);
int arx_large_list_take_data(const ArxBinarySpan* values, int elem_shift, int64_t num_indices, const void* ws,
                             size_t ws_bytes, const int64_t* out_offsets, int64_t total_elements, void* out_data,
                             void* stream);

/* ---------------------------------------------------------------------------
 * Cast float64 -> float32 — replaces CastPrimitive<FloatType,DoubleType>::Exec
 * (cpp/src/arrow/compute/kernels/scalar_cast_internal.cc:41-53): every slot is
 * converted (null slots included), IEEE round-to-nearest-even.
 * `in` is pre-offset (points at element 0).
 * ------------------------------------------------------------------------- */
int arx_cast_f64_f32(const double* in, int64_t length, float* out, void* stream);

/* Integer casts — CastIntegerToInteger (cpp/src/arrow/compute/kernels/scalar_cast_numeric.cc:46-54).
 * int64 -> int32: unless allow_int_overflow, the first valid slot (row order) whose value does not fit
 * fails with the reference's text "Integer value V not in range: -2147483648 to 2147483647"
 * (IntegersInRange, util/int_util.cc:594-665; ARX_INVALID, synchronous; ws: >= 8 device bytes);
 * every slot is converted with static_cast either way.  int32 -> int64 widens (asynchronous). */
int arx_cast_i64_i32(const ArxSpan* values, int allow_int_overflow, void* ws, size_t ws_bytes, int32_t* out,
                     void* stream);
int arx_cast_i32_i64(const int32_t* values, int64_t length, int64_t* out, void* stream);
/* int64 -> float64 — CastIntegerToFloating (scalar_cast_numeric.cc:270-279): unless
 * allow_float_truncate, CheckIntegerFloatTruncateImpl (:218-227) applies the same range check with
 * the bounds -2^53 .. 2^53 ("Integer value V not in range: -9007199254740992 to 9007199254740992"). */
int arx_cast_i64_f64(const ArxSpan* values, int allow_float_truncate, void* ws, size_t ws_bytes, double* out,
                     void* stream);

/* Every numeric pair — CastIntegerToInteger / CastFloatingToInteger / CastIntegerToFloating / CastFloatingToFloating
 * (scalar_cast_numeric.cc:46-60, 190-207, 270-279; registration :797-858): all slots go through static_cast
 * (CastNumberToNumberUnsafe), and on the VALID slots
 *   integer -> integer, unless allow_int_overflow: IntegersCanFit (util/int_util.cc:795-900) —
 *       "Integer value V not in range: LO to HI" with the bounds GetSafeMinMax derives for the pair;
 *   integer -> floating, unless allow_float_truncate: |V| <= 2^24 (float) / 2^53 (double) for the 32/64-bit inputs
 *       the reference checks (:229-268), same message;
 *   floating -> integer, unless allow_float_truncate: static_cast<In>(out) == in, else
 *       "Float value V was truncated converting to TYPE" (out-of-range values and NaN fail it too).  With
 *       allow_float_truncate an out-of-range value converts to an unspecified integer (undefined behaviour in the
 *       reference as well).
 * The first offender in row order names the error (ARX_INVALID; synchronous when a check applies, ws >= 8 device
 * bytes).  in_type / out_type: ARX_NUM_*. */
enum {
  ARX_NUM_INT8 = 0, ARX_NUM_UINT8 = 1, ARX_NUM_INT16 = 2, ARX_NUM_UINT16 = 3, ARX_NUM_INT32 = 4, ARX_NUM_UINT32 = 5,
  ARX_NUM_INT64 = 6, ARX_NUM_UINT64 = 7, ARX_NUM_FLOAT32 = 8, ARX_NUM_FLOAT64 = 9
};
int arx_cast_numeric(const ArxSpan* values, int in_type, int out_type, int allow_int_overflow, int allow_float_truncate,
                     void* ws, size_t ws_bytes, void* out, void* stream);

/* ---------------------------------------------------------------------------
 * Compare — replaces ComparePrimitiveArrayArray/ArrayScalar/ScalarArray<DoubleType,
 * Greater> (cpp/src/arrow/compute/kernels/scalar_compare.cc:165-247): bit i =
 * left[i] > right[i] (any NaN -> 0), LSB-first, computed on all slots.
 * Pointers are pre-offset.  out_bits: ceil(length/64) 64-bit words.
 * ------------------------------------------------------------------------- */
int arx_greater_f64(const double* left, const double* right, int64_t length,
                    uint64_t* out_bits, void* stream);
int arx_greater_f64_array_scalar(const double* left, double right, int64_t length,
                                 uint64_t* out_bits, void* stream);
int arx_greater_f64_scalar_array(double left, const double* right, int64_t length,
                                 uint64_t* out_bits, void* stream);
int arx_greater_i64(const int64_t* left, const int64_t* right, int64_t length,
                    uint64_t* out_bits, void* stream);
int arx_greater_i64_array_scalar(const int64_t* left, int64_t right, int64_t length,
                                 uint64_t* out_bits, void* stream);
int arx_greater_i64_scalar_array(int64_t left, const int64_t* right, int64_t length,
                                 uint64_t* out_bits, void* stream);

/* The whole comparison family — equal, not_equal, greater, greater_equal, less, less_equal
 * (Equal ... LessEqual, cpp/src/arrow/compute/kernels/scalar_compare.cc:38-64; less / less_equal
 * are registered there as the flipped greater / greater_equal, :436-445, and run here the same
 * way).  left / right: pre-offset arrays, or NULL for "this side is the scalar" (then *_scalar is
 * its value).  Floats: IEEE — NaN makes every ordered comparison and equal false, not_equal true. */
#define ARX_CMP_EQUAL 0
#define ARX_CMP_NOT_EQUAL 1
#define ARX_CMP_GREATER 2
#define ARX_CMP_GREATER_EQUAL 3
#define ARX_CMP_LESS 4
#define ARX_CMP_LESS_EQUAL 5
int arx_compare_f64(int op, const double* left, double left_scalar, const double* right, double right_scalar,
                    int64_t length, uint64_t* out_bits, void* stream);
int arx_compare_i64(int op, const int64_t* left, int64_t left_scalar, const int64_t* right, int64_t right_scalar,
                    int64_t length, uint64_t* out_bits, void* stream);

/* ---------------------------------------------------------------------------
 * Arithmetic — replaces ScalarBinary<Int64,Int64,Int64,Add> / <Double,...>
 * (cpp/src/arrow/compute/kernels/base_arithmetic_internal.h:45-80,
 * codegen_internal.h:814): unchecked integer add wraps around.
 * ------------------------------------------------------------------------- */
int arx_add_i64(const int64_t* left, const int64_t* right, int64_t length, int64_t* out,
                void* stream);
int arx_add_f64(const double* left, const double* right, int64_t length, double* out,
                void* stream);
/* add / subtract / multiply in one entry point (Add, Subtract, Multiply,
 * base_arithmetic_internal.h:45-120,290-330): left / right are pre-offset arrays or NULL for
 * "this side is the scalar".  Integer results wrap. */
#define ARX_ARITH_ADD 0
#define ARX_ARITH_SUBTRACT 1
#define ARX_ARITH_MULTIPLY 2
int arx_arith_i64(int op, const int64_t* left, int64_t left_scalar, const int64_t* right, int64_t right_scalar,
                  int64_t length, int64_t* out, void* stream);
int arx_arith_f64(int op, const double* left, double left_scalar, const double* right, double right_scalar,
                  int64_t length, double* out, void* stream);
/* add_checked / subtract_checked / multiply_checked(int64) (AddChecked ..., :70-120,341-364): the
 * wrapped results are written like above and *overflow_flag (device uint32, caller-zeroed) is set
 * if a slot where BOTH operands are valid overflowed — the reference visits only those slots
 * (ScalarBinaryNotNull) and then fails with Status::Invalid("overflow"); the caller reads the flag
 * after the stream has drained.  left/right_validity: bitmaps (bit offsets) or NULL = all valid.
 * For doubles the checked functions are the plain ones. */
int arx_arith_checked_i64(int op, const int64_t* left, int64_t left_scalar, const void* left_validity,
                          int64_t left_offset, const int64_t* right, int64_t right_scalar,
                          const void* right_validity, int64_t right_offset, int64_t length, int64_t* out,
                          unsigned int* overflow_flag, void* stream);

/* The comparison family and add / subtract / multiply (+ _checked) for EVERY numeric element type (num_type:
 * ARX_NUM_*; both operands and, for arithmetic, the result have that type): the same Call bodies as above
 * instantiated per type, as the reference registers them for all of NumericTypes() (scalar_compare.cc:398-446,
 * scalar_arithmetic.cc AddArithmeticFunctions).  Unchecked integer results wrap in the type's width (the reference
 * computes them in the unsigned type, base_arithmetic_internal.h:45-68; int16 / uint16 multiply through uint32,
 * :303-325 — the same bits); the checked forms report an overflow of that type.  float / double: IEEE.
 * left / right == NULL: that operand is the scalar *left_scalar / *right_scalar (host pointer to one value of the type).
 * out_bits / out / overflow_flag / validity as for the 64-bit entry points above.  Asynchronous. */
int arx_compare_numeric(int op, int num_type, const void* left, const void* left_scalar, const void* right,
                        const void* right_scalar, int64_t length, uint64_t* out_bits, void* stream);
int arx_arith_numeric(int op, int checked, int num_type, const void* left, const void* left_scalar,
                      const void* left_validity, int64_t left_offset, const void* right, const void* right_scalar,
                      const void* right_validity, int64_t right_offset, int64_t length, void* out,
                      uint32_t* overflow_flag, void* stream);
/* divide / divide_checked (Divide, DivideChecked, base_arithmetic_internal.h:366-424), visited only where both
 * operands are valid.  int64: truncating division; a zero divisor fails with "divide by zero" in both forms;
 * INT64_MIN / -1 is 0 unchecked and "overflow" checked.  double: IEEE division, the checked form fails on a zero
 * divisor.  errors: two device uint64 zeroed by the caller; after the stream has drained errors[0] / errors[1] hold
 * 1 + the LAST failing row of the overflow / zero-divisor kind (the reference overwrites its Status on every
 * failing slot, so the larger of the two names the message).  A NULL operand pointer broadcasts its scalar. */
int arx_divide_i64(const int64_t* left, int64_t left_scalar, const void* left_validity, int64_t left_offset,
                   const int64_t* right, int64_t right_scalar, const void* right_validity,
                   int64_t right_offset, int64_t length, int checked, int64_t* out, uint64_t* errors,
                   void* stream);
int arx_divide_f64(const double* left, double left_scalar, const void* left_validity, int64_t left_offset,
                   const double* right, double right_scalar, const void* right_validity,
                   int64_t right_offset, int64_t length, int checked, double* out, uint64_t* errors,
                   void* stream);
/* The same for EVERY numeric element type (num_type: ARX_NUM_*; the reference registers divide / divide_checked for all
 * of NumericTypes(), scalar_arithmetic.cc): the signed integer types fail / give 0 on min / -1 of their own width, the
 * unsigned ones only on a zero divisor (DivideWithOverflowGeneric, util/int_util_overflow.h:124-138), float like double.
 * left / right == NULL: that operand is the scalar *left_scalar / *right_scalar (host pointer to one value of the type). */
int arx_divide_numeric(int checked, int num_type, const void* left, const void* left_scalar, const void* left_validity,
                       int64_t left_offset, const void* right, const void* right_scalar, const void* right_validity,
                       int64_t right_offset, int64_t length, void* out, uint64_t* errors, void* stream);
/* array + valid scalar (ScalarBinary::ArrayScalar, codegen_internal.h; add commutes, so
 * scalar + array is the same call) */
int arx_add_i64_array_scalar(const int64_t* left, int64_t right, int64_t length, int64_t* out,
                             void* stream);
int arx_add_f64_array_scalar(const double* left, double right, int64_t length, double* out,
                             void* stream);

/* ---------------------------------------------------------------------------
 * Validity plumbing — what ScalarExecutor's null propagation does on the host
 * (PropagateNullsSpans, cpp/src/arrow/compute/exec.cc:1222-1281; BitmapAnd /
 * CopyBitmap, cpp/src/arrow/util/bitmap_ops.cc).  Inputs carry bit offsets,
 * outputs start at bit 0 and are zero-padded to a 64-bit boundary.
 * ------------------------------------------------------------------------- */
/* Device-to-device copy of nbytes (ranges must not overlap) — the kROCM -> kROCM leg of MemoryManager::CopyBufferTo /
 * CopyNonOwned (cpp/src/arrow/device.h:214-222, gpu/cuda_memory.cc's CopyBufferTo) as a kernel: 16 bytes per lane,
 * one-shot grid.  bench.py times it on 1 GiB as the box's copy ceiling (`copy_ceiling_GBps`).  Asynchronous. */
int arx_buffer_copy(const void* src, void* dst, int64_t nbytes, void* stream);
int arx_bitmap_copy(const void* bits, int64_t bit_offset, int64_t length, void* out,
                    void* stream);
int arx_bitmap_and(const void* left, int64_t left_offset, const void* right,
                   int64_t right_offset, int64_t length, void* out, void* stream);
/* Concatenation plumbing (arrow::Concatenate, cpp/src/arrow/array/concatenate.cc: ConcatenateBitmaps
 * and PutOffsets) for operators that must see all their input at once (OrderByNode accumulates every
 * batch, cpp/src/arrow/acero/order_by_node.cc:100-122).
 * arx_bitmap_copy_at ORs bits [bit_offset, bit_offset+length) of `bits` (NULL = all ones) into `out`
 * starting at bit `out_bit_offset`; the words of `out` from that bit on must be zero (memset once, then
 * append chunk after chunk on one stream).  arx_binary_rebase_offsets writes
 * out[i] = offsets[i] - offsets[0] + base for i in [0, length].  Asynchronous. */
int arx_bitmap_copy_at(const void* bits, int64_t bit_offset, int64_t length, void* out,
                       int64_t out_bit_offset, void* stream);

/* Many device-to-device copies in ONE launch: the data half of Concatenate (array/concatenate.cc ConcatenateBuffers)
 * when the chunks are small and many — Acero hands an operator 32K-row batches (exec_plan.h kMaxBatchSize), and a
 * hipMemcpy or a launch per batch and column is what bounds such plans.  segments: DEVICE array of {src, dst, nbytes}
 * with absolute device addresses (so one table can fill several destination columns); max_segment_bytes sizes the
 * grid (workgroups per segment).  Overlapping destinations are the caller's problem.  Asynchronous. */
typedef struct ArxCopySeg {
  const void* src;
  void* dst;
  uint64_t nbytes;
} ArxCopySeg;
int arx_copy_segments(const ArxCopySeg* segments, int64_t num_segments, uint64_t max_segment_bytes, void* stream);
/* The validity half of the same concatenation (ConcatenateBitmaps, array/concatenate.cc): bit ranges, each at its own
 * bit offset, ORed into ZEROED destination bitmaps at their own bit positions (ranges may meet inside a word).
 * src == NULL stands for a chunk without a validity buffer: all ones.  dst: 8-byte aligned, writable up to the 64-bit
 * word that holds the range's last bit.  segments: DEVICE array.  Asynchronous. */
typedef struct ArxBitSeg {
  const void* src;
  int64_t src_bit_offset;
  void* dst;
  int64_t dst_bit_offset;
  int64_t nbits;
} ArxBitSeg;
int arx_bitmap_copy_segments(const ArxBitSeg* segments, int64_t num_segments, int64_t max_segment_bits, void* stream);
int arx_binary_rebase_offsets(const int32_t* offsets, int64_t length, int32_t base, int32_t* out,
                              void* stream);
/* Scalar aggregates over an int64 column in one pass — the state SumImpl / CountImpl / MinMaxImpl keep
 * (cpp/src/arrow/compute/kernels/aggregate_basic.inc.cc:49-110,776-860): acc = {wrap-around sum of the
 * valid values, their count, min, max} as 4 x int64 in device memory; init once, consume per batch
 * (Consume / MergeFrom accumulate the same way).  The options (skip_nulls, min_count) are applied by
 * the caller from count and the inputs' null counts, as Finalize does.  Asynchronous. */
int arx_reduce_i64_init(void* acc, void* stream);
int arx_reduce_i64_consume(const ArxSpan* values, void* acc, void* stream);
/* coalesce(values, fill) of ONE fixed-width type — what fill_null(values, fill_value) calls (CoalesceFunctor,
 * compute/kernels/scalar_if_else.cc): out[i] = values[i] where valid, else fill[i] (fill != NULL: an array of the same
 * length) or *fill_scalar (byte_width bytes; booleans: one byte 0 / 1); fill == NULL && fill_scalar == NULL: a null
 * scalar (the values' own validity survives).  byte_width 1 / 2 / 4 / 8, or 0 for booleans (bitmaps).  out_validity
 * (ceil(length / 64) words, always written) = valid(values) | valid(fill); null result slots are zeroed.  Asynchronous. */
int arx_coalesce2(int byte_width, const ArxSpan* values, const ArxSpan* fill, const void* fill_scalar, int64_t length,
                  void* out_data, void* out_validity, void* stream);
/* min_max of a float32 / float64 column (num_type ARX_NUM_FLOAT32 / _FLOAT64) — MinMaxState<floating>
 * (aggregate_basic.inc.cc:681-701: fmin / fmax over NaN anti-extrema): acc (arx_reduce_i64_init) gets [1] += the valid
 * values, [2] / [3] = min / max of the ORDER KEYS of the valid non-NaN values (the int64 whose signed order is the
 * doubles' numeric order, -0.0 just below +0.0); the untouched INT64_MAX / INT64_MIN read back as NaN.  Asynchronous. */
int arx_reduce_float_minmax(const ArxSpan* values, int num_type, void* acc, void* stream);
/* sum of a float32 / float64 column, BIT FOR BIT the reference's: SumArray's pairwise summation
 * (compute/kernels/aggregate_internal.h:155-232 — the valid values of every run in blocks of 16, left to right; block sums
 * merged by a binary counter; float32 values are widened to double first, as SumImpl's accumulator type asks) is a fixed tree
 * of additions, evaluated here block by block and level by level on the device and finished on the host over the few
 * thousand partial sums that are left.  *out_count = the valid values (MeanImpl divides by it).  One call = one
 * Consume of one batch (the caller adds batch sums in batch order, as SumImpl does).  ws: device scratch of
 * arx_sum_float_workspace_bytes(length, null_count) bytes (null_count < 0: unknown).  Synchronous. */
size_t arx_sum_float_workspace_bytes(int64_t length, int64_t null_count);
int arx_sum_float(const ArxSpan* values, int num_type, void* ws, size_t ws_bytes, double* out_sum, int64_t* out_count,
                  void* stream);

/* Kleene logic on boolean arrays — KleeneAndOp / KleeneOrOp (array, array) and InvertOp,
 * cpp/src/arrow/compute/kernels/scalar_boolean.cc:138-260.  left/right: boolean ArxSpans (data =
 * LSB-first bitmap, offset in bits, validity NULL or null_count == 0 = no nulls).  out_data /
 * out_validity start at bit 0, zero padded; out_validity may be NULL only when neither input can
 * have nulls.  and: data = l_true & r_true, valid = l_false | r_false | (l_true & r_true);
 * or: data = l_true | r_true, valid = l_true | r_true | (l_false & r_false).  Asynchronous. */
#define ARX_AND_KLEENE 0
#define ARX_OR_KLEENE 1
int arx_boolean_kleene(int op, const ArxSpan* left, const ArxSpan* right, void* out_data, void* out_validity,
                       void* stream);
/* out bit i = !bits[bit_offset + i] (validity is the caller's: arx_bitmap_copy). */
int arx_boolean_invert(const void* bits, int64_t bit_offset, int64_t length, void* out, void* stream);
/* BytesToBits (util/bitmap_builders.cc) on the device: out bit i = (bytes[i] != 0), whole 64-bit words written, zero
 * padded; *set_count (device, may be NULL) is INCREMENTED by the number of set bits.  Asynchronous. */
int arx_bytes_to_bitmap(const uint8_t* bytes, int64_t length, void* out_bits, int64_t* set_count, void* stream);
/* Synchronous popcount of [bit_offset, bit_offset+length) (CountSetBits). */
int arx_bitmap_popcount(const void* bits, int64_t bit_offset, int64_t length, void* ws,
                        size_t ws_bytes, int64_t* out_count, void* stream);

/* run_end_encoded<boolean> filter masks (vector_selection_filter_internal.cc:1090,1115-; the run visitor
 * VisitPlainxREEFilterOutputSegments, vector_selection_internal.cc:79-153): expands the runs covering the logical rows
 * [logical_offset, logical_offset + length) into the plain mask layout — out_bits (selection) and out_validity (NULL
 * allowed when the run values have no nulls), bit i = row i, whole 64-bit words written — which arx_filter_* then takes
 * as an ordinary boolean mask: every row of a run carries its run's (valid, selected) pair, exactly what the reference
 * visits.  run_ends: int16 / int32 / int64 (run_end_width 2 / 4 / 8), values: the boolean run values.  Asynchronous. */
int arx_ree_bool_expand(const void* run_ends, int run_end_width, int64_t num_runs, const ArxSpan* values,
                        int64_t logical_offset, int64_t length, void* out_bits, void* out_validity, void* stream);

/* ---------------------------------------------------------------------------
 * array_sort_indices (uint64/int64 keys) — replaces ArraySortIndices<UInt64Type,
 * UInt64Type>::Exec -> ArrayCountOrCompareSorter / ArrayCompareSorter
 * (cpp/src/arrow/compute/kernels/vector_array_sort.cc:144-178,404-446,524-540):
 * a STABLE argsort producing uint64 indices; nulls are stably partitioned to the
 * end or the start (PartitionNullsOnly, vector_sort_internal.h:225-293);
 * descending keeps ties in ascending index order (rhs < lhs comparator).
 * ------------------------------------------------------------------------- */
/* Key types of arx_sort_indices (the physical types AddArraySortingKernels registers,
 * vector_array_sort.cc:554-610, that are on the gfx950 path).  For floating point keys NaNs are
 * "null-likes" (PartitionNulls, vector_sort_internal.h): values, NaNs, nulls (at_end) or nulls,
 * NaNs, values (at_start) whatever the order; -0.0 and 0.0 tie. */
enum {
  ARX_KEY_UINT64 = 0, ARX_KEY_INT64 = 1, ARX_KEY_UINT32 = 2, ARX_KEY_INT32 = 3,
  ARX_KEY_FLOAT64 = 4, ARX_KEY_FLOAT32 = 5
};
size_t arx_sort_indices_workspace_bytes(int64_t length);
int arx_sort_indices(const ArxSpan* values, int key_type, int order, int null_placement, void* ws,
                     size_t ws_bytes, uint64_t* out_indices, void* stream);
/* rank / rank_quantile of a column from its sorted order — the second half of RankMetaFunction
 * (compute/kernels/vector_rank.cc:36-245: MarkDuplicates :40-72, OrdinalRanker::CreateRankings :203-263,
 * BaseQuantileRanker::CreateRankings :163-196).  sorted_rows = arx_sort_indices(values, key_type, order, null_placement):
 * the order and the null placement are in it; all NaNs tie, all nulls tie, -0.0 ties with 0.0.  out: uint64[length]
 * 1-based ranks (ARX_RANK_MIN / MAX / FIRST / DENSE = RankOptions::Tiebreaker, api_vector.h:201-212) or, ARX_RANK_QUANTILE,
 * double[length] = (rows below the row's run of ties + half the run) / length, or, ARX_RANK_NORMAL ("rank_normal",
 * NormalRanker :204-209), the normal percent-point function of that quantile (arrow::internal::NormalPPF,
 * util/math_internal.cc:26-137 = Wichura's AS 241, evaluated without fused multiply-add: bit for bit in the centre of the
 * distribution, within the log() of the two math libraries in the tails — the reference's own test bar is 4 ULPs).
 * ws: arx_rank_workspace_bytes(length), 256-byte aligned (not used by ARX_RANK_FIRST).  Asynchronous. */
enum { ARX_RANK_MIN = 0, ARX_RANK_MAX = 1, ARX_RANK_FIRST = 2, ARX_RANK_DENSE = 3, ARX_RANK_QUANTILE = 4, ARX_RANK_NORMAL = 5 };
size_t arx_rank_workspace_bytes(int64_t length);
int arx_rank(const ArxSpan* values, int key_type, const uint64_t* sorted_rows, int tiebreaker, void* ws, size_t ws_bytes,
             void* out, void* stream);
/* 64-bit integer keys (is_signed: 0 = uint64, 1 = int64): same as arx_sort_indices. */
int arx_sort_indices_64(const ArxSpan* values, int is_signed, int order, int null_placement,
                        void* ws, size_t ws_bytes, uint64_t* out_indices, void* stream);

/* ---------------------------------------------------------------------------
 * Multi-GPU sort_indices: every rank owns a contiguous row shard; the sort
 * needs ONE exchange step.  These are the device pieces around it:
 *   arx_sort_key_histogram      counts of the top `bits` (<= 12) bits of the order-transformed key
 *                               over the non-null rows, added into out_hist (device u64[2^bits],
 *                               caller-zeroed); an all-reduce of it yields the splitters;
 *   arx_sort_partition_by_bins  rows -> destination rank d = #{j : splitter_bins[j] <= bin(row)},
 *                               a STABLE partition (row order kept inside a destination, which is
 *                               what keeps the global sort stable): out_keys = transformed keys,
 *                               out_rows = local row ids, both destination-major; out_counts
 *                               (device int64[num_parts]); *out_num_valid (host) = non-null rows.
 *                               ws = arx_sort_indices_workspace_bytes(length), 256-byte aligned.
 *                               Synchronous when the shard has nulls or num_parts > 1.
 *   arx_bitmap_to_indices       ascending positions of the set (invert: clear) bits — the null
 *                               rows of a shard (PartitionNullsOnly keeps them in row order).
 *                               ws = arx_filter_workspace_bytes(length).  Synchronous.
 * The receiving rank sorts the transformed keys it was sent with arx_sort_indices_64
 * (unsigned, ascending: the transform already encodes sign and order).
 * ------------------------------------------------------------------------- */
int arx_sort_key_histogram(const ArxSpan* values, int is_signed, int order, int bits,
                           uint64_t* out_hist, void* stream);
int arx_sort_partition_by_bins(const ArxSpan* values, int is_signed, int order, int bits,
                               const uint32_t* splitter_bins /* host */, int num_parts, void* ws,
                               size_t ws_bytes, uint64_t* out_keys, uint32_t* out_rows,
                               int64_t* out_counts, int64_t* out_num_valid /* host */, void* stream);
int arx_bitmap_to_indices(const void* bits, int64_t bit_offset, int64_t length, int invert, void* ws,
                          size_t ws_bytes, uint32_t* out_indices, int64_t* out_count /* host */,
                          void* stream);
/* The same exchange with ONE buffer per peer.  arx_sort_partition_records = arx_sort_partition_by_bins with the
 * output packed as 12-byte records {order-transformed key, local row}, destination-major; the shard's null rows
 * (row order kept, keys 0) ride in the same buffer — after all valid records (ARX_NULLS_AT_END: they belong to the
 * last rank's block) or before them (ARX_NULLS_AT_START: the first rank's block); *out_num_valid (host) tells the
 * two parts apart.  arx_sort_unpack_records is the receiver: `records` = the blocks of all source ranks in rank
 * order, block_meta (device int64[num_blocks][3]) = {valid records, null records, global row number of the source's
 * row 0}; it writes the valid records' keys and GLOBAL rows compacted in source order (what keeps equal keys in
 * global row order) and the null rows' global numbers likewise.  Asynchronous. */
typedef struct ArxSortRecord {
  uint32_t key_lo, key_hi; /* order-transformed key: unsigned ascending order = the requested order */
  uint32_t row;            /* row number inside the sender's shard */
} ArxSortRecord;
int arx_sort_partition_records(const ArxSpan* values, int is_signed, int order, int null_placement, int bits,
                               const uint32_t* splitter_bins /* host */, int num_parts, void* ws, size_t ws_bytes,
                               ArxSortRecord* out_records /* values->length entries */,
                               int64_t* out_counts /* device int64[num_parts]: valid records per destination */,
                               int64_t* out_num_valid /* host */, void* stream);
int arx_sort_unpack_records(const ArxSortRecord* records, int64_t num_records, const int64_t* block_meta,
                            int num_blocks, int nulls_first, uint64_t* out_keys, int64_t* out_rows,
                            int64_t* out_null_rows, void* stream);
/* Splitter bins inside the WINDOW of the keys that exist.  Row ids, timestamps and small integers share their top
 * bits: bins of the raw key would put every row of every shard into one bin and the whole sort onto one rank.
 * arx_sort_key_range: out_range (device u64[2], caller-zeroed) = {max of ~key, max of key} over the shard's non-null
 * order-transformed keys — both combine by MAX, so ONE 16-byte all-reduce (as int64 with the sign bit flipped) gives
 * the global {~min, max}.  The window is then key_min = min, shift = leading zeros of (max - min), and
 * bin = top `bits` bits of (key - key_min) << shift (monotone in the key: splitters stay valid).  The _window forms
 * of the histogram and of the partition take it; window == NULL is the raw key (the plain forms above).
 * (No reference counterpart: Arrow has no multi-device sort; the result is still ArraySortIndices' permutation,
 * vector_array_sort.cc:524-540.) */
typedef struct ArxSortKeyWindow {
  uint64_t key_min; /* smallest order-transformed key of all shards */
  int32_t shift;    /* 0..63 */
  int32_t reserved;
} ArxSortKeyWindow;
int arx_sort_key_range(const ArxSpan* values, int is_signed, int order, uint64_t* out_range, void* stream);
/* The same two over a SAMPLE of the rows (round 6): one tile of 8192 rows in 2^sample_shift (0 .. 8; 0 = every row).  A window
 * built from a sampled range is widened by its caller, and keys outside a window fall into its first / last bin (the bin stays
 * monotone in the key: destinations stay ordered); a sampled histogram places the splitters, the exact per-destination
 * counts come from the partition. */
int arx_sort_key_range_sampled(const ArxSpan* values, int is_signed, int order, int sample_shift, uint64_t* out_range, void* stream);
int arx_sort_key_histogram_window(const ArxSpan* values, int is_signed, int order, int bits,
                                  const ArxSortKeyWindow* window, uint64_t* out_hist, void* stream);
int arx_sort_key_histogram_window_sampled(const ArxSpan* values, int is_signed, int order, int bits,
                                          const ArxSortKeyWindow* window, int sample_shift, uint64_t* out_hist, void* stream);
int arx_sort_partition_records_window(const ArxSpan* values, int is_signed, int order, int null_placement, int bits,
                                      const ArxSortKeyWindow* window, const uint32_t* splitter_bins /* host */,
                                      int num_parts, void* ws, size_t ws_bytes, ArxSortRecord* out_records,
                                      int64_t* out_counts, int64_t* out_num_valid /* host */, void* stream);
/* Round 6: the exchange WITHOUT a stable pass.  arx_sort_partition_records_global writes a null-free shard's records with
 * GLOBAL row numbers (row_base + row < 2^32), grouped by destination in no particular order inside a block (one tile-level
 * pass straight from the column: counts, then runs reserved with one atomic per tile and destination); the receiver sorts
 * what it got by (key, row) with arx_sort_records — the same order a stable sort of the keys gives (ArraySortIndices:
 * kernels/vector_array_sort.cc:524-540, ties in row order) without unpack, without a gather.  out_counts: device
 * int64[num_parts]; ws: 1 KB, 8-byte aligned.  Shards with nulls: ARX_NOT_IMPLEMENTED (the stable form above).  Asynchronous
 * but for the copy of the splitters.
 * arx_sort_records: out_rows[i] = row of the i-th smallest (key, row) record, widened; ws = arx_sort_indices_workspace_bytes(
 * num_records), 256-byte aligned. */
int arx_sort_partition_records_global(const ArxSpan* values, int is_signed, int order, int bits, const ArxSortKeyWindow* window,
                                      const uint32_t* splitter_bins /* host */, int num_parts, uint32_t row_base, void* ws,
                                      size_t ws_bytes, ArxSortRecord* out_records, int64_t* out_counts, void* stream);
int arx_sort_records(const ArxSortRecord* records, int64_t num_records, void* ws, size_t ws_bytes, uint64_t* out_rows, void* stream);

/* ---------------------------------------------------------------------------
 * Group-by hash_sum(int64) BY int32 key — replaces, as one fused device operator,
 * Grouper::Consume (cpp/src/arrow/compute/row/grouper.cc:662-815) +
 * HashAggregateKernel{resize,consume,merge,finalize} of
 * GroupedReducingAggregator<Int64Type,GroupedSumImpl> (compute/kernel.h:720-769,
 * kernels/hash_aggregate_numeric.cc:44-187,273-293) as driven by
 * GroupByNode::Consume/Merge/Finalize (acero/groupby_aggregate_node.cc:210-337).
 *
 * The state is an open-addressing table in HBM owned by the caller:
 * arx_groupby_state_bytes(capacity) bytes, capacity = a power of two > the
 * number of distinct keys (+1 for the null-key group).  Group order in the
 * output is unspecified, exactly as for the reference under threads
 * (python/pyarrow/table.pxi:5632-5634); parity is defined on the key-sorted result.
 * ------------------------------------------------------------------------- */
size_t arx_groupby_state_bytes(int64_t capacity);
/* Synchronous: clears the table (HashAggregateKernel::init + resize). */
int arx_groupby_init(void* state, int64_t capacity, void* stream);
/* consume: for every row, state[key].sum += value (wrap-around), count++,
 * a null value clears no_nulls; a null key is its own group.  Asynchronous.
 * `ws` (256-byte aligned, arx_groupby_consume_workspace_bytes; may be smaller, or NULL) is the
 * scratch of the radix-partitioned path: rows are partitioned by key hash until a partition's
 * groups fit an LDS table, aggregated there, and only per-partition partial aggregates touch the
 * HBM table.  With ws == NULL (or a batch below the partitioning threshold) every row goes
 * straight to the HBM table with device atomics — same results, ~12 Grows/s at best. */
size_t arx_groupby_consume_workspace_bytes(int64_t length, int64_t capacity);
int arx_groupby_sum_i64_consume(void* state, int64_t capacity, const ArxSpan* keys_i32,
                                const ArxSpan* values_i64, void* ws, size_t ws_bytes,
                                void* stream);
/* merge: fold partial aggregates (keys, key_is_valid, sums, counts, no_nulls; one byte
 * per group for the two flags, either may be NULL = all 1) of another state / another
 * GPU into this one (Merge, hash_aggregate_numeric.cc:85-107).  Asynchronous. */
int arx_groupby_sum_i64_merge(void* state, int64_t capacity, const int32_t* keys,
                              const uint8_t* key_is_valid, const int64_t* sums,
                              const int64_t* counts, const uint8_t* no_nulls,
                              int64_t num_groups, void* stream);
/* out_min_max (device int32[2], initialised by the caller to {INT32_MAX, INT32_MIN}) = {min, max} of the key slots
 * (null slots included: they only widen the range).  max - min + 1 bounds the number of groups, usually far below the
 * number of rows for an int32 id / code column: what aggregate_rocm sizes its table from (the reference's Grouper grows
 * its table instead, compute/row/grouper.cc).  Asynchronous. */
int arx_groupby_key_range_i32(const ArxSpan* keys, int32_t* out_min_max, void* stream);
/* Synchronous: number of groups currently in the table (host int64); ARX_INVALID if the
 * table overflowed. */
int arx_groupby_num_groups(void* state, int64_t* out_num_groups, void* stream);
/* export: dense partial-aggregate columns, arx_groupby_num_groups entries each, in an
 * unspecified order (Grouper::GetUniques + the aggregator state).  Synchronous on entry
 * (reads the header), kernels asynchronous. */
int arx_groupby_sum_i64_export(void* state, int32_t* out_keys, uint8_t* out_key_is_valid,
                               int64_t* out_sums, int64_t* out_counts, uint8_t* out_no_nulls,
                               void* stream);
/* hash_min / hash_max(int64) on the same table — GroupedMinMaxImpl (kernels/hash_aggregate.cc:
 * 330-419): a second caller-owned buffer (arx_groupby_minmax_bytes) holds mins | maxes per slot,
 * initialised to the anti-extrema (:349-350).  consume inserts the keys like the sum consume
 * (a group exists even if all its values are null), folds valid values into both extrema and sets
 * the group's null flag for null values; sums / counts are not touched, so it can run next to
 * arx_groupby_sum_i64_consume on the same rows.  Synchronous (reads the overflow flag).
 * arx_groupby_export is the sum export with the extrema as two more columns in the SAME group
 * order (minmax / out_mins / out_maxs may all be NULL).  finalize: out_valid[g] = the group saw a
 * value (min <= max) && (skip_nulls || no_nulls[g]) — min_count is not consulted (:401-410). */
/* Read-only probe: out[i] = (int32) the sum column of keys[i]'s group, -1 if the key (or, for a null
 * key, the null group) is not in the table.  dictionary_encode (DictEncodeAction,
 * kernels/vector_hash.cc:192-270) = unique in first-appearance order, the groups' positions merged
 * back in as their "sums", then this lookup over the rows.  Asynchronous. */
int arx_groupby_lookup_i32(void* state, int64_t capacity, const ArxSpan* keys_i32, int32_t* out, void* stream);
size_t arx_groupby_minmax_bytes(int64_t capacity);
int arx_groupby_minmax_init(void* minmax, int64_t capacity, void* stream);
int arx_groupby_minmax_i64_consume(void* state, void* minmax, int64_t capacity, const ArxSpan* keys_i32,
                                   const ArxSpan* values_i64, void* stream);
int arx_groupby_export(void* state, const void* minmax, int32_t* out_keys, uint8_t* out_key_is_valid,
                       int64_t* out_sums, int64_t* out_counts, uint8_t* out_no_nulls, int64_t* out_mins,
                       int64_t* out_maxs, void* stream);
/* merge (hash_aggregate.cc:371-399): exported extrema of another state fold into this one. */
int arx_groupby_minmax_merge(void* state, void* minmax, int64_t capacity, const int32_t* keys,
                             const uint8_t* key_is_valid, const int64_t* mins, const int64_t* maxs,
                             const uint8_t* no_nulls, int64_t num_groups, void* stream);
int arx_groupby_minmax_finalize(const int64_t* mins, const int64_t* maxs, const uint8_t* no_nulls,
                                int64_t num_groups, int skip_nulls, uint8_t* out_valid, void* stream);
/* finalize (Finalize, hash_aggregate_numeric.cc:130-152, ScalarAggregateOptions
 * {skip_nulls, min_count}): out_valid[g] (one byte) = counts[g] >= min_count &&
 * (skip_nulls || no_nulls[g]); the sums column is returned as is.  Asynchronous. */
int arx_groupby_sum_i64_finalize(const int64_t* counts, const uint8_t* no_nulls,
                                 int64_t num_groups, int skip_nulls, uint32_t min_count,
                                 uint8_t* out_valid, void* stream);
/* hash_mean(int64) — GroupedMeanImpl (hash_aggregate_numeric.cc:352-430): the reference sums DOUBLES in row order,
 * which is reproducible by a parallel reduction exactly when every partial sum of a group is an exactly
 * representable integer: count * max(|min|, |max|) < 2^53.  Inputs = the columns of arx_groupby_export WITH the
 * extrema (the rows must also have gone through arx_groupby_minmax_i64_consume).  out_means[g] = (double)sum / count
 * where count >= min_count (0 / 0 = NaN as in the reference), else 0; out_valid as for the sum; *out_inexact
 * (device, caller-zeroed) is set if any group breaks the bound — the caller must then decline (the result would
 * depend on row order in the reference too).  Asynchronous. */
int arx_groupby_mean_i64_finalize(const int64_t* sums, const int64_t* counts, const int64_t* mins, const int64_t* maxs,
                                  const uint8_t* no_nulls, int64_t num_groups, int skip_nulls, uint32_t min_count,
                                  double* out_means, uint8_t* out_valid, uint32_t* out_inexact, void* stream);
/* Radix partition of partial aggregates by hash(key) % num_parts for the multi-GPU
 * exchange: rows are written grouped by destination (order inside a
 * destination unspecified); out_part_counts = device int64[num_parts].  Asynchronous. */
size_t arx_groupby_partition_workspace_bytes(int num_parts);
int arx_groupby_partition(const int32_t* keys, const uint8_t* key_is_valid, const int64_t* sums,
                          const int64_t* counts, const uint8_t* no_nulls, int64_t num_groups,
                          int num_parts, void* ws, size_t ws_bytes, int32_t* out_keys,
                          uint8_t* out_key_is_valid, int64_t* out_sums, int64_t* out_counts,
                          uint8_t* out_no_nulls, int64_t* out_part_counts, void* stream);

/* The same exchange with ONE buffer per peer: a state's groups leave as 24-byte records grouped by
 * destination rank (no dense column export in between), so the multi-GPU group-by is one count
 * exchange + one all-to-all(v) of bytes; the receiver folds the records in with Merge semantics
 * (hash_aggregate_numeric.cc:85-107).  out_records needs arx_groupby_num_groups entries.
 * export: synchronous on entry (reads the header), kernels asynchronous.  merge: asynchronous. */
typedef struct ArxGroupPartial {
  int64_t sum;          /* wrap-around partial sum */
  int64_t count;        /* valid values folded into it */
  int32_t key;
  uint8_t key_is_valid; /* 0 = the null-key group */
  uint8_t no_nulls;     /* 0 once a null value hit the group */
  uint8_t pad[2];
} ArxGroupPartial;
/* The ROW-level exchange of the same group-by (the variant BASELINE.json words as "shard by radix
 * partition ... all-to-all"): the rows themselves are grouped by the rank that owns their key (the same hash(key) %
 * num_parts) as 16-byte records, exchanged, unpacked into key / value columns with validity bitmaps and consumed by the
 * receiver's own table.  Pays off only when almost every row is its own group; otherwise the partials exchange moves
 * G x 24 bytes instead of N x 16.  flags: bit 0 = key valid, bit 1 = value valid.  Asynchronous. */
typedef struct ArxRowRecord {
  int32_t key;
  uint32_t flags;
  int64_t value;
} ArxRowRecord;
int arx_groupby_partition_rows(const ArxSpan* keys_i32, const ArxSpan* values_i64, int num_parts, void* ws, size_t ws_bytes,
                               ArxRowRecord* out_records, int64_t* out_part_counts /* device int64[num_parts] */, void* stream);
/* out_key_validity / out_value_validity: ceil(num_records / 64) 64-bit words each. */
int arx_groupby_unpack_rows(const ArxRowRecord* records, int64_t num_records, int32_t* out_keys, int64_t* out_values,
                            void* out_key_validity, void* out_value_validity, void* stream);
/* The sharded group-by's local pass WITHOUT the local table (round 5): the partitioned consume as
 * arx_groupby_sum_i64_consume runs it, but a work unit's groups leave as ArxGroupPartial records written straight into the
 * region of the rank that owns each key (hash(key) % num_parts — the owner arx_groupby_export_partitioned assigns) instead
 * of being folded into the HBM table and exported afterwards: no table probe and two atomics per group, no export pass
 * (ThreadLocalState + Merge of acero/groupby_aggregate_node.cc:211-337 with the "state" being the record stream itself).
 * A key may appear in several records of a region (once per slice / work unit): the receiver's merge adds them up.
 * out_records: num_parts regions of records_per_part records (arx_groupby_partials_capacity); out_part_counts (device
 * int64[num_parts]): records in every region.  ARX_NOT_IMPLEMENTED: rows with nulls, or a batch too small for the
 * partitioned consume — use the table path; ARX_CAPACITY_ERROR: a region overflowed (nothing else was touched: use the table path).
 * `capacity` (the slots a local table WOULD have: > 2 x the distinct keys expected) only plans the pass; `state` is not
 * used and may be NULL.  Synchronous at its end (reads the counts). */
int64_t arx_groupby_partials_capacity(int64_t num_rows, int64_t capacity, int num_parts);
int arx_groupby_sum_i64_consume_partials(void* state, int64_t capacity, const ArxSpan* keys_i32, const ArxSpan* values_i64,
                                         void* ws, size_t ws_bytes, int num_parts, ArxGroupPartial* out_records,
                                         int64_t records_per_part, int64_t* out_part_counts, void* stream);
int arx_groupby_export_partitioned(void* state, int num_parts, void* ws /* arx_groupby_partition_workspace_bytes */,
                                   size_t ws_bytes, ArxGroupPartial* out_records,
                                   int64_t* out_part_counts /* device int64[num_parts] */, void* stream);
int arx_groupby_sum_i64_merge_records(void* state, int64_t capacity, const ArxGroupPartial* records,
                                      int64_t num_records, void* stream);

/* ---------------------------------------------------------------------------
 * The RANGE-PARTITIONED group-by state (round 6): hash_sum(int64) BY int32 for keys from a narrow range — ids, codes,
 * dictionary indices; BASELINE configs[3] draws its 1e7 keys from [0, 1e7).  Same operator as arx_groupby_* above
 * (ThreadLocalState -> Merge -> Finalize of acero/groupby_aggregate_node.cc:210-337 around
 * GroupedReducingAggregator<Int64,Sum>, compute/kernels/hash_aggregate_numeric.cc:44-187), another state: NO hash table.
 * The keys [key_min, key_min + partitions * width) are cut into `partitions` slices of `width` keys, and the state is one
 * dense block per partition, { uint64 sums[width] | uint64 counts[width] } = 16 * width bytes, zero-initialised by the
 * caller.  consume radix-partitions the rows by key slice with a write-combined scatter (every global store a whole
 * 128-byte line of 12 {value, 16-bit key remainder} records) and aggregates every partition in a direct-indexed LDS table
 * (csrc/groupby_lines.h); Merge is a vector add of two states (hash_aggregate_numeric.cc:85-107: sums wrap, counts add);
 * Finalize compacts the non-empty slots in key order (:109-152: valid where count >= min_count).  Sharded over P GPUs the
 * owner of a key is the owner of its PARTITION (contiguous runs of partitions per rank): a rank sends every owner one
 * contiguous run of blocks whose size every rank knows (no count exchange), the owner adds P runs and finalizes.
 * Rows with nulls are not taken (ARX_NOT_IMPLEMENTED: the table operator above serves them).
 * ------------------------------------------------------------------------- */
typedef struct ArxRangePlan {
  int32_t key_min;          /* first key of partition 0 */
  int32_t width;            /* keys per partition: 8 ... 8192 (powers of two) or 12288 */
  int32_t partitions;       /* 384 ... 1216 */
  int32_t reserved;
  int64_t slots;            /* partitions * width */
  uint64_t state_bytes;     /* slots * 16 */
  uint64_t workspace_bytes; /* scratch of a consume of up to max_rows rows (256-byte aligned) */
} ArxRangePlan;
/* The plan for keys in [key_min, key_max] and consumes of up to max_rows rows: ARX_NOT_IMPLEMENTED when the range needs
 * more than 1216 partitions of 12288 keys (14.9M keys) or fewer than 384 of 8 (3072 keys: one LDS table holds those
 * groups — the table operator's direct plan is the tool).  Pure host arithmetic. */
int arx_groupby_range_plan(int64_t max_rows, int32_t key_min, int32_t key_max, ArxRangePlan* out);
/* {min, max} of a SAMPLE of the key slots (one 64-row unit per stratum, about sample_rows rows in all; sample_rows >= length
 * reads every key) folded into out_min_max (device int32[2], initialised by the caller to {INT32_MAX, INT32_MIN}).  A
 * caller that plans from a sample widens the range (the sample misses a few keys at both ends).  Asynchronous. */
int arx_groupby_key_range_sampled_i32(const ArxSpan* keys, int64_t sample_rows, int32_t* out_min_max, void* stream);
/* Adds the rows' groups into `state` (plan->state_bytes, zeroed before the first consume).  ARX_CAPACITY_ERROR — and
 * nothing consumed — when a key lies outside the plan's range, a hot key makes the scatter give up, or `ws` is too small
 * (use the table operator for these rows); ARX_NOT_IMPLEMENTED for rows with nulls.  Synchronous (reads the scatter's flags). */
int arx_groupby_range_sum_i64_consume(void* state, const ArxRangePlan* plan, const ArxSpan* keys_i32, const ArxSpan* values_i64,
                                      void* ws, size_t ws_bytes, void* stream);
/* partitions[0 .. num_partitions) at `state` += the same partitions of num_others other states, the r-th starting at
 * others + r * others_stride_bytes (all pointers at a partition's first byte; the blocks a rank received from its peers
 * lie one behind the other).  Asynchronous. */
int arx_groupby_range_merge(void* state, const void* others, int32_t width, int64_t num_partitions, int num_others,
                            int64_t others_stride_bytes, void* stream);
/* The groups of num_partitions partitions starting at `partitions` (the first one's keys start at first_key), ascending by
 * key: out_keys / out_sums / out_counts (may be NULL) / out_valid (1 where count >= min_count) need num_partitions * width
 * entries at most; *out_num_groups (device int64) = how many were written.  ws: arx_groupby_range_finalize_workspace_bytes.
 * Asynchronous. */
size_t arx_groupby_range_finalize_workspace_bytes(int64_t slots);
int arx_groupby_range_finalize(const void* partitions, int32_t first_key, int32_t width, int64_t num_partitions, uint32_t min_count,
                               void* ws, size_t ws_bytes, int32_t* out_keys, int64_t* out_sums, int64_t* out_counts,
                               uint8_t* out_valid, int64_t* out_num_groups, void* stream);

/* ---------------------------------------------------------------------------
 * hash_sum(int64, uint32 group id) — the HashAggregateKernel boundary itself
 * (compute/kernel.h:720-769): the caller's Grouper already produced dense group ids.
 * Replaces GroupedReducingAggregator<Int64Type,GroupedSumImpl>::{Consume,Merge,Finalize}
 * (compute/kernels/hash_aggregate_numeric.cc:70-152).  State = three dense device arrays of
 * num_groups entries that the caller owns, zero-initialises and grows (Resize, :61-68):
 * sums (wrap-around int64), counts, null_seen (bit 0 set once a null value hit the group;
 * the reference keeps the complement, `no_nulls`).  All asynchronous.
 * ------------------------------------------------------------------------- */
/* values: int64 ArxSpan of `length` rows, or (values_is_scalar != 0) a broadcast scalar
 * `scalar_value` whose validity is values->null_count == 0 (hash_aggregate_internal.h:165-175). */
int arx_hash_sum_i64_consume(const ArxSpan* values, int values_is_scalar, int64_t scalar_value,
                             const uint32_t* group_ids, int64_t length, int64_t* sums,
                             int64_t* counts, uint32_t* null_seen, void* stream);
/* The same consume with scratch (256-byte aligned, arx_hash_sum_consume_workspace_bytes): rows are radix-partitioned by
 * the top bits of the group id until a partition's ids fit an LDS table (<= 2048 ids: a perfect, probe-free layout),
 * aggregated there with LDS atomics and flushed once per partition — device atomics into the state arrays run at
 * ~12 Grows/s at best and collapse when a few groups are hot.  num_groups = current size of the state arrays
 * (every id < num_groups).  Falls back to the per-row form for broadcast scalars, small batches or short scratch. */
size_t arx_hash_sum_consume_workspace_bytes(int64_t length, int64_t num_groups);
int arx_hash_sum_i64_consume_ws(const ArxSpan* values, int values_is_scalar, int64_t scalar_value,
                                const uint32_t* group_ids, int64_t length, int64_t num_groups, int64_t* sums,
                                int64_t* counts, uint32_t* null_seen, void* ws, size_t ws_bytes, void* stream);
/* hash_mean(int64) on the same dense state — GroupedMeanImpl::Finalize (hash_aggregate_numeric.cc:352-430) divides a
 * sum accumulated as DOUBLES in row order; out_means[g] = double(sums[g]) / counts[g] equals it bit for bit when every
 * partial sum is an integer below 2^53, guaranteed when counts[g] * abs_bound < 2^53 with abs_bound >= |value| of
 * every consumed row (the caller's min_max of the column).  *inexact (device, caller-zeroed) is set when a group
 * breaks the bound: the caller then declines, the reference's own result depends on row order there.  Validity of a
 * group is arx_hash_sum_i64_finalize's.  Groups with count 0 get 0.0. */
int arx_hash_mean_i64_finalize(const int64_t* sums, const int64_t* counts, int64_t num_groups, uint64_t abs_bound,
                               double* out_means, uint32_t* inexact, void* stream);

/* hash_min / hash_max / hash_min_max(int64, uint32 group id) — GroupedMinMaxImpl (compute/kernels/hash_aggregate.cc:
 * 330-419) on dense state arrays the caller owns: mins / maxs (arx_hash_minmax_i64_fill sets the anti-extrema of new
 * groups — Resize, :343-353), null_seen (zero-initialised; bit 0 = a null value hit the group).  Consume :355-379, Merge
 * :381-399, Finalize :401-419: out_validity bit g = the group saw a value && (skip_nulls || saw no null) — min_count is
 * not consulted by the reference either; the data of a null group keeps the anti-extremum.  values / scalar conventions
 * of arx_hash_sum_i64_consume.  All asynchronous. */
int arx_hash_minmax_i64_fill(int64_t* mins, int64_t* maxs, int64_t first_group, int64_t num_new_groups, void* stream);
int arx_hash_minmax_i64_consume(const ArxSpan* values, int values_is_scalar, int64_t scalar_value, const uint32_t* group_ids,
                                int64_t length, int64_t* mins, int64_t* maxs, uint32_t* null_seen, void* stream);
int arx_hash_minmax_i64_merge(int64_t* mins, int64_t* maxs, uint32_t* null_seen, const int64_t* other_mins,
                              const int64_t* other_maxs, const uint32_t* other_null_seen, const uint32_t* group_id_mapping,
                              int64_t other_num_groups, void* stream);
int arx_hash_minmax_i64_finalize(const int64_t* mins, const int64_t* maxs, const uint32_t* null_seen, int64_t num_groups,
                                 int skip_nulls, void* out_validity, int64_t* valid_count, void* stream);
/* The same for float32 / float64 values (num_type ARX_NUM_FLOAT32 / _FLOAT64; MinMaxOp = fmin / fmax over NaN
 * anti-extrema, hash_aggregate.cc:306-326).  The state arrays are the int64 arrays above and hold the values' ORDER KEYS
 * (the int64 whose signed order is the doubles' numeric order; -0.0 sorts just below +0.0 — the tie the reference
 * leaves to row order), so arx_hash_minmax_i64_fill and _merge serve them unchanged; null_seen bit 1 = the group saw a
 * value (a NaN row sets only that: fmin / fmax skip NaNs).  Finalize writes the extrema back as num_type values
 * (out_mins / out_maxs: either may be NULL; a group of NaNs only gets NaN) and the validity bit g = saw a value &&
 * (skip_nulls || saw no null).  scalar_value: the broadcast value of a scalar column. */
int arx_hash_minmax_float_consume(const ArxSpan* values, int num_type, int values_is_scalar, double scalar_value,
                                  const uint32_t* group_ids, int64_t length, int64_t* mins, int64_t* maxs, uint32_t* null_seen,
                                  void* stream);
int arx_hash_minmax_float_finalize(const int64_t* mins, const int64_t* maxs, const uint32_t* null_seen, int64_t num_groups,
                                   int skip_nulls, int num_type, void* out_mins, void* out_maxs, void* out_validity,
                                   int64_t* valid_count, void* stream);
/* hash_count(any, uint32 group id) — GroupedCountImpl (compute/kernels/hash_aggregate.cc:107-212): counts[g] += 1 for
 * the rows of group g whose value is valid (mode 0, CountOptions::ONLY_VALID), null (1, ONLY_NULL) or either (2, ALL).
 * Only the values' validity is read: values_validity + values_offset (NULL: no nulls; values_null_count != 0 with a
 * NULL bitmap: a null broadcast scalar, every row null).  counts: dense, caller-owned, zero-initialised.  Merge adds
 * other_counts[i] to counts[group_id_mapping[i]] (:156-170); Finalize is the counts array itself, never null. */
#define ARX_COUNT_ONLY_VALID 0
#define ARX_COUNT_ONLY_NULL 1
#define ARX_COUNT_ALL 2
int arx_hash_count_consume(const void* values_validity, int64_t values_offset, int64_t values_null_count, int mode,
                           const uint32_t* group_ids, int64_t length, int64_t* counts, void* stream);
int arx_hash_count_merge(int64_t* counts, const int64_t* other_counts, const uint32_t* group_id_mapping,
                         int64_t other_num_groups, void* stream);

/* hash_any / hash_all(boolean, uint32 group id) — GroupedBooleanAggregator<GroupedAnyImpl / GroupedAllImpl>
 * (compute/kernels/hash_aggregate.cc:1232-1398).  The state is three dense count arrays kept with arx_hash_count_consume /
 * _merge: n_valid (the boolean column's validity, ARX_COUNT_ONLY_VALID), n_null (ONLY_NULL) and n_true (ONLY_VALID over
 * the bitmap validity AND value, arx_bitmap_and) — counts[g] of the reference is n_valid, no_nulls[g] is n_null == 0,
 * reduced[g] is n_true > 0 (any) or n_true == n_valid (all).  This call is Finalize (:1321-1350): out_values /
 * out_validity = LSB-first bitmaps of num_groups bits (whole 64-bit words written); valid iff n_valid >= min_count and —
 * unless skip_nulls — no null was seen or the value is already decided by what was seen (AdjustForMinCount, :1376-1398);
 * *valid_count (device, may be NULL) is incremented by the number of valid groups.  Asynchronous. */
int arx_hash_bool_finalize(const int64_t* n_valid, const int64_t* n_null, const int64_t* n_true, int64_t num_groups, int is_all,
                           int skip_nulls, uint32_t min_count, void* out_values, void* out_validity, int64_t* valid_count,
                           void* stream);

/* ---------------------------------------------------------------------------
 * Grouper: key rows of one or several fixed-width columns -> dense group ids.
 * arrow::compute::Grouper (cpp/src/arrow/compute/row/grouper.h:104-137): Consume :121, Lookup :126, GetUniques :134,
 * num_groups :137; the implementation followed is GrouperFastImpl (row/grouper.cc:555-973: encode the key columns as
 * rows, map rows through a hash table, ids in order of first appearance, ConsumeImpl :695-815; uniques decoded from
 * the stored rows, GetUniques :835-880).  This is the piece GroupByNode calls before every hash_* kernel
 * (acero/groupby_aggregate_node.cc:210-258), so with arx_hash_sum_i64_consume_ws & co. it covers the keys the fused
 * 32-bit operator (arx_groupby_*) does not: int64 / uint64 keys over their whole range and several key columns.
 *
 * Key columns: byte widths 1, 2, 4 or 8 (integers, temporal types ... compared by their bits), at most 8 columns and
 * 16 bytes per row in total (one 32-byte slot per key row: wider rows are the HOST's to chain — level s consumes the
 * row (uint32 ids of level s-1, next columns), as arrow_amd.compute.Grouper and aggregate_rocm do).
 * A null is a key value of its own (all rows whose column j is null agree in column j),
 * as in the reference.  Group ids are in order of first appearance: the k-th distinct key row in row order gets id k,
 * across calls (GrouperImpl's order and that of every expectation in grouper_test.cc; GrouperFastImpl's ids are a
 * bijection away, which is what the reference's own AssertEquivalentIds accepts).  max_groups bounds the distinct key rows over the Grouper's life; exceeding it fails the call with
 * ARX_INVALID and leaves the state unusable (arx_grouper_init again).
 * state: device, arx_grouper_state_bytes(max_groups), 256-byte aligned.  ws: device, 256-byte aligned.  Synchronous. */
size_t arx_grouper_state_bytes(int64_t max_groups);
int arx_grouper_init(void* state, int64_t max_groups, void* stream);
size_t arx_grouper_consume_workspace_bytes(int64_t length);
/* out_group_ids: device uint32[length]. */
int arx_grouper_consume(void* state, int64_t max_groups, const ArxSpan* key_columns, const int32_t* key_byte_widths,
                        int num_keys, void* ws, size_t ws_bytes, uint32_t* out_group_ids, void* stream);
/* Lookup (:126): rows whose key has not been consumed get a null id: out_validity = bitmap of length bits (device,
 * 8-byte aligned, ceil(length/64) words), bit clear = unseen (its id slot holds 0).  Adds no group. */
int arx_grouper_lookup(void* state, int64_t max_groups, const ArxSpan* key_columns, const int32_t* key_byte_widths,
                       int num_keys, void* ws, size_t ws_bytes, uint32_t* out_group_ids, uint8_t* out_validity,
                       void* stream);
/* group_ids[i] -= 1 where group_ids[i] > skipped_id: the renumbering DictionaryEncode needs when the null's group is
 * left out of the dictionary (null_encoding MASK; DictEncodeAction, kernels/vector_hash.cc:173-270).  Asynchronous. */
int arx_group_ids_skip_group(uint32_t* group_ids, int64_t length, uint32_t skipped_id, void* stream);
int arx_grouper_num_groups(void* state, int64_t* out_num_groups, void* stream);
/* GetUniques (:134), one key column per call: out_values = num_groups values of key_byte_widths[key_index] bytes in
 * group-id order, out_validity = their validity bitmap (ceil(num_groups/64) words, 8-byte aligned). */
int arx_grouper_get_uniques(void* state, int64_t max_groups, const int32_t* key_byte_widths, int num_keys, int key_index,
                            void* out_values, uint8_t* out_validity, int64_t* out_null_count, void* stream);

/* Var-width (utf8 / binary, int32 offsets) KEY columns for the chain of Grouper tables — the var-length part of the
 * reference's key rows (row/grouper.cc:559-611; RowTableEncoder's varbinary columns): the column becomes fixed-width
 * virtual key columns, its LENGTH (uint32; 0xFFFFFFFF for a null, so that null is a key value of its own and differs
 * from "") and then 12 bytes of the string per table level as one uint64 (bytes 0-7, little endian) and one uint32
 * (bytes 8-11), zero-padded past the string's end; ceil(max_length / 12) chunks tell any two different strings apart.
 * arx_binary_key_lengths: synchronous (returns the longest valid string's length); ws: >= 8 device bytes.
 * arx_group_first_rows: out_first_rows[g] = the smallest row of group g (the row GetUniques reports for g): the
 * unique strings are `take(values, first_rows)`.  length < 2^32 - 1. */
int arx_binary_key_lengths(const ArxBinarySpan* values, uint32_t* out_lengths, int64_t* out_max_length, void* ws,
                           void* stream);
int arx_binary_key_chunk(const ArxBinarySpan* values, int64_t chunk_index, uint64_t* out_lo, uint32_t* out_hi,
                         void* stream);
/* utf8 / binary SORT keys (array_sort_indices of BaseBinary types, kernels/vector_array_sort.cc:144-178; strings compare
 * bytewise, the shorter first on a common prefix): out_keys[i] = bytes [8 c, 8 c + 8) of string i as one BIG-endian uint64,
 * zero-padded, 0 for a null.  The keys (chunk 0, chunk 1, ..., chunk ceil(max_length / 8) - 1, length) — lengths from
 * arx_binary_key_lengths — ordered as unsigned integers are that order; sorted as a chain of stable sorts, last key first. */
int arx_binary_sort_chunk(const ArxBinarySpan* values, int64_t chunk_index, uint64_t* out_keys, void* stream);
/* Var-width keys in ONE pass: out_hash[i] = a 64-bit hash of the bytes of string i (0 for a null; only the low
 * hash_bits bits are kept — 64 in production, fewer to force collisions in tests); with the length column
 * (arx_binary_key_lengths) that is a 12-byte stand-in for the string whatever its length.  arx_binary_key_verify then
 * counts the rows whose bytes differ from the bytes of their group's first row (first_rows: arx_group_first_rows) —
 * 0 means the hashed groups are the exact groups (the reference compares the encoded rows on a hash match the same way,
 * compute/row/grouper.cc:695-815); otherwise the caller groups by the exact chunk columns instead.  verify is
 * synchronous; ws: >= 8 device bytes. */
int arx_binary_key_hash(const ArxBinarySpan* values, int hash_bits, uint64_t* out_hash, void* stream);
int arx_binary_key_verify(const ArxBinarySpan* values, const uint32_t* group_ids, const uint32_t* first_rows,
                          int64_t* out_mismatches, void* ws, void* stream);
/* hash_first / hash_last / hash_one (GroupedFirstLastImpl kernels/hash_aggregate.cc:738-925, GroupedOneImpl :1556-1625 keep the
 * first / last NON-NULL value of every group in row order): out_rows[g] = the smallest (last = 0) or largest (last != 0) row
 * of group g whose bit in `validity` (NULL: every row) is set, 0 where the group has none; out_has_row = a bitmap of
 * ceil(num_groups / 64) words saying which groups have one — the validity of the index array `take(values, out_rows)`
 * is called with.  length < 2^32 - 2.  Asynchronous. */
int arx_group_edge_rows(const uint32_t* group_ids, const void* validity, int64_t validity_offset, int64_t length,
                        int64_t num_groups, int last, uint32_t* out_rows, void* out_has_row, void* stream);
int arx_group_first_rows(const uint32_t* group_ids, int64_t length, int64_t num_groups, uint32_t* out_first_rows,
                         void* stream);


/* hash_sum / hash_mean of float32 / float64 values (num_type ARX_NUM_FLOAT32 / _FLOAT64) —
 * GroupedReducingAggregator<FloatType / DoubleType, GroupedSumImpl | GroupedMeanImpl> (hash_aggregate_numeric.cc:44-152,
 * 352-430): every row is added to its group's DOUBLE accumulator in ROW ORDER (Reduce = double(u) + double(v)); the value of
 * such a sum depends on the order, so the device keeps it — the rows are stably sorted by group id and one thread walks each
 * group's run from the group's running sum.  State: sums (double), counts, null_seen as for the integer kernels; batches
 * continue where the last one stopped.  ws: arx_hash_sum_float_workspace_bytes(length).  A broadcast scalar: scalar_value,
 * valid iff values->null_count == 0.  _merge: sums[mapping[g]] += other_sums[g] (each target once per call, :85-107);
 * _mean_finalize: sums[g] / counts[g] (DoMean :381-385; 0 where the count is 0 — validity comes from
 * arx_hash_sum_i64_finalize, which reads counts / null_seen only).  Asynchronous but for the sort's own synchronisation. */
size_t arx_hash_sum_float_workspace_bytes(int64_t length);
int arx_hash_sum_float_consume(const ArxSpan* values, int num_type, int values_is_scalar, double scalar_value,
                               const uint32_t* group_ids, int64_t length, void* ws, size_t ws_bytes, double* sums,
                               int64_t* counts, uint32_t* null_seen, void* stream);
/* hash_product — GroupedProductImpl (kernels/hash_aggregate_numeric.cc:311-347): per group the product of the valid values in row
 * order from 1, in the sum's accumulator type (int64 / uint64: MultiplyTraits wraps in the unsigned type; double).  products: 8
 * bytes per group, arx_hash_product_init fills them with 1 / 1.0; counts / null_seen / workspace as for arx_hash_sum_float_consume
 * (whose finalize serves min_count / skip_nulls here too).  num_type: any ARX_NUM_*.  Asynchronous. */
int arx_hash_product_init(void* products, int num_type, int64_t num_groups, void* stream);
int arx_hash_product_consume(const ArxSpan* values, int num_type, const uint32_t* group_ids, int64_t length, void* ws,
                             size_t ws_bytes, void* products, int64_t* counts, uint32_t* null_seen, void* stream);
int arx_hash_sum_f64_merge(double* sums, int64_t* counts, uint32_t* null_seen, const double* other_sums,
                           const int64_t* other_counts, const uint32_t* other_null_seen,
                           const uint32_t* group_id_mapping, int64_t other_num_groups, void* stream);
int arx_hash_mean_f64_finalize(const double* sums, const int64_t* counts, int64_t num_groups, double* out_means,
                               void* stream);
/* hash_variance / hash_stddev / hash_skew / hash_kurtosis — GroupedStatisticImpl (kernels/hash_aggregate_numeric.cc:457-843),
 * the two-pass form of its ConsumeGeneric (:555-615) over ALL rows of the node at once: pass 1 is arx_hash_sum_float_consume
 * (sums, counts: mean = sum / count); arx_group_central_power writes (x - mean of x's group)^power (power 2, 3 or 4; 0 for null
 * rows, which keep their validity) as a float64 column whose arx_hash_sum_float_consume is the moment m2 / m3 / m4;
 * arx_hash_moments_finalize applies Moments::Variance / Stddev / Skew / Kurtosis (kernels/aggregate_var_std_internal.h:83-116).
 * The reference computes the same per batch and merges batches (Moments::Merge): equal up to floating-point rounding, not bit
 * for bit — its own tests compare with a tolerance (hash_aggregate_test.cc VarianceAndStddev / SkewAndKurtosis).  A group the
 * reference leaves null for its count (count <= ddof; unbiased skew: count <= 2, kurtosis: count <= 3) reads 0; the validity
 * comes from arx_hash_sum_i64_finalize with min_count raised to that bound.  values: float32 / float64 (integers are cast to
 * float64 first, ToDouble :537-539).  Asynchronous. */
enum { ARX_STAT_VARIANCE = 0, ARX_STAT_STDDEV = 1, ARX_STAT_SKEW = 2, ARX_STAT_KURTOSIS = 3 };
int arx_group_central_power(const ArxSpan* values, int num_type, const uint32_t* group_ids, int64_t length, const double* sums,
                            const int64_t* counts, int power, double* out, void* stream);
int arx_hash_moments_finalize(const int64_t* counts, const double* m2, const double* m3, const double* m4, int64_t num_groups,
                              int stat, int ddof, int biased, double* out, void* stream);
/* hash_sum of decimal128 values (16 bytes a value, little-endian two's complement) —
 * GroupedReducingAggregator<Decimal128Type, GroupedSumImpl> (hash_aggregate_numeric.cc:44-152,189-215): a Decimal128 per
 * group, Reduce = BasicDecimal128 addition, i.e. modulo 2^128 — associative and commutative, kept with atomics: the low word
 * first, the carry (seen by exactly one row per wrap) into the high word's addend.  State: sums_lo / sums_hi / counts /
 * null_seen, zero-initialised and grown by the caller; a broadcast scalar: (scalar_lo, scalar_hi), valid iff
 * values->null_count == 0.  _merge: each target group once per call (:85-107).  arx_dec128_pack: (lo, hi) -> 16-byte values
 * (the output's type is the input's widened to precision 38, :165-167; validity from arx_hash_sum_i64_finalize).
 * Asynchronous. */
int arx_hash_sum_dec128_consume(const ArxSpan* values, int values_is_scalar, uint64_t scalar_lo, uint64_t scalar_hi,
                                const uint32_t* group_ids, int64_t length, uint64_t* sums_lo, uint64_t* sums_hi,
                                int64_t* counts, uint32_t* null_seen, void* stream);
int arx_hash_sum_dec128_merge(uint64_t* sums_lo, uint64_t* sums_hi, int64_t* counts, uint32_t* null_seen,
                              const uint64_t* other_lo, const uint64_t* other_hi, const int64_t* other_counts,
                              const uint32_t* other_null_seen, const uint32_t* group_id_mapping,
                              int64_t other_num_groups, void* stream);
int arx_dec128_pack(const uint64_t* lo, const uint64_t* hi, int64_t n, void* out_values, void* stream);
/* The inverse: out_lo[i] / out_hi[i] = the low / high 64 bits of value i (`values`: pre-offset, 16 bytes a value).  A
 * decimal128 SORT KEY is the pair (high word as int64, low word as uint64) — BasicDecimal128's order, what
 * ConcreteColumnComparator<Decimal128Type> compares (kernels/vector_sort_internal.h) — and is sorted as those two keys. */
int arx_dec128_split(const void* values, int64_t n, uint64_t* out_lo, uint64_t* out_hi, void* stream);
/* hash_min / hash_max of decimal128 values — GroupedMinMaxImpl<Decimal128Type> (kernels/hash_aggregate.cc:330-419): per group
 * the smallest and largest value in signed 128-bit order.  No 128-bit atomics: the rows are stably sorted by group id, every
 * group is one run, one owner per group (a thread, or a wave for runs > 1024 rows) folds the run into the state.  State the
 * caller owns: mins / maxs (16 bytes a group, any content), seen (zeroed: bit 0 = a null hit the group, bit 1 = the group has a
 * value and mins / maxs hold its extrema); ws: arx_hash_minmax_dec128_workspace_bytes(length).  _finalize: validity =
 * has a value && (skip_nulls || no null) (:401-419), valid_count (device, caller-zeroed, may be NULL) += popcount. */
size_t arx_hash_minmax_dec128_workspace_bytes(int64_t length);
int arx_hash_minmax_dec128_consume(const ArxSpan* values, const uint32_t* group_ids, int64_t length, void* ws, size_t ws_bytes,
                                   void* mins, void* maxs, uint32_t* seen, void* stream);
int arx_hash_minmax_dec128_finalize(const uint32_t* seen, int64_t num_groups, int skip_nulls, void* out_validity,
                                    int64_t* valid_count, void* stream);
/* sum / mean / min_max of a decimal128 column — SumImpl / MeanImpl / MinMaxImpl<Decimal128Type>
 * (kernels/aggregate_basic.inc.cc:49-110,229-258,776-860): out8 (HOST) = {sum lo, sum hi (modulo 2^128), count of valid values,
 * 1 if any, min lo, min hi, max lo, max hi}.  ws: arx_reduce_dec128_workspace_bytes() device bytes.  Synchronous. */
size_t arx_reduce_dec128_workspace_bytes(void);
int arx_reduce_dec128(const ArxSpan* values, void* ws, size_t ws_bytes, uint64_t* out8, void* stream);
int arx_hash_sum_i64_merge(int64_t* sums, int64_t* counts, uint32_t* null_seen,
                           const int64_t* other_sums, const int64_t* other_counts,
                           const uint32_t* other_null_seen, const uint32_t* group_id_mapping,
                           int64_t other_num_groups, void* stream);
/* out_validity: ceil(num_groups/64) words, bit g = counts[g] >= min_count &&
 * (skip_nulls || !null_seen[g]); valid_count (device, caller-zeroed, may be NULL) += popcount. */
int arx_hash_sum_i64_finalize(const int64_t* counts, const uint32_t* null_seen, int64_t num_groups,
                              int skip_nulls, uint32_t min_count, void* out_validity,
                              int64_t* valid_count, void* stream);

/* ---------------------------------------------------------------------------
 * Parquet page decode (first slice): the RLE / bit-packed hybrid of definition
 * levels and dictionary indices — RleBitPackedDecoder, cpp/src/arrow/util/rle_encoding_internal.h:
 * 40-90,462-; LevelDecoder::SetData, cpp/src/parquet/column_reader.cc:128-172.  The caller walks the
 * (sequential, variable-length) run headers once on the host and passes a run table in device memory:
 * out_start = index of the run's first value, kind 0 = repeated run (payload = the value), 1 = literal
 * run (payload = byte offset of its first bit-packed group inside `bytes`); bits 8.. of kind, when
 * non-zero, override bit_width for that run (pages of one chunk may use different widths).  Asynchronous.
 *   arx_rle_decode_u32           : out[i] = value i                       (dictionary indices)
 *   arx_rle_decode_equals_bitmap : bit i = (value i == equals), LSB-first (def levels -> validity)
 * arx_expand_by_mask spreads `dense` (one element per set mask bit) over the mask's slots, zero
 * elsewhere — the inverse of a DROP filter; ws = the workspace arx_filter_count filled for this
 * mask (ARX_FILTER_DROP).  What the reader's "spaced" decode does for optional columns.
 * ------------------------------------------------------------------------- */
typedef struct ArxRleRun {
  uint32_t out_start;
  uint32_t kind;
  uint64_t payload;
} ArxRleRun;
/* HOST function (no device work): walks the run headers of one block in host memory and fills the run
 * table (runs may be NULL to only count: *num_runs, and for bit_width 1 *ones = number of values equal
 * to 1, i.e. the non-null count of a definition-level block).  out_start / payload are shifted by
 * out_base / byte_base so that several pages can share one table and one byte buffer. */
int arx_rle_scan_runs(const void* data, size_t nbytes, int bit_width, int64_t num_values, uint32_t out_base,
                      uint64_t byte_base, ArxRleRun* runs, int64_t max_runs, int64_t* num_runs, int64_t* ones);
/* DELTA_BINARY_PACKED (DeltaBitPackDecoder, cpp/src/parquet/decoder.cc; INT32 / INT64 columns and the
 * length streams of DELTA_LENGTH_BYTE_ARRAY): value_i = value_{i-1} + min_delta(block) + unpack(miniblock),
 * wrap-around.  arx_delta_scan_miniblocks is a HOST function: it walks the block headers of one page and
 * fills one entry per miniblock that holds values (bit_start = absolute bit position in the buffer the device
 * will see: data sits at byte_base in it).  arx_delta_decode unpacks + prefix-sums on the device: `bytes` must be
 * 8-byte aligned and readable 8 bytes past the last miniblock; out_byte_width 4 (INT32) or 8 (INT64).
 * Asynchronous. */
typedef struct ArxDeltaMiniblock {
  uint64_t bit_start;  /* first delta of the miniblock */
  int64_t min_delta;   /* of its block */
  uint32_t bit_width;  /* 0..64 */
  uint32_t reserved;
} ArxDeltaMiniblock;
int arx_delta_scan_miniblocks(const void* data, size_t nbytes, uint64_t byte_base, ArxDeltaMiniblock* out,
                              int64_t max_miniblocks, int64_t* num_miniblocks,
                              int64_t* values_per_miniblock, int64_t* total_values, int64_t* first_value,
                              size_t* bytes_consumed);
size_t arx_delta_decode_workspace_bytes(int64_t num_values);
int arx_delta_decode(const void* bytes, const ArxDeltaMiniblock* miniblocks, int64_t num_miniblocks,
                     int64_t values_per_miniblock, int64_t first_value, int64_t num_values,
                     int out_byte_width, void* ws, size_t ws_bytes, void* out, void* stream);

/* All DELTA_BINARY_PACKED pages of a column chunk in ONE launch sequence (three launches whatever the page count;
 * arx_delta_decode is one sequence per page).  Every page restarts at its own first value (DeltaBitPackDecoder::
 * InitHeader per page, decoder.cc), so page p owns whole tiles of 4096 values: first_tile = tiles of the pages before
 * it, total_tiles = their sum.  pages: DEVICE array; miniblocks: the chunk's table (first_miniblock indexes into it);
 * values of page p land at out + out_start (in values).  ws: arx_delta_decode_workspace_bytes(total_tiles * 4096). */
typedef struct ArxDeltaPage {
  int64_t out_start;            /* first output value of the page */
  int64_t first_miniblock;      /* index into `miniblocks` */
  int64_t values_per_miniblock;
  int64_t first_value;
  int64_t num_values;
  int64_t first_tile;
} ArxDeltaPage;
int arx_delta_decode_pages(const void* bytes, const ArxDeltaMiniblock* miniblocks, const ArxDeltaPage* pages, int64_t num_pages,
                           int64_t total_tiles, int out_byte_width, void* ws, size_t ws_bytes, void* out, void* stream);
/* DELTA_LENGTH_BYTE_ARRAY (DeltaLengthByteArrayDecoder, cpp/src/parquet/decoder.cc): the lengths are a
 * DELTA_BINARY_PACKED stream (arx_delta_decode, width 4), the offsets their running sum: out[0] = base,
 * out[i] = out[i-1] + lengths[i-1], i in [1, n].  ws: arx_delta_decode_workspace_bytes(n + 1).  Asynchronous. */
int arx_lengths_to_offsets_i32(const int32_t* lengths, int64_t n, int32_t base, int32_t* out, void* ws,
                               size_t ws_bytes, void* stream);
/* DELTA_BYTE_ARRAY (DeltaByteArrayDecoderImpl, cpp/src/parquet/decoder.cc:1974-2204): value i = the first prefix[i]
 * bytes of value i - 1 ++ suffix i, every page starting from the empty string (SetData :1988-2018).  The prefix lengths
 * are a DELTA_BINARY_PACKED stream and the suffixes a DELTA_LENGTH_BYTE_ARRAY block: arx_delta_decode (width 4) gives
 * `prefix` and `suffix_len` — the pages of a column chunk one after the other, page p holding values
 * [page_first[p], page_first[p + 1]) (device array of num_pages + 1 entries).
 *   _lengths: out_len[i] = prefix[i] + suffix_len[i] and the decoder's checks.  state (device, 2 x uint64, zeroed
 *             here): state[0] bit 0 "negative prefix length in DELTA_BYTE_ARRAY" (:2096), bit 1 "prefix length too large in
 *             DELTA_BYTE_ARRAY" (:2039: longer than the previous value; a page's first value: than ""), bit 2 a negative
 *             suffix length, bit 3 "excess expansion" (:2105: a value past the int32 offsets); state[1] = the bytes all
 *             values take (64-bit: what the int32 offsets must be able to hold).
 *   _expand:  suffix_offsets / out_offsets = arx_lengths_to_offsets_i32 of suffix_len (base 0) / out_len (any base;
 *             out_data[0] is the byte at offset out_base; out_data holds state[1] bytes); suffix_bytes: the pages' suffix
 *             bytes one after the other, 4-byte aligned and readable up to the next multiple of 4 past suffix_size;
 *             page_suffix_first (device, num_pages + 1, or NULL): where each page's suffix bytes start — a page whose
 *             suffix lengths do not add up to that is skipped and ORs bit 4 into state[0] (not zeroed here).  One wave
 *             walks a page value by value, the lanes copy bytes.  Call it only when _lengths left state[0] == 0 and
 *             state[1] fits the int32 offsets (the caller reads both to size out_data anyway); a value with a negative
 *             length or a suffix outside suffix_bytes is skipped rather than written.  Asynchronous. */
int arx_delta_byte_array_lengths(const int32_t* prefix, const int32_t* suffix_len, int64_t n, const int64_t* page_first,
                                 int64_t num_pages, int32_t* out_len, uint64_t* state, void* stream);
int arx_delta_byte_array_expand(const int32_t* prefix, const int32_t* suffix_offsets, const void* suffix_bytes,
                                int64_t suffix_size, const int32_t* out_offsets, int32_t out_base, const int64_t* page_first,
                                const int64_t* page_suffix_first, int64_t num_pages, void* out_data, uint64_t* state,
                                void* stream);
/* BYTE_STREAM_SPLIT (ByteStreamSplitDecoder, cpp/src/parquet/decoder.cc; arrow/util/byte_stream_split_internal.h):
 * `in` holds byte_width streams of num_values bytes each (stream k = byte k of every value); out[i] is value i.
 * byte_width 2, 4 or 8; `out` aligned to it.  Asynchronous. */
int arx_byte_stream_split_decode(const void* in, int64_t num_values, int byte_width, void* out, void* stream);
/* HOST function: PLAIN BYTE_ARRAY values (4-byte length + bytes each; PlainByteArrayDecoder,
 * cpp/src/parquet/decoder.cc) described as 2 * count + 1 int32 offsets of alternating {length prefix,
 * value} entries (shifted by `base`), so that a var-width take of the odd entries compacts the values. */
int arx_plain_byte_array_offsets(const void* data, size_t nbytes, int64_t count, int32_t base, int32_t* out_offsets);
int arx_rle_decode_u32(const void* bytes, size_t nbytes, const ArxRleRun* runs, int64_t nruns, int bit_width,
                       int64_t num_values, uint32_t* out, void* stream);
int arx_rle_decode_equals_bitmap(const void* bytes, size_t nbytes, const ArxRleRun* runs, int64_t nruns,
                                 int bit_width, int64_t num_values, uint32_t equals, void* out_bits, void* stream);
int arx_expand_by_mask(const void* dense, int byte_width, const ArxSpan* mask, const void* ws, void* out_data,
                       void* stream);
/* LZ4 frame bodies on the device — Lz4FrameCodec::Decompress (cpp/src/arrow/util/compression_lz4.cc; lz4 is a bundled
 * third-party dependency, cpp/thirdparty/versions.txt, not vendored in the tree: frame and block formats restated from
 * its published lz4_Frame_format.md / lz4_Block_format.md) as DecompressBuffers applies it to every buffer of a
 * compressed IPC record batch (cpp/src/arrow/ipc/reader.cc).  The HOST walks each frame's header and block sizes; a
 * stream = one buffer's blocks in order (blocks of a frame may reference the previous blocks' output, so one wave
 * decodes a stream sequentially; streams are independent).  `compressed`, `streams`, `blocks`: device.  status[i]: 0 ok,
 * 1 the stream did not produce dst_size bytes, 2 a sequence runs past its block or the output, 3 a match offset
 * outside the output so far — read it back before trusting the bytes.  Asynchronous. */
typedef struct ArxLz4Block {
  uint64_t src_offset;  /* of the block's data inside `compressed` */
  uint32_t src_size;
  uint32_t stored;      /* 1: the block is not compressed (copied as it is) */
} ArxLz4Block;
typedef struct ArxLz4Stream {
  uint64_t first_block; /* index into `blocks` */
  uint32_t num_blocks;
  uint32_t reserved;
  uint64_t dst_offset;  /* where the buffer's bytes go inside `out` */
  uint64_t dst_size;    /* the buffer's uncompressed length (the 8-byte prefix of the IPC body buffer) */
} ArxLz4Stream;
/* HOST function: the blocks of the LZ4 frame at [data, data + nbytes) (blocks may be NULL to only count); src_offset =
 * byte_base + the block's position, so that the frames of many buffers can share one device copy and one table.
 * *content_size = the frame's own content size field or 0.  Checksums are skipped. */
int arx_lz4_frame_scan(const void* data, size_t nbytes, uint64_t byte_base, ArxLz4Block* blocks, int64_t max_blocks,
                       int64_t* num_blocks, uint64_t* content_size);
int arx_lz4_decompress_streams(const void* compressed, const ArxLz4Stream* streams, const ArxLz4Block* blocks,
                               int64_t num_streams, void* out, uint32_t* status, void* stream);

/* Definition levels of a flat optional column (bit width 1; LevelDecoder, cpp/src/parquet/column_reader.cc:95-190) ->
 * validity bits WITHOUT a host walk: `bytes` (device) holds the pages' level blocks where the chunk read put them,
 * pages[i] (device) = {where block i starts, its length, the page's value count, the page's first row}; one wave
 * walks the run headers of one page and the 64 lanes write the bits.  out_bits: caller-ZEROED bitmap over all rows
 * (pages share words: atomicOr).  ones[i] = number of non-null values of page i, status[i] = 0 ok / 1 corrupt block
 * (header or payload past the block, a level above 1) — read both back before trusting the bits.  Asynchronous. */
typedef struct ArxLevelPage {
  uint64_t byte_start;
  uint32_t nbytes;
  uint32_t num_values;
  uint64_t row_start;
} ArxLevelPage;
int arx_rle_levels_to_bitmap(const void* bytes, const ArxLevelPage* pages, int64_t num_pages, void* out_bits,
                             uint32_t* ones, uint32_t* status, void* stream);

/* Repeated (list) columns — DefRepLevelsToList and DefLevelsToBitmap, cpp/src/parquet/level_conversion.cc:40-124,126-146
 * and level_conversion.h:31-135 (LevelInfo: def_level, rep_level, repeated_ancestor_def_level), over level arrays the
 * hybrid decoder (arx_rle_decode_u32) left in HBM.
 *   arx_rle_scan_runs_equals   : HOST, arx_rle_scan_runs for levels of any width up to 16 bits: *count = values equal to
 *                                `equals` (the number of values a page of a nested column stores: def == max level).
 *   arx_def_rep_levels_to_list : one list level of the column.  A level slot is skipped when def < repeated_ancestor_def_
 *                                level or rep > rep_level, continues the current entry when rep == rep_level (its offset
 *                                grows by one) and starts an entry otherwise (offset grows by one when def >= def_level;
 *                                the entry is valid when def >= def_level - 1).  offsets: max_entries + 1 int32 (device),
 *                                valid_bits: caller-ZEROED bitmap of max_entries bits padded to 32-bit words, or NULL;
 *                                counts (device, 4 words): entries, elements, null entries, 1 if the levels hold more than
 *                                max_entries entries (the reference's "Definition levels exceeded upper bound") — read it
 *                                back before trusting the rest.  rep_levels may be NULL when rep_level == 0 never occurs
 *                                (every slot then starts an entry).  Asynchronous.
 *   arx_levels_ge_bitmap       : bit i = levels[i] >= threshold, every slot in place (out_bits: ceil(n / 64) words, all
 *                                written); *ones (device, caller-zeroed, may be NULL) += the set bits.  With threshold =
 *                                the leaf's def_level this is its validity over ALL level slots, with threshold = its
 *                                repeated_ancestor_def_level the slots that exist in the leaf array at all — a filter by
 *                                the second (arx_filter_*, DROP) is DefLevelsToBitmap<has_repeated_parent> plus the
 *                                compaction of the values.  Asynchronous. */
int arx_rle_scan_runs_equals(const void* data, size_t nbytes, int bit_width, int64_t num_values, uint32_t equals,
                             uint32_t out_base, uint64_t byte_base, ArxRleRun* runs, int64_t max_runs, int64_t* num_runs,
                             int64_t* count);
size_t arx_levels_to_list_workspace_bytes(int64_t num_levels);
int arx_def_rep_levels_to_list(const uint32_t* def_levels, const uint32_t* rep_levels, int64_t num_levels, int def_level,
                               int rep_level, int repeated_ancestor_def_level, int64_t max_entries, int32_t* offsets,
                               void* valid_bits, uint64_t* counts, void* ws, size_t ws_bytes, void* stream);
int arx_levels_ge_bitmap(const uint32_t* levels, int64_t num_levels, uint32_t threshold, void* out_bits, uint64_t* ones,
                         void* stream);

/* Snappy page decompression on the device — SnappyCodec::Decompress (cpp/src/arrow/util/compression_snappy.cc:42-62)
 * for the pages of a column chunk in ONE launch: `compressed` holds the raw Snappy blocks (device), pages[i] says
 * where block i sits, how many bytes it must produce and where they go in `out`; one wave decodes one page (the
 * element stream is sequential, the bytes of every literal / copy are moved by 64 lanes).  status[i] (device): 0 ok,
 * 1 bad preamble / length mismatch, 2 an element runs past the block or the output, 3 a copy offset outside the
 * output so far — the caller must read it back before trusting the bytes (a corrupt page never writes outside its
 * dst range).  Asynchronous. */
typedef struct ArxSnappyPage {
  uint64_t src_offset;  /* of the block inside `compressed` */
  uint32_t src_size;
  uint32_t dst_size;    /* the uncompressed size the page header announced */
  uint64_t dst_offset;  /* inside `out` */
} ArxSnappyPage;
int arx_snappy_decompress_pages(const void* compressed, const ArxSnappyPage* pages, int64_t num_pages, void* out,
                                uint32_t* status, void* stream);

/* GZIP page decompression on the device — GZipCodec::Decompress (cpp/src/arrow/util/compression_zlib.cc:88-180: inflateInit2
 * with window bits 15 | 32, i.e. gzip / zlib header auto-detection).  zlib is a bundled third-party dependency
 * (cpp/thirdparty/versions.txt:122 pins 1.3.1), not vendored in the tree: the formats are restated from RFC 1952 (gzip),
 * RFC 1950 (zlib) and RFC 1951 (deflate: stored, fixed and dynamic Huffman blocks).  Same page table and calling convention
 * as arx_snappy_decompress_pages (one wave decodes one page; the symbol stream is sequential, matches are copied by 64
 * lanes); checksums are skipped, a gzip member's ISIZE must equal dst_size.  status[i] (device): 0 ok, 1 bad container header
 * or the stream does not produce dst_size bytes, 2 the stream runs past its block or the output, 3 a distance before the
 * start of the output, 4 an invalid Huffman code / block type / stored length — read it back before trusting the bytes (a
 * corrupt page never writes outside its dst range).  Asynchronous. */
int arx_gzip_decompress_pages(const void* compressed, const ArxSnappyPage* pages, int64_t num_pages, void* out,
                              uint32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ARROW_AMD_H_ */
