"""ctypes binding of libarrow_amd.so (include/arrow_amd.h).

The shared library is the product: hand-written HIP kernels for gfx950 behind a C ABI.
There is no fallback of any kind — if the library is missing or no GPU is present the
calls raise.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
(or `python -m arrow_amd.build`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libarrow_amd.so")

# arrow::StatusCode twins (include/arrow_amd.h)
ARX_OK = 0
ARX_OUT_OF_MEMORY = -1
ARX_INVALID = -4
ARX_CAPACITY_ERROR = -6
ARX_INDEX_ERROR = -7
ARX_NOT_IMPLEMENTED = -10
ARX_DEVICE_ERROR = -100

ABI_VERSION = 3  # ARX_ABI_VERSION of include/arrow_amd.h

FILTER_DROP, FILTER_EMIT_NULL = 0, 1
SORT_ASCENDING, SORT_DESCENDING = 0, 1
NULLS_AT_START, NULLS_AT_END = 0, 1


class ArrowAmdError(Exception):
    """Base class; subclasses mirror pyarrow.lib.ArrowInvalid & friends."""


class ArrowInvalid(ArrowAmdError, ValueError):
    pass


class ArrowIndexError(ArrowAmdError, IndexError):
    pass


class ArrowCapacityError(ArrowAmdError):
    pass


class ArrowNotImplementedError(ArrowAmdError, NotImplementedError):
    pass


class ArrowDeviceError(ArrowAmdError, RuntimeError):
    pass


class ArrowMemoryError(ArrowAmdError, MemoryError):
    pass


_ERRORS = {
    ARX_INVALID: ArrowInvalid,
    ARX_CAPACITY_ERROR: ArrowCapacityError,
    ARX_INDEX_ERROR: ArrowIndexError,
    ARX_NOT_IMPLEMENTED: ArrowNotImplementedError,
    ARX_DEVICE_ERROR: ArrowDeviceError,
    ARX_OUT_OF_MEMORY: ArrowMemoryError,
}


class ArxSpan(C.Structure):
    """struct ArxSpan of include/arrow_amd.h (device twin of arrow::ArraySpan)."""

    _fields_ = [
        ("validity", C.c_void_p),
        ("data", C.c_void_p),
        ("offset", C.c_int64),
        ("length", C.c_int64),
        ("null_count", C.c_int64),
    ]


class ArxSortKeyWindow(C.Structure):
    """struct ArxSortKeyWindow of include/arrow_amd.h (the key range the sharded sort takes its splitter bins from)."""

    _fields_ = [("key_min", C.c_uint64), ("shift", C.c_int32), ("reserved", C.c_int32)]


class ArxRangePlan(C.Structure):
    """struct ArxRangePlan of include/arrow_amd.h (the range-partitioned group-by state, round 6)."""

    _fields_ = [("key_min", C.c_int32), ("width", C.c_int32), ("partitions", C.c_int32), ("reserved", C.c_int32),
                ("slots", C.c_int64), ("state_bytes", C.c_uint64), ("workspace_bytes", C.c_uint64)]


class ArxBinarySpan(C.Structure):
    """struct ArxBinarySpan of include/arrow_amd.h (binary / utf8 values, int32 offsets)."""

    _fields_ = [
        ("validity", C.c_void_p),
        ("offsets", C.c_void_p),
        ("data", C.c_void_p),
        ("offset", C.c_int64),
        ("length", C.c_int64),
        ("null_count", C.c_int64),
    ]


_p, _i64, _int, _u64, _u32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_uint32, C.c_size_t
_span = C.POINTER(ArxSpan)
_bspan = C.POINTER(ArxBinarySpan)

# name -> (restype, argtypes); must list every symbol declared in include/arrow_amd.h
SIGNATURES = {
    "arx_last_error": (C.c_char_p, []),
    "arx_abi_version": (_int, []),
    "arx_device_count": (_int, []),
    "arx_set_option": (_int, [C.c_char_p, _i64]),
    "arx_get_counter": (_i64, [C.c_char_p]),
    "arx_filter_workspace_bytes": (_sz, [_i64]),
    "arx_filter_count": (_int, [_span, _int, _p, _sz, C.POINTER(_i64), _p]),
    "arx_filter_count_async": (_int, [_span, _int, _p, _sz, _p]),
    "arx_filter_count_nulls": (_int, [_span, _span, _int, _p, _sz, C.POINTER(_i64), C.POINTER(_i64), _p]),
    "arx_filter_exec": (_int, [_span, _int, _span, _int, _p, _i64, _p, _p, _p]),
    "arx_mask_to_indices": (_int, [_span, _int, _p, _i64, _int, _p, _p, _p]),
    "arx_take_workspace_bytes": (_sz, []),
    "arx_check_index_bounds": (_int, [_span, _int, _u64, _p, _sz, _p]),
    "arx_take": (_int, [_span, _int, _span, _int, _p, _p, _p, _p]),
    "arx_take_rows": (_int, [_span, _i64, _span, _int, _p, _p, _p, _p]),
    "arx_take_bits": (_int, [_span, _span, _int, _p, _p, _p, _p]),
    "arx_binary_take_workspace_bytes": (_sz, [_i64]),
    "arx_binary_take_offsets": (_int, [_bspan, _span, _int, _p, _sz, _p, _p, _p, C.POINTER(_i64), _p]),
    "arx_binary_take_data": (_int, [_bspan, _i64, _p, _sz, _p, _i64, _p, _p]),
    "arx_large_binary_take_workspace_bytes": (_sz, [_i64]),
    "arx_large_binary_take_offsets": (_int, [_bspan, _span, _int, _p, _sz, _p, _p, _p, C.POINTER(_i64), _p]),
    "arx_large_binary_take_data": (_int, [_bspan, _i64, _p, _sz, _p, _i64, _p, _p]),
    "arx_list_take_data": (_int, [_bspan, _int, _i64, _p, _sz, _p, _i64, _p, _p]),
    "arx_large_list_take_data": (_int, [_bspan, _int, _i64, _p, _sz, _p, _i64, _p, _p]),
    "arx_plain_byte_array_offsets": (_int, [_p, _sz, _i64, C.c_int32, _p]),
    "arx_rle_scan_runs": (_int, [_p, _sz, _int, _i64, _u32, _u64, _p, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "arx_rle_decode_u32": (_int, [_p, _sz, _p, _i64, _int, _i64, _p, _p]),
    "arx_rle_decode_equals_bitmap": (_int, [_p, _sz, _p, _i64, _int, _i64, _u32, _p, _p]),
    "arx_expand_by_mask": (_int, [_p, _int, _span, _p, _p, _p]),
    "arx_rle_levels_to_bitmap": (_int, [_p, _p, _i64, _p, _p, _p, _p]),
    "arx_rle_scan_runs_equals": (_int, [_p, _sz, _int, _i64, _u32, _u32, _u64, _p, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "arx_levels_to_list_workspace_bytes": (_sz, [_i64]),
    "arx_def_rep_levels_to_list": (_int, [_p, _p, _i64, _int, _int, _int, _i64, _p, _p, _p, _p, _sz, _p]),
    "arx_levels_ge_bitmap": (_int, [_p, _i64, _u32, _p, _p, _p]),
    "arx_lz4_decompress_streams": (_int, [_p, _p, _p, _i64, _p, _p, _p]),
    "arx_lz4_frame_scan": (_int, [_p, _sz, _u64, _p, _i64, C.POINTER(_i64), C.POINTER(_u64)]),
    "arx_cast_f64_f32": (_int, [_p, _i64, _p, _p]),
    "arx_cast_i64_i32": (_int, [_span, _int, _p, _sz, _p, _p]),
    "arx_cast_i32_i64": (_int, [_p, _i64, _p, _p]),
    "arx_cast_i64_f64": (_int, [_span, _int, _p, _sz, _p, _p]),
    "arx_greater_f64": (_int, [_p, _p, _i64, _p, _p]),
    "arx_greater_f64_array_scalar": (_int, [_p, C.c_double, _i64, _p, _p]),
    "arx_greater_f64_scalar_array": (_int, [C.c_double, _p, _i64, _p, _p]),
    "arx_greater_i64": (_int, [_p, _p, _i64, _p, _p]),
    "arx_arith_i64": (_int, [_int, _p, _i64, _p, _i64, _i64, _p, _p]),
    "arx_arith_f64": (_int, [_int, _p, C.c_double, _p, C.c_double, _i64, _p, _p]),
    "arx_arith_checked_i64": (_int, [_int, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _p, _p, _p]),
    "arx_compare_f64": (_int, [_int, _p, C.c_double, _p, C.c_double, _i64, _p, _p]),
    "arx_compare_i64": (_int, [_int, _p, _i64, _p, _i64, _i64, _p, _p]),
    "arx_greater_i64_array_scalar": (_int, [_p, _i64, _i64, _p, _p]),
    "arx_greater_i64_scalar_array": (_int, [_i64, _p, _i64, _p, _p]),
    "arx_add_i64_array_scalar": (_int, [_p, _i64, _i64, _p, _p]),
    "arx_add_f64_array_scalar": (_int, [_p, C.c_double, _i64, _p, _p]),
    "arx_add_i64": (_int, [_p, _p, _i64, _p, _p]),
    "arx_add_f64": (_int, [_p, _p, _i64, _p, _p]),
    "arx_bitmap_copy": (_int, [_p, _i64, _i64, _p, _p]),
    "arx_buffer_copy": (_int, [_p, _p, _i64, _p]),
    "arx_delta_scan_miniblocks": (_int, [_p, _sz, _u64, _p, _i64, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64),
                                         C.POINTER(_i64), C.POINTER(_sz)]),
    "arx_delta_decode_workspace_bytes": (_sz, [_i64]),
    "arx_delta_decode": (_int, [_p, _p, _i64, _i64, _i64, _i64, _int, _p, _sz, _p, _p]),
    "arx_lengths_to_offsets_i32": (_int, [_p, _i64, C.c_int32, _p, _p, _sz, _p]),
    "arx_byte_stream_split_decode": (_int, [_p, _i64, _int, _p, _p]),
    "arx_delta_byte_array_lengths": (_int, [_p, _p, _i64, _p, _i64, _p, _p, _p]),
    "arx_delta_byte_array_expand": (_int, [_p, _p, _p, _i64, _p, C.c_int32, _p, _p, _i64, _p, _p, _p]),
    "arx_divide_i64": (_int, [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _int, _p, _p, _p]),
    "arx_divide_f64": (_int, [_p, C.c_double, _p, _i64, _p, C.c_double, _p, _i64, _i64, _int, _p, _p, _p]),
    "arx_bitmap_copy_at": (_int, [_p, _i64, _i64, _p, _i64, _p]),
    "arx_binary_rebase_offsets": (_int, [_p, _i64, C.c_int32, _p, _p]),
    "arx_bitmap_and": (_int, [_p, _i64, _p, _i64, _i64, _p, _p]),
    "arx_reduce_i64_init": (_int, [_p, _p]),
    "arx_reduce_i64_consume": (_int, [_span, _p, _p]),
    "arx_reduce_float_minmax": (_int, [_span, _int, _p, _p]),
    "arx_coalesce2": (_int, [_int, _span, _span, _p, _i64, _p, _p, _p]),
    "arx_sum_float_workspace_bytes": (_sz, [_i64, _i64]),
    "arx_sum_float": (_int, [_span, _int, _p, _sz, C.POINTER(C.c_double), C.POINTER(_i64), _p]),
    "arx_boolean_kleene": (_int, [_int, _span, _span, _p, _p, _p]),
    "arx_boolean_invert": (_int, [_p, _i64, _i64, _p, _p]),
    "arx_bitmap_popcount": (_int, [_p, _i64, _i64, _p, _sz, C.POINTER(_i64), _p]),
    "arx_bytes_to_bitmap": (_int, [_p, _i64, _p, _p, _p]),
    "arx_groupby_key_range_i32": (_int, [_span, _p, _p]),
    "arx_groupby_range_plan": (_int, [_i64, C.c_int32, C.c_int32, _p]),
    "arx_groupby_key_range_sampled_i32": (_int, [_span, _i64, _p, _p]),
    "arx_groupby_range_sum_i64_consume": (_int, [_p, _p, _span, _span, _p, _sz, _p]),
    "arx_groupby_range_merge": (_int, [_p, _p, C.c_int32, _i64, _int, _i64, _p]),
    "arx_groupby_range_finalize_workspace_bytes": (_sz, [_i64]),
    "arx_groupby_range_finalize": (_int, [_p, C.c_int32, C.c_int32, _i64, C.c_uint32, _p, _sz, _p, _p, _p, _p, _p, _p]),
    "arx_hash_bool_finalize": (_int, [_p, _p, _p, _i64, _int, _int, C.c_uint32, _p, _p, _p, _p]),
    "arx_sort_indices_workspace_bytes": (_sz, [_i64]),
    "arx_sort_partition_records_global": (_int, [_span, _int, _int, _int, _p, _p, _int, C.c_uint32, _p, _sz, _p, _p, _p]),
    "arx_sort_records": (_int, [_p, _i64, _p, _sz, _p, _p]),
    "arx_sort_indices": (_int, [_span, _int, _int, _int, _p, _sz, _p, _p]),
    "arx_rank_workspace_bytes": (_sz, [_i64]),
    "arx_rank": (_int, [_span, _int, _p, _int, _p, _sz, _p, _p]),
    "arx_sort_indices_64": (_int, [_span, _int, _int, _int, _p, _sz, _p, _p]),
    "arx_sort_key_histogram": (_int, [_span, _int, _int, _int, _p, _p]),
    "arx_sort_partition_by_bins": (_int, [_span, _int, _int, _int, _p, _int, _p, _sz, _p, _p, _p,
                                          C.POINTER(_i64), _p]),
    "arx_bitmap_to_indices": (_int, [_p, _i64, _i64, _int, _p, _sz, _p, C.POINTER(_i64), _p]),
    "arx_sort_partition_records": (_int, [_span, _int, _int, _int, _int, _p, _int, _p, _sz, _p, _p, C.POINTER(_i64), _p]),
    "arx_sort_key_range": (_int, [_span, _int, _int, _p, _p]),
    "arx_sort_key_histogram_window": (_int, [_span, _int, _int, _int, _p, _p, _p]),
    "arx_sort_key_range_sampled": (_int, [_span, _int, _int, _int, _p, _p]),
    "arx_sort_key_histogram_window_sampled": (_int, [_span, _int, _int, _int, _p, _int, _p, _p]),
    "arx_sort_partition_records_window": (_int, [_span, _int, _int, _int, _int, _p, _p, _int, _p, _sz, _p, _p,
                                                 C.POINTER(_i64), _p]),
    "arx_sort_unpack_records": (_int, [_p, _i64, _p, _int, _int, _p, _p, _p, _p]),
    "arx_groupby_export_partitioned": (_int, [_p, _int, _p, _sz, _p, _p, _p]),
    "arx_groupby_partials_capacity": (_i64, [_i64, _i64, _int]),
    "arx_groupby_sum_i64_consume_partials": (_int, [_p, _i64, _span, _span, _p, _sz, _int, _p, _i64, _p, _p]),
    "arx_groupby_partition_rows": (_int, [_span, _span, _int, _p, _sz, _p, _p, _p]),
    "arx_groupby_unpack_rows": (_int, [_p, _i64, _p, _p, _p, _p, _p]),
    "arx_hash_sum_consume_workspace_bytes": (_sz, [_i64, _i64]),
    "arx_hash_sum_i64_consume_ws": (_int, [_span, _int, _i64, _p, _i64, _i64, _p, _p, _p, _p, _sz, _p]),
    "arx_hash_mean_i64_finalize": (_int, [_p, _p, _i64, C.c_uint64, _p, _p, _p]),
    "arx_compare_numeric": (_int, [_int, _int, _p, _p, _p, _p, _i64, _p, _p]),
    "arx_arith_numeric": (_int, [_int, _int, _int, _p, _p, _p, _i64, _p, _p, _p, _i64, _i64, _p, _p, _p]),
    "arx_divide_numeric": (_int, [_int, _int, _p, _p, _p, _i64, _p, _p, _p, _i64, _i64, _p, _p, _p]),
    "arx_delta_decode_pages": (_int, [_p, _p, _p, _i64, _i64, _int, _p, _sz, _p, _p]),
    "arx_copy_segments": (_int, [_p, _i64, _u64, _p]),
    "arx_bitmap_copy_segments": (_int, [_p, _i64, _i64, _p]),
    "arx_take_columns": (_int, [_p, _p, _int, _span, _int, _p, _p, _p, _p]),
    "arx_grouper_state_bytes": (_sz, [_i64]),
    "arx_grouper_init": (_int, [_p, _i64, _p]),
    "arx_grouper_consume_workspace_bytes": (_sz, [_i64]),
    "arx_grouper_consume": (_int, [_p, _i64, _p, _p, _int, _p, _sz, _p, _p]),
    "arx_grouper_lookup": (_int, [_p, _i64, _p, _p, _int, _p, _sz, _p, _p, _p]),
    "arx_grouper_num_groups": (_int, [_p, _p, _p]),
    "arx_group_ids_skip_group": (_int, [_p, _i64, _u32, _p]),
    "arx_grouper_get_uniques": (_int, [_p, _i64, _p, _int, _int, _p, _p, _p, _p]),
    "arx_binary_key_lengths": (_int, [_bspan, _p, _p, _p, _p]),
    "arx_binary_key_chunk": (_int, [_bspan, _i64, _p, _p, _p]),
    "arx_binary_sort_chunk": (_int, [_bspan, _i64, _p, _p]),
    "arx_binary_key_hash": (_int, [_bspan, _int, _p, _p]),
    "arx_binary_key_verify": (_int, [_bspan, _p, _p, C.POINTER(_i64), _p, _p]),
    "arx_group_first_rows": (_int, [_p, _i64, _i64, _p, _p]),
    "arx_group_edge_rows": (_int, [_p, _p, _i64, _i64, _i64, _int, _p, _p, _p]),
    "arx_ree_bool_expand": (_int, [_p, _int, _i64, _span, _i64, _i64, _p, _p, _p]),
    "arx_snappy_decompress_pages": (_int, [_p, _p, _i64, _p, _p, _p]),
    "arx_gzip_decompress_pages": (_int, [_p, _p, _i64, _p, _p, _p]),
    "arx_cast_numeric": (_int, [_span, _int, _int, _int, _int, _p, _sz, _p, _p]),
    "arx_groupby_mean_i64_finalize": (_int, [_p, _p, _p, _p, _p, _i64, _int, _u32, _p, _p, _p, _p]),
    "arx_groupby_sum_i64_merge_records": (_int, [_p, _i64, _p, _i64, _p]),
    "arx_groupby_state_bytes": (_sz, [_i64]),
    "arx_groupby_init": (_int, [_p, _i64, _p]),
    "arx_groupby_consume_workspace_bytes": (_sz, [_i64, _i64]),
    "arx_groupby_sum_i64_consume": (_int, [_p, _i64, _span, _span, _p, _sz, _p]),
    "arx_groupby_sum_i64_merge": (_int, [_p, _i64, _p, _p, _p, _p, _p, _i64, _p]),
    "arx_groupby_num_groups": (_int, [_p, C.POINTER(_i64), _p]),
    "arx_groupby_sum_i64_export": (_int, [_p, _p, _p, _p, _p, _p, _p]),
    "arx_groupby_lookup_i32": (_int, [_p, _i64, _span, _p, _p]),
    "arx_groupby_minmax_bytes": (_sz, [_i64]),
    "arx_groupby_minmax_init": (_int, [_p, _i64, _p]),
    "arx_groupby_minmax_i64_consume": (_int, [_p, _p, _i64, _span, _span, _p]),
    "arx_groupby_export": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "arx_groupby_minmax_merge": (_int, [_p, _p, _i64, _p, _p, _p, _p, _p, _i64, _p]),
    "arx_groupby_minmax_finalize": (_int, [_p, _p, _p, _i64, _int, _p, _p]),
    "arx_groupby_sum_i64_finalize": (_int, [_p, _p, _i64, _int, _u32, _p, _p]),
    "arx_hash_sum_i64_consume": (_int, [_span, _int, _i64, _p, _i64, _p, _p, _p, _p]),
    "arx_hash_sum_i64_merge": (_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "arx_hash_sum_dec128_consume": (_int, [_span, _int, C.c_uint64, C.c_uint64, _p, _i64, _p, _p, _p, _p, _p]),
    "arx_hash_sum_dec128_merge": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "arx_dec128_pack": (_int, [_p, _p, _i64, _p, _p]),
    "arx_dec128_split": (_int, [_p, _i64, _p, _p, _p]),
    "arx_reduce_dec128_workspace_bytes": (_sz, []),
    "arx_reduce_dec128": (_int, [_span, _p, _sz, _p, _p]),
    "arx_hash_minmax_dec128_workspace_bytes": (_sz, [_i64]),
    "arx_hash_minmax_dec128_consume": (_int, [_span, _p, _i64, _p, _sz, _p, _p, _p, _p]),
    "arx_hash_minmax_dec128_finalize": (_int, [_p, _i64, _int, _p, _p, _p]),
    "arx_hash_sum_float_workspace_bytes": (_sz, [_i64]),
    "arx_hash_sum_float_consume": (_int, [_span, _int, _int, C.c_double, _p, _i64, _p, _sz, _p, _p, _p, _p]),
    "arx_hash_sum_f64_merge": (_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "arx_hash_product_init": (_int, [_p, _int, _i64, _p]),
    "arx_hash_product_consume": (_int, [_span, _int, _p, _i64, _p, _sz, _p, _p, _p, _p]),
    "arx_hash_mean_f64_finalize": (_int, [_p, _p, _i64, _p, _p]),
    "arx_group_central_power": (_int, [_span, _int, _p, _i64, _p, _p, _int, _p, _p]),
    "arx_hash_moments_finalize": (_int, [_p, _p, _p, _p, _i64, _int, _int, _int, _p, _p]),
    "arx_hash_minmax_i64_fill": (_int, [_p, _p, _i64, _i64, _p]),
    "arx_hash_minmax_i64_consume": (_int, [_span, _int, _i64, _p, _i64, _p, _p, _p, _p]),
    "arx_hash_minmax_i64_merge": (_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "arx_hash_minmax_i64_finalize": (_int, [_p, _p, _p, _i64, _int, _p, _p, _p]),
    "arx_hash_minmax_float_consume": (_int, [_span, _int, _int, C.c_double, _p, _i64, _p, _p, _p, _p]),
    "arx_hash_minmax_float_finalize": (_int, [_p, _p, _p, _i64, _int, _int, _p, _p, _p, _p, _p]),
    "arx_hash_count_consume": (_int, [_p, _i64, _i64, _int, _p, _i64, _p, _p]),
    "arx_hash_count_merge": (_int, [_p, _p, _p, _i64, _p]),
    "arx_hash_sum_i64_finalize": (_int, [_p, _p, _i64, _int, _u32, _p, _p, _p]),
    "arx_groupby_partition_workspace_bytes": (_sz, [_int]),
    "arx_groupby_partition": (_int, [_p, _p, _p, _p, _p, _i64, _int, _p, _sz, _p, _p, _p, _p, _p,
                                     _p, _p]),
}

_lib = None


def load(path: str | None = None):
    """dlopen the library and attach the prototypes.  Raises if it is not built."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ArrowDeviceError(
            f"{path} is not built: arrow_amd has no CPU fallback. "
            "Run `python -m arrow_amd.build` (needs hipcc, gfx950).")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.arx_abi_version() != ABI_VERSION:
        raise ArrowDeviceError("libarrow_amd.so ABI version mismatch")
    return lib


def get_lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def check(rc: int) -> None:
    """Turn an ArxStatus into the matching exception (message from arx_last_error)."""
    if rc == ARX_OK:
        return
    msg = get_lib().arx_last_error().decode("utf-8", "replace")
    raise _ERRORS.get(rc, ArrowAmdError)(msg)
