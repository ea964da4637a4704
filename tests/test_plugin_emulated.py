"""The Arrow registration shim's DEVICE-RESIDENT paths in the GPU-less CPU tier.

The same C++ sources (arrow_amd/csrc/arrow_plugin.cc + plugin/*.inc) are built against the emulated
kernel library and a host-memory stand-in for the HIP runtime (tests/emu/plugin_hip): Arrow sees kROCM
buffers (non-CPU, `Buffer::data()` is null), the shim reads their addresses, the kernel sources run
under the SIMT emulator.  The scripts are the very ones the GPU tier runs (tests/test_gpu_arrow_plugin.py),
scaled down: kROCM memory manager + C Device Data round trips, device-aware filter / take / casts /
comparisons / arithmetic / Kleene logic / sorts / utf8, the aggregate_rocm Acero node and whole Acero
plans over device-resident tables.  Test infrastructure only — the product build is untouched."""
import os
import subprocess
import sys

import pytest

from . import test_gpu_arrow_plugin as G

pytestmark = pytest.mark.emu


def _run(script, marker, scale):
    pytest.importorskip("pyarrow")
    env = dict(os.environ, ARROW_AMD_PLUGIN_EMULATED="1", ARROW_AMD_TEST_SCALE=str(scale), ARROW_AMD_TEST_LIGHT="1")
    r = subprocess.run([sys.executable, "-c", f"ROOT = {G.ROOT!r}\n" + script], capture_output=True, text=True,
                       timeout=1500, cwd=G.ROOT, env=env)
    assert r.returncode == 0 and marker in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_device_resident_arrays_through_callfunction_emulated():
    _run(G.DEVICE_SCRIPT, "DEVICE_OK", 0.025)


def test_acero_fused_group_by_node_emulated():
    _run(G.ACERO_SCRIPT, "ACERO_OK", 0.03)


def test_acero_plan_over_a_device_resident_table_emulated():
    _run(G.ACERO_DEVICE_SCRIPT, "ACERO_DEVICE_OK", 0.1)


def test_pyarrow_compute_dispatches_to_the_hip_kernels_emulated():
    _run(G.SCRIPT, "PLUGIN_OK", 0.04)


def test_parquet_column_chunks_through_the_plugin_emulated():
    _run(G.PARQUET_SCRIPT, "PARQUET_OK", 0.03)


def test_acero_order_by_node_emulated():
    _run(G.ORDER_BY_SCRIPT, "ORDER_BY_OK", 0.02)


def test_scalar_aggregates_on_device_resident_columns_emulated():
    _run(G.AGGREGATE_SCRIPT, "AGGREGATE_OK", 0.02)


def test_parquet_delta_and_split_encodings_through_the_plugin_emulated():
    _run(G.PARQUET_ENCODINGS_SCRIPT, "PARQUET_ENCODINGS_OK", 0.02)


def test_boolean_values_filter_and_take_emulated():
    _run(G.BOOLEAN_VALUES_SCRIPT, "BOOLEAN_VALUES_OK", 0.025)


def test_single_sync_filter_path_emulated():
    _run(G.MORSEL_FILTER_SCRIPT, "MORSEL_FILTER_OK", 0.05)


def test_filter_and_take_of_device_batches_and_tables_emulated():
    _run(G.SELECTION_META_SCRIPT, "SELECTION_META_OK", 0.02)


def test_divide_on_device_resident_arrays_emulated():
    _run(G.DIVIDE_SCRIPT, "DIVIDE_OK", 0.02)


def test_reference_golden_vectors_through_callfunction_emulated():
    _run(G.GOLDEN_SCRIPT, "GOLDEN_OK", 1)


def test_compare_and_arithmetic_on_every_numeric_type_emulated():
    _run(G.NUMERIC_OPS_SCRIPT, "NUMERIC_OPS_OK", 0.01)


def test_hash_count_min_max_mean_vtables_emulated():
    _run(G.HASH_KERNELS_SCRIPT, "HASH_KERNELS_OK", 0.02)


def test_vector_hash_kernels_and_numeric_casts_emulated():
    _run(G.VECTOR_HASH_SCRIPT, "VECTOR_HASH_OK", 0.01)


def test_aggregate_rocm_general_keys_emulated():
    _run(G.GENERAL_GROUP_BY_SCRIPT, "GENERAL_GROUP_BY_OK", 0.01)


def test_aggregate_rocm_key_rows_wider_than_16_bytes_emulated():
    from . import test_gpu_group_keys as W

    _run(W.WIDE_KEYS_SCRIPT, "WIDE_KEYS_OK", 0.004)


def test_aggregate_rocm_utf8_and_binary_keys_emulated():
    from . import test_gpu_group_keys as W

    _run(W.STRING_KEYS_SCRIPT, "STRING_KEYS_OK", 0.005)


def test_device_streams_events_reader_writer_dlpack_emulated():
    _run(G.DEVICE_INTERFACES_SCRIPT, "DEVICE_INTERFACES_OK", 1)


def test_table_source_rocm_whole_chunk_batches_emulated():
    _run(G.TABLE_SOURCE_SCRIPT, "TABLE_SOURCE_OK", 0.1)     # (several 32Ki-row batches for coalesce_rocm to join)


def test_run_end_encoded_filter_masks_emulated():
    _run(G.REE_FILTER_SCRIPT, "REE_FILTER_OK", 0.02)


def test_reference_golden_grouped_aggregates_through_acero_emulated():
    _run(G.GOLDEN_HASH_AGGREGATE_SCRIPT, "GOLDEN_HASH_AGGREGATE_OK", 1)


def test_reference_golden_compare_and_arithmetic_through_callfunction_emulated():
    _run(G.GOLDEN_SCALAR_OPS_SCRIPT, "GOLDEN_SCALAR_OPS_OK", 1)


def test_reference_kernels_of_the_extended_functions_refuse_device_arrays_emulated():
    _run(G.DEVICE_GUARD_SCRIPT, "DEVICE_GUARD_OK", 1)


def test_hash_min_max_of_floats_and_temporal_types_emulated():
    _run(G.FLOAT_EXTREMA_SCRIPT, "FLOAT_EXTREMA_OK", 0.02)


def test_scalar_aggregates_of_float_boolean_and_temporal_device_columns_emulated():
    _run(G.FLOAT_AGGREGATE_SCRIPT, "FLOAT_AGGREGATE_OK", 0.02)


def test_fill_null_on_device_resident_arrays_emulated():
    _run(G.FILL_NULL_SCRIPT, "FILL_NULL_OK", 0.02)


def test_wrap_device_memory_zero_copy_and_uint64_row_numbers_emulated():
    _run(G.WRAP_SCRIPT, "WRAP_OK", 0.02)


def test_stock_acero_plans_land_on_the_plugin_nodes_after_the_factory_override_emulated():
    _run(G.ACERO_OVERRIDE_SCRIPT, "ACERO_OVERRIDE_OK", 0.05)


def test_filter_and_take_of_large_utf8_and_large_binary_on_device_arrays_emulated():
    _run(G.LARGE_BINARY_SCRIPT, "LARGE_BINARY_OK", 0.02)


def test_filter_and_take_of_fixed_size_list_and_list_on_device_arrays_emulated():
    _run(G.NESTED_SELECTION_SCRIPT, "NESTED_SELECTION_OK", 0.02)


def test_hash_sum_and_mean_of_floats_are_the_references_row_order_sums_emulated():
    _run(G.FLOAT_GROUPED_SUM_SCRIPT, "FLOAT_GROUPED_SUM_OK", 0.02)


def test_hash_count_distinct_in_aggregate_rocm_emulated():
    _run(G.COUNT_DISTINCT_SCRIPT, "COUNT_DISTINCT_OK", 0.02)


def test_hash_sum_of_decimal128_emulated():
    _run(G.DECIMAL_SUM_SCRIPT, "DECIMAL_SUM_OK", 0.02)
