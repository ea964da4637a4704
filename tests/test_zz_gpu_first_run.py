"""GPU tests of what landed after round 1's GPU minutes were spent: none of these has run on an MI355X yet (all
are green on the SIMT emulator and the emulated plugin, same scripts and checks): boolean-valued filter / take, the
Parquet C++ binding, the device scalar aggregates, concatenate / order_by, the DELTA_* and BYTE_STREAM_SPLIT decoders.  They live in a file that sorts
last so that a first-run failure here does not hide the suite that has been green on the device
(`pytest -x` stops at the first failure).  Fold them back into their topical files once they have passed on a GPU."""
import subprocess
import sys

import numpy as np
import pytest

from . import parity_cases as P
from . import test_gpu_arrow_plugin as G
from . import test_parquet as TP
from . import util as U
from .test_gpu_parity import rng_for

pytestmark = pytest.mark.gpu


def test_boolean_values_filter_and_take_on_device_resident_arrays():
    """Filter / take of BOOLEAN (bit-packed) device values through Arrow's CallFunction (arx_take_bits behind the
    array_filter / array_take shims), incl. sliced operands and EMIT_NULL."""
    pytest.importorskip("pyarrow")
    code = f"ROOT = {G.ROOT!r}\n" + G.BOOLEAN_VALUES_SCRIPT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=G.ROOT)
    assert r.returncode == 0 and "BOOLEAN_VALUES_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_parquet_column_chunks_through_the_plugin():
    """parquet::PageReader (headers, decompression) + the C-ABI kernels (levels,
    indices, dictionary gather, null expansion) -> device-resident arrays equal to the reference's reader."""
    pytest.importorskip("pyarrow")
    code = f"ROOT = {G.ROOT!r}\n" + G.PARQUET_SCRIPT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=G.ROOT)
    assert r.returncode == 0 and "PARQUET_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_single_sync_filter_path_on_device_resident_arrays():
    """The opt-in single-synchronisation device filter (arrow_amd_plugin_set_filter_morsel_rows): worst-case allocation,
    count -> compact back to back, one read-back — identical output to the default path and to the reference."""
    pytest.importorskip("pyarrow")
    code = f"ROOT = {G.ROOT!r}\n" + G.MORSEL_FILTER_SCRIPT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=G.ROOT)
    assert r.returncode == 0 and "MORSEL_FILTER_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_filter_and_take_of_device_resident_batches_and_tables():
    """FilterMetaFunction / TakeMetaFunction shapes (F4 / T4 of SURVEY.md 8a: record batch, table, chunked array) over
    device-resident data: per-column array_filter / array_take in HBM, equal to the reference on the host copies;
    multi-chunk device columns and uncovered device casts are refused instead of read from the CPU."""
    pytest.importorskip("pyarrow")
    code = f"ROOT = {G.ROOT!r}\n" + G.SELECTION_META_SCRIPT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=G.ROOT)
    assert r.returncode == 0 and "SELECTION_META_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_divide_on_device_resident_arrays():
    """divide / divide_checked (int64, double) on device arrays through CallFunction: values, validity, and the error the
    last failing valid slot names ("divide by zero" / "overflow"), equal to the reference; `/` in an Acero projection."""
    pytest.importorskip("pyarrow")
    code = f"ROOT = {G.ROOT!r}\n" + G.DIVIDE_SCRIPT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=G.ROOT)
    assert r.returncode == 0 and "DIVIDE_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_scalar_aggregates_on_device_resident_columns():
    """SumImpl / CountImpl / MinMaxImpl (aggregate_basic.inc.cc:49-110,776-860) as ScalarAggregateKernel shims: `sum`,
    `count`, `min_max`, `min`, `max` of int64 device columns with every option combination, chunked input (state
    merge), a refused host+device mix, and Acero's key-less `aggregate` node over a filtered device table."""
    pytest.importorskip("pyarrow")
    code = f"ROOT = {G.ROOT!r}\n" + G.AGGREGATE_SCRIPT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=G.ROOT)
    assert r.returncode == 0 and "AGGREGATE_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_acero_order_by_over_a_device_resident_table():
    """OrderByNode (acero/order_by_node.cc:100-108) as `order_by_rocm`: table_source ->
    [filter] -> order_by_rocm over device-resident and host tables, one to three sort keys with their own
    direction and null placement (int32 / int64 / float64 with NaNs / timestamp), payload columns of int64,
    utf8, boolean; equal to the stock `order_by` over the host table, with and without threads."""
    pytest.importorskip("pyarrow")
    code = f"ROOT = {G.ROOT!r}\n" + G.ORDER_BY_SCRIPT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=G.ROOT)
    assert r.returncode == 0 and "ORDER_BY_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_parquet_delta_and_split_encodings_through_the_plugin():
    """DELTA_BINARY_PACKED, DELTA_LENGTH_BYTE_ARRAY and BYTE_STREAM_SPLIT column chunks through arrow_amd_parquet_read_column
    (parquet::PageReader for the pages, the C-ABI kernels for the values), equal to the reference's reader."""
    pytest.importorskip("pyarrow")
    code = f"ROOT = {G.ROOT!r}\n" + G.PARQUET_ENCODINGS_SCRIPT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=G.ROOT)
    assert r.returncode == 0 and "PARQUET_ENCODINGS_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("idx_dtype", [np.uint8, np.int32, np.int64])
@pytest.mark.parametrize("vnull,inull,voff", [(0.0, 0.0, 0), (0.2, 0.1, 5), (1.0, 0.5, 67)])
def test_boolean_values_take_and_filter(gpu_ctx, idx_dtype, vnull, inull, voff):
    """filter / take on BOOLEAN values (1-bit Gather, gather_internal.h; PrimitiveFilter's bit-width-1 case)."""
    rng = rng_for("booltake", str(idx_dtype), vnull, inull, voff)
    nv = 200 if np.dtype(idx_dtype).itemsize == 1 else 500003
    v = U.random_mask(rng, nv, 0.5, null_p=vnull, offset=voff, tail=3)
    i = U.random_array(rng, idx_dtype, 500003, null_p=inull, offset=1, lo=0, hi=nv - 1)
    m = U.random_mask(rng, nv, 0.3, null_p=0.05, offset=2)
    P.check_boolean_take_and_filter(gpu_ctx, v, i, m)


def test_scalar_aggregates_int64(gpu_ctx):
    """SumImpl / CountImpl / MinMaxImpl (aggregate_basic.inc.cc): wrap-around sum, options, batches."""
    P.check_scalar_aggregates(gpu_ctx, rng_for("scalaragg"), n=1000003)


def test_divide_and_divide_checked(gpu_ctx):
    """Divide / DivideChecked (base_arithmetic_internal.h:366-424) through the C ABI against the oracle and pyarrow."""
    P.check_divide(gpu_ctx, rng_for("divide"), n=300_007)


# ------------------------------------------------------------------ concatenate / order_by (OrderByNode::DoFinish)
@pytest.mark.parametrize("kind", ["int64", "int8", "bool", "utf8", "binary_nonull"])
def test_concat_arrays(gpu_ctx, kind):
    """Concatenate (array/concatenate.cc): sliced chunks glued at arbitrary bit positions, empty chunks."""
    rng = rng_for("concat", kind)
    specs = [(77, 0.2, 3), (0, 0.0, 0), (1, 0.0, 0), (130_001, 0.0, 65), (64, 1.0, 7), (300_000, 0.1, 0), (5, 0.5, 1)]
    if kind == "bool":
        chunks = [U.random_mask(rng, n, 0.5, null_p=p, offset=o, tail=2) for n, p, o in specs]
    elif kind in ("utf8", "binary_nonull"):
        chunks = [U.random_binary(rng, n, null_p=0.0 if kind == "binary_nonull" else p, offset=o, tail=2, utf8=kind == "utf8")
                  for n, p, o in specs]
    else:
        chunks = [U.random_array(rng, np.dtype(kind).type, n, null_p=p, offset=o, tail=2) for n, p, o in specs]
    P.check_concat_arrays(gpu_ctx, chunks)
    P.check_concat_arrays(gpu_ctx, chunks[1:2])
    P.check_concat_arrays(gpu_ctx, [chunks[3]])


@pytest.mark.parametrize("null_placement", ["at_end", "at_start"])
def test_order_by_several_keys(gpu_ctx, null_placement):
    """OrderByNode::DoFinish (acero/order_by_node.cc:100-108) with up to three keys of mixed direction and
    per-key null placement; few distinct values per key, NaNs in the float key, payload columns ride along."""
    rng = rng_for("orderby", null_placement)
    sizes = [(32_768, 3), (0, 0), (32_768, 0), (20_001, 5)]

    def col(make):
        return [make(n, o) for n, o in sizes]

    def fkey(n, o):
        a = U.random_array(rng, np.float64, n, null_p=0.1, offset=o, tail=1)
        a.values[:] = np.round(a.values * 2) / 2
        a.values[rng.random(len(a.values)) < 0.1] = np.nan
        return a

    k0 = col(lambda n, o: U.random_array(rng, np.int32, n, null_p=0.1, offset=o, tail=1, lo=-30, hi=30))
    k1 = col(lambda n, o: U.random_array(rng, np.int64, n, null_p=0.1, offset=o, tail=1, lo=0, hi=40))
    k2 = col(fkey)
    payload = col(lambda n, o: U.random_array(rng, np.int64, n, null_p=0.2, offset=o, tail=1))
    strs = col(lambda n, o: U.random_binary(rng, n, null_p=0.1, offset=o, tail=1, utf8=True))
    flags = col(lambda n, o: U.random_mask(rng, n, 0.5, null_p=0.1, offset=o, tail=1))
    cols = [k0, k1, k2, payload, strs, flags]
    P.check_order_by(gpu_ctx, cols, [(0, "ascending"), (1, "descending")], null_placement)
    P.check_order_by(gpu_ctx, cols, [(2, "descending"), (0, "descending"), (1, "ascending")], null_placement)
    other = "at_start" if null_placement == "at_end" else "at_end"
    P.check_order_by(gpu_ctx, cols, [(0, "descending"), (2, "ascending")], [null_placement, other])


@pytest.mark.parametrize("null_p", [0.0, 0.1])
def test_parquet_byte_stream_split_gpu(gpu_ctx, tmp_path, null_p):
    TP._write_split_and_check(gpu_ctx, str(tmp_path), 600_000, null_p, 79, compression="snappy")


@pytest.mark.parametrize("null_p", [0.0, 0.1])
def test_parquet_delta_length_byte_array_gpu(gpu_ctx, tmp_path, null_p):
    TP._write_delta_length_and_check(gpu_ctx, str(tmp_path), 300_000, null_p, 83, compression="snappy")


def test_delta_decode_kernel_vs_restatement_gpu(gpu_ctx):
    P.check_delta_decode(gpu_ctx, np.random.default_rng(5), 128, 4)
    P.check_delta_decode(gpu_ctx, np.random.default_rng(6), 512, 2)


@pytest.mark.parametrize("null_p", [0.0, 0.1])
def test_parquet_delta_binary_packed_gpu(gpu_ctx, tmp_path, null_p):
    TP._write_delta_and_check(gpu_ctx, str(tmp_path), 600_000, null_p, 77, compression="snappy")
    TP._write_delta_and_check(gpu_ctx, str(tmp_path), 70_001, null_p, 78, data_page_version="2.0", data_page_size=8192)
