"""hash_sum of 4e9 rows / 1e7 keys through `table_source_rocm -> aggregate_rocm` (bench.py's acero_hash_sum_full) for several
batch sizes of the source (arrow_amd_plugin_set_table_source_rows); ARROW_AMD_AGGREGATE_TIMING=1 prints the node's phases.
usage: exp_acero_hash_sum_full.py [log2 of rows per batch ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
plug = bench.plugin_session()
plug.lib.arrow_amd_plugin_set_table_source_rows.argtypes = [bench.ctypes.c_int64] if hasattr(bench, "ctypes") else None
import ctypes  # noqa: E402

plug.lib.arrow_amd_plugin_set_table_source_rows.argtypes = [ctypes.c_int64]
rows = int(os.environ.get("ROWS", 4_000_000_000))
for lg in [int(x) for x in sys.argv[1:]] or [27]:
    plug.lib.arrow_amd_plugin_set_table_source_rows(1 << lg)
    print("== rows per batch 2^%d" % lg, flush=True)
    print(bench.acero_hash_sum_full(dev, rows, 10_000_000, reps=2), flush=True)
