"""Throughput of the kernels that landed after round 1's GPU minutes were spent (never timed on a device):
bitmap append / concatenate, multi-key order_by, divide, DELTA_BINARY_PACKED / BYTE_STREAM_SPLIT decode.
Run: gpurun -- 'python scripts/exp_new_kernels.py'.  Prints one line per kernel: rows, ms, Grows/s, GB/s of
algorithmic bytes (stated per line).  Results are checked against numpy on a sample before timing."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd  # noqa: E402
from arrow_amd import compute as cp  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker only)

EMU = os.environ.get("ARROW_AMD_EXP_EMULATED") == "1"      # dry run of this script's own logic on the SIMT emulator (tiny sizes)
if EMU:
    from tests.emu.build_emu import build
    amd._lib._lib = amd._lib.load(build())
    amd.array.set_default_device("cpu")
dev = torch.device("cpu") if EMU else torch.device("cuda", 0)
SHIFT = 14 if EMU else 0
rng = np.random.default_rng(0)


def sync():
    if not EMU:
        torch.cuda.synchronize(dev)


def timed(fn, reps=5):
    reps = 1 if EMU else reps
    fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    sync()
    return (time.perf_counter() - t0) / reps * 1e3, out


def report(name, rows, ms, bytes_per_row):
    print(f"{name:46s} {rows:>12d} rows {ms:9.3f} ms {rows / ms / 1e6:8.2f} Grows/s {rows * bytes_per_row / ms / 1e6:9.1f} GB/s "
          f"({bytes_per_row} B/row algorithmic)", flush=True)


n = 1 << (27 - SHIFT)
vals = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device=dev)
valid = torch.randint(0, 256, (n // 8,), dtype=torch.uint8, device=dev)
a = amd.Array(amd.array.int64, n, [valid, vals.view(torch.uint8)], -1, 0)
chunks = [a.slice(i * (n // 8) + 3, n // 8 - 7) for i in range(8)]
ms, out = timed(lambda: cp.concat_arrays(chunks))
report("concat_arrays (8 sliced int64 chunks + validity)", out.length, ms, 16.25)

small = torch.randint(1, 1000, (n,), dtype=torch.int64, device=dev)
b = amd.Array(amd.array.int64, n, [None, small.view(torch.uint8)], 0, 0)
ms, out = timed(lambda: cp.divide(a, b))
assert torch.equal(out.data.view(torch.int64)[:1000], torch.div(vals[:1000], small[:1000], rounding_mode="trunc"))
report("divide(int64, int64), left validity", n, ms, 24.25)
ms, out = timed(lambda: cp.divide_checked(a, b))
report("divide_checked(int64, int64)", n, ms, 24.25)

m = 1 << (26 - SHIFT)
k0 = amd.Array(amd.array.int32, m, [None, torch.randint(0, 100, (m,), dtype=torch.int32, device=dev).view(torch.uint8)], 0, 0)
k1 = amd.Array(amd.array.int64, m, [None, torch.randint(0, 2**40, (m,), dtype=torch.int64, device=dev).view(torch.uint8)], 0, 0)
pay = amd.Array(amd.array.int64, m, [None, torch.arange(m, dtype=torch.int64, device=dev).view(torch.uint8)], 0, 0)
ms, cols = timed(lambda: cp.order_by([[k0], [k1], [pay]], [(0, "ascending"), (1, "descending")]), reps=3)
s0 = cols[0].data.view(torch.int32)[:m]
assert bool((s0[1:] >= s0[:-1]).all())
report("order_by 2 keys (int32 asc, int64 desc) + payload", m, ms, 20 + 20)

nd = 1 << (24 - SHIFT // 2)
walk = np.cumsum(rng.integers(-50, 60, nd)).astype(np.int64)
page = O.delta_binary_packed_encode(walk[: 1 << 16])          # (the python encoder is slow: one 64K-value page, repeated)
ms, out = timed(lambda: [amd.parquet.decode_delta_binary_packed(page, 8, dev) for _ in range(2 if EMU else 64)][-1], reps=3)
assert np.array_equal(out.to_numpy()[0], walk[: 1 << 16])
report("DELTA_BINARY_PACKED, 64 pages x 64K int64 (+ host walk)", (2 if EMU else 64) << 16, ms, round(8 + len(page) / (1 << 16), 2))

f = rng.standard_normal(nd)
split = np.ascontiguousarray(f.view(np.uint8).reshape(nd, 8).T).reshape(-1)
d_split = torch.from_numpy(split).to(dev)
outb = torch.empty(nd * 8, dtype=torch.uint8, device=dev)
lib = amd._lib.get_lib()
stream = amd.array.current_stream(dev)
ms, _ = timed(lambda: amd._lib.check(lib.arx_byte_stream_split_decode(d_split.data_ptr(), nd, 8, outb.data_ptr(), stream)))
assert np.array_equal(outb.cpu().numpy().view(np.float64), f)
report("BYTE_STREAM_SPLIT decode (double)", nd, ms, 16)
print("NEW_KERNELS_OK")
