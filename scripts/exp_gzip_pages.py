"""arx_gzip_decompress_pages on the GPU: pages per launch and bytes per second, against zlib on one host core.
Usage (GPU box): python scripts/exp_gzip_pages.py > gpurun_out/<dir>/gzip_pages.txt"""
import ctypes as C
import gzip
import time
import zlib

import numpy as np
import torch

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd
from arrow_amd import _lib
from arrow_amd.array import to_device

PAGE = np.dtype([("src_offset", "<u8"), ("src_size", "<u4"), ("dst_size", "<u4"), ("dst_offset", "<u8")])
lib = _lib.get_lib()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
for kind in ("int64 walk", "doubles 1 decimal", "int32 0..49"):
    for page_bytes, npages in ((1 << 20, 64), (1 << 20, 1024), (64 << 10, 4096)):
        n = page_bytes // 8
        raws = []
        for p in range(min(npages, 64)):             # 64 distinct pages, repeated
            if kind == "int64 walk":
                raws.append(np.cumsum(rng.integers(-3, 4, n)).astype(np.int64).tobytes())
            elif kind == "doubles 1 decimal":
                raws.append(np.round(rng.standard_normal(n), 1).tobytes())
            else:
                raws.append(rng.integers(0, 50, n * 2).astype(np.int32).tobytes())
        blocks = [gzip.compress(r, 6) for r in raws]
        t0 = time.perf_counter()
        for b in blocks:
            zlib.decompress(b, 31)
        host_s = (time.perf_counter() - t0) / len(blocks)
        table = np.zeros(npages, PAGE)
        so = do = 0
        parts = []
        for i in range(npages):
            b = blocks[i % len(blocks)]
            table[i] = (so, len(b), page_bytes, do)
            so += len(b)
            do += page_bytes
            parts.append(b)
        src = to_device(np.frombuffer(b"".join(parts) + b"\0" * 8, dtype=np.uint8), dev)
        out = torch.empty(do + 64, dtype=torch.uint8, device=dev)
        st = torch.zeros(npages, dtype=torch.int32, device=dev)
        d_table = to_device(table.view(np.uint8), dev)
        stream = torch.cuda.current_stream().cuda_stream
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.check(lib.arx_gzip_decompress_pages(src.data_ptr(), d_table.data_ptr(), npages, out.data_ptr(), st.data_ptr(), stream))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        assert int(st.abs().sum().item()) == 0
        got = out[:page_bytes].cpu().numpy().tobytes()
        assert got == raws[0]
        ratio = so / do
        print(f"{kind:18s} {npages:5d} pages of {page_bytes >> 10:5d} KB (compressed to {ratio:.2f}): device {dt * 1e3:8.2f} ms = "
              f"{do / dt / 1e9:7.2f} GB/s of output ({dt / npages * 1e3 * min(npages, 1024) / min(npages, 1024):.3f} ms a page at this occupancy); "
              f"zlib on one host core {host_s * 1e3:.2f} ms a page = {page_bytes / host_s / 1e9:.2f} GB/s")
