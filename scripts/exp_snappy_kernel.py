"""arx_snappy_decompress_pages alone: incompressible pages (64 KB literals), compressible pages (short copies), page sizes."""
import os, sys, time
import numpy as np, torch, pyarrow as pa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
from arrow_amd import _lib
from arrow_amd.array import current_stream, default_device, to_device
lib, dev = _lib.get_lib(), default_device()
PAGE = np.dtype([("src_offset", "<u8"), ("src_size", "<u4"), ("dst_size", "<u4"), ("dst_offset", "<u8")])
codec = pa.Codec("snappy")
rng = np.random.default_rng(1)
total = int(os.environ.get("TOTAL_MB", 160)) << 20
for kind in ("random int64", "cumsum of small ints", "zeros"):
    for page_bytes in (160_000, 1 << 20, 16_384):
        npages = total // page_bytes
        n64 = page_bytes // 8
        if kind == "random int64":
            one = [rng.integers(-2**62, 2**62, n64).tobytes() for _ in range(8)]
        elif kind == "zeros":
            one = [bytes(page_bytes)] * 8
        else:
            one = [np.cumsum(rng.integers(-3, 4, n64)).tobytes() for _ in range(8)]
        blocks8 = [codec.compress(r).to_pybytes() for r in one]
        blocks = [blocks8[i % 8] for i in range(npages)]
        pages = np.zeros(npages, PAGE)
        so = 0
        for i, b in enumerate(blocks):
            pages[i] = (so, len(b), n64 * 8, i * n64 * 8)
            so += len(b)
        src = to_device(np.frombuffer(b"".join(blocks) + b"\0" * 8, dtype=np.uint8), dev)
        out = torch.empty(npages * n64 * 8 + 64, dtype=torch.uint8, device=dev)
        st = torch.zeros(npages, dtype=torch.int32, device=dev)
        table = to_device(pages.view(np.uint8), dev)
        stream = current_stream(dev)
        def run():
            _lib.check(lib.arx_snappy_decompress_pages(src.data_ptr(), table.data_ptr(), npages, out.data_ptr(), st.data_ptr(), stream))
        res = []
        for lds in (1, 0):
            assert lib.arx_set_option(b"snappy_lds", lds) == 0
            out.zero_()
            run(); torch.cuda.synchronize()
            assert int(st.max().item()) == 0
            got = out[: n64 * 8].cpu().numpy().tobytes()
            assert got == one[0]
            last = out[(npages - 1) * n64 * 8: npages * n64 * 8].cpu().numpy().tobytes()
            assert last == one[(npages - 1) % 8]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 5)
        lib.arx_set_option(b"snappy_lds", -1)
        print(f"{kind:22s} {npages:6d} pages x {page_bytes:8d} B ({so/1e6:7.1f} MB compressed): LDS form {res[0]:7.3f} ms = {npages*n64*8/res[0]/1e6:8.1f} GB/s of output | global form {res[1]:7.3f} ms", flush=True)
