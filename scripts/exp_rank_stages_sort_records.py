"""Per-rank stage times of the sharded array_sort_indices (2e9 uint64 rows) in its RECORDS form (round 6) for P = 1, 2, 4, 8
VIRTUAL ranks on ONE GPU: one rank's shard (N / P rows) goes through the local stages of parallel.sharded_sort_indices —
key range + splitter histogram, arx_sort_partition_records_global (records with global rows, no stable pass),
arx_sort_records over as many records as the rank would receive (its own, same count: uniform keys).  The exchange needs P
GPUs: its bytes per rank are printed and priced at 7 xGMI links x 153 GB/s.  P = 1 is the single-GPU sort (arx_sort_indices).
Wall clock per stage, stream synchronised at every mark, best of 4."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
from arrow_amd import _lib
from arrow_amd.array import Array, alloc, current_stream, uint64
dev = torch.device("cuda", 0)
lib, st = _lib.get_lib(), current_stream(dev)
def fill(t, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    for b in range(0, t.numel(), 1 << 27):
        e = min(t.numel(), b + (1 << 27))
        t[b:e] = torch.randint(-2**63, 2**63 - 1, (e - b,), dtype=torch.int64, device=dev, generator=g)
ROWS = int(os.environ.get("SORT_ROWS", 2_000_000_000))
XGMI_GBS = 7 * 153.0
base = None
print(f"array_sort_indices {ROWS} uint64 rows, records form: per-rank stage ms, best of 4 (wall clock, stream synchronised at every mark)")
for world in [int(w) for w in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
    n = ROWS // world
    k = torch.empty(n, dtype=torch.int64, device=dev); fill(k, 10)
    arr = Array(uint64, n, [None, k.view(torch.uint8)], 0, 0)
    best = None
    for rep in range(5):
        torch.cuda.synchronize(); t = [time.perf_counter()]
        def mark():
            torch.cuda.synchronize(); t.append(time.perf_counter())
        if world == 1:
            out = amd.compute.sort_indices(arr); mark()
            ms = [0.0, 0.0, (t[1] - t[0]) * 1e3]
        else:
            span = arr.span()
            # (as parallel.sharded_sort_indices since the second half of round 6: range and splitter histogram from 1 tile in 16)
            SHIFT = int(os.environ.get("SAMPLE_SHIFT", 4))
            key_range = torch.zeros(4, dtype=torch.int64, device=dev)
            _lib.check(lib.arx_sort_key_range_sampled(C.byref(span), 0, _lib.SORT_ASCENDING, SHIFT, key_range.data_ptr(), st))
            kr = key_range.cpu().tolist()
            window = _lib.ArxSortKeyWindow(0, 0, 0)
            stats = torch.zeros(4096, dtype=torch.int64, device=dev)
            _lib.check(lib.arx_sort_key_histogram_window_sampled(C.byref(span), 0, _lib.SORT_ASCENDING, 12, C.byref(window), SHIFT, stats.data_ptr(), st))
            cum = torch.cumsum(stats.cpu(), 0)
            sampled_rows = int(cum[-1])
            split = [min(int(torch.searchsorted(cum, torch.tensor(sampled_rows * p // world, dtype=cum.dtype)).item()) + 1, 4096) for p in range(1, world)]
            split_arr = (C.c_uint32 * len(split))(*split)
            mark()                                    # key range + histogram + splitters
            records = torch.empty(n * 12, dtype=torch.uint8, device=dev)
            counts = torch.zeros(world, dtype=torch.int64, device=dev)
            pws = alloc(1024, dev)
            _lib.check(lib.arx_sort_partition_records_global(C.byref(span), 0, _lib.SORT_ASCENDING, 12, C.byref(window), split_arr, world, 0,
                                                             pws.data_ptr(), pws.numel(), records.data_ptr(), counts.data_ptr(), st))
            c = counts.cpu().tolist()
            mark()                                    # partition
            out = torch.empty(n, dtype=torch.int64, device=dev)
            ws = alloc(lib.arx_sort_indices_workspace_bytes(n) + 256, dev)
            ws_ptr = (ws.data_ptr() + 255) & ~255
            torch.cuda.synchronize(); t[-1] = time.perf_counter()      # (allocations are not a stage)
            _lib.check(lib.arx_sort_records(records.data_ptr(), n, ws_ptr, ws.numel() - (ws_ptr - ws.data_ptr()), out.data_ptr(), st))
            mark()                                    # local sort of the records a rank receives
            ms = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
            del records, ws
        if rep and (best is None or sum(ms) < sum(best)):
            best = ms
        del out
    sent_mb = n * 12 * (world - 1) / world / 1e6 if world > 1 else 0.0
    exch = sent_mb / 1e3 / XGMI_GBS * 1e3
    total = sum(best) + exch
    if base is None:
        base = total
    print(f"P={world}: rows/rank={n} histogram={best[0]:.3f} partition={best[1]:.3f} local_sort={best[2]:.3f} exchange={sent_mb:.0f} MB ~{exch:.3f} ms  "
          f"total={total:.3f} ms  speed-up={base / total:.2f}x  efficiency={base / total / world:.2f}", flush=True)
    del k, arr
    torch.cuda.empty_cache()
