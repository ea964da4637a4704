"""Per-rank stage times of the sharded hash_sum (4e9 rows / 1e7 keys) and the sharded array_sort_indices (2e9 uint64 rows)
for P = 1, 2, 4, 8 VIRTUAL ranks on ONE GPU: one rank's shard (N / P rows) goes through every local stage of
arrow_amd.parallel with the current kernels; what a rank would receive is emulated with its own blocks (same sizes: uniform
keys).  The exchange itself needs P GPUs (its bytes per rank are printed); everything else is what bounds the scaling.
VERDICT r3 next 5(i).  `hash_sum*` = the local pass without the local table (consume_partials, round 5).  Output: one line per (workload, P) with the stage ms (best of 3) and the predicted speed-up."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
from arrow_amd import _lib, parallel
from arrow_amd import compute as cp
from arrow_amd.array import Array, alloc, current_stream, int64, uint64
from arrow_amd.compute import GroupBySum
dev = torch.device("cuda", 0)
lib, st = _lib.get_lib(), current_stream(dev)
def ev(): return torch.cuda.Event(enable_timing=True)
def fill(t, lo, hi, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    for b in range(0, t.numel(), 1 << 27):
        e = min(t.numel(), b + (1 << 27))
        t[b:e] = torch.randint(lo, hi, (e - b,), dtype=t.dtype, device=dev, generator=g)
def best(fn, reps=3):
    out = None
    for _ in range(reps):
        r = fn()
        out = r if out is None else {k: min(out[k], r[k]) for k in r}
    return out
GB_ROWS, SORT_ROWS, GROUPS = int(os.environ.get("GB_ROWS", 4_000_000_000)), int(os.environ.get("SORT_ROWS", 2_000_000_000)), 10_000_000
base = {}
for world in (1, 2, 4, 8):
    n = GB_ROWS // world
    keys = torch.empty(n, dtype=torch.int32, device=dev); vals = torch.empty(n, dtype=torch.int64, device=dev)
    fill(keys, 0, GROUPS, 8); fill(vals, -2**63, 2**63 - 1, 9)
    kk = Array(amd.array.int32, n, [None, keys.view(torch.uint8)], 0, 0); vv = Array(amd.array.int64, n, [None, vals.view(torch.uint8)], 0, 0)
    cap = 1 << 25
    def gb(direct):
        t = [ev() for _ in range(5)]
        t[0].record()
        if not direct:
            local = GroupBySum(cap, dev)
        if world == 1:
            local.consume(kk, vv); t[1].record()
            out = local.finalize(); t[2].record(); torch.cuda.synchronize()
            return {"consume": t[0].elapsed_time(t[1]), "export": 0.0, "merge": 0.0, "finalize": t[1].elapsed_time(t[2]), "exchange_MB_per_rank": 0.0}
        if direct:      # round 5: the partials leave the aggregate as records in the owners' regions (no local table, no export)
            records, counts = parallel.consume_partials(kk, vv, cap, world); t[1].record(); t[2].record()
            assert records is not None
        else:
            local.consume(kk, vv); t[1].record()
            records, counts = parallel.export_partitioned(local, world); t[2].record()
        c = counts.cpu().tolist()
        mine = records[: c[0] * parallel.RECORD_BYTES]
        owned = GroupBySum(max(16, 2 * world * c[0] + 2), dev)
        for r in range(world):
            parallel.merge_records(owned, mine)
        t[3].record(); out = owned.finalize(); t[4].record(); torch.cuda.synchronize()
        return {"consume": t[0].elapsed_time(t[1]), "export": t[1].elapsed_time(t[2]), "merge": t[2].elapsed_time(t[3]),
                "finalize": t[3].elapsed_time(t[4]), "exchange_MB_per_rank": sum(c[1:]) * parallel.RECORD_BYTES / 1e6}
    for direct in ((False,) if world == 1 else (False, True)):
        r = best(lambda: gb(direct))
        local_ms = r["consume"] + r["export"] + r["merge"] + r["finalize"]
        xgmi_ms = (r["exchange_MB_per_rank"] / max(world - 1, 1)) / 153e3 * 1e3 if world > 1 else 0.0    # bytes to ONE peer / one link's 153 GB/s, all pairs at once
        total = local_ms + xgmi_ms
        base.setdefault("gb", total)
        print(f"hash_sum{'*' if direct else ' '} P={world}: rows/rank {n:>11d}  consume {r['consume']:7.2f}  export {r['export']:5.2f}  merge {r['merge']:5.2f}  finalize {r['finalize']:5.2f}"
              f"  exchange {r['exchange_MB_per_rank']:6.1f} MB/rank ~{xgmi_ms:5.2f} ms  => {total:7.2f} ms/rank, speed-up x{base['gb'] / total:4.2f}, efficiency {base['gb'] / total / world:4.2f}", flush=True)
    del keys, vals, kk, vv
    torch.cuda.empty_cache()
for world in (() if os.environ.get("SKIP_SORT") else (1, 2, 4, 8)):
    n = SORT_ROWS // world
    k = torch.empty(n, dtype=torch.int64, device=dev); fill(k, -2**63, 2**63 - 1, 10)
    arr = Array(uint64, n, [None, k.view(torch.uint8)], 0, 0)
    k2 = torch.empty(n if world > 1 else 1, dtype=torch.int64, device=dev)
    arr2 = Array(uint64, k2.numel(), [None, k2.view(torch.uint8)], 0, 0)
    def so():
        t = [ev() for _ in range(6)]
        if world == 1:
            t[0].record(); perm = cp.call_function("array_sort_indices", [arr], cp.ArraySortOptions("ascending", "at_end")); t[1].record(); torch.cuda.synchronize()
            return {"histogram": 0.0, "partition": 0.0, "unpack": 0.0, "local_sort": t[0].elapsed_time(t[1]), "gather": 0.0, "exchange_MB_per_rank": 0.0}
        bits, nbins = 12, 1 << 12
        span = arr.span()
        t[0].record()
        key_range = torch.zeros(2, dtype=torch.int64, device=dev)
        _lib.check(lib.arx_sort_key_range(C.byref(span), 0, _lib.SORT_ASCENDING, key_range.data_ptr(), st))
        window = _lib.ArxSortKeyWindow(0, 0, 0)
        stats = torch.zeros(nbins + 2 * world, dtype=torch.int64, device=dev)
        _lib.check(lib.arx_sort_key_histogram_window(C.byref(span), 0, _lib.SORT_ASCENDING, bits, C.byref(window), stats.data_ptr(), st))
        cum = torch.cumsum(stats[:nbins].cpu(), 0)
        t[1].record()
        split = [min(int(torch.searchsorted(cum, torch.tensor((n * p + world - 1) // world)).item()) + 1, nbins) for p in range(1, world)]
        split_arr = (C.c_uint32 * max(1, len(split)))(*split)
        ws_bytes = lib.arx_sort_indices_workspace_bytes(n) + 256
        ws = alloc(ws_bytes, dev); ws_ptr = (ws.data_ptr() + 255) & ~255
        records = torch.empty(max(n, 1) * parallel.SORT_RECORD_BYTES, dtype=torch.uint8, device=dev)
        counts = torch.zeros(world, dtype=torch.int64, device=dev); n_valid = C.c_int64(0)
        _lib.check(lib.arx_sort_partition_records_window(C.byref(span), 0, _lib.SORT_ASCENDING, _lib.NULLS_AT_END, bits, C.byref(window), split_arr, world,
                                                         ws_ptr, ws.numel() - (ws_ptr - ws.data_ptr()), records.data_ptr(), counts.data_ptr(), C.byref(n_valid), st))
        t[2].record()
        c = counts.cpu().tolist()
        # what rank 0 would receive: block 0 of every rank's shard — other ranks' shards are other random keys (a replica
        # of this rank's block would give every key `world` times: ties the local sort resolves by row, not the real load)
        blocks, sizes = [records[: c[0] * parallel.SORT_RECORD_BYTES].clone()], [c[0]]
        for other in range(1, world):
            fill(k2, -2**63, 2**63 - 1, 100 + other)
            sp2 = arr2.span()
            cnt2 = torch.zeros(world, dtype=torch.int64, device=dev); nv2 = C.c_int64(0)
            _lib.check(lib.arx_sort_partition_records_window(C.byref(sp2), 0, _lib.SORT_ASCENDING, _lib.NULLS_AT_END, bits, C.byref(window), split_arr, world,
                                                             ws_ptr, ws.numel() - (ws_ptr - ws.data_ptr()), records.data_ptr(), cnt2.data_ptr(), C.byref(nv2), st))
            c2 = cnt2.cpu().tolist()
            blocks.append(records[: c2[0] * parallel.SORT_RECORD_BYTES].clone()); sizes.append(c2[0])
        got = torch.cat(blocks)
        m = sum(sizes)
        meta = torch.tensor([[sizes[s], 0, s * n] for s in range(world)], dtype=torch.int64).to(dev)
        keys_recv = torch.empty(max(m, 1), dtype=torch.int64, device=dev); gidx = torch.empty(max(m, 1), dtype=torch.int64, device=dev); nulls = torch.empty(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(); t[2].record()
        _lib.check(lib.arx_sort_unpack_records(got.data_ptr(), m, meta.data_ptr(), world, 0, keys_recv.data_ptr(), gidx.data_ptr(), nulls.data_ptr(), st))
        t[3].record()
        karr = Array(uint64, m, [None, keys_recv.view(torch.uint8)], 0, 0)
        perm = cp.call_function("array_sort_indices", [karr], cp.ArraySortOptions("ascending", "at_end")); t[4].record()
        garr = Array(int64, m, [None, gidx.view(torch.uint8)], 0, 0)
        rows = cp.take(garr, perm, boundscheck=False); t[5].record(); torch.cuda.synchronize()
        return {"histogram": t[0].elapsed_time(t[1]), "partition": t[1].elapsed_time(t[2]) if False else 0.0, "unpack": t[2].elapsed_time(t[3]),
                "local_sort": t[3].elapsed_time(t[4]), "gather": t[4].elapsed_time(t[5]), "exchange_MB_per_rank": sum(c[1:]) * parallel.SORT_RECORD_BYTES / 1e6}
    def so_partition():      # (timed apart: the emulated receive above re-records the event behind a host synchronisation)
        bits, nbins = 12, 1 << 12
        span = arr.span(); window = _lib.ArxSortKeyWindow(0, 0, 0)
        split = [min(nbins * p // world, nbins) for p in range(1, world)]
        split_arr = (C.c_uint32 * max(1, len(split)))(*split)
        ws_bytes = lib.arx_sort_indices_workspace_bytes(n) + 256
        ws = alloc(ws_bytes, dev); ws_ptr = (ws.data_ptr() + 255) & ~255
        records = torch.empty(max(n, 1) * parallel.SORT_RECORD_BYTES, dtype=torch.uint8, device=dev)
        counts = torch.zeros(world, dtype=torch.int64, device=dev); n_valid = C.c_int64(0)
        a, b = ev(), ev(); a.record()
        _lib.check(lib.arx_sort_partition_records_window(C.byref(span), 0, _lib.SORT_ASCENDING, _lib.NULLS_AT_END, bits, C.byref(window), split_arr, world,
                                                         ws_ptr, ws.numel() - (ws_ptr - ws.data_ptr()), records.data_ptr(), counts.data_ptr(), C.byref(n_valid), st))
        b.record(); torch.cuda.synchronize()
        return {"partition": a.elapsed_time(b)}
    r = best(so, reps=2)
    if world > 1:
        r["partition"] = best(so_partition)["partition"]
    local_ms = r["histogram"] + r["partition"] + r["unpack"] + r["local_sort"] + r["gather"]
    xgmi_ms = (r["exchange_MB_per_rank"] / max(world - 1, 1)) / 153e3 * 1e3 if world > 1 else 0.0
    total = local_ms + xgmi_ms
    base.setdefault("sort", total)
    print(f"sort      P={world}: rows/rank {n:>11d}  histogram {r['histogram']:5.2f}  partition {r['partition']:6.2f}  unpack {r['unpack']:5.2f}  local_sort {r['local_sort']:6.2f}  gather {r['gather']:5.2f}"
          f"  exchange {r['exchange_MB_per_rank']:7.1f} MB/rank ~{xgmi_ms:5.2f} ms  => {total:7.2f} ms/rank, speed-up x{base['sort'] / total:4.2f}, efficiency {base['sort'] / total / world:4.2f}", flush=True)
    del k, arr, k2, arr2
    torch.cuda.empty_cache()
