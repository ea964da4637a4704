#!/usr/bin/env python
"""bench.py — the headline benchmark of BASELINE.json, on synthetic data resident in HBM.

Default workload (N=1): BASELINE.json configs[1] — Filter + Take on a 1B-row int64 array with a
validity bitmap (10 % null), boolean mask with 10 % selectivity, FilterOptions::DROP:
    step = compute.filter(values, mask)                       # array_filter (count, scan, compact)
         + compute.get_take_indices(mask)                     # GetTakeIndices -> uint32[S]
         + compute.take(values, indices, boundscheck=False)   # the Table/RecordBatch filter path
`value` = input rows per second through one whole step (Mrows/s), all N ranks together.
Filter/take do not shard (north_star: "filter/take/cast stay single-GPU"): with --gpus N > 1
every rank runs an independent replica on its own GPU ("replicas only", weak scaling).
`--workload hash_sum` runs the sharded group-by instead (local aggregate -> partition ->
one all-to-all -> merge), rows split across ranks.

One JSON line on stdout (rank 0).  `roofline` prices the dominant kernel (the filter compaction,
arx_filter_exec) in ALGORITHMIC bytes (8N + N/8 + N/8 + 8S + S/8) over its HIP-event
duration measured live; `cpu_baseline` times the reference's own CPU kernels (pyarrow wheel =
libarrow.so.2500, kind "reference"; the C oracle as kind "port" if the wheel is absent) on a
bounded sample of the same data, one thread.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def pack_bits_device(bools: torch.Tensor) -> torch.Tensor:
    """bool[n] (n % 8 == 0) -> LSB-first bitmap bytes on the device (data generation only)."""
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=bools.device)
    return (bools.view(-1, 8).to(torch.uint8) * w).sum(dim=1, dtype=torch.uint8)


def gen_filter_inputs(n: int, device, seed: int, null_p: float, true_p: float):
    """values int64[n] (full range), validity bitmap (null_p nulls), mask bitmap (true_p set)."""
    from arrow_amd.array import alloc

    g = torch.Generator(device=device).manual_seed(seed)
    n8 = (n + 7) // 8 * 8
    values = alloc(n * 8, device)
    validity = alloc((n8 // 8 + 7) // 8 * 8, device, zero=True)
    mask = alloc((n8 // 8 + 7) // 8 * 8, device, zero=True)
    v64 = values[: n * 8].view(torch.int64)
    chunk = 1 << 26
    for b in range(0, n, chunk):
        e = min(n, b + chunk)
        v64[b:e] = torch.randint(-2**63, 2**63 - 1, (e - b,), dtype=torch.int64, device=device,
                                 generator=g)
    for b in range(0, n8, chunk):
        e = min(n8, b + chunk)
        r = torch.rand(e - b, device=device, generator=g)
        if e > n:
            r[n - b:] = 2.0
        validity[b // 8: e // 8] = pack_bits_device(r >= null_p)
        r = torch.rand(e - b, device=device, generator=g)
        if e > n:
            r[n - b:] = 2.0
        mask[b // 8: e // 8] = pack_bits_device(r < true_p)
    return values, validity, mask


def cpu_baseline_filter_take(values, validity, mask, n_total: int, sample_rows: int, budget_s: float):
    """Reference CPU kernels on the first `sample_rows` rows of the very same buffers."""
    n = min(n_total, sample_rows)
    n -= n % 64
    hv = values[: n * 8].cpu().numpy()
    hval = validity[: n // 8].cpu().numpy()
    hm = mask[: n // 8].cpu().numpy()
    try:
        import pyarrow as pa
        import pyarrow.compute as pc
    except Exception:
        pa = None
    reps, t_total = 0, 0.0
    if pa is not None:
        varr = pa.Array.from_buffers(pa.int64(), n, [pa.py_buffer(hval), pa.py_buffer(hv)], null_count=-1)
        marr = pa.Array.from_buffers(pa.bool_(), n, [None, pa.py_buffer(hm)], null_count=0)
        table = pa.table({"v": varr})
        while reps < 2 or (t_total < budget_s and reps < 8):
            t0 = time.perf_counter()
            a = pc.filter(varr, marr)          # array_filter: PrimitiveFilterExec
            b = table.filter(marr)             # FilterTable: GetTakeIndices + Take
            t_total += time.perf_counter() - t0
            reps += 1
            del a, b
        kind, what = "reference", f"pyarrow {pa.__version__} (libarrow CPU kernels), pc.filter + Table.filter"
    else:
        from oracle import oracle as O
        n = min(n, 1 << 25)
        hv64 = hv[: n * 8].view(np.int64)
        while reps < 2 or (t_total < budget_s and reps < 4):
            t0 = time.perf_counter()
            O.filter(hv64, hval, 0, hm, None, 0, n, 0, True)
            idx, _ = O.mask_to_indices(hm, None, 0, n, 0, False)
            O.take(hv64, hval, 0, idx, None, 0, len(idx), True)
            t_total += time.perf_counter() - t0
            reps += 1
        kind, what = "port", "oracle/arx_oracle.c filter + mask_to_indices + take"
    mrows = n * reps / t_total / 1e6
    return {"value": round(mrows, 2), "unit": "Mrows/s", "cores": 1, "kind": kind,
            "sample": f"first {n} rows of the same HBM buffers, {reps} reps, {what}",
            "host_cpus": os.cpu_count()}


def parity_spot_check(amd, values, validity, mask, n_total: int):
    """Outside the timed region: device filter+take on a prefix vs the C oracle."""
    from oracle import oracle as O

    n = min(n_total, 1 << 21)
    n -= n % 64
    dv = amd.Array(amd.array.int64, n, [validity, values], -1, 0)
    dm = amd.Array(amd.array.bool_, n, [None, mask], 0, 0)
    out = amd.compute.filter(dv, dm)
    idx = amd.compute.get_take_indices(dm)
    tk = amd.compute.take(dv, idx, boundscheck=False)
    torch.cuda.synchronize()
    hv = values[: n * 8].cpu().numpy().view(np.int64)
    hval = validity[: n // 8].cpu().numpy()
    hm = mask[: n // 8].cpu().numpy()
    want, want_bm = O.filter(hv, hval, 0, hm, None, 0, n, 0, True)
    got = out.data[: out.length * 8].cpu().numpy().view(np.int64)
    ok = out.length == len(want) and bool((got == want).all())
    gv = np.unpackbits(out.validity.cpu().numpy(), bitorder="little")[: out.length].astype(bool)
    ok = ok and bool((gv == O.unpack_bits(want_bm, 0, out.length)).all())
    gt = tk.data[: tk.length * 8].cpu().numpy().view(np.int64)
    tvalid = np.unpackbits(tk.validity.cpu().numpy(), bitorder="little")[: tk.length].astype(bool)
    ok = ok and tk.length == out.length and bool((tvalid == gv).all()) and bool((gt[tvalid] == want[gv]).all())
    return ok


def run_filter_take(args, rank, world, device):
    import arrow_amd as amd
    from arrow_amd import tracing

    n = args.rows
    t0 = time.time()
    values, validity, mask = gen_filter_inputs(n, device, 1234 + rank, args.null_p, args.selectivity)
    torch.cuda.synchronize(device)
    log(f"[rank {rank}] generated {n} rows in {time.time() - t0:.1f}s")
    dv = amd.Array(amd.array.int64, n, [validity, values], -1, 0)
    dm = amd.Array(amd.array.bool_, n, [None, mask], 0, 0)

    def step():
        out = amd.compute.filter(dv, dm)
        idx = amd.compute.get_take_indices(dm)
        tk = amd.compute.take(dv, idx, boundscheck=False)
        return out, idx, tk

    for _ in range(args.warmup):
        out, idx, tk = step()
    selected = out.length if args.warmup else None
    timer = tracing.KernelTimer(device)
    tracing.install(timer)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, idx, tk = step()
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    tracing.install(None)
    selected = out.length
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    filt_ms = timer.elapsed_ms("arx_filter_exec")
    take_ms = timer.elapsed_ms("arx_take")
    m2i_ms = timer.elapsed_ms("arx_mask_to_indices")
    avg_filter_ms = float(np.mean(filt_ms))
    alg_bytes = 8 * n + n / 8 + n / 8 + 8 * selected + selected / 8
    achieved = alg_bytes / (avg_filter_ms * 1e-3) / 1e9
    take_bytes = selected * 20.25
    result = {
        "metric": "filter_take_mrows_per_s",
        "value": round(world * n * args.steps / elapsed / 1e6, 2),
        "unit": "Mrows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {
            "workload": f"Filter+Take int64[{n}] + validity ({args.null_p:.0%} null), boolean mask "
                        f"{args.selectivity:.0%} true, FilterOptions::DROP; take indices = "
                        "GetTakeIndices(mask) (uint32, monotonic), no boundscheck",
            "rows": n, "selected_rows": int(selected), "parallelism": "replicas" if world > 1 else "single-gpu",
        },
        "roofline": {
            "bound": "hbm", "kernel": "compact_sparse_kernel<8> (arx_filter_exec)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_launch": int(alg_bytes),
            "avg_kernel_ms": round(avg_filter_ms, 4),
            "traffic": load_measured_traffic(n),
        },
        "kernel_ms": {"arx_filter_exec": round(avg_filter_ms, 4),
                      "arx_mask_to_indices": round(float(np.mean(m2i_ms)), 4),
                      "arx_take": round(float(np.mean(take_ms)), 4),
                      "take_algorithmic_GBps": round(take_bytes / (float(np.mean(take_ms)) * 1e-3) / 1e9, 1)},
    }
    if rank == 0 and world == 1:
        result["parity_spot_check"] = "ok" if parity_spot_check(amd, values, validity, mask, n) else "MISMATCH"
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_filter_take(values, validity, mask, n,
                                                              args.cpu_sample_rows, args.cpu_budget_s)
        if args.extras:
            del out, idx, tk
            result["other_paths"] = run_extras(amd, device)
    watchdog = None
    if args.extras and world > 1:
        # The sharded legs below are the only part of this file that depends on collectives with
        # uneven splits; should one of them wedge on a node, the headline measured above must not be
        # lost with it: after --extras-timeout seconds rank 0 prints the line it has and every rank exits.
        watchdog = _ExtrasWatchdog(rank, result, args.extras_timeout)
        watchdog.start()
    if args.extras:
        # the second half of BASELINE.json's metric: hash_sum group-by, 4B rows / 10M keys, rows
        # sharded over the N ranks (strong scaling), outside the timed region of `value`
        try:
            out = idx = tk = None
            del values, validity, mask, dv, dm
            torch.cuda.empty_cache()
            sec, rows, groups_out, checksum, ok = measure_hash_sum(rank, world, device, args.hash_sum_rows,
                                                                   args.groups, 3, 1)
            result["hash_sum"] = {"rows": rows, "groups": groups_out, "n_gpus": world,
                                  "ms": round(sec * 1e3, 3), "mrows_per_s": round(rows / sec / 1e6, 1),
                                  "scaling": "strong", "algorithmic_GBps_per_gpu": round(12 * rows / world / sec / 1e9, 1),
                                  "checksum_matches_sum_of_values": ok}
        except Exception as e:  # never lose the headline line to the secondary measurement
            result["hash_sum"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        # configs[4]: array_sort_indices of a 2B-row uint64 array sharded over the N ranks
        try:
            torch.cuda.empty_cache()
            sec, rows, ok = measure_sort(rank, world, device, args.sort_rows, 2, 1)
            result["sort_indices"] = {"rows": rows, "n_gpus": world, "ms": round(sec * 1e3, 3),
                                      "mrows_per_s": round(rows / sec / 1e6, 1), "scaling": "strong",
                                      "algorithmic_GBps_per_gpu": round(16 * rows / world / sec / 1e9, 1),
                                      "permutation_and_order_checks": ok}
        except Exception as e:
            result["sort_indices"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if watchdog is not None:
        watchdog.cancel()
    return result


class _ExtrasWatchdog:
    """Prints the headline line and ends the process if the multi-rank extras do not finish."""

    def __init__(self, rank, result, seconds):
        import threading

        self.rank, self.result, self.seconds = rank, result, seconds
        self.done = threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)

    def start(self):
        self.thread.start()

    def cancel(self):
        self.done.set()

    def _run(self):
        if self.done.wait(self.seconds):
            return
        if self.rank == 0:
            line = dict(self.result)
            for leg in ("hash_sum", "sort_indices"):
                line.setdefault(leg, {"error": f"did not finish within {self.seconds} s (watchdog)"})
            print(json.dumps(line), flush=True)
        else:
            time.sleep(2.0)   # let rank 0 print first
        os._exit(0)


def load_measured_traffic(n):
    """HBM bytes per launch from the committed PMC pass (profiles/), or null."""
    p = os.path.join(ROOT, "profiles", "filter_traffic.json")
    try:
        d = json.load(open(p))
        if int(d.get("rows", -1)) == int(n):
            return d.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def _time_gpu(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def run_extras(amd, device):
    """Informational single-GPU numbers for the other rows of SURVEY.md section 8 (not `value`)."""
    out = {}
    g = torch.Generator(device=device).manual_seed(99)
    n = 1 << 28
    x = torch.randn(n, dtype=torch.float64, device=device, generator=g)
    y = torch.randn(n, dtype=torch.float64, device=device, generator=g)
    ax = amd.Array(amd.array.float64, n, [None, x.view(torch.uint8)], 0, 0)
    ay = amd.Array(amd.array.float64, n, [None, y.view(torch.uint8)], 0, 0)
    ms = _time_gpu(lambda: amd.compute.cast(ax, amd.array.float32))
    out["cast_f64_f32"] = {"rows": n, "ms": round(ms, 4), "algorithmic_GBps": round(12 * n / ms / 1e6, 1)}
    ms = _time_gpu(lambda: amd.compute.greater(ax, ay))
    out["greater_f64"] = {"rows": n, "ms": round(ms, 4), "algorithmic_GBps": round(16.125 * n / ms / 1e6, 1)}
    del x, y, ax, ay
    n = 1 << 27
    k = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=device, generator=g)
    ak = amd.Array(amd.array.uint64, n, [None, k.view(torch.uint8)], 0, 0)
    ms = _time_gpu(lambda: amd.compute.sort_indices(ak), reps=2, warm=1)
    out["sort_indices_u64"] = {"rows": n, "ms": round(ms, 3), "mrows_per_s": round(n / ms / 1e3, 1),
                               "algorithmic_GBps": round(16 * n / ms / 1e6, 1)}
    del k, ak
    n = 1 << 28
    keys = torch.randint(0, 10_000_000, (n,), dtype=torch.int32, device=device, generator=g)
    vals = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=device, generator=g)
    kk = amd.Array(amd.array.int32, n, [None, keys.view(torch.uint8)], 0, 0)
    vv = amd.Array(amd.array.int64, n, [None, vals.view(torch.uint8)], 0, 0)
    ms = _time_gpu(lambda: amd.compute.group_by_sum(kk, vv, capacity=1 << 25), reps=2, warm=1)
    out["hash_sum_i64_by_i32"] = {"rows": n, "groups": 10_000_000, "ms": round(ms, 3),
                                  "mrows_per_s": round(n / ms / 1e3, 1),
                                  "algorithmic_GBps": round(12 * n / ms / 1e6, 1)}
    return out


def _sync(device):
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def measure_hash_sum(rank, world, device, rows_total, groups, steps, warmup):
    """Sharded group-by: each rank owns rows_total // world contiguous rows.  Returns
    (seconds per step [max over ranks], rows actually processed, groups out, checksum)."""
    import arrow_amd as amd
    from arrow_amd import parallel

    n = rows_total // world
    g = torch.Generator(device=device).manual_seed(4321 + rank)
    keys = torch.empty(n, dtype=torch.int32, device=device)
    vals = torch.empty(n, dtype=torch.int64, device=device)
    chunk = 1 << 26
    for b in range(0, n, chunk):
        e = min(n, b + chunk)
        keys[b:e] = torch.randint(0, groups, (e - b,), dtype=torch.int32, device=device, generator=g)
        vals[b:e] = torch.randint(-2**63, 2**63 - 1, (e - b,), dtype=torch.int64, device=device, generator=g)
    kk = amd.Array(amd.array.int32, n, [None, keys.view(torch.uint8)], 0, 0)
    vv = amd.Array(amd.array.int64, n, [None, vals.view(torch.uint8)], 0, 0)
    cap = 1
    while cap < 2 * groups + 2:
        cap <<= 1

    def step():
        return parallel.sharded_group_by_sum(kk, vv, cap)

    for _ in range(warmup):
        res = step()
    if world > 1:
        torch.distributed.barrier()
    _sync(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    _sync(device)
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    # parity property independent of the sharding: sum of all group sums == wrap-around sum of
    # all values; number of groups == distinct keys
    sums = res[2]
    cs = torch.stack([sums.sum(), vals.sum(), torch.tensor(sums.numel(), device=device, dtype=torch.int64)])
    if world > 1:
        torch.distributed.all_reduce(cs)
    ok = int(cs[0].item()) == int(cs[1].item())
    del keys, vals, kk, vv, res
    return elapsed / steps, n * world, int(cs[2].item()), int(cs[0].item()), ok


def measure_sort(rank, world, device, rows_total, steps, warmup):
    """Sharded array_sort_indices (configs[4]): each rank owns rows_total // world contiguous rows of
    a uint64 array; splitters -> one all-to-all -> local sort.  Returns (seconds per step [max over
    ranks], rows processed, ok) where ok = the size-independent properties that can be checked
    without a second exchange: the ranks' slices tile [0, N) and hold a permutation checksum of the
    global row numbers (sum and sum of squares mod 2^64); on one rank also full sortedness."""
    import arrow_amd as amd
    from arrow_amd import parallel

    n = rows_total // world
    g = torch.Generator(device=device).manual_seed(1010 + rank)
    keys = torch.empty(n, dtype=torch.int64, device=device)
    chunk = 1 << 26
    for b in range(0, n, chunk):
        e = min(n, b + chunk)
        keys[b:e] = torch.randint(-2**63, 2**63 - 1, (e - b,), dtype=torch.int64, device=device, generator=g)
    ak = amd.Array(amd.array.uint64, n, [None, keys.view(torch.uint8)], 0, 0)

    def step():
        return parallel.sharded_sort_indices(ak)

    for _ in range(warmup):
        rows, start = step()
    if world > 1:
        torch.distributed.barrier()
    _sync(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        rows, start = step()
    _sync(device)
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    total = n * world
    m = int(rows.numel())
    s1, s2 = 0, 0
    for b in range(0, m, chunk):          # wrap-around int64 arithmetic == mod 2^64
        r = rows[b: b + chunk]
        s1 += int(r.sum().item())
        s2 += int((r * r).sum().item())
    cs = torch.tensor([m, s1 % (1 << 63), s2 % (1 << 63)], dtype=torch.int64, device=device)
    starts = torch.zeros(world, dtype=torch.int64, device=device)
    starts[rank] = start
    lens = torch.zeros(world, dtype=torch.int64, device=device)
    lens[rank] = m
    if world > 1:
        torch.distributed.all_reduce(starts)
        torch.distributed.all_reduce(lens)
        parts = [torch.zeros_like(cs) for _ in range(world)]
        torch.distributed.all_gather(parts, cs)
    else:
        parts = [cs]
    mask64 = (1 << 64) - 1
    got_s1 = sum(int(p[1].item()) for p in parts)
    got_s2 = sum(int(p[2].item()) for p in parts)
    ok = sum(int(p[0].item()) for p in parts) == total
    # the per-rank values were reduced mod 2^63 from wrapped int64 sums: compare mod 2^63
    want_s1 = (total * (total - 1) // 2)
    want_s2 = ((total - 1) * total * (2 * total - 1) // 6)
    ok = ok and (got_s1 - want_s1) % (1 << 63) == 0 and (got_s2 - want_s2) % (1 << 63) == 0
    pos = 0
    for r_ in range(world):
        ok = ok and int(starts[r_].item()) == pos
        pos += int(lens[r_].item())
    if world == 1:
        for b in range(0, m - 1, chunk):
            e = min(m, b + chunk + 1)
            kk = keys[rows[b:e]] ^ (-2**63)
            ok = ok and bool((kk[1:] >= kk[:-1]).all())
    del keys, ak, rows
    return elapsed / steps, total, bool(ok)


def run_hash_sum(args, rank, world, device):
    sec, rows, groups_out, checksum, ok = measure_hash_sum(rank, world, device, args.rows, args.groups,
                                                           args.steps, args.warmup)
    ms = sec * 1e3
    return {
        "metric": "hash_sum_mrows_per_s", "value": round(rows / sec / 1e6, 2),
        "unit": "Mrows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": f"hash_sum(int64) GROUP BY int32, {rows} rows, {args.groups} distinct keys",
                   "rows": rows, "groups_out": groups_out,
                   "sum_of_sums_checksum": checksum, "checksum_matches_sum_of_values": ok,
                   "parallelism": f"row shards x{world} + partition + all-to-all of partials"},
        "roofline": {"bound": "hbm", "kernel": "gbp_scatter1/2 + gbp_aggregate (whole consume)",
                     "achieved": round(12 * rows / world / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(12 * rows / world / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": None},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="filter_take", choices=["filter_take", "hash_sum"])
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--groups", type=int, default=10_000_000)
    ap.add_argument("--hash-sum-rows", dest="hash_sum_rows", type=int, default=4_000_000_000)
    ap.add_argument("--sort-rows", dest="sort_rows", type=int, default=2_000_000_000)
    ap.add_argument("--extras-timeout", dest="extras_timeout", type=float, default=300.0)
    ap.add_argument("--selectivity", type=float, default=0.10)
    ap.add_argument("--null-p", dest="null_p", type=float, default=0.10)
    ap.add_argument("--cpu-sample-rows", type=int, default=250_000_000)
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", dest="extras", action="store_false")
    ap.add_argument("--option", action="append", default=[], help="name=value for arx_set_option")
    args = ap.parse_args()
    if args.rows is None:
        args.rows = 1_000_000_000 if args.workload == "filter_take" else 4_000_000_000

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=device)

    import arrow_amd as amd
    lib = amd._lib.get_lib()
    for kv in args.option:
        k, v = kv.split("=")
        assert lib.arx_set_option(k.encode(), int(v)) == 0, kv

    if args.workload == "filter_take":
        result = run_filter_take(args, rank, world, device)
    else:
        result = run_hash_sum(args, rank, world, device)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
