"""Per-rank stage times of the sharded hash_sum (4e9 rows / 1e7 keys) on the RANGE-PARTITIONED state (round 6) for P = 1, 2,
4, 8 VIRTUAL ranks on ONE GPU: one rank's shard (N / P rows) goes through every local stage of
parallel._sharded_range_group_by_sum with the current kernels — sampled key range (+ its read-back), plan, consume, the merge
of the P blocks a rank receives for its partitions (its own block stands in for the peers': same size), finalize of its
partitions.  The exchange itself needs P GPUs: its bytes per rank are printed and priced at 7 xGMI links x 153 GB/s.
Wall-clock per stage with the stream synchronised at every mark (host work and read-backs included), best of 5."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
from arrow_amd import _lib, parallel
from arrow_amd.array import Array, current_stream
from arrow_amd.compute import RangeGroupBySum
dev = torch.device("cuda", 0)
lib = _lib.get_lib()
def fill(t, lo, hi, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    for b in range(0, t.numel(), 1 << 27):
        e = min(t.numel(), b + (1 << 27))
        t[b:e] = torch.randint(lo, hi, (e - b,), dtype=t.dtype, device=dev, generator=g)
GB_ROWS, GROUPS = int(os.environ.get("GB_ROWS", 4_000_000_000)), 10_000_000
XGMI_GBS = 7 * 153.0
base = None
print(f"hash_sum {GB_ROWS} rows / {GROUPS} keys on the range-partitioned state: per-rank stage ms, best of 5 (wall clock, stream synchronised at every mark)")
for world in (1, 2, 4, 8):
    n = GB_ROWS // world
    keys = torch.empty(n, dtype=torch.int32, device=dev); vals = torch.empty(n, dtype=torch.int64, device=dev)
    fill(keys, 0, GROUPS, 8); fill(vals, -2**63, 2**63 - 1, 9)
    kk = Array(amd.array.int32, n, [None, keys.view(torch.uint8)], 0, 0); vv = Array(amd.array.int64, n, [None, vals.view(torch.uint8)], 0, 0)
    best = None
    for rep in range(6):
        torch.cuda.synchronize(); t = [time.perf_counter()]
        def mark():
            torch.cuda.synchronize(); t.append(time.perf_counter())
        rng = RangeGroupBySum.sampled_key_range(kk)
        neg_lo, hi = [int(x) for x in rng.cpu().tolist()]
        plan = RangeGroupBySum.plan_for(1, -neg_lo, hi)
        st = RangeGroupBySum(plan, dev)
        mark()                                   # range + plan + state allocation
        assert st.consume(kk, vv)
        mark()                                   # consume
        parts, pb = int(plan.partitions), st.partition_bytes()
        first, mine = parallel.range_partitions_of(0, world, parts)
        if world > 1:
            got = st.state[: mine * pb // 8].repeat(world)      # what a rank receives: P blocks of its partitions
            torch.cuda.synchronize(); t[-1] = time.perf_counter()        # (building the stand-in is not a stage)
            _lib.check(lib.arx_groupby_range_merge(got.data_ptr(), got.data_ptr() + mine * pb, plan.width, mine, world - 1, mine * pb, current_stream(dev)))
            mark()                               # merge
            out = st.finalize(first, mine, blocks=got)
        else:
            mark()
            out = st.finalize()
        mark()                                   # finalize
        ms = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
        if rep and (best is None or sum(ms) < sum(best)):
            best = ms
        del st, out
    sent_mb = (parts - mine) * pb / 1e6 if world > 1 else 0.0
    exch = sent_mb / 1e3 / XGMI_GBS * 1e3
    total = sum(best) + exch
    if base is None:
        base = total
    print(f"P={world}: rows/rank={n} range+plan={best[0]:.3f} consume={best[1]:.3f} merge={best[2]:.3f} finalize={best[3]:.3f} "
          f"exchange={sent_mb:.1f} MB ~{exch:.3f} ms  total={total:.3f} ms  speed-up={base / total:.2f}x  efficiency={base / total / world:.2f}", flush=True)
    del keys, vals, kk, vv
    torch.cuda.empty_cache()
