"""world_size-2 run of the sharded hash_sum group-by (arrow_amd/parallel.py) on CPU: `gloo`
backend, one process per rank, the kernel sources running under the SIMT emulator.  Checks the
N>1 data path (local aggregate -> device partition of the partials -> ONE all-to-all -> merge ->
finalize) against the oracle on the concatenated shards: the union of the ranks' results must be
the oracle's groups, each key owned by exactly one rank."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True, scope="module")
def _emulated_library_is_built_once():
    """Build the emulated kernel library in THIS process before any rank starts: two ranks
    compiling the same objects concurrently would race."""
    from tests.emu.build_emu import build

    build()

WORKER = textwrap.dedent(r'''
    import os, sys, pickle
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, ROOT)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import arrow_amd
    from arrow_amd import _lib, array, parallel
    from tests.emu.build_emu import build
    from tests import util as U
    _lib._lib = _lib.load(build())          # kernel sources under the CPU emulator (test plumbing)
    array.set_default_device("cpu")
    rng = np.random.default_rng(1000 + rank)
    n = 6000 + 500 * rank                    # ragged shards
    direct = EXCHANGE == "partials_direct"    # shards without nulls, large enough for the partitioned consume: the
    if direct:                                # local pass writes its partials straight into the owners' regions
        lib = _lib.get_lib()
        assert lib.arx_set_option(b"groupby_partition_min_rows", 0) == 0
        assert lib.arx_set_option(b"groupby_partition_bits", 3 + 2 * rank) == 0
        assert lib.arx_set_option(b"groupby_agg_chunk_rows", 1 << 12) == 0
    ranged = EXCHANGE in ("range", "range_hot")   # round 6: ids from a narrow range, no nulls -> the range-partitioned state
    if ranged:
        lib = _lib.get_lib()
        assert lib.arx_set_option(b"groupby_lines_wgs", 2) == 0
        assert lib.arx_set_option(b"groupby_lines_unit_rows", 4096) == 0
    k = U.random_array(rng, np.int32, n, null_p=0.0 if direct or ranged else 0.02, offset=rank, lo=-20000 if ranged else -300,
                       hi=30000 if ranged else 300)
    if EXCHANGE == "range_hot" and rank == 1:      # ONE rank's consume gives up on a hot key: both ranks take the table path together
        k.values[rng.random(len(k.values)) < 0.9] = 777
    v = U.random_array(rng, np.int64, n, null_p=0.0 if direct or ranged else 0.15, offset=2)
    opts = arrow_amd.compute.ScalarAggregateOptions(skip_nulls=SKIP_NULLS, min_count=MIN_COUNT)
    stages = parallel.Stages(torch.device("cpu"))
    gk, gkv, gs, gvalid = parallel.sharded_group_by_sum(k.to_device(arrow_amd), v.to_device(arrow_amd),
                                                        2048 if not ranged else 1 << 17, opts,
                                                        exchange="partials" if direct or ranged else EXCHANGE, stages=stages,
                                                        local_table=False if direct else None,
                                                        range_state=True if EXCHANGE == "range" else None)
    want_stages = {"partials": ["consume", "export", "exchange", "merge", "finalize"],     # (nulls: through the local table)
                   "partials_direct": ["consume", "exchange", "merge", "finalize"],
                   "range": ["consume", "exchange", "merge", "finalize"],
                   "range_hot": ["consume", "export", "exchange", "merge", "finalize"],      # (declined together: small shards go through the local table)
                   "rows": ["partition_rows", "exchange", "consume", "finalize"]}[EXCHANGE]
    assert list(stages.ms) == want_stages, stages.ms
    if EXCHANGE == "range":      # a rank's slice is a contiguous run of the key range, ascending
        ks = gk.numpy()
        assert (np.diff(ks) > 0).all(), "the range-partitioned state finalizes in key order"
    mine = dict(keys=gk.numpy(), key_is_valid=gkv.numpy(), sums=gs.numpy(), valid=gvalid.numpy(),
                shard=(k.values[k.offset:k.offset + n].copy(), None if k.valid is None else k.valid[k.offset:k.offset + n].copy(),
                       v.values[v.offset:v.offset + n].copy(), None if v.valid is None else v.valid[v.offset:v.offset + n].copy()))
    out = [None] * world
    dist.all_gather_object(out, mine)
    if rank == 0:
        with open(OUT, "wb") as f:
            pickle.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()
''')


@pytest.mark.parametrize("skip_nulls,min_count,exchange", [(True, 1, "partials"), (False, 2, "partials"), (False, 2, "rows"),
                                                           (True, 1, "rows"), (True, 1, "partials_direct"),
                                                           (False, 8, "partials_direct"), (True, 1, "range"), (True, 3, "range"),
                                                           (True, 1, "range_hot")])
def test_sharded_group_by_sum_world2_gloo(tmp_path, skip_nulls, min_count, exchange):
    import pickle

    import numpy as np

    from oracle import oracle as O

    out = str(tmp_path / "result.pkl")
    code = (f"ROOT = {ROOT!r}\nOUT = {out!r}\nSKIP_NULLS = {skip_nulls!r}\nMIN_COUNT = {min_count!r}\nEXCHANGE = {exchange!r}\n" + WORKER)
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    ranks = pickle.load(open(out, "rb"))

    # oracle on the concatenation of the shards
    keys = np.concatenate([r["shard"][0] for r in ranks])
    kval = np.concatenate([np.ones(len(r["shard"][0]), bool) if r["shard"][1] is None else r["shard"][1] for r in ranks])
    vals = np.concatenate([r["shard"][2] for r in ranks])
    vval = np.concatenate([np.ones(len(r["shard"][2]), bool) if r["shard"][3] is None else r["shard"][3] for r in ranks])
    w = O.groupby_sum_i64(keys, O.pack_bits(kval), 0, vals, O.pack_bits(vval), 0, len(keys), skip_nulls, min_count)
    want = {(bool(kv), int(k) if kv else 0): (int(s) if ok else None)
            for k, kv, s, ok in zip(w["keys"], w["key_is_valid"], w["sums"], w["valid"])}
    got = {}
    for r in ranks:
        for k, kv, s, ok in zip(r["keys"], r["key_is_valid"], r["sums"], r["valid"]):
            key = (bool(kv), int(k) if kv else 0)
            assert key not in got, f"key {key} owned by two ranks"
            got[key] = int(s) if ok else None
    assert got == want
    assert all(len(r["keys"]) > 0 for r in ranks), "every rank should own part of the key space"


SORT_WORKER = textwrap.dedent(r'''
    import os, sys, pickle
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, ROOT)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import arrow_amd
    from arrow_amd import _lib, array, parallel
    from tests.emu.build_emu import build
    from tests import util as U
    _lib._lib = _lib.load(build())
    array.set_default_device("cpu")
    rng = np.random.default_rng(77 + rank)
    n = 5000 + 700 * rank
    if SAMPLE:      # round 6: key range and splitter histogram from a sample of the tiles (1 tile of 8192 rows in 2^SAMPLE)
        parallel.SORT_SAMPLE_SHIFT, parallel.SORT_SAMPLE_MIN_ROWS = SAMPLE, 0
        n = 40000 + 700 * rank
    else:
        parallel.SORT_SAMPLE_SHIFT = 0
    dt = np.int64 if SIGNED else np.uint64
    a = U.random_array(rng, dt, n, null_p=0.05 if NULLS else 0.0, offset=rank + 1)
    a.values[a.offset:a.offset + n:3] = (a.values[a.offset:a.offset + n:3] % 40).astype(dt)  # many ties across ranks
    if WINDOW is not None:   # ids / timestamps: every key inside [lo, lo + span) — the top bits say nothing
        lo, span = WINDOW
        v = a.values[a.offset:a.offset + n]
        v[:] = (v.astype(np.uint64) % np.uint64(span)).astype(dt) + dt(lo)
    # (no shard has a null: round 6's records form — global rows in the records, no stable pass, the receiver sorts by (key, row))
    rows, start = parallel.sharded_sort_indices(a.to_device(arrow_amd), ORDER, PLACEMENT, records_form=None if NULLS else True)
    mine = dict(rows=rows.numpy(), start=start, vals=a.values[a.offset:a.offset + n].copy(),
                valid=None if a.valid is None else a.valid[a.offset:a.offset + n].copy())
    out = [None] * world
    dist.all_gather_object(out, mine)
    if rank == 0:
        with open(OUT, "wb") as f:
            pickle.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()
''')


@pytest.mark.parametrize("signed,order,placement,window,nulls,sample", [
    (False, "ascending", "at_end", None, True, 0), (True, "descending", "at_start", None, True, 0), (False, "descending", "at_end", None, True, 0),
    (True, "ascending", "at_end", (1_700_000_000_000_000, 86_400_000_000), True, 0),   # a day of microsecond timestamps
    (True, "descending", "at_end", (-3000, 9000), True, 0), (False, "ascending", "at_start", ((1 << 40) - 700, 1500), True, 0),
    (False, "ascending", "at_end", (12345, 1), True, 0),
    (False, "ascending", "at_end", None, False, 0), (True, "descending", "at_start", None, False, 0),
    (True, "ascending", "at_end", (1_700_000_000_000_000, 86_400_000_000), False, 0), (False, "descending", "at_end", (12345, 1), False, 0),
    # sampled window + splitters (records form): full-range keys, a narrow window, and a sample asked for while a shard has nulls
    (False, "ascending", "at_end", None, False, 1), (True, "descending", "at_start", (-3000, 9000), False, 2),
    (True, "ascending", "at_end", (1_700_000_000_000_000, 86_400_000_000), True, 1)])
def test_sharded_sort_indices_world2_gloo(tmp_path, signed, order, placement, window, nulls, sample):
    """One-exchange multi-rank sort_indices == the oracle's stable argsort of the concatenation; keys that share their
    top bits (window) must still be split between the ranks."""
    import pickle

    import numpy as np

    from oracle import oracle as O

    out = str(tmp_path / "sort.pkl")
    code = (f"ROOT = {ROOT!r}\nOUT = {out!r}\nSIGNED = {signed!r}\nORDER = {order!r}\nPLACEMENT = {placement!r}\n"
            f"WINDOW = {window!r}\nNULLS = {nulls!r}\nSAMPLE = {sample!r}\n" + SORT_WORKER)
    port = 31500 + (os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    ranks = pickle.load(open(out, "rb"))
    vals = np.concatenate([r["vals"] for r in ranks])
    valid = np.concatenate([np.ones(len(r["vals"]), bool) if r["valid"] is None else r["valid"] for r in ranks])
    want = O.sort_indices_64(np.ascontiguousarray(vals), O.pack_bits(valid), 0, len(vals),
                             descending=(order == "descending"), nulls_at_start=(placement == "at_start"))
    assert ranks[0]["start"] == 0 and ranks[1]["start"] == len(ranks[0]["rows"])
    got = np.concatenate([r["rows"] for r in ranks]).astype(np.uint64)
    assert (got == want).all()
    if window is None or window[1] > 1:   # (one distinct key cannot be split)
        assert min(len(r["rows"]) for r in ranks) > len(vals) // 4, "splitters should balance the ranks"


def test_bench_launcher_starts_two_ranks_end_to_end():
    """`python bench.py --gpus 2` with NO launcher around it (WORLD_SIZE unset): bench.py must start the two ranks
    itself, rendezvous on 127.0.0.1, run the replica headline and both sharded legs (hash_sum: one all-to-all of
    partial records; sort_indices: one all-to-all of (key, row) records) and rank 0 must print ONE JSON line with
    n_gpus = 2.  The emulated backend (CPU tensors + gloo + kernel sources under the SIMT emulator) stands in for
    HIP + RCCL: same file, same code path above the device."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "emu", "--rows", "20000",
           "--steps", "1", "--warmup", "1", "--hash-sum-rows", "40000", "--groups", "700", "--sort-rows", "30000",
           "--stream-rows", "20000", "--no-cpu-baseline"]
    # (round 6: at two ranks the sharded sort loses to one GPU and the leg DECLINES — checked first; --force-sharded-sort
    #  runs the leg for the rest of this test)
    r = subprocess.run(cmd + ["--no-extras"][:0], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    declined = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["sort_indices"]
    assert "declined" in declined and "loses to one GPU" in declined["declined"] and declined["n_gpus"] == 2, declined
    r = subprocess.run(cmd + ["--force-sharded-sort"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "replicas"
    assert line["hash_sum"]["n_gpus"] == 2 and line["hash_sum"]["rows"] == 40000 and line["hash_sum"]["groups"] == 700
    assert line["hash_sum"]["checksum_matches_sum_of_values"] is True
    assert "ONE all-to-all" in line["hash_sum"]["exchange"]
    assert line["sort_indices"]["n_gpus"] == 2 and line["sort_indices"]["rows"] == 30000
    assert line["sort_indices"]["permutation_and_order_checks"] is True
    for leg in ("hash_sum", "sort_indices"):
        assert line[leg]["roofline"]["bound"] == "hbm"


def test_bench_single_rank_line_has_every_leg():
    """N = 1 on the emulated backend: the JSON line carries roofline, the secondary configurations of SURVEY.md 8(d)
    and both sharded legs; the device streams equal their numpy twins (parity_spot_check covers that)."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "emu", "--rows", "20000", "--steps", "1",
           "--warmup", "1", "--hash-sum-rows", "20000", "--groups", "300", "--sort-rows", "10000", "--stream-rows",
           "10000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["parity_spot_check"] == "ok"
    other = line["other_paths"]
    assert other["cast_f64_f32"]["bit_exact_vs_round_to_nearest_even_sample"] is True
    for name in ("filter_drop_5pct_mask_nulls", "filter_emit_null_5pct_mask_nulls", "take_random_uint32",
                 "take_monotonic_boundscheck", "filter_drop_selectivity_25pct", "filter_drop_selectivity_50pct",
                 "take_record_batch_4_columns_one_launch", "take_record_batch_4_columns_per_column"):
        assert name in other["secondary_configs"]
    assert {"greater_f64", "sort_indices_u64_1pct_nulls", "hash_sum_1pct_null_values"} <= set(other)
    assert line["hash_sum"]["checksum_matches_sum_of_values"] and line["sort_indices"]["permutation_and_order_checks"]


def test_bench_streams_match_their_numpy_twins():
    import numpy as np
    import torch

    import bench

    for first, seed in ((0, 1), (12345, 9), (2**40 + 7, 10)):
        dev = bench.splitmix64(first, 1000, seed, torch.device("cpu")).numpy().view(np.uint64)
        assert (dev == bench.splitmix64_np(first, 1000, seed)).all()
    z = bench.splitmix64(0, 5000, 2, torch.device("cpu"))
    assert (bench.umod(z, 100).numpy() == (bench.splitmix64_np(0, 5000, 2) % np.uint64(100)).astype(np.int64)).all()
    x = bench.gen_cast_mix(20000, torch.device("cpu")).numpy()
    big = np.abs(x[np.isfinite(x)]) > 3.4028234663852886e38
    assert 0.03 < big.mean() < 0.08 and np.isnan(x).any() and np.isinf(x).any()
