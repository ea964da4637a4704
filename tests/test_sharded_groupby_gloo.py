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
    k = U.random_array(rng, np.int32, n, null_p=0.02, offset=rank, lo=-300, hi=300)
    v = U.random_array(rng, np.int64, n, null_p=0.15, offset=2)
    opts = arrow_amd.compute.ScalarAggregateOptions(skip_nulls=SKIP_NULLS, min_count=MIN_COUNT)
    gk, gkv, gs, gvalid = parallel.sharded_group_by_sum(k.to_device(arrow_amd), v.to_device(arrow_amd),
                                                        2048, opts)
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


@pytest.mark.parametrize("skip_nulls,min_count", [(True, 1), (False, 2)])
def test_sharded_group_by_sum_world2_gloo(tmp_path, skip_nulls, min_count):
    import pickle

    import numpy as np

    from oracle import oracle as O

    out = str(tmp_path / "result.pkl")
    code = (f"ROOT = {ROOT!r}\nOUT = {out!r}\nSKIP_NULLS = {skip_nulls!r}\nMIN_COUNT = {min_count!r}\n" + WORKER)
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
    dt = np.int64 if SIGNED else np.uint64
    a = U.random_array(rng, dt, n, null_p=0.05, offset=rank + 1)
    a.values[a.offset:a.offset + n:3] = (a.values[a.offset:a.offset + n:3] % 40).astype(dt)  # many ties across ranks
    rows, start = parallel.sharded_sort_indices(a.to_device(arrow_amd), ORDER, PLACEMENT)
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


@pytest.mark.parametrize("signed,order,placement", [(False, "ascending", "at_end"), (True, "descending", "at_start"),
                                                    (False, "descending", "at_end")])
def test_sharded_sort_indices_world2_gloo(tmp_path, signed, order, placement):
    """One-exchange multi-rank sort_indices == the oracle's stable argsort of the concatenation."""
    import pickle

    import numpy as np

    from oracle import oracle as O

    out = str(tmp_path / "sort.pkl")
    code = (f"ROOT = {ROOT!r}\nOUT = {out!r}\nSIGNED = {signed!r}\nORDER = {order!r}\nPLACEMENT = {placement!r}\n"
            + SORT_WORKER)
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
    assert min(len(r["rows"]) for r in ranks) > len(vals) // 4, "splitters should balance the ranks"


BENCH_WORKER = textwrap.dedent(r'''
    import os, sys, json
    import torch, torch.distributed as dist
    sys.path.insert(0, ROOT)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arrow_amd import _lib, array
    from tests.emu.build_emu import build
    _lib._lib = _lib.load(build())
    array.set_default_device("cpu")
    import bench
    sec, rows, groups_out, checksum, ok = bench.measure_hash_sum(rank, world, torch.device("cpu"), 40000, 700, 1, 1)
    assert ok and rows == 40000 and groups_out == 700, (ok, rows, groups_out)
    sec, srows, sok = bench.measure_sort(rank, world, torch.device("cpu"), 30000, 1, 1)
    assert sok and srows == 30000, (sok, srows)
    if rank == 0:
        print("BENCH_HASH_SUM_OK", json.dumps(dict(rows=rows, groups=groups_out, sort_rows=srows)))
    dist.barrier()
    dist.destroy_process_group()
''')


def test_bench_measure_hash_sum_world2_gloo():
    """bench.py's own multi-rank hash_sum and sort_indices legs (the code the driver's --gpus N run executes),
    world_size 2 over gloo: rows sharded, partials exchanged, checksum == sum of all values."""
    code = f"ROOT = {ROOT!r}\n" + BENCH_WORKER
    port = 33500 + (os.getpid() % 2000)
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
    assert any("BENCH_HASH_SUM_OK" in l for l in logs)
