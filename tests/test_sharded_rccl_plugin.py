"""The plugin's sharded hash_sum group-by and sharded array_sort_indices (arrow_amd/csrc/plugin/sharded.inc,
sharded_sort.inc: RCCL called directly from C++, no torch).

CPU tier: world_size 2, one process per rank, the shim built against the emulated kernels and a file-based stand-in for
the RCCL entry points it resolves with dlsym (tests/emu/fake_rccl) — the real counts all-gather, the real per-peer
send / recv group, the real merge.  The union of the ranks' groups must be pyarrow's group_by of the concatenated shards,
every key owned by exactly one rank; both exchanges (partial aggregates, rows).
GPU tier (-m gpu): the same entry points on ONE GPU over the real librccl (a one-rank communicator: the all-gather and
the self send / recv go through RCCL)."""
import os
import pickle
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(r'''
    import ctypes, os, pickle, sys, time
    import numpy as np
    import pyarrow as pa
    sys.path.insert(0, ROOT)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
        from tests.emu.build_plugin_emu import build_plugin
    else:
        from arrow_amd.plugin_build import build_plugin
    lib = ctypes.CDLL(build_plugin(verbose=False))
    lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
    assert lib.arrow_amd_register() == 0, lib.arrow_amd_plugin_last_error()

    def to_device(arr):
        c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
        arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)

    def to_host(darr):
        c_dev, c_schema, c_arr, c_schema2 = (ctypes.create_string_buffer(m) for m in (128, 72, 80, 72))
        darr._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
        assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, c_schema2) == 0, lib.arrow_amd_plugin_last_error()
        return pa.Array._import_from_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema2))

    # the unique id: rank 0 makes it, the others read it from a file (any channel will do)
    id_file = OUT + ".id"
    ident = ctypes.create_string_buffer(128)
    if rank == 0:
        assert lib.arrow_amd_sharded_unique_id(ident) == 0, lib.arrow_amd_plugin_last_error()
        with open(id_file + ".tmp", "wb") as f:
            f.write(ident.raw)
        os.rename(id_file + ".tmp", id_file)
    else:
        for _ in range(6000):
            if os.path.exists(id_file):
                break
            time.sleep(0.01)
        ident = ctypes.create_string_buffer(open(id_file, "rb").read(), 128)
    comm = ctypes.c_void_p()
    assert lib.arrow_amd_sharded_comm_create(ident, world, rank, ctypes.byref(comm)) == 0, lib.arrow_amd_plugin_last_error()

    rng = np.random.default_rng(2000 + rank)
    n = N_ROWS + 777 * rank                       # ragged shards
    k = pa.array(rng.integers(-KEY_RANGE, KEY_RANGE, n).astype(np.int32), mask=rng.random(n) < 0.02)
    v = pa.array(rng.integers(-2**63, 2**63 - 1, n), mask=rng.random(n) < 0.15)
    results = {}
    lib.arrow_amd_sharded_group_by_sum.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_uint32, ctypes.c_int] + \
        [ctypes.c_void_p] * 4 + [ctypes.c_void_p]
    for exchange in (0, 1):
        for skip_nulls, min_count in ((1, 1), (0, 2)):
            dk, dv = to_device(k), to_device(v)
            bufs = [ctypes.create_string_buffer(m) for m in (128, 72, 128, 72, 128, 72, 128, 72)]
            dk._export_to_c_device(ctypes.addressof(bufs[0]), ctypes.addressof(bufs[1]))
            dv._export_to_c_device(ctypes.addressof(bufs[2]), ctypes.addressof(bufs[3]))
            stage_ms = (ctypes.c_double * 5)()
            rc = lib.arrow_amd_sharded_group_by_sum(comm, *[ctypes.addressof(b) for b in bufs[:4]], skip_nulls, min_count, exchange,
                                                    *[ctypes.addressof(b) for b in bufs[4:]], stage_ms)
            assert rc == 0, lib.arrow_amd_plugin_last_error()
            gk = pa.Array._import_from_c_device(ctypes.addressof(bufs[4]), ctypes.addressof(bufs[5]))
            gs = pa.Array._import_from_c_device(ctypes.addressof(bufs[6]), ctypes.addressof(bufs[7]))
            assert not gk.is_cpu and not gs.is_cpu
            assert sum(1 for x in stage_ms if x > 0) >= 4, list(stage_ms)
            results[(exchange, skip_nulls, min_count)] = (to_host(gk).to_pylist(), to_host(gs).to_pylist())
    # ---- shards without nulls: the local pass writes its partials straight into the owners' regions (no local table,
    # no export stage: stage_ms[1] is exactly 0) and the sends go out of the regions
    k2 = pa.array(rng.integers(-KEY_RANGE, KEY_RANGE, n).astype(np.int32))
    v2 = pa.array(rng.integers(-2**63, 2**63 - 1, n))
    lib.arx_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    if n < (1 << 17):     # (the CPU tier's shards are below the partitioned consume's row threshold)
        assert lib.arx_set_option(b"groupby_partition_min_rows", 0) == 0
        assert lib.arx_set_option(b"groupby_partition_bits", 2 + 3 * rank) == 0     # the ranks need not agree on a plan
    results2 = {}
    lib.arrow_amd_plugin_set_sharded_range_state.argtypes = [ctypes.c_int, ctypes.c_int64]
    lib.arrow_amd_plugin_sharded_range_runs.restype = ctypes.c_int64
    lib.arrow_amd_plugin_set_sharded_range_state(0, 1 << 22)      # (this block is about the records path: the range-partitioned state comes next)
    for skip_nulls, min_count in ((1, 1), (0, 3)):
        dk, dv = to_device(k2), to_device(v2)
        bufs = [ctypes.create_string_buffer(m) for m in (128, 72, 128, 72, 128, 72, 128, 72)]
        dk._export_to_c_device(ctypes.addressof(bufs[0]), ctypes.addressof(bufs[1]))
        dv._export_to_c_device(ctypes.addressof(bufs[2]), ctypes.addressof(bufs[3]))
        stage_ms = (ctypes.c_double * 5)()
        rc = lib.arrow_amd_sharded_group_by_sum(comm, *[ctypes.addressof(b) for b in bufs[:4]], skip_nulls, min_count, 0,
                                                *[ctypes.addressof(b) for b in bufs[4:]], stage_ms)
        assert rc == 0, lib.arrow_amd_plugin_last_error()
        assert stage_ms[0] > 0 and stage_ms[1] == 0.0, ("the direct local pass did not run", list(stage_ms))
        gk = pa.Array._import_from_c_device(ctypes.addressof(bufs[4]), ctypes.addressof(bufs[5]))
        gs = pa.Array._import_from_c_device(ctypes.addressof(bufs[6]), ctypes.addressof(bufs[7]))
        results2[(0, skip_nulls, min_count)] = (to_host(gk).to_pylist(), to_host(gs).to_pylist())
    if n < (1 << 17):
        lib.arx_set_option(b"groupby_partition_min_rows", 1 << 17)
        lib.arx_set_option(b"groupby_partition_bits", -1)
    # ---- round 6: the same null-free shards on the RANGE-PARTITIONED state (keys from a narrow range): the owner of a key is
    # the owner of its slice of the key range, one exchange of dense blocks of known size, groups come back in key order
    lib.arrow_amd_plugin_set_sharded_range_state(1, 0)
    if n < (1 << 22):     # (sized for small shards: two scatter workgroups, small aggregate units)
        assert lib.arx_set_option(b"groupby_lines_wgs", 2) == 0 and lib.arx_set_option(b"groupby_lines_unit_rows", 4096) == 0
    runs0 = lib.arrow_amd_plugin_sharded_range_runs()
    k3 = pa.array(rng.integers(-20 * KEY_RANGE, 20 * KEY_RANGE, n).astype(np.int32))      # (a range the state plans for: >= 3072 keys)
    v3 = pa.array(rng.integers(-2**63, 2**63 - 1, n))
    results3 = {}
    for skip_nulls, min_count in ((1, 1), (0, 3)):
        dk, dv = to_device(k3), to_device(v3)
        bufs = [ctypes.create_string_buffer(m) for m in (128, 72, 128, 72, 128, 72, 128, 72)]
        dk._export_to_c_device(ctypes.addressof(bufs[0]), ctypes.addressof(bufs[1]))
        dv._export_to_c_device(ctypes.addressof(bufs[2]), ctypes.addressof(bufs[3]))
        stage_ms = (ctypes.c_double * 5)()
        rc = lib.arrow_amd_sharded_group_by_sum(comm, *[ctypes.addressof(b) for b in bufs[:4]], skip_nulls, min_count, 0,
                                                *[ctypes.addressof(b) for b in bufs[4:]], stage_ms)
        assert rc == 0, lib.arrow_amd_plugin_last_error()
        gk = pa.Array._import_from_c_device(ctypes.addressof(bufs[4]), ctypes.addressof(bufs[5]))
        gs = pa.Array._import_from_c_device(ctypes.addressof(bufs[6]), ctypes.addressof(bufs[7]))
        got_keys = to_host(gk).to_pylist()
        assert got_keys == sorted(got_keys), "the range-partitioned state's groups come back in key order"
        results3[(0, skip_nulls, min_count)] = (got_keys, to_host(gs).to_pylist())
    assert lib.arrow_amd_plugin_sharded_range_runs() - runs0 == 2, "the range-partitioned state did not run"
    assert lib.arx_set_option(b"groupby_lines_wgs", 0) == 0 and lib.arx_set_option(b"groupby_lines_unit_rows", 1 << 21) == 0
    # ---- the sharded sort: this rank's slice of array_sort_indices of the concatenated shards
    lib.arrow_amd_sharded_sort_indices.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p]
    m = N_ROWS // 2 + 555 * rank
    sort_inputs = {
        "u64_full": pa.array(rng.integers(0, 2**64, m, dtype=np.uint64), mask=rng.random(m) < 0.03),
        "i64_window": pa.array(1_700_000_000_000_000 + rng.integers(-5000, 5000, m), mask=rng.random(m) < 0.03),   # shared top bits, many ties
        "i64_no_nulls": pa.array(rng.integers(-2**62, 2**62, m)),
        "u64_all_null": pa.array([None] * 9, pa.uint64()),
    }
    sorts = {}
    lib.arrow_amd_plugin_sharded_sort_records_runs.restype = ctypes.c_int64
    lib.arrow_amd_plugin_set_sharded_sort_sample.argtypes = [ctypes.c_int, ctypes.c_int64]
    # round 6: the null-free input once more with the key window and the splitters from a SAMPLE (1 tile of 8192 rows in 4,
    # forced at this size) — the records form either way; every other input keeps the exact form
    sort_inputs["i64_no_nulls_sampled"] = sort_inputs["i64_no_nulls"]
    for name, arr in sort_inputs.items():
        lib.arrow_amd_plugin_set_sharded_sort_sample(2 if name.endswith("_sampled") else 4, 0 if name.endswith("_sampled") else 1 << 20)
        runs0 = lib.arrow_amd_plugin_sharded_sort_records_runs()
        for descending, nulls_first, bits in ((0, 0, 12), (1, 1, 5)):
            dv = to_device(arr)
            bufs = [ctypes.create_string_buffer(sz) for sz in (128, 72, 128, 72)]
            dv._export_to_c_device(ctypes.addressof(bufs[0]), ctypes.addressof(bufs[1]))
            start = ctypes.c_int64(-1)
            stage_ms = (ctypes.c_double * 4)()
            rc = lib.arrow_amd_sharded_sort_indices(comm, ctypes.addressof(bufs[0]), ctypes.addressof(bufs[1]), descending, nulls_first,
                                                    bits, ctypes.addressof(bufs[2]), ctypes.addressof(bufs[3]), ctypes.byref(start),
                                                    stage_ms)
            assert rc == 0, lib.arrow_amd_plugin_last_error()
            idx = pa.Array._import_from_c_device(ctypes.addressof(bufs[2]), ctypes.addressof(bufs[3]))
            assert not idx.is_cpu and idx.type == pa.uint64()
            assert all(x >= 0 for x in stage_ms), list(stage_ms)
            sorts[(name, descending, nulls_first)] = (start.value, to_host(idx).to_pylist())
        assert lib.arrow_amd_plugin_sharded_sort_records_runs() - runs0 == (2 if name.startswith("i64_no_nulls") else 0), name
    lib.arrow_amd_sharded_comm_destroy(comm)
    with open(OUT + f".rank{rank}", "wb") as f:
        pickle.dump(dict(keys=k.to_pylist(), values=v.to_pylist(), results=results, keys2=k2.to_pylist(), values2=v2.to_pylist(), results2=results2, keys3=k3.to_pylist(), values3=v3.to_pylist(), results3=results3,
                         sort_inputs={name: (str(a.type), a.to_pylist()) for name, a in sort_inputs.items()}, sorts=sorts), f)
''')


def _run_ranks(tmp_path, world, env_extra, n_rows, key_range):
    out = str(tmp_path / "result")
    code = f"ROOT = {ROOT!r}\nOUT = {out!r}\nN_ROWS = {n_rows}\nKEY_RANGE = {key_range}\n" + WORKER
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), **env_extra)
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    return [pickle.load(open(out + f".rank{r}", "rb")) for r in range(world)]


def _check(ranks, which=""):
    import pyarrow as pa
    import pyarrow.compute as pc

    if which == "":
        _check(ranks, "2")      # the shards without nulls (the direct local pass)
        _check(ranks, "3")      # ... and on the range-partitioned state (round 6)
    keys = pa.array([x for r in ranks for x in r["keys" + which]], pa.int32())
    vals = pa.array([x for r in ranks for x in r["values" + which]], pa.int64())
    t = pa.table({"k": keys, "v": vals})
    for (exchange, skip_nulls, min_count) in ranks[0]["results" + which]:
        opts = pc.ScalarAggregateOptions(skip_nulls=bool(skip_nulls), min_count=min_count)
        ref = t.group_by("k", use_threads=False).aggregate([("v", "sum", opts)])
        want = dict(zip(ref.column("k").to_pylist(), ref.column("v_sum").to_pylist()))
        got = {}
        for r in ranks:
            gk, gs = r["results" + which][(exchange, skip_nulls, min_count)]
            for key, s in zip(gk, gs):
                assert key not in got, f"key {key} owned by two ranks (exchange {exchange})"
                got[key] = s
        assert got == want, (exchange, skip_nulls, min_count, len(got), len(want))
        if len(ranks) > 1:
            assert all(len(r["results" + which][(exchange, skip_nulls, min_count)][0]) > 0 for r in ranks)


def _check_sorts(ranks):
    """Concatenating the ranks' slices in rank order must give pyarrow's own array_sort_indices of the concatenated shards
    (stable: equal keys in global row order; nulls at the chosen end, in row order)."""
    import pyarrow as pa
    import pyarrow.compute as pc

    for (name, descending, nulls_first) in ranks[0]["sorts"]:
        typ = pa.uint64() if ranks[0]["sort_inputs"][name][0] == "uint64" else pa.int64()
        whole = pa.array([x for r in ranks for x in r["sort_inputs"][name][1]], typ)
        want = pc.array_sort_indices(whole, order="descending" if descending else "ascending",
                                     null_placement="at_start" if nulls_first else "at_end").to_pylist()
        got = [None] * len(whole)
        at = 0
        for r in ranks:
            start, idx = r["sorts"][(name, descending, nulls_first)]
            assert start == at, (name, "slices must tile the result in rank order", start, at)
            got[start:start + len(idx)] = idx
            at += len(idx)
        assert at == len(whole) and got == want, (name, descending, nulls_first, at, len(whole))
        if len(ranks) > 1 and name in ("u64_full", "i64_no_nulls"):
            sizes = [len(r["sorts"][(name, descending, nulls_first)][1]) for r in ranks]
            assert min(sizes) > len(whole) // (3 * len(ranks)), ("splitters should balance the ranks", sizes)


@pytest.mark.emu
def test_sharded_group_by_sum_cpp_world2_over_a_file_based_rccl_stand_in(tmp_path):
    pytest.importorskip("pyarrow")
    from tests.emu.build_plugin_emu import build_plugin

    build_plugin(verbose=False)       # once, before the ranks start (they would race on the objects)
    fake = os.path.join(ROOT, "tests", "emu", "_build", "libfake_rccl.so")
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-o", fake, os.path.join(ROOT, "tests", "emu", "fake_rccl", "fake_rccl.c")])
    ranks = _run_ranks(tmp_path, 2, dict(ARROW_AMD_PLUGIN_EMULATED="1", ARROW_AMD_RCCL_LIBRARY=fake), 6000, 300)
    _check(ranks)
    _check_sorts(ranks)


@pytest.mark.gpu
def test_sharded_group_by_sum_cpp_one_rank_over_the_real_rccl(tmp_path):
    pytest.importorskip("pyarrow")
    ranks = _run_ranks(tmp_path, 1, {}, 2_000_000, 70_000)
    _check(ranks)
    _check_sorts(ranks)


@pytest.mark.gpu
def test_sharded_group_by_and_sort_cpp_world2_on_one_gpu(tmp_path):
    """the C++ sharded group-by and sort with MORE than a self send — two processes on the one GPU
    of the box, real HIP kernels, the exchange over the file transport staged through host memory (the real librccl
    refuses two ranks on one device).  Same checks as the CPU tier's world-2 run."""
    pytest.importorskip("pyarrow")
    fake = str(tmp_path / "libfake_rccl_device.so")
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-DFAKE_RCCL_DEVICE", "-I/opt/rocm/include", "-o", fake,
                           os.path.join(ROOT, "tests", "emu", "fake_rccl", "fake_rccl.c"), "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath,/opt/rocm/lib"])
    ranks = _run_ranks(tmp_path, 2, dict(ARROW_AMD_RCCL_LIBRARY=fake), 1_000_000, 70_000)
    _check(ranks)
    _check_sorts(ranks)
