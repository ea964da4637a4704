"""The emulator's two schedules (tests/emu/hip_emu_runtime.cpp).  The hardware promises no order between
workgroups, nor between the waves of one: HIPEMU_SCHEDULE=reverse runs workgroups last-to-first and resumes
fibers in descending thread order, and every kernel result has to be the same under both.  The whole CPU tier
is meant to be run once under it before a GPU call (`HIPEMU_SCHEDULE=reverse pytest -m "not gpu"`); this
file checks that the switch does what it says and keeps a few order-sensitive kernels under it permanently
(atomics on shared edge words, workgroup-level compaction, the tiled prefix sums, the hash table)."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.emu


@pytest.fixture
def reverse_schedule():
    saved = os.environ.get("HIPEMU_SCHEDULE")
    os.environ["HIPEMU_SCHEDULE"] = "reverse"   # read by the emulator at every launch
    try:
        yield
    finally:
        if saved is None:
            del os.environ["HIPEMU_SCHEDULE"]
        else:
            os.environ["HIPEMU_SCHEDULE"] = saved


def _order(nblocks, nthreads):
    from tests.emu.build_emu import build

    lib = ctypes.CDLL(build())
    out = (ctypes.c_int * (nblocks * nthreads))()
    n = lib.hipemu_selftest_order(out, nblocks, nthreads)
    assert n == nblocks * nthreads
    return list(out)


def test_default_schedule_is_ascending(monkeypatch):
    monkeypatch.delenv("HIPEMU_SCHEDULE", raising=False)
    assert _order(3, 130) == list(range(3 * 130))


def test_reverse_schedule_is_descending(reverse_schedule):
    assert _order(3, 130) == list(range(3 * 130))[::-1]


def test_filter_take_sort_groupby_under_reverse_schedule(reverse_schedule, emu_ctx):
    import pyarrow as pa
    import pyarrow.compute as pc

    amd = emu_ctx
    rng = np.random.default_rng(77)
    n = 40_000
    vals = pa.array(rng.integers(-2**62, 2**62, n), mask=rng.random(n) < 0.2)
    mask = pa.array(rng.random(n) < 0.3, mask=rng.random(n) < 0.1)
    dv, dm = amd.Array.from_pyarrow(vals), amd.Array.from_pyarrow(mask)
    assert amd.compute.filter(dv, dm).to_pyarrow().equals(pc.filter(vals, mask))
    idx = pa.array(rng.integers(0, n, 10_000).astype(np.int32), mask=rng.random(10_000) < 0.1)
    assert amd.compute.take(dv, amd.Array.from_pyarrow(idx)).to_pyarrow().equals(pc.take(vals, idx))
    keys = pa.array(rng.integers(0, 50, n).astype(np.uint64))   # many ties: stability
    assert amd.compute.sort_indices(amd.Array.from_pyarrow(keys)).to_pyarrow().equals(pc.sort_indices(keys))
    gk = rng.integers(0, 300, n).astype(np.int32)
    gv = rng.integers(-2**62, 2**62, n)
    keys_t, _, sums_t, _ = amd.compute.group_by_sum(amd.Array.from_numpy(gk), amd.Array.from_numpy(gv))
    want = pa.table({"k": gk, "v": gv}).group_by("k", use_threads=False).aggregate([("v", "sum")])
    g = want.num_rows
    got_k = keys_t.cpu().numpy().view(np.int32)[:g]
    got_s = sums_t.cpu().numpy().view(np.int64)[:g]
    # this entry point hands groups back in table order (the plugin's node re-orders them): compare by key
    wk, ws = want["k"].to_numpy(), want["v_sum"].to_numpy()
    go, wo = np.argsort(got_k, kind="stable"), np.argsort(wk, kind="stable")
    assert np.array_equal(got_k[go], wk[wo]) and np.array_equal(got_s[go], ws[wo])
