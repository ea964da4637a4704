"""Key rows wider than one device Grouper table (16 bytes / 8 columns) and utf8 / binary keys: the chain of tables, and
the one-pass (length, hash) form of var-width keys with its verification and forced-collision fallback.  (Written at the
end of round 3 in a last-sorting file because it had not yet run on gfx950; green there since round 4 —
profiles/r04_c_*, r04_e_* — and named like its siblings since.  The emulator tier runs the same checks.  The two
aggregate_rocm scripts over such keys are rows of tests/plugin_scripts.py::CASES.)"""
import numpy as np
import pytest

from . import parity_cases as P


def _rng(*key):
    return np.random.default_rng([20260924, *[abs(hash(k)) % (1 << 31) for k in key]])


@pytest.mark.gpu
@pytest.mark.parametrize("dtypes,n,card,null_p,batches", [
    ((np.int64, np.int64, np.int64), 200_000, 5000, 0.1, 2), ((np.int64,) * 5, 100_000, 90_000, 0.05, 3),
    ((np.uint8,) * 20, 150_000, 400, 0.2, 2), ((np.int32, np.int64, np.int16, np.float64, np.int8, np.int64), 100_000, 30, 0.3, 1),
    ((np.int64, np.int64, np.int32), 0, 1, 0.0, 1)])
def test_grouper_rows_wider_than_one_table(gpu_ctx, dtypes, n, card, null_p, batches):
    """arrow_amd.compute.Grouper over rows of 20-40 bytes / 20 columns: ids in order of first appearance, uniques and
    Lookup equal to the oracle's GrouperImpl restatement (row/grouper.cc:695-815, :835-940)."""
    P.check_grouper(gpu_ctx, _rng("wide", len(dtypes), n, card), dtypes, n, card, null_p, batches)


@pytest.mark.gpu
def test_grouper_chain_levels_and_partial_lookups(gpu_ctx):
    P.check_grouper_chain(gpu_ctx)


@pytest.mark.gpu
def test_group_by_three_int64_keys(gpu_ctx):
    P.check_group_by_keys(gpu_ctx, _rng("wide-gb"), (np.int64, np.int64, np.int64), 300_000, 12, 0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,null_p,offset,max_len,card", [(0, 0.0, 0, 8, 1), (200_000, 0.1, 0, 30, 400), (100_000, 0.0, 5, 11, 90_000),
                                                         (50_000, 0.3, 3, 50, 7), (1000, 1.0, 0, 5, 5), (60_000, 0.1, 2, 700, 9000)])
def test_binary_key_columns_and_first_rows(gpu_ctx, n, null_p, offset, max_len, card):
    from . import util as U

    P.check_binary_key_columns(gpu_ctx, U.random_binary_pool(_rng("bkey", n, max_len), n, card, null_p, offset, max_len), _rng("bkey2", n))
