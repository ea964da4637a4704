"""Random-shape soak of the wide sort and the wide group-by on the EMULATED tier (no GPU): random row counts around tile
boundaries (8192 / 16384 / 24576 rows), key types and distributions (full range, narrow window, few distinct keys, sorted,
blocky), null rates, both orders; random values of the round-3 knobs (tile rows per thread at either level, partition bits,
bucket finish form, sub-bucket counters, sampled / exact sizes, rooms) — against numpy's stable argsort and a dictionary
group-by.  Usage: soak_sort_groupby_emulated.py <first seed> <trials>."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import arrow_amd as amd
from arrow_amd import _lib, array
from tests import parity_cases as P
from tests import util as U
from tests.emu.build_emu import build

_lib._lib = _lib.load(build())
array.set_default_device("cpu")
lib = _lib.get_lib()
seed0, trials = int(sys.argv[1]), int(sys.argv[2])
SORT_DEFAULTS = {b"sort_msd": -1, b"sort_msd_segment_rows": 1 << 27, b"sort_msd_wide": 1, b"sort_msd_wide_rpt1": 24,
                 b"sort_msd_wide_rpt2": 16, b"sort_msd_wide_bits": 0, b"sort_msd_wide_b2max": 11, b"sort_msd_tiny_bucket": 2,
                 b"sort_msd_bucket_cpt": 4, b"sort_msd_wide_sample_shift": 4, b"sort_msd_wide_gap2": 1}
GB_DEFAULTS = {b"groupby_partition_min_rows": 1 << 17, b"groupby_wide_max_bits": 11, b"groupby_wide_rooms": 1,
               b"groupby_wide_room_min_mean": 1 << 14, b"groupby_probe_rows": 1 << 25}


def set_all(opts):
    for k, v in opts.items():
        assert lib.arx_set_option(k, int(v)) == 0, k


for trial in range(trials):
    rng = np.random.default_rng(seed0 + trial)
    # ---- sort
    n = int(rng.choice([rng.integers(4097, 9000), rng.integers(16000, 17000), rng.integers(24000, 26000), rng.integers(40000, 120000)]))
    dtype = rng.choice([np.uint64, np.int64])
    kind = rng.choice(["full", "window", "few", "sorted", "blocky"])
    arr = U.random_array(rng, dtype, n, null_p=float(rng.choice([0, 0.02, 0.3])), offset=int(rng.integers(0, 9)))
    v = arr.values[arr.offset:arr.offset + n]
    if kind == "window":
        v[:] = (v % np.array(1 << 20, dtype)) + np.array(1_700_000_000_000, dtype)
    elif kind == "few":
        v[:] = v % np.array(int(rng.integers(1, 50)), dtype)
    elif kind == "sorted":
        v[:] = np.sort(v)
    elif kind == "blocky":
        v[(np.arange(n) // 8192) % 3 == 0] >>= np.array(5, dtype)
    opts = dict(SORT_DEFAULTS)
    opts.update({b"sort_msd": 1, b"sort_msd_segment_rows": 4096,
                 b"sort_msd_wide_rpt1": int(rng.choice([8, 16, 24])), b"sort_msd_wide_rpt2": int(rng.choice([8, 16, 24])),
                 b"sort_msd_wide_bits": int(rng.choice([0, 0, 6, 9, 13])), b"sort_msd_wide_b2max": int(rng.choice([0, 10, 11, 12])),
                 b"sort_msd_tiny_bucket": int(rng.integers(0, 3)), b"sort_msd_bucket_cpt": int(rng.choice([4, 8])),
                 b"sort_msd_wide_sample_shift": int(rng.choice([0, 2, 4])), b"sort_msd_wide_gap2": int(rng.integers(0, 2))})
    set_all(opts)
    try:
        P.check_sort_indices(amd, arr, str(rng.choice(["ascending", "descending"])), str(rng.choice(["at_end", "at_start"])),
                             use_pyarrow=False)
    except Exception:
        print("SORT FAILED", seed0 + trial, n, dtype, kind, {k.decode(): v for k, v in opts.items()}, flush=True)
        raise
    finally:
        set_all(SORT_DEFAULTS)
    # ---- group-by (the wide one-level plan with few bins so that small inputs reach it)
    m = int(rng.choice([rng.integers(1000, 30000), rng.integers(24000, 26000), rng.integers(60000, 140000)]))
    distinct = int(rng.choice([3, 100, 5000, 200000]))
    k = U.random_array(rng, np.int32, m, null_p=float(rng.choice([0, 0.01])), lo=-distinct // 2, hi=distinct // 2 + 1,
                       offset=int(rng.integers(0, 5)))
    val = U.random_array(rng, np.int64, m, null_p=float(rng.choice([0, 0.1])))
    gopts = dict(GB_DEFAULTS)
    gopts.update({b"groupby_partition_min_rows": 0, b"groupby_wide_max_bits": int(rng.choice([3, 6, 11])),
                  b"groupby_wide_rooms": int(rng.integers(0, 2)), b"groupby_wide_room_min_mean": int(rng.choice([1, 64, 1 << 14])),
                  b"groupby_probe_rows": int(rng.choice([24576, 1 << 25]))})
    set_all(gopts)
    try:
        need = 1 << max(12, int(np.ceil(np.log2(2 * min(distinct, m) + 4))))
        P.check_groupby_sum(amd, k, val, capacity=need << int(rng.integers(0, 3)), batches=int(rng.integers(1, 3)), use_pyarrow=False)
    except Exception:
        print("GROUPBY FAILED", seed0 + trial, m, distinct, {k_.decode(): v_ for k_, v_ in gopts.items()}, flush=True)
        raise
    finally:
        set_all(GB_DEFAULTS)
    if trial % 10 == 9:
        print(f"  {trial + 1} trials ok", flush=True)
print("SORT_GROUPBY_SOAK_OK")
