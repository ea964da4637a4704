"""One sort of the failing rec8 sequence, alone in a process (stderr visible): uniform uint64 keys, the given knobs.
usage: exp_sort_rec8_step.py n "k=v k=v ..." [repeat]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd  # noqa: E402
from tests import parity_cases as P, util  # noqa: E402

n = int(sys.argv[1])
lib = arrow_amd._lib.get_lib()
for kv in sys.argv[2].split():
    k, v = kv.split("=")
    assert lib.arx_set_option(k.encode(), int(v)) == 0, kv
rng = np.random.default_rng(5)
arr = util.random_array(rng, np.uint64, n, offset=3)
names = (b"sort_wide_runs", b"sort_wide_rec8_runs", b"sort_wide_rec8_ties", b"sort_wide_rec8_given_up", b"sort_wide_wc_runs")
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
    c0 = {c: lib.arx_get_counter(c) for c in names}
    P.check_sort_indices(arrow_amd, arr, "ascending", "at_end", use_pyarrow=False)
    print("STEP_OK", rep, {c.decode(): lib.arx_get_counter(c) - v for c, v in c0.items()}, flush=True)
