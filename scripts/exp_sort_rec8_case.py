"""One parametrisation of tests/parity_cases.py::check_sort_wide_rec8 on the GPU, outside pytest (stderr visible).
usage: exp_sort_rec8_case.py n bits gap2 shift rpt1 rpt2 b2max wc prefetch"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd  # noqa: E402
from tests import parity_cases as P  # noqa: E402

n, bits, gap2, shift, r1, r2, b2max, wc, pf = [int(x) for x in sys.argv[1:10]]
lib = arrow_amd._lib.get_lib()
P.check_sort_wide_rec8(arrow_amd, lib, np.random.default_rng(5), n, bits=bits, gap2=gap2, shift=shift, rpt=(r1, r2), b2max=b2max, wc=wc, prefetch=pf)
print("CASE_OK", sys.argv[1:], flush=True)
