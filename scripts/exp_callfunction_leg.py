"""bench.py's `callfunction` leg alone (pyarrow.compute / Acero on device-resident arrays through the registration
shim), printed as JSON.  Usage: exp_callfunction_leg.py [rows]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
args = argparse.Namespace(rows=rows, callfunction_rows=rows, groups=10_000_000, null_p=0.10, selectivity=0.10)
device = torch.device("cuda", 0)
values, validity, mask, _ = bench.gen_filter_inputs(rows, device, 0, args.null_p, args.selectivity)
print(json.dumps(bench.callfunction_leg(args, values, validity, mask, device), indent=1))
