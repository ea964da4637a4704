"""Headline Filter+Take + the streaming legs (cast / greater / filter selectivities / take forms) of bench.py, without the
CPU baselines, the 4e9-row group-by, the 2e9-row sort and the CallFunction leg: the quick A/B loop for kernel-form
changes of selection.hip / scalar.hip (one JSON line per run; `--tag` names it)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="")
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--no-other", action="store_true")
    a = ap.parse_args()
    import torch
    args = argparse.Namespace(gpus=1, steps=a.steps, warmup=3, workload="filter_take", rows=a.rows, groups=10_000_000,
                              hash_sum_rows=0, sort_rows=1 << 26, stream_rows=a.rows, callfunction_rows=a.rows,
                              extras_timeout=300.0, selectivity=0.10, null_p=0.10, cpu_sample_rows=0, cpu_groupby_rows=0,
                              cpu_sort_rows=0, cpu_budget_s=0.0, no_cpu_baseline=True, extras=False, option=[], backend="hip")
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    import arrow_amd as amd
    res = bench.run_filter_take(args, 0, 1, device)
    line = {"tag": a.tag, "ms_per_step": res["ms_per_step"], "kernel_ms": res["kernel_ms"], "filter_frac": res["roofline"]["frac"],
            "parity": res.get("parity_spot_check")}
    if not a.no_other:
        torch.cuda.empty_cache()
        other = bench.run_other_paths(amd, device, args)
        line["cast_ms"] = other["cast_f64_f32"]["ms"]
        line["cast_frac"] = other["cast_f64_f32"]["roofline_frac"]
        line["cast_exact"] = other["cast_f64_f32"]["bit_exact_vs_round_to_nearest_even_sample"]
        line["greater_ms"] = other["greater_f64"]["ms"]
        line["greater_frac"] = other["greater_f64"]["roofline_frac"]
        line["secondary"] = {k: (v.get("ms"), v.get("roofline_frac")) for k, v in other["secondary_configs"].items() if isinstance(v, dict)}
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
