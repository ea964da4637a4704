"""profiles/<name>_traffic.json from a PMC summary (scripts/rocprof_summary.py pmc output of a FETCH_SIZE pass and a
WRITE_SIZE pass of scripts/prof_sort_groupby.py): HBM bytes of ONE run = sum over the pipeline's kernels of
(FETCH_SIZE x 2 [gfx950 tallies 128-B requests at 64 B] + WRITE_SIZE) x KiB x launches per run.
usage: traffic_from_pmc.py groupby|sort <pmc.txt> <runs in the pass> <rows> <source note> [form]"""
import json
import re
import sys

name, path, runs, rows, source = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
form = sys.argv[6] if len(sys.argv) > 6 else None
prefix = {"groupby": ("arx::gbp_", "arx::groupby_", "arx::gbl_"), "sort": ("arx::msd", "arx::sort_", "arx::msdw_")}[name]
kern = {}
for line in open(path):
    m = re.match(r"^(arx::\S+?)[<(].*\s(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if not m or not m.group(1).startswith(prefix):
        continue
    full = line.split(m.group(2))[0].strip()
    k = kern.setdefault(full, {"fetch_bytes_x2": 0, "write_bytes": 0, "launches_per_run": 0.0})
    n, big_n, mean = int(m.group(3)), int(m.group(4)), float(m.group(5))
    # `mean` is over the big launches only; the small ones (probe slices, tails) are charged at the same mean: an upper bound
    per_run = mean * 1024.0 * n / runs if big_n == n else mean * 1024.0 * big_n / runs
    k["launches_per_run"] = max(k["launches_per_run"], n / runs)
    if m.group(2) == "FETCH_SIZE":
        k["fetch_bytes_x2"] = int(2 * per_run)
    else:
        k["write_bytes"] = int(per_run)
total = sum(k["fetch_bytes_x2"] + k["write_bytes"] for k in kern.values())
out = {"rows": rows, "form": form, "what": f"{name}: sum over its kernels of FETCH_SIZE x 2 + WRITE_SIZE (KiB units) per run; "
       "launches smaller than a kernel's largest (probe slice, tails) are left out of its mean and of the sum",
       "source": source, "hbm_bytes_per_launch": total, "bytes_per_row": round(total / rows, 1),
       "algorithmic_bytes_per_row": 12 if name == "groupby" else 16, "kernels": kern}
json.dump(out, open(f"profiles/{name}_traffic.json", "w"), indent=1)
print(name, total, round(total / rows, 2), "B/row")
