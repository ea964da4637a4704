"""Summarise rocprofv3 (rocpd sqlite) output: per-kernel duration stats from a --kernel-trace run
and per-kernel counter means from --pmc runs.  Usage:
  python scripts/rocprof_summary.py trace <results.db> [name-filter]
  python scripts/rocprof_summary.py pmc   <results.db> [name-filter]
Only dispatches longer than 1/4 of the kernel's longest dispatch are averaged in the `big_*`
columns (bench.py also launches the same kernels on a small parity sample); `big_med_us` is their median — one
kernel template can serve several bench legs (the filter compaction runs the 10 % headline and the 25 / 50 % legs),
the median is the duration of the leg with the most dispatches."""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name if len(name) <= 88 else name[:85] + "..."


def _grid_expr(cur, table):
    """SQL expression for a dispatch's grid size from whatever columns this rocpd schema has (or None)."""
    cols = {r[1] for r in cur.execute(f"pragma table_info({table})").fetchall()}
    for cand in (("grid_size",), ("grid_size_x", "grid_size_y", "grid_size_z"), ("grid_x", "grid_y", "grid_z")):
        if all(c in cols for c in cand):
            return " * ".join(f"max({c}, 1)" if len(cand) > 1 else c for c in cand)
    return None


def trace(db, flt):
    """One row per kernel, then — VERDICT r4 item 7 — one row per (kernel, grid size) for every kernel that was launched
    with more than one grid: a template that serves several bench legs (the filter compaction: 10 % headline, 25 / 50 %
    legs, the 2 M-row parity sample) gets an average PER LEG, and the headline leg's row reproduces bench.py's
    roofline.avg_kernel_ms."""
    cur = sqlite3.connect(db).cursor()
    grid = _grid_expr(cur, "kernels")
    rows = cur.execute(f"select name, duration, {grid or '0'} from kernels").fetchall()
    agg, legs = defaultdict(list), defaultdict(lambda: defaultdict(list))
    for n, d, g in rows:
        if flt in n:
            agg[n].append(d)
            legs[n][g].append(d)
    total = sum(sum(v) for v in agg.values())
    print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'big_n':>6s} {'big_avg_us':>11s} {'big_min_us':>11s} {'big_med_us':>11s} {'pct':>6s}")
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        big = [x for x in v if x >= max(v) / 4]
        med = sorted(big)[len(big) // 2]
        print(f"{short(n):90s} {len(v):6d} {sum(v) / 1e6:10.3f} {sum(v) / len(v) / 1e3:10.1f} {len(big):6d} "
              f"{sum(big) / len(big) / 1e3:11.1f} {min(big) / 1e3:11.1f} {med / 1e3:11.1f} {100 * sum(v) / total:6.1f}")
    if grid is None:
        print("(this rocpd schema has no grid-size column: no per-leg rows)")
        return
    print()
    print("per bench leg = per (kernel, grid size, duration cluster); kernels with one leg only are complete above")
    print(f"{'kernel':90s} {'grid':>12s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'med_us':>10s}")
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if max(v) < 50_000 or (len(legs[n]) < 2 and max(v) <= min(v) * 1.12):      # (never longer than 50 us, or one leg)
            continue
        for g, d in sorted(legs[n].items(), key=lambda kv: -sum(kv[1])):
            if max(d) < max(v) / 50:
                continue
            # legs that share a grid (the same rows at 10 / 25 / 50 % selectivity) differ in duration by far more than
            # run-to-run noise: split the sorted durations wherever the next one is > 12 % longer
            d = sorted(d)
            groups = [[d[0]]]
            for x in d[1:]:
                (groups[-1].append(x) if x <= groups[-1][-1] * 1.12 else groups.append([x]))
            for c in groups:
                print(f"{short(n):90s} {g:12d} {len(c):6d} {sum(c) / len(c) / 1e3:10.1f} {min(c) / 1e3:10.1f} {c[len(c) // 2] / 1e3:10.1f}")


def pmc(db, flt):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, value, duration from counters_collection").fetchall()
    agg = defaultdict(list)
    for n, c, v, d in rows:
        if flt in n:
            agg[(n, c)].append((v, d))
    print(f"{'kernel':90s} {'counter':>14s} {'n':>5s} {'big_n':>6s} {'big_mean_value':>16s} {'big_avg_us':>11s}")
    for (n, c), v in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        dmax = max(x[1] for x in v)
        big = [x for x in v if x[1] >= dmax / 4]
        print(f"{short(n):90s} {c:>14s} {len(v):5d} {len(big):6d} {sum(x[0] for x in big) / len(big):16.1f} "
              f"{sum(x[1] for x in big) / len(big) / 1e3:11.1f}")


if __name__ == "__main__":
    mode, db = sys.argv[1], sys.argv[2]
    flt = sys.argv[3] if len(sys.argv) > 3 else "arx::"
    (trace if mode == "trace" else pmc)(db, flt)
