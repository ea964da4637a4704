"""Kernel timeline of the LAST `count` dispatches of a rocprofv3 --kernel-trace run: start offset, duration and the idle
gap before each kernel (us).  Usage: rocprof_timeline.py <results.db> [count]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = cur.execute("select name, start, end from kernels order by start").fetchall()[-count:]
t0 = rows[0][1]
prev = None
for n, s, e in rows:
    gap = 0 if prev is None else (s - prev) / 1e3
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f} us  gap {gap:8.1f}  {n.replace('void ', '')[:90]}")
    prev = e
print(f"span {(rows[-1][2] - t0) / 1e3:.1f} us, busy {sum(e - s for _, s, e in rows) / 1e3:.1f} us")
