import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
prev_end = None
for n, s, e in rows:
    if prev_end is not None and s - prev_end > 2_000_000:
        print(f"gap {(s - prev_end)/1e6:8.2f} ms before {n[:60]}")
    prev_end = e
print("span ms", (rows[-1][2] - rows[0][1]) / 1e6, "kernels", len(rows))
