#!/usr/bin/env python
"""The last N dispatches of a rocprofv3 rocpd (.db) kernel trace in start order: offset from the first one, duration, gap to the previous end, grid, queue.
Usage: python tools/rocpd_seq.py results.db [N=80]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
print("# columns:", cols)
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = c.execute(f"select name, start, end, grid_x, {q} from kernels order by start desc limit {n}").fetchall()[::-1]
t0, prev_end = rows[0][1], None
for name, s, e, gx, qid in rows:
    gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:8.2f}"
    print(f"{(s - t0) / 1e3:10.2f} us  dur {(e - s) / 1e3:8.2f}  gap {gap:>8s}  q {qid}  grid {gx:9d}  {name.split('(')[0][:80]}")
    prev_end = e if prev_end is None else max(prev_end, e)
