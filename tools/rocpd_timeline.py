#!/usr/bin/env python
"""Print the kernel timeline (start offset, duration, gap) of the last N dispatches of a rocpd database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
rows = rows[-n:]
t0 = rows[0][1]
prev_end = None
for name, s, e, st in rows:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f} us  gap {gap:8.1f} us  stream {st}  {name.split('(')[0][:70]}")
    prev_end = max(prev_end or e, e)
