#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) result: per-kernel time stats and, if present, PMC counter sums per kernel.
Usage: python tools/rocpd_summary.py results.db [--last N] [> profiles/xyz.txt]
--last N [--skip K]: the PMC section covers only the last N dispatches of each kernel (a window at the end of a long run), after skipping the K most recent."""
import sqlite3
import sys


def short(name):
    name = name.split("(")[0]
    return name if len(name) < 70 else name[:67] + "..."


def main(path, last=0, skip=0):
    db = sqlite3.connect(path)
    c = db.cursor()
    print(f"# {path}")
    print("## kernel time (rocprofv3 --kernel-trace), ns")
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':70s} {'calls':>6s} {'total_ns':>14s} {'avg_ns':>12s} {'min_ns':>12s} {'max_ns':>12s} {'pct':>6s}")
    for name, n, tot, avg, mn, mx in rows:
        print(f"{short(name):70s} {n:6d} {tot:14d} {avg:12.0f} {mn:12d} {mx:12d} {100.0 * tot / total:6.2f}")
    regs = c.execute("select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x), max(grid_x) from kernels group by name order by sum(duration) desc").fetchall()
    print("\n## resources (vgpr, agpr, sgpr, lds bytes, workgroup, max grid)")
    for r in regs:
        print(f"{short(r[0]):70s} {r[1]} {r[2]} {r[3]} {r[4]} {r[5]} {r[6]}")
    try:
        if last > 0:
            pm = []
            for (kname,) in c.execute("select distinct name from kernels").fetchall():
                ids = [r[0] for r in c.execute("select dispatch_id from kernels where name = ? order by dispatch_id desc", (kname,)).fetchall()][skip:skip + last]
                if not ids:
                    continue
                q = ",".join(str(i) for i in ids)
                pm += c.execute(f"select ?, p.counter_name, count(*), sum(p.value), avg(p.value) from counters_collection p where p.dispatch_id in ({q}) group by p.counter_name", (kname,)).fetchall()
                t = c.execute(f"select count(*), avg(duration) from kernels where dispatch_id in ({q})").fetchone()
                print(f"## last {last} dispatches: {short(kname):60s} n={t[0]} avg_ns={t[1]:.0f}")
            pm.sort(key=lambda r: (r[0], r[1]))
        else:
            pm = c.execute("select k.name, p.counter_name, count(*), sum(p.value), avg(p.value) from counters_collection p join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name order by k.name").fetchall()
    except Exception as e:  # schema differences
        pm = []
        try:
            cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
            print("counters_collection columns:", cols)
        except Exception:
            pass
    if pm:
        print("\n## PMC counters per kernel (sum over dispatches, avg per dispatch)")
        for name, cn, n, s, a in pm:
            print(f"{short(name):70s} {cn:28s} n={n:4d} sum={s:.6g} avg={a:.6g}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0, int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0)
