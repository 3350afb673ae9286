#!/usr/bin/env bash
# HIP API trace of the default bench window: how often the host synchronises (mpm_run_fixed: once per sync_interval substeps)
cd "$(dirname "$0")/.."
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ht
timeout 600 rocprofv3 --hip-trace --stats -d /tmp/ht -o t -- python $R/bench.py --no-cpu-baseline --flow-start 0 --steps 96 --warmup 8 > $R/gpurun_out/hip_trace_bench.json 2>/dev/null
python - <<PY > $R/gpurun_out/hip_trace.txt
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/ht/*.db")[0]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
t = "regions" if "regions" in tabs else None
print("# rocprofv3 --hip-trace --stats -- python bench.py --no-cpu-baseline --flow-start 0 --steps 96 --warmup 8  (C3; 104 substeps = 13 windows of 8)")
if t:
    rows = c.execute("select name, count(*), sum(end - start) from regions group by name order by count(*) desc").fetchall()
    for name, n, tot in rows[:25]:
        print(f"{name:45s} calls {n:7d}  total {tot/1e6:10.3f} ms")
else:
    print("tables:", tabs)
PY
cat $R/gpurun_out/hip_trace.txt
