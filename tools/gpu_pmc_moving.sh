#!/usr/bin/env bash
# SQ / HBM counters of G2P2G in the moving window of C3 (after --start-step substeps) vs at rest.
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/pmc_moving.txt
rm -f $O
cd /tmp && export TMPDIR=/tmp
for START in 0 3000; do
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pm
  timeout 600 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pm -o p -- python $R/bench.py --no-cpu-baseline --start-step $START --steps 3 --warmup 2 > /dev/null 2>&1
  echo "# start-step $START: $SET" >> $O
  python - >> $O <<PY
import sqlite3
db = sqlite3.connect("/tmp/pm/p_results.db"); c = db.cursor()
# last 3 dispatches of g2p2g only (the timed ones)
ids = [r[0] for r in c.execute("select dispatch_id from kernels where name like '%g2p2g%' order by dispatch_id desc limit 3")]
dur = [r[0] for r in c.execute("select duration from kernels where name like '%g2p2g%' order by dispatch_id desc limit 3")]
print("  g2p2g duration (last 3):", [d/1e6 for d in dur])
q = "select counter_name, avg(value) from counters_collection where dispatch_id in (%s) group by counter_name" % ",".join(str(i) for i in ids)
for name, v in c.execute(q): print(f"  {name:32s} {v:.5g}")
PY
done
done
