#!/usr/bin/env bash
# SQ counters of G2P2G for prebuilt library variants (gpurun_libs/*.so) on one scene: tools/gpu_pmc_libs.sh "<bench args>" libA.so libB.so ...
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/pmc_libs.txt
rm -f $O
ARGS=$1; shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
cp $R/gpurun_libs/$L $R/claymore_amd/csrc/libclaymore_hip.so
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_IFETCH SQ_INSTS_BRANCH"; do
  rm -rf /tmp/pm
  timeout 600 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pm -o p -- python $R/bench.py --no-cpu-baseline $ARGS --steps 3 --warmup 2 > /dev/null 2>&1
  echo "# [$L] $ARGS: $SET" >> $O
  python - >> $O <<PY
import sqlite3
db = sqlite3.connect("/tmp/pm/p_results.db"); c = db.cursor()
ids = [r[0] for r in c.execute("select dispatch_id from kernels where name like '%g2p2g%' order by dispatch_id desc limit 3")]
dur = [r[0] for r in c.execute("select duration from kernels where name like '%g2p2g%' order by dispatch_id desc limit 3")]
print("  g2p2g duration (last 3):", [d/1e6 for d in dur])
q = "select counter_name, avg(value) from counters_collection where dispatch_id in (%s) group by counter_name" % ",".join(str(i) for i in ids)
for name, v in c.execute(q): print(f"  {name:32s} {v:.5g}")
PY
done
done
