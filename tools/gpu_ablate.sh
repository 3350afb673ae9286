#!/usr/bin/env bash
# where do G2P2G's time and VALU instructions go?  One launch (substep 8) with parts switched off (-DMPM_DEBUG_ABLATE build):
# 1 particle loads from one hot region, 2 stores to it, 4 no scatter chain, 8 no gather, 16 no material update
cd "$(dirname "$0")/.."
R=$PWD
(cd claymore_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-unused-value -fno-slp-vectorize -DMPM_DEBUG_ABLATE -o libclaymore_hip.so claymore_hip.hip) 2>/dev/null
cat > /tmp/abl.py <<PY
import sys; sys.path.insert(0, "$R")
from claymore_amd import scenes
from claymore_amd.engine import build_engine
sc = scenes.sand_column(9)
e = build_engine(sc); e.initial_setup(); e.run_fixed(8, 1e-4)
e.run_fixed(1, 1e-4)
print("g2p2g_ms %.3f" % e.timers().g2p2g_ms)
PY
: > $R/gpurun_out/ablate.txt
cd /tmp && export TMPDIR=/tmp
for A in 0 4 8 16 28 31; do
  rm -rf /tmp/pa
  V=$(MPM_ABLATE=$A MPM_ABLATE_FROM=8 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d /tmp/pa -o p -- python /tmp/abl.py 2>/dev/null | grep g2p2g_ms)
  C=$(python - <<PY
import sqlite3
db = sqlite3.connect("/tmp/pa/p_results.db"); c = db.cursor()
i = c.execute("select dispatch_id, duration from kernels where name like '%g2p2g%' order by dispatch_id desc limit 1").fetchone()
vals = dict(c.execute("select counter_name, value from counters_collection where dispatch_id = %d" % i[0]).fetchall())
print("kernel %.3f ms  VALU/iter %.0f  LDS/iter %.0f  SALU/iter %.0f  active_valu_quads/iter %.0f  wave_quads/iter %.0f" % (i[1]/1e6, vals.get("SQ_INSTS_VALU",0)/626688, vals.get("SQ_INSTS_LDS",0)/626688, vals.get("SQ_INSTS_SALU",0)/626688, vals.get("SQ_ACTIVE_INST_VALU",0)/626688, vals.get("SQ_WAVE_CYCLES",0)/626688))
PY
)
  echo "ablate=$A: $V | $C" >> $R/gpurun_out/ablate.txt
done
