#!/usr/bin/env bash
# latency-side counters of G2P2G (default window and flow): average LDS / VMEM instruction latency (SQ_INST_LEVEL_* / SQ_INSTS_*), issue-side breakdown
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/pmc_latency.txt
: > $O
cd /tmp && export TMPDIR=/tmp
for START in 0 3000; do
for SET in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES"; do
  rm -rf /tmp/pm
  timeout 600 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pm -o p -- python $R/bench.py --no-cpu-baseline --flow-start 0 --start-step $START --steps 3 --warmup 2 > /dev/null 2>&1
  echo "# start-step $START: $SET" >> $O
  python $R/tools/rocpd_summary.py /tmp/pm/p_results.db | grep -E "g2p2g" | grep -v "^void.*[0-9]  *[0-9]*  *[0-9]" >> $O 2>&1
done
done
cat $O
