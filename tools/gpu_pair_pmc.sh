#!/usr/bin/env bash
# wave-level counters of the J-fluid G2P2G, one particle per lane (MPM_G2P2G_PAIRS=0) against two (=1): where a wave's cycles go
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/pair_pmc.txt
cat build_stamp.txt > $O
cd /tmp && export TMPDIR=/tmp
for P in 0 1; do
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES"; do
  rm -rf /tmp/pm
  MPM_G2P2G_PAIRS=$P timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pm -o p -- python $R/bench.py --no-cpu-baseline --flow-start 0 --scene ${SCENE:-fluid12m} --steps 3 --warmup 2 > /dev/null 2>&1
  echo "# MPM_G2P2G_PAIRS=$P: $SET" >> $O
  python $R/tools/rocpd_summary.py /tmp/pm/p_results.db | grep -E "g2p2g" >> $O 2>&1
done
done
cat $O
