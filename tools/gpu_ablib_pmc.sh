#!/usr/bin/env bash
# wave-level counters of G2P2G for prebuilt libraries through tools/ab_libs.py (its loader tolerates libraries of earlier rounds): tools/gpu_ablib_pmc.sh scenes lib...
cd "$(dirname "$0")/.."
R=$PWD
SC=$1; shift
O=$R/gpurun_out/ablib_pmc.txt
: > $O
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    rm -rf /tmp/pm
    timeout 600 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pm -o p -- python $R/tools/ab_libs.py --scenes $SC $L > /dev/null 2>&1
    echo "# $L $SC: $SET" >> $O
    python $R/tools/rocpd_summary.py /tmp/pm/p_results.db | grep -E "g2p2g" >> $O 2>&1
  done
done
cat $O
