#!/usr/bin/env bash
# second counter set for prebuilt libraries through tools/ab_libs.py: instruction mix, instruction fetch, L2 hits: tools/gpu_ablib_pmc2.sh scenes lib...
cd "$(dirname "$0")/.."
R=$PWD
SC=$1; shift
O=$R/gpurun_out/ablib_pmc2.txt
: > $O
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  for SET in "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAVES" "SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"; do
    rm -rf /tmp/pm
    timeout 600 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pm -o p -- python $R/tools/ab_libs.py --scenes $SC $L > /dev/null 2>&1
    echo "# $L $SC: $SET" >> $O
    python $R/tools/rocpd_summary.py /tmp/pm/p_results.db | grep -E "g2p2g" | grep "n=" >> $O 2>&1
  done
done
cat $O
