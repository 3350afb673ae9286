#!/usr/bin/env bash
# round 4, call C: A/B of library variants (all four scenes), then the SQ counter passes of the flow window on the in-tree library
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab_libs.txt
timeout 1200 python tools/ab_libs.py --scenes c3,c3flow,c2,c5 "$@" > gpurun_out/r04c_ab.log 2>&1
tail -8 gpurun_out/r04c_ab.log
O=$R/gpurun_out/prof_r04c; rm -rf $O; mkdir -p $O
cd /tmp
pmc() { # name, counters (quoted), bench args...
  local name=$1; local ctr=$2; shift; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$name -o p -- python $R/bench.py --no-cpu-baseline --flow-start 0 "$@" > /dev/null 2>&1
  echo "# rocprofv3 --kernel-trace --pmc $ctr -- python bench.py --no-cpu-baseline --flow-start 0 $*" >> $O/${name}_pmc.txt
  python $R/tools/rocpd_summary.py /tmp/pmc_$name/p_results.db $LAST | grep -E "g2p2g|prepare_blocks" >> $O/${name}_pmc.txt 2>&1
  rm -rf /tmp/pmc_$name
}
SQA="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQB="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES"
SQC="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"
pmc c3_default "$SQA" --steps 3 --warmup 2
pmc c3_default "$SQB" --steps 3 --warmup 2
pmc c3_default "$SQC" --steps 3 --warmup 2
LAST="--last 5"
pmc c3_moving "$SQA" --start-step 3000 --steps 3 --warmup 2
pmc c3_moving "$SQB" --start-step 3000 --steps 3 --warmup 2
pmc c3_moving "$SQC" --start-step 3000 --steps 3 --warmup 2
cat $O/*.txt
