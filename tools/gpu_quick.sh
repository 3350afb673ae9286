#!/usr/bin/env bash
# On the GPU box: kernel time of G2P2G on C3 plus SQ counters per 64-particle iteration.
# usage (through gpurun): bash tools/gpu_quick.sh [tag]
TAG=${1:-q}
cd "$(dirname "$0")/.."
R=$PWD
OUT=$R/gpurun_out/quick_$TAG.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*|\"ms_per_step\": [0-9.]*" > $OUT
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_LEVEL_WAVES"; do
  rm -rf $R/gpurun_out/quick_pmc
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $R/gpurun_out/quick_pmc -o pmc -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $R/gpurun_out/quick_pmc/pmc_results.db | grep "g2p2g" | grep "SQ_" | awk '{name=""; v=0; for(i=1;i<=NF;i++){ if($i ~ /^SQ_/) name=$i; if($i ~ /^avg=/){split($i,b,"="); v=b[2]} }; printf "%-22s per dispatch %.4g  per iteration %.1f\n", name, v, v/626688.0}' >> $OUT
  python $R/tools/rocpd_summary.py $R/gpurun_out/quick_pmc/pmc_results.db | grep "g2p2g" | grep -v "SQ_" | head -2 >> $OUT
done
rm -rf $R/gpurun_out/quick_pmc
