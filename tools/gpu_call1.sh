#!/usr/bin/env bash
# round-2 call 1: cost-model microbench + dynamic VALU counts per ablation of the current kernel
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
timeout 300 tools/valu_microbench2 > gpurun_out/mb2.txt 2>&1
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_base.json 2> gpurun_out/bench_base.err
cd /tmp && export TMPDIR=/tmp
for A in 0 1 2 4 3 7 15; do
  rm -rf $R/gpurun_out/abl_$A
  MPM_G2P2G_ABLATE=$A timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/abl_$A -o pmc -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo "ABL=$A" >> $R/gpurun_out/abl_summary.txt
  python $R/tools/rocpd_summary.py $R/gpurun_out/abl_$A/pmc_results.db | grep g2p2g >> $R/gpurun_out/abl_summary.txt
  rm -rf $R/gpurun_out/abl_$A
done
