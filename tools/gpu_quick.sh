#!/usr/bin/env bash
# On the GPU box: kernel time of G2P2G on C3 plus VALU / SALU / LDS instructions per 64-particle iteration.
# usage (through gpurun): bash tools/gpu_quick.sh [warmup]
W=${1:-2}
cd "$(dirname "$0")/.."
R=$PWD
python bench.py --steps 5 --warmup $W --no-cpu-baseline 2>&1 | grep -oE "\"g2p2g_ms\": [0-9.]*"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/quick_pmc
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $R/gpurun_out/quick_pmc -o pmc -- python $R/bench.py --steps 2 --warmup $W --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py $R/gpurun_out/quick_pmc/pmc_results.db | grep "g2p2g" | grep "SQ_" | awk '{name=""; v=0; for(i=1;i<=NF;i++){ if($i ~ /^SQ_/) name=$i; if($i ~ /^avg=/){split($i,b,"="); v=b[2]} }; printf "%-22s per dispatch %.4g  per iteration %.1f\n", name, v, v/626688.0}'
