#!/usr/bin/env bash
# HBM-side traffic of g2p2g in the flow window for two engine libraries (gpurun_libs/$1, $2): tools/ab_libs.py under rocprofv3 --pmc, 20 timed + 5 warm-up launches per library
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/abp
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/abp -o p -- python $R/tools/ab_libs.py --scenes c3flow $1 $2 > /dev/null 2>&1
  echo "# $ctr: $2 (last 25 launches)"; python $R/tools/rocpd_summary.py /tmp/abp/p_results.db --last 25 | grep -E "g2p2g|prepare_blocks|carry_grid"
  echo "# $ctr: $1 (the 25 launches before)"; python $R/tools/rocpd_summary.py /tmp/abp/p_results.db --last 25 --skip 25 | grep -E "g2p2g|prepare_blocks|carry_grid"
done
