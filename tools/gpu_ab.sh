#!/usr/bin/env bash
# same-box A/B of library variants (gpurun_libs/): tools/gpu_ab.sh <scenes> lib...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
SC=$1; shift
rm -f gpurun_out/ab_libs.txt
timeout 1500 python tools/ab_libs.py --scenes $SC "$@" > gpurun_out/ab.log 2>&1
tail -12 gpurun_out/ab.log
