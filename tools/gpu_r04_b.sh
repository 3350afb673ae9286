#!/usr/bin/env bash
# round 4, call B: GPU suite on the working tree, then same-box A/B of library variants (gpurun_libs/) incl. the leave-a-part-out hacks
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04b_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04b_pytest.log
tail -5 gpurun_out/r04b_pytest.log
rm -f gpurun_out/ab_libs.txt
timeout 1200 python tools/ab_libs.py --scenes c3,c3flow "$@" > gpurun_out/r04b_ab.log 2>&1
tail -12 gpurun_out/r04b_ab.log
