#!/usr/bin/env bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04f_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04f_pytest.log
tail -6 gpurun_out/r04f_pytest.log
rm -f gpurun_out/ab_libs.txt
timeout 1200 python tools/ab_libs.py --scenes c3,c3flow,c2,c5 "$@" > gpurun_out/r04f_ab.log 2>&1
tail -8 gpurun_out/r04f_ab.log
