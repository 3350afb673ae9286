#!/usr/bin/env bash
# round 4, call A: the GPU suite on the b-state engine, then the same-box A/B against the round-3 library
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04a_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04a_pytest.log
tail -15 gpurun_out/r04a_pytest.log
rm -f gpurun_out/ab_libs.txt
timeout 900 python tools/ab_libs.py --scenes c3,c3flow,c2,c5 r03.so bstate.so > gpurun_out/r04a_ab.log 2>&1
tail -8 gpurun_out/r04a_ab.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
tail -c 3000 gpurun_out/r04a_bench.json
