#!/usr/bin/env bash
# round 5, first GPU call: the whole GPU suite on the new library, a same-box A/B of the stay fast path, the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
cat build_stamp.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/suite_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/suite_pytest.log
tail -30 gpurun_out/suite_pytest.log
rm -f gpurun_out/ab_libs.txt
timeout 600 python tools/ab_libs.py --scenes c3,c3flow,c2,c5 --reps 2 r5base.so r5fast.so 2>&1 | grep -v amdgpu.ids | tail -8
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"
cat gpurun_out/bench_default.json | head -c 6000
