#!/usr/bin/env bash
# round 4, call E: the two new full-size tests + the bench line (same-input CPU baseline, watchdog)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "full_size_flow_invariants_c3 or full_size_c5_fluid_dam_8" ) > gpurun_out/r04e_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04e_pytest.log
tail -12 gpurun_out/r04e_pytest.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04e_bench.json 2> gpurun_out/r04e_bench.err ) 2>&1 | tail -4
tail -c 1500 gpurun_out/r04e_bench.json; tail -5 gpurun_out/r04e_bench.err
free -g | head -2; nproc
