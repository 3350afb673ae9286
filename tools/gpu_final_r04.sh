#!/usr/bin/env bash
# round 4 final: GPU suite, the driver's bench line, the profile set
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_final_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_final_pytest.log
tail -4 gpurun_out/r04_final_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_smoke.log 2>&1; tail -1 gpurun_out/r04_smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_default_line.json 2> gpurun_out/r04_bench.err
tail -c 600 gpurun_out/r04_bench_default_line.json
bash tools/gpu_profile_r04.sh all > gpurun_out/r04_profile.log 2>&1
tail -3 gpurun_out/r04_profile.log
