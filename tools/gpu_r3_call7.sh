#!/usr/bin/env bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/c7_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c7_pytest.log
tail -16 gpurun_out/c7_pytest.log
MPM_GROUP_EARLY_EXCHANGE=1 timeout 600 python -m pytest tests/test_mgsp_gpu.py -m gpu -x -q -k "cpp_group" > gpurun_out/c7_pytest_early.log 2>&1; echo "pytest early rc $?" >> gpurun_out/c7_pytest_early.log
tail -4 gpurun_out/c7_pytest_early.log
bash tools/gpu_mgsp_timeline.sh
