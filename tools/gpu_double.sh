#!/usr/bin/env bash
# the group driver's RCCL branch through the in-process double in its stream-ordered mode (and, for comparison, the synchronising one)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_mgsp_gpu.py -m gpu -x -q -k "rccl_double or failing_rank" > gpurun_out/double_async_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/double_async_pytest.log
tail -15 gpurun_out/double_async_pytest.log
RCCL_DOUBLE_SYNC=1 timeout 1500 python -m pytest tests/test_mgsp_gpu.py -m gpu -x -q -k "rccl_double or failing_rank" > gpurun_out/double_sync_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/double_sync_pytest.log
tail -4 gpurun_out/double_sync_pytest.log
