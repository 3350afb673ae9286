#!/usr/bin/env bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/suite_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/suite_pytest.log
tail -24 gpurun_out/suite_pytest.log
