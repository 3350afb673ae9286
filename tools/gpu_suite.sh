#!/usr/bin/env bash
# the whole GPU suite (all failures reported, not only the first)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
cat build_stamp.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=15 "$@" > gpurun_out/suite_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/suite_pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|pytest rc" gpurun_out/suite_pytest.log | tail -30
