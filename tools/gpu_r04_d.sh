#!/usr/bin/env bash
# round 4, call D: GPU suite (incl. the windowed group loop, the failing-rank test), rank-alone timing of the group loop (deferred / not)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04d_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04d_pytest.log
tail -30 gpurun_out/r04d_pytest.log
timeout 600 python tools/mgsp_rank_alone.py 40 4,8 > gpurun_out/r04d_rank_alone.txt 2>&1
MPM_GROUP_DEFER=0 timeout 600 python tools/mgsp_rank_alone.py 40 4,8 > gpurun_out/r04d_rank_alone_nodefer.txt 2>&1
tail -4 gpurun_out/r04d_rank_alone.txt gpurun_out/r04d_rank_alone_nodefer.txt
