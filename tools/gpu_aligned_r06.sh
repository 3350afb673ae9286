#!/usr/bin/env bash
# round 6: block-aligned slabs (scenes.split_slabs(block=...), bench.py's default for --gpus N) against equal-count slabs on the one-GPU proxies
cd "$(dirname "$0")/.."
O=gpurun_out/aligned; mkdir -p $O
export TMPDIR=/tmp
f() { grep -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path"; }
{ cat build_stamp.txt; timeout 900 python tools/mgsp_rank_alone.py 40 4,8 2>&1 | f; } > $O/rank_alone.txt
{ cat build_stamp.txt; timeout 1200 python tools/mgsp_strong_local.py 20 1,4,8 y,y+aligned 0 2>&1 | f; timeout 1500 python tools/mgsp_strong_local.py 20 1,8 y,y+aligned,x,x+aligned 3000 2>&1 | f; } > $O/strong_local.txt
{ cat build_stamp.txt; WORLDS="8" FULL=1 bash tools/gpu_mp.sh 2>&1 | f | cut -c1-1500; } > $O/mp_launch.txt
timeout 900 python -m pytest tests/test_mgsp_gpu.py -m gpu -x -q > $O/pytest_mgsp.log 2>&1; tail -3 $O/pytest_mgsp.log
ls -la $O
