#!/usr/bin/env bash
# end-of-round verification: GPU test suite, smoke, the profile set, the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/final_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/final_pytest.log
tail -12 gpurun_out/final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/gpu_profile_r03.sh all > gpurun_out/final_profile.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
cat gpurun_out/final_bench.json | cut -c1-700
python tools/mgsp_rank_alone.py 20 2,4,8 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/final_rank_alone.txt
python tools/mgsp_strong_local.py 20 1,2,4,8 2>&1 | grep -v "amdgpu.ids" > gpurun_out/final_strong_local.txt
tail -3 gpurun_out/final_rank_alone.txt | cut -c1-300
