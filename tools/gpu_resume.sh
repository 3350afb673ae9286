#!/usr/bin/env bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O1 -std=c++17 -fPIC -shared -o /tmp/librccl_double.so tests/rccl_double/rccl_double.cpp -lpthread || exit 1
for w in 2 3; do MPM_RCCL_LIBRARY=/tmp/librccl_double.so timeout 300 python tests/rccl_double/run_group.py $w resume 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600; done
