#!/usr/bin/env bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O1 -std=c++17 -fPIC -shared -o /tmp/librccl_double.so tests/rccl_double/rccl_double.cpp -lpthread || exit 1
export MPM_RCCL_LIBRARY=/tmp/librccl_double.so
for spec in "$@"; do   # world:defer
  W=${spec%%:*}; D=${spec##*:}   # world:defer
  MPM_GROUP_DEFER=$D timeout 900 python tools/mgsp_flow_check.py $W ${STEPS:-3030} ${CHUNK:-500} ${SCENE:-c3} 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/flow_check.txt
done
