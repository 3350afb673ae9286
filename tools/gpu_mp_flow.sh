#!/usr/bin/env bash
# 8 processes on the one GPU (multi-process RCCL double), the full C3 column, 3000 untimed substeps into the collapse, then the timed window:
# capacity growth, halo growth and the padded key lists of the windowed group loop in the flow regime.  Not a measurement.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/rcclmp
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O1 -std=c++17 -fPIC -shared -o /tmp/librccl_double_mp.so tests/rccl_double/rccl_double_mp.cpp -lpthread || exit 1
W=${W:-8}
MPM_RCCL_LIBRARY=/tmp/librccl_double_mp.so RCCL_DOUBLE_DIR=/tmp/rcclmp RCCL_DOUBLE_TIMEOUT_S=300 timeout 1700 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$W --master-addr 127.0.0.1 --master-port 29677 \
  bench.py --gpus $W --steps 20 --warmup 5 --start-step ${START:-3000} --oversubscribe --watchdog 1600 > gpurun_out/mp_flow_w$W.json 2> gpurun_out/mp_flow_w$W.err
echo "flow world $W rc $?"; tail -c 1500 gpurun_out/mp_flow_w$W.json; echo; grep -v "^\[W\|^W0\|^\*\*\*\*\|OMP_NUM\|amdgpu.ids\|Gloo" gpurun_out/mp_flow_w$W.err | tail -12
