#!/usr/bin/env bash
# the multi-process bench launch through the multi-process RCCL double: one manual run per world size (stdout / stderr kept), then the tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/rcclmp
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O1 -std=c++17 -fPIC -shared -o /tmp/librccl_double_mp.so tests/rccl_double/rccl_double_mp.cpp -lpthread || exit 1
for W in ${WORLDS:-2 4 8}; do
  MPM_RCCL_LIBRARY=/tmp/librccl_double_mp.so RCCL_DOUBLE_DIR=/tmp/rcclmp RCCL_DOUBLE_TIMEOUT_S=200 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$W --master-addr 127.0.0.1 --master-port $((29600 + W)) \
    bench.py --gpus $W --steps 6 --warmup 3 --oversubscribe --fraction ${FRACTION:-0.03125} --watchdog 500 > gpurun_out/mp_w$W.json 2> gpurun_out/mp_w$W.err
  echo "world $W rc $?"; tail -c 1500 gpurun_out/mp_w$W.json; echo; grep -v "^\[W\|^W0\|^\*\*\*\*\|OMP_NUM" gpurun_out/mp_w$W.err | tail -${ERRTAIL:-25}
done
if [ -n "$PYTEST" ]; then
  timeout 1200 python -m pytest tests/test_mgsp_gpu.py -m gpu -x -q -k "multi_process_bench" > gpurun_out/mp_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/mp_pytest.log
  tail -5 gpurun_out/mp_pytest.log
fi
if [ -n "$FULL" ]; then   # the driver's command at full size, 8 processes on the one GPU
  MPM_RCCL_LIBRARY=/tmp/librccl_double_mp.so RCCL_DOUBLE_DIR=/tmp/rcclmp RCCL_DOUBLE_TIMEOUT_S=300 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29655 \
    bench.py --gpus 8 --steps 20 --warmup 5 --oversubscribe > gpurun_out/mp_full_w8.json 2> gpurun_out/mp_full_w8.err
  echo "full world 8 rc $?"; tail -c 1800 gpurun_out/mp_full_w8.json; echo; grep -v "^\[W\|^W0\|^\*\*\*\*\|OMP_NUM\|amdgpu.ids\|Gloo" gpurun_out/mp_full_w8.err | tail -25
fi
