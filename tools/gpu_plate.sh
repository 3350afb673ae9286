#!/usr/bin/env bash
# the plate scenes (run_group.py 2 plate / plate-fall) with the library before the fix (gpurun_libs/opt8.so) and with the in-tree one
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O1 -std=c++17 -fPIC -shared -o /tmp/librccl_double.so tests/rccl_double/rccl_double.cpp -lpthread || exit 1
cp claymore_amd/csrc/libclaymore_hip.so /tmp/product.so
for v in ${LIBS:-opt8 product}; do
  [ $v = product ] && cp /tmp/product.so claymore_amd/csrc/libclaymore_hip.so || cp gpurun_libs/$v.so claymore_amd/csrc/libclaymore_hip.so
  for k in plate plate-fall; do
    out=$(MPM_RCCL_LIBRARY=/tmp/librccl_double.so timeout 300 python tests/rccl_double/run_group.py 2 $k 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-700)
    echo "[$v, $k] $out" | tee -a gpurun_out/plate.txt
  done
done
cp /tmp/product.so claymore_amd/csrc/libclaymore_hip.so
