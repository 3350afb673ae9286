#!/usr/bin/env bash
# does the stream-ordered RCCL double notice a forgotten stream dependency?  Two mutants of the engine library (one hipStreamWaitEvent removed
# each) run the 4-rank "fixed-big" case; the product library runs it beside them.  (Works on the GPU box's scratch copy of the tree.)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O1 -std=c++17 -fPIC -shared -o /tmp/librccl_double.so tests/rccl_double/rccl_double.cpp -lpthread || exit 1
cp claymore_amd/csrc/libclaymore_hip.so /tmp/product.so
for v in product mut_a mut_b; do
  [ $v = product ] && cp /tmp/product.so claymore_amd/csrc/libclaymore_hip.so || cp gpurun_libs/$v.so claymore_amd/csrc/libclaymore_hip.so
  for mode in ordered sync; do
    [ $mode = sync ] && export RCCL_DOUBLE_SYNC=1 || unset RCCL_DOUBLE_SYNC
    for w in 4 8; do
      out=$(MPM_RCCL_LIBRARY=/tmp/librccl_double.so timeout 300 python tests/rccl_double/run_group.py $w fixed-big 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300)
      echo "[$v, double $mode, world $w] $out" | tee -a gpurun_out/mutants.txt
    done
  done
done
cp /tmp/product.so claymore_amd/csrc/libclaymore_hip.so
