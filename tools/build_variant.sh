#!/usr/bin/env bash
# build an engine library variant into gpurun_libs/<name>.so for a same-box A/B (tools/gpu_ab.sh -> tools/ab_libs.py): tools/build_variant.sh name [-D...]
cd "$(dirname "$0")/.."
n=$1; shift
mkdir -p gpurun_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-unused-value -fno-slp-vectorize "$@" -o gpurun_libs/$n.so claymore_amd/csrc/claymore_hip.hip -ldl -lpthread
