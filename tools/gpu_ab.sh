#!/usr/bin/env bash
# same-box A/B of prebuilt libraries: tools/gpu_ab.sh [--scenes ...] lib...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab_libs.txt
timeout 1500 python tools/ab_libs.py "$@" 2>&1 | grep -v amdgpu.ids | tail -20
