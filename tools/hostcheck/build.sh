#!/usr/bin/env bash
# x86 build of the device math for tools/hostcheck/run.py (development aid; the product is the gfx950 build)
cd "$(dirname "$0")/../.."
/opt/rocm/lib/llvm/bin/clang++ -O2 -std=c++17 -fPIC -shared -ffp-contract=fast -Itools/hostcheck -Iclaymore_amd/csrc -o tools/hostcheck/libhostmath.so tools/hostcheck/check_math.cpp
