#!/usr/bin/env bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python tools/ab_libs.py --scenes c3,c3flow,c2 lib_pk.so lib_sc128.so lib_sc96.so lib_pk.so lib_sc128.so lib_sc96.so > gpurun_out/c6_ab.log 2>&1
tail -8 gpurun_out/c6_ab.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c6_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c6_pytest.log
tail -4 gpurun_out/c6_pytest.log
