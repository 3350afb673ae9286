#!/usr/bin/env bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/c3_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c3_pytest.log
tail -5 gpurun_out/c3_pytest.log
timeout 600 python tools/ab_libs.py lib_dev.so lib_new.so lib_dev.so lib_new.so > gpurun_out/c3_ab.log 2>&1
tail -6 gpurun_out/c3_ab.log
