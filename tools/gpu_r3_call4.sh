#!/usr/bin/env bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python tools/ab_libs.py --scenes c3,c3flow lib_dev.so lib_noloop.so lib_dev.so lib_noloop.so > gpurun_out/c4_ab.log 2>&1
tail -6 gpurun_out/c4_ab.log
