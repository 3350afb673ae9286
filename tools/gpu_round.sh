#!/usr/bin/env bash
# one GPU call: claim / retry statistics of the current design over the collapse, then a same-box A/B of library variants
cd "$(dirname "$0")/.."
cp gpurun_libs/lib_stats.so claymore_amd/csrc/libclaymore_hip.so
timeout 300 python tools/g2p2g_stats_run.py > gpurun_out/stats3.txt 2>&1
timeout 1500 tools/gpu_ab_libs.sh "$@"
