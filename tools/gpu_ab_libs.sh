#!/usr/bin/env bash
# same-box A/B of prebuilt engine libraries (gpurun_libs/*.so): C3 rest / flow, C2 FC, C5 fluid
cd "$(dirname "$0")/.."
R=$PWD
: > $R/gpurun_out/ab_libs.txt
for L in "$@"; do
  cp gpurun_libs/$L claymore_amd/csrc/libclaymore_hip.so
  A=$(python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
  A2=$(python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
  B=$(python bench.py --no-cpu-baseline --steps 20 --warmup 5 --start-step 3000 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
  C=$(python bench.py --scene sphere5m --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
  D=$(python bench.py --scene fluid12m --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
  echo "[$L] C3 rest 5+20 $A 10+100 $A2 | C3 flow $B | C2 FC $C | C5 fluid $D" >> $R/gpurun_out/ab_libs.txt
done
