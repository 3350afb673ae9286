#!/usr/bin/env bash
# same-box A/B of build variants on the J-fluid (C5 per-rank share) and FC (C2) scenes
cd "$(dirname "$0")/.."
R=$PWD
: > $R/gpurun_out/ab_fluid.txt
for FLAGS in "$@"; do
  (cd claymore_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-unused-value -fno-slp-vectorize $FLAGS -o libclaymore_hip.so claymore_hip.hip) 2>/dev/null
  for S in fluid12m sphere5m; do
    A=$(python bench.py --scene $S --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
    B=$(python bench.py --scene $S --no-cpu-baseline --steps 50 --warmup 10 --start-step 2000 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
    echo "[$FLAGS] $S rest $A  after 2000 steps $B" >> $R/gpurun_out/ab_fluid.txt
  done
done
