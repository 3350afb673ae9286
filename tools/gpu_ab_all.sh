#!/usr/bin/env bash
# same-box A/B of build variants on C3 (rest / flow), C2 (FC) and the C5 per-rank fluid
cd "$(dirname "$0")/.."
R=$PWD
: > $R/gpurun_out/ab_all.txt
for FLAGS in "$@"; do
  (cd claymore_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-unused-value -fno-slp-vectorize $FLAGS -o libclaymore_hip.so claymore_hip.hip) 2>/dev/null
  A=$(python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
  A2=$(python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
  B=$(python bench.py --no-cpu-baseline --steps 20 --warmup 5 --start-step 3000 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
  C=$(python bench.py --scene sphere5m --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
  D=$(python bench.py --scene fluid12m --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
  echo "[$FLAGS] C3 rest $A $A2 | C3 flow $B | C2 FC $C | C5 fluid $D" >> $R/gpurun_out/ab_all.txt
done
