#!/usr/bin/env bash
# A/B of compile-time variants of the engine on one box: tools/gpu_ab.sh "<flags A>" "<flags B>" ... ; prints g2p2g ms at rest / moving
cd "$(dirname "$0")/.."
R=$PWD
: > $R/gpurun_out/ab.txt
for FLAGS in "$@"; do
  (cd claymore_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-unused-value -fno-slp-vectorize $FLAGS -o libclaymore_hip.so claymore_hip.hip) 2>/dev/null
  for rep in 1 2; do
    A=$(python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
    echo "[$FLAGS] rest $A" >> $R/gpurun_out/ab.txt
  done
  B=$(python bench.py --no-cpu-baseline --steps 20 --warmup 5 --start-step 3000 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
  echo "[$FLAGS] moving $B" >> $R/gpurun_out/ab.txt
done
