#!/usr/bin/env bash
# energy per operation: one instruction class at a time on all CUs, rocm-smi power / sclk sampled beside it (tools/energy_microbench.hip)
cd "$(dirname "$0")/.."
OUT=gpurun_out/energy.txt
: > $OUT
for MODE in 0 1 2 3 9 4 5 6 7 8; do
  ./gpurun_bin/energy_microbench $MODE 2.5 > gpurun_out/energy_mode.txt 2>&1 &
  BP=$!
  sleep 0.8
  S=""
  for i in 1 2 3 4; do
    L=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed -E 's/.*\(([0-9]+)Mhz\).*/sclk \1/; s/.*Power \(W\): ([0-9.]+).*/W \1/' | tr '\n' ' ')
    S="$S | $L"
    sleep 0.3
  done
  wait $BP
  echo "$(cat gpurun_out/energy_mode.txt) $S" >> $OUT
done
cat $OUT
