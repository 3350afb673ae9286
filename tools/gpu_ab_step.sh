#!/usr/bin/env bash
# same-box A/B of prebuilt engine libraries (gpurun_libs/*.so) on the whole substep: ms_per_step and phase times of the default bench window
cd "$(dirname "$0")/.."
: > gpurun_out/ab_step.txt
for L in "$@"; do
  cp gpurun_libs/$L claymore_amd/csrc/libclaymore_hip.so
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$L]', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['config']['phases_ms'].items()})" >> gpurun_out/ab_step.txt
done
