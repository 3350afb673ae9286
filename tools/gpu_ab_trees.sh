#!/usr/bin/env bash
# same-box comparison of two source trees (the repo and a copy of an older commit under .ab_old/): default bench windows
cd "$(dirname "$0")/.."
R=$PWD
: > $R/gpurun_out/ab_trees.txt
for rep in 1 2; do
for T in . .ab_old; do
  A=$(cd $T && python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
  B=$(cd $T && python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
  echo "[$T] 5+20: $A   10+100: $B" >> $R/gpurun_out/ab_trees.txt
done
done
C=$(cd .ab_old && sed -i 's/ap.add_argument("--fraction"/ap.add_argument("--start-step", type=int, default=0); ap.add_argument("--fraction"/; s/        eng.run_fixed(args.warmup, dt)/        eng.run_fixed(args.start_step, dt) if args.start_step else None; eng.run_fixed(args.warmup, dt)/' bench.py && python bench.py --no-cpu-baseline --steps 20 --warmup 5 --start-step 3000 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
D=$(python bench.py --no-cpu-baseline --steps 20 --warmup 5 --start-step 3000 2>/dev/null | grep -oE "\"g2p2g_ms\": [0-9.]*" | head -1)
echo "moving (start 3000): old $C   new $D" >> $R/gpurun_out/ab_trees.txt
