#!/usr/bin/env bash
# Round-6 profile set (run through gpurun): kernel traces + HBM / SQ counters of the C3 launch in the default (rest) window and in the
# flow (--start-step 3000), FC (C2) trace.  Summaries land in gpurun_out/prof_r06/*.txt; tools/make_pmc_json.py turns them into profiles/r06_pmc.json.
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/prof_r06
rm -rf $O; mkdir -p $O
STAMP=$(cat $R/build_stamp.txt 2>/dev/null || echo "no build stamp: run tools/stamp.sh before gpurun")
cd /tmp && export TMPDIR=/tmp
trace() { # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o t -- python $R/bench.py --no-cpu-baseline --flow-start 0 "$@" > $O/${name}_bench.json 2> /dev/null
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --flow-start 0 $*   ($STAMP)" > $O/${name}_kernel_trace.txt
  python $R/tools/rocpd_summary.py /tmp/prof_$name/t_results.db >> $O/${name}_kernel_trace.txt 2>&1
  rm -rf /tmp/prof_$name
}
pmc() { # name, counters (quoted), bench args...
  local name=$1; local ctr=$2; shift; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$name -o p -- python $R/bench.py --no-cpu-baseline --flow-start 0 "$@" > /dev/null 2>&1
  echo "# rocprofv3 --kernel-trace --pmc $ctr -- python bench.py --no-cpu-baseline --flow-start 0 $*   ($STAMP)" >> $O/${name}_pmc.txt
  python $R/tools/rocpd_summary.py /tmp/pmc_$name/p_results.db $LAST | grep -E "g2p2g|carry_grid|prepare_blocks|grid_update|substep_clear|compact_blocks|register_blocks" >> $O/${name}_pmc.txt 2>&1
  rm -rf /tmp/pmc_$name
}
SQA="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQC="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"
SQB="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES"
trace c3_default --steps 100 --warmup 10
pmc c3_default "FETCH_SIZE" --steps 3 --warmup 2
pmc c3_default "WRITE_SIZE" --steps 3 --warmup 2
pmc c3_default "$SQA" --steps 3 --warmup 2
pmc c3_default "$SQB" --steps 3 --warmup 2
pmc c3_default "$SQC" --steps 3 --warmup 2
trace c3_moving --start-step 3000 --steps 100 --warmup 10
LAST="--last 5"   # counters of the flow window only (the 5 launches after the 3000 untimed substeps), not of the ramp into it
pmc c3_moving "FETCH_SIZE" --start-step 3000 --steps 3 --warmup 2
pmc c3_moving "WRITE_SIZE" --start-step 3000 --steps 3 --warmup 2
pmc c3_moving "$SQA" --start-step 3000 --steps 3 --warmup 2
pmc c3_moving "$SQB" --start-step 3000 --steps 3 --warmup 2
pmc c3_moving "$SQC" --start-step 3000 --steps 3 --warmup 2
LAST=""
# round 6: a third window deep in the collapse (substeps 9000+: the pile spreads against the walls, cells hold hundreds of particles)
trace c3_deep --start-step 9000 --steps 40 --warmup 10
if [ "$1" = "all" ]; then
  trace c2_fc --scene sphere5m
  pmc c2_fc "$SQA" --scene sphere5m --steps 3 --warmup 2
  pmc c2_fc "$SQB" --scene sphere5m --steps 3 --warmup 2
  trace c5_fluid --scene fluid12m
  # the J-fluid instantiation with the full set the sand kernel has (LDS bank conflicts, the wait split)
  pmc c5_fluid "$SQA" --scene fluid12m --steps 3 --warmup 2
  pmc c5_fluid "$SQB" --scene fluid12m --steps 3 --warmup 2
  pmc c5_fluid "$SQC" --scene fluid12m --steps 3 --warmup 2
  pmc c5_fluid "FETCH_SIZE" --scene fluid12m --steps 3 --warmup 2
  pmc c5_fluid "WRITE_SIZE" --scene fluid12m --steps 3 --warmup 2
  # the two multi-GPU configurations whole on ONE GPU (they fit: 288 GB): the per-particle rate of FC / the J-fluid without C2's launch tail
  trace c4_fc_one_gpu --scene spheres40m --steps 40 --warmup 10
  trace c5_fluid_one_gpu --scene fluid100m --steps 40 --warmup 10
fi
ls -la $O
