#!/usr/bin/env bash
# Round-2 profile set (run through gpurun): kernel traces + HBM / SQ counters for C3 (rest and moving windows), C2 (FC),
# C5 per-rank J-fluid.  Summaries land in gpurun_out/prof_r02/*.txt; the ones to keep are copied to profiles/ by hand.
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
trace() { # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o t -- python $R/bench.py --no-cpu-baseline "$@" > $O/${name}_bench.json 2> /dev/null
  python $R/tools/rocpd_summary.py /tmp/prof_$name/t_results.db > $O/${name}_kernel_trace.txt 2>&1
  rm -rf /tmp/prof_$name
}
pmc() { # name, counters (quoted), bench args...
  local name=$1; local ctr=$2; shift; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$name -o p -- python $R/bench.py --no-cpu-baseline "$@" > /dev/null 2>&1
  echo "# rocprofv3 --kernel-trace --pmc $ctr -- python bench.py --no-cpu-baseline $*" >> $O/${name}_pmc.txt
  python $R/tools/rocpd_summary.py /tmp/pmc_$name/p_results.db | grep -E "g2p2g|carry_grid|prepare_blocks|grid_update" >> $O/${name}_pmc.txt 2>&1
  rm -rf /tmp/pmc_$name
}
trace c3_default
pmc c3 "FETCH_SIZE" --steps 3 --warmup 2
pmc c3 "WRITE_SIZE" --steps 3 --warmup 2
pmc c3 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --steps 3 --warmup 2
pmc c3 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES" --steps 3 --warmup 2
trace c3_moving --start-step 3000 --steps 100 --warmup 10
pmc c3_moving "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --start-step 3000 --steps 3 --warmup 2
trace c2_fc --scene sphere5m
pmc c2_fc "FETCH_SIZE" --scene sphere5m --steps 3 --warmup 2
pmc c2_fc "WRITE_SIZE" --scene sphere5m --steps 3 --warmup 2
pmc c2_fc "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --scene sphere5m --steps 3 --warmup 2
trace c5_fluid --scene fluid12m
pmc c5_fluid "FETCH_SIZE" --scene fluid12m --steps 3 --warmup 2
pmc c5_fluid "WRITE_SIZE" --scene fluid12m --steps 3 --warmup 2
pmc c5_fluid "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --scene fluid12m --steps 3 --warmup 2
