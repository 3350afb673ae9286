#!/usr/bin/env bash
# wave-level counters of G2P2G for several prebuilt libraries (gpurun_libs/*.so): tools/gpu_libs_pmc.sh scene lib...   (on the GPU box: the in-tree library is overwritten)
cd "$(dirname "$0")/.."
R=$PWD
SCENE=$1; shift
O=$R/gpurun_out/libs_pmc.txt
: > $O
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  cp $R/gpurun_libs/$L $R/claymore_amd/csrc/libclaymore_hip.so
  for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH"; do
    rm -rf /tmp/pm
    timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pm -o p -- python $R/bench.py --no-cpu-baseline --flow-start 0 --scene $SCENE --steps 3 --warmup 2 > /dev/null 2>&1
    echo "# $L $SCENE: $SET" >> $O
    python $R/tools/rocpd_summary.py /tmp/pm/p_results.db | grep -E "g2p2g" >> $O 2>&1
  done
done
cat $O
