#!/usr/bin/env bash
# WARNING: this pass did not finish on the MI355X pool of round 3 (the first rocprofv3 run with TCC_EA0_* counters was still running after 15 minutes): bound it with a short timeout before trying again.
# L2 (TCC) request mix of the C3 G2P2G launch at rest and in the flow window: reads / writes / atomics at the L2 and towards the fabric
cd "$(dirname "$0")/.."
R=$PWD
cd /tmp && export TMPDIR=/tmp
A="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
B="TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum"
run() { # label, counters, extra bench args
  local label=$1 ctr=$2; shift; shift
  rm -rf /tmp/tcc
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/tcc -o p -- python $R/bench.py --no-cpu-baseline --flow-start 0 --steps 3 --warmup 2 "$@" > /dev/null 2>&1
  echo "# $label: $ctr"
  python $R/tools/rocpd_summary.py /tmp/tcc/p_results.db --last 5 | grep -E "g2p2g" | grep -v "^void mpm::g2p2g_kernel<2>  *[0-9]"
}
run rest "$A"
run rest "$B"
run flow "$A" --start-step 3000
run flow "$B" --start-step 3000
