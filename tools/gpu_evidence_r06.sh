#!/usr/bin/env bash
# round 6, after the retake: three pieces of evidence on the shipped library that the profile set does not hold
#   1. MPM_G2P2G_STATS build of the shipped source (gpurun_libs/stats.so): claim losers / edge lanes / split pairs / iterations with a serial entry, at rest and in the flow
#   2. the driver's 8-rank launch of bench.py at full size as 8 processes on the one GPU (multi-process RCCL stand-in), at rest and 3000 substeps into the collapse
#   3. randomised HIP-vs-oracle parity (tools/fuzz_parity.py) with seeds the test suite does not use
cd "$(dirname "$0")/.."
O=gpurun_out/evidence; mkdir -p $O
export TMPDIR=/tmp
f() { grep -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path"; }
{ cat build_stamp.txt; echo "# python tools/g2p2g_stats_run.py gpurun_libs/stats.so  (the shipped source built with -DMPM_EXPERIMENT -DMPM_G2P2G_STATS)"; timeout 600 python tools/g2p2g_stats_run.py gpurun_libs/stats.so 2>&1 | f; } > $O/g2p2g_stats.txt
{ cat build_stamp.txt; WORLDS="" FULL=1 bash tools/gpu_mp.sh 2>&1 | f; W=8 bash tools/gpu_mp_flow.sh 2>&1 | f; } > $O/mp_launch.txt
cp gpurun_out/mp_full_w8.json gpurun_out/mp_full_w8.err gpurun_out/mp_flow_w8.json gpurun_out/mp_flow_w8.err $O/ 2>/dev/null
{ cat build_stamp.txt; echo "# python tools/fuzz_parity.py 24 601 3 120 0123 ; 24 602 6 80 123 (violent)"; timeout 900 python tools/fuzz_parity.py 24 601 3 120 0123 2>&1 | f | tail -30; timeout 900 python tools/fuzz_parity.py 24 602 6 80 123 2>&1 | f | tail -30; } > $O/fuzz.txt
ls -la $O
