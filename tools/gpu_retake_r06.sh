#!/usr/bin/env bash
# round 6, LAST call: everything profiles/r06_* holds, re-taken on the shipped library in one go ("re-take the proxies last").
# Outputs: gpurun_out/ (profile set, bench line: tools/gpu_final_r06.sh) and gpurun_out/retake/ (proxies, long runs, probes), each headed by build_stamp.txt
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/retake
mkdir -p $O
export TMPDIR=/tmp
f() { grep -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path"; }
bash tools/gpu_final_r06.sh "" 2>&1 | tail -6
{ cat build_stamp.txt; timeout 900 python tools/mgsp_strong_local.py 20 1,2,4,8 y,y+aligned,x,octants,xz-columns 0 2>&1 | f; timeout 1500 python tools/mgsp_strong_local.py 20 1,2,4,8 y,y+aligned,x,octants,xz-columns 3000 2>&1 | f; } > $O/mgsp_partition.txt
{ cat build_stamp.txt; timeout 600 python tools/mgsp_rank_alone.py 40 2,4,8 2>&1 | f; } > $O/mgsp_rank_alone.txt
bash tools/gpu_rank_alone_prof.sh > /dev/null 2>&1
{ echo "# $(cat build_stamp.txt): tools/gpu_rank_alone_prof.sh (1/8 slab of C3 on the group driver, world 1 on RCCL, 40 substeps)"; cat gpurun_out/rank_alone_trace.txt; } > $O/rank_alone_kernel_trace.txt
{ echo "# $(cat build_stamp.txt): the last dispatches of the same run in start order (tools/rocpd_seq.py): the plain engine's substeps of the 1/8 slab first, the group driver's last"; grep -n "" gpurun_out/rank_alone_seq.txt | sed -n '2,60p;' | cut -d: -f2-; echo "..."; tail -52 gpurun_out/rank_alone_seq.txt; } | cut -c1-150 > $O/rank_alone_seq.txt
{ echo "# tools/long_run_check.py on the shipped library: self-check (particle count, lost / discarded / dropped, grid mass, finite positions) every 1000 substeps; G2P2G ms of the last launch"; cat build_stamp.txt
  timeout 900 python tools/long_run_check.py sand40m 12000 1000 2>&1 | f; timeout 600 python tools/long_run_check.py fluid12m 6000 1000 2>&1 | f; timeout 600 python tools/long_run_check.py sphere5m 3000 1000 2>&1 | f; } > $O/long_run.txt
timeout 600 python tools/probe_grid_parity.py > /dev/null 2>&1; { cat build_stamp.txt; cat gpurun_out/grid_parity_probe.txt; } > $O/grid_parity_probe.txt
{ echo "# MPM_PRINT_WORST=1 pytest -k one_particle ($(cat build_stamp.txt)): largest deviation of the HIP kernel from the reference's own statements (G16-G18) per material"
  MPM_PRINT_WORST=1 timeout 600 python -m pytest tests/test_parity_gpu.py -q -s -m gpu -k one_particle 2>&1 | f | grep -i "worst\|passed\|failed"; } > $O/one_particle_worst.txt
ls -la $O
