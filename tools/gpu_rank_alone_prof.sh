#!/usr/bin/env bash
# kernel trace of one rank's share (1/8 slab of C3) on the group driver, world 1 on RCCL: what the multi-GPU loop adds to a substep
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ra
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ra -o t -- python $R/tools/mgsp_rank_alone.py 40 8 > $R/gpurun_out/rank_alone_prof.log 2>&1
python $R/tools/rocpd_summary.py /tmp/prof_ra/t_results.db > $R/gpurun_out/rank_alone_trace.txt 2>&1
python $R/tools/rocpd_seq.py /tmp/prof_ra/t_results.db 1600 > $R/gpurun_out/rank_alone_seq.txt 2>&1
grep -v amdgpu.ids $R/gpurun_out/rank_alone_prof.log | tail -4
head -40 $R/gpurun_out/rank_alone_trace.txt | cut -c1-150
