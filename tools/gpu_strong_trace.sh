#!/usr/bin/env bash
cd /root/repo
R=$PWD
cd /tmp && export TMPDIR=/tmp
for W in 1 8; do
rm -rf /tmp/t8
MPM_GROUP_EARLY_EXCHANGE=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/t8 -o t -- python $R/tools/mgsp_strong_local.py 20 $W > $R/gpurun_out/strong_trace_run_$W.txt 2>&1
python $R/tools/rocpd_summary.py /tmp/t8/t_results.db > $R/gpurun_out/strong_trace_$W.txt 2>&1
grep -v "amdgpu.ids\|^W2026" $R/gpurun_out/strong_trace_run_$W.txt | tail -4
head -40 $R/gpurun_out/strong_trace_$W.txt
done
