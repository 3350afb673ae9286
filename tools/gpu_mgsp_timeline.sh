#!/usr/bin/env bash
# kernel timeline of one substep of two FULL-SIZE ranks (40.1 M particles each, touching C3 columns) as contexts of one GPU, with the
# exchange enqueued BEFORE the interior G2P2G (MPM_GROUP_EARLY_EXCHANGE=1: the RCCL transport's ordering, here with asynchronous
# device-to-device copies in place of ncclSend / ncclRecv) -> gpurun_out/mgsp_timeline.txt
cd "$(dirname "$0")/.."
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
MPM_GROUP_EARLY_EXCHANGE=1 timeout 900 rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $R/tools/mgsp_weak_local.py 2 6 > $R/gpurun_out/mgsp_timeline_run.txt 2>&1
python $R/tools/rocpd_timeline.py /tmp/tl/t_results.db 70 > $R/gpurun_out/mgsp_timeline.txt 2>&1
tail -3 $R/gpurun_out/mgsp_timeline_run.txt
head -70 $R/gpurun_out/mgsp_timeline.txt
