#!/usr/bin/env bash
# round 3, GPU call 1: correctness of the new kernel pieces, same-box A/B of the variants, clocks / power under the kernel, counter list
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/ab_libs.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c1_pytest.log
timeout 900 python tools/ab_libs.py lib_r2.so lib_new.so lib_nopre.so lib_tol5.so lib_r2.so lib_new.so > gpurun_out/c1_ab.log 2>&1
# clocks and power while the G2P2G loop runs (C3, 1500 substeps ~ 3 s)
( python bench.py --no-cpu-baseline --steps 1500 --warmup 10 > gpurun_out/c1_long.json 2> gpurun_out/c1_long.err ) &
BP=$!
: > gpurun_out/c1_smi.txt
for i in $(seq 1 60); do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|busy" | tr '\n' ' ' >> gpurun_out/c1_smi.txt; echo >> gpurun_out/c1_smi.txt
  kill -0 $BP 2>/dev/null || break
  sleep 0.4
done
wait $BP
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L 2>/dev/null | grep -iE "IFETCH|ICACHE|SQC_|SQ_INST_LEVEL|SQ_WAIT|SQ_ACTIVE|SQ_INSTS_|SQ_BUSY|SQ_LDS|TA_BUSY|TCP_|GRBM" | head -150) > gpurun_out/c1_counters.txt 2>&1
echo done
