#!/usr/bin/env bash
# round 6: [tests selected by $1] + smoke + the driver's bench line + the profile set (tools/gpu_profile_r06.sh all)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
cat build_stamp.txt
if [ -n "$1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -k "$1" > gpurun_out/r06_select_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_select_pytest.log
  grep -E "^(FAILED|ERROR)|passed|failed|pytest rc" gpurun_out/r06_select_pytest.log | tail -12
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_smoke.log 2>&1; tail -1 gpurun_out/r06_smoke.log
# the profile set first: bench.py attaches the counters of profiles/r06_pmc.json only if they were taken with the library it loads
bash tools/gpu_profile_r06.sh all > gpurun_out/r06_profile.log 2>&1
tail -3 gpurun_out/r06_profile.log
python tools/make_pmc_json.py gpurun_out/prof_r06 profiles/r06_pmc.json > /dev/null && cp profiles/r06_pmc.json gpurun_out/r06_pmc.json
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default_line.json 2> gpurun_out/r06_bench.err; echo "bench rc $?"
tail -c 300 gpurun_out/r06_bench_default_line.json
