#!/usr/bin/env bash
# copies what tools/gpu_retake_r06.sh left under gpurun_out/ into profiles/r06_* (run here, after the gpurun call): the raw files as they are;
# profiles/r06_mgsp_partition.txt and r06_c3_deep_window.txt carry a hand-written summary and are written by tools/summarise_retake_r06.py
cd "$(dirname "$0")/.."
P=gpurun_out/prof_r06
for n in c2_fc c3_deep c3_default c3_moving c4_fc_one_gpu c5_fluid c5_fluid_one_gpu; do
  cp $P/${n}_kernel_trace.txt profiles/r06_${n}_kernel_trace.txt
  [ -f $P/${n}_pmc.txt ] && cp $P/${n}_pmc.txt profiles/r06_${n}_pmc.txt
done
cp gpurun_out/r06_pmc.json profiles/r06_pmc.json
cp gpurun_out/r06_bench_default_line.json profiles/r06_bench_default_line.json
R=gpurun_out/retake
for n in grid_parity_probe long_run mgsp_rank_alone one_particle_worst rank_alone_kernel_trace rank_alone_seq; do cp $R/$n.txt profiles/r06_$n.txt; done
python tools/summarise_retake_r06.py
git status --short profiles | head -40
