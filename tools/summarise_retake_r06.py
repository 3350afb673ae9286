"""Writes the two round-6 profile files that carry a summary in front of the raw output of tools/gpu_retake_r06.sh:
profiles/r06_mgsp_partition.txt (the 8-rank table of the partition-shape study) and profiles/r06_c3_deep_window.txt (the window at
substep 9000).  Every number is parsed from the files of ONE gpurun call under gpurun_out/; the history lines (earlier libraries of the
round) are kept as text.  Run by tools/install_retake_r06.sh."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")

raw = open(os.path.join(G, "retake", "mgsp_partition.txt")).read().split("\n")
stamp = raw[0]
lib = stamp.split()[-1]
rows = {}   # (window, shape) -> (work, bound, halo %, MB)
t1 = {}
window = None
for i, l in enumerate(raw):
    if l.startswith("# C3:"):
        window = "flow" if "in the flow" in l else "rest"
    m = re.match(r"world 1 .*wall ([\d.]+) ms", l)
    if m:
        t1[window] = float(m.group(1))
    m = re.match(r"shape (\S+)( block-aligned)? \((\d), (\d), (\d)\) world 8: .*work vs 1 rank x([\d.]+); speed-up bound ([\d.]+) of 8", l)
    if m:
        nxt = raw[i + 1]
        halo = re.search(r"\((\d+) % of all\)", nxt).group(1)
        mb = re.search(r"\(([\d.]+) MB max\)", nxt).group(1)
        rows[(window, m.group(1) + ("+aligned" if m.group(2) else ""))] = (float(m.group(6)), float(m.group(7)), int(halo), float(mb))
names = [("y", "y slabs (1,8,1)"), ("y+aligned", "y slabs, block-aligned"), ("x", "x slabs (8,1,1)"), ("octants", "octants (2,2,2)"), ("xz-columns", "x-z columns (4,1,2)")]
names = [n for n in names if ("rest", n[0]) in rows and ("flow", n[0]) in rows]
best_rest = max(names, key=lambda n: rows[("rest", n[0])][1])
best_flow = max(names, key=lambda n: rows[("flow", n[0])][1])
out = [stamp,
       "# VERDICT r5 #3: the SHAPE of the static particle partition of C3, one-GPU proxy (tools/mgsp_strong_local.py; every rank a context of the one GPU, in-process transport), re-taken LAST on the shipped library (tools/gpu_retake_r06.sh; this header: tools/summarise_retake_r06.py):",
       "# N * T1 / wall is an UPPER bound of the N-GPU speed-up.  Caveat of the proxy: with 8 ranks the wall is close to the host threads' CPU time (2.2-4.0 ms per",
       "# substep: ~20 launches per rank and substep into ONE device), i.e. partly the runtime's launch throughput on one device, which 8 processes on 8 GPUs do not share.",
       "#",
       "# summary, 8 ranks:      at rest: work x, bound of 8, halo blocks, max MB sent | in the flow (3000 substeps first): work x, bound of 8, halo blocks, max MB sent"]
for key, label in names:
    r, f = rows[("rest", key)], rows[("flow", key)]
    out.append(f"#   {label:<22s} x{r[0]:.2f}  {r[1]:.2f}  {r[2]:2d} %  {r[3]:5.1f} MB                        |  x{f[0]:.2f}  {f[1]:.2f}  {f[2]:2d} %  {f[3]:5.1f} MB")
fmt = lambda w: " ".join(f"{rows[(w, k)][1]:.2f}" for k, _ in names if k != "y+aligned")
out += [
    f"# -> at rest (the window bench.py times) the best cut is {best_rest[1]} (bound {rows[('rest', best_rest[0])][1]:.2f} of 8); once the column has collapsed the y slabs are pancakes",
    f"#    ({rows[('flow', 'y')][2]} % of a rank's particle blocks are halo blocks, {rows[('flow', 'y')][3]:.0f} MB per substep) and {best_flow[1]} win: bound {rows[('flow', best_flow[0])][1]:.2f} of 8.  bench.py --partition picks the shape",
    "#    (default: the longest axis = y for C3, named in config.parallelism), and since the end of round 6 it moves the cut planes to particle-block faces (\"block-aligned\", scenes.split_slabs: no block shared by two ranks",
    "#    at the start - 83 853 instead of 90 948 particle blocks over the 8 ranks, half the halo blocks and half the bytes sent; the material cut drifts off the faces as the column moves, so the flow rows do not differ beyond the proxy's noise).  The targets of the verdict (inflation <= 1.5, bound >= 4.8) are NOT met at rest by any shape; in the flow the x slabs reach the",
    "#    bound (4.76-5.13 over the round's libraries) but not the inflation (1.56-1.68): a rank's fixed cost (five launch-bound rebuild kernels, tagging, exchange, one host synchronisation per window) does not",
    "#    shrink with its share - and a faster single-rank kernel LOWERS the bound: the same table earlier in the round, y / x / octants / x-z columns at rest | in the flow:",
    "#      library 9f9471531d09f319 (T1 1.199 / 2.246 ms, seven stream events per substep): 3.64 3.52 2.85 3.29 | 4.30 4.88 4.20 4.65",
    "#      library 555f16501d61c901 (T1 1.190 / 2.302, three events):                        3.82 3.69 2.88 3.27 | 4.52 5.13 4.15 4.76",
    "#      library 7191df0aae204ac3 (T1 1.145 / 2.084: LDS waits by hand):                   3.74 3.53 2.78 3.16 | 4.13 4.76 3.87 4.40",
    f"#      this library {lib} (T1 {t1['rest']:.3f} / {t1['flow']:.3f}):" + " " * 20 + f"{fmt('rest')} | {fmt('flow')}",
]
out += raw[1:]
open(os.path.join(P, "r06_mgsp_partition.txt"), "w").write("\n".join(out).rstrip("\n") + "\n")

# ---- the deep window
deep = json.loads(open(os.path.join(G, "prof_r06", "c3_deep_bench.json")).read().strip().split("\n")[-1])
line = json.loads(open(os.path.join(G, "r06_bench_default_line.json")).read().strip().split("\n")[-1])
rd, rl = deep["roofline"], line["roofline"]["deep"]
txt = [
    f"# {stamp}: the window deep in the collapse (VERDICT r5 #2: a third window at step 9000); also roofline.deep of the default bench line (profiles/r06_bench_default_line.json); written by tools/summarise_retake_r06.py",
    f"python bench.py --no-cpu-baseline --flow-start 0 --start-step 9000 --steps 40 --warmup 10 (under rocprofv3, profiles/r06_c3_deep_kernel_trace.txt): G2P2G {rd['kernel_ms']:.4f} ms per launch = {rd['frac']:.3f} of the HBM roofline (round 5: 2.28 ms = 0.317); {deep['ms_per_step']:.4f} ms per substep; blocks {deep['config']['blocks']}",
    f"default bench line, roofline.deep (substeps 9005-9025): G2P2G {rl['kernel_ms']:.4f} ms = {rl['frac']:.3f}; {rl['ms_per_step']:.4f} ms per substep; blocks {rl['blocks']}",
    "# earlier in the round: the pair kernel before the LDS waits were issued by hand 2.115 / 2.061 ms = 0.341 / 0.350 on one box (library 9f9471531d09f319), 2.197 / 2.216 = 0.329 / 0.326 on another (555f16501d61c901);",
    "# with the waits by hand and plain 512-record chunks (7191df0aae204ac3) 2.070 / 2.009 = 0.349 / 0.359.  This window sits behind 9000 substeps of sustained load and moves by 5-7 % from box to box, the rest window of the same runs by 1 %.",
]
open(os.path.join(P, "r06_c3_deep_window.txt"), "w").write("\n".join(txt) + "\n")
print("profiles/r06_mgsp_partition.txt, profiles/r06_c3_deep_window.txt written for", stamp)
