"""Error growth of selected fuzz cases: pos_rel (HIP vs oracle) after increasing numbers of substeps.  usage: fuzz_growth.py seed vmax maxsteps materials case [case ...]"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as g
from fuzz_scenes import random_scene
from parity_util import match_and_compare, run_pair
g.build_oracle()
seed, vmax, maxsteps = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
mats = tuple(int(c) for c in sys.argv[4])
want = [int(a) for a in sys.argv[5:]]
rng = np.random.default_rng(seed)
for case in range(max(want) + 1):
    sc, nsteps = random_scene(rng, case, vmax, maxsteps, mats)
    if case not in want:
        continue
    row = []
    for k in sorted(set([nsteps // 8, nsteps // 4, nsteps // 2, 3 * nsteps // 4, nsteps])):
        w = match_and_compare(run_pair(sc, k))
        row.append(f"{k}: {w['pos_rel']:.1e}")
    print(f"case {case} materials {[m['material'] for m in sc['models']]} v0 {[tuple(round(v, 1) for v in m['v0']) for m in sc['models']]}: " + ", ".join(row), flush=True)
