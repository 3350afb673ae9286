"""Where the HIP engine and the oracle drift apart in violent fuzz cases: the same cases against the oracle with its converged
double-precision SVD in place of the reference's four-sweep one (experiment knob mpmo_set_exact_svd).  usage: fuzz_exact_svd.py seed vmax maxsteps materials case [case ...]"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as g
from fuzz_scenes import random_scene
from parity_util import match, run_engine
from oracle_ffi import oracle_api
g.build_oracle()
seed, vmax, maxsteps = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
mats = tuple(int(c) for c in sys.argv[4])
want = [int(a) for a in sys.argv[5:]]
rng = np.random.default_rng(seed)
api = oracle_api()
def rel(a, b):
    w = 0.0
    for (xa, _, _), (xb, _, _) in zip(a, b):
        idx, _ = match(xb.astype(np.float64), xa.astype(np.float64))
        w = max(w, float((np.abs(xa[idx].astype(np.float64) - xb).max(axis=1) / np.abs(xb).max(axis=1)).max()))
    return w
for case in range(max(want) + 1):
    sc, nsteps = random_scene(rng, case, vmax, maxsteps, mats)
    if case not in want:
        continue
    hip = run_engine(sc, nsteps)["state"]
    api.raw.mpmo_set_exact_svd(0)
    ref = run_engine(sc, nsteps, api=api)["state"]
    api.raw.mpmo_set_exact_svd(1)
    exact = run_engine(sc, nsteps, api=api)["state"]
    api.raw.mpmo_set_exact_svd(0)
    print(f"case {case} ({nsteps} substeps, materials {[m['material'] for m in sc['models']]}): HIP vs oracle {rel(hip, ref):.2e} | HIP vs oracle with exact SVD {rel(hip, exact):.2e} | oracle vs oracle with exact SVD {rel(ref, exact):.2e}", flush=True)
