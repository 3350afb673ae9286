"""N-rank vs 1-rank agreement (positions) for a colliding sand / FC scene: how much the partitioning changes the result."""
import sys, threading
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from claymore_amd import _ffi, scenes
from claymore_amd.mgsp import LocalGroup, MgspGroupRank
from parity_util import match, run_engine

def run_group(sc, world, nsteps, dt):
    lg = LocalGroup(world)
    ranks = [MgspGroupRank(sc, r, world, device=0, local_group=lg) for r in range(world)]
    lg.create()
    out, errs = [None] * world, []
    def work(r):
        try:
            ranks[r].initial_setup(); ranks[r].run_fixed(nsteps, dt); out[r] = ranks[r].local_state()
        except Exception as e: errs.append(repr(e))
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    for r in ranks: r.close()
    assert not errs, errs
    return out

for mat, name in ((_ffi.SAND, "sand"), (_ffi.FIXED_COROTATED, "fc")):
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=0.5, speed=2.0, youngs=2e4, material=mat)
    if mat == _ffi.SAND:
        for m in sc["models"]: m["params"] = {}
    nsteps = 150
    one = run_engine(sc, nsteps, 1e-4)
    for world in (2, 3):
        res = run_group(sc, world, nsteps, 1e-4)
        worst = 0.0
        for m in range(2):
            xm = np.concatenate([r[m][0] for r in res]); xo = one["state"][m][0]
            idx, _ = match(xo.astype(np.float64), xm.astype(np.float64))
            worst = max(worst, (np.abs(xm[idx].astype(np.float64) - xo).max(axis=1) / np.abs(xo).max(axis=1)).max())
        print(f"{name}: {world} ranks vs 1 rank after {nsteps} substeps: max rel position difference {worst:.3e}", flush=True)
