"""The flow regime against the ORACLE at 5 M particles (test infrastructure; tests/test_parity_gpu.py::test_mid_size_flow_parity_630k is the same scene at 1/8 of
the size and 160 substeps): a 64 x 154 x 64-cell column of Drucker-Prager sand (5 046 272 particles, 512^3) on the floor's wall zone, thrown at it obliquely at
speed x (1.5, -4, 0.8) m/s; both engines are stepped side by side (the oracle on `threads` OpenMP threads) and compared every `every` substeps: particles matched by
nearest neighbour, the DISTRIBUTION of the position deviation (a yielding granular flow amplifies rounding differences - the largest deviation of 5 M particles grows
with the number of substeps, the bulk does not), b, log Jp and the block counts.
With `self` as the fifth argument the HIP engine is replaced by a SECOND ORACLE on 3/4 of the threads: the oracle's G2P2G walks the particle blocks on OpenMP threads
(dynamic schedule) and adds its arenas to the grid with atomic float adds, like the reference's kernel - the order of those sums, and of the bucket appends, differs
from run to run as it does on a GPU.  Oracle against oracle is therefore the deviation the reference's ALGORITHM shows against itself when nothing but the order of
float additions changes: the floor under any parity statement after that many substeps.
usage: flow_parity_large.py [nsteps=300] [threads=64] [speed=0.5: 0.1 cells per substep, as the 630 k test; 1.0: 0.2] [every=100] [self]"""
import sys
import time

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as g
from claymore_amd import scenes
from claymore_amd.engine import build_engine
from parity_util import match, to_b
from test_parity_gpu import oracle_api_threads

g.build_oracle()
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 64
speed = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
every = int(sys.argv[4]) if len(sys.argv) > 4 else 100
self_mode = len(sys.argv) > 5 and sys.argv[5] == "self"
bits = 9
sc = scenes.sand_column(bits, (64, 154, 64), min_corner=(224, 12, 224))
v0 = (1.5 * speed, -4.0 * speed, 0.8 * speed)
sc["models"][0]["v0"] = v0
n = scenes.total_particles(sc)
dt = 1e-4
other = f"a second oracle run on {3 * threads // 4} threads" if self_mode else "the HIP engine"
print(f"{n} sand particles, 512^3, thrown at the floor at {v0} m/s ({abs(v0[1]) * dt * 512:.2f} cells per substep), dt {dt}, oracle on {threads} threads; deviation of {other} from the oracle, relative = |dx| / max |coordinate| of the particle")
hip = build_engine(sc, api=oracle_api_threads(3 * threads // 4)) if self_mode else build_engine(sc)
ora = build_engine(sc, api=oracle_api_threads(threads))
hip.initial_setup()
ora.initial_setup()
done, worst, t_ora = 0, 0.0, 0.0
while done < nsteps:
    k = min(every, nsteps - done)
    hip.run_fixed(k, dt)
    t0 = time.time()
    ora.run_fixed(k, dt)
    t_ora += time.time() - t0
    done += k
    xh, sh, lh = hip.retrieve_state(0)
    xo, so, lo = ora.retrieve_state(0)
    idx, _ = match(xo.astype(np.float64), xh.astype(np.float64))
    dx = np.abs(xh[idx].astype(np.float64) - xo.astype(np.float64)).max(axis=1)
    rel = dx / np.abs(xo).max(axis=1)
    bh, bo = to_b(sh, not self_mode), to_b(so, False)
    db = (np.abs(bh[idx] - bo).max(axis=1) / np.maximum(1.0, np.abs(bo).max(axis=1))).max()
    ch, co = hip.counts(), ora.counts()
    same = (ch.particle_blocks, ch.neighbor_blocks, ch.exterior_blocks) == (co.particle_blocks, co.neighbor_blocks, co.exterior_blocks)
    q = np.quantile(rel, [0.5, 0.99, 0.9999])
    worst = max(worst, float(rel.max()))
    print(f"  substep {done:4d}: positions max {rel.max():.3e} (absolute {dx.max():.2e} = {dx.max() * 512:.4f} cells), median {q[0]:.1e}, 99 % {q[1]:.1e}, 99.99 % {q[2]:.1e}, above 1e-5: {int((rel >= 1e-5).sum())} of {n}; "
          f"b {db:.2e}, log Jp {np.abs(lh[idx] - lo).max():.2e} (largest |log Jp| {np.abs(lo).max():.2f}); blocks {ch.particle_blocks} / {ch.neighbor_blocks} / {ch.exterior_blocks} {'equal' if same else 'DIFFER'}; "
          f"lowest particle y = {xo[:, 1].min() * 512:.2f} cells", flush=True)
print(f"  oracle {t_ora:.0f} s for {nsteps} substeps; worst position deviation {worst:.3e}")
hip.close()
ora.close()
