"""Why a few violent fuzz cases (tools/fuzz_parity.py 200 7 6 300 012) end 1-2e-5 apart: the z trajectory of one particle of case 146's free-flying body,
substep by substep, in both engines.  v_z dt = 8920.497 ulp(0.5): the sum x + v dt sits 0.003 ulp from a rounding tie, the two engines' gathered velocities
differ by 3e-7 (summation order), so one rounds every step up and the other down - 1 ulp per substep, the same way for every particle of the lattice in
that binade (they share their mantissa alignment).  Both are correct fp32 roundings; the difference grows linearly (1.2e-7 of the domain per substep)."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as g
from claymore_amd.engine import build_engine
from fuzz_scenes import random_scene
from parity_util import match
from oracle_ffi import oracle_api
g.build_oracle()
rng = np.random.default_rng(7)
for case in range(147):
    sc, nsteps = random_scene(rng, case, 6.0, 300, (0, 1, 2))
sc = dict(sc, models=[sc["models"][0]])
x0 = sc["models"][0]["xyz"]
def traj(api, n):
    eng = build_engine(sc, api=api); eng.initial_setup()
    out = [x0.copy()]
    for _ in range(n):
        eng.run_fixed(1, 1e-4)
        x = eng.retrieve_positions(0)
        idx, _ = match(out[-1].astype(np.float64), x.astype(np.float64))
        out.append(x[idx])
    eng.close(); return np.stack(out)
n = 8
th, to = traj(None, n), traj(oracle_api(), n)
d = (th[-1, :, 2].astype(np.float64) - to[-1, :, 2])
p = int(np.argmax(np.abs(d)))
q = int(np.argmin(np.abs(d) + (np.abs(d) == 0) * 1.0)) if (d != 0).any() else 0
c = np.float32(sc["models"][0]["v0"][2]) * np.float32(1e-4)
print("v0z*dt (fp32) =", repr(float(c)), "ulp(0.5) = 5.96e-8")
for name, pid in (("worst", p), ("a same one", int(np.argmin(np.abs(d))))):
    print(f"particle {name}: z0 = {x0[pid,2]!r}")
    for s in range(1, n + 1):
        ih = float(th[s, pid, 2]) - float(th[s - 1, pid, 2]); io = float(to[s, pid, 2]) - float(to[s - 1, pid, 2])
        print(f"  step {s}: z HIP {float(th[s,pid,2])!r} oracle {float(to[s,pid,2])!r}  increment HIP {ih:.10e} oracle {io:.10e}  (x incr HIP {float(th[s,pid,0])-float(th[s-1,pid,0]):.10e} oracle {float(to[s,pid,0])-float(to[s-1,pid,0]):.10e})")
