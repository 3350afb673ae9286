"""Long runs of the bench scenes with the self-check (particle count, lost / discarded counters, grid mass, finite positions) every
`chunk` substeps.  usage: long_run_check.py scene steps chunk [drop]   (scene: sphere5m | fluid12m | sand40m; drop: mpm_config.drop_overflow = 1)"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from claymore_amd import scenes
from claymore_amd.engine import build_engine

name, steps, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sc = {"sphere5m": lambda: scenes.sphere_drop(), "fluid12m": lambda: scenes.fluid_dam(10, (32, 192, 256)), "sand40m": lambda: scenes.sand_column(9)}[name]()
n = scenes.total_particles(sc)
if len(sys.argv) > 4:
    sc["config"]["drop_overflow"] = 1
eng = build_engine(sc)
eng.initial_setup()
mass = n * eng.model_mass(0)
done = 0
while done < steps:
    eng.run_fixed(chunk, sc["dt"])
    done += chunk
    c, d, tot, t = eng.counts(), eng.diagnostics(), eng.grid_totals(), eng.timers()
    x = eng.retrieve_positions(0)
    n_now = n - d.dropped_particles
    mass = n_now * eng.model_mass(0)
    ok = c.particles[0] == n_now and d.lost_particles == 0 and d.discarded_p2g == 0 and abs(tot[0] - mass) / mass < 1e-4 and np.isfinite(x).all()
    print(f"{name} step {done} (t = {done * sc['dt']:.4f} s): {'ok' if ok else 'FAILED'} particles {c.particles[0]}/{n} lost {d.lost_particles} discarded {d.discarded_p2g} dropped {d.dropped_particles} "
          f"mass err {abs(tot[0] - mass) / mass:.1e} blocks {c.particle_blocks} y range [{x[:, 1].min():.3f}, {x[:, 1].max():.3f}] g2p2g {t.g2p2g_ms:.3f} ms", flush=True)
eng.close()
