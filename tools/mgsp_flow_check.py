"""Per-rank particle accounting of the strong-scaling layout deep in the flow: WORLD ranks as threads on one GPU (RCCL branch of the group driver
through the in-process double, MPM_RCCL_LIBRARY), the full C3 column, STEPS substeps of mpm_group_run_fixed in chunks; after every chunk each
rank's bucketed particle count against its share, and its diagnostics.  usage: mgsp_flow_check.py WORLD [STEPS=3030] [CHUNK=500] [SCENE=c3|c4|c5|c3small]"""
import os
import sys
import threading

sys.path.insert(0, "/root/repo")
from claymore_amd import scenes
from claymore_amd.mgsp import MgspGroupRank

world = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3030
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 500
which = sys.argv[4] if len(sys.argv) > 4 else "c3"
assert os.environ.get("MPM_RCCL_LIBRARY"), "MPM_RCCL_LIBRARY is not set"
sc = {"c3": lambda: scenes.sand_column(9), "c4": scenes.two_spheres_c4, "c5": lambda: scenes.fluid_dam(10), "c3small": lambda: scenes.scaled_sand_column(9, 1.0 / 16.0)}[which]()
dt = sc["dt"]
ident, have_id = {}, threading.Event()


def bootstrap(raw):
    if raw is not None:
        ident["raw"] = raw
        have_id.set()
    else:
        assert have_id.wait(120)
    return ident["raw"]


bar = threading.Barrier(world)
lines, errors = [[] for _ in range(world)], []
sums = [None] * world   # per rank: (n, sum of positions, sum of squared heights) at the end


def work(rank):
    try:
        sim = MgspGroupRank(sc, rank, world, device=0, bootstrap=bootstrap)
        sim.initial_setup()
        done = 0
        while done < steps:
            n = min(chunk, steps - done)
            sim.run_fixed(n, dt)
            done += n
            c, d = sim.eng.counts(), sim.eng.diagnostics()
            have = sum(c.particles[i] for i in range(c.model_count))
            lines[rank].append(f"step {done}: rank {rank} bucketed {have} of {sim.n_local} ({have - sim.n_local:+d}) lost {d.lost_particles} discarded {d.discarded_p2g} dropped {d.dropped_particles} "
                               f"overflow {d.overflow_flags} blocks {c.particle_blocks}/{c.neighbor_blocks}/{c.exterior_blocks} halo blocks {sim.n_halo_blocks}")
            bar.wait(600)
        import numpy as np
        acc = [0, np.zeros(3), 0.0]
        for xyz, *_ in sim.local_state():
            x = xyz.astype(np.float64)
            acc[0] += x.shape[0]
            acc[1] += x.sum(axis=0)
            acc[2] += float((x[:, 1] ** 2).sum())
        sums[rank] = acc
        sim.close()
    except Exception as e:  # noqa: BLE001
        errors.append((rank, repr(e)))
        bar.abort()


th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
for t in th:
    t.start()
for t in th:
    t.join(timeout=1500)
print(f"# scene {which}, world {world}, {steps} substeps, MPM_GROUP_DEFER={os.environ.get('MPM_GROUP_DEFER', '(unset)')}")
for k in range(max(len(x) for x in lines)):
    for r in range(world):
        if k < len(lines[r]) and ("+0)" not in lines[r][k] or k == len(lines[r]) - 1):
            print(lines[r][k])
print("errors:", errors)

if not errors and all(x is not None for x in sums):
    # the same substeps on ONE engine: centre of mass and the second moment of the heights of all particles must agree (sums over 4e7 particles:
    # the order of the float additions on the grid differs between the runs, a wrong or stale halo block would not average out)
    import numpy as np
    from claymore_amd.engine import build_engine
    n = sum(x[0] for x in sums)
    com = sum(x[1] for x in sums) / n
    h2 = sum(x[2] for x in sums) / n
    eng = build_engine(sc)
    eng.initial_setup()
    eng.run_fixed(steps, dt)
    n1, s1, q1 = 0, np.zeros(3), 0.0
    for m in range(len(sc["models"])):
        x = eng.retrieve_positions(m).astype(np.float64)
        n1 += x.shape[0]
        s1 += x.sum(axis=0)
        q1 += float((x[:, 1] ** 2).sum())
    eng.close()
    com1, h21 = s1 / n1, q1 / n1
    print(f"centre of mass, {world} ranks: {com}  one engine: {com1}  difference {np.abs(com - com1).max():.3e}; mean squared height {h2:.9f} vs {h21:.9f} (relative {abs(h2 - h21) / h21:.2e}); particles {n} vs {n1}")
