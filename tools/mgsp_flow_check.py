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
