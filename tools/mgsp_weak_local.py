"""The weak-scaling workload of `bench.py --gpus N` at FULL size with every rank as a context on ONE GPU (in-process transport of
the C++ group driver: device-to-device copies instead of RCCL): N touching C3 columns, 40.1 M particles each.  Reports halo
sizes, the self-check over all ranks and the wall time per substep (the contexts share the GPU, so that is ~N x one rank's).
usage: mgsp_weak_local.py [world=2] [steps=20]"""
import sys
import threading
import time

sys.path.insert(0, "/root/repo")
from claymore_amd import scenes
from claymore_amd.mgsp import LocalGroup, MgspGroupRank

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lg = LocalGroup(world)
ranks = []
for r in range(world):
    sc = scenes.sand_columns_rank(r, world)
    ranks.append(MgspGroupRank(sc, r, world, device=0, local_group=lg, prepartitioned=True))
    del sc
lg.create()
dt = 1e-4
out, errs = [None] * world, []


def work(r):
    try:
        sim = ranks[r]
        sim.initial_setup()
        sim.run_fixed(5, dt)
        t0 = time.perf_counter()
        sim.run_fixed(steps, dt)
        el = time.perf_counter() - t0
        c, d = sim.eng.counts(), sim.eng.diagnostics()
        out[r] = dict(ms_per_step=1e3 * el / steps, g2p2g_ms=sim.g2p2g_ms_avg, halo_blocks_sent=sum(sim.send_counts), halo_particle_blocks=sim.n_halo_blocks,
                      blocks=sim.block_counts(), particles=int(c.particles[0]), lost=int(d.lost_particles), discarded=int(d.discarded_p2g), n_local=sim.n_local)
    except Exception as e:  # noqa: BLE001
        errs.append(repr(e))


th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
for t in th:
    t.start()
for t in th:
    t.join()
for r in ranks:
    r.close()
print("errors:", errs)
for r, o in enumerate(out):
    print(f"rank {r}: {o}")
if not errs:
    tot = sum(o["particles"] for o in out)
    print(f"world {world}: {tot} particles bucketed of {sum(o['n_local'] for o in out)}, lost {sum(o['lost'] for o in out)}, discarded {sum(o['discarded'] for o in out)}, "
          f"wall {max(o['ms_per_step'] for o in out):.3f} ms per substep for all ranks on one GPU")
