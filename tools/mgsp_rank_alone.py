"""One rank's share of the strong-scaling layout ALONE on the GPU: the C3 column is cut into N equal-count slabs and a middle slab is run
(a) by the plain engine (mpm_run_fixed: device-side substep loop) and (b) by the group driver with world = 1 on the RCCL transport
(mpm_group_run_fixed: halo-first / interior split, padded key export, ncclAllGather, tagging, one host synchronisation per substep - everything a
rank does except talk to a peer).  T1 / (b) is what strong scaling would reach if the exchange were free and perfectly hidden; (b) - (a) is the price of
the multi-GPU loop itself.  usage: mgsp_rank_alone.py [steps=20] [worlds=2,4,8]"""
import sys
import time

sys.path.insert(0, "/root/repo")
from claymore_amd import scenes
from claymore_amd.engine import build_engine
from claymore_amd.mgsp import MgspGroupRank, partition_scene

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
worlds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,4,8").split(",")]
sc = scenes.sand_column(9)
dt = sc["dt"]


def plain(scene):
    eng = build_engine(scene)
    eng.initial_setup()
    eng.run_fixed(5, dt)
    t0 = time.perf_counter()
    eng.run_fixed(steps, dt)
    wall = 1e3 * (time.perf_counter() - t0) / steps
    tm = eng.timers()
    c = eng.counts()
    eng.close()
    return wall, tm, c.particle_blocks


def group(scene):
    sim = MgspGroupRank(scene, 0, 1, device=0, prepartitioned=True)
    sim.initial_setup()
    sim.run_fixed(5, dt)
    t0 = time.perf_counter()
    sim.run_fixed(steps, dt)
    wall = 1e3 * (time.perf_counter() - t0) / steps
    g = sim.g2p2g_ms_avg
    sim.close()
    return wall, g


w1, tm1, pb1 = plain(sc)
print(f"# C3 whole column: plain engine {w1:.3f} ms per substep (device {tm1.total_ms:.3f}, g2p2g {tm1.g2p2g_ms:.3f}); {pb1} particle blocks; warm-up 5 + {steps} timed substeps")
gw, gg = group(sc)
print(f"  world 1 group driver (RCCL, one rank) on the whole column: {gw:.3f} ms per substep (g2p2g {gg:.3f})")
for world, align in [(w, al) for w in worlds for al in (False, True)]:
    # equal-count slabs: a middle slab; block-aligned slabs (scenes.split_slabs, bench.py's default): the piece with the most particles
    local = partition_scene(sc, world // 2, world, align=False) if not align else max((partition_scene(sc, r, world, align=True) for r in range(world)), key=scenes.total_particles)
    n = scenes.total_particles(local)
    a, tma, pba = plain(local)
    b, bg = group(local)
    print(f"1/{world} slab{', block-aligned (the largest piece)' if align else ''} ({n} particles, {pba} particle blocks) alone: plain engine {a:.3f} ms per substep (device {tma.total_ms:.3f}, g2p2g {tma.g2p2g_ms:.3f}, partition {tma.partition_ms:.3f}) | "
          f"group driver {b:.3f} (g2p2g {bg:.3f}) | T1 / plain = {w1 / a:.2f}, T1 / group = {w1 / b:.2f} of {world}")
