"""Single-GPU proxy of the STRONG-scaling curve of BASELINE's metric (40.1 M-particle C3 column, 512^3, 1/2/4/8 ranks): the column is cut
into N equal-count slabs (the static particle partition of `bench.py --gpus N`), every rank is a context on the ONE GPU of the box
(in-process transport of the C++ group driver: device-to-device copies instead of RCCL), all ranks step concurrently.  The contexts
share the GPU, so the wall time per substep is the SUM of the ranks' work; wall / N is what a rank would take on a GPU of its own if
the exchange were free and nothing else were exposed - a lower bound for the N-GPU substep, i.e. N * T1 / wall is an UPPER bound for
the strong-scaling speed-up.  What the proxy does show: how much work the partitioning adds (halo blocks processed on two ranks, smaller
launches, tagging and exchange kernels), the halo share, and the host time a rank's thread spends per substep.
No multi-GPU hardware curve exists for this engine (the build box has one GPU).
Round 6: the SHAPE of the partition is a parameter - y slabs (the longest axis of the C3 column: what bench.py --gpus N used
so far), x or z slabs and 2 x 2 x 2 octants (the reference's scenarios put their per-GPU bodies side by side in x / z, Projects/MGSP/mgsp.cu:52-57,
:67-72), x-z columns - and every shape is measured at rest AND in the flow (`flow` substeps run first: the column has collapsed into a pile,
where thin horizontal slabs turn into wide pancakes).
usage: mgsp_strong_local.py [steps=20] [worlds=1,2,4,8] [shapes=y,x,octants] [flow=0]"""
import sys
import threading
import time

sys.path.insert(0, "/root/repo")
from claymore_amd import scenes
from claymore_amd.engine import build_engine
from claymore_amd.mgsp import LocalGroup, MgspGroupRank

from claymore_amd import mgsp as mgsp_mod

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
worlds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
shapes = (sys.argv[3] if len(sys.argv) > 3 else "y").split(",")
flow = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sc = scenes.sand_column(9)
n_total = scenes.total_particles(sc)
dt = sc["dt"]
t1 = None
print(f"# C3: {n_total} sand particles, 512^3; {'at rest' if not flow else f'in the flow ({flow} substeps first)'}; warm-up 5 + {steps} timed substeps; every rank a context on one GPU")
for shape, world in [(None, 1)] * (1 in worlds) + [(sh, w) for sh in shapes for w in worlds if w > 1]:
    mgsp_mod.ALIGN_TO_BLOCKS = bool(shape) and shape.endswith("+aligned")   # "y+aligned": the cut planes moved to particle-block faces (scenes.split_slabs)
    shape = shape[:-len("+aligned")] if mgsp_mod.ALIGN_TO_BLOCKS else shape
    mgsp_mod.PARTITION_SHAPE = shape
    if world == 1:
        eng = build_engine(sc)
        eng.initial_setup()
        eng.run_fixed(5 + flow, dt)
        t0 = time.perf_counter()
        eng.run_fixed(steps, dt)
        wall = 1e3 * (time.perf_counter() - t0) / steps
        tm, c = eng.timers(), eng.counts()
        print(f"world 1 (plain engine, device-side substep loop): wall {wall:.3f} ms per substep, device {tm.total_ms:.3f} (g2p2g {tm.g2p2g_ms:.3f}, partition {tm.partition_ms:.3f}), "
              f"host gap {wall - tm.total_ms:.3f}; particle blocks {c.particle_blocks}")
        t1 = wall
        eng.close()
        continue
    lg = LocalGroup(world)
    ranks = [MgspGroupRank(sc, r, world, device=0, local_group=lg) for r in range(world)]
    lg.create()
    out, errs = [None] * world, []

    def work(r):
        try:
            sim = ranks[r]
            sim.initial_setup()
            sim.run_fixed(5 + flow, dt)
            t0 = time.perf_counter()
            c0 = time.thread_time()
            sim.run_fixed(steps, dt)
            el = time.perf_counter() - t0
            cpu = time.thread_time() - c0
            c, d = sim.eng.counts(), sim.eng.diagnostics()
            out[r] = dict(ms=1e3 * el / steps, cpu_ms=1e3 * cpu / steps, g2p2g_ms=sim.g2p2g_ms_avg, sent=sum(sim.send_counts), halo_pb=sim.n_halo_blocks,
                          pb=c.particle_blocks, nb=c.neighbor_blocks, particles=int(c.particles[0]), lost=int(d.lost_particles), disc=int(d.discarded_p2g), n=sim.n_local)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for r in ranks:
        r.close()
    if errs:
        print(f"{shape} world {world}: errors {errs}")
        continue
    wall = max(o["ms"] for o in out)
    tot = sum(o["particles"] for o in out)
    assert tot == n_total and sum(o["lost"] for o in out) == 0 and sum(o["disc"] for o in out) == 0, out
    pb = sum(o["pb"] for o in out)
    print(f"shape {shape}{' block-aligned' if mgsp_mod.ALIGN_TO_BLOCKS else ''} {scenes.PARTITION_SHAPES[shape](world)} world {world}: wall {wall:.3f} ms per substep for ALL ranks on one GPU -> {wall / world:.3f} per rank (lower bound of the {world}-GPU substep); "
          f"work vs 1 rank x{wall / t1:.3f}; speed-up bound {world * t1 / wall:.2f} of {world}" if t1 else f"world {world}: wall {wall:.3f}")
    print(f"         particle blocks {pb} over all ranks; per rank: particles {min(o['n'] for o in out)}-{max(o['n'] for o in out)}, particle blocks {min(o['pb'] for o in out)}-{max(o['pb'] for o in out)}, "
          f"halo particle blocks {min(o['halo_pb'] for o in out)}-{max(o['halo_pb'] for o in out)} ({100.0 * sum(o['halo_pb'] for o in out) / pb:.0f} % of all), "
          f"grid blocks sent {min(o['sent'] for o in out)}-{max(o['sent'] for o in out)} ({max(o['sent'] for o in out) * 1036 / 1e6:.1f} MB max), host thread CPU time {max(o['cpu_ms'] for o in out):.3f} ms per substep; all {tot} particles accounted for")
