#!/usr/bin/env python
"""Drive the RCCL transport of the C++ group driver (claymore_amd/csrc/mpm_group.inc: mpm_group_create with a unique id,
ncclCommInitRank, the grouped ncclSend / ncclRecv of the halo exchange, the ncclAllGather of the block keys, the ncclAllReduce
of the maximum velocity) with WORLD > 1 ranks on ONE GPU: every rank a thread with its own engine context, the collectives
served by the in-process double tests/rccl_double/rccl_double.cpp (MPM_RCCL_LIBRARY must point at its build; a separate process
per run because the library loads its collective library once).  The union of the ranks' particles must follow the single-engine
CPU oracle.  Prints "OK ..." on success.

    MPM_RCCL_LIBRARY=/path/to/librccl_double.so python tests/rccl_double/run_group.py WORLD KIND   (KIND: fixed | substeps | adaptive | fixed-big | fixed-tightpad | plate | plate-fall | plate-stale | resume | fail)

KIND fail: rank 1 runs with a block capacity it outgrows after a few substeps (no growth): it must come back with MPM_ERR_CAPACITY, and so
must EVERY other rank, in the same substep (its status word travels in row 0 of the key all-gather) - nobody may be left waiting in a
collective.  Prints "OK ..." on success.
"""
import os
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402

from claymore_amd import scenes  # noqa: E402
from claymore_amd.engine import build_engine  # noqa: E402
from claymore_amd.mgsp import MgspGroupRank  # noqa: E402
from oracle_ffi import oracle_api  # noqa: E402
from parity_util import match  # noqa: E402


def run_failing_rank(world):
    """One rank outgrows its (fixed) block capacity in the middle of mpm_group_run_fixed: every rank returns MPM_ERR_CAPACITY, none hangs."""
    import copy
    from claymore_amd import _ffi
    from claymore_amd.engine import EngineError
    from claymore_amd.mgsp import partition_scene
    # a sand ball thrown at the floor: its block count wobbles by a few while it falls and doubles once it splashes (400-600 substeps in)
    sc = scenes.sphere_drop(bits=6, radius_cells=6.0, center=(0.5, 0.26, 0.5), material=_ffi.SAND)
    sc["models"][0]["params"] = {}
    sc["models"][0]["v0"] = (0.0, -4.0, 0.0)
    locals_ = [copy.deepcopy(partition_scene(sc, r, world)) for r in range(world)]
    probe = build_engine(locals_[1])                    # how many blocks does rank 1 start with?
    probe.initial_setup()
    ebc0 = probe.counts().exterior_blocks
    probe.close()
    locals_[1]["config"] = dict(locals_[1].get("config", {}), max_blocks=ebc0 + 8, grow=0)
    ident, have_id = {}, threading.Event()

    def bootstrap(raw):
        if raw is not None:
            ident["raw"] = raw
            have_id.set()
        else:
            assert have_id.wait(120)
        return ident["raw"]

    outcome, sims = [None] * world, [None] * world

    def work(rank):
        try:
            sim = sims[rank] = MgspGroupRank(locals_[rank], rank, world, device=0, bootstrap=bootstrap, prepartitioned=True)
            sim.initial_setup()
            sim.run_fixed(900, 1e-4)
            outcome[rank] = ("finished", "")
        except EngineError as e:
            outcome[rank] = (e.code, str(e))
        except Exception as e:  # noqa: BLE001
            outcome[rank] = ("exception", repr(e))

    threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), f"a rank is stuck in a collective: {outcome}"
    assert all(o is not None and o[0] == _ffi.MPM_ERR_CAPACITY for o in outcome), outcome
    assert "rank 1 reported" in outcome[0][1], outcome
    for s in sims:
        if s is not None:
            s.close()
    print(f"OK world {world} fail: every rank returned MPM_ERR_CAPACITY ({[o[1] for o in outcome]})")


def variant_api():
    """RUN_GROUP_LIBRARY: another build of the engine library (a mutant with an experiment switch) instead of the shipped one."""
    path = os.environ.get("RUN_GROUP_LIBRARY")
    if not path:
        return None
    import ctypes as C
    from claymore_amd import _ffi
    return _ffi.bind(C.CDLL(os.path.abspath(path), mode=os.RTLD_LOCAL | os.RTLD_NOW), "mpm_", hip=True)


def run_plate(world, fall=False, stale=False):
    """A rank whose particle blocks are ALL halo blocks at first and all interior later: a plate of elastic material two cells above a body
    that belongs to the other rank (no contact, but the same grid blocks), thrown upwards.  The windowed loop launches a substep's G2P2G passes before the host has seen the
    counts of the tagging they run on; the interior pass must not be skipped because the host's count of interior blocks (one tagging
    old) is still zero - those blocks would be done by neither pass and their particles would silently disappear (round 4: they did)."""
    from claymore_amd import _ffi
    assert world == 2
    bits = 7
    prm = {"volume": (1.0 / (1 << bits)) ** 3 / 8.0, "youngs_modulus": 5e3, "poisson_ratio": 0.4, "rho": 1e3}
    base = {"name": "plate_on_body", "bits": bits, "dt": 1e-4, "config": {"max_ppc": 128}}
    body = dict(base, models=[{"material": _ffi.FIXED_COROTATED, "xyz": scenes.lattice_box(bits, (44, 40, 44), (84, 58, 84)), "v0": (0.0, 0.0, 0.0), "params": dict(prm)}])
    plate = dict(base, models=[{"material": _ffi.FIXED_COROTATED, "xyz": scenes.lattice_box(bits, (52, 60, 52), (76, 62, 76)), "v0": (0.0, 5.0, 0.0), "params": dict(prm)}])
    if fall:   # the other way round: the plate starts 14 cells above the body - no block in common - and is thrown at it (no halo -> all halo -> contact)
        plate["models"][0]["xyz"] = scenes.lattice_box(bits, (52, 72, 52), (76, 74, 76))
        plate["models"][0]["v0"] = (0.0, -5.0, 0.0)
    locals_ = [body, plate]
    ident, have_id = {}, threading.Event()

    def bootstrap(raw):
        if raw is not None:
            ident["raw"] = raw
            have_id.set()
        else:
            assert have_id.wait(120)
        return ident["raw"]

    log, errors, sims, start, final = [[] for _ in range(world)], [], [None] * world, [None] * world, [None] * world
    bar = threading.Barrier(world)
    api = variant_api()
    if stale:
        assert api is not None and "MPM_HACK_STALE_INTERIOR" in api.build_info().decode(), "plate-stale needs RUN_GROUP_LIBRARY = a build with -DMPM_EXPERIMENT -DMPM_HACK_STALE_INTERIOR"

    def work(rank):
        try:
            sim = sims[rank] = MgspGroupRank(locals_[rank], rank, world, device=0, bootstrap=bootstrap, prepartitioned=True, api=api)
            sim.initial_setup()
            c0 = sim.eng.counts()
            start[rank] = (c0.particle_blocks, sim.n_halo_blocks)
            for _ in range(40):
                sim.run_fixed(10, 1e-4)
                c, d = sim.eng.counts(), sim.eng.diagnostics()
                log[rank].append((sum(c.particles[i] for i in range(c.model_count)), c.particle_blocks, sim.n_halo_blocks, d.lost_particles, d.discarded_p2g))
                bar.wait(120)
            final[rank] = sim.local_state()[0][0]
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            bar.abort()

    threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in a collective"
    if stale:
        # Round 4's bug put back (the interior pass skipped on a block count one tagging old): the plate's particles vanish with nothing counted
        # as lost.  The library's own books must catch that - MPM_ERR_INTERNAL on the rank that lost them, and on its peer in the same
        # substep (the status word of the key all-gather) - where round 4 needed bench.py's end-of-run self-check to notice.
        from claymore_amd import _ffi
        for s_ in sims:
            if s_ is not None:
                s_.close()
        assert len(errors) == world, f"the books did not trip on every rank: {errors} / {[e[-1:] for e in log]}"
        text = {r: e for r, e in errors}
        assert all(f"mpm status {_ffi.MPM_ERR_INTERNAL}:" in e for e in text.values()), errors
        assert "do not balance" in text[1] and "rank 1 reported" in text[0], errors
        steps_done = 10 * len(log[1])
        print(f"OK world {world} plate-stale: the stale-count mutant tripped the library's books after {steps_done}+ substeps: {text[1][:200]} | peer: {text[0][:160]}")
        return
    assert not errors, errors
    n_local = [s.n_local for s in sims]
    for s in sims:
        s.close()
    last = log[1][-1]
    if fall:
        assert start[1][1] == 0 and start[1][0] > 0, f"the plate shares blocks with the body at the start: (particle blocks, halo particle blocks) = {start[1]}"
        assert max(e[2] for e in log[1]) == max(e[1] for e in log[1] if e[2]), f"the plate never became all halo blocks: {[e[1:3] for e in log[1]]}"
    else:
        assert start[1][1] == start[1][0] and start[1][0] > 0, f"the plate's blocks are not all halo blocks at the start: (particle blocks, halo particle blocks) = {start[1]}"
        assert last[2] == 0, f"the plate never left the body: {last}"
    for r in range(world):
        bad = [(10 * (k + 1), e) for k, e in enumerate(log[r]) if e[0] != n_local[r] or e[3] or e[4]]
        assert not bad, f"rank {r} ({n_local[r]} particles) lost particles: (substep, (bucketed, particle blocks, halo particle blocks, lost, discarded)) {bad[:4]}"
    # the physics: both bodies as two models of ONE engine, the same substeps
    both = dict(base, models=[body["models"][0], plate["models"][0]])
    one = build_engine(both)
    one.initial_setup()
    one.run_fixed(10 * len(log[0]), 1e-4)
    worst = 0.0
    for m in range(2):
        xo = one.retrieve_state(m)[0]
        xm = final[m]
        assert xm.shape == xo.shape, (xm.shape, xo.shape)
        idx, _ = match(xo.astype(np.float64), xm.astype(np.float64))
        worst = max(worst, float((np.abs(xm[idx].astype(np.float64) - xo).max(axis=1) / np.abs(xo).max(axis=1)).max()))
    one.close()
    assert worst < 1e-5, worst
    print(f"OK world {world} plate{'-fall' if fall else ''}: worst relative position error against one engine {worst:.2e}; every particle bucketed through {10 * len(log[0])} substeps; the plate's halo particle blocks went {start[1][1]} (of {start[1][0]} particle blocks) -> {[e[2] for e in log[1]][:6]} ...")


def run_resume(world):
    """Restart of a group: 30 substeps, every rank saves its checkpoint, 30 more (A).  Then NEW contexts and a NEW communicator: initial
    set-up, every rank loads its checkpoint, mpm_group_resume rebuilds the tagging, 30 substeps (B).  B must be A (to the float atomics'
    last bits) and the CPU oracle's uninterrupted 60 substeps."""
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=0.5, speed=2.0, youngs=2e4)
    dt = 1e-4

    def group_run(body):
        ident, have_id = {}, threading.Event()

        def bootstrap(raw):
            if raw is not None:
                ident["raw"] = raw
                have_id.set()
            else:
                assert have_id.wait(120)
            return ident["raw"]

        out, errors = [None] * world, []

        def work(rank):
            sim = None
            try:
                sim = MgspGroupRank(sc, rank, world, device=0, bootstrap=bootstrap)
                sim.initial_setup()
                out[rank] = body(rank, sim)
            except Exception as e:  # noqa: BLE001
                errors.append((rank, repr(e)))
            finally:
                if sim is not None:
                    sim.close()

        threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in threads), "a rank is stuck in a collective"
        assert not errors, errors
        return out

    def first(rank, sim):
        sim.run_fixed(30, dt)
        ck = sim.save_checkpoint().copy()
        sim.run_fixed(30, dt)
        return ck, sim.local_state(), sim.n_halo_blocks

    def second(rank, sim):
        sim.load_checkpoint(saved[rank][0])
        sim.run_fixed(30, dt)
        return sim.local_state(), sim.n_halo_blocks

    saved = group_run(first)
    again = group_run(second)
    assert max(x[2] for x in saved) > 0, "no halo: the exchange was not exercised"
    worst = 0.0
    for r in range(world):
        for m in range(len(sc["models"])):
            a, b = saved[r][1][m][0].astype(np.float64), again[r][0][m][0].astype(np.float64)
            assert a.shape == b.shape, (a.shape, b.shape)
            if a.shape[0]:   # (the order of the particles in the bins need not be the same: the runs' atomics order the lists differently)
                idx, _ = match(a, b)
                worst = max(worst, float((np.abs(b[idx] - a).max(axis=1) / np.abs(a).max(axis=1)).max()))
    ora = build_engine(sc, api=oracle_api())
    ora.initial_setup()
    ora.run_fixed(60, dt)
    worst_o = 0.0
    for m in range(len(sc["models"])):
        xo = ora.retrieve_state(m)[0]
        xm = np.concatenate([again[r][0][m][0] for r in range(world)])
        idx, _ = match(xo.astype(np.float64), xm.astype(np.float64))
        worst_o = max(worst_o, float((np.abs(xm[idx].astype(np.float64) - xo).max(axis=1) / np.abs(xo).max(axis=1)).max()))
    ora.close()
    assert worst < 2e-6 and worst_o < 1e-5, (worst, worst_o)
    print(f"OK world {world} resume: restarted run against the uninterrupted one {worst:.2e}, against the oracle's 60 substeps {worst_o:.2e}; halo particle blocks {[x[2] for x in saved]} / {[x[1] for x in again]}")


def main():
    world, kind = int(sys.argv[1]), sys.argv[2]
    assert os.environ.get("MPM_RCCL_LIBRARY"), "MPM_RCCL_LIBRARY is not set"
    if kind == "fail":
        return run_failing_rank(world)
    if kind == "resume":
        return run_resume(world)
    if kind in ("plate", "plate-fall", "plate-stale"):
        return run_plate(world, fall=kind == "plate-fall", stale=kind == "plate-stale")
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=0.5, speed=2.0, youngs=2e4)
    nsteps, dt = 60, 1e-4
    big = kind.endswith("-big")
    if big:
        # 2 x 2.1 M particles in contact: launches of a fraction of a millisecond, megabytes of halo per substep - long enough for a forgotten
        # stream dependency (the double is stream-ordered, like RCCL) to show; compared with the plain single-GPU engine, not the CPU oracle
        kind = kind[:-4]
        sc = scenes.two_spheres(bits=8, radius_cells=40.0, gap_cells=0.5, speed=2.0, youngs=2e4)
        nsteps = 40
    ident, have_id = {}, threading.Event()

    def bootstrap(raw):            # what bench.py does with one gloo broadcast
        if raw is not None:
            ident["raw"] = raw
            have_id.set()
        else:
            assert have_id.wait(120)
        return ident["raw"]

    results, errors, sims = [None] * world, [], [None] * world

    def work(rank):
        try:
            sim = sims[rank] = MgspGroupRank(sc, rank, world, device=0, bootstrap=bootstrap)   # ncclCommInitRank meets the other ranks here
            sim.initial_setup()
            steps = nsteps
            if kind == "fixed":
                sim.run_fixed(nsteps, dt)
            elif kind == "substeps":
                for _ in range(nsteps):
                    sim.substep(dt, dt)
            else:
                steps = sim.main_loop(2, 240, dt)
            results[rank] = (sim.local_state(), sum(sim.send_counts), sim.n_halo_blocks, steps)
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in a collective"
    assert not errors, errors
    for s in sims:
        s.close()
    assert max(r[1] for r in results) > 0 and max(r[2] for r in results) > 0, "no halo: the exchange was not exercised"

    # the single-engine oracle on the whole scene
    if kind == "adaptive":
        from test_mgsp_gpu import _oracle_mgsp_main_loop
        want, steps_o = _oracle_mgsp_main_loop(sc, 2, 240, dt)
        assert [r[3] for r in results] == [steps_o] * world, ([r[3] for r in results], steps_o)
        want = [w[0] for w in want]
    else:
        ora = build_engine(sc) if big else build_engine(sc, api=oracle_api())
        ora.initial_setup()
        ora.run_fixed(nsteps, dt)
        want = [ora.retrieve_state(m)[0] for m in range(len(sc["models"]))]
        ora.close()
    worst = 0.0
    for m, xo in enumerate(want):
        xm = np.concatenate([r[0][m][0] for r in results])
        assert xm.shape == xo.shape, (xm.shape, xo.shape)
        idx, _ = match(xo.astype(np.float64), xm.astype(np.float64))
        rel = np.abs(xm[idx].astype(np.float64) - xo).max(axis=1) / np.abs(xo).max(axis=1)
        worst = max(worst, float(rel.max()))
    assert worst < (1e-6 if big else 1e-5), worst   # (big: 2.3e-7 is the sum-order difference; a mutant without the comm stream's wait for the collect kernel reaches 1.6e-6)
    print(f"OK world {world} {kind}{' (big, vs the single-GPU engine)' if big else ''}: {results[0][3]} substeps, worst relative position error {worst:.2e}, grid blocks sent {[r[1] for r in results]}, halo particle blocks {[r[2] for r in results]}")


if __name__ == "__main__":
    main()
