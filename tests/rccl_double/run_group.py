#!/usr/bin/env python
"""Drive the RCCL transport of the C++ group driver (claymore_amd/csrc/mpm_group.inc: mpm_group_create with a unique id,
ncclCommInitRank, the grouped ncclSend / ncclRecv of the halo exchange, the ncclAllGather of the block keys, the ncclAllReduce
of the maximum velocity) with WORLD > 1 ranks on ONE GPU: every rank a thread with its own engine context, the collectives
served by the in-process double tests/rccl_double/rccl_double.cpp (MPM_RCCL_LIBRARY must point at its build; a separate process
per run because the library loads its collective library once).  The union of the ranks' particles must follow the single-engine
CPU oracle.  Prints "OK ..." on success.

    MPM_RCCL_LIBRARY=/path/to/librccl_double.so python tests/rccl_double/run_group.py WORLD KIND   (KIND: fixed | substeps | adaptive)
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


def main():
    world, kind = int(sys.argv[1]), sys.argv[2]
    assert os.environ.get("MPM_RCCL_LIBRARY"), "MPM_RCCL_LIBRARY is not set"
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=0.5, speed=2.0, youngs=2e4)
    nsteps, dt = 60, 1e-4
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
        ora = build_engine(sc, api=oracle_api())
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
    assert worst < 1e-5, worst
    print(f"OK world {world} {kind}: {results[0][3]} substeps, worst relative position error {worst:.2e}, grid blocks sent {[r[1] for r in results]}, halo particle blocks {[r[2] for r in results]}")


if __name__ == "__main__":
    main()
