"""Pipeline-level pins of the CPU oracle (no reference run exists to compare with): conservation and consistency
invariants of SURVEY.md section 8c."""
import ctypes as C

import numpy as np
import pytest

from claymore_amd import _ffi, scenes
from claymore_amd.engine import build_engine
from oracle_ffi import oracle_api


def _mass(scene):
    tot = 0.0
    for m in scene["models"]:
        p = m["params"]
        tot += m["xyz"].shape[0] * np.float32(p["volume"]) * np.float32(p.get("rho", 1e3))
    return float(tot)


def test_two_spheres_invariants():
    api = oracle_api()
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=3.0, speed=0.5)
    n = scenes.total_particles(sc)
    eng = build_engine(sc, api=api)
    eng.initial_setup()
    tot0 = eng.grid_totals()
    c0 = eng.counts()
    assert c0.particle_blocks <= c0.neighbor_blocks <= c0.exterior_blocks
    assert sum(c0.particles[i] for i in range(c0.model_count)) == n
    assert abs(tot0[0] - _mass(sc)) / _mass(sc) < 1e-5          # rasterize conserves mass
    assert abs(tot0[1]) < 1e-6 * _mass(sc)                       # +v and -v spheres cancel
    dt = 1e-4
    g = -9.8
    for step in range(1, 31):
        mv2 = eng.grid_update(dt)
        assert np.isfinite(mv2) and mv2 > 0
        eng.g2p2g(dt, dt)
        c = eng.rebuild_partition()
        assert api.raw.mpmo_check_table(eng.ctx) == 0           # partition.query(active_keys[i]) == i
        assert sum(c.particles[i] for i in range(c.model_count)) == n   # particle count conserved
        tot = eng.grid_totals()
        assert abs(tot[0] - _mass(sc)) / _mass(sc) < 2e-5       # P2G conserves mass
        # momentum: x stays ~0 (internal forces cancel), y = gravity impulse accumulated so far
        assert abs(tot[1]) < 2e-4 * _mass(sc)
        assert abs(tot[2] - g * dt * step * _mass(sc)) < 2e-3 * abs(g * dt * step * _mass(sc)) + 1e-7 * _mass(sc)
    x0 = eng.retrieve_positions(0)
    x1 = eng.retrieve_positions(1)
    # free flight of the centroids: x(t) = x0 +- v t
    t = 30 * dt
    assert abs(x0[:, 0].mean() - (sc["models"][0]["xyz"][:, 0].mean() + 0.5 * t)) < 2e-5
    assert abs(x1[:, 0].mean() - (sc["models"][1]["xyz"][:, 0].mean() - 0.5 * t)) < 2e-5
    eng.close()


@pytest.mark.parametrize("material", [_ffi.J_FLUID, _ffi.SAND, _ffi.NACC])
def test_other_materials_run_and_conserve(material):
    api = oracle_api()
    sc = scenes.sphere_drop(bits=6, radius_cells=5.0, center=(0.5, 0.5, 0.5), material=material)
    sc["models"][0]["params"] = {}
    n = scenes.total_particles(sc)
    eng = build_engine(sc, api=api)
    eng.initial_setup()
    m0 = eng.grid_totals()[0]
    eng.run_fixed(10, 1e-4)
    c = eng.counts()
    assert c.particles[0] == n
    assert abs(eng.grid_totals()[0] - m0) / m0 < 2e-5
    xyz, st, lj = eng.retrieve_state(0)
    assert np.isfinite(xyz).all() and np.isfinite(st).all()
    eng.close()


def test_adaptive_dt_matches_compute_dt():
    api = oracle_api()
    sc = scenes.two_spheres(bits=6, radius_cells=4.0, gap_cells=2.0, speed=3.0)
    eng = build_engine(sc, api=api)
    eng.initial_setup()
    dx = 1.0 / 64
    nd, mv = eng.substep(1e-4, 0.0, 1.0 / 24, 1e-3)
    assert abs(mv - 3.0) < 0.05                                  # max grid speed ~ |v0|
    assert abs(nd - min(1e-3, dx * 0.5 / mv)) < 1e-9
    eng.close()


def test_capacity_error_is_reported():
    api = oracle_api()
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=3.0)
    sc["config"]["max_blocks"] = 8
    sc["config"]["grow"] = 0          # fixed capacity: exceeding it is an error (gmpm_simulator.cuh:473-476)
    eng = build_engine(sc, api=api)
    with pytest.raises(Exception):
        eng.initial_setup()
    eng.close()
