"""Level-set collision object of the MGSP grid update.

CPU: the oracle's restatement of SignedDistanceGrid::detect_and_resolve_collision (Projects/MGSP/boundary_condition.cuh:
164-248) against closed-form expectations; GPU: the HIP kernel against the oracle through the C ABI."""
import numpy as np
import pytest

from claymore_amd import scenes
from claymore_amd.engine import build_engine
from oracle_ffi import oracle_api
from parity_util import match


def _run(sc, steps, api=None):
    eng = build_engine(sc, api=api) if api is not None else build_engine(sc)
    eng.initial_setup()
    mv = []
    for _ in range(steps):
        mv.append(eng.grid_update(sc["dt"]))
        eng.g2p2g(sc["dt"], sc["dt"])
        eng.rebuild_partition()
    xyz = eng.retrieve_positions(0)
    tot = eng.grid_totals()
    eng.close()
    return xyz, np.array(mv), tot


@pytest.mark.parametrize("boundary", ["sticky", "slip", "separate"])
def test_oracle_obstacle_stops_the_sphere(boundary):
    """Without the object the sphere's lowest point falls through y = top of the obstacle; with it, no particle ends up deeper than
    about one cell inside the level set (grid velocities are resolved at nodes, particles interpolate)."""
    api = oracle_api()
    sc = scenes.sphere_on_obstacle(bits=6, radius_cells=4.0, obstacle_cells=5.0, boundary=boundary, speed=2.0)
    steps = 300
    xyz, mv, _ = _run(sc, steps, api)
    free = dict(sc)
    free.pop("collision")
    xyz_free, mv_free, _ = _run(free, steps, api)
    dx = 1.0 / 64
    c = np.array([0.5, 0.34, 0.5])
    depth = 5.0 * dx - np.linalg.norm(xyz.astype(np.float64) - c, axis=1)
    depth_free = 5.0 * dx - np.linalg.norm(xyz_free.astype(np.float64) - c, axis=1)
    assert depth_free.max() > 2.0 * dx          # the free sphere is well inside where the obstacle would be
    assert depth.max() < 1.25 * dx, depth.max() / dx
    # the reference's boundary overload reports 2 |v|^2 (mgmpm_kernels.cuh:365-373): before contact exactly twice the free value
    assert np.allclose(mv[:10], 2.0 * mv_free[:10], rtol=1e-6)


def test_oracle_sticky_zeroes_nodes_inside():
    """STICKY with a static object: every grid node inside the level set has zero velocity after the grid update."""
    api = oracle_api()
    sc = scenes.sphere_on_obstacle(bits=6, radius_cells=4.0, obstacle_cells=5.0, boundary="sticky", speed=2.0)
    eng = build_engine(sc, api=api)
    eng.initial_setup()
    for _ in range(150):
        eng.grid_update(sc["dt"])
        eng.g2p2g(sc["dt"], sc["dt"])
        eng.rebuild_partition()
    eng.grid_update(sc["dt"])
    keys, blocks = eng.dump_grid()
    eng.close()
    dx = 1.0 / 64
    c = np.array([0.5, 0.34, 0.5])
    inside = 0
    for key, blk in zip(keys, blocks):
        for cell in range(64):
            node = (key * 4 + np.array([cell >> 4, (cell >> 2) & 3, cell & 3])) * dx
            if blk[0, cell] > 0 and np.linalg.norm(node - c) < 5.0 * dx - 1e-6 and 8 * dx <= node.min() and node.max() < 1 - 8 * dx:
                inside += 1
                assert blk[1, cell] == 0 and blk[2, cell] == 0 and blk[3, cell] == 0
    assert inside > 10


@pytest.mark.gpu
@pytest.mark.parametrize("boundary,friction", [("sticky", 0.3), ("slip", 0.3), ("slip", 0.0), ("separate", 0.3)])
def test_hip_matches_oracle_with_collision_object(boundary, friction):
    sc = scenes.sphere_on_obstacle(bits=6, radius_cells=4.0, obstacle_cells=5.0, boundary=boundary, friction=friction, speed=2.0)
    steps = 220
    xo, mvo, to = _run(sc, steps, oracle_api())
    xh, mvh, th = _run(sc, steps)
    assert xh.shape == xo.shape
    idx, _ = match(xo.astype(np.float64), xh.astype(np.float64))
    rel = np.abs(xh[idx].astype(np.float64) - xo).max(axis=1) / np.abs(xo).max(axis=1)
    assert rel.max() < 1e-5, rel.max()
    assert np.allclose(mvh, mvo, rtol=2e-4, atol=1e-9)
    assert abs(th[0] - to[0]) <= 1e-6 * abs(to[0])


@pytest.mark.gpu
def test_hip_collision_object_can_be_removed():
    sc = scenes.sphere_on_obstacle(bits=6, radius_cells=4.0, obstacle_cells=5.0, boundary="sticky", speed=2.0)
    eng = build_engine(sc)
    eng.set_collision_object(None)
    eng.initial_setup()
    free = dict(sc)
    free.pop("collision")
    ref = build_engine(free)
    ref.initial_setup()
    for e in (eng, ref):
        e.run_fixed(100, sc["dt"])
    a, b = eng.retrieve_positions(0), ref.retrieve_positions(0)
    eng.close()
    ref.close()
    idx, _ = match(b.astype(np.float64), a.astype(np.float64))
    assert np.abs(a[idx] - b).max() < 1e-6
