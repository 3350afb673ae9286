"""Full-state checkpoint / restart: a run resumed from a checkpoint follows the uninterrupted run."""
import numpy as np
import pytest

from claymore_amd import _ffi, scenes
from claymore_amd.engine import build_engine
from oracle_ffi import oracle_api
from parity_util import match

pytestmark = pytest.mark.gpu


def _close(a, b, tol):
    idx, _ = match(b.astype(np.float64), a.astype(np.float64))
    rel = np.abs(a[idx].astype(np.float64) - b).max(axis=1) / np.abs(b).max(axis=1)
    return rel.max() < tol, rel.max()


@pytest.mark.parametrize("scene", ["spheres", "sand", "fluid"])
def test_restart_follows_the_uninterrupted_run(scene):
    if scene == "spheres":
        sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=3.0, speed=2.0)
    elif scene == "sand":
        sc = scenes.scaled_sand_column(7, 1.0 / 4096)
    else:
        sc = scenes.fluid_dam(bits=6, size_cells=(12, 10, 12), min_corner=(12, 12, 12))
    dt = sc["dt"]
    ref = build_engine(sc)
    ref.initial_setup()
    ref.run_fixed(30, dt)
    ckpt = ref.save_checkpoint().copy()
    keys_at_save, grid_at_save = ref.dump_grid()
    ref.run_fixed(30, dt)
    want = [ref.retrieve_positions(m) for m in range(len(sc["models"]))]
    ref.close()
    # the yardstick is the ORACLE's uninterrupted 60-substep run (the HIP run above only says what float-atomic noise to expect)
    ora = build_engine(sc, api=oracle_api())
    ora.initial_setup()
    ora.run_fixed(60, dt)
    want_oracle = [ora.retrieve_positions(m) for m in range(len(sc["models"]))]
    ora.close()
    for m, w in enumerate(want_oracle):
        ok, err = _close(want[m], w, 1e-5)
        assert ok, (scene, m, err)

    # (1) a fresh context: same models, initial_setup, then load
    eng = build_engine(sc)
    eng.initial_setup()
    eng.load_checkpoint(ckpt)
    k, g = eng.dump_grid()
    assert np.array_equal(k, keys_at_save) and np.array_equal(g, grid_at_save)     # the grid came back bit for bit
    eng.run_fixed(30, dt)
    for m, w in enumerate(want):
        got = eng.retrieve_positions(m)
        ok, err = _close(got, w, 2e-6)     # float atomics make two runs differ in the last bits
        assert ok, (scene, m, err)
        ok, err = _close(got, want_oracle[m], 1e-5)              # the restarted run against the oracle's uninterrupted one
        assert ok, (scene, m, err)
    # (2) rewinding the same context
    eng.load_checkpoint(ckpt)
    eng.run_fixed(30, dt)
    for m, w in enumerate(want):
        got = eng.retrieve_positions(m)
        ok, err = _close(got, w, 2e-6)
        assert ok, (scene, m, err)
        ok, err = _close(got, want_oracle[m], 1e-5)
        assert ok, (scene, m, err)
    eng.close()


def test_checkpoint_rejects_a_different_setup():
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=3.0)
    a = build_engine(sc)
    a.initial_setup()
    ckpt = a.save_checkpoint().copy()
    a.close()
    other = scenes.two_spheres(bits=6, radius_cells=4.0, gap_cells=3.0)
    b = build_engine(other)
    b.initial_setup()
    with pytest.raises(Exception):
        b.load_checkpoint(ckpt)
    with pytest.raises(Exception):
        b.load_checkpoint(ckpt[:100])
    b.close()


def test_checkpoint_rejects_other_physics_and_corrupt_headers():
    """Same geometry, different material parameters or gravity: the parameter hash refuses the load; header counts that do
    not fit together are refused before anything is copied."""
    import struct
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=3.0)
    a = build_engine(sc)
    a.initial_setup()
    a.run_fixed(3, sc["dt"])
    ckpt = a.save_checkpoint().copy()
    a.close()
    stiff = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=3.0, youngs=2e4)
    b = build_engine(stiff)
    b.initial_setup()
    with pytest.raises(Exception):
        b.load_checkpoint(ckpt)
    b.close()
    heavy = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=3.0)
    heavy.setdefault("config", {})["gravity"] = -1.0
    c = build_engine(heavy)
    c.initial_setup()
    with pytest.raises(Exception):
        c.load_checkpoint(ckpt)
    c.close()
    d = build_engine(sc)
    d.initial_setup()
    raw = bytearray(ckpt.tobytes())
    # header: magic u64, domain_bits/max_ppc/nmodels/rollid i32 x4, pbc/nbc/ebc/prev_count i32 x4
    pbc, nbc, ebc, prev = struct.unpack_from("<4i", raw, 24)
    for bad in ((nbc + 1, nbc, ebc, prev), (pbc, ebc + 1, ebc, prev), (pbc, nbc, 1 << 30, prev), (-1, nbc, ebc, prev)):
        broken = bytearray(raw)
        struct.pack_into("<4i", broken, 24, *bad)
        with pytest.raises(Exception):
            d.load_checkpoint(np.frombuffer(bytes(broken), dtype=np.uint8))
    d.load_checkpoint(ckpt)      # the untouched checkpoint still loads into the same context
    d.close()


def test_checkpoint_into_a_smaller_capacity_grows_it():
    """The checkpoint needs more blocks than the loading context has: the load grows the capacity (check_capacity path)."""
    # a sphere centred on a block corner occupies 2^3 particle blocks; two cells further along the diagonal it straddles 3^3
    dx = 1.0 / 64
    sc = {"name": "diag_sphere", "bits": 6, "dt": 1e-4, "config": {},
          "models": [{"material": _ffi.FIXED_COROTATED, "xyz": scenes.lattice_sphere(6, (0.5, 0.5, 0.5), 3.5), "v0": (4.0, 4.0, 4.0),
                      "params": {"volume": float(np.float32(dx ** 3 / 8)), "youngs_modulus": 5e3, "poisson_ratio": 0.4, "rho": 1e3}}]}
    a = build_engine(sc)
    a.initial_setup()
    ebc0 = a.counts().exterior_blocks
    ckpt = None
    for _ in range(80):                      # the moving spheres straddle more blocks at some point
        a.run_fixed(5, sc["dt"])
        if a.counts().exterior_blocks > ebc0:
            ckpt = a.save_checkpoint().copy()
            break
    if ckpt is None:
        a.close()
        pytest.skip("block count never exceeded its initial value")
    a.run_fixed(10, sc["dt"])
    want = a.retrieve_positions(0)
    a.close()
    sc["config"]["max_blocks"] = ebc0        # exactly what set-up needs
    b = build_engine(sc)
    b.initial_setup()
    cap0, _, _ = b.capacity()
    assert cap0 == ebc0
    b.load_checkpoint(ckpt)
    assert b.capacity()[0] > cap0
    b.run_fixed(10, sc["dt"])
    ok, err = _close(b.retrieve_positions(0), want, 2e-6)
    assert ok, err
    b.close()
