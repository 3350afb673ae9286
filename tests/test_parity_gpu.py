"""Parity tests proper: the HIP engine (through the C ABI) against the CPU oracle on identical seeded inputs,
plus the device math against the reference's golden vectors.  Tolerance: 1e-5 relative on particle positions
(BASELINE.json north_star), written out per test."""
import ctypes as C
import os

import numpy as np
import pytest

from claymore_amd import _ffi, scenes
from claymore_amd.engine import build_engine
from parity_util import grid_compare, grid_velocity_compare, match_and_compare, run_engine, run_pair, to_b

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
POS_TOL = 1e-5


def oracle_api_threads(threads):
    """The oracle with OpenMP over particle blocks in its G2P2G (test infrastructure: large cases only)."""
    from oracle_ffi import oracle_api
    api = oracle_api()
    return _ThreadedApi(api, threads)


class _ThreadedApi:
    """Forwards to the oracle's Api; sets the thread count right after initial_setup (the oracle's contexts start serial)."""

    def __init__(self, api, threads):
        self._api, self._threads = api, threads

    def __getattr__(self, name):
        fn = getattr(self._api, name)
        if name != "initial_setup":
            return fn

        def wrapped(ctx):
            rc = fn(ctx)
            self._api.raw.mpmo_set_threads(ctx, self._threads)
            return rc
        return wrapped


def f32(name):
    return np.fromfile(os.path.join(G, name), dtype=np.float32)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _params(material):
    hip = _ffi.load_hip()
    p = _ffi.MaterialParams()
    hip.default_material(material, 8, C.byref(p))
    p.volume = float(f32("g456_params.f32")[0])
    return p


def _bform(F9):
    """b = F F^T (n, 9) in float64 of column-major deformation gradients: what the device's stress functions return for a (projected) F."""
    F = np.asarray(F9, dtype=np.float64).reshape(-1, 3, 3).transpose(0, 2, 1)
    return np.einsum("nij,nkj->nik", F, F).reshape(-1, 9)


def _lame(p):
    e, nu = np.float32(p.youngs_modulus), np.float32(p.poisson_ratio)
    return float(e / (2 * (1 + nu))), float(e * nu / ((1 + nu) * (1 - 2 * nu)))


def test_device_fixed_corotated_vs_reference_golden():
    """P F^T of the fixed-corotated model against (1) the reference's golden output and (2) the float64 closed form.
    The reference's float result carries the residual of its approximate 4-sweep SVD (up to 3.5e-4 of the stiffness scale
    vol * E on these inputs); the device's symmetric eigen-decomposition converges, so the device must be (a) within 1e-5 of
    vol * E of the exact stress on every well-conditioned input (classes 0-5) and (b) never further from the truth than
    twice the reference itself."""
    import exact_models as X
    hip = _ffi.load_hip()
    F = f32("g3_F_in.f32")
    n = F.size // 9
    want = f32("g4_fc_out.f32").reshape(n, 9).astype(np.float64)
    got = np.empty((n, 19), dtype=np.float32)
    p = _params(_ffi.FIXED_COROTATED)
    assert hip.test_stress(_ffi.FIXED_COROTATED, C.byref(p), ptr(F), None, n, ptr(got), 0) == 0
    mu, lam = _lame(p)
    exact = X.fixed_corotated(F.reshape(n, 9), mu, lam, p.volume)
    scale = p.volume * p.youngs_modulus
    e_dev = np.abs(got[:, 9:18] - exact).max(axis=1) / scale
    e_ref = np.abs(want - exact).max(axis=1) / scale
    cls = np.arange(n) % 8
    ok = cls <= 5
    assert e_dev[ok].max() < 1e-5, e_dev[ok].max()
    assert (e_dev[ok] <= 2.0 * e_ref[ok] + 3e-6).all()
    assert np.median(e_dev) < 1e-6
    # near-singular / reflected inputs (classes 6, 7): same order as the reference
    assert e_dev[~ok].max() <= max(2.0 * e_ref[~ok].max(), 1e-4)


def test_device_sand_vs_reference_golden_and_closed_form():
    """Drucker-Prager return mapping: the projected state (b = F F^T of the projected F: the engine carries b, mpm_device_math.hpp),
    P F^T and log Jp against the float64 closed form, with the reference's own golden output as the yardstick (see the
    fixed-corotated test)."""
    import exact_models as X
    hip = _ffi.load_hip()
    F = f32("g3_F_in.f32").reshape(-1, 9)
    n = F.shape[0]
    lj = f32("g5_sand_logjp_in.f32")
    want = f32("g5_sand_out.f32").reshape(n, 19).astype(np.float64)
    idx = np.arange(n)
    sel = np.where((idx % 5 != 4) & (idx % 7 != 6))[0]      # the generator's default-parameter subset
    p = _params(_ffi.SAND)
    Fi, li = np.ascontiguousarray(F[sel]), np.ascontiguousarray(lj[sel])
    got = np.empty((sel.size, 19), dtype=np.float32)
    assert hip.test_stress(_ffi.SAND, C.byref(p), ptr(Fi), ptr(li), sel.size, ptr(got), 0) == 0
    mu, lam = _lame(p)
    exF, exPF, exL = X.sand(Fi, li, mu, lam, p.volume, p.cohesion, p.beta, p.yield_surface, p.volume_correction)
    w = want[sel]
    scale = p.volume * p.youngs_modulus
    cls = sel % 8
    ok = (cls <= 5) & np.isfinite(w).all(axis=1)
    e_dev = np.abs(got[:, 9:18] - exPF).max(axis=1) / scale
    e_ref = np.abs(w[:, 9:18] - exPF).max(axis=1) / scale
    assert e_dev[ok].max() < 1e-5, e_dev[ok].max()
    assert (e_dev[ok] <= 2.0 * e_ref[ok] + 3e-6).all()
    exB = _bform(exF)
    f_dev = np.abs(got[:, 0:9] - exB).max(axis=1)
    f_ref = np.abs(_bform(w[:, 0:9]) - exB).max(axis=1)
    assert f_dev[ok].max() < 1e-5, f_dev[ok].max()              # (b is quadratic in F: twice the bound the projected F had)
    assert (f_dev[ok] <= 2.0 * f_ref[ok] + 4e-6).all()
    assert np.abs(got[ok, 18] - exL[ok]).max() < 2e-6
    assert np.abs(got[ok, 18] - w[ok, 18]).max() < 1e-5


def test_device_nacc_vs_closed_form_and_reference_golden():
    """NACC (flagged unstable in the reference, constitutive_models.cuh:80): projected F, P F^T and log Jp against the float64
    closed form (tests/exact_models.py: nacc, which itself reproduces the reference's golden output to 1.1e-5 of vol * E), like the
    fixed-corotated and sand tests.  The model is discontinuous in its case selection (tips / surface / inside), so rows whose case
    changes under a 3e-6 perturbation of F are left out; what remains covers all four cases."""
    import exact_models as X
    hip = _ffi.load_hip()
    F = f32("g3_F_in.f32").reshape(-1, 9)
    n = F.shape[0]
    lj = f32("g6_nacc_logjp_in.f32")
    want = f32("g6_nacc_out.f32").reshape(n, 19).astype(np.float64)
    idx = np.arange(n)
    sel = np.where((idx % 5 != 4) & (idx % 7 != 6))[0]
    p = _params(_ffi.NACC)
    Fi, li = np.ascontiguousarray(F[sel]), np.ascontiguousarray(lj[sel])
    got = np.empty((sel.size, 19), dtype=np.float32)
    assert hip.test_stress(_ffi.NACC, C.byref(p), ptr(Fi), ptr(li), sel.size, ptr(got), 0) == 0
    mu, lam = _lame(p)
    args = (mu, lam, p.volume, p.beta, p.xi, p.msqr, p.hardening_on)
    exF, exPF, exL, case = X.nacc(Fi, li, *args)
    rng = np.random.default_rng(1)
    stable = np.ones(sel.size, dtype=bool)
    for _ in range(4):
        stable &= X.nacc(Fi.astype(np.float64) * (1 + 3e-6 * rng.standard_normal(Fi.shape)), li, *args)[3] == case
    w = want[sel]
    cls = sel % 8
    ok = (cls <= 5) & np.isfinite(w).all(axis=1) & stable & np.isfinite(exPF).all(axis=1)
    assert np.bincount(case[ok], minlength=4).min() >= 20          # every case is represented
    assert np.isfinite(got[ok]).all()
    scale = p.volume * p.youngs_modulus
    e_dev = np.abs(got[:, 9:18] - exPF).max(axis=1) / scale
    e_ref = np.abs(w[:, 9:18] - exPF).max(axis=1) / scale
    # (the reference's own float output is off by up to 1.1e-5 of vol * E in P F^T and 2e-4 in F on these rows: its approximate SVD)
    assert e_dev[ok].max() < 5e-6, (e_dev[ok].max(), e_ref[ok].max())
    assert (e_dev[ok] <= 2.0 * e_ref[ok] + 3e-6).all()
    exB = _bform(exF)
    f_dev = np.abs(got[:, 0:9] - exB).max(axis=1)
    f_ref = np.abs(_bform(w[:, 0:9]) - exB).max(axis=1)
    assert f_dev[ok].max() < 1e-5, (f_dev[ok].max(), f_ref[ok].max())
    assert (f_dev[ok] <= 2.0 * f_ref[ok] + 6e-6).all()
    assert np.abs(got[ok, 18] - exL[ok]).max() < 3e-6, np.abs(got[ok, 18] - exL[ok]).max()
    # the remaining rows (unstable case, near-singular / reflected inputs): bulk agreement with the reference's own output
    rest = np.isfinite(w).all(axis=1) & (cls <= 5) & ~ok
    if rest.any():
        wb = _bform(w[rest, 0:9])
        relF = np.abs(got[rest, 0:9] - wb).max(axis=1) / np.maximum(1.0, np.abs(wb).max(axis=1))
        assert np.median(relF) < 2e-5


@pytest.mark.parametrize("material", [_ffi.FIXED_COROTATED, _ffi.SAND, _ffi.NACC])
def test_device_undeformed_wave_early_exit(material):
    """A whole wave of undeformed particles (F F^T = I to rounding: free fall, rigid translation - the default window of the C3 bench)
    leaves the stress functions early: P F^T = 0 exactly, b = F F^T and log Jp untouched - which is what the models give there anyway (the
    float64 closed forms say so).  A wave that holds ONE deformed particle takes the full path: its undeformed lanes must come out the same."""
    import exact_models as X
    hip = _ffi.load_hip()
    n = 256
    F = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (n, 1))
    F[:, 1] = 2e-10                       # the off-diagonal dust a free-falling particle carries (dt grad v of a rounding-level velocity gradient)
    F[:, 5] = -3e-10
    p = _params(material)
    lj0 = float(p.log_jp0)
    lj = np.full(n, lj0, dtype=np.float32)
    got = np.empty((n, 19), dtype=np.float32)
    assert hip.test_stress(material, C.byref(p), ptr(F), ptr(lj), n, ptr(got), 0) == 0
    b_in = _bform(F)                                      # identity + the symmetrised dust
    assert np.abs(got[:, 0:9] - b_in).max() < 1e-12 and np.all(got[:, [0, 4, 8]] == 1.0) and np.all(got[:, 9:18] == 0.0) and np.all(got[:, 18] == lj0)
    mixed = F.copy()
    mixed[70] = np.array([0.95, 0.02, 0, -0.01, 0.90, 0.03, 0, 0.01, 0.93], dtype=np.float32)     # lane 6 of the second wave: compressed (sand carries no tension)
    got2 = np.empty((n, 19), dtype=np.float32)
    assert hip.test_stress(material, C.byref(p), ptr(mixed), ptr(lj), n, ptr(got2), 0) == 0
    keep = np.arange(n) != 70
    scale = p.volume * p.youngs_modulus
    assert np.abs(got2[keep, 9:18]).max() / scale < 1e-7 and np.abs(got2[keep, 0:9] - b_in[keep]).max() < 4e-7 and np.abs(got2[keep, 18] - lj0).max() < 1e-7
    # the deformed lane took the full path: it carries stress, or the return mapping moved its F / log Jp
    assert np.abs(got2[70, 9:18]).max() / scale > 1e-3 or np.abs(got2[70, 0:9] - _bform(mixed[70:71])[0]).max() > 1e-4 or abs(got2[70, 18] - lj0) > 1e-5
    mu, lam = _lame(p)
    if material == _ffi.FIXED_COROTATED:
        ex = X.fixed_corotated(F, mu, lam, p.volume)
    elif material == _ffi.SAND:
        ex = X.sand(F, lj, mu, lam, p.volume, p.cohesion, p.beta, p.yield_surface, p.volume_correction)[1]
    else:
        ex = X.nacc(F, lj, mu, lam, p.volume, p.beta, p.xi, p.msqr, p.hardening_on)[1]
    assert np.abs(ex).max() / scale < 1e-8                                                          # the truth is zero too


def test_device_sym_eig3_vs_float64():
    """The eigen-decomposition the stress functions are built on (sym_eig3 in mpm_device_math.hpp: cyclic Jacobi on F F^T with exact
    rotations), tested directly on the reference SVD's golden inputs (G3): U orthogonal, U diag(lam) U^T = F F^T, eigenvalues equal to
    numpy's float64 ones."""
    hip = _ffi.load_hip()
    F = f32("g3_F_in.f32")
    n = F.size // 9
    got = np.empty((n, 12), dtype=np.float32)
    assert hip.test_eig(ptr(F), n, ptr(got), 0) == 0
    Fm = F.reshape(n, 3, 3).transpose(0, 2, 1).astype(np.float64)
    b = np.einsum("nij,nkj->nik", Fm, Fm)
    U = got[:, 0:9].reshape(n, 3, 3).transpose(0, 2, 1).astype(np.float64)
    lam = got[:, 9:12].astype(np.float64)
    bmax = np.abs(b).reshape(n, -1).max(axis=1)
    assert np.abs(np.einsum("nji,njk->nik", U, U) - np.eye(3)).max() < 2e-6                  # exact rotations: orthogonal to rounding
    assert np.abs(np.linalg.det(U) - 1.0).max() < 5e-6
    rec = np.abs(np.einsum("nij,nj,nkj->nik", U, lam, U) - b).reshape(n, -1).max(axis=1) / bmax
    assert rec.max() < 3e-6, rec.max()
    ev = np.linalg.eigvalsh(b)
    assert (np.abs(np.sort(lam, axis=1) - ev).max(axis=1) / bmax).max() < 2e-6
    # the residual off-diagonal of U^T b U relative to the smallest eigenvalue, on well-conditioned inputs: the wave-uniform exit (1e-6)
    cls = np.arange(n) % 8
    off = np.einsum("nji,njk,nkl->nil", U, b, U)
    off[:, [0, 1, 2], [0, 1, 2]] = 0
    good = cls <= 5
    assert (np.abs(off).reshape(n, -1).max(axis=1)[good] / ev[good, 0]).max() < 2e-5


@pytest.mark.parametrize("nsteps", [1, 10, 100])
def test_two_spheres_parity(nsteps):
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=3.0, speed=0.5)
    res = run_pair(sc, nsteps, 1e-4)
    err = match_and_compare(res)
    assert err["pos_rel"] < POS_TOL, err
    assert err["state_rel"] < 1e-4, err
    # (grid momentum TOTAL: 3 x the largest value measured over 1 / 10 / 100 substeps of this scene and of C1, 3.9e-5 - the total of a symmetric
    #  collision is a difference of two large sums, its scale the larger of |momentum| and 1e-3 of the mass; tools/probe_grid_parity.py)
    assert err["grid_mass_rel"] < 1e-5 and err["grid_mom_rel"] < 1.2e-4, err
    ch, co = res["hip"]["counts"], res["oracle"]["counts"]
    assert (ch.particle_blocks, ch.neighbor_blocks, ch.exterior_blocks) == (co.particle_blocks, co.neighbor_blocks, co.exterior_blocks)
    assert [ch.particles[i] for i in range(2)] == [co.particles[i] for i in range(2)]


def test_collision_parity():
    """Spheres in contact: exercises stress response, block activation/deactivation and cross-block advection."""
    sc = scenes.two_spheres(bits=6, radius_cells=6.0, gap_cells=0.5, speed=2.0, youngs=2e4)
    res = run_pair(sc, 150, 1e-4)
    err = match_and_compare(res)
    assert err["pos_rel"] < POS_TOL, err


def test_grid_node_parity_after_one_step():
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=1.0, speed=1.0)
    res = run_pair(sc, 1, 1e-4, collect_grid=True)
    assert grid_compare(res) < 2e-5


@pytest.mark.parametrize("nsteps", [10, 100])
def test_grid_velocity_parity_over_many_substeps(nsteps):
    """north_star: "positions/velocities match ... within 1e-5".  The reference's particles carry no velocity - it lives on the grid
    (update_grid_velocity_query_max, mgmpm_kernels.cuh:325-420) -, so velocity parity is GRID parity, node by node with the block keys matched,
    after many substeps and not only one: C1 (BASELINE config 1, 48 840 particles, 128^3) after 10 and 100 substeps.
    grid_compare: every node's {mass | momentum} within 1e-5 of the largest entry of its channel group (measured 1.2e-6 / 3.8e-6,
    tools/probe_grid_parity.py -> profiles/r06_grid_parity_probe.txt); grid_velocity_compare: momentum / mass per node, nodes lighter than
    1e-3 of the heaviest left out, relative to the largest |v| (measured 2.2e-6 / 7.9e-6: the quotient of two sums of ~27 x 8 float terms each)."""
    sc = scenes.two_spheres()
    res = run_pair(sc, nsteps, 1e-4, collect_grid=True)
    assert grid_compare(res) < 1e-5, grid_compare(res)
    v = grid_velocity_compare(res)
    assert v < (1e-5 if nsteps <= 10 else 2.4e-5), v     # (100 substeps: 3 x the measured 7.9e-6)
    err = match_and_compare(res)
    assert err["pos_rel"] < POS_TOL, err


@pytest.mark.parametrize("material,steps", [(_ffi.J_FLUID, 40), (_ffi.SAND, 40), (_ffi.NACC, 20)])
def test_other_materials_parity(material, steps):
    sc = scenes.sphere_drop(bits=6, radius_cells=6.0, center=(0.5, 0.3, 0.5), material=material)
    sc["models"][0]["params"] = {}
    res = run_pair(sc, steps, 1e-4)
    err = match_and_compare(res)
    assert err["pos_rel"] < POS_TOL, err
    assert err["logjp_abs"] < 1e-4, err


def test_adaptive_dt_parity():
    # fast spheres: dt is CFL-limited (dx * 0.5 / |v| ~ 9.7e-4 < dt_default) from the second step on
    sc = scenes.two_spheres(bits=6, radius_cells=4.0, gap_cells=20.0, speed=8.0)
    res = run_pair(sc, 15, 1e-4, adaptive=True, dt_default=2e-3)
    assert np.allclose(res["hip"]["dts"], res["oracle"]["dts"], rtol=1e-4)
    assert 5e-4 < res["hip"]["dts"][-1] < 1.5e-3
    err = match_and_compare(res)
    assert err["pos_rel"] < POS_TOL, err


def test_wall_boundary_parity():
    """A sphere dropped onto the floor zone: slip-wall branch of the grid update (mgmpm_kernels.cuh:339,:367-370)."""
    sc = scenes.sphere_drop(bits=6, radius_cells=5.0, center=(0.5, 0.22, 0.5), material=_ffi.FIXED_COROTATED)
    sc["models"][0]["v0"] = (0.0, -3.0, 0.0)
    res = run_pair(sc, 120, 1e-4)
    err = match_and_compare(res)
    assert err["pos_rel"] < POS_TOL, err


def test_c1_config_parity_50k():
    """BASELINE config 1 (two spheres, ~49 k particles, 128^3): 200 substeps against the oracle."""
    sc = scenes.two_spheres()
    res = run_pair(sc, 200, 1e-4)
    err = match_and_compare(res)
    assert err["pos_rel"] < POS_TOL, err


def test_full_size_invariants_5m():
    """C2-size run (5 M particles, 256^3): size-independent properties - mass, particle count, momentum."""
    sc = scenes.sphere_drop()
    n = scenes.total_particles(sc)
    eng = build_engine(sc)
    eng.initial_setup()
    m0 = eng.grid_totals()
    mass = n * float(np.float32(sc["models"][0]["params"]["volume"]) * np.float32(1e3))
    assert abs(m0[0] - mass) / mass < 1e-4
    steps, dt = 20, 1e-4
    eng.run_fixed(steps, dt)
    c = eng.counts()
    assert c.particles[0] == n
    tot = eng.grid_totals()
    assert abs(tot[0] - mass) / mass < 1e-4
    assert abs(tot[2] - (-9.8) * dt * steps * mass) < 5e-3 * abs(9.8 * dt * steps * mass)
    assert abs(tot[1]) < 1e-4 * mass and abs(tot[3]) < 1e-4 * mass
    xyz = eng.retrieve_positions(0)
    assert xyz.shape[0] == n and np.isfinite(xyz).all()
    # free fall of the centroid
    y0 = sc["models"][0]["xyz"][:, 1].astype(np.float64).mean()
    t = steps * dt
    assert abs(xyz[:, 1].astype(np.float64).mean() - (y0 - 0.5 * 9.8 * t * t)) < 2e-6
    eng.close()


def _lattice_order(x, bits):
    """Order of the particles by the lattice site they started from (site spacing dx / 2): unique while they have moved less than dx / 4."""
    q = np.rint(x.astype(np.float64) * (1 << (bits + 2))).astype(np.int64)      # units of dx / 4: sites sit at odd multiples
    key = (q[:, 0] << 42) | (q[:, 1] << 21) | q[:, 2]
    assert np.unique(key).size == key.size
    return np.argsort(key, kind="stable")


def test_full_size_c2_parity_with_contact_5m():
    """C2 at size against the ORACLE, not only invariants: the 5 M-particle fixed-corotated sphere (R = 53 dx, 256^3) placed so that its
    lowest layers sit in the floor's wall zone - the grid update zeroes their velocity from the first substep on, a stress wave runs up
    the sphere - 12 substeps on both engines, particles matched by the lattice site they started from: positions within 1e-5 relative,
    F within 1e-4 (the test_parity bound), block counts equal."""
    bits = 8
    sc = scenes.sphere_drop(bits=bits, radius_cells=53.0, center=(0.5, (53.0 + 6.5) / 256.0, 0.5))
    sc["models"][0]["v0"] = (0.0, -0.5, 0.0)                 # thrown at the floor: 5e-5 per substep = 1.3 % of a cell (0.15 dx in all)
    n = scenes.total_particles(sc)
    assert n > 4.9e6
    nsteps, dt = 12, 1e-4
    hip = run_engine(sc, nsteps, dt)
    api = oracle_api_threads(32)
    ora = run_engine(sc, nsteps, dt, api=api)
    xh, fh, _ = hip["state"][0]
    xo, fo, _ = ora["state"][0]
    assert xh.shape == xo.shape == (n, 3)
    # both engines moved every particle by < dx / 4 from its site: undo the (common) rigid part before snapping to sites
    shift = np.array([0.0, -0.5 * nsteps * dt, 0.0])
    oh, oo = _lattice_order(xh - shift, bits), _lattice_order(xo - shift, bits)
    dx = np.abs(xh[oh].astype(np.float64) - xo[oo].astype(np.float64)).max(axis=1)
    rel = dx / np.abs(xo[oo]).max(axis=1)
    assert rel.max() < POS_TOL, rel.max()
    assert np.abs(to_b(fh, True)[oh] - to_b(fo, False)[oo]).max() < 1e-4               # b = F F^T: the state the HIP engine carries
    assert np.abs(fo - np.eye(3, dtype=np.float32).T.reshape(1, 9)).max() > 1e-3        # the contact really deformed something
    ch, co = hip["counts"], ora["counts"]
    assert (ch.particle_blocks, ch.neighbor_blocks, ch.exterior_blocks) == (co.particle_blocks, co.neighbor_blocks, co.exterior_blocks)


def test_full_size_invariants_c5_rank_share_12m():
    """One rank's share of BASELINE config 5 at full size (12.6 M weakly compressible J-fluid particles, 1024^3 grid), 20
    substeps at the scene's acoustic-CFL time step: particle count, nothing lost or discarded, grid mass, gravity's impulse."""
    sc = scenes.fluid_dam(10, (32, 192, 256))
    n = scenes.total_particles(sc)
    assert n == 32 * 192 * 256 * 8
    eng = build_engine(sc)
    eng.initial_setup()
    mass = n * eng.model_mass(0)
    steps, dt = 20, sc["dt"]
    eng.run_fixed(steps, dt)
    c = eng.counts()
    assert c.particles[0] == n
    d = eng.diagnostics()
    assert d.lost_particles == 0 and d.discarded_p2g == 0 and d.overflow_flags == 0
    tot = eng.grid_totals()
    assert np.isfinite(tot).all() and abs(tot[0] - mass) / mass < 1e-4
    xyz = eng.retrieve_positions(0)
    assert xyz.shape[0] == n and np.isfinite(xyz).all()
    eng.close()


def test_full_size_flow_invariants_c3_40m():
    """BASELINE config 3 at full size IN THE FLOW (the regime "sand column collapse" means; the 20-substep test below sees free fall
    only): 3 000 substeps to get the collapse going, then the size-independent properties over the next 40 - every particle still
    bucketed, nothing lost / discarded / dropped, the grid carries the whole mass, all particles inside the walls, the pile is lower
    and wider than the column was, the fastest particle is no faster than a free fall from the column's top, and between the two
    looks the centre of mass keeps sinking while the total momentum stays below weight x time (the floor pushes back)."""
    sc = scenes.sand_column(9)
    n = scenes.total_particles(sc)
    x0 = sc["models"][0]["xyz"]
    lo0, hi0 = x0.min(axis=0).astype(np.float64), x0.max(axis=0).astype(np.float64)
    com0 = x0.mean(axis=0, dtype=np.float64)
    del x0
    eng = build_engine(sc)
    eng.initial_setup()
    mass = n * eng.model_mass(0)
    dt = 1e-4
    eng.run_fixed(3000, dt)

    def look():
        c, d = eng.counts(), eng.diagnostics()
        assert c.particles[0] == n and d.lost_particles == 0 and d.discarded_p2g == 0 and d.overflow_flags == 0 and d.dropped_particles == 0
        tot = eng.grid_totals()
        assert np.isfinite(tot).all() and abs(tot[0] - mass) / mass < 1e-4
        x = eng.retrieve_positions(0)
        assert x.shape[0] == n and np.isfinite(x).all()
        wall = 8.0 / 512.0                                       # two blocks of slip wall: the grid velocity normal to it is zero in the whole zone,
        assert x.min() >= wall - 2.0 / 512.0 and x.max() <= 1.0 - wall + 2.0 / 512.0   # a particle gets a cell or so into it before it stops
        return tot, x.mean(axis=0, dtype=np.float64), x.min(axis=0).astype(np.float64), x.max(axis=0).astype(np.float64)

    tot_a, com_a, lo_a, hi_a = look()
    assert com_a[1] < com0[1] - 0.02                             # it has collapsed: the centre of mass came down by > 10 cells
    assert hi_a[1] < hi0[1] - 0.02                               # ... the top too
    assert lo_a[0] < lo0[0] - 0.02 and hi_a[0] > hi0[0] + 0.02   # ... and the pile spreads in x
    assert lo_a[2] < lo0[2] - 0.02 and hi_a[2] > hi0[2] + 0.02   # ... and in z
    eng.run_fixed(40, dt)
    tot_b, com_b, _, _ = look()
    assert com_b[1] < com_a[1]                                   # still sinking
    vcom = (com_b - com_a) / (40 * dt)
    vmax = np.sqrt(2 * 9.8 * (hi0[1] - 8.0 / 512.0))             # free fall from the top of the column
    assert np.abs(vcom).max() < vmax
    assert abs(vcom[0]) < 0.05 and abs(vcom[2]) < 0.05           # the column is symmetric in x and z: no net sideways drift
    assert abs(tot_b[2]) < mass * 9.8 * 0.304                    # |momentum_y| far below weight x elapsed time: the floor carries the pile
    eng.close()


def test_full_size_invariants_c3_40m():
    """BASELINE config 3 at full size (40.1 M Drucker-Prager particles, 512^3), 20 substeps: the size-independent
    properties the 1e-5 parity tests cannot reach - particle count (gmpm_simulator.cuh:617), nothing lost or discarded,
    grid mass, momentum = weight * time until the floor pushes back, finite positions, column centroid in free fall."""
    sc = scenes.sand_column(9)
    n = scenes.total_particles(sc)
    assert n == 128 * 306 * 128 * 8
    eng = build_engine(sc)
    eng.initial_setup()
    mass = n * eng.model_mass(0)
    m0 = eng.grid_totals()
    assert abs(m0[0] - mass) / mass < 1e-4
    steps, dt = 20, 1e-4
    eng.run_fixed(steps, dt)
    c = eng.counts()
    assert c.particles[0] == n
    d = eng.diagnostics()
    assert d.lost_particles == 0 and d.discarded_p2g == 0 and d.overflow_flags == 0
    tot = eng.grid_totals()
    assert np.isfinite(tot).all()
    assert abs(tot[0] - mass) / mass < 1e-4
    assert abs(tot[1]) < 1e-4 * mass and abs(tot[3]) < 1e-4 * mass
    # the column starts at rest 12 cells above the floor zone: in 2 ms nothing has reached a wall yet, so the total momentum is
    # gravity's impulse; the elastic wave set off by the missing support only redistributes it
    assert abs(tot[2] - (-9.8) * dt * steps * mass) < 2e-2 * abs(9.8 * dt * steps * mass)
    xyz = eng.retrieve_positions(0)
    assert xyz.shape[0] == n and np.isfinite(xyz).all()
    assert xyz.min() > 0.0 and xyz.max() < 1.0
    eng.close()


def test_fast_flow_mass_is_conserved_through_the_shell_path():
    """Particles that cross into a neighbouring block send part of their P2G stencil straight to the grid (p2g_shell in
    mpm_g2p2g.hpp) instead of the LDS arena.  A fast translating sphere keeps ~10 % of its particles on that path every substep:
    mass, momentum and the particle count must come out exactly as for particles at rest."""
    sc = scenes.two_spheres(bits=6, radius_cells=6.0, gap_cells=6.0, speed=3.0)
    n = scenes.total_particles(sc)
    eng = build_engine(sc)
    eng.initial_setup()
    mass = sum(eng.model_mass(i) * m["xyz"].shape[0] for i, m in enumerate(sc["models"]))
    for _ in range(60):
        eng.run_fixed(1, 1e-4)
        tot = eng.grid_totals()
        assert abs(tot[0] - mass) / mass < 2e-5
    c = eng.counts()
    assert c.particles[0] + c.particles[1] == n
    d = eng.diagnostics()
    assert d.lost_particles == 0 and d.discarded_p2g == 0
    eng.close()


def test_capacity_overflow_reports_error():
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=3.0)
    sc["config"]["max_ppc"] = 4          # 256 particles per block < 512 needed
    eng = build_engine(sc)
    with pytest.raises(Exception):
        eng.initial_setup()
    eng.close()


def test_fuzz_parity():
    """Random small scenes (tests/fuzz_scenes.py: 1-3 bodies, J-fluid / fixed-corotated / sand, random sizes, positions - some next
    to a wall -, velocities up to 3 m/s, 20-120 substeps): positions within 1e-5 relative of the oracle's, block counts equal.
    (NACC, which the reference itself flags unstable, constitutive_models.cuh:80, has its own test; tools/fuzz_parity.py runs
    the long / fast / all-material variant.)"""
    from fuzz_scenes import random_scene
    rng = np.random.default_rng(2024)
    for case in range(12):
        sc, nsteps = random_scene(rng, case, 3.0, 120, (_ffi.J_FLUID, _ffi.FIXED_COROTATED, _ffi.SAND))
        res = run_pair(sc, nsteps)
        co, ch = res["oracle"]["counts"], res["hip"]["counts"]
        assert [co.particles[i] for i in range(len(sc["models"]))] == [m["xyz"].shape[0] for m in sc["models"]]
        w = match_and_compare(res)
        assert w["pos_rel"] < 1e-5, (case, w)
        assert (ch.particle_blocks, ch.neighbor_blocks, ch.exterior_blocks) == (co.particle_blocks, co.neighbor_blocks, co.exterior_blocks), case


def test_fuzz_parity_violent_tier():
    """The same generator with bodies up to 6 m/s and up to 300 substeps (J-fluid / fixed-corotated / sand).  In this tier 1e-5 does
    not hold everywhere - and the reason is a fact, not noise: a free-flying lattice body sits within 0.003 ulp of a rounding tie in
    x + v dt, which the two engines' gathered velocities (3e-7 apart: summation order) resolve in opposite directions at EVERY substep
    (tools/fuzz_tie_probe.py, profiles/r02_fuzz_parity.txt), so the distance grows linearly, by at most one ulp of the position per
    substep.  Asserted here: (1) that growth bound - |dx| <= one ulp of the [0.5, 1) binade per substep -, (2) 2.5e-5 relative for
    every case, (3) block counts equal, (4) at most a sixth of the cases above 1e-5 (6 of 200 in the long fuzz)."""
    from fuzz_scenes import random_scene
    rng = np.random.default_rng(7)
    ulp = 2.0 ** -24                      # spacing of float32 in [0.5, 1) is 2^-23: positions live in (0, 1)
    above = 0
    ncases = 24
    for case in range(ncases):
        sc, nsteps = random_scene(rng, case, 6.0, 300, (_ffi.J_FLUID, _ffi.FIXED_COROTATED, _ffi.SAND))
        res = run_pair(sc, nsteps)
        co, ch = res["oracle"]["counts"], res["hip"]["counts"]
        n_in = [m["xyz"].shape[0] for m in sc["models"]]
        if any(co.particles[i] < n_in[i] for i in range(len(n_in))):
            continue                      # the ORACLE dropped particles (the reference's per-cell capacity): the two engines then simulate different particle sets;
                                          # that regime has its own test (test_overflow_regime_drop_count_matches_the_oracle).  None of the 24 cases of this seed takes this branch.
        w = match_and_compare(res)
        assert w["pos_abs"] <= 2 * ulp * nsteps + 2e-6, (case, nsteps, w)          # <= 1 ulp (2^-23) per substep
        assert w["pos_rel"] < 2.5e-5, (case, nsteps, w)
        assert (ch.particle_blocks, ch.neighbor_blocks, ch.exterior_blocks) == (co.particle_blocks, co.neighbor_blocks, co.exterior_blocks), case
        above += w["pos_rel"] >= 1e-5
    assert above <= ncases // 6, above


def test_runtime_overflow_error_or_drop():
    """A block that receives more particles than it has list slots (max_ppc * 64): MPM_ERR_CAPACITY by default; with
    mpm_config.drop_overflow the surplus is dropped and counted - what the reference does silently and per cell
    (particle_buffer.cuh:122-130) - and the run goes on with a consistent particle count."""
    def scene():
        sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=0.5, speed=3.0, youngs=2e3)
        sc["config"]["max_ppc"] = 8      # 512 per block: the lattice fills interior blocks exactly, any compression overflows
        return sc
    sc = scene()
    n = scenes.total_particles(sc)
    eng = build_engine(sc)
    eng.initial_setup()
    with pytest.raises(Exception):
        eng.run_fixed(400, 1e-4)
    eng.close()
    sc = scene()
    sc["config"]["drop_overflow"] = 1
    eng = build_engine(sc)
    eng.initial_setup()
    eng.run_fixed(400, 1e-4)
    c, d = eng.counts(), eng.diagnostics()
    assert d.dropped_particles > 0 and (d.overflow_flags & 2)
    assert c.particles[0] + c.particles[1] == n - d.dropped_particles
    x = np.concatenate([eng.retrieve_positions(0), eng.retrieve_positions(1)])
    assert x.shape[0] == n - d.dropped_particles and np.isfinite(x).all()
    tot = eng.grid_totals()
    mass = sum(eng.model_mass(i) * (c.particles[i]) for i in range(2))
    assert abs(tot[0] - mass) / mass < 1e-4
    eng.close()


def test_block_capacity_grows_like_check_capacity():
    """gmpm_simulator.cuh:283-300: capacities grow by 3/2 once 3/4 full.  A run that starts with a block capacity just above
    its exterior block count grows at the first rebuild and must then follow the same trajectory as a generously sized run."""
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=4.0, speed=2.0)
    ref = build_engine(sc)
    ref.initial_setup()
    ebc = ref.counts().exterior_blocks
    ref.run_fixed(40, sc["dt"])
    xyz_ref = ref.retrieve_positions(0)
    ref.close()

    sc["config"]["max_blocks"] = ebc + 8   # initial capacity, 3/4 rule trips immediately
    eng = build_engine(sc)
    eng.initial_setup()
    cap0, bins0, ev0 = eng.capacity()
    assert cap0 == ebc + 8 and ev0 == 0
    eng.run_fixed(40, sc["dt"])
    cap1, bins1, ev1 = eng.capacity()
    assert ev1 >= 1 and cap1 >= (cap0 * 3) // 2 and bins1[0] >= bins0[0]
    xyz = eng.retrieve_positions(0)
    eng.close()
    from parity_util import match
    idx, _ = match(xyz_ref.astype(np.float64), xyz.astype(np.float64))
    rel = np.abs(xyz[idx].astype(np.float64) - xyz_ref).max(axis=1) / np.abs(xyz_ref).max(axis=1)
    assert xyz.shape == xyz_ref.shape and rel.max() < POS_TOL, rel.max()


def test_capacity_grows_inside_a_long_run_between_windows():
    """Capacity growth happens at host synchronisations only, i.e. once per window of sync_interval substeps: two touching
    spheres flying apart keep entering new blocks (190 -> ~265 exterior blocks over 300 substeps, never more than ~8 % inside
    one window).  The context starts 70 % full - below the 3/4 mark, so nothing grows at the entry of the run - and must be
    grown in time at some window boundary of ONE long mpm_run_fixed call; same trajectory as a roomy context."""
    def scene():
        return scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=0.5, speed=-3.0)     # negative speed: apart
    sc = scene()
    ref = build_engine(sc)
    ref.initial_setup()
    nm = len(sc["models"])
    ebc0 = ref.counts().exterior_blocks
    peak = ebc0
    for _ in range(30):
        ref.run_fixed(10, sc["dt"])
        peak = max(peak, ref.counts().exterior_blocks)
    want = [ref.retrieve_positions(m) for m in range(nm)]
    ref.close()
    sc = scene()
    sc["config"]["max_blocks"] = int(ebc0 / 0.70)          # 70 % full at set-up: below the 3/4 growth mark
    eng = build_engine(sc)
    eng.initial_setup()
    cap0, _, ev0 = eng.capacity()
    assert ev0 == 0 and peak > cap0 * 3 // 4, (ebc0, peak, cap0)      # the run will have to grow
    eng.run_fixed(300, sc["dt"])                           # one call: 38 windows of 8 substeps
    cap1, _, ev1 = eng.capacity()
    assert ev1 >= 1 and cap1 > cap0
    from parity_util import match
    for m in range(nm):
        got = eng.retrieve_positions(m)
        idx, _ = match(want[m].astype(np.float64), got.astype(np.float64))
        rel = np.abs(got[idx].astype(np.float64) - want[m]).max(axis=1) / np.abs(want[m]).max(axis=1)
        assert rel.max() < 2e-6, rel.max()
    eng.close()


def test_fixed_capacity_reports_error_when_exceeded():
    """grow = 0: the reference's abort path (gmpm_simulator.cuh:473-476) becomes MPM_ERR_CAPACITY."""
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=4.0)
    probe = build_engine(sc)
    probe.initial_setup()
    ebc = probe.counts().exterior_blocks
    probe.close()
    sc["config"]["max_blocks"] = ebc - 1
    sc["config"]["grow"] = 0
    eng = build_engine(sc)
    with pytest.raises(Exception) as ei:
        eng.initial_setup()
        eng.run_fixed(5, sc["dt"])
    assert "block" in str(ei.value).lower() or "capacity" in str(ei.value).lower()
    eng.close()


def _dense_cube_scene(ppc_axis=3, cells=8, bits=6, material=_ffi.FIXED_COROTATED):
    """A cube with ppc_axis^3 particles per cell: 27 per cell -> 1728 per block, i.e. more than one 1024-record sort
    chunk and more than kSortRounds (24) particles per sort key - the ragged / overflow paths of the in-LDS sort."""
    dx = 1.0 / (1 << bits)
    lo = (1 << bits) // 2 - cells // 2
    sub = (np.arange(ppc_axis) + 0.5) / ppc_axis - 0.5
    ax = (np.arange(lo, lo + cells)[:, None] + sub[None, :]).reshape(-1) * dx
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    xyz = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1).astype(np.float32)
    vol = float(np.float32(dx ** 3 / ppc_axis ** 3))
    return {"name": "dense_cube", "bits": bits, "dt": 1e-4, "config": {"max_ppc": 32},
            "models": [{"material": material, "xyz": xyz, "v0": (0.3, -0.2, 0.1), "params": {"volume": vol, "youngs_modulus": 5e3, "poisson_ratio": 0.4, "rho": 1e3}}]}


def test_dense_blocks_multichunk_parity():
    sc = _dense_cube_scene()
    res = run_pair(sc, 25, 1e-4)
    err = match_and_compare(res)
    assert err["n"] == 8 ** 3 * 27
    assert err["pos_rel"] < POS_TOL, err


def test_reference_capacity_max_ppc_128_parity():
    """The reference's own capacity (128 particles per cell -> 8192 per block, 13-bit slots in the advection record)."""
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=1.0, speed=1.5)
    sc["config"]["max_ppc"] = 128
    res = run_pair(sc, 30, 1e-4)
    err = match_and_compare(res)
    assert err["pos_rel"] < POS_TOL, err


def test_fast_flow_many_cell_changes_parity():
    """~25 % of the particles change their stencil base every substep (|v| dt = 0.25 dx): exercises the predicted sort
    key, the conflict/retry path of the scatter and cross-block advection in every step."""
    sc = scenes.two_spheres(bits=6, radius_cells=4.0, gap_cells=24.0, speed=4.0)
    dx = 1.0 / 64
    res = run_pair(sc, 40, 0.25 * dx / 4.0)
    err = match_and_compare(res)
    assert err["pos_rel"] < POS_TOL, err


def test_particle_outside_domain_is_rejected():
    sc = scenes.two_spheres(bits=6, radius_cells=4.0, gap_cells=3.0)
    sc["models"][0]["xyz"] = sc["models"][0]["xyz"].copy()
    sc["models"][0]["xyz"][0] = (1.5, 0.5, 0.5)
    eng = build_engine(sc)
    with pytest.raises(Exception):
        eng.initial_setup()
    eng.close()


def test_empty_context_is_rejected():
    from claymore_amd.engine import Engine
    eng = Engine(domain_bits=6)
    with pytest.raises(Exception):
        eng.initial_setup()
    eng.close()


def test_mixed_materials_share_one_grid_parity():
    """Four models of four materials in one context (the reference instantiates one g2p2g per material over the same grid,
    gmpm_simulator.cuh:364-413): an elastic sphere, a sand block, a NACC block and a J-fluid block falling side by side and touching."""
    bits, dx = 6, 1.0 / 64
    vol = float(np.float32(dx ** 3 / 8))

    def box(lo, hi):
        return scenes.lattice_box(bits, np.array(lo), np.array(hi))
    models = [
        {"material": _ffi.FIXED_COROTATED, "xyz": scenes.lattice_sphere(bits, (0.42, 0.5, 0.5), 4.0), "v0": (1.0, 0.0, 0.0),
         "params": {"volume": vol, "youngs_modulus": 5e3, "poisson_ratio": 0.4, "rho": 1e3}},
        {"material": _ffi.SAND, "xyz": box((33, 28, 28), (39, 36, 36)), "v0": (-1.0, 0.0, 0.0), "params": {"volume": vol}},
        {"material": _ffi.NACC, "xyz": box((28, 38, 28), (36, 43, 36)), "v0": (0.0, -1.0, 0.0), "params": {"volume": vol}},
        {"material": _ffi.J_FLUID, "xyz": box((28, 20, 28), (36, 26, 36)), "v0": (0.0, 1.0, 0.0), "params": {"volume": vol}},
    ]
    sc = {"name": "mixed", "bits": bits, "dt": 5e-5, "config": {}, "models": models}
    res = run_pair(sc, 120, 5e-5)
    err = match_and_compare(res)
    assert err["n"] == sum(m["xyz"].shape[0] for m in models)
    assert err["pos_rel"] < POS_TOL, err
    ch, co = res["hip"]["counts"], res["oracle"]["counts"]
    assert [ch.particles[i] for i in range(4)] == [co.particles[i] for i in range(4)]


def test_single_particle_and_tiny_models_parity():
    """Degenerate sizes: one particle, and a 3-particle model next to it (partial bins, one-lane iterations, empty sort rounds)."""
    dx = 1.0 / 64
    vol = float(np.float32(dx ** 3 / 8))
    p = {"volume": vol, "youngs_modulus": 5e3, "poisson_ratio": 0.4, "rho": 1e3}
    one = np.array([[0.5 + 0.25 * dx, 0.5 + 0.25 * dx, 0.5 + 0.25 * dx]], dtype=np.float32)
    three = np.array([[0.5 + 2.25 * dx, 0.5, 0.5], [0.5 + 2.75 * dx, 0.5, 0.5], [0.5 + 2.25 * dx, 0.5 + 0.5 * dx, 0.5]], dtype=np.float32)
    sc = {"name": "tiny", "bits": 6, "dt": 1e-4, "config": {},
          "models": [{"material": _ffi.FIXED_COROTATED, "xyz": one, "v0": (0.5, 0.0, -0.5), "params": p},
                     {"material": _ffi.SAND, "xyz": three, "v0": (-0.5, 0.0, 0.0), "params": {"volume": vol}}]}
    res = run_pair(sc, 200, 1e-4)
    err = match_and_compare(res)
    assert err["n"] == 4 and err["pos_rel"] < POS_TOL, err


def test_long_run_sand_stays_close_to_the_oracle():
    """500 substeps of a 9.7 k-particle sand column with plastic flow: north_star's 1e-5 holds here too.  (Round 1 needed
    5e-5: its device SVD was the source.  With the eigen-decomposition of round 2 the engine stays within 3e-7 of the oracle,
    the level at which two runs of the oracle itself separate when only the summation order of the grid accumulation is
    permuted - tools/sand_drift_study.py, profiles/r02_sand_drift_study.txt.)"""
    sc = scenes.scaled_sand_column(7, 1.0 / 64)
    res = run_pair(sc, 500, sc["dt"])
    err = match_and_compare(res)
    assert err["n"] == scenes.total_particles(sc)
    assert err["pos_rel"] < 2e-6, err
    assert err["grid_mass_rel"] < 1e-6, err


def test_setup_time_overflow_is_an_error_whatever_the_drop_policy():
    """drop_overflow covers particles that ARRIVE in a full block at run time; a block that starts with more particles than it has
    list slots is a configuration error (a dropped particle must not have been rasterised): MPM_ERR_CAPACITY, also with the policy on."""
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=3.0)
    sc["config"]["max_ppc"] = 4          # 256 particles per block < 512 needed
    sc["config"]["drop_overflow"] = 1
    eng = build_engine(sc)
    with pytest.raises(Exception) as ei:
        eng.initial_setup()
    assert "set-up" in str(ei.value) or "capacity" in str(ei.value).lower()
    eng.close()


@pytest.mark.parametrize("material", [_ffi.FIXED_COROTATED, _ffi.SAND])
def test_run_fixed_is_independent_of_the_sync_interval(material):
    """mpm_run_fixed enqueues sync_interval substeps between two host synchronisations (every kernel reads its block counts from the
    status block, launches are sized by stale estimates).  One synchronisation per substep, the default 8 and a window longer than the
    run must give the same trajectory (to float-atomic noise) and the same counts, here in a fast collision where the block counts
    change every few substeps - and all of them the oracle's."""
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=2.0, speed=3.0, material=material)
    nsteps = 90
    runs = {}
    for k in (1, 8, 64):
        sc["config"]["sync_interval"] = k
        eng = build_engine(sc)
        eng.initial_setup()
        eng.run_fixed(nsteps, sc["dt"])
        c = eng.counts()
        runs[k] = (np.concatenate([eng.retrieve_positions(0), eng.retrieve_positions(1)]), (c.particle_blocks, c.neighbor_blocks, c.exterior_blocks))
        d = eng.diagnostics()
        assert d.lost_particles == 0 and d.discarded_p2g == 0
        eng.close()
    from parity_util import match
    for k in (8, 64):
        assert runs[k][1] == runs[1][1], (k, runs[k][1], runs[1][1])
        idx, _ = match(runs[1][0].astype(np.float64), runs[k][0].astype(np.float64))
        rel = np.abs(runs[k][0][idx].astype(np.float64) - runs[1][0]).max(axis=1) / np.abs(runs[1][0]).max(axis=1)
        assert rel.max() < 2e-6, (k, rel.max())
    del sc["config"]["sync_interval"]
    w = match_and_compare(run_pair(sc, nsteps))
    assert w["pos_rel"] < POS_TOL, w


def test_failed_run_then_checkpoint_load_recovers():
    """A run that stops with an error between two substeps (here: block capacity exhausted with grow = 0) leaves the grid in the
    pre-updated state of the fused carry-over; loading a checkpoint must bring the canonical state back - flags included - and the
    run must follow the reference trajectory again (round-2 advisor finding: the flags survived the load)."""
    from parity_util import match
    sc = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=0.5, speed=-3.0)      # flying apart: 190 -> 270 exterior blocks in 300 substeps
    ref = build_engine(sc)
    ref.initial_setup()
    ebc = ref.counts().exterior_blocks
    ckpt = ref.save_checkpoint().copy()
    ref.run_fixed(3, sc["dt"])
    want = ref.retrieve_positions(0)
    ref.close()
    small = scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=0.5, speed=-3.0)
    small["config"]["max_blocks"] = ebc + 2
    small["config"]["grow"] = 0
    eng = build_engine(small)
    eng.initial_setup()
    failed = False
    for _ in range(40):                      # the spheres separate: soon more blocks than the fixed capacity holds
        try:
            eng.run_fixed(10, sc["dt"])
        except Exception as e:
            failed = "block" in str(e).lower() or "capacity" in str(e).lower()
            break
    assert failed, "the block count never exceeded the fixed capacity"
    eng.load_checkpoint(ckpt)                # rewind to t = 0 (fits: set-up needed ebc blocks)
    eng.run_fixed(3, sc["dt"])
    got = eng.retrieve_positions(0)
    idx, _ = match(want.astype(np.float64), got.astype(np.float64))
    rel = np.abs(got[idx].astype(np.float64) - want).max(axis=1) / np.abs(want).max(axis=1)
    assert rel.max() < 2e-6, rel.max()
    eng.close()


@pytest.mark.parametrize("material", [_ffi.J_FLUID, _ffi.FIXED_COROTATED, _ffi.SAND, _ffi.NACC])
def test_one_particle_scenes_through_the_real_kernel_against_the_references_statements(material):
    """G16-G18 on the GPU, through the PUBLIC C ABI and the real g2p2g_kernel (no test kernel): the golden rows whose particle is in the state
    initial_setup leaves one in (F = I, the model's log Jp_0) are replayed as one-particle scenes - mpm_add_model at the row's position,
    mpm_initial_setup, the row's velocity arena written into the eight grid blocks of the particle's node cube (mpm_halo_reduce adds blocks by
    key into the current grid, whose momentum channels are exactly zero for a particle at rest), mpm_g2p2g(dt, new_dt), mpm_rebuild_partition -
    and what comes out (mpm_retrieve_state: position, b, log Jp; mpm_dump_grid: every node) is compared with what the REFERENCE'S OWN
    STATEMENTS of mgmpm_kernels.cuh:772-905 + :518-663 produced for that particle (tests/golden/gen/gen_golden_kernel.sh): position <= 1e-6
    relative, b 1.2e-5 and log Jp 6e-6 absolute (3 x what was measured, on every arena), the 27 x {m, mv} node values 1e-5 of the stencil's largest, nothing anywhere else, a discarded particle
    (:877-885) counted and absent from the grid, the particle bucketed in the block add_advection was given (:863)."""
    import torch
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    par = np.fromfile(os.path.join(G, "g16_params.f32"), np.float32)
    names = "bits vol mass mu lam cohesion beta_sand yield_surface volume_correction bm xi msqr hardening_on dt new_dt beta_nacc E nu rho bulk gamma viscosity".split()
    P = dict(zip(names, [float(v) for v in par]))
    bits = int(P["bits"])
    dx = 1.0 / (1 << bits)
    arenas = np.fromfile(os.path.join(G, "g16_arenas.f32"), np.float32).reshape(-1, 3, 8, 8, 8)
    rin = np.fromfile(os.path.join(G, "g16_particle_in.f32"), np.float32).reshape(-1, 15)
    wf = np.fromfile(os.path.join(G, "g16_particle_out.f32"), np.float32).reshape(-1, 154)
    wi = np.fromfile(os.path.join(G, "g16_particle_out.i32"), np.int32).reshape(-1, 14)
    eye = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], np.float32) if material != _ffi.J_FLUID else np.array([1, 0, 0, 0, 0, 0, 0, 0, 0], np.float32)   # (J-fluid: J = 1 in the slot of F[0])
    lj0 = np.float32(-0.01 if material == _ffi.NACC else 0.0)
    plain = (rin[:, 0] == material) & np.all(rin[:, 5:14] == eye, axis=1) & (rin[:, 14] == lj0)
    rows = np.flatnonzero(plain)
    assert rows.size >= 60, rows.size
    prm = dict(rho=P["rho"], volume=P["vol"], youngs_modulus=P["E"], poisson_ratio=P["nu"])
    if material == _ffi.J_FLUID:
        prm = dict(rho=P["rho"], volume=P["vol"], bulk=P["bulk"], gamma=P["gamma"], viscosity=P["viscosity"])
    if material == _ffi.SAND:
        prm.update(cohesion=P["cohesion"], beta=P["beta_sand"], yield_surface=P["yield_surface"], volume_correction=int(P["volume_correction"]))
    if material == _ffi.NACC:
        prm.update(beta=P["beta_nacc"], xi=P["xi"], msqr=P["msqr"], hardening_on=int(P["hardening_on"]))
    bad, seen = [], dict(crossed=0, discarded=0, arenas=set())
    worst = {}                                                     # measured deviations by (quantity, arena is CFL-abiding): printed with MPM_PRINT_WORST=1

    def note(what, a, e):
        worst[(what, a < 3)] = max(worst.get((what, a < 3), 0.0), float(e))
    for r in rows:
        a = int(rin[r, 1])
        pos = rin[r, 2:5].astype(np.float32)
        base, arena_c, adv_cell, dirtag, narena, disc = wi[r, 0:3], wi[r, 3:6], wi[r, 6:9], int(wi[r, 9]), wi[r, 10:13], int(wi[r, 13])
        blk = (base - 1) >> 2
        assert np.array_equal(((base - 1) & 3) + 1, arena_c)
        sc = {"name": "one", "bits": bits, "dt": P["dt"], "config": {"max_ppc": 8}, "models": [{"material": material, "xyz": pos.reshape(1, 3), "v0": (0.0, 0.0, 0.0), "params": dict(prm)}]}
        eng = build_engine(sc)
        eng.initial_setup()
        # the velocity arena -> the eight grid blocks of the node cube (channel 0, the mass, stays what the rasteriser left: G2P2G does not read it)
        keys = np.zeros((8, 3), np.int32)
        blocks = np.zeros((8, 4, 64), np.float32)
        for lb in range(8):
            ox, oy, oz = (lb >> 2) & 1, (lb >> 1) & 1, lb & 1
            keys[lb] = blk + (ox, oy, oz)
            blocks[lb, 1:4] = arenas[a][:, 4 * ox:4 * ox + 4, 4 * oy:4 * oy + 4, 4 * oz:4 * oz + 4].reshape(3, 64)
        dk, db = torch.from_numpy(keys).cuda(), torch.from_numpy(blocks).cuda()
        eng._check(eng.api.halo_reduce(eng.ctx, 0, C.c_void_p(dk.data_ptr()), C.c_void_p(db.data_ptr()), 8))
        eng.g2p2g(P["dt"], P["new_dt"])
        cnt = eng.rebuild_partition()
        x, st, lj = eng.retrieve_state(0)
        gk, gb = eng.dump_grid()
        dg = eng.diagnostics()
        eng.close()
        tag = f"row {r} (arena {a}, dirtag {dirtag}, discarded {disc})"
        want = wf[r]
        # position (G16)
        e = np.abs(x[0].astype(np.float64) - want[12:15]).max() / np.abs(want[12:15]).max()
        if not e <= 1e-6:
            bad.append((tag, "pos", e))
        # state: b = F F^T of what the body stored, log Jp (G18)
        F = want[15:24].astype(np.float64).reshape(3, 3).T
        b = to_b(st, True)[0].reshape(3, 3)
        if material == _ffi.J_FLUID:
            if np.isfinite(want[15]) and not abs(float(st[0, 0]) - float(want[15])) <= 2e-6 * max(1.0, abs(float(want[15]))):
                bad.append((tag, "J", float(st[0, 0]), float(want[15])))
        elif np.isfinite(F).all():
            e = np.abs(b - F @ F.T).max() / max(1.0, np.abs(F @ F.T).max())
            note("b", a, e)
            if not e <= 1.2e-5:                      # (3 x the largest deviation measured over all rows and materials, 3.9e-6: sand)
                bad.append((tag, "b", e))
            # log Jp: an ABSOLUTE bound (it is a sum of logarithms; rows with log Jp ~ 0 have no relative error to speak of), 3 x the largest deviation
            # measured over all rows, 2.0e-6 - the two CFL-violating arenas (70 m/s noise, a 90 m/s stream: a particle torn apart in ONE substep, where
            # the reference's own float arithmetic ends in NaN for some rows, which are skipped) included: round 5 allowed them 5e-3 relative, which
            # the measurement (MPM_PRINT_WORST=1, profiles/r06_one_particle_worst.txt) shows was never needed
            if material != _ffi.FIXED_COROTATED and np.isfinite(want[24]):
                note("logjp_abs", a, abs(float(lj[0]) - float(want[24])))
            if material != _ffi.FIXED_COROTATED and np.isfinite(want[24]) and not abs(float(lj[0]) - float(want[24])) <= 6e-6:
                bad.append((tag, "logjp", float(lj[0]), float(want[24])))
        # the block the particle is bucketed in (add_advection's cell, :863)
        if cnt.particle_blocks != 1 or int(cnt.particles[0]) != 1:
            bad.append((tag, "counts", cnt.particle_blocks, int(cnt.particles[0])))
        # the grid (G17): the 27 nodes of the new stencil, nothing else
        nodes = {}
        for k, blkv in zip(gk, gb):
            nz = np.flatnonzero(np.abs(blkv).sum(axis=0))
            for cell in nz:
                nodes[(4 * int(k[0]) + (cell >> 4), 4 * int(k[1]) + ((cell >> 2) & 3), 4 * int(k[2]) + (cell & 3))] = blkv[:, cell].astype(np.float64)
        if disc:
            if nodes or dg.discarded_p2g != 1:
                bad.append((tag, "a discarded particle reached the grid / was not counted", len(nodes), int(dg.discarded_p2g)))
        else:
            stencil = want[46:154].astype(np.float64).reshape(27, 4)
            if np.isfinite(stencil).all():
                scale_m, scale_p = np.abs(stencil[:, 0]).max(), max(np.abs(stencil[:, 1:]).max(), 1e-30)
                origin = 4 * blk + narena
                for s in range(27):
                    node = tuple(int(v) for v in origin + (s // 9, (s // 3) % 3, s % 3))
                    got = nodes.pop(node, np.zeros(4))
                    e = max(abs(got[0] - stencil[s, 0]) / scale_m, np.abs(got[1:] - stencil[s, 1:]).max() / scale_p)
                    if not e <= 1e-5:
                        bad.append((tag, "node", s, e))
                        break
                if nodes:
                    bad.append((tag, "mass outside the stencil", list(nodes)[:3]))
            if dg.discarded_p2g != 0 or dg.lost_particles != 0:
                bad.append((tag, "counted as discarded / lost", int(dg.discarded_p2g), int(dg.lost_particles)))
        seen["crossed"] += dirtag != 13
        seen["discarded"] += disc
        seen["arenas"].add(a)
    if os.environ.get("MPM_PRINT_WORST"):
        print("WORST", material, {f"{k[0]}{'' if k[1] else '_cfl_violating'}": v for k, v in sorted(worst.items())})
    assert not bad, (len(bad), bad[:12])
    assert len(seen["arenas"]) == arenas.shape[0] and seen["crossed"] >= 4 and seen["discarded"] >= 2, seen


def test_full_size_c3_parity_40m():
    """C3 at size against the ORACLE (VERDICT r4 #4a; until now the 40 M-particle runs were property checks): the BASELINE column (128 x 306 x 128
    cells of Drucker-Prager sand, 40 108 032 particles, 512^3) set down with its lowest layers inside the floor's wall zone and thrown at it at
    0.4 m/s - the grid update zeroes their vertical velocity from the first substep, the bottom of the column is compressed and yields -,
    8 substeps on both engines (the oracle with OpenMP over particle blocks), particles matched by the lattice site they started from:
    positions within 1e-5 relative, b and log Jp within the test_parity bounds, block counts equal, and - round 6 - the GRID node by node
    (mass / momentum within 3e-5 of the channel group's largest entry, velocity = momentum / mass within 1.1e-4 of the largest |v|: 3 x what was
    measured, see below)."""
    bits = 9
    sc = scenes.sand_column(bits, min_corner=(192, 7, 192))
    v0 = -0.4
    sc["models"][0]["v0"] = (0.0, v0, 0.0)
    n = scenes.total_particles(sc)
    assert n == 40108032
    nsteps, dt = 8, 1e-4
    hip = run_engine(sc, nsteps, dt, collect_grid=True)
    grid_h = hip["grid"]
    xh, bh, lh = hip["state"][0]
    oh = _lattice_order(xh - np.array([0.0, v0 * nsteps * dt, 0.0]), bits)
    xh, bh, lh = xh[oh], to_b(bh, True)[oh], lh[oh]
    ch = hip["counts"]
    hc = (ch.particle_blocks, ch.neighbor_blocks, ch.exterior_blocks)
    del hip, oh
    api = oracle_api_threads(64)
    ora = run_engine(sc, nsteps, dt, api=api, collect_grid=True)
    # velocity parity at size: the two grids node by node after the 8 substeps, ~94 k blocks with their keys matched
    g = {"hip": {"grid": grid_h}, "oracle": {"grid": ora["grid"]}}
    gc, gv = grid_compare(g), grid_velocity_compare(g)
    # Measured (two runs): every node within 9.1 - 9.6e-6 of the largest mass / momentum, its velocity within 3.2 - 3.6e-5 of the largest |v| - NOT
    # the 1e-5 the small scenes reach (test_grid_velocity_parity_over_many_substeps: 4e-6 / 8e-6).  In a column compressed against the floor a
    # node's momentum change is the SUM of large stress terms of its ~200 particle-node pairs that cancel to almost nothing: each engine rounds
    # that sum in its own order, and the residue is 1e-7 of the TERMS, not of the sum.  Positions integrate it harmlessly (1e-5 above).
    # The bounds are 3 x the measurement.
    assert gc < 3e-5 and gv < 1.1e-4, (gc, gv)
    del g, grid_h
    xo, fo, lo = ora["state"][0]
    assert xh.shape == xo.shape == (n, 3)
    oo = _lattice_order(xo - np.array([0.0, v0 * nsteps * dt, 0.0]), bits)
    xo, lo = xo[oo], lo[oo]
    rel = np.abs(xh.astype(np.float64) - xo.astype(np.float64)).max(axis=1) / np.abs(xo).max(axis=1)
    assert rel.max() < POS_TOL, rel.max()
    bo = to_b(fo[oo], False)
    assert np.abs(bh - bo).max() < 1e-4, np.abs(bh - bo).max()
    assert np.abs(lh - lo).max() < 5e-5, np.abs(lh - lo).max()       # (measured 2.0e-5 over 40 M particles: the tail of a sum of three logarithms after a 15 % compression)
    assert np.abs(bo - np.eye(3).reshape(1, 9)).max() > 1e-3 and np.abs(lo).max() > 1e-5     # the floor really deformed the sand, and some of it yielded
    co = ora["counts"]
    assert hc == (co.particle_blocks, co.neighbor_blocks, co.exterior_blocks)


def test_reflected_deformation_gradient_in_a_pipeline_scene():
    """The reflection bit of the b-state (sign of b00: det F < 0) reached through the PIPELINE, not only by the function-level tests
    (VERDICT r4, weak #3): two fixed-corotated blocks driven into each other at +-3 m/s with a substep 30 x beyond the CFL limit
    (dt = 5e-3 on a 64^3 grid: dt grad v ~ 1.3 in the contact layer), so det(I + dt grad v) < 0 for the particles of that layer in the
    first substep - 512 of 8 192 in the oracle, whose SVD then carries the sign in the smallest singular value (svd.cuh:590-770).  The HIP
    engine must flag exactly those particles, keep them flagged, and still follow the oracle's positions."""
    bits = 6
    prm = {"volume": (1.0 / (1 << bits)) ** 3 / 8.0, "youngs_modulus": 5e3, "poisson_ratio": 0.4, "rho": 1e3}
    V, dt = 3.0, 5e-3
    sc = {"name": "clash", "bits": bits, "dt": dt, "config": {"max_ppc": 128, "gravity": 0.0},
          "models": [{"material": _ffi.FIXED_COROTATED, "xyz": scenes.lattice_box(bits, (24, 28, 28), (32, 36, 36)), "v0": (V, 0, 0), "params": dict(prm)},
                     {"material": _ffi.FIXED_COROTATED, "xyz": scenes.lattice_box(bits, (32, 28, 28), (40, 36, 36)), "v0": (-V, 0, 0), "params": dict(prm)}]}
    for nsteps in (1, 3):
        res = run_pair(sc, nsteps=nsteps, dt=dt)
        flagged = reflected = 0
        for (xh, sh, _), (xo, so, _) in zip(res["hip"]["state"], res["oracle"]["state"]):
            from parity_util import match
            idx, _ = match(xo.astype(np.float64), xh.astype(np.float64))
            F = so.reshape(-1, 3, 3).transpose(0, 2, 1).astype(np.float64)
            neg_o = np.linalg.det(F) < 0
            neg_h = sh[idx][:, 0] < 0                     # the sign bit of b00
            assert np.array_equal(neg_o, neg_h), (nsteps, int(neg_o.sum()), int(neg_h.sum()))
            reflected += int(neg_o.sum())
            flagged += int(neg_h.sum())
            rel = np.abs(xh[idx].astype(np.float64) - xo).max(axis=1) / np.abs(xo).max(axis=1)
            assert rel.max() < POS_TOL, (nsteps, rel.max())
            b_o, b_h = to_b(so, False), to_b(sh[idx], True)
            assert np.abs(b_h - b_o).max() / max(1.0, np.abs(b_o).max()) < 1e-4, (nsteps, np.abs(b_h - b_o).max())
        assert reflected == flagged and reflected >= 256, (nsteps, reflected)


@pytest.mark.parametrize("material,speed", [(_ffi.SAND, 4.0), (_ffi.FIXED_COROTATED, 6.0)])
def test_overflow_regime_drop_count_matches_the_oracle(material, speed):
    """The overflow regime, where the two engines diverge BY DESIGN: the reference drops the particles beyond max_ppc per
    CELL, silently (particle_buffer.cuh:122-130) - which ones is decided by the order its atomics happen to take -; this engine caps per
    BLOCK (max_ppc * 64 list slots) and reports, so in the same situation it keeps them.  What CAN be compared is compared: a ball thrown at
    the floor with a cell capacity of 16, (1) up to the substep before the oracle's first drop both engines hold every particle and agree
    to 1e-5; (2) in the substep of the first drop the number the oracle loses equals the surplus this engine's own particles show over
    the cell capacity, sum over cells of max(0, count - max_ppc), cell = the bucket add_advection files a particle under (:863: stencil base - 1);
    (3) this engine loses none of them and says so (nothing lost, nothing dropped: its block capacity is not reached)."""
    from parity_util import match
    from oracle_ffi import oracle_api
    bits, cap, dt = 6, 16, 1e-4
    sc = scenes.sphere_drop(bits=bits, radius_cells=6.0, center=(0.5, 0.26, 0.5), material=material)
    if material != _ffi.FIXED_COROTATED:
        sc["models"][0]["params"] = {}
    sc["models"][0]["v0"] = (0.0, -speed, 0.0)
    sc["config"]["max_ppc"] = cap
    n = scenes.total_particles(sc)
    ora = build_engine(sc, api=oracle_api())
    ora.initial_setup()
    onset, before = None, None
    for step in range(1, 600):
        xb = ora.retrieve_state(0)[0].copy()
        ora.run_fixed(1, dt)
        if int(ora.counts().particles[0]) < n:
            onset, before = step, xb
            break
    assert onset is not None and onset > 50, onset
    lost_by_the_oracle = n - int(ora.counts().particles[0])
    ora.close()
    hip = build_engine(sc)
    hip.initial_setup()
    hip.run_fixed(onset - 1, dt)
    xh = hip.retrieve_state(0)[0]
    idx, _ = match(before.astype(np.float64), xh.astype(np.float64))
    rel = np.abs(xh[idx].astype(np.float64) - before).max(axis=1) / np.abs(before).max(axis=1)
    assert rel.max() < POS_TOL, (onset, rel.max())                                   # (1)
    hip.run_fixed(1, dt)
    x = hip.retrieve_state(0)[0]
    c, d = hip.counts(), hip.diagnostics()
    hip.close()
    assert int(c.particles[0]) == n and d.lost_particles == 0 and d.dropped_particles == 0 and d.overflow_flags == 0   # (3)
    cell = np.rint(x.astype(np.float64) * (1 << bits)).astype(np.int64) - 2          # get_block_id(pos) - 1 - 1 (mgmpm_kernels.cuh:852-863)
    key = (cell[:, 0] << 40) | (cell[:, 1] << 20) | cell[:, 2]
    counts = np.unique(key, return_counts=True)[1]
    surplus = int(np.maximum(0, counts - cap).sum())
    assert surplus == lost_by_the_oracle, (onset, surplus, lost_by_the_oracle, int(counts.max()))   # (2)


def _hip_block_sets(eng, nch, ppb, dx):
    """The HIP engine's own bookkeeping read out of a checkpoint (mpm_checkpoint.inc: header 448 B, then 16-byte aligned sections): per particle
    block its key and the POSITIONS of the particles its advection list names - record {dir tag 5 b, sort key 8 b, slot} -> the source block (the
    block's key + the tag's direction, looked up in the previous numbering) -> that block's bins -> the record of the slot."""
    buf = eng.save_checkpoint()
    hdr = np.frombuffer(buf[:48].tobytes(), np.int32)
    nmodels, pbc, nbc, ebc, prev_count = int(hdr[4]), int(hdr[6]), int(hdr[7]), int(hdr[8]), int(hdr[9])
    assert nmodels == 1
    m0 = np.frombuffer(buf[64:64 + 48].tobytes(), np.int64)           # models[0]: {material, nch | list_in, layout | n | bincount | bincount_src | bucketed}
    bincount_src, bucketed, layout = int(m0[4]), int(m0[5]), int(np.frombuffer(buf[64 + 12:64 + 16].tobytes(), np.int32)[0])
    o = [448]

    def take(nbytes, dtype):
        a = np.frombuffer(buf[o[0]:o[0] + nbytes].tobytes(), dtype)
        o[0] += (nbytes + 15) & ~15
        return a
    cur_keys = take(4 * 3 * ebc, np.int32).reshape(-1, 3)
    prev_keys = take(4 * 3 * prev_count, np.int32).reshape(-1, 3)
    take(4 * 256 * nbc, np.float32)
    size = take(4 * (ebc + 1), np.int32)
    take(4 * (ebc + 1), np.int32)
    binoff_src = take(4 * (prev_count + 1), np.int32)
    take(4 * (ebc + 1), np.int32)
    bins = take(4 * bincount_src * nch * 64, np.float32)
    lists = take(4 * bucketed, np.int32)
    if layout:
        take(4 * pbc * 16, np.int32)
    assert o[0] == buf.size, (o[0], buf.size)
    prev = {tuple(int(v) for v in k): i for i, k in enumerate(prev_keys)}
    pid_bits = int(np.log2(ppb))
    rec_floats = 4 if nch == 4 else 8
    out, at = {}, 0
    for b in range(pbc):
        key = tuple(int(v) for v in cur_keys[b])
        pts = []
        for rec in lists[at:at + int(size[b])]:
            rec = int(rec) & 0xffffffff
            tag, sp = (rec >> (pid_bits + 8)) & 31, rec & (ppb - 1)
            src = prev[(key[0] + tag // 9 - 1, key[1] + (tag // 3) % 3 - 1, key[2] + tag % 3 - 1)]
            base = (int(binoff_src[src]) + (sp >> 6)) * nch * 64 + (sp & 63) * rec_floats
            pts.append(tuple(float(v) * dx for v in bins[base:base + 3]))
        at += int(size[b])
        out[key] = sorted(pts)
    assert at == bucketed
    return out, (pbc, nbc, ebc), cur_keys


def test_g20_block_keys_particle_sets_and_table_against_the_references_statements():
    """G20 on the GPU: the integer bookkeeping of the reference - Partition::insert / reinsert, activate_blocks, the bucket kernels,
    register_neighbor / exterior_blocks, cut out of the reference as text (tests/golden/gen/gen_golden_book.sh) - against the HIP engine's own
    bookkeeping, which is built differently on purpose (block-level advection lists, one compaction kernel, no scans: DESIGN.md 2): after
    initial_setup on the G20 scene the same block counts, the same key SET per tier, per particle block the same SET of particles (read out of the
    engine's lists and bins through a checkpoint), query(active_keys[i]) == i; then 40 substeps of a moving scene: the key sets of the oracle's
    partition, block by block the particles the oracle has there, and the table still consistent."""
    from test_oracle_golden import book_tables, book_groups
    T = book_tables()
    bits, dx = 8, 1.0 / 256.0
    prm = {"volume": dx ** 3 / 8.0, "youngs_modulus": 5e3, "poisson_ratio": 0.4, "rho": 1e3}
    sc = {"name": "g20", "bits": bits, "dt": 1e-4, "config": {"max_ppc": 128}, "models": [{"material": _ffi.FIXED_COROTATED, "xyz": np.ascontiguousarray(T["xyz"]), "v0": (0.0, 0.0, 0.0), "params": prm}]}
    eng = build_engine(sc)
    eng.initial_setup()
    assert eng.api.check_table(eng.ctx) == 0
    sets, counts, keys = _hip_block_sets(eng, 9, 8192, dx)
    eng.close()
    assert counts == (T["pbc0"], T["nbc0"], T["ebc0"])
    tier = lambda k, a, b: set(map(tuple, np.asarray(k)[a:b].tolist()))
    for a, b in ((0, T["pbc0"]), (T["pbc0"], T["nbc0"]), (T["nbc0"], T["ebc0"])):
        assert tier(keys, a, b) == tier(T["keys0"], a, b)
    want = {k: sorted(tuple(float(v) for v in T["xyz"][p]) for p in pids) for k, pids in book_groups(T["keys0"], T["sizes0"], T["buckets0"]).items()}
    assert sets == want                                   # per block: exactly the particles build_particle_cell_buckets + cell_bucket_to_block put there
    # a moving scene, HIP against the oracle driven through the same calls: two spheres in contact, 40 substeps (blocks appear and disappear)
    sc = scenes.two_spheres(bits=6, radius_cells=6.0, gap_cells=0.5, speed=2.0, youngs=2e4)
    eng = build_engine(sc)
    eng.initial_setup()
    eng.run_fixed(40, 1e-4)
    assert eng.api.check_table(eng.ctx) == 0
    ch = eng.counts()
    kh, _ = eng.dump_grid()
    xh = [eng.retrieve_state(m)[0] for m in range(2)]
    eng.close()
    from oracle_ffi import oracle_api
    ora = build_engine(sc, api=oracle_api())
    ora.initial_setup()
    ora.run_fixed(40, 1e-4)
    co = ora.counts()
    ko, _ = ora.dump_grid()
    xo = [ora.retrieve_state(m)[0] for m in range(2)]
    ora.close()
    assert (ch.particle_blocks, ch.neighbor_blocks, ch.exterior_blocks) == (co.particle_blocks, co.neighbor_blocks, co.exterior_blocks)
    assert tier(kh, 0, ch.particle_blocks) == tier(ko, 0, co.particle_blocks) and tier(kh, 0, len(kh)) == tier(ko, 0, len(ko))
    # per block the same particles: the block a particle is bucketed in is a function of its position (get_block_id - 2 over the block size)
    dxs = 1.0 / 64.0
    blk = lambda x: [tuple(v) for v in ((np.rint(x.astype(np.float64) / dxs).astype(np.int64) - 2) >> 2).tolist()]
    for m in range(2):
        from parity_util import match
        idx, _ = match(xo[m].astype(np.float64), xh[m].astype(np.float64))
        assert blk(xh[m][idx]) == blk(xo[m])


def _ckpt_lists(eng, ppb):
    """(size, records per block, pair counts per block) of model 0 out of a checkpoint (layout: see _hip_block_sets)."""
    buf = eng.save_checkpoint()
    hdr = np.frombuffer(buf[:48].tobytes(), np.int32)
    pbc, nbc, ebc, prev_count = int(hdr[6]), int(hdr[7]), int(hdr[8]), int(hdr[9])
    m0 = np.frombuffer(buf[64:64 + 48].tobytes(), np.int64)
    nch = int(np.frombuffer(buf[64 + 4:64 + 8].tobytes(), np.int32)[0])
    bincount_src, bucketed, layout = int(m0[4]), int(m0[5]), int(np.frombuffer(buf[64 + 12:64 + 16].tobytes(), np.int32)[0])
    nmodels = int(hdr[4])
    assert nmodels == 1 and layout == 2, "the model is not in the pair layout"
    o = [448]

    def take(nbytes, dtype):
        a = np.frombuffer(buf[o[0]:o[0] + nbytes].tobytes(), dtype)
        o[0] += (nbytes + 15) & ~15
        return a
    take(4 * 3 * ebc, np.int32), take(4 * 3 * prev_count, np.int32), take(4 * 256 * nbc, np.float32)
    size = take(4 * (ebc + 1), np.int32)
    take(4 * (ebc + 1), np.int32), take(4 * (prev_count + 1), np.int32), take(4 * (ebc + 1), np.int32)
    take(4 * bincount_src * nch * 64, np.float32)
    lists = take(4 * bucketed, np.int32)
    pinfo = take(4 * pbc * 16, np.int32).reshape(pbc, 16)
    out, at = [], 0
    for b in range(pbc):
        out.append((int(size[b]), lists[at:at + int(size[b])].astype(np.int64) & 0xffffffff, pinfo[b]))
        at += int(size[b])
    return out


def test_the_sort_writes_the_pair_layout_it_promises():
    """White-box check of prepare_blocks_kernel's pair layout (claymore_amd/csrc/mpm_kernels.hpp) on a scene in motion - a sand block thrown
    into a corner, 220 substeps: cells fill unevenly, keys have odd counts, chunks run full.  Read out of a checkpoint, for every 512-record chunk of every
    particle block: with n records and Pf full pairs (pairinfo) the S = ceil(n / 128) slices hold L = min(Pf + n1, 64 S) slots, dense; every pair slot's two
    members carry the SAME sort key and the keys of the pair slots ascend (key-major order, the wrap-around rule); the mismatched slots are exactly
    X = max(0, Pf + n1 - 64 S) and hold two different keys; no key has two singles; both members of a pair slot carry the same arena bit - its lane
    parity -, and a single never uses the arena its own key's pair slot of the same slice uses."""
    bits = 6
    sc = scenes.sphere_drop(bits=bits, radius_cells=7.0, center=(0.3, 0.3, 0.3), material=_ffi.SAND)
    sc["models"][0]["params"] = {}
    sc["models"][0]["v0"] = (-6.0, -8.0, -5.0)
    sc.setdefault("config", {})["max_ppc"] = 128
    eng = build_engine(sc)
    eng.initial_setup()
    eng.run_fixed(220, 1e-4)
    ppb = 8192
    blocks = _ckpt_lists(eng, ppb)
    eng.close()
    pid_bits = 13
    from test_pair_layout_model import pair_chunks
    seen = dict(chunks=0, pairs=0, mismatched=0, singles=0, multi_chunk=0, full_slices=0, merged=0)
    for size, recs, pinfo in blocks:
        chunks = pair_chunks(size)                                   # 512-record chunks, the last one with a tail of up to 256 records more
        seen["multi_chunk"] += len(chunks) > 1
        seen["merged"] += bool(chunks) and chunks[-1][1] > 512
        for c, (first, n) in enumerate(chunks):
            assert first == 512 * c
            r = recs[first:first + n]
            key, arena = (r >> pid_bits) & 255, (r >> 30) & 1
            pf = int(pinfo[c])
            n1 = n - 2 * pf
            assert 0 <= pf and n1 >= 0
            S = (n + 127) // 128
            X = max(0, pf + n1 - 64 * S)
            px, L = pf + X, pf + n1 - X
            assert L <= 64 * S and L + px == n
            pos = 0
            slot_key = {}
            for d in range(S):
                ca, cb = L // S + (d < L % S), px // S + (d < px % S)
                ka, kb = key[pos:pos + ca], key[pos + ca:pos + ca + cb]
                aa, ab = arena[pos:pos + ca], arena[pos + ca:pos + ca + cb]
                for ln in range(ca):
                    slot_key[ln * S + d] = (int(ka[ln]), int(kb[ln]) if ln < cb else None, int(aa[ln]), ln, int(ab[ln]) if ln < cb else None)
                pos += ca + cb
                seen["full_slices"] += ca == 64
            assert pos == n and sorted(slot_key) == list(range(L))
            pair_keys = [slot_key[p][0] for p in range(pf)]
            assert all(slot_key[p][0] == slot_key[p][1] for p in range(pf))                       # a pair slot: two records of one key
            assert pair_keys == sorted(pair_keys)                                                 # key-major order
            assert all(slot_key[p][2] == slot_key[p][4] == (slot_key[p][3] & 1) for p in range(pf))   # its arena, on both members: the lane parity (the kernel reads A's bit: a mismatched slot's B carries its own key's, unused)
            assert all(slot_key[p][1] is not None and slot_key[p][0] != slot_key[p][1] for p in range(pf, px))   # a mismatched slot: two singles of different keys
            assert all(slot_key[p][1] is None for p in range(px, L))                              # a single slot: A only
            singles = [slot_key[p][0] for p in range(pf, L)] + [slot_key[p][1] for p in range(pf, px)]
            assert len(singles) == len(set(singles)) == n1                                        # one single per key at most
            from collections import Counter
            cnt = Counter(pair_keys)
            assert all(2 * cnt.get(k, 0) + (k in set(singles)) == int((key == k).sum()) for k in set(key.tolist()))
            # the arena of a single (and of a mismatched slot, which carries its A's rule): not the one its key's pair slot of the same slice uses
            by_slice = {}
            for p in range(pf):
                by_slice.setdefault((p % S, slot_key[p][0]), []).append(slot_key[p][2])
            for p in range(pf, L):
                used = by_slice.get((p % S, slot_key[p][0]), [])
                if len(used) == 1:
                    assert slot_key[p][2] != used[0]
            seen["chunks"] += 1
            seen["pairs"] += pf
            seen["mismatched"] += X
            seen["singles"] += n1
    print("pair layout seen:", seen)
    assert seen["chunks"] > 30 and seen["pairs"] > 2000 and seen["singles"] > 100 and seen["mismatched"] > 0 and seen["multi_chunk"] > 0 and seen["merged"] > 0, seen


def test_mid_size_flow_parity_630k():
    """The flow regime against the ORACLE at a size where every path of G2P2G carries thousands of particles per substep (at scale
    the claim losers, edge lanes, shell atomics - and since round 6 the split pairs - had been compared with the oracle on <= 10 k-particle scenes only;
    the 40 M-particle runs are 8 substeps against the oracle or invariants): a 32 x 77 x 32-cell column of Drucker-Prager sand (630 784 particles, 256^3)
    one block above the floor's wall zone, thrown at it obliquely at (1.5, -4, 0.8) m/s - it hits after ~25 substeps, the lower half yields and spreads,
    2 % of the particles change block per substep -, 160 substeps on both engines (the oracle on 32 threads), particles matched by nearest neighbour
    (unambiguous while the deviation is far below the spacing of 0.5 dx): positions, b, log Jp and the block counts."""
    bits = 8
    sc = scenes.sand_column(bits, (32, 77, 32), min_corner=(112, 12, 112))
    sc["models"][0]["v0"] = (1.5, -4.0, 0.8)
    n = scenes.total_particles(sc)
    assert n == 630784
    nsteps, dt = 160, 1e-4
    hip = run_engine(sc, nsteps, dt)
    ora = run_engine(sc, nsteps, dt, api=oracle_api_threads(32))
    res = {"hip": hip, "oracle": ora, "scene": sc}
    err = match_and_compare(res)
    print("mid-size flow parity:", err)
    assert err["pos_rel"] < POS_TOL, err
    # b and log Jp of a particle that yields in a 4 m/s impact: 3 x the measured 4.0e-4 (relative to max(1, |b|)) / 1.7e-4 - the return mapping
    # amplifies a 1e-7 difference of the trial strain by the stiffness ratio; the positions above do not feel it
    assert err["state_rel"] < 1.2e-3 and err["logjp_abs"] < 5e-4, err
    ch, co = hip["counts"], ora["counts"]
    assert (ch.particle_blocks, ch.neighbor_blocks, ch.exterior_blocks) == (co.particle_blocks, co.neighbor_blocks, co.exterior_blocks)
    xo = ora["state"][0][0]
    assert xo[:, 1].min() < 13.5 / 256 and np.abs(ora["state"][0][2]).max() > 1e-3      # it reached the floor and yielded


@pytest.mark.parametrize("mask", ["0", "0xF"])
def test_both_g2p2g_kernels_pass_the_material_parity_tests(mask):
    """The library ships two G2P2G kernels - one particle per lane (mpm_g2p2g.hpp) and two (mpm_g2p2g_pair.hpp; the default) - selected per material by a mask the library reads once per process (MPM_G2P2G_PAIRS).  The suite runs with the
    default (all four materials on pairs); this test runs the material parity tests against the oracle in a subprocess with the mask forced to NO pairs
    and to ALL FOUR materials, so that neither kernel nor layout rots: both pass the same bounds."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MPM_G2P2G_PAIRS=mask)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_parity_gpu.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "other_materials_parity or mixed_materials or two_spheres_parity or wall_boundary or test_the_sort_writes or checkpoint"],
                       env=env, capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-1500:]
    if mask == "0":       # (without pairs there is no pair layout to read out of a checkpoint: that one test fails by design and is deselected)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_parity_gpu.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                            "-k", "other_materials_parity or mixed_materials or two_spheres_parity or wall_boundary"], env=env, capture_output=True, text=True, timeout=1500)
        tail = r.stdout[-1500:]
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, (mask, tail, r.stderr[-1500:])
