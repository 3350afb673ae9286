"""Pins the CPU oracle's per-particle functions against golden vectors produced by the REFERENCE's own
functions (tests/golden/gen/gen_golden.cpp, run where /root/reference is mounted).  Integer/index helpers
must be bit-exact; fp32 functions must be bit-exact too wherever only IEEE +,-,*,/ and sqrt are involved, and
within a few ulp where libm (logf/expf/powf/sinhf) is involved."""
import ctypes as C
import os

import numpy as np
import pytest

from claymore_amd import _ffi
from oracle_ffi import oracle_api

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def f32(name):
    return np.fromfile(os.path.join(G, name), dtype=np.float32)


def i32(name):
    return np.fromfile(os.path.join(G, name), dtype=np.int32)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def test_g1_bspline_bit_exact():
    api = oracle_api()
    p = f32("g1_bspline_in.f32")
    want = f32("g1_bspline_out.f32")
    got = np.empty_like(want)
    api.raw.mpmo_fn_bspline(ptr(p), p.size, 256.0, ptr(got))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_g2_index_helpers_bit_exact():
    api = oracle_api()
    d = i32("g2_dirs.i32").reshape(27, 7)
    for dx, dy, dz, tag, bx, by, bz in d:
        assert api.raw.mpmo_fn_dir_offset(int(dx), int(dy), int(dz)) == tag
        back = (C.c_int * 3)()
        api.raw.mpmo_fn_dir_components(int(tag), back)
        assert tuple(back) == (bx, by, bz) == (dx, dy, dz)
    x = f32("g2_node_in.f32")
    want = i32("g2_node_out.i32")
    got = np.empty_like(want)
    api.raw.mpmo_fn_node_index(ptr(x), x.size, 256.0, ptr(got))
    assert np.array_equal(got, want)


def test_g3_svd_bit_exact():
    api = oracle_api()
    F = f32("g3_F_in.f32")
    want = f32("g3_svd_out.f32")
    got = np.empty_like(want)
    assert api.raw.mpmo_test_svd(ptr(F), F.size // 9, ptr(got), 0) == 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # and it is an SVD: U S V^T == F to fp32 accuracy
    n = F.size // 9
    U = got.reshape(n, 21)[:, 0:9].reshape(n, 3, 3).transpose(0, 2, 1)
    S = got.reshape(n, 21)[:, 9:12]
    V = got.reshape(n, 21)[:, 12:21].reshape(n, 3, 3).transpose(0, 2, 1)
    Fm = F.reshape(n, 3, 3).transpose(0, 2, 1)
    rec = np.einsum("nij,nj,nkj->nik", U, S, V)
    # 4 Jacobi sweeps are approximate by design (svd.cuh:167): ~1e-6 near identity, up to ~1e-3 for
    # strongly stretched / near-singular inputs
    err = np.abs(rec - Fm).reshape(n, -1).max(axis=1)
    assert err.max() < 2e-3
    assert np.median(err[np.arange(n) % 8 == 0]) < 2e-6  # the near-identity class


def _params(material):
    api = oracle_api()
    p = _ffi.MaterialParams()
    api.default_material(material, 8, C.byref(p))
    vol, mu, lam = f32("g456_params.f32")
    p.volume = float(vol)
    return p


def test_g4_fixed_corotated_bit_exact():
    api = oracle_api()
    F = f32("g3_F_in.f32")
    n = F.size // 9
    want = f32("g4_fc_out.f32").reshape(n, 9)
    got = np.empty((n, 19), dtype=np.float32)
    assert api.test_stress(_ffi.FIXED_COROTATED, C.byref(_params(_ffi.FIXED_COROTATED)), ptr(F), None, n, ptr(got), 0) == 0
    assert np.array_equal(got[:, 9:18].view(np.uint32), want.view(np.uint32))


def _close_ulp(got, want, rel, absolute):
    ok = np.isclose(got, want, rtol=rel, atol=absolute) | (np.isnan(got) & np.isnan(want)) | (np.isinf(got) & np.isinf(want) & (np.sign(got) == np.sign(want)))
    return ok


def test_g5_sand():
    """Sand uses logf/expf: glibc vs the container's libm used for the golden run are the same library, so this
    is expected bit-exact here; the assertion allows 4 ulp so that a different libm does not fail it."""
    api = oracle_api()
    F = f32("g3_F_in.f32")
    n = F.size // 9
    lj = f32("g5_sand_logjp_in.f32")
    want = f32("g5_sand_out.f32").reshape(n, 19)
    got = np.empty((n, 19), dtype=np.float32)
    idx = np.arange(n)
    coh = np.where(idx % 5 == 4, np.float32(0.01), np.float32(0.0))
    vc = (idx % 7 != 6)
    for c in (0.0, 0.01):
        for v in (True, False):
            sel = np.where((coh == np.float32(c)) & (vc == v))[0]
            if sel.size == 0:
                continue
            p = _params(_ffi.SAND)
            p.cohesion = c
            p.volume_correction = int(v)
            Fi = np.ascontiguousarray(F.reshape(n, 9)[sel])
            li = np.ascontiguousarray(lj[sel])
            out = np.empty((sel.size, 19), dtype=np.float32)
            assert api.test_stress(_ffi.SAND, C.byref(p), ptr(Fi), ptr(li), sel.size, ptr(out), 0) == 0
            got[sel] = out
    scale = np.abs(want).max(axis=1, keepdims=True) + 1e-30
    assert _close_ulp(got, want, 5e-7, 0).all() or (np.abs(got - want) / scale).max() < 1e-6


def test_g6_nacc():
    api = oracle_api()
    F = f32("g3_F_in.f32")
    n = F.size // 9
    lj = f32("g6_nacc_logjp_in.f32")
    want = f32("g6_nacc_out.f32").reshape(n, 19)
    got = np.empty((n, 19), dtype=np.float32)
    idx = np.arange(n)
    hard = (idx % 7 != 6)
    for h in (True, False):
        sel = np.where(hard == h)[0]
        p = _params(_ffi.NACC)
        p.hardening_on = int(h)
        Fi = np.ascontiguousarray(F.reshape(n, 9)[sel])
        li = np.ascontiguousarray(lj[sel])
        out = np.empty((sel.size, 19), dtype=np.float32)
        assert api.test_stress(_ffi.NACC, C.byref(p), ptr(Fi), ptr(li), sel.size, ptr(out), 0) == 0
        got[sel] = out
    finite = np.isfinite(want).all(axis=1)
    assert (np.isfinite(got).all(axis=1) == finite).all()
    scale = np.abs(want[finite]).max(axis=1, keepdims=True) + 1e-30
    assert (np.abs(got[finite] - want[finite]) / scale).max() < 1e-6


def test_g8_compute_dt_bit_exact():
    api = oracle_api()
    rows = f32("g8_dt_in.f32").reshape(-1, 4)
    want = f32("g8_dt_out.f32")
    dx = np.float32(1.0 / 256.0)
    for (mv, cur, nxt, dtd), w in zip(rows, want):
        got = np.float32(api.raw.mpmo_fn_compute_dt(float(mv), float(cur), float(nxt), float(dtd), float(dx), 0.5))
        assert got.view(np.uint32) == np.float32(w).view(np.uint32)


def test_g9_matrix_helpers_bit_exact():
    api = oracle_api()
    F = f32("g3_F_in.f32").reshape(-1, 9)
    want = f32("g9_mat_out.f32").reshape(-1, 36)
    for i in range(want.shape[0]):
        a = np.ascontiguousarray(F[i])
        b = np.ascontiguousarray(F[i + 1])
        s = np.array([a[0], b[4], a[8]], dtype=np.float32)
        out = np.empty(36, dtype=np.float32)
        api.raw.mpmo_fn_mat(ptr(a), ptr(b), ptr(s), ptr(out))
        assert np.array_equal(out.view(np.uint32), want[i].view(np.uint32)), i


def test_g7_jfluid_closed_form():
    """G7: the J-fluid block is inline in the reference's g2p2g kernel (mgmpm_kernels.cuh:476-505) and cannot be
    host-compiled; it is pinned by closed-form cases in float64."""
    api = oracle_api()
    rng = np.random.RandomState(7)
    n = 256
    J = (0.5 + rng.rand(n)).astype(np.float32)
    J[:8] = 0.1  # clamp region
    A = (1e-3 * rng.randn(n, 9)).astype(np.float32)
    A[:8] = -1.0
    dt, d_inv, vol, bulk, gamma, visc = 1e-4, 4.0 * 256.0 * 256.0, 7.45e-9, 4e4, 7.15, 0.01
    out = np.empty((n, 10), dtype=np.float32)
    api.raw.mpmo_fn_jfluid(ptr(J), ptr(A), n, dt, d_inv, vol, bulk, gamma, visc, ptr(out))
    J64 = J.astype(np.float64)
    A64 = A.astype(np.float64)
    Jn = J64 + (A64[:, 0] + A64[:, 4] + A64[:, 8]) * dt * d_inv * J64
    Jn = np.maximum(Jn, 0.1)
    press = bulk * (Jn ** (-gamma) - 1.0)
    voln = Jn * vol
    sym = (A64.reshape(n, 3, 3) + A64.reshape(n, 3, 3).transpose(0, 2, 1)) * d_inv * visc
    want = (sym - press[:, None, None] * np.eye(3)[None]) * voln[:, None, None]
    assert np.allclose(out[:, 0], Jn, rtol=2e-6)
    # J^-gamma - 1 cancels near J = 1: allow an absolute slack of a few fp32 ulps of bulk * voln
    slack = (2e-6 * bulk * voln * gamma)[:, None]
    assert (np.abs(out[:, 1:] - want.reshape(n, 9)) <= 2e-5 * np.abs(want.reshape(n, 9)) + slack).all()


def test_g7_jfluid_reference_statements():
    """G7 against the reference's own statements: the J update, the 0.1 clamp, the Tait pressure and the nine contrib[] lines of
    calculate_contribution_and_store_particle_data<J_FLUID> (Projects/GMPM/mgmpm_kernels.cuh:473-504), cut out of the file as text by
    tests/golden/gen/gen_golden.sh and compiled between locals of the names they use.  powf comes from the same libm on both sides."""
    api = oracle_api()
    vol, bulk, gamma, visc, dt, d_inv = (float(v) for v in f32("g7_jfluid_params.f32"))
    rows = f32("g7_jfluid_in.f32").reshape(-1, 10)
    want = f32("g7_jfluid_out.f32").reshape(-1, 10)
    n = rows.shape[0]
    J, A = np.ascontiguousarray(rows[:, 0]), np.ascontiguousarray(rows[:, 1:])
    out = np.empty((n, 10), dtype=np.float32)
    api.raw.mpmo_fn_jfluid(ptr(J), ptr(A), n, dt, d_inv, vol, bulk, gamma, visc, ptr(out))
    assert (want[:, 0] == np.float32(0.1)).sum() >= 8 and (want[:, 0] > 0.5).sum() > 400        # the clamp and the regular branch
    assert np.array_equal(out[:, 0].view(np.uint32), want[:, 0].view(np.uint32))
    assert _close_ulp(out[:, 1:], want[:, 1:], 5e-7, 0).all(), np.abs(out[:, 1:] - want[:, 1:]).max()


# ---- the MGSP project's host-compilable functions (tests/golden/gen/gen_golden_mgsp.cpp) ---------------------------------------------
def test_g12_mgsp_compute_dt_bit_exact():
    """compute_dt of the MGSP project (Projects/MGSP/utility_funcs.hpp:32-55: CFL 0.3, 0.51 frame-remainder rule): the oracle's
    restatement AND the product package's Python restatement (MgspRank.compute_dt_mgsp, the dt rule of the gloo / torch driver)
    against the reference's own outputs.  (The C++ group driver's mpm_group_compute_dt needs a context: tests/test_mgsp_gpu.py.)"""
    import types
    from claymore_amd.mgsp import MgspRank
    api = oracle_api()
    rows = f32("g12_mgsp_dt_in.f32").reshape(-1, 4)
    want = f32("g12_mgsp_dt_out.f32")
    assert rows.shape[0] == want.size > 1000
    dx = np.float32(1.0 / 256.0)                                  # Projects/MGSP/settings.h: DOMAIN_BITS = 8
    shim = types.SimpleNamespace(eng=types.SimpleNamespace(dx=float(dx)))
    for (mv, cur, nxt, dtd), w in zip(rows, want):
        got = np.float32(api.raw.mpmo_fn_compute_dt_mgsp(float(mv), float(cur), float(nxt), float(dtd), float(dx)))
        assert got.view(np.uint32) == np.float32(w).view(np.uint32), (mv, cur, nxt, dtd, got, w)
        py = np.float32(MgspRank.compute_dt_mgsp(shim, float(mv), float(cur), float(nxt), float(dtd)))
        assert py.view(np.uint32) == np.float32(w).view(np.uint32), (mv, cur, nxt, dtd, py, w)
    assert (want == 0).any() and (want < rows[:, 3]).any() and (want == rows[:, 3]).any()      # every branch is represented


def test_g13_rot_angle_to_matrix():
    """SignedDistanceGrid::rot_angle_to_matrix (boundary_condition.cuh:67-91): cosf / sinf come from libm on both sides."""
    api = oracle_api()
    rows = f32("g13_rot_in.f32").reshape(-1, 2)
    want = f32("g13_rot_out.f32").reshape(-1, 9)
    got = np.empty(9, dtype=np.float32)
    for (a, dim), w in zip(rows, want):
        api.raw.mpmo_fn_rot_angle_to_matrix(float(a), int(dim), ptr(got))
        assert np.abs(got - w).max() <= 2 * np.spacing(np.float32(1.0)), (a, dim, got, w)
        assert np.array_equal((got == 0), (w == 0)) and np.array_equal(np.sign(got), np.sign(w))


def hashed_sdf_field(n):
    """The signed-distance field of the G14 / G15 goldens (gen_golden_mgsp.cpp: field_value): node (i, j, k), channel c -> a float in
    [-1, 1) from integer hashing - exact in uint32 and float32, so numpy rebuilds the generator's 256^3 x 4 field bit for bit."""
    i = (np.arange(n, dtype=np.uint32) * np.uint32(73856093))[:, None, None]
    j = (np.arange(n, dtype=np.uint32) * np.uint32(19349663))[None, :, None]
    k = (np.arange(n, dtype=np.uint32) * np.uint32(83492791))[None, None, :]
    base = i ^ j ^ k
    out = []
    for c in range(4):
        h = base ^ np.uint32((c * 0x9E3779B1) & 0xFFFFFFFF)
        h = h * np.uint32(2654435761)
        h ^= h >> np.uint32(15)
        h = h * np.uint32(2246822519)
        h ^= h >> np.uint32(13)
        out.append(((h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)) * np.float32(2.0) - np.float32(1.0))
    return out


@pytest.fixture(scope="module")
def collision_oracle():
    """An oracle context on the MGSP project's grid (256^3, wall zone of 2 blocks) with the hashed field installed."""
    api = oracle_api()
    cfg = _ffi.Config()
    assert api.default_config(8, C.byref(cfg)) == 0
    ctx = C.c_void_p()
    assert api.create(C.byref(cfg), 0, C.byref(ctx)) == 0
    field = hashed_sdf_field(256)

    def install(obj):
        assert api.set_collision_object(ctx, C.byref(obj), ptr(field[0]), ptr(field[1]), ptr(field[2]), ptr(field[3])) == 0

    yield api, ctx, install
    api.destroy(ctx)


def test_g15_signed_distance_and_normal(collision_oracle):
    """get_signed_distance_and_normal / query_sdf (boundary_condition.cuh:99-146) at 768 points, a part of them in the wall zone or on
    a node plane: the wall-zone test and the hit flag exactly, the interpolated distance and the normal bit for bit."""
    api, ctx, install = collision_oracle
    obj = _ffi.CollisionObject()
    api.default_collision_object(C.byref(obj))
    install(obj)
    x = f32("g15_sdf_x.f32")
    want = f32("g15_sdf_out.f32").reshape(-1, 6)
    got = np.empty_like(want)
    assert api.raw.mpmo_fn_query_sdf(ctx, ptr(x), want.shape[0], ptr(got)) == 0
    assert np.array_equal(got[:, 0], want[:, 0]) and np.array_equal(got[:, 1], want[:, 1])
    assert 0.05 < want[:, 0].mean() < 0.98 and 0.2 < want[want[:, 0] == 1, 1].mean() < 0.8
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_g14_detect_and_resolve_collision(collision_oracle):
    """detect_and_resolve_collision (boundary_condition.cuh:164-248): STICKY / SLIP / SEPARATE x friction 0 / 0.3 x (object at rest,
    t = 0 | translating, rotating, growing object at t = 0.37 with a start orientation), 384 grid nodes each."""
    api, ctx, install = collision_oracle
    cfgs = f32("g14_col_cfg.f32").reshape(-1, 23)
    nodes = i32("g14_col_nodes.i32")
    vin = f32("g14_col_vel_in.f32")
    want = f32("g14_col_vel_out.f32").reshape(cfgs.shape[0], -1)
    m = nodes.size // 3
    assert cfgs.shape[0] == 12
    changed = 0
    for cf, w in zip(cfgs, want):
        obj = _ffi.CollisionObject()
        api.default_collision_object(C.byref(obj))
        obj.type, obj.friction, obj.scale, obj.dsdt = int(cf[0]), float(cf[1]), float(cf[2]), float(cf[3])
        for d in range(3):
            obj.trans[d], obj.trans_vel[d], obj.omega[d] = float(cf[4 + d]), float(cf[7 + d]), float(cf[10 + d])
        for e in range(9):
            obj.rot_mat[e] = float(cf[13 + e])
        install(obj)
        vel = vin.copy()
        assert api.raw.mpmo_fn_collision_resolve(ctx, ptr(nodes), m, float(cf[22]), ptr(vel)) == 0
        hit = (w.reshape(m, 3) != vin.reshape(m, 3)).any(axis=1)
        changed += int(hit.sum())
        assert 0.2 < hit.mean() < 0.8                       # the hashed field puts about half of the nodes inside the object
        if cf[22] == 0.0:                                   # t = 0: only IEEE +, -, *, / and sqrt between the inputs and the result
            assert np.array_equal(vel.view(np.uint32), w.view(np.uint32)), (cf[:4],)
        else:                                               # t != 0: cosf / sinf of the rotation angles go through libm
            assert np.array_equal(vel != vin, w != vin)
            assert np.abs(vel - w).max() <= 4e-6 * max(1.0, np.abs(w).max()), (cf[:4], np.abs(vel - w).max())
    assert changed > 12 * m // 4


# ---- G16-G19: the BODY of the reference's g2p2g kernel and of its grid update at statement level ----------------------------
# tests/golden/gen/gen_golden_kernel.sh cuts the statements of Projects/GMPM/mgmpm_kernels.cuh:772-838 (stencil base, weights, gather,
# advection), :518-663 (the FC / sand / NACC bodies: F <- (I + dt grad v) F, compute_stress, what they store), :845-905 (the contrib
# line, the re-bucketing arguments, the arena test, the 27 x 4 scatter) and :353-388 (the grid update's cell arithmetic) out of the
# reference AS TEXT and runs them between locals, one particle per row, on five velocity arenas.  The oracle's pipeline goes through
# the SAME function (orc_particle_body / orc_grid_cell in oracle/mpm_oracle.c) that mpmo_fn_particle_step / mpmo_fn_grid_cells expose.
def kernel_rows():
    par = f32("g16_params.f32")
    names = "bits vol mass mu lam cohesion beta_sand yield_surface volume_correction bm xi msqr hardening_on dt new_dt beta_nacc E nu rho bulk gamma viscosity".split()
    P = dict(zip(names, [float(v) for v in par]))
    arenas = f32("g16_arenas.f32").reshape(-1, 3 * 512)
    rin = f32("g16_particle_in.f32").reshape(-1, 15)
    return P, arenas, rin, f32("g16_particle_out.f32").reshape(-1, 154), i32("g16_particle_out.i32").reshape(-1, 14)


def kernel_params(P, material):
    api = oracle_api()
    p = _ffi.MaterialParams()
    assert api.default_material(material, int(P["bits"]), C.byref(p)) == 0
    p.rho, p.volume, p.youngs_modulus, p.poisson_ratio = P["rho"], P["vol"], P["E"], P["nu"]
    p.cohesion, p.yield_surface, p.volume_correction = P["cohesion"], P["yield_surface"], int(P["volume_correction"])
    p.beta = P["beta_sand"] if material == _ffi.SAND else P["beta_nacc"]
    p.xi, p.msqr, p.hardening_on = P["xi"], P["msqr"], int(P["hardening_on"])
    p.bulk, p.gamma, p.viscosity = P["bulk"], P["gamma"], P["viscosity"]
    return p


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.mark.parametrize("material", [_ffi.J_FLUID, _ffi.FIXED_COROTATED, _ffi.SAND, _ffi.NACC])
def test_g16_g17_g18_kernel_body_against_the_references_own_statements(material):
    api = oracle_api()
    fn = api.raw.mpmo_fn_particle_step
    fn.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    P, arenas, rin, want_f, want_i = kernel_rows()
    p = kernel_params(P, material)
    # the derived constants are the reference's (particle_buffer.cuh:155-162, :187-188, :255)
    assert np.float32(P["vol"]) * np.float32(P["rho"]) == np.float32(P["mass"])
    seen = dict(rows=0, crossed=0, discarded=0)
    for a in range(arenas.shape[0]):
        sel = (rin[:, 0] == material) & (rin[:, 1] == a)
        rows = np.ascontiguousarray(rin[sel][:, 2:])
        n = rows.shape[0]
        assert n == 48
        got_f, got_i = np.zeros((n, 154), np.float32), np.zeros((n, 14), np.int32)
        assert fn(material, C.byref(p), int(P["bits"]), ptr(arenas[a]), ptr(rows), n, P["dt"], P["new_dt"], ptr(got_f), ptr(got_i)) == 0
        wf, wi = want_f[sel], want_i[sel]
        # every integer the body produces: stencil base, its image in the arena, the cell and direction tag handed to add_advection (:863),
        # the new base in the arena, whether the contribution is discarded (:877-885)
        assert np.array_equal(got_i, wi), (a, np.argwhere(got_i != wi)[:5])
        # G16 gather / advect (:772-838): vel, A, the advected position - IEEE +, -, * only
        assert same_bits(got_f[:, :15], wf[:, :15]), a
        fin = np.isfinite(wf).all(axis=1)        # (the violent arenas drive a few sand / NACC rows to inf / nan - in the reference's statements too)
        assert np.array_equal(np.isfinite(got_f).all(axis=1), fin), a
        if material == _ffi.J_FLUID:   # J' exact (IEEE +, *), the pressure goes through powf: G7's bound on the stress and what follows from it
            assert same_bits(got_f[:, 15], wf[:, 15]), a
            scale = np.maximum(1e-30, np.abs(wf[fin]).max(axis=0, keepdims=True))
            assert (np.abs(got_f[fin] - wf[fin]) / scale)[:, np.r_[16:24, 25:154]].max() <= 2e-6, a
        elif material == _ffi.FIXED_COROTATED:    # (the FC body stores no log Jp: column 24 is whatever the row carried)
            cols = np.r_[15:24, 25:154]
            assert same_bits(got_f[:, cols], wf[:, cols]), a
        else:        # sand / NACC go through logf / expf / sinhf: same libm here -> same bits in this image; the bound is G5 / G6's
            scale = np.maximum(1e-30, np.abs(wf[fin]).max(axis=0, keepdims=True))
            assert (np.abs(got_f[fin] - wf[fin]) / scale).max() <= 2e-6, a
        seen["rows"] += n
        seen["crossed"] += int((wi[:, 9] != 13).sum())
        seen["discarded"] += int(wi[:, 13].sum())
    # the set exercises what it claims to: particles that change block, particles thrown out of the arena
    assert seen["rows"] == 240 and seen["crossed"] >= 15 and seen["discarded"] >= 8, seen


def test_g19_grid_update_cell_arithmetic_bit_exact():
    api = oracle_api()
    fn = api.raw.mpmo_fn_grid_cells
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_void_p]
    fn.restype = None
    P = kernel_rows()[0]
    cells = f32("g19_gridcell_in.f32").reshape(-1, 5)
    want = f32("g19_gridcell_out.f32").reshape(-1, 4)
    got = np.zeros_like(want)
    fn(ptr(np.ascontiguousarray(cells)), cells.shape[0], -9.8, P["dt"], ptr(got))
    live = cells[:, 0] > 0                      # (a cell without mass is left alone: its velocity entries are whatever they were)
    assert same_bits(got[live][:, :3], want[live][:, :3])
    assert same_bits(got[:, 3], want[:, 3])     # |v|^2, +inf where the arithmetic produced a NaN (:380-383)
    assert np.isinf(want[:, 3]).sum() >= 6 and (cells[:, 0] <= 0).sum() >= 100 and (cells[:, 4] > 0).sum() > 400


# ---- G20: the INTEGER BOOKKEEPING of the particle path at statement level (tests/golden/gen/gen_golden_book.sh / .cpp) ----------------
# Partition::insert / query / reinsert (hash_table.cuh:118-135), add_advection (particle_buffer.cuh:100-135), activate_blocks,
# build_particle_cell_buckets, cell_bucket_to_block, compute_bin_capacity, register_neighbor / exterior_blocks, mark_active_particle_blocks,
# update_partition, update_buckets (mgmpm_kernels.cuh:21-151, :954-1000) and exclusive_scan_inverse (MappingKernels.cuh:44-55), cut out of the
# reference AS TEXT and run over a serial thread loop on 174 particles / 9 -> 25 particle blocks.  A serial loop fixes one of the orders the
# GPU's atomics may produce: block keys and per-block particle sets are compared as SETS, everything order-free exactly.
def book_tables():
    hdr = i32("g20_hdr.i32")
    names = "n pbc0 nbc0 ebc0 pbc1 nbc1 ebc1 ncalls".split()
    T = dict(zip(names, [int(v) for v in hdr]))
    T["xyz"] = f32("g20_xyz.f32").reshape(-1, 3)
    T["delta"] = i32("g20_delta.i32").reshape(-1, 3)
    for k, w in (("keys0", 3), ("keys1", 3), ("adv", 7)):
        T[k] = i32(f"g20_{k}.i32").reshape(-1, w)
    for k in ("sizes0", "buckets0", "binoff0", "marks", "scan", "scan_inverse", "sizes1", "buckets1", "binoff1"):
        T[k] = i32(f"g20_{k}.i32")
    return T


def book_groups(keys, sizes, buckets):
    """{block key: list of bucket entries}"""
    out, o = {}, 0
    for b, n in enumerate(sizes):
        out[tuple(int(v) for v in keys[b])] = [int(v) for v in buckets[o:o + n]]
        o += n
    assert o == len(buckets)
    return out


def book_resolve(groups1, groups0, ppb=8192):
    """The particle ids behind the rebuilt buckets: an entry (dirtag * ppb) | particle_id_in_block names slot particle_id_in_block of the SOURCE
    block, which lies at the new block's key + dir_components(dirtag) (mgmpm_kernels.cuh:860-862: dirtag = dir_offset(old block - new block))."""
    out = {}
    for key, entries in groups1.items():
        pids = []
        for e in entries:
            tag, pidib = e // ppb, e % ppb
            d = (tag // 9 - 1, (tag // 3) % 3 - 1, tag % 3 - 1)
            src = tuple(key[i] + d[i] for i in range(3))
            pids.append(groups0[src][pidib])
        out[key] = sorted(pids)
    return out


def test_g20_integer_bookkeeping_against_the_references_own_statements():
    T = book_tables()
    assert (T["n"], T["pbc0"], T["pbc1"]) == (174, 9, 25) and T["ncalls"] == T["n"]
    api = oracle_api()
    cfg = _ffi.Config()
    assert api.default_config(8, C.byref(cfg)) == 0          # the reference's compile-time grid: 256^3, 128 particles per cell, bins of 32
    cfg.max_ppc = 128
    ctx = C.c_void_p()
    assert api.create(C.byref(cfg), 0, C.byref(ctx)) == 0
    p = _ffi.MaterialParams()
    assert api.default_material(_ffi.FIXED_COROTATED, 8, C.byref(p)) == 0
    xyz = np.ascontiguousarray(T["xyz"])
    v0 = (C.c_float * 3)(0, 0, 0)
    assert api.add_model(ctx, _ffi.FIXED_COROTATED, C.byref(p), ptr(xyz), xyz.shape[0], v0, None) == 0
    assert api.initial_setup(ctx) == 0
    dump = api.raw.mpmo_fn_book_dump
    dump.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
    dump.restype = C.c_int
    adv = api.raw.mpmo_fn_book_advect
    adv.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    adv.restype = C.c_int

    def state(which):
        cnt, keys = np.zeros(3, np.int32), np.zeros((1024, 3), np.int32)
        sizes, buckets, binoff = np.zeros(1024, np.int32), np.zeros(4096, np.int32), np.zeros(1025, np.int32)
        assert dump(ctx, 0, which, ptr(cnt), ptr(keys), ptr(sizes), ptr(buckets), ptr(binoff)) == 0
        pbc, nbc, ebc = (int(v) for v in cnt)
        return pbc, nbc, ebc, keys[:ebc], sizes[:pbc], buckets[:int(sizes[:pbc].sum())], binoff[:pbc + 1]

    def tiers(keys, pbc, nbc):
        s = [set(map(tuple, keys[a:b].tolist())) for a, b in ((0, pbc), (pbc, nbc), (nbc, len(keys)))]
        assert sum(len(t) for t in s) == len(keys)            # no key twice (Partition::insert claims a key once)
        return s

    # A + B: the initial partition and buckets (activate_blocks, register_*, build_particle_cell_buckets, cell_bucket_to_block)
    pbc, nbc, ebc, keys, sizes, buckets, binoff = state(0)
    assert (pbc, nbc, ebc) == (T["pbc0"], T["nbc0"], T["ebc0"])
    assert tiers(keys, pbc, nbc) == tiers(T["keys0"], T["pbc0"], T["nbc0"])
    g0o, g0g = book_groups(keys, sizes, buckets), book_groups(T["keys0"], T["sizes0"], T["buckets0"])
    assert {k: sorted(v) for k, v in g0o.items()} == {k: sorted(v) for k, v in g0g.items()}      # per block: the same particle ids
    # compute_bin_capacity + exclusive scan (bins of 32): a function of the sizes in the engine's own block order
    assert np.array_equal(binoff, np.concatenate([[0], np.cumsum((sizes + 31) // 32)]))
    assert np.array_equal(T["binoff0"], np.concatenate([[0], np.cumsum((T["sizes0"] + 31) // 32)]))
    serial_order = np.array_equal(keys, T["keys0"])          # (a serial oracle and a serial thread loop happen to insert in the same order)
    # C: add_advection for every bucketed particle, D: the rebuild
    assert adv(ctx, 0, ptr(np.ascontiguousarray(T["delta"]))) == 0
    assert api.rebuild_partition(ctx, None) == 0
    assert api.raw.mpmo_check_table(ctx) == 0                 # query(active_keys[i]) == i (update_partition's reinsert + the two register passes)
    pbc, nbc, ebc, keys, sizes, buckets, binoff = state(1)
    assert (pbc, nbc, ebc) == (T["pbc1"], T["nbc1"], T["ebc1"])
    assert tiers(keys, pbc, nbc) == tiers(T["keys1"], T["pbc1"], T["nbc1"])
    r_o = book_resolve(book_groups(keys, sizes, buckets), g0o)
    r_g = book_resolve(book_groups(T["keys1"], T["sizes1"], T["buckets1"]), g0g)
    assert r_o == r_g and sum(len(v) for v in r_g.values()) == T["n"]        # every particle in the block the reference's statements put it in, none lost
    # the direction tags themselves, per particle (a particle's entry carries dir_offset(old block - new block))
    tags = lambda groups1, groups0: sorted((groups0[tuple(k[i] + (e // 8192 // 9 - 1, (e // 8192 // 3) % 3 - 1, e // 8192 % 3 - 1)[i] for i in range(3))][e % 8192], e // 8192) for k, es in groups1.items() for e in es)
    assert tags(book_groups(keys, sizes, buckets), g0o) == tags(book_groups(T["keys1"], T["sizes1"], T["buckets1"]), g0g) == sorted((int(r[2]), int(r[6])) for r in T["adv"])
    assert np.array_equal(binoff, np.concatenate([[0], np.cumsum((sizes + 31) // 32)]))
    # mark_active_particle_blocks, the exclusive scan and exclusive_scan_inverse are functions of the OLD block order: internally consistent in the
    # table, and equal to the oracle's whenever the two old orders coincide
    marks, scan, inv = T["marks"], T["scan"], T["scan_inverse"]
    assert np.array_equal(scan, np.concatenate([[0], np.cumsum(marks)[:-1]])) and scan[-1] == T["pbc1"]
    assert np.array_equal(inv, np.flatnonzero(marks[:-1])) and len(inv) == T["pbc1"]
    assert np.array_equal(T["keys1"][:T["pbc1"]], T["keys0"][inv])            # update_partition: new block i is old block sources[i]
    if serial_order:
        assert np.array_equal(keys[:pbc], T["keys1"][:T["pbc1"]])
    api.destroy(ctx)
