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
