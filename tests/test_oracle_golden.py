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
