#!/usr/bin/env python
"""x86 check of mpm_device_math.hpp (built by tools/hostcheck/build.sh) against the CPU oracle: the same comparisons
tests/test_parity_gpu.py makes on the GPU, plus random deformation gradients.  Development aid only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from claymore_amd import _ffi  # noqa: E402
from oracle_ffi import oracle_api  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


class MC(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("mass", "volume", "mu", "lam", "bulk", "gamma", "viscosity", "cohesion", "beta",
                                         "yield_surface", "bm", "xi", "msqr", "log_jp0")] + [("volume_correction", C.c_int), ("hardening_on", C.c_int)]


def make_mc(p):
    e, nu = np.float32(p.youngs_modulus), np.float32(p.poisson_ratio)
    mc = MC()
    mc.volume = p.volume
    mc.mass = p.volume * p.rho
    mc.lam = e * nu / ((1 + nu) * (1 - 2 * nu))
    mc.mu = e / (2 * (1 + nu))
    mc.bm = np.float32(2.0 / 3.0) * mc.mu + mc.lam
    for k in ("bulk", "gamma", "viscosity", "cohesion", "beta", "yield_surface", "xi", "msqr", "log_jp0", "volume_correction", "hardening_on"):
        setattr(mc, k, getattr(p, k))
    return mc


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def f32(name):
    return np.fromfile(os.path.join(G, name), dtype=np.float32)


def main():
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhostmath.so"))
    api = oracle_api()
    rng = np.random.default_rng(7)
    F = f32("g3_F_in.f32").reshape(-1, 9)
    n0 = F.shape[0]
    # random near-identity + moderate strain + rotations (the regime the pipeline lives in)
    def rand_F(n, amp):
        A = rng.standard_normal((n, 3, 3)) * amp
        Q, _ = np.linalg.qr(rng.standard_normal((n, 3, 3)))
        Q[np.linalg.det(Q) < 0, :, 0] *= -1
        Fm = Q @ (np.eye(3) + A)
        return np.ascontiguousarray(Fm.transpose(0, 2, 1).reshape(n, 9).astype(np.float32))
    sets = {"golden": F, "rand 1e-3": rand_F(20000, 1e-3), "rand 0.05": rand_F(20000, 0.05), "rand 0.3": rand_F(20000, 0.3)}
    vol = float(f32("g456_params.f32")[0])
    for mat, name in ((_ffi.FIXED_COROTATED, "FC"), (_ffi.SAND, "SAND"), (_ffi.NACC, "NACC")):
        p = _ffi.MaterialParams()
        api.default_material(mat, 8, C.byref(p))
        p.volume = vol
        mc = make_mc(p)
        for sname, Fs in sets.items():
            n = Fs.shape[0]
            lj = (rng.standard_normal(n) * 0.01).astype(np.float32) if mat != _ffi.FIXED_COROTATED else None
            if mat == _ffi.NACC:
                lj = (p.log_jp0 + rng.standard_normal(n) * 0.01).astype(np.float32)
            want = np.empty((n, 19), np.float32)
            got = np.empty((n, 19), np.float32)
            assert api.test_stress(mat, C.byref(p), ptr(Fs), ptr(lj) if lj is not None else None, n, ptr(want), 0) == 0
            lib.host_test_stress(mat, C.byref(mc), ptr(Fs), ptr(lj) if lj is not None else None, C.c_size_t(n), ptr(got))
            fin = np.isfinite(want).all(axis=1) & np.isfinite(got).all(axis=1)
            det = np.linalg.det(Fs.reshape(n, 3, 3).astype(np.float64))
            ok = fin & (det > 1e-3)
            sF = np.maximum(1.0, np.abs(want[:, 0:9]).max(axis=1))
            eF = np.abs(got[:, 0:9] - want[:, 0:9]).max(axis=1) / sF
            sP = np.abs(want[:, 9:18]).max(axis=1) + 1e-30
            eP = np.abs(got[:, 9:18] - want[:, 9:18]).max(axis=1) / sP
            # stress error relative to the stiffness scale (what moves particles): |dPF| / (vol * E)
            eE = np.abs(got[:, 9:18] - want[:, 9:18]).max(axis=1) / (vol * p.youngs_modulus)
            eL = np.abs(got[:, 18] - want[:, 18])
            nf = (~np.isfinite(got).all(axis=1) & np.isfinite(want).all(axis=1)).sum()
            print(f"{name:5s} {sname:10s} n={n:6d} ok={ok.sum():6d}  F: med {np.median(eF[ok]):.1e} p99 {np.quantile(eF[ok], .99):.1e} max {eF[ok].max():.1e} | "
                  f"PF rel: med {np.median(eP[ok]):.1e} p99 {np.quantile(eP[ok], .99):.1e} | PF/(vol E): med {np.median(eE[ok]):.1e} max {eE[ok].max():.1e} | logJp max {eL[ok].max():.1e} | newly non-finite {nf}")
    # eigen-decomposition sanity
    Fs = sets["rand 0.3"]
    n = Fs.shape[0]
    out = np.empty((n, 12), np.float32)
    lib.host_test_eig(ptr(Fs), C.c_size_t(n), ptr(out))
    U = out[:, :9].reshape(n, 3, 3).transpose(0, 2, 1).astype(np.float64)
    lam = out[:, 9:].astype(np.float64)
    Fm = Fs.reshape(n, 3, 3).transpose(0, 2, 1).astype(np.float64)
    b = Fm @ Fm.transpose(0, 2, 1)
    rec = np.einsum("nij,nj,nkj->nik", U, lam, U)
    print("eig: |U lam U^T - b| max", np.abs(rec - b).max(), " |U^T U - I| max", np.abs(U.transpose(0, 2, 1) @ U - np.eye(3)).max())


if __name__ == "__main__":
    main()
