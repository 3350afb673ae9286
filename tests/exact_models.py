"""Float64 closed forms of the constitutive models (ground truth for the device-function tests): the reference's own
float results deviate from these by its approximate 4-sweep SVD (Library/MnBase/Math/Matrix/svd.cuh:167); the device
math must be at least as close to the truth as the reference is.  Test infrastructure only."""
import numpy as np


def _svd_rot(F):
    """F = U S V^T with U, V rotations; the smallest singular value carries the sign of det F (svd.cuh:590-770)."""
    U, S, Vt = np.linalg.svd(F)
    V = Vt.transpose(0, 2, 1)
    du, dv = np.linalg.det(U) < 0, np.linalg.det(V) < 0
    U[du, :, 2] *= -1
    V[dv, :, 2] *= -1
    S = S.copy()
    S[du ^ dv, 2] *= -1
    return U, S, V


def to_mats(flat9):
    """column-major 9-vectors -> (n, 3, 3) float64 matrices"""
    return np.asarray(flat9, dtype=np.float64).reshape(-1, 3, 3).transpose(0, 2, 1)


def to_flat(M):
    return M.transpose(0, 2, 1).reshape(-1, 9)


def fixed_corotated(F9, mu, lam, volume):
    """P F^T volume, constitutive_models.cuh:36-73"""
    F = to_mats(F9)
    U, S, V = _svd_rot(F)
    J = S.prod(axis=1, keepdims=True)
    Ph = 2 * mu * (S - 1) + lam * (J - 1) * J / S
    PF = np.einsum("nij,nj,nkj->nik", U, Ph * S, U) * volume
    return to_flat(PF)


def sand(F9, logjp, mu, lam, volume, cohesion, beta, yield_surface, volume_correction):
    """Drucker-Prager return mapping, constitutive_models.cuh:238-335: returns (F_new 9, PF 9, logJp)."""
    F = to_mats(F9)
    U, S, V = _svd_rot(F)
    lj = np.asarray(logjp, dtype=np.float64).copy()
    aS = np.maximum(np.abs(S), 1e-4)
    eps = np.log(aS) - cohesion
    sum_eps = eps.sum(axis=1)
    tr = sum_eps + lj
    eps_hat = eps - tr[:, None] / 3
    norm = np.sqrt((eps_hat ** 2).sum(axis=1))
    H = eps + cohesion
    tip = tr >= 0
    dg = norm + (3 * lam + 2 * mu) / (2 * mu) * tr * yield_surface
    proj = ~tip & (dg > 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        H[proj] = (eps - (dg / norm)[:, None] * eps_hat + cohesion)[proj]
    H[tip] = cohesion
    lj_new = np.where(tip, (beta * sum_eps + lj) if volume_correction else lj, 0.0)
    newS = np.exp(H)
    Fn = np.einsum("nij,nj,nkj->nik", U, newS, V)
    d = (2 * mu * H + lam * H.sum(axis=1, keepdims=True)) * volume
    PF = np.einsum("nij,nj,nkj->nik", U, d, U)
    return to_flat(Fn), to_flat(PF), lj_new
