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


def nacc(F9, logjp, mu, lam, volume, beta, xi, msqr, hardening_on):
    """Non-associated Cam-Clay, constitutive_models.cuh:77-234 (USE_JOSH_FRACTURE_PAPER branch): returns (F_new 9, PF 9, logJp, case)
    with case 0 = inside the yield surface, 1 / 2 = projected to the max / min tip (:113-143), 3 = projected to the surface (:151-203).
    The model is discontinuous in its case selection; callers compare only rows whose case is stable under a small perturbation."""
    F = to_mats(F9)
    U, S, V = _svd_rot(F)
    lj = np.asarray(logjp, dtype=np.float64).copy()
    bm = 2.0 / 3.0 * mu + lam                                     # particle_buffer.cuh:255
    p0 = bm * (1e-5 + np.sinh(xi * np.maximum(-lj, 0.0)))
    p_min = -beta * p0
    Je = S.prod(axis=1)
    B = S * S
    trB3 = B.sum(axis=1) / 3.0
    with np.errstate(invalid="ignore", divide="ignore"):
        Jm = mu * np.power(Je, -2.0 / 3.0)
        s_hat = Jm[:, None] * (B - trB3[:, None])
        p_trial = -bm * 0.5 * (Je - 1.0 / Je) * Je
        a = 1.5 * (1.0 + 2.0 * beta)
        y_p_half = msqr * (p_trial - p_min) * (p_trial - p0)
        sq = (s_hat ** 2).sum(axis=1)
        y = a * sq + y_p_half
        case = np.where(p_trial > p0, 1, np.where(p_trial < p_min, 2, np.where(y >= 1e-4, 3, 0)))
        Sn = S.copy()
        lj_new = lj.copy()
        for c, ptip in ((1, p0), (2, p_min)):
            m = case == c
            Je_new = np.sqrt(-2.0 * ptip / bm + 1.0)
            Sn[m] = np.power(Je_new[m], 1.0 / 3.0)[:, None]
            if hardening_on:
                lj_new[m] += np.log(Je[m] / Je_new[m])
        m3 = case == 3
        coeff = np.power(Je, 2.0 / 3.0) / mu * np.sqrt(-y_p_half / a) / np.sqrt(sq)
        Sn[m3] = np.sqrt(s_hat * coeff[:, None] + trB3[:, None])[m3]
        if hardening_on:
            h = m3 & (p0 > 1e-4) & (p_trial < p0 - 1e-4) & (p_trial > 1e-4 + p_min)
            pc = (1.0 - beta) * p0 / 2.0
            q = np.sqrt(1.5 * sq)
            d0, d1 = pc - p_trial, -q
            dn = np.sqrt(d0 * d0 + d1 * d1)
            d0, d1 = d0 / dn, d1 / dn
            Cc = msqr * (pc - p_min) * (pc - p0)
            Bc = msqr * d0 * (2 * pc - p0 - p_min)
            Ac = msqr * d0 * d0 + (1 + 2 * beta) * d1 * d1
            disc = np.sqrt(Bc * Bc - 4 * Ac * Cc)
            p1 = pc + (-Bc + disc) / (2 * Ac) * d0
            p2 = pc + (-Bc - disc) / (2 * Ac) * d0
            pf = np.where((p_trial - pc) * (p1 - pc) > 0, p1, p2)
            Jf = np.sqrt(np.abs(-2 * pf / bm + 1.0))
            add = h & (Jf > 1e-4)
            lj_new[add] += np.log(Je[add] / Jf[add])
        Fn = np.where((case > 0)[:, None, None], np.einsum("nij,nj,nkj->nik", U, Sn, V), F)
        J = Sn.prod(axis=1)
        b = np.einsum("nij,nkj->nik", Fn, Fn)
        dev = b - np.trace(b, axis1=1, axis2=2)[:, None, None] / 3.0 * np.eye(3)
        PF = (mu * np.power(J, -2.0 / 3.0))[:, None, None] * dev + (bm * 0.5 * ((J * J - 1.0) * 0.5 - np.log(J)))[:, None, None] * np.eye(3)
    return to_flat(Fn), to_flat(PF * volume), lj_new, case
