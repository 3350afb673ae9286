"""Helpers shared by the parity tests, smoke() and bench.py's cpu_baseline leg: run the same scene through an
engine (HIP product or CPU oracle) and compare particle sets.  Particle order is unspecified on both sides
(atomics), so particles are matched by nearest neighbour, which is unambiguous while the error (<=1e-5) is
far below the particle spacing (0.5 dx >= 5e-4)."""
import numpy as np
from scipy.spatial import cKDTree

from claymore_amd.engine import build_engine
from oracle_ffi import oracle_api


def run_engine(scene, nsteps, dt=None, api=None, adaptive=False, dt_default=None, collect_grid=False):
    eng = build_engine(scene, api=api)
    dt = scene["dt"] if dt is None else dt
    eng.initial_setup()
    out = {"counts0": eng.counts(), "totals0": eng.grid_totals()}
    if adaptive:
        t, cur = 0.0, dt
        frame = 1.0 / 24.0
        dts = []
        for _ in range(nsteps):
            nd, mv = eng.substep(cur, t, frame, dt_default or dt)
            t += cur
            dts.append(cur)
            cur = nd
        out["dts"] = dts
    else:
        eng.run_fixed(nsteps, dt)
    out["counts"] = eng.counts()
    out["totals"] = eng.grid_totals()
    out["state"] = [eng.retrieve_state(m) for m in range(len(scene["models"]))]
    if collect_grid:
        out["grid"] = eng.dump_grid()
    eng.close()
    return out


def run_pair(scene, nsteps, dt=None, **kw):
    hip = run_engine(scene, nsteps, dt, api=None, **kw)
    ora = run_engine(scene, nsteps, dt, api=oracle_api(), **kw)
    return {"hip": hip, "oracle": ora, "scene": scene}


def to_b(state9, from_hip):
    """The comparable per-particle state: the left Cauchy-Green tensor b = F F^T (n, 9).  The HIP engine carries b itself
    (claymore_amd/csrc/mpm_device_math.hpp; the sign of its first entry marks a reflected F), the oracle carries the
    reference's F; the J-fluid's state is J in column 0 on both sides."""
    s = np.asarray(state9, dtype=np.float64)
    if s.size == 0 or not np.any(s[:, 1:]):
        return s
    if from_hip:
        s = s.copy()
        s[:, 0] = np.abs(s[:, 0])
        return s
    F = s.reshape(-1, 3, 3).transpose(0, 2, 1)          # column-major 9 -> F[i, j]
    return np.einsum("nij,nkj->nik", F, F).reshape(-1, 9)


def match(xa, xb):
    """Index array idx with xb[idx[i]] the match of xa[i]; asserts a bijection."""
    tree = cKDTree(xb)
    d, idx = tree.query(xa, k=1)
    assert np.unique(idx).size == xa.shape[0], "particle matching is not one-to-one"
    return idx, d


def match_and_compare(res):
    worst = {"pos_rel": 0.0, "pos_abs": 0.0, "state_rel": 0.0, "logjp_abs": 0.0, "n": 0}
    for (xh, sh, lh), (xo, so, lo) in zip(res["hip"]["state"], res["oracle"]["state"]):
        assert xh.shape == xo.shape, (xh.shape, xo.shape)
        idx, d = match(xo.astype(np.float64), xh.astype(np.float64))
        dx = np.abs(xh[idx].astype(np.float64) - xo.astype(np.float64)).max(axis=1)
        rel = dx / np.abs(xo).max(axis=1)
        worst["pos_rel"] = max(worst["pos_rel"], float(rel.max()))
        worst["pos_abs"] = max(worst["pos_abs"], float(dx.max()))
        bh, bo = to_b(sh, True), to_b(so, False)
        dF = np.abs(bh[idx] - bo).max(axis=1)
        worst["state_rel"] = max(worst["state_rel"], float((dF / np.maximum(1.0, np.abs(bo).max(axis=1))).max()))
        worst["logjp_abs"] = max(worst["logjp_abs"], float(np.abs(lh[idx] - lo).max()))
        worst["n"] += xh.shape[0]
    th, to = res["hip"]["totals"], res["oracle"]["totals"]
    worst["grid_mass_rel"] = float(abs(th[0] - to[0]) / max(abs(to[0]), 1e-30))
    scale = max(np.abs(to[1:]).max(), abs(to[0]) * 1e-3, 1e-30)
    worst["grid_mom_rel"] = float(np.abs(th[1:] - to[1:]).max() / scale)
    return worst


def grid_compare(res):
    """Node-by-node comparison of the two grids (keys matched)."""
    kh, bh = res["hip"]["grid"]
    ko, bo = res["oracle"]["grid"]
    mh = {tuple(k): i for i, k in enumerate(kh)}
    worst = 0.0
    scale = np.abs(bo).max(axis=(0, 2))  # per channel; momentum channels share one scale (a channel with no
    scale[1:] = scale[1:].max()          # net motion holds only stress-rounding noise)
    for j, k in enumerate(ko):
        i = mh.get(tuple(k))
        if i is None:
            assert np.all(bo[j] == 0), "oracle has a non-empty grid block the HIP grid lacks"
            continue
        worst = max(worst, float((np.abs(bh[i] - bo[j]) / scale[:, None]).max()))
    return worst


def grid_velocity_compare(res, mass_floor=1e-3):
    """Node-by-node comparison of the grid VELOCITIES (momentum / mass per node, keys matched; north_star: "positions/velocities ... within
    1e-5"; the reference's particles carry no velocity, it lives on the grid: mgmpm_kernels.cuh:325-420).  Nodes lighter than mass_floor of
    the heaviest node are skipped (a quotient of two rounding residues); the result is relative to the largest |v| on the oracle's grid."""
    kh, bh = res["hip"]["grid"]
    ko, bo = res["oracle"]["grid"]
    mh = {tuple(k): i for i, k in enumerate(kh)}
    mmax = float(np.abs(bo[:, 0]).max())
    heavy_o = bo[:, 0] > mass_floor * mmax                                       # (blocks, 64 cells)
    vo_all = bo[:, 1:].astype(np.float64) / np.where(heavy_o, bo[:, 0], 1.0).astype(np.float64)[:, None, :]
    vmax = float(np.abs(np.where(heavy_o[:, None, :], vo_all, 0.0)).max())
    worst = 0.0
    for j, k in enumerate(ko):
        i = mh.get(tuple(k))
        if i is None:
            assert np.all(bo[j] == 0), "oracle has a non-empty grid block the HIP grid lacks"
            continue
        heavy = heavy_o[j]
        if not heavy.any():
            continue
        vo = bo[j][1:][:, heavy].astype(np.float64) / bo[j][0][heavy].astype(np.float64)
        vh = bh[i][1:][:, heavy].astype(np.float64) / bh[i][0][heavy].astype(np.float64)
        worst = max(worst, float(np.abs(vh - vo).max()))
    return worst / max(vmax, 1e-30)
