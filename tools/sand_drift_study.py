#!/usr/bin/env python
"""Is the 2e-5 position difference between the HIP engine and the oracle after 500 substeps of plastic sand (bound 5e-5 in
tests/test_parity_gpu.py::test_long_run_sand_stays_close_to_the_oracle) an approximation error or rounding noise of the
dynamics itself?  Oracle vs ORACLE: the serial oracle against the very same C code run with OpenMP threads over particle
blocks - identical arithmetic per particle, only the float summation order of the P2G grid accumulation (and the cell bucket
order) differs; and oracle vs oracle with a CONVERGED double-precision SVD in place of the reference's approximate
four-sweep one (svd.cuh:167).  The first measures chaos under rounding noise, the second how far the reference's own SVD
residual moves the run: the HIP engine's stress functions are within 1e-5 of the exact closed forms
(tests/test_parity_gpu.py), i.e. they differ from the reference by the reference's own SVD error.

    python tools/sand_drift_study.py            (CPU only, ~1 min)   -> profiles/r02_sand_drift_study.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from claymore_amd import scenes  # noqa: E402
from claymore_amd.engine import build_engine  # noqa: E402
from oracle_ffi import oracle_api  # noqa: E402
from parity_util import match  # noqa: E402


def run(threads, checkpoints, exact_svd=False):
    api = oracle_api()
    api.raw.mpmo_set_exact_svd(1 if exact_svd else 0)
    sc = scenes.scaled_sand_column(7, 1.0 / 64)
    eng = build_engine(sc, api=api)
    eng.initial_setup()
    api.raw.mpmo_set_threads(eng.ctx, threads)
    out, done = {}, 0
    for n in checkpoints:
        eng.run_fixed(n - done, sc["dt"])
        done = n
        out[n] = eng.retrieve_positions(0).copy()
    eng.close()
    return out


def main():
    cps = [100, 200, 300, 400, 500]
    a = run(1, cps)
    lines = ["# oracle (serial) vs oracle (OpenMP over particle blocks): max relative position difference, 9.7 k-particle sand column",
             "# threads  " + "  ".join(f"step {n:4d}" for n in cps)]
    for t in (2, 4, 8):
        b = run(t, cps)
        row = []
        for n in cps:
            idx, _ = match(a[n].astype(np.float64), b[n].astype(np.float64))
            rel = np.abs(b[n][idx].astype(np.float64) - a[n]).max(axis=1) / np.abs(a[n]).max(axis=1)
            row.append(rel.max())
        lines.append(f"{t:8d}  " + "  ".join(f"{v:9.2e}" for v in row))
    lines.append("# oracle (reference's approximate 4-sweep SVD) vs oracle (converged double-precision SVD, everything else identical)")
    b = run(1, cps, exact_svd=True)
    row = []
    for n in cps:
        idx, _ = match(a[n].astype(np.float64), b[n].astype(np.float64))
        rel = np.abs(b[n][idx].astype(np.float64) - a[n]).max(axis=1) / np.abs(a[n]).max(axis=1)
        row.append(rel.max())
    lines.append("exact svd " + "  ".join(f"{v:9.2e}" for v in row))
    txt = "\n".join(lines)
    print(txt)
    open(os.path.join(ROOT, "profiles", "r02_sand_drift_study.txt"), "w").write(txt + "\n")


if __name__ == "__main__":
    main()
