#!/usr/bin/env python
"""HIP engine vs oracle on the long plastic-sand run (9.7 k particles, 500 substeps): max relative position difference at
checkpoints.  usage (gpurun): python tools/gpu_drift.py [--exact-svd]"""
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

api = oracle_api()
api.raw.mpmo_set_exact_svd(1 if "--exact-svd" in sys.argv else 0)
sc = scenes.scaled_sand_column(7, 1.0 / 64)
hip, ora = build_engine(sc), build_engine(sc, api=api)
hip.initial_setup()
ora.initial_setup()
done = 0
print("# HIP engine vs oracle%s: max relative position / max |F| difference" % (" (oracle with converged double-precision SVD)" if "--exact-svd" in sys.argv else ""))
for n in (50, 100, 200, 300, 400, 500):
    hip.run_fixed(n - done, sc["dt"])
    ora.run_fixed(n - done, sc["dt"])
    done = n
    xh, sh, lh = hip.retrieve_state(0)
    xo, so, lo = ora.retrieve_state(0)
    idx, _ = match(xo.astype(np.float64), xh.astype(np.float64))
    rel = np.abs(xh[idx].astype(np.float64) - xo).max(axis=1) / np.abs(xo).max(axis=1)
    print(f"step {n:4d}: pos {rel.max():.2e} (median {np.median(rel):.1e})  F {np.abs(sh[idx] - so).max():.2e}  logJp {np.abs(lh[idx] - lo).max():.2e}")
