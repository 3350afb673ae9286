#!/usr/bin/env python
"""Measured bounds for the grid (= velocity) parity tests: node-by-node deviation of the HIP grid from the oracle's after N substeps
(tests/parity_util.grid_compare: per channel group, relative to the largest entry) and the grid-momentum total.  GPU box only.
    python tools/probe_grid_parity.py            -> gpurun_out/grid_parity_probe.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from claymore_amd import scenes, _ffi  # noqa: E402
from parity_util import run_pair, match_and_compare, grid_compare, grid_velocity_compare  # noqa: E402

out = open(os.path.join(ROOT, "gpurun_out", "grid_parity_probe.txt"), "w")


def emit(s):
    print(s, flush=True)
    out.write(s + "\n")
    out.flush()


for name, sc in (("two_spheres(6)", scenes.two_spheres(bits=6, radius_cells=5.0, gap_cells=3.0, speed=0.5)),
                 ("C1 two_spheres(7)", scenes.two_spheres())):
    for n in (1, 10, 100):
        res = run_pair(sc, n, 1e-4, collect_grid=True)
        e = match_and_compare(res)
        emit(f"{name} nsteps {n}: grid_compare {grid_compare(res):.3e} velocity {grid_velocity_compare(res):.3e} grid_mom_rel {e['grid_mom_rel']:.3e} grid_mass_rel {e['grid_mass_rel']:.3e} pos_rel {e['pos_rel']:.3e}")
