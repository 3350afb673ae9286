#!/usr/bin/env python
"""How spatially coherent is the block numbering?  G2P2G deals the particle blocks to the 8 XCDs as contiguous eighths of the
block range; a particle block reads 8 grid blocks and adds into 8 grid blocks of the neighbour list.  For the C3 column at rest
and after --flow-start substeps: the fraction of (particle block, grid block) pairs whose grid block is ALSO touched by a
particle block of another eighth (i.e. is fetched into / updated in more than one L2), and the mean index distance of spatial
neighbours.

    python tools/block_coherence.py [--flow-start 3000]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from claymore_amd import scenes  # noqa: E402
from claymore_amd.engine import build_engine  # noqa: E402


def report(tag, eng):
    cnt = eng.counts()
    keys, _ = eng.dump_grid()
    pbc = cnt.particle_blocks
    k = keys.astype(np.int64)
    code = (k[:, 0] << 40) | (k[:, 1] << 20) | k[:, 2]
    order = np.argsort(code)
    sorted_code = code[order]
    pk = k[:pbc]
    share = pbc // 8
    xcd_of = np.minimum(np.arange(pbc) // max(share, 1), 7)
    touched = np.zeros((keys.shape[0], 8), dtype=bool)
    dist = []
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                c = ((pk[:, 0] + dx) << 40) | ((pk[:, 1] + dy) << 20) | (pk[:, 2] + dz)
                pos = np.searchsorted(sorted_code, c)
                pos = np.minimum(pos, sorted_code.size - 1)
                ok = sorted_code[pos] == c
                nb = order[pos[ok]]
                touched[nb, xcd_of[ok]] = True
                if dx + dy + dz == 1:
                    isp = nb < pbc
                    dist.append(np.abs(nb[isp] - np.arange(pbc)[ok][isp]))
    nx = touched.sum(axis=1)
    used = nx > 0
    d = np.concatenate(dist)
    print(f"{tag}: particle blocks {pbc}, grid blocks touched {used.sum()}, touched by 1 XCD {np.mean(nx[used] == 1):.3f}, by 2 {np.mean(nx[used] == 2):.3f}, by >=3 {np.mean(nx[used] >= 3):.3f}; "
          f"L2 copies per touched grid block {nx[used].mean():.3f}; face-neighbour index distance: median {np.median(d):.0f}, mean {d.mean():.0f}, share beyond one eighth {np.mean(d > share):.3f}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flow-start", type=int, default=3000)
    args = ap.parse_args()
    sc = scenes.sand_column(9)
    eng = build_engine(sc)
    eng.initial_setup()
    eng.run_fixed(10, sc["dt"])
    report("rest (10 substeps)", eng)
    eng.run_fixed(args.flow_start - 10, sc["dt"])
    report(f"flow ({args.flow_start} substeps)", eng)
    eng.close()


if __name__ == "__main__":
    main()
