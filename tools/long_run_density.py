"""Long C3 run: every `chunk` substeps retrieve the positions and report the densest 4^3-cell block / cell (capacity planning for
max_ppc; the reference drops particles beyond 128 per cell, particle_buffer.cuh:122-130).  usage: long_run_density.py [steps] [chunk] [max_ppc]"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from claymore_amd import scenes
from claymore_amd.engine import build_engine, EngineError

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
sc = scenes.sand_column(9)
if len(sys.argv) > 3:
    sc["config"]["max_ppc"] = int(sys.argv[3])
eng = build_engine(sc)
eng.initial_setup()
done = 0
while done < steps:
    try:
        eng.run_fixed(chunk, sc["dt"])
    except EngineError as e:
        print(f"after step {done}: {e}")
        break
    done += chunk
    x = eng.retrieve_positions(0)
    cell = np.floor(x.astype(np.float64) * 512 + 0.5).astype(np.int64) - 1          # stencil base node of every particle
    cid = (cell[:, 0] * 512 + cell[:, 1]) * 512 + cell[:, 2]
    cc = np.bincount(np.unique(cid, return_inverse=True)[1])
    blk = cell >> 2
    bid = (blk[:, 0] * 128 + blk[:, 1]) * 128 + blk[:, 2]
    ub, inv = np.unique(bid, return_inverse=True)
    bc = np.bincount(inv)
    k = int(np.argmax(bc))
    bx, by, bz = (ub[k] // (128 * 128), (ub[k] // 128) % 128, ub[k] % 128)
    c = eng.counts()
    t = eng.timers()
    print(f"step {done}: blocks {c.particle_blocks}, densest block {bc.max()} particles at block ({bx},{by},{bz}), densest cell {cc.max()}, "
          f"y range [{x[:,1].min()*512:.1f}, {x[:,1].max()*512:.1f}] cells, x range [{x[:,0].min()*512:.1f}, {x[:,0].max()*512:.1f}], g2p2g {t.g2p2g_ms:.3f} ms", flush=True)
