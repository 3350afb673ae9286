import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from claymore_amd import scenes
from claymore_amd.engine import build_engine
sc = scenes.sand_column(9)
api = None
if len(sys.argv) > 1:   # an engine library built with -DMPM_G2P2G_STATS (tools/build_variant.sh stats -DMPM_G2P2G_STATS)
    import ctypes as C, os
    from claymore_amd import _ffi
    api = _ffi.bind(C.CDLL(os.path.abspath(sys.argv[1]), mode=os.RTLD_LOCAL | os.RTLD_NOW), "mpm_", hip=True)
eng = build_engine(sc, api=api); eng.initial_setup()
prev = np.zeros(5)
done = 0
for upto in (10, 1000, 3000, 3100):
    eng.run_fixed(upto - done, 1e-4); done = upto
    d = eng.diagnostics(); cur = np.array([d.reserved[i] for i in range(4)] + [0], dtype=np.float64)
    # counters are int32 cumulative: use per-window deltas modulo 2^32
    delta = (cur - prev) % 2**32; prev = cur
    eng.run_fixed(1, 1e-4); done += 1
    d2 = eng.diagnostics(); c2 = np.array([d2.reserved[i] for i in range(4)] + [0], dtype=np.float64)
    one = (c2 - cur) % 2**32; prev = c2
    c = eng.counts(); t = eng.timers()
    print(f"step {done}: blocks {c.particle_blocks} g2p2g {t.g2p2g_ms:.3f} ms | per substep: iterations {one[0]:.0f}, loser lanes {one[1]:.0f} ({100*one[1]/40108032:.2f} % of particles), edge lanes {one[2]:.0f} ({100*one[2]/40108032:.2f} %), iterations with a serial entry {one[3]:.0f} ({100*one[3]/one[0]:.1f} %)")
