"""Per-substep G2P2G time of a scene (one substep per call: the engine's event pair around the launches of all models).
usage: launch_series.py [scene=c4|c4rest|c4one] [steps=60]"""
import sys

sys.path.insert(0, "/root/repo")
from claymore_amd import scenes
from claymore_amd.engine import build_engine

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
if which == "c4":
    sc = scenes.two_spheres_c4()
elif which == "c4rest":
    sc = scenes.two_spheres_c4(speed=0.0)
elif which == "c4one":
    sc = scenes.two_spheres_c4()
    sc["models"] = sc["models"][:1]
else:
    raise SystemExit(which)
dt = sc["dt"]
eng = build_engine(sc)
eng.initial_setup()
ms = []
for i in range(steps):
    eng.run_fixed(1, dt)
    ms.append(eng.last_g2p2g_ms())
c = eng.counts()
print(f"# {which}: {scenes.total_particles(sc)} particles, {c.particle_blocks} particle blocks after {steps} substeps; g2p2g ms per substep:")
for i in range(0, steps, 10):
    print("  " + " ".join(f"{v:6.3f}" for v in ms[i:i + 10]))
s = sorted(ms[5:])
print(f"  min {s[0]:.3f} median {s[len(s) // 2]:.3f} mean {sum(s) / len(s):.3f} max {s[-1]:.3f}")
eng.close()
