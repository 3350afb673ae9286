import sys
sys.path.insert(0, "/root/repo")
from claymore_amd import scenes
from claymore_amd.engine import build_engine
sc = scenes.sand_column(9); dt = sc["dt"]
eng = build_engine(sc); eng.initial_setup()
eng.run_fixed(2990, dt)
out = []
for w in range(16):
    eng.run_fixed(10, dt); t = eng.timers(); out.append((3000 + 10 * w, t.g2p2g_ms, t.partition_ms))
print(" ".join(f"{s}:{g:.3f}" for s, g, p in out))
eng.run_fixed(8, dt); eng.run_fixed(20, dt); t = eng.timers(); print("20-step window after 3158:", round(t.g2p2g_ms, 4))
