"""The block-size distribution of the C3 column after N substeps and what the pair layout's 512-record chunks make of it: slices (= iterations of the pair kernel)
against ceil(n / 128) per block and against N / 128.   usage: flow_blocksizes.py [steps=3000]"""
import sys
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
import numpy as np
from claymore_amd import scenes
from claymore_amd.engine import build_engine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sc = scenes.sand_column(9)
eng = build_engine(sc)
eng.initial_setup()
eng.run_fixed(steps, sc["dt"])
buf = eng.save_checkpoint()
hdr = np.frombuffer(buf[:48].tobytes(), np.int32)
pbc, nbc, ebc, prev_count = int(hdr[6]), int(hdr[7]), int(hdr[8]), int(hdr[9])
o = 448
for nbytes in (4 * 3 * ebc, 4 * 3 * prev_count, 4 * 256 * nbc):
    o += (nbytes + 15) & ~15
size = np.frombuffer(buf[o:o + 4 * (ebc + 1)].tobytes(), np.int32)[:pbc].astype(np.int64)
n = int(size.sum())
full, tail = size // 512, size % 512
slices = full * 4 + (tail + 127) // 128
print(f"# C3 after {steps} substeps: {pbc} particle blocks, {n} particles, {n / pbc:.0f} per block")
print(f"slices (pair-kernel iterations) {int(slices.sum())}; sum ceil(n / 128) per block {int(((size + 127) // 128).sum())}; N / 128 = {n / 128:.0f}")
h = np.bincount(np.minimum(size // 64, 20))
print("blocks by size / 64:", " ".join(f"{64 * i}:{c}" for i, c in enumerate(h) if c))
over = (size > 512)
print(f"blocks with more than 512 particles: {int(over.sum())} ({100.0 * over.mean():.1f} %), of them with a last chunk of <= 64 records: {int(((tail > 0) & (tail <= 64) & over).sum())}; mean fill of the slices {n / (128.0 * slices.sum()):.3f}")
