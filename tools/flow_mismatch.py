"""How many second members of MISMATCHED slots (two singles of different keys in one lane slot: the second one takes the serial path) the pair layout has in the C3 flow,
under the chunking in force (512-record chunks) and under variants of it: a last chunk that absorbs a tail of <= 128 / <= 256 records.  From a checkpoint (keys of the records).
usage: flow_mismatch.py [steps=3000]"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from claymore_amd import scenes
from claymore_amd.engine import build_engine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sc = scenes.sand_column(9)
eng = build_engine(sc)
eng.initial_setup()
eng.run_fixed(steps, sc["dt"])
ppb = 64 * sc.get("config", {}).get("max_ppc", 128) if False else None
buf = eng.save_checkpoint()
hdr = np.frombuffer(buf[:48].tobytes(), np.int32)
pbc, nbc, ebc, prev_count = int(hdr[6]), int(hdr[7]), int(hdr[8]), int(hdr[9])
m0 = np.frombuffer(buf[64:64 + 48].tobytes(), np.int64)
nch = int(np.frombuffer(buf[64 + 4:64 + 8].tobytes(), np.int32)[0])
bincount_src, bucketed = int(m0[4]), int(m0[5])
o = [448]


def take(nbytes, dtype):
    a = np.frombuffer(buf[o[0]:o[0] + nbytes].tobytes(), dtype)
    o[0] += (nbytes + 15) & ~15
    return a


take(4 * 3 * ebc, np.int32), take(4 * 3 * prev_count, np.int32), take(4 * 256 * nbc, np.float32)
size = take(4 * (ebc + 1), np.int32)[:pbc].astype(np.int64)
take(4 * (ebc + 1), np.int32), take(4 * (prev_count + 1), np.int32), take(4 * (ebc + 1), np.int32)
take(4 * bincount_src * nch * 64, np.float32)
lists = take(4 * bucketed, np.int32).astype(np.int64) & 0xffffffff
pid_bits = int(np.log2(eng.ppb)) if hasattr(eng, "ppb") else 13
keys = (lists >> pid_bits) & 255
n = int(size.sum())
start = np.concatenate([[0], np.cumsum(size)])


def chunks(sz, merge):
    nfull, tail = sz // 512, sz % 512
    if nfull >= 1 and 0 < tail <= merge:
        return [512] * (nfull - 1) + [512 + tail]
    return [512] * nfull + ([tail] if tail else [])


out = {}
for merge in (0, 128, 256):
    x_tot = slices = 0
    for b in range(pbc):
        at = start[b]
        for nrec in chunks(int(size[b]), merge):
            c = np.bincount(keys[at:at + nrec], minlength=256)
            pf, n1 = int((c >> 1).sum()), int((c & 1).sum())
            S = (nrec + 127) // 128
            x_tot += max(0, pf + n1 - 64 * S)
            slices += S
            at += nrec
    out[merge] = (x_tot, slices)
    print(f"after {steps} substeps, last chunk absorbs a tail of <= {merge}: mismatched slots (second members on the serial path) {x_tot} = {100.0 * x_tot / n:.2f} % of the particles; slices {slices}")
