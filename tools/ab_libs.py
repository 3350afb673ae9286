#!/usr/bin/env python
"""Same-box A/B of prebuilt engine libraries in ONE process: scenes are built once, the flow state of C3 (after --flow-start
substeps of the FIRST library) is checkpointed once and loaded into every variant, so a variant costs a few seconds of GPU time.

    python tools/ab_libs.py [--flow-start 3000] [--scenes c3,c3flow,c2,c5] lib_a.so lib_b.so ...   (paths or names in gpurun_libs/)

Prints G2P2G ms (HIP events inside the library) and whole-substep ms per window; writes gpurun_out/ab_libs.txt.
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from claymore_amd import _ffi, scenes  # noqa: E402
from claymore_amd.engine import build_engine  # noqa: E402


def load(path):
    """(a library of an earlier round may lack entry points that were added since: those stay unbound)"""
    lib = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    names = {n: sig for n, sig in {**_ffi.SIGNATURES, **_ffi.HIP_ONLY}.items() if hasattr(lib, "mpm_" + n)}
    return _ffi.Api(lib, "mpm_", names)


def window(eng, warm, steps, dt):
    eng.run_fixed(warm, dt)
    t0 = time.perf_counter()
    eng.run_fixed(steps, dt)
    el = time.perf_counter() - t0
    tm = eng.timers()
    return tm.g2p2g_ms, tm.partition_ms, 1e3 * el / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--flow-start", type=int, default=3000)
    ap.add_argument("--scenes", default="c3,c3flow,c2,c5")
    ap.add_argument("--reps", type=int, default=1)
    args = ap.parse_args()
    want = args.scenes.split(",")
    libs = [p if os.path.exists(p) else os.path.join(ROOT, "gpurun_libs", p) for p in args.libs]
    apis = [(os.path.basename(p), load(p)) for p in libs]
    out = open(os.path.join(ROOT, "gpurun_out", "ab_libs.txt"), "a")

    def emit(s):
        print(s, flush=True)
        out.write(s + "\n")
        out.flush()

    emit("# tools/ab_libs.py " + " ".join(sys.argv[1:]))
    sc3 = scenes.sand_column(9) if ("c3" in want or "c3flow" in want) else None
    ckpt = None
    if "c3flow" in want:
        eng = build_engine(sc3, api=apis[0][1])
        eng.initial_setup()
        eng.run_fixed(args.flow_start, sc3["dt"])
        ckpt = eng.save_checkpoint().copy()
        eng.close()
        emit(f"# flow checkpoint after {args.flow_start} substeps of {apis[0][0]}: {ckpt.nbytes / 1e9:.2f} GB")
    sc2 = scenes.sphere_drop() if "c2" in want else None
    sc5 = scenes.fluid_dam(10, (32, 192, 256)) if "c5" in want else None
    for rep in range(args.reps):
        for name, api in apis:
            row = [f"[{name}]"]
            if "c3" in want:
                eng = build_engine(sc3, api=api)
                eng.initial_setup()
                g, p, w = window(eng, 5, 20, sc3["dt"])
                row.append(f"C3 rest 5+20: g2p2g {g:.4f} part {p:.4f} step {w:.4f}")
                g, p, w = window(eng, 0, 85, sc3["dt"])
                row.append(f"next 85: g2p2g {g:.4f} step {w:.4f}")
                eng.close()
            if "c3flow" in want:
                eng = build_engine(sc3, api=api)
                eng.initial_setup()
                try:
                    eng.load_checkpoint(ckpt)
                except Exception:      # another checkpoint format: this library walks into the flow by itself
                    eng.run_fixed(args.flow_start, sc3["dt"])
                g, p, w = window(eng, 5, 20, sc3["dt"])
                row.append(f"| C3 flow: g2p2g {g:.4f} part {p:.4f} step {w:.4f}")
                eng.close()
            if "c2" in want:
                eng = build_engine(sc2, api=api)
                eng.initial_setup()
                g, p, w = window(eng, 10, 50, sc2["dt"])
                row.append(f"| C2 FC: g2p2g {g:.4f} step {w:.4f}")
                eng.close()
            if "c5" in want:
                eng = build_engine(sc5, api=api)
                eng.initial_setup()
                g, p, w = window(eng, 10, 50, sc5["dt"])
                row.append(f"| C5 fluid: g2p2g {g:.4f} step {w:.4f}")
                eng.close()
            if "c4" in want:       # one of C4's two spheres, translating at 1 m/s: every ~10th substep half of the particles change their stencil base at once
                sc4 = scenes.two_spheres_c4()
                sc4["models"] = sc4["models"][:1]
                eng = build_engine(sc4, api=api)
                eng.initial_setup()
                g, p, w = window(eng, 10, 40, sc4["dt"])
                row.append(f"| C4 one sphere: g2p2g {g:.4f} step {w:.4f}")
                eng.close()
            if "c2nacc" in want:   # C2's sphere as NACC (no BASELINE config uses NACC: a scene for its kernel instantiation alone)
                from claymore_amd import _ffi as _f
                scn = scenes.sphere_drop(material=_f.NACC)
                scn["models"][0]["params"] = {}
                eng = build_engine(scn, api=api)
                eng.initial_setup()
                g, p, w = window(eng, 10, 50, scn["dt"])
                row.append(f"| C2 as NACC at rest: g2p2g {g:.4f} step {w:.4f}")
                g, p, w = window(eng, 400, 50, scn["dt"])     # (in free fall; the default NACC parameters do not survive the impact at step ~2500 in either kernel)
                row.append(f"after 500: g2p2g {g:.4f} step {w:.4f}")
                eng.close()
            if "c5flow" in want:   # the dam break under way
                sc5f = scenes.fluid_dam(10, (32, 192, 256))
                eng = build_engine(sc5f, api=api)
                eng.initial_setup()
                g, p, w = window(eng, 3000, 30, sc5f["dt"])
                row.append(f"| C5 fluid after 3000: g2p2g {g:.4f} step {w:.4f}")
                eng.close()
            emit(" ".join(row))


if __name__ == "__main__":
    main()
