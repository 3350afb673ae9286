#!/usr/bin/env python
"""Rebuild libclaymore_hip.so with -Rpass-analysis and print one resource line per g2p2g instantiation."""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(os.path.join(root, "claymore_amd", "csrc"))
r = subprocess.run("hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-unused-value -fno-slp-vectorize "
                   "-Rpass-analysis=kernel-resource-usage -o libclaymore_hip.so claymore_hip.hip", shell=True, capture_output=True, text=True)
lines = r.stderr.split("\n")
pat = sys.argv[1] if len(sys.argv) > 1 else "g2p2g"
for i, l in enumerate(lines):
    if "error" in l:
        print("\n".join(lines[i:i + 6]))
    if "Function Name" in l and pat in l:
        keys = ["Name", "VGPRs:", "AGPRs", "SGPRs:", "Occupancy", "Spill", "LDS", "Scratch"]
        print(" | ".join(x.split("remark: ")[-1].split("[-R")[0].strip().replace("Function Name: ", "")[:60] for x in lines[i:i + 12] if any(k in x for k in keys)))
sys.exit(r.returncode)
