"""What the hand-placed LDS waits of the G2P2G kernels rely on, checked in the ISA the build flags produce (CPU only: hipcc cross-compiles the device code to
assembly in ~10 s).

gather_apic / ScatterChain2 (claymore_amd/csrc/mpm_g2p2g.hpp, mpm_g2p2g_pair.hpp) request LDS reads with inline `ds_read_b128` and wait for them with
`s_waitcnt lgkmcnt(N)`, N > 0: "all but the newest N operations of the counter are done".  That holds because LDS operations complete in order; the counter is
shared with the scalar memory loads, which do NOT - so no scalar load may be outstanding when such a read is awaited.  The source has none inside the particle loop;
this test makes sure the compiler put none there either (it could: a kernel argument re-loaded under scalar register pressure), in every instantiation of both kernels.
It also pins the claim of DESIGN.md 3.2b-2: no scratch reload inside a particle loop, beyond the ones listed there."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

KERNELS = [f"_ZN3mpm17g2p2g_pair_kernelILi{m}E" for m in range(4)] + [f"_ZN3mpm12g2p2g_kernelILi{m}E" for m in range(4)]


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("isa") / "claymore_hip.s")
    flags = [f for f in entry.HIP_FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [entry.HIPCC] + flags + ["--offload-device-only", "-S", "-o", out, os.path.join(entry.CSRC, "claymore_hip.hip")]
    r = subprocess.run(cmd, cwd=entry.CSRC, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read().splitlines()


def kernel_body(lines, sym):
    start = next(i for i, l in enumerate(lines) if l.startswith(sym) and ":" in l.split(";")[0])
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].strip().startswith(".Lfunc_end"))
    return lines[start:end]


def instr(line):
    t = line.strip()
    return "" if (not t or t.startswith(";") or t.startswith(".")) else t


SMEM = re.compile(r"^s_(load|buffer_load|scratch_load)_")


@pytest.mark.parametrize("sym", KERNELS)
def test_no_scalar_load_is_outstanding_at_a_hand_placed_lds_wait(device_asm, sym):
    body = kernel_body(device_asm, sym)
    # walk the text twice (the second pass stands for "the next trip of whatever loop encloses this"): a scalar load is `pending` until an s_waitcnt lgkmcnt(0)
    pending_since = None
    hand_reads = 0
    in_asm = False
    for rep in range(2):
        for i, line in enumerate(body):
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            t = instr(line)
            if not t:
                continue
            if SMEM.match(t):
                pending_since = i
            elif t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                pending_since = None
            elif in_asm and (t.startswith("ds_read_b128") or (t.startswith("s_waitcnt") and "lgkmcnt(" in t)):
                hand_reads += 1
                assert pending_since is None, f"{sym}: scalar load of line {pending_since} may be outstanding at the hand-placed LDS operation of line {i}: {t}"
    assert hand_reads > 0, f"{sym}: no hand-placed LDS operation found - the check looks at the wrong kernel text"


def particle_loop(body):
    """[first, last] line of the innermost loop that contains the hand-issued gather reads: from the loop header comment in front of the first one to the last
    branch back to it"""
    first_read = next(i for i, l in enumerate(body) if "ds_read_b128" in l and "#ASMSTART" in body[i - 1])
    header = max(i for i in range(first_read) if "Loop Header: Depth=2" in body[i])
    label = next(body[i].split(":")[0].strip() for i in range(header, header - 3, -1) if body[i].startswith(".LBB"))
    last = max(i for i, l in enumerate(body) if re.search(r"s_c?branch\S*\s+" + re.escape(label) + r"\b", l))
    return header, last


@pytest.mark.parametrize("sym,allowed", [(KERNELS[0], 0), (KERNELS[1], 0), (KERNELS[2], 0), (KERNELS[3], 0)])
def test_no_scratch_access_inside_the_particle_loop_of_the_pair_kernels(device_asm, sym, allowed):
    body = kernel_body(device_asm, sym)
    a, b = particle_loop(body)
    assert b > a + 500, (a, b)  # (the loop is some thousand instructions long)
    inside = [instr(l) for l in body[a:b + 1] if instr(l).startswith("scratch_")]
    assert len(inside) <= allowed, f"{sym}: scratch accesses inside the particle loop: {inside[:6]}"
