"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header declares,
its host-only entry points agree with the oracle, and it refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

from claymore_amd import _ffi
from oracle_ffi import oracle_api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "claymore_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mpm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_ffi.HIP_LIB_PATH)
    names = header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/claymore_amd.h but not exported"


def test_ffi_table_matches_header():
    declared = set(header_symbols())
    bound = {"mpm_" + n for n in list(_ffi.SIGNATURES) + list(_ffi.HIP_ONLY)}
    assert bound == declared, (sorted(bound - declared), sorted(declared - bound))


def test_oracle_exports_same_surface():
    api = oracle_api()
    for n in _ffi.SIGNATURES:
        assert hasattr(api.raw, "mpmo_" + n)


@pytest.mark.parametrize("bits", [7, 8, 9, 10])
def test_defaults_agree_with_oracle(bits):
    hip = _ffi.load_hip()
    ora = oracle_api()
    a, b = _ffi.Config(), _ffi.Config()
    assert hip.default_config(bits, C.byref(a)) == 0 and ora.default_config(bits, C.byref(b)) == 0
    assert bytes(a) == bytes(b)
    assert (a.max_ppc, a.boundary_blocks, a.cfl) == (128, 2, 0.5) and abs(a.gravity + 9.8) < 1e-6
    for mat in range(4):
        p, q = _ffi.MaterialParams(), _ffi.MaterialParams()
        assert hip.default_material(mat, bits, C.byref(p)) == 0 and ora.default_material(mat, bits, C.byref(q)) == 0
        assert bytes(p) == bytes(q)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback_without_gpu():
    hip = _ffi.load_hip()
    cfg = _ffi.Config()
    hip.default_config(7, C.byref(cfg))
    ctx = C.c_void_p()
    assert hip.create(C.byref(cfg), 0, C.byref(ctx)) == _ffi.MPM_ERR_DEVICE
    from claymore_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(domain_bits=7)


def test_shipped_library_has_no_experiment_switch():
    """Every switch that changes what the kernels compute (timing hacks that leave physics out, A/B variants) lives behind
    -DMPM_EXPERIMENT; the library build() ships reports none, and build()'s flags contain no MPM_ define at all."""
    hip = _ffi.load_hip()
    info = hip.build_info().decode()
    assert info.startswith("claymore_hip ") and info.endswith("experiment=none"), info
    import __graft_entry__ as ge
    assert not [f for f in ge.HIP_FLAGS if "MPM_" in f]


def test_state_kind_says_what_retrieve_state_returns():
    """mpm_retrieve_state kept its symbol when the solid models' state became b = F F^T (ABI 5): the library says so through
    mpm_state_kind() and the ABI number in mpm_build_info(); the CPU oracle still carries the reference's F."""
    from oracle_ffi import oracle_api
    hip = _ffi.load_hip()
    assert hip.state_kind() == 1 and " abi7 " in hip.build_info().decode() and "state=b" in hip.build_info().decode()
    assert oracle_api().state_kind() == 0


def test_experiment_switch_without_the_guard_does_not_compile():
    """A stray -DMPM_HACK_* (or any other experiment switch) without -DMPM_EXPERIMENT is a compile error (preprocessor only: fast)."""
    import subprocess
    import __graft_entry__ as ge
    src = os.path.join(ROOT, "claymore_amd", "csrc", "mpm_device_math.hpp")
    base = [ge.HIPCC, "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-x", "hip", "-E", "-o", os.devnull, src]
    assert subprocess.run(base, capture_output=True).returncode == 0
    for sw in ("MPM_HACK_NOSERIAL", "MPM_HACK_NOSHELL", "MPM_HACK_NOWB", "MPM_HACK_UNDEF", "MPM_G2P2G_WAVES=2", "MPM_HACK_NOSPLIT", "MPM_PAIR_WAVES=2", "MPM_PAIR_DUAL=1"):
        r = subprocess.run(base + ["-D" + sw], capture_output=True, text=True)
        assert r.returncode != 0 and "MPM_EXPERIMENT" in r.stderr, (sw, r.stderr[-300:])
        assert subprocess.run(base + ["-D" + sw, "-DMPM_EXPERIMENT"], capture_output=True).returncode == 0


def test_bench_without_a_gpu_fails_loudly_with_one_json_error_line():
    """The product path has no CPU fallback: on a box without a GPU bench.py exits non-zero and its stdout is ONE JSON record with an
    `error` field (what a driver that parses the line sees instead of a traceback), never a number computed some other way."""
    import json
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["value"] is None and "needs a GPU" in rec["error"] and rec["n_gpus"] == 1


def test_bench_attaches_counters_only_of_the_library_it_loaded():
    """bench.py's `traffic` / `physical_frac` / `valu_executed` come from profiles/rNN_pmc.json, a separate rocprofv3 run: they are attached only when
    that file's stamp carries the sha256 of the library this run loaded (a kernel edit without a re-profile must not report
    stale traffic), and the newest committed counter file is stamped at all."""
    import glob
    import json
    import re
    import sys
    sys.path.insert(0, ROOT)
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))
    assert files
    pmc = json.load(open(files[-1]))
    m = re.search(r"sha256 ([0-9a-f]{16})", pmc.get("stamp", ""))
    assert m, "the newest counter file carries no library hash"
    n, bpp = 40108032, 144
    ok, stale = {}, {}
    bench.attach_counters(ok, pmc, "profiles/x.json", "flow", m.group(1), 1.9, n, bpp)
    assert ok["traffic"] == pmc["flow"]["traffic_bytes"] and ok["traffic_range"][0] <= ok["traffic"]
    assert ok["physical_frac"] == pytest.approx(pmc["flow"]["traffic_bytes"] / 1.9e-3 / 8e12) and 0.2 < ok["physical_frac"] < 1.0
    assert ok["valu_executed"]["executed_per_particle"] == pytest.approx(pmc["flow"]["valu_insts"] * 64 / n)
    bench.attach_counters(stale, pmc, "profiles/x.json", "flow", "0123456789abcdef", 1.9, n, bpp)
    assert stale["traffic"] is None and stale["traffic_source"].startswith("REFUSED") and "physical_frac" not in stale and "valu_executed" not in stale
    assert stale["algorithmic_bytes"] == n * bpp
    assert len(bench.loaded_library_sha16()) == 16
