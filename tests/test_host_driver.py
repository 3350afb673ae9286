"""Host-side pieces above the C ABI: scene JSON reader, lattice sampler, BGEO frame writer (CPU), and the `gmpm`
executable end to end on the GPU (same scene through the Python Engine must give the same frames)."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from claymore_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "claymore_amd", "host")


def read_bgeo(path):
    """Reader for the position-only classic BGEO v5 files partio writes (Externals/partio/io/BGEO.cpp:311-477)."""
    raw = open(path, "rb").read()
    magic, vchar, version, npoints = struct.unpack(">IcII", raw[:13])
    assert magic == 0x4267656F and vchar == b"V" and version == 5
    rest = struct.unpack(">7I", raw[13:41])
    assert rest == (0, 0, 0, 0, 0, 0, 0)          # nPrims, groups, attribute counts
    pts = np.frombuffer(raw[41:41 + 16 * npoints], dtype=">f4").reshape(npoints, 4)
    assert np.all(pts[:, 3] == 1.0)               # homogeneous coordinate
    assert raw[41 + 16 * npoints:] == b"\x00\xff"
    return pts[:, :3].astype(np.float32)


def test_host_selftest(tmp_path):
    exe = tmp_path / "host_selftest"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-o", str(exe), os.path.join(HOST, "host_selftest.cpp")])
    out = subprocess.check_output([str(exe), str(tmp_path / "t.bgeo")], text=True).splitlines()
    assert out[0] == 'json_ok 2 5e-06 s"q -1'
    assert out[1] == "json_bad rejected"
    n = int(out[2].split()[1])
    assert n == scenes.lattice_sphere(6, (0.5, 0.5, 0.5), 5.0).shape[0]     # same lattice rule as the Python sampler
    pts = read_bgeo(tmp_path / "t.bgeo")
    assert np.array_equal(pts, np.array([[0.25, 0.5, 0.75], [-1.5, 2.0, 3.25]], dtype=np.float32))
    # the asynchronous IO worker (reference IO.h:10-67): all queued frames are on disk after flush()
    assert out[4] == "async 8"
    assert np.array_equal(read_bgeo(str(tmp_path / "t.bgeo") + ".async7"), pts)


def test_bgeo_writer_equals_partio_byte_for_byte(tmp_path):
    """tests/golden/g10_partio.bgeo was written by the reference's own partio library (Externals/partio compiled from its
    sources by tests/golden/gen/gen_bgeo.sh) through the calls of write_partio (Library/MnSystem/IO/ParticleIO.hpp:14-29);
    the host drivers' writer must produce the same bytes from the same points."""
    exe = tmp_path / "host_selftest"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-o", str(exe), os.path.join(HOST, "host_selftest.cpp")])
    gold = os.path.join(ROOT, "tests", "golden")
    out = tmp_path / "mine.bgeo"
    subprocess.check_call([str(exe), "--bgeo-from", os.path.join(gold, "g10_points.f32"), str(out)])
    want = open(os.path.join(gold, "g10_partio.bgeo"), "rb").read()
    got = open(out, "rb").read()
    assert len(got) == len(want) == 41 + 16 * 1000 + 2
    assert got == want
    # and the reader used by the other tests understands partio's file
    pts = read_bgeo(os.path.join(gold, "g10_partio.bgeo"))
    assert np.array_equal(pts, np.fromfile(os.path.join(gold, "g10_points.f32"), dtype=np.float32).reshape(-1, 3))


@pytest.mark.gpu
def test_gmpm_executable_matches_engine(tmp_path):
    import __graft_entry__ as g
    g.build_host()
    scene = {
        "simulation": {"gpuid": 0, "fps": 500, "frames": 2, "default_dt": 1e-4, "domain_bits": 6, "output_dir": str(tmp_path)},
        "models": [
            {"file": "sphere", "constitutive": "fixed_corotated", "rho": 1e3, "volume": float(np.float32((1 / 64) ** 3 / 8)),
             "youngs_modulus": 5e3, "poisson_ratio": 0.4, "offset": [0.34375, 0.421875, 0.421875], "span": [0.15625] * 3, "velocity": [0.5, 0, 0]},
            {"file": "sphere", "constitutive": "sand", "offset": [0.53125, 0.421875, 0.421875], "span": [0.15625] * 3, "velocity": [-0.5, 0, 0]},
        ],
    }
    fn = tmp_path / "scene.json"
    fn.write_text(json.dumps(scene))
    out = subprocess.check_output([os.path.join(HOST, "gmpm"), "-f", str(fn)], text=True)
    assert "total number of particles" in out
    from claymore_amd import _ffi
    from claymore_amd.engine import Engine
    eng = Engine(domain_bits=6, max_ppc=32)
    p0 = read_bgeo(tmp_path / "model_id[0]_frame[0].bgeo")
    p1 = read_bgeo(tmp_path / "model_id[1]_frame[0].bgeo")
    eng.init_model(_ffi.FIXED_COROTATED, p0, (0.5, 0, 0), volume=scene["models"][0]["volume"], rho=1e3, youngs_modulus=5e3, poisson_ratio=0.4)
    eng.init_model(_ffi.SAND, p1, (-0.5, 0, 0))
    frames = {}
    eng.main_loop(2, 500, 1e-4, on_frame=lambda f, e: frames.update({f: [e.retrieve_positions(0), e.retrieve_positions(1)]}))
    from parity_util import match
    for f in (1, 2):
        for m in (0, 1):
            a = read_bgeo(tmp_path / f"model_id[{m}]_frame[{f}].bgeo")
            b = frames[f][m]
            assert a.shape == b.shape
            idx, _ = match(a.astype(np.float64), b.astype(np.float64))
            assert np.abs(a - b[idx]).max() < 1e-6
    eng.close()


def _sample_sdf_numpy(sdf_path, bits, offset, span):
    """Independent restatement of the level-set branch of sample_model (gmpm.cu:60-165, ParticleIO.hpp:32-70 as re-specified
    in SURVEY f2): lattice particles (8 per node at +-0.25 dx) where the trilinear interpolation of phi is negative; the level
    set box is scaled uniformly into `span` and moved to `offset`."""
    tok = open(sdf_path).read().split()
    n = [int(t) for t in tok[:3]]
    mn = np.array([float(t) for t in tok[3:6]], dtype=np.float32)
    sdx = np.float32(tok[6])
    phi = np.array(tok[7:], dtype=np.float32).reshape(n[2], n[1], n[0]).transpose(2, 1, 0)
    dx = np.float32(1.0 / (1 << bits))
    offset, span = np.array(offset, dtype=np.float32), np.array(span, dtype=np.float32)
    ext = np.array(n, dtype=np.float32) * sdx - mn
    scale = np.float32((span / ext).min())
    lo = np.floor(offset / dx).astype(int) - 1
    hi = np.ceil((offset + span) / dx).astype(int) + 2
    pts = scenes.lattice_box(bits, lo, hi)
    q = (pts - offset) / scale + mn
    g = (q - mn) / sdx
    c = np.floor(g).astype(int)
    t = g - c
    ok = np.all((c >= 0) & (c + 1 < np.array(n)), axis=1)
    cc = np.clip(c, 0, np.array(n) - 2)
    r = np.zeros(len(pts), dtype=np.float32)
    for a in (0, 1):
        for b in (0, 1):
            for e in (0, 1):
                w = (t[:, 0] if a else 1 - t[:, 0]) * (t[:, 1] if b else 1 - t[:, 1]) * (t[:, 2] if e else 1 - t[:, 2])
                r += w.astype(np.float32) * phi[cc[:, 0] + a, cc[:, 1] + b, cc[:, 2] + e]
    return pts[ok & (r < 0)]


@pytest.mark.gpu
def test_gmpm_executable_matches_oracle_with_level_set_model(tmp_path):
    """The `gmpm` executable end to end against the CPU ORACLE driven through the same scene: one analytic sphere
    (fixed-corotated) and one model sampled from a text level set (tests/golden/g11_ball.sdf, the *.sdf branch of
    sample_model), two output frames with adaptive dt."""
    import shutil
    import __graft_entry__ as g
    g.build_host()
    from claymore_amd import _ffi
    from claymore_amd.engine import Engine
    from oracle_ffi import oracle_api
    from parity_util import match
    shutil.copy(os.path.join(ROOT, "tests", "golden", "g11_ball.sdf"), tmp_path / "ball.sdf")
    vol = float(np.float32((1 / 64) ** 3 / 8))
    scene = {
        "simulation": {"gpuid": 0, "fps": 500, "frames": 2, "default_dt": 1e-4, "domain_bits": 6, "output_dir": str(tmp_path)},
        "models": [
            {"file": "sphere", "constitutive": "fixed_corotated", "rho": 1e3, "volume": vol, "youngs_modulus": 5e3, "poisson_ratio": 0.4,
             "offset": [0.34375, 0.421875, 0.421875], "span": [0.15625] * 3, "velocity": [0.5, 0, 0]},
            {"file": "ball.sdf", "constitutive": "fixed_corotated", "rho": 1e3, "volume": vol, "youngs_modulus": 5e3, "poisson_ratio": 0.4,
             "offset": [0.53125, 0.40625, 0.40625], "span": [0.1875] * 3, "velocity": [-0.5, 0, 0]},
        ],
    }
    fn = tmp_path / "scene.json"
    fn.write_text(json.dumps(scene))
    subprocess.check_output([os.path.join(HOST, "gmpm"), "-f", str(fn)], text=True)
    p0 = read_bgeo(tmp_path / "model_id[0]_frame[0].bgeo")
    p1 = read_bgeo(tmp_path / "model_id[1]_frame[0].bgeo")
    # the level-set sampler against an independent numpy evaluation of the same rule
    want = _sample_sdf_numpy(tmp_path / "ball.sdf", 6, scene["models"][1]["offset"], scene["models"][1]["span"])
    assert p1.shape == want.shape and p1.shape[0] > 500
    idx, _ = match(want.astype(np.float64), p1.astype(np.float64))
    assert np.array_equal(p1[idx], want)
    # the same scene through the oracle
    eng = Engine(domain_bits=6, max_ppc=32, api=oracle_api())
    prm = dict(volume=vol, rho=1e3, youngs_modulus=5e3, poisson_ratio=0.4)
    eng.init_model(_ffi.FIXED_COROTATED, p0, (0.5, 0, 0), **prm)
    eng.init_model(_ffi.FIXED_COROTATED, p1, (-0.5, 0, 0), **prm)
    frames = {}
    eng.main_loop(2, 500, 1e-4, on_frame=lambda f, e: frames.update({f: [e.retrieve_positions(0), e.retrieve_positions(1)]}))
    for f in (1, 2):
        for m in (0, 1):
            a = read_bgeo(tmp_path / f"model_id[{m}]_frame[{f}].bgeo")
            b = frames[f][m]
            assert a.shape == b.shape
            idx, _ = match(a.astype(np.float64), b.astype(np.float64))
            rel = np.abs(a.astype(np.float64) - b[idx]).max(axis=1) / np.abs(a).max(axis=1)
            assert rel.max() < 1e-5, rel.max()
    eng.close()


@pytest.mark.gpu
def test_mgsp_executable_same_device(tmp_path):
    """The in-process multi-context driver (reference structure: one process, N devices) on one GPU: two lattice boxes
    one cell apart share halo blocks; particle counts are conserved and the centroids fall with MGSP's gravity."""
    import __graft_entry__ as g
    g.build_host()
    out = subprocess.check_output([os.path.join(HOST, "mgsp"), "--devices", "2", "--same-device", "--bits", "7", "--frames", "2",
                                   "--fps", "100", "--out", str(tmp_path)], text=True)
    assert out.count("total number of particles") == 4
    for d in (0, 1):
        p0 = read_bgeo(tmp_path / f"model_dev[{d}]_frame[0].bgeo")
        p2 = read_bgeo(tmp_path / f"model_dev[{d}]_frame[2].bgeo")
        assert p0.shape == p2.shape and p0.shape[0] == 27 ** 3 * 8
        t = 2 / 100.0
        dy = p2[:, 1].astype(np.float64).mean() - p0[:, 1].astype(np.float64).mean()
        assert abs(dy - (-0.5 * 4.9 * t * t)) < 0.03 * 0.5 * 4.9 * t * t    # free fall under -9.8 * 0.5 (settings.h:108)
        assert abs(p2[:, 0].mean() - p0[:, 0].mean()) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["in-process", "peer"])
def test_mgsp_executable_equals_oracle(tmp_path, transport):
    """The `mgsp` executable (scenario 2 of Projects/MGSP/mgsp.cu:34-81: one lattice cube per device, MGSP gravity -4.9 and CFL 0.3,
    adaptive dt) against the single-rank CPU oracle driven through MgspBenchmark::main_loop on the union of the cubes: the frames the
    executable writes must hold the oracle's particles within 1e-5 relative.  "peer": the peer-direct transport's code path
    (MPM_GROUP_TRANSPORT=peer: hipMemcpyPeerAsync on the comm stream behind the peer's event, no host synchronisation - the reference's
    cudaMemcpyPeerAsync, halo_buffer.cuh:54-59) with both contexts on the one GPU a test box has; between two devices the same code
    runs with peer access enabled."""
    import __graft_entry__ as g
    from test_mgsp_gpu import _oracle_mgsp_main_loop
    from parity_util import match
    from claymore_amd import _ffi
    g.build_host()
    bits, frames, fps = 7, 2, 200
    env = dict(os.environ, MPM_GROUP_VERBOSE="1")
    if transport == "peer":
        env["MPM_GROUP_TRANSPORT"] = "peer"
    else:
        env.pop("MPM_GROUP_TRANSPORT", None)
    run = subprocess.run([os.path.join(HOST, "mgsp"), "--devices", "2", "--same-device", "--bits", str(bits), "--frames", str(frames),
                          "--fps", str(fps), "--out", str(tmp_path)], text=True, capture_output=True, env=env, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    out = run.stdout
    assert ("peer-direct (hipMemcpyPeerAsync)" in run.stderr) == (transport == "peer"), run.stderr[-1000:]   # the library says which copies it issues
    steps_exe = int(out.strip().splitlines()[-1].split()[-2])
    n = 1 << bits
    dx = 1.0 / n
    length, stride, o = 54 * n // 256, 56 * n // 256, 18 * n // 256        # host/mgsp.cpp scenario 2 == mgsp.cu:51-79
    models = []
    for d in range(2):
        lo = (o + (stride if d & 1 else 0), o + (d >> 1) * stride, o)
        hi = tuple(c + length for c in lo)
        xyz = scenes.lattice_box(bits, lo, hi)
        assert np.array_equal(np.sort(read_bgeo(tmp_path / f"model_dev[{d}]_frame[0].bgeo"), axis=0), np.sort(xyz, axis=0))   # same lattice as the executable's sampler
        models.append({"material": _ffi.FIXED_COROTATED, "xyz": xyz, "v0": (0.0, 0.0, 0.0), "params": {"volume": float(np.float32(dx) ** 3 / np.float32(8.0))}})
    sc = {"name": "mgsp_scenario2", "bits": bits, "dt": 1e-4, "config": {"max_ppc": 128, "gravity": -9.8 * 0.5, "cfl": 0.3}, "models": models}
    want, steps = _oracle_mgsp_main_loop(sc, frames, fps, 1e-4)
    assert steps == steps_exe, (steps, steps_exe)
    for d in range(2):
        got = read_bgeo(tmp_path / f"model_dev[{d}]_frame[{frames}].bgeo")
        xo = want[d][0]
        assert got.shape == xo.shape
        idx, _ = match(xo.astype(np.float64), got.astype(np.float64))
        rel = np.abs(got[idx].astype(np.float64) - xo).max(axis=1) / np.abs(xo).max(axis=1)
        assert rel.max() < 1e-5, (d, rel.max())
