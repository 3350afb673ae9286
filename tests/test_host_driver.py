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
