import numpy as np

from claymore_amd import scenes


def test_lattice_rule():
    pts = scenes.lattice_box(7, (10, 10, 10), (12, 11, 11))
    assert pts.shape == (2 * 8, 3) and pts.dtype == np.float32
    dx = 1 / 128
    # every particle sits at node +- 0.25 dx and rounds to its node (GeometrySampler.h:11-37)
    frac = pts / dx - np.round(pts / dx)
    assert np.allclose(np.abs(frac), 0.25)
    assert set(np.round(pts[:, 0] / dx).astype(int)) == {10, 11}


def test_config_sizes():
    c1 = scenes.two_spheres()
    n1 = scenes.total_particles(c1)
    assert 45_000 < n1 < 53_000                      # "50k particles"
    s = scenes.lattice_sphere(8, (0.5, 0.6, 0.5), 53.0)
    assert 4.9e6 < s.shape[0] < 5.1e6                 # C2 ~5.0 M
    assert 128 * 306 * 128 * 8 == 40_108_032          # C3 box
    parts = scenes.split_slabs(c1["models"][0]["xyz"], 4)
    assert sum(p.shape[0] for p in parts) == c1["models"][0]["xyz"].shape[0]
    assert max(p.shape[0] for p in parts) - min(p.shape[0] for p in parts) <= 1
