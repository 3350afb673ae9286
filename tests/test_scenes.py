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


def test_weak_scaling_columns_layout():
    """bench.py --gpus N (weak scaling): N touching, block-aligned C3 columns that fit the 512^3 grid, one per rank."""
    assert scenes.sand_columns_layout(1) == [tuple(int(v) for v in ((512 - 128) // 2, 12, (512 - 128) // 2))]  # N = 1 is C3 itself
    for world in (2, 4, 8):
        lay = scenes.sand_columns_layout(world)
        assert len(set(lay)) == world
        for (x, y, z) in lay:
            assert x % 4 == 0 and z % 4 == 0 and y == 12
            assert 12 <= x and x + 128 <= 500 and 12 <= z and z + 128 <= 500   # inside the walls
        # every column shares a face with another one (halo blocks from the first substep)
        for a in lay:
            assert any((abs(a[0] - b[0]) == 128 and a[2] == b[2]) or (abs(a[2] - b[2]) == 128 and a[0] == b[0]) for b in lay if b != a)
    # small instance: the ranks' particle sets are disjoint and each equals a plain column
    parts = [scenes.sand_columns_rank(r, 4, bits=6, size_cells=(8, 14, 8)) for r in range(4)]
    allp = np.concatenate([p["models"][0]["xyz"] for p in parts])
    assert all(p["models"][0]["xyz"].shape[0] == 8 * 14 * 8 * 8 for p in parts)
    assert np.unique(allp, axis=0).shape[0] == allp.shape[0]
    try:
        scenes.sand_columns_layout(10)
        assert False
    except ValueError:
        pass


def test_block_aligned_slabs_share_no_particle_block():
    """scenes.split_slabs(block=...): the cut planes of bench.py's strong-scaling partition are particle-block faces - no block of the initial
    lattice is shared by two ranks, the pieces stay within 10 % of the equal share, and a cut that cannot keep that bound stays equal-count."""
    bits = 8
    dx = 1.0 / (1 << bits)
    xyz = scenes.lattice_box(bits, (13, 5, 19), (13 + 6, 5 + 154, 19 + 5))   # 154 cells = 38.5 block layers along y (C3: 77 layers, 8 ranks)
    block = scenes.block_faces(bits)

    def layers(p, axis):
        return set(((np.rint(p[:, axis] / np.float32(dx)).astype(np.int64) - 2) // 4).tolist())

    for world in (2, 4, 8):
        parts = scenes.split_slabs(xyz, world, 1, block=block)
        assert sum(p.shape[0] for p in parts) == xyz.shape[0]
        assert max(p.shape[0] for p in parts) <= 1.10 * xyz.shape[0] / world
        ls = [layers(p, 1) for p in parts]
        for a in range(world):
            for b in range(a + 1, world):
                assert not (ls[a] & ls[b]), (world, a, b)
        eq = scenes.split_slabs(xyz, world, 1)
        assert sum(len(layers(p, 1)) for p in eq) >= sum(len(s) for s in ls)
    # a body two blocks thick cannot be cut 8 ways on block faces: the equal-count cut stands
    thin = scenes.lattice_box(bits, (12, 12, 12), (20, 20, 20))
    parts = scenes.split_slabs(thin, 8, 1, block=block)
    assert max(p.shape[0] for p in parts) - min(p.shape[0] for p in parts) <= 1
    # the partition is a partition: same multiset of particles
    parts = scenes.split_slabs(xyz, 4, 1, block=block)
    allp = np.concatenate(parts)
    assert np.array_equal(np.sort(allp.view([("x", "f4"), ("y", "f4"), ("z", "f4")]), axis=0), np.sort(np.ascontiguousarray(xyz).view([("x", "f4"), ("y", "f4"), ("z", "f4")]), axis=0))
