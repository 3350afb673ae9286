"""Random small scenes for the HIP-vs-oracle fuzz (tests/test_parity_gpu.py::test_fuzz_parity, tools/fuzz_parity.py): one to
three bodies (lattice spheres / boxes) of random material, size, position (sometimes next to a wall) and velocity."""
import numpy as np

from claymore_amd import _ffi, scenes

ALL_MATERIALS = (_ffi.J_FLUID, _ffi.FIXED_COROTATED, _ffi.SAND, _ffi.NACC)


def random_scene(rng, case, vmax=3.0, maxsteps=120, materials=ALL_MATERIALS):
    bits = int(rng.choice([5, 6]))
    n = 1 << bits
    models = []
    for _ in range(int(rng.integers(1, 4))):
        mat = int(rng.choice(materials))
        r = float(rng.uniform(2.5, 5.5 if bits == 6 else 3.5))
        c = rng.uniform(0.25, 0.75, size=3)
        if rng.random() < 0.3:       # near a wall / the floor
            c[int(rng.integers(0, 3))] = float(rng.choice([(r + 8.5) / n, 1 - (r + 8.5) / n]))
        if rng.random() < 0.5:
            xyz = scenes.lattice_sphere(bits, tuple(c), r)
        else:
            lo = np.floor(c * n - r).astype(int)
            xyz = scenes.lattice_box(bits, lo, lo + np.maximum(2, np.round(rng.uniform(0.6, 1.0, 3) * 2 * r)).astype(int))
        v0 = tuple(float(x) for x in rng.uniform(-vmax, vmax, size=3) * (rng.random() < 0.8))
        prm = {}
        if mat == _ffi.FIXED_COROTATED:
            prm = {"volume": scenes._vol(bits), "youngs_modulus": float(rng.choice([2e3, 5e3, 2e4])), "poisson_ratio": 0.4, "rho": 1e3}
        models.append({"material": mat, "xyz": xyz, "v0": v0, "params": prm})
    sc = {"name": f"fuzz{case}", "bits": bits, "dt": 1e-4, "config": {"max_ppc": 128}, "models": models}
    return sc, int(rng.integers(20, maxsteps))
