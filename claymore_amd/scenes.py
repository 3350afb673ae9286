"""Deterministic synthetic scenes of BASELINE.json's configs.

Particles come from the reference's lattice rule `sample_uniform_box`
(Library/MnBase/Geometry/GeometrySampler.h:11-37): 8 particles per grid node at node +- 0.25 dx;
spheres are that lattice clipped by |x - c| <= R.  No RNG, no Data/ files.
"""
import numpy as np

from ._ffi import FIXED_COROTATED, J_FLUID, NACC, SAND  # noqa: F401


def lattice_box(bits, minc, maxc):
    """All lattice particles of nodes minc <= (i,j,k) < maxc; float32 (n,3)."""
    dx = np.float32(1.0 / (1 << bits))
    ax = [np.arange(minc[d], maxc[d], dtype=np.float64) for d in range(3)]
    off = np.array([-0.25, 0.25])
    # per axis: node*dx +- 0.25dx  (exact in fp32: dx is a power of two)
    coords = [((a[:, None] + off[None, :]) * float(dx)).reshape(-1) for a in ax]
    # the reference nests i,j,k then ii,jj,kk; the order of particles is irrelevant to the physics,
    # a plain tensor product is used here
    X, Y, Z = np.meshgrid(coords[0], coords[1], coords[2], indexing="ij")
    out = np.empty((X.size, 3), dtype=np.float32)
    out[:, 0] = X.reshape(-1)
    out[:, 1] = Y.reshape(-1)
    out[:, 2] = Z.reshape(-1)
    return out


def lattice_sphere(bits, center, radius_cells):
    dx = 1.0 / (1 << bits)
    c = np.asarray(center, dtype=np.float64)
    r = radius_cells * dx
    lo = np.floor((c - r) / dx).astype(int) - 1
    hi = np.ceil((c + r) / dx).astype(int) + 2
    pts = lattice_box(bits, lo, hi)
    d2 = ((pts.astype(np.float64) - c[None, :]) ** 2).sum(axis=1)
    return np.ascontiguousarray(pts[d2 <= r * r])


def _vol(bits):
    dx = 1.0 / (1 << bits)
    return float(np.float32(dx * dx * dx / 8.0))


def two_spheres(bits=7, radius_cells=9.0, gap_cells=None, speed=0.5, material=FIXED_COROTATED, youngs=5e3):
    """C1: two elastic spheres colliding (BASELINE config 1).  Default: centres (0.40,0.5,0.5)/(0.60,0.5,0.5),
    R = 9 dx, v = (+-0.5, 0, 0), fixed-corotated E = 5e3, nu = 0.4, rho = 1e3, vol = dx^3/8."""
    dx = 1.0 / (1 << bits)
    if gap_cells is None:
        c0, c1 = (0.40, 0.5, 0.5), (0.60, 0.5, 0.5)
    else:
        half = (radius_cells + gap_cells / 2.0) * dx
        c0, c1 = (0.5 - half, 0.5, 0.5), (0.5 + half, 0.5, 0.5)
    prm = {"volume": _vol(bits), "youngs_modulus": youngs, "poisson_ratio": 0.4, "rho": 1e3}
    return {
        "name": "two_spheres", "bits": bits, "dt": 1e-4,
        "config": {"max_ppc": 128},
        "models": [
            {"material": material, "xyz": lattice_sphere(bits, c0, radius_cells), "v0": (speed, 0.0, 0.0), "params": dict(prm)},
            {"material": material, "xyz": lattice_sphere(bits, c1, radius_cells), "v0": (-speed, 0.0, 0.0), "params": dict(prm)},
        ],
    }


def two_spheres_c4(bits=9, radius_cells=84.0, speed=1.0):
    """C4 (BASELINE config 4): two fixed-corotated spheres of R = 84 dx on a 512^3 grid, centres (0.30, 0.5, 0.5) / (0.70, 0.5, 0.5),
    approaching each other at +-1 m/s: 2 x 19.9 M particles; run as MGSP with 4 ranks (static particle partition, two slabs per sphere)."""
    prm = {"volume": _vol(bits), "youngs_modulus": 5e3, "poisson_ratio": 0.4, "rho": 1e3}
    return {
        "name": "two_spheres_c4", "bits": bits, "dt": 1e-4, "config": {"max_ppc": 128},
        "models": [
            {"material": FIXED_COROTATED, "xyz": lattice_sphere(bits, (0.30, 0.5, 0.5), radius_cells), "v0": (speed, 0.0, 0.0), "params": dict(prm)},
            {"material": FIXED_COROTATED, "xyz": lattice_sphere(bits, (0.70, 0.5, 0.5), radius_cells), "v0": (-speed, 0.0, 0.0), "params": dict(prm)},
        ],
    }


def sphere_drop(bits=8, radius_cells=53.0, center=(0.5, 0.6, 0.5), material=FIXED_COROTATED):
    """C2: one elastic sphere dropped under gravity (~5.0 M particles at bits 8, R = 53 dx)."""
    prm = {"volume": _vol(bits)} if material != SAND else {}
    return {"name": "sphere_drop", "bits": bits, "dt": 1e-4, "config": {"max_ppc": 128},
            "models": [{"material": material, "xyz": lattice_sphere(bits, center, radius_cells), "v0": (0, 0, 0), "params": prm}]}


def sand_column(bits=9, size_cells=(128, 306, 128), min_corner=None):
    """C3: Drucker-Prager sand column collapse; at bits 9 the default box holds 128*306*128*8 = 40.1 M particles.
    Reference sand defaults (particle_buffer.cuh:202-218) incl. its volume."""
    if min_corner is None:
        n = 1 << bits
        min_corner = ((n - size_cells[0]) // 2, 12, (n - size_cells[2]) // 2)
    lo = np.array(min_corner)
    hi = lo + np.array(size_cells)
    return {"name": "sand_column", "bits": bits, "dt": 1e-4, "config": {"max_ppc": 128},
            "models": [{"material": SAND, "xyz": lattice_box(bits, lo, hi), "v0": (0, 0, 0), "params": {}}]}


def scaled_sand_column(bits, fraction):
    """A geometrically similar, smaller sand column (same aspect ratio) for bounded CPU samples / tests."""
    s = fraction ** (1.0 / 3.0)
    full = np.array([128, 306, 128]) * (1 << bits) / 512.0
    size = np.maximum(4, np.round(full * s)).astype(int)
    return sand_column(bits, tuple(int(x) for x in size))


def sand_columns_layout(world, bits=9, size_cells=(128, 306, 128)):
    """Min corners of `world` adjacent sand columns on the floor of the domain, ceil(sqrt(world)) per row along x, rows
    along z, centred and block-aligned: the weak-scaling workload of bench.py (one C3 column per rank; touching columns
    share a face, so every rank has halo blocks from the first substep)."""
    n = 1 << bits
    cols = int(np.ceil(np.sqrt(world)))
    rows = (world + cols - 1) // cols
    x0 = ((n - size_cells[0] * cols) // 2) & ~3
    z0 = ((n - size_cells[2] * rows) // 2) & ~3
    if x0 < 12 or z0 < 12 or 12 + size_cells[1] > n - 12:
        raise ValueError(f"{world} columns of {size_cells} cells do not fit a {n}^3 grid")
    return [(x0 + size_cells[0] * (r % cols), 12, z0 + size_cells[2] * (r // cols)) for r in range(world)]


def sand_columns_rank(rank, world, bits=9, size_cells=(128, 306, 128)):
    """The particles of ONE rank of the weak-scaling workload (a rank never builds the other ranks' columns)."""
    return sand_column(bits, size_cells, sand_columns_layout(world, bits, size_cells)[rank])


def fluid_dam(bits=10, size_cells=(256, 192, 256), min_corner=(12, 12, 12)):
    """C5: weakly compressible J-fluid dam break (reference defaults particle_buffer.cuh:148-153).  The fixed substep of the
    bench / tests follows the acoustic CFL limit: sound speed sqrt(bulk * gamma / rho) = 16.9, dx / c = 2.3e-4 at bits 8 and
    5.8e-5 at bits 10 - dt = 1e-4 is fine up to bits 8 and halves with every bit beyond (2.5e-5 at bits 10; at 1e-4 the
    1024^3 run blows up within 2000 substeps, which bench.py's self-check reports as lost particles)."""
    lo = np.array(min_corner)
    hi = lo + np.array(size_cells)
    dt = 1e-4 if bits <= 8 else 1e-4 * 2.0 ** (8 - bits)
    return {"name": "fluid_dam", "bits": bits, "dt": dt, "config": {"max_ppc": 128},
            "models": [{"material": J_FLUID, "xyz": lattice_box(bits, lo, hi), "v0": (0, 0, 0), "params": {}}]}


def split_slabs(xyz, parts, axis=0, block=None, tolerance=0.10):
    """Static particle partition of MGSP: equal-count slabs of the initial lattice along `axis`.

    block = (origin, width): the particle blocks' faces along the axis lie at origin + k * width (block_faces()).  The equal-count cuts are then
    moved to the nearest face, so that no particle block is shared by two ranks at the start: a slab whose two cut planes pass through blocks
    owns two layers of half-filled blocks, and G2P2G's cost is per iteration of a block, not per particle (C3 cut 8 ways along y: 77 block layers,
    9.6 per rank - an equal-count slab touches 11 layers, an aligned one holds 9 or 10).  Kept only if no piece grows beyond (1 + tolerance) of
    the equal share (and none is empty); otherwise the equal-count cut stands."""
    col = xyz[:, axis]
    n = xyz.shape[0]
    sorted_already = col.size < 2 or bool(np.all(col[1:] >= col[:-1]))  # the lattice samplers emit particles sorted along x
    bounds = np.linspace(0, n, parts + 1).astype(np.int64)
    if block is not None and parts > 1 and n >= parts:
        order = None if sorted_already else np.argsort(col, kind="stable")
        cs = col if sorted_already else col[order]
        origin, width = block
        cuts = 0.5 * (cs[bounds[1:-1] - 1].astype(np.float64) + cs[bounds[1:-1]].astype(np.float64))  # between the two particles an equal-count cut separates
        faces = origin + np.round((cuts - origin) / width) * width
        snapped = np.concatenate([[0], np.searchsorted(cs, faces.astype(cs.dtype), side="left"), [n]]).astype(np.int64)
        sizes = np.diff(snapped)
        if sizes.min() > 0 and sizes.max() <= (1.0 + tolerance) * n / parts:
            bounds = snapped
        if order is not None:
            return [np.ascontiguousarray(xyz[order[bounds[i]:bounds[i + 1]]]) for i in range(parts)]
        return [np.ascontiguousarray(xyz[bounds[i]:bounds[i + 1]]) for i in range(parts)]
    if sorted_already:
        return [np.ascontiguousarray(xyz[bounds[i]:bounds[i + 1]]) for i in range(parts)]
    order = np.argsort(col, kind="stable")
    return [np.ascontiguousarray(xyz[idx]) for idx in np.array_split(order, parts)]


def block_faces(bits):
    """(origin, width) of the particle blocks' faces along any axis, in world units: a particle at x belongs to block (lround(x / dx) - 2) / 4
    (particle_block_key, mpm_kernels.hpp; reference: Projects/GMPM/mgmpm_kernels.cuh:21-36), i.e. block k holds x / dx in [4 k + 1.5, 4 k + 5.5)."""
    dx = 1.0 / (1 << bits)
    return 1.5 * dx, 4.0 * dx


def split_boxes(xyz, shape, block=None):
    """Static particle partition into shape = (nx, ny, nz) equal-count boxes of the initial lattice: nx slabs along x, each cut into ny along
    y, each of those into nz along z; block: cut planes moved to particle-block faces (split_slabs) (the reference's scenarios put their per-GPU bodies side by side in x and z, Projects/MGSP/mgsp.cu:52-57,
    :67-72; SURVEY 8(e): "slabs / octants").  Returns the nx * ny * nz particle arrays, x-major."""
    parts = [xyz]
    for axis, n in enumerate(shape):
        parts = [q for p in parts for q in split_slabs(p, n, axis, block=block)] if n > 1 else parts
    return parts


PARTITION_SHAPES = {  # name -> world size -> (nx, ny, nz); "octants" halves every axis it can (x first: 2 -> x, 4 -> x z, 8 -> x y z)
    "y": lambda w: (1, w, 1), "x": lambda w: (w, 1, 1), "z": lambda w: (1, 1, w),
    "octants": lambda w: {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 1, 2), 8: (2, 2, 2)}[w],
    "xz-columns": lambda w: {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 1, 2), 8: (4, 1, 2)}[w],
    "y-x": lambda w: {1: (1, 1, 1), 2: (1, 2, 1), 4: (2, 2, 1), 8: (2, 4, 1)}[w],
}


def total_particles(scene):
    return int(sum(m["xyz"].shape[0] for m in scene["models"]))


def sphere_level_set(bits, centre, radius):
    """Node-sampled signed distance (negative inside) and gradient of a sphere, in the layout of the reference's
    `<name>_sdf.bin` / `_grad_{0,1,2}.bin` files (Projects/MGSP/boundary_condition.cuh:252-321): (N, N, N) and (3, N, N, N)."""
    n = 1 << bits
    ax = np.arange(n, dtype=np.float64) / n
    X, Y, Z = np.meshgrid(ax - centre[0], ax - centre[1], ax - centre[2], indexing="ij")
    r = np.sqrt(X * X + Y * Y + Z * Z)
    rs = np.maximum(r, 1e-12)
    return (r - radius).astype(np.float32), np.stack([X / rs, Y / rs, Z / rs]).astype(np.float32)


def sphere_on_obstacle(bits=6, radius_cells=5.0, obstacle_cells=6.0, boundary="slip", friction=0.3, speed=1.0, material=FIXED_COROTATED):
    """An elastic sphere thrown onto a fixed spherical level-set obstacle (the MGSP collision-object path)."""
    dx = 1.0 / (1 << bits)
    c_obs = (0.5, 0.34, 0.5)
    c_sph = (0.5 + 1.5 * dx, c_obs[1] + (obstacle_cells + radius_cells + 1.5) * dx, 0.5)
    vol = float(np.float32(dx ** 3 / 8))
    sdf, grad = sphere_level_set(bits, c_obs, obstacle_cells * dx)
    return {"name": "sphere_on_obstacle", "bits": bits, "dt": 1e-4, "config": {},
            "models": [{"material": material, "xyz": lattice_sphere(bits, c_sph, radius_cells), "v0": (0.0, -speed, 0.0),
                        "params": {"volume": vol, "youngs_modulus": 5e3, "poisson_ratio": 0.4, "rho": 1e3}}],
            "collision": {"sdf": sdf, "grad": grad, "type": {"sticky": 0, "slip": 1, "separate": 2}[boundary], "friction": friction}}
