"""ctypes mirror of include/claymore_amd.h.

`bind(lib, prefix)` attaches argument/return types for every entry point the header declares; the
product binds libclaymore_hip.so with prefix "mpm_", the tests additionally bind the CPU oracle
(oracle/libmpm_oracle.so) with prefix "mpmo_" so both are driven through identical call sequences.
"""
import ctypes as C
import os

MPM_OK, MPM_ERR_INVALID, MPM_ERR_DEVICE, MPM_ERR_CAPACITY, MPM_ERR_NONFINITE, MPM_ERR_NOT_READY, MPM_ERR_INTERNAL = range(7)
J_FLUID, FIXED_COROTATED, SAND, NACC = range(4)
MATERIAL_NAMES = {"jfluid": J_FLUID, "fixed_corotated": FIXED_COROTATED, "sand": SAND, "nacc": NACC}


class Config(C.Structure):
    _fields_ = [("domain_bits", C.c_int), ("max_ppc", C.c_int), ("boundary_blocks", C.c_int),
                ("gravity", C.c_float), ("cfl", C.c_float), ("max_blocks", C.c_int64),
                ("grow", C.c_int), ("drop_overflow", C.c_int), ("sync_interval", C.c_int), ("reserved", C.c_int * 3)]


class MaterialParams(C.Structure):
    _fields_ = [("rho", C.c_float), ("volume", C.c_float), ("youngs_modulus", C.c_float),
                ("poisson_ratio", C.c_float), ("bulk", C.c_float), ("gamma", C.c_float),
                ("viscosity", C.c_float), ("beta", C.c_float), ("xi", C.c_float),
                ("cohesion", C.c_float), ("yield_surface", C.c_float), ("msqr", C.c_float),
                ("log_jp0", C.c_float), ("volume_correction", C.c_int), ("hardening_on", C.c_int),
                ("reserved", C.c_int * 5)]


class CollisionObject(C.Structure):
    _fields_ = [("type", C.c_int), ("friction", C.c_float), ("scale", C.c_float), ("dsdt", C.c_float),
                ("trans", C.c_float * 3), ("trans_vel", C.c_float * 3), ("omega", C.c_float * 3),
                ("rot_mat", C.c_float * 9), ("time", C.c_float), ("reserved", C.c_int * 3)]


class Counts(C.Structure):
    _fields_ = [("particle_blocks", C.c_int), ("neighbor_blocks", C.c_int), ("exterior_blocks", C.c_int),
                ("model_count", C.c_int), ("bins", C.c_int64 * 8), ("particles", C.c_int64 * 8)]


class Diagnostics(C.Structure):
    _fields_ = [("lost_particles", C.c_int64), ("discarded_p2g", C.c_int64), ("overflow_flags", C.c_int),
                ("dropped_particles", C.c_int), ("reserved", C.c_int * 4)]


class Timers(C.Structure):
    _fields_ = [("grid_update_ms", C.c_float), ("g2p2g_ms", C.c_float), ("partition_ms", C.c_float),
                ("halo_ms", C.c_float), ("total_ms", C.c_float), ("reserved", C.c_float * 3)]


_P = C.POINTER
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_fp, _ip = _P(C.c_float), _P(C.c_int)

# name -> (restype, argtypes); names are without prefix.  Keep in sync with include/claymore_amd.h.
SIGNATURES = {
    "default_config": (_i, [_i, _P(Config)]),
    "default_material": (_i, [_i, _i, _P(MaterialParams)]),
    "create": (_i, [_P(Config), _i, _P(_vp)]),
    "destroy": (None, [_vp]),
    "last_error": (C.c_char_p, [_vp]),
    "add_model": (_i, [_vp, _i, _P(MaterialParams), _vp, _sz, _fp, _ip]),
    "initial_setup": (_i, [_vp]),
    "grid_update": (_i, [_vp, _f, _fp]),
    "compute_dt": (_f, [_vp, _f, _f, _f, _f]),
    "g2p2g": (_i, [_vp, _f, _f]),
    "rebuild_partition": (_i, [_vp, _P(Counts)]),
    "substep": (_i, [_vp, _f, _f, _f, _f, _fp, _fp]),
    "run_fixed": (_i, [_vp, _i, _f]),
    "retrieve_positions": (_i, [_vp, _i, _vp, _P(_sz)]),
    "retrieve_state": (_i, [_vp, _i, _vp, _vp, _vp, _P(_sz)]),
    "state_kind": (_i, []),
    "get_counts": (_i, [_vp, _P(Counts)]),
    "get_timers": (_i, [_vp, _P(Timers)]),
    "grid_totals": (_i, [_vp, _P(C.c_double)]),
    "dump_grid": (_i, [_vp, _vp, _vp, _P(_sz)]),
    "check_table": (_i, [_vp]),
    "default_collision_object": (_i, [_P(CollisionObject)]),
    "set_collision_object": (_i, [_vp, _P(CollisionObject), _vp, _vp, _vp, _vp]),
    "test_eig": (_i, [_vp, _sz, _vp, _i]),
    "test_stress": (_i, [_i, _P(MaterialParams), _vp, _vp, _sz, _vp, _i]),
}
HALO = {
    "halo_keys": (_i, [_vp, _vp, _i, _ip]),
    "halo_tag_begin": (_i, [_vp]),
    "halo_tag_peer": (_i, [_vp, _i, _vp, _i]),
    "halo_tag_end": (_i, [_vp, _ip, _ip]),
    "g2p2g_halo": (_i, [_vp, _f, _f]),
    "g2p2g_interior": (_i, [_vp, _f, _f]),
    "halo_collect": (_i, [_vp, _i, _i, _vp, _vp, _i, _ip]),
    "halo_reduce": (_i, [_vp, _i, _vp, _vp, _i]),
    "mgsp_begin": (_i, [_vp, _f, _f]),
    "mgsp_rebuild_export": (_i, [_vp, _vp, _i]),
    "mgsp_tag": (_i, [_vp, _vp, _i, _i, _i]),
    "mgsp_end": (_i, [_vp, _ip, _ip, _ip, _fp]),
}
SIGNATURES.update(HALO)
# entry points only the HIP library has (stream plumbing + kernel timing)
HIP_ONLY = {
    "build_info": (C.c_char_p, []),
    "last_g2p2g_ms": (_i, [_vp, _fp]),
    "streams": (_i, [_vp, _P(_vp), _P(_vp)]),
    "sync": (_i, [_vp]),
    "get_capacity": (_i, [_vp, _P(C.c_int64), _P(C.c_int64), _ip]),
    "get_diagnostics": (_i, [_vp, _P(Diagnostics)]),
    "checkpoint_size": (_i, [_vp, _P(_sz)]),
    "checkpoint_save": (_i, [_vp, _vp, _sz, _P(_sz)]),
    "checkpoint_load": (_i, [_vp, _vp, _sz]),
    "group_unique_id": (_i, [_vp]),
    "group_create": (_i, [_vp, _i, _i, _vp, _P(_vp)]),
    "group_create_local": (_i, [_P(_vp), _i, _P(_vp)]),
    "group_destroy": (None, [_vp]),
    "group_last_error": (C.c_char_p, [_vp]),
    "group_transport": (C.c_char_p, [_vp]),
    "group_initial_setup": (_i, [_vp]),
    "group_resume": (_i, [_vp]),
    "group_substep": (_i, [_vp, _f, _f, _fp]),
    "group_run_fixed": (_i, [_vp, _i, _f]),
    "group_compute_dt": (_f, [_vp, _f, _f, _f, _f]),
    "group_main_loop": (_i, [_vp, _i, _i, _f, _vp, _vp, _ip]),
    "group_stats": (_i, [_vp, _ip, _ip, _fp]),
}


class Api:
    """Namespace of bound functions without their prefix (api.create, api.substep ...)."""

    def __init__(self, lib, prefix, names):
        self.lib, self.prefix = lib, prefix
        for name, (res, args) in names.items():
            fn = getattr(lib, prefix + name)  # AttributeError if the symbol is missing: fail loudly
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)


def bind(lib, prefix, hip=False):
    names = dict(SIGNATURES)
    if hip:
        names.update(HIP_ONLY)
    return Api(lib, prefix, names)


def repo_root():
    return os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HIP_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libclaymore_hip.so")


def load_hip():
    """Load the HIP engine.  No fallback: a missing library is an error."""
    if not os.path.exists(HIP_LIB_PATH):
        raise RuntimeError(
            f"{HIP_LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback in the product path)")
    lib = C.CDLL(HIP_LIB_PATH, mode=C.RTLD_GLOBAL)
    return bind(lib, "mpm_", hip=True)
