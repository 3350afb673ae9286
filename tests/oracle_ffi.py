"""Loads the CPU oracle (oracle/libmpm_oracle.so) for the tests.  Test infrastructure only."""
import ctypes as C
import os

from claymore_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "libmpm_oracle.so")
_api = None


def oracle_api():
    global _api
    if _api is None:
        lib = C.CDLL(ORACLE_PATH)
        _api = _ffi.bind(lib, "mpmo_", hip=False)
        _api.raw = lib
        fp, ip, sz, f, i = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_size_t, C.c_float, C.c_int
        lib.mpmo_fn_bspline.argtypes = [C.c_void_p, sz, f, C.c_void_p]
        lib.mpmo_fn_node_index.argtypes = [C.c_void_p, sz, f, C.c_void_p]
        lib.mpmo_fn_dir_offset.argtypes = [i, i, i]
        lib.mpmo_fn_dir_offset.restype = i
        lib.mpmo_fn_dir_components.argtypes = [i, ip]
        lib.mpmo_fn_compute_dt.argtypes = [f] * 6
        lib.mpmo_fn_compute_dt.restype = f
        lib.mpmo_fn_compute_dt_mgsp.argtypes = [f] * 5
        lib.mpmo_fn_compute_dt_mgsp.restype = f
        lib.mpmo_fn_rot_angle_to_matrix.argtypes = [f, i, C.c_void_p]
        lib.mpmo_fn_query_sdf.argtypes = [C.c_void_p, C.c_void_p, sz, C.c_void_p]
        lib.mpmo_fn_query_sdf.restype = i
        lib.mpmo_fn_collision_resolve.argtypes = [C.c_void_p, C.c_void_p, sz, f, C.c_void_p]
        lib.mpmo_fn_collision_resolve.restype = i
        lib.mpmo_fn_mat.argtypes = [C.c_void_p] * 4
        lib.mpmo_fn_jfluid.argtypes = [C.c_void_p, C.c_void_p, sz, f, f, f, f, f, f, C.c_void_p]
        lib.mpmo_check_table.argtypes = [C.c_void_p]
        lib.mpmo_check_table.restype = i
        lib.mpmo_set_threads.argtypes = [C.c_void_p, i]
        lib.mpmo_set_threads.restype = i
        lib.mpmo_test_svd.argtypes = [C.c_void_p, sz, C.c_void_p, i]    # math::svd itself: oracle only (the HIP engine has no SVD on its path)
        lib.mpmo_test_svd.restype = i
    return _api
