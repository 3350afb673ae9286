"""Host-side mirror of the reference's GmpmSimulator surface (Projects/GMPM/gmpm_simulator.cuh:25-783)
on top of the C ABI.  Method names follow the reference: init_model, update_fr_parameters,
update_j_fluid_parameters, update_nacc_parameters, main_loop / substep, output_model."""
import ctypes as C
import math

import numpy as np

from . import _ffi


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mpm status {code}: {msg}")
        self.code = code


class Engine:
    """One simulation context on one device.

    api: a bound _ffi.Api.  The product always uses the HIP library (`Engine(cfg)`); tests may hand in
    the oracle's Api to drive the checker through the same call sequence.
    """

    def __init__(self, cfg=None, device=0, api=None, domain_bits=None, **overrides):
        self.api = api if api is not None else _ffi.load_hip()
        if cfg is None:
            cfg = _ffi.Config()
            self._check(self.api.default_config(int(domain_bits), C.byref(cfg)))
        for k, v in overrides.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self.ctx = C.c_void_p()
        rc = self.api.create(C.byref(cfg), int(device), C.byref(self.ctx))
        if rc != 0:
            raise EngineError(rc, "mpm_create failed (no usable HIP device? there is no CPU fallback)")
        self.dx = 1.0 / (1 << cfg.domain_bits)
        self.models = []
        self.cur_time = 0.0
        self.dt = None

    # ---- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "ctx", None):
            self.api.destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self.api.last_error(self.ctx).decode() if getattr(self, "ctx", None) else ""
            raise EngineError(rc, msg)

    # ---- model setup (gmpm_simulator.cuh:168-254) -------------------------------------------------
    def default_material(self, material):
        p = _ffi.MaterialParams()
        self._check(self.api.default_material(int(material), self.cfg.domain_bits, C.byref(p)))
        return p

    def init_model(self, material, positions, v0=(0.0, 0.0, 0.0), params=None, **param_overrides):
        xyz = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        p = params if params is not None else self.default_material(material)
        for k, v in param_overrides.items():
            setattr(p, k, v)
        v = (C.c_float * 3)(*[float(x) for x in v0])
        mid = C.c_int(-1)
        self._check(self.api.add_model(self.ctx, int(material), C.byref(p), xyz.ctypes.data_as(C.c_void_p),
                                       xyz.shape[0], v, C.byref(mid)))
        self.models.append({"material": int(material), "n": xyz.shape[0], "params": p})
        return mid.value

    def initial_setup(self):
        self._check(self.api.initial_setup(self.ctx))

    # ---- phases (gmpm_simulator.cuh:326-579) ------------------------------------------------------
    def grid_update(self, dt):
        mv = C.c_float(0)
        self._check(self.api.grid_update(self.ctx, dt, C.byref(mv)))
        return mv.value

    def compute_dt(self, max_vel, cur, nxt, dt_default):
        return self.api.compute_dt(self.ctx, max_vel, cur, nxt, dt_default)

    def g2p2g(self, dt, next_dt):
        self._check(self.api.g2p2g(self.ctx, dt, next_dt))

    def rebuild_partition(self):
        c = _ffi.Counts()
        self._check(self.api.rebuild_partition(self.ctx, C.byref(c)))
        return c

    def substep(self, dt, step_time, frame_time, dt_default):
        nd, mv = C.c_float(0), C.c_float(0)
        self._check(self.api.substep(self.ctx, dt, step_time, frame_time, dt_default, C.byref(nd), C.byref(mv)))
        return nd.value, mv.value

    def run_fixed(self, nsteps, dt):
        self._check(self.api.run_fixed(self.ctx, int(nsteps), dt))

    def main_loop(self, nframes, fps, dt_default, on_frame=None):
        """GmpmSimulator::main_loop (gmpm_simulator.cuh:303-592): adaptive dt, `nframes` frames."""
        spf = 1.0 / fps
        max_v0 = 0.0
        dt = self.compute_dt(max_v0, 0.0, spf, dt_default)
        self.initial_setup()
        steps = 0
        for frame in range(1, nframes + 1):
            t = 0.0
            while t < spf:
                next_dt, _ = self.substep(dt, t, spf, dt_default)
                t += dt
                self.cur_time += dt
                dt = next_dt
                steps += 1
            if on_frame:
                on_frame(frame, self)
        return steps

    # ---- output (gmpm_simulator.cuh:594-634) ------------------------------------------------------
    def retrieve_positions(self, model=0):
        n = C.c_size_t(self.models[model]["n"])
        xyz = np.empty((n.value, 3), dtype=np.float32)
        self._check(self.api.retrieve_positions(self.ctx, model, xyz.ctypes.data_as(C.c_void_p), C.byref(n)))
        return xyz[: n.value]

    def retrieve_state(self, model=0):
        n = C.c_size_t(self.models[model]["n"])
        xyz = np.empty((n.value, 3), dtype=np.float32)
        st = np.empty((n.value, 9), dtype=np.float32)
        lj = np.empty((n.value,), dtype=np.float32)
        self._check(self.api.retrieve_state(self.ctx, model, xyz.ctypes.data_as(C.c_void_p),
                                            st.ctypes.data_as(C.c_void_p), lj.ctypes.data_as(C.c_void_p), C.byref(n)))
        k = n.value
        return xyz[:k], st[:k], lj[:k]

    def counts(self):
        c = _ffi.Counts()
        self._check(self.api.get_counts(self.ctx, C.byref(c)))
        return c

    def set_collision_object(self, sdf=None, grad=None, **fields):
        """Install the level-set collision object of the MGSP grid update (MgspBenchmark::init_boundary,
        mgsp_benchmark.cuh:257-266).  sdf: (N, N, N) float32 node values, grad: (3, N, N, N); fields: type, friction, scale,
        dsdt, trans, trans_vel, omega, rot_mat, time.  sdf=None removes the object."""
        if sdf is None:
            self._check(self.api.set_collision_object(self.ctx, None, None, None, None, None))
            return
        obj = _ffi.CollisionObject()
        self._check(self.api.default_collision_object(C.byref(obj)))
        for k, v in fields.items():
            cur = getattr(obj, k)
            if hasattr(cur, "__len__"):
                for i, x in enumerate(np.asarray(v, dtype=np.float32).ravel()):
                    cur[i] = float(x)
            else:
                setattr(obj, k, v)
        n = 1 << self.cfg.domain_bits
        sdf = np.ascontiguousarray(sdf, dtype=np.float32)
        grad = np.ascontiguousarray(grad, dtype=np.float32)
        assert sdf.shape == (n, n, n) and grad.shape == (3, n, n, n), (sdf.shape, grad.shape)
        self._check(self.api.set_collision_object(self.ctx, C.byref(obj), sdf.ctypes.data, grad[0].ctypes.data,
                                                  grad[1].ctypes.data, grad[2].ctypes.data))

    def save_checkpoint(self):
        """Full state at the current substep boundary as a numpy byte array (HIP engine only)."""
        n = C.c_size_t(0)
        self._check(self.api.checkpoint_size(self.ctx, C.byref(n)))
        buf = np.empty(n.value, dtype=np.uint8)
        w = C.c_size_t(0)
        self._check(self.api.checkpoint_save(self.ctx, buf.ctypes.data, buf.size, C.byref(w)))
        return buf[: w.value]

    def load_checkpoint(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self._check(self.api.checkpoint_load(self.ctx, buf.ctypes.data, buf.size))

    def capacity(self):
        """(block capacity, per-model bin capacities, number of growth events) - HIP engine only."""
        blocks, bins, events = C.c_int64(0), (C.c_int64 * 8)(), C.c_int(0)
        self._check(self.api.get_capacity(self.ctx, C.byref(blocks), bins, C.byref(events)))
        return blocks.value, list(bins), events.value

    def model_mass(self, model=0):
        """Particle mass of a model: volume * rho (particle_buffer.cuh:158,:186,:252)."""
        p = self.models[model]["params"]
        return float(np.float32(p.volume) * np.float32(p.rho))

    def diagnostics(self):
        """Particles / P2G contributions the reference would have lost silently (HIP library only)."""
        d = _ffi.Diagnostics()
        self._check(self.api.get_diagnostics(self.ctx, C.byref(d)))
        return d

    def timers(self):
        t = _ffi.Timers()
        self._check(self.api.get_timers(self.ctx, C.byref(t)))
        return t

    def grid_totals(self):
        out = (C.c_double * 4)()
        self._check(self.api.grid_totals(self.ctx, out))
        return np.array(list(out))

    def dump_grid(self):
        cnt = self.counts()
        n = C.c_size_t(cnt.neighbor_blocks)
        keys = np.empty((n.value, 3), dtype=np.int32)
        blocks = np.empty((n.value, 4, 64), dtype=np.float32)
        self._check(self.api.dump_grid(self.ctx, keys.ctypes.data_as(C.c_void_p), blocks.ctypes.data_as(C.c_void_p), C.byref(n)))
        return keys[: n.value], blocks[: n.value]

    def last_g2p2g_ms(self):
        ms = C.c_float(0)
        self._check(self.api.last_g2p2g_ms(self.ctx, C.byref(ms)))
        return ms.value


def build_engine(scene, device=0, api=None):
    """Create an Engine for a scene dict (claymore_amd.scenes) and add its models (no setup yet)."""
    cfg = _ffi.Config()
    a = api if api is not None else _ffi.load_hip()
    rc = a.default_config(scene["bits"], C.byref(cfg))
    if rc:
        raise EngineError(rc, "default_config")
    for k, v in scene.get("config", {}).items():
        setattr(cfg, k, v)
    eng = Engine(cfg=cfg, device=device, api=a)
    for m in scene["models"]:
        eng.init_model(m["material"], m["xyz"], m.get("v0", (0, 0, 0)), **m.get("params", {}))
    if scene.get("collision"):
        eng.set_collision_object(**scene["collision"])
    return eng
