"""MGSP: static-particle-partition multi-GPU driver (Projects/MGSP/mgsp_benchmark.cuh:361-776), one process per
GPU over torch.distributed (backend "nccl" == RCCL over xGMI; "gloo" on CPU in the tests).

Each rank owns a fixed subset of the particles for the whole run (reference: one model per device,
mgsp_benchmark.cuh:240-253) and builds its own sparse partition and grid wherever those particles are.  Grids of
different ranks overlap only in blocks touched by particles of both.  Per substep:

  grid update (redundant but identical on every owner of a shared block)
  G2P2G on the halo particle blocks                                       (compute stream)
  collect shared grid blocks -> ONE all-to-all-v -> reduce into own grid   (comm stream, overlaps with ...)
  G2P2G on the interior particle blocks                                    (compute stream)
  partition rebuild
  halo tagging: ONE all-gather of neighbor-block keys -> overlap marks, send lists, halo/interior split

The reference moves the same data with pairwise cudaMemcpyPeerAsync (halo_buffer.cuh:54-59) and copies of the key
lists to every peer (mgsp_benchmark.cuh:681-686); on MI355X the all-to-all-v is a grouped send/recv that drives all
seven xGMI links of a GPU at once, and the exchange is symmetric (a block shared with p is sent to p and received
from p), so receive counts equal send counts and no count exchange is needed.

The class is generic over the engine API (`api`): the product uses the HIP library with CUDA tensors; the tests run
the very same logic on CPU tensors with the oracle's API and the gloo backend.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _ffi, scenes
from .engine import EngineError, build_engine

ROW = 3 + 256  # one halo record: key (3 x int32) + grid block (4 x 64 x f32)


PARTITION_SHAPE = None  # None: slabs along every model's longest axis; or a key of scenes.PARTITION_SHAPES ("y", "x", "z", "octants", ...): set by bench.py / the tools


ALIGN_TO_BLOCKS = False  # True (bench.py's strong scaling, the tools): the slabs' cut planes are moved to the nearest particle-block faces (scenes.split_slabs)


def partition_scene(scene, rank, world, axis=None, shape=None, align=None):
    """Static particle partition: every model is cut into `world` equal-count pieces of the initial lattice.  Default: slabs along the
    model's LONGEST axis (axis=None): slabs as thick as possible keep the interfaces - the halo blocks that have to be computed first, sent
    and reduced every substep - small against the interior that hides the exchange (the C3 column of 128 x 306 x 128 cells cut 8 ways along
    x would be 4 blocks thick, i.e. almost all halo; along y it is 9.5 blocks thick).  `shape` (or the module's PARTITION_SHAPE) names another
    cut of scenes.PARTITION_SHAPES - x / z slabs, octants, ... - compared in tools/mgsp_strong_local.py (profiles/r06_mgsp_partition.txt).
    `align` (default: the module's ALIGN_TO_BLOCKS): slabs whose cut planes are particle-block faces (scenes.split_slabs: pieces within 10 % of the
    equal share, no block shared by two ranks at the start)."""
    shape = shape if shape is not None else PARTITION_SHAPE
    align = ALIGN_TO_BLOCKS if align is None else align
    block = scenes.block_faces(scene["bits"]) if align and "bits" in scene else None
    sc = dict(scene)
    sc["models"] = []
    for m in scene["models"]:
        if shape is not None:
            mm = dict(m)
            mm["xyz"] = scenes.split_boxes(m["xyz"], scenes.PARTITION_SHAPES[shape](world), block=block)[rank]
            sc["models"].append(mm)
            continue
        ax = axis
        if ax is None:
            xyz = m["xyz"]
            ax = int(np.argmax(xyz.max(axis=0) - xyz.min(axis=0))) if xyz.shape[0] else 0
        part = scenes.split_slabs(m["xyz"], world, ax, block=block)[rank]
        mm = dict(m)
        mm["xyz"] = part
        sc["models"].append(mm)
    return sc


class TorchDistComm:
    """The three collectives of MGSP on torch.distributed (RCCL on GPUs, gloo in the CPU tests)."""

    def all_reduce_max(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.MAX)

    def all_gather(self, out, inp):
        dist.all_gather_into_tensor(out, inp)

    def all_to_all(self, recv, send, splits):
        dist.all_to_all_single(recv, send, splits, splits)


class MgspRank:
    def __init__(self, scene, rank, world, device=0, api=None, gravity=None, cfl=None, comm=None):
        self.rank, self.world = rank, world
        self.comm = comm if comm is not None else TorchDistComm()
        self.hip = api is None
        self.api = api if api is not None else _ffi.load_hip()
        local = partition_scene(scene, rank, world)
        if gravity is not None:
            local.setdefault("config", {})["gravity"] = gravity
        if cfl is not None:
            local.setdefault("config", {})["cfl"] = cfl
        self.n_local = scenes.total_particles(local)
        self.eng = build_engine(local, device=device, api=self.api)
        self.ctx = self.eng.ctx
        self.tdev = torch.device("cuda", device) if self.hip else torch.device("cpu")
        self.pad = 0            # padded key-list length of the tagging all-gather
        self.send_counts = [0] * world
        self.buf_send = self.buf_recv = None
        self._comm_stream = self._compute_stream = None
        if self.hip:
            cs, ms = C.c_void_p(), C.c_void_p()
            self._check(self.api.streams(self.ctx, C.byref(cs), C.byref(ms)))
            self._compute_stream = torch.cuda.ExternalStream(cs.value, device=self.tdev)
            self._comm_stream = torch.cuda.ExternalStream(ms.value, device=self.tdev)
        self._t = {"g2p2g": 0.0, "steps": 0}

    # ---- helpers ----------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, self.api.last_error(self.ctx).decode())

    def _on(self, stream):
        """Make torch (and therefore the RCCL collectives it enqueues) use one of the engine's own HIP streams."""
        return torch.cuda.stream(stream) if self.hip else _Null()

    def close(self):
        self.eng.close()

    # ---- setup ------------------------------------------------------------------------------------------
    def initial_setup(self):
        self.eng.initial_setup()
        self._tag()
        self._exchange(gid=0)  # initial rasterised grids are summed once (mgsp_benchmark.cuh:653-654)
        if self.hip:
            self._check(self.api.sync(self.ctx))

    # ---- halo tagging (mgsp_benchmark.cuh:661-720) --------------------------------------------------------
    def _tag(self):
        api, ctx, w = self.api, self.ctx, self.world
        n = self.eng.counts().neighbor_blocks
        if self.pad == 0:  # first call: agree on a padded length (every later length is derived from gathered counts)
            t = torch.tensor([n], dtype=torch.int32, device=self.tdev)
            with self._on(self._compute_stream):
                self.comm.all_reduce_max(t)
            self.pad = int(1.25 * int(t.item())) + 64
        while True:
            pad = self.pad  # identical on all ranks by construction
            with self._on(self._compute_stream):
                mine = torch.zeros((pad, 3), dtype=torch.int32, device=self.tdev)
                cnt = C.c_int(0)
                self._check(api.halo_keys(ctx, C.c_void_p(mine[1:].data_ptr()), pad - 1, C.byref(cnt)))
                mine[0, 0] = cnt.value          # row 0 carries the true count (keys are truncated if it exceeds pad-1)
                allk = torch.empty((w * pad, 3), dtype=torch.int32, device=self.tdev)
                self.comm.all_gather(allk, mine)
                head = allk.view(w, pad, 3)[:, 0, 0].cpu()
            counts = [int(head[p]) for p in range(w)]
            need = max(counts) + 1
            if need > pad:  # some rank outgrew the padding: every rank sees it and repeats with the same larger length
                self.pad = int(1.25 * need) + 64
                continue
            if need > 0.9 * pad:
                self.pad = int(1.25 * need) + 64  # grow ahead of time (takes effect at the next tagging)
            break
        self._check(api.halo_tag_begin(ctx))
        for p in range(w):
            if p != self.rank:
                seg = allk[p * pad + 1:]
                self._check(api.halo_tag_peer(ctx, p, C.c_void_p(seg.data_ptr()), counts[p]))
        nh = C.c_int(0)
        sc = (C.c_int * 32)()
        self._check(api.halo_tag_end(ctx, C.byref(nh), sc))
        self.send_counts = [int(sc[p]) if p != self.rank else 0 for p in range(w)]
        self.n_halo_blocks = nh.value
        self._keep = allk  # keep the gathered keys alive until the tagging kernels have run

    # ---- halo exchange (mgsp_benchmark.cuh:723-776) ---------------------------------------------------------
    def _exchange(self, gid, overlap_with=None):
        """collect -> all-to-all-v -> reduce on the comm stream; `overlap_with` (the interior G2P2G launch) is issued
        on the compute stream right after the transfer has been enqueued (mgsp_benchmark.cuh:449-466)."""
        api, ctx, w = self.api, self.ctx, self.world
        total = sum(self.send_counts)
        splits = [ROW * c for c in self.send_counts]
        need = max(ROW * total, 1)
        if self.buf_send is None or self.buf_send.numel() < need:
            self.buf_send = torch.empty(int(need * 1.5) + ROW, dtype=torch.float32, device=self.tdev)
            self.buf_recv = torch.empty_like(self.buf_send)
        send, recv = self.buf_send[: ROW * total], self.buf_recv[: ROW * total]
        # The big interior G2P2G is enqueued FIRST (compute stream): the collect kernels and the collective below go to
        # the comm stream behind an event that mgsp_begin / g2p2g_halo recorded after the halo G2P2G, so the device-side
        # order is unchanged, but the ~100 us of host time that torch spends in the collective no longer delay the launch
        # of the kernel the GPU is waiting for (0.49 -> 0.43 ms per substep at 5 M particles per rank).
        if overlap_with is not None:
            overlap_with()
        off = 0
        for p in range(w):
            c = self.send_counts[p]
            if c:
                base = send.data_ptr() + 4 * off
                ns = C.c_int(0)
                self._check(api.halo_collect(ctx, p, gid, C.c_void_p(base), C.c_void_p(base + 12 * c), c, C.byref(ns)))
                assert ns.value == c
            off += ROW * c
        with self._on(self._comm_stream):
            self.comm.all_to_all(recv, send, splits)  # symmetric exchange: recv counts == send counts
        off = 0
        for p in range(w):
            c = self.send_counts[p]
            if c:
                base = recv.data_ptr() + 4 * off
                self._check(api.halo_reduce(ctx, gid, C.c_void_p(base), C.c_void_p(base + 12 * c), c))
            off += ROW * c
        if total == 0:  # nothing shared: still order the compute stream behind the comm stream
            self._check(api.halo_reduce(ctx, gid, None, None, 0))

    # ---- one substep (mgsp_benchmark.cuh:361-559) -------------------------------------------------------------
    def _advance(self, dt, next_dt):
        """Phase-by-phase variant (one host synchronisation per phase, like the reference's issue()/sync())."""
        api, ctx = self.api, self.ctx
        self._check(api.g2p2g_halo(ctx, dt, next_dt))
        self._exchange(1, overlap_with=lambda: self._check(api.g2p2g_interior(ctx, dt, next_dt)))
        self._check(api.rebuild_partition(ctx, None))
        self._tag()

    def _key_buffers(self):
        if getattr(self, "_kb_pad", None) != self.pad:
            with self._on(self._compute_stream):
                self._kb_mine = torch.zeros((self.pad, 3), dtype=torch.int32, device=self.tdev)
                self._kb_all = torch.zeros((self.world * self.pad, 3), dtype=torch.int32, device=self.tdev)
            self._kb_pad = self.pad
        return self._kb_mine, self._kb_all

    def substep(self, dt, next_dt):
        """Fused variant: everything is enqueued, ONE host synchronisation at the end (mpm_mgsp_end)."""
        api, ctx = self.api, self.ctx
        self._check(api.mgsp_begin(ctx, dt, next_dt))
        self._exchange(1, overlap_with=lambda: self._check(api.g2p2g_interior(ctx, dt, next_dt)))
        pad = self.pad
        mine, allk = self._key_buffers()
        with self._on(self._compute_stream):
            self._check(api.mgsp_rebuild_export(ctx, C.c_void_p(mine.data_ptr()), pad))
            self.comm.all_gather(allk, mine)
        self._check(api.mgsp_tag(ctx, C.c_void_p(allk.data_ptr()), pad, self.world, self.rank))
        sc, nh, mx, mv = (C.c_int * 32)(), C.c_int(0), C.c_int(0), C.c_float(0)
        self._check(api.mgsp_end(ctx, sc, C.byref(nh), C.byref(mx), C.byref(mv)))
        if mx.value > pad:      # a key list was truncated (same verdict on every rank): tag again with a larger padding
            self.pad = int(1.25 * mx.value) + 64
            self._tag()
        else:
            if mx.value > 0.9 * pad:
                self.pad = int(1.25 * mx.value) + 64
            self.send_counts = [int(sc[p]) if p != self.rank else 0 for p in range(self.world)]
            self.n_halo_blocks = nh.value
        return mv.value

    def substep_phased(self, dt, next_dt):
        mv = C.c_float(0)
        self._check(self.api.grid_update(self.ctx, dt, C.byref(mv)))
        self._advance(dt, next_dt)
        return mv.value

    def run_fixed(self, nsteps, dt):
        for _ in range(nsteps):
            mv = self.substep(dt, dt)
            if not np.isfinite(mv):
                raise EngineError(_ffi.MPM_ERR_NONFINITE, "Maximum velocity is infinity")
            if self.hip:
                self._t["g2p2g"] += self.eng.last_g2p2g_ms()
            self._t["steps"] += 1

    def compute_dt_mgsp(self, max_vel, cur, nxt, dt_default):
        """compute_dt of the MGSP project (Projects/MGSP/utility_funcs.hpp:32-55): CFL 0.3 and the 0.51 frame-remainder
        rule, in float32 like the reference."""
        f = np.float32
        max_vel, cur, nxt, dt_default = f(max_vel), f(cur), f(nxt), f(dt_default)
        if nxt < cur:
            return 0.0
        dt = dt_default
        if max_vel > 0:
            lim = f(f(self.eng.dx) * f(0.3)) / max_vel
            if lim < dt_default:
                dt = lim
        if f(cur + dt) >= nxt:
            dt = f(nxt - cur)
        else:
            rest = f(f(nxt - cur) * f(0.51))
            if rest < dt:
                dt = rest
        return float(dt)

    def run_adaptive(self, nsteps, dt0, dt_default, frame_time=1.0 / 24.0):
        """Adaptive dt as MgspBenchmark::main_loop (mgsp_benchmark.cuh:375-418): the CFL limit uses the maximum velocity over
        all ranks, and the step clock restarts with every frame."""
        t, cur, dts = 0.0, dt0, []
        for _ in range(nsteps):
            assert cur > 0.0, "time step is not positive"
            api, ctx = self.api, self.ctx
            mv2 = C.c_float(0)
            self._check(api.grid_update(ctx, cur, C.byref(mv2)))
            m = torch.tensor([mv2.value], dtype=torch.float32, device=self.tdev)
            self.comm.all_reduce_max(m)   # host max over GPUs in the reference
            mv = float(np.sqrt(m.item()))
            nd = self.compute_dt_mgsp(mv, t + cur, frame_time, dt_default)
            if not nd > 0.0:  # the frame is complete: first dt of the next one
                nd = self.compute_dt_mgsp(mv, 0.0, frame_time, dt_default)
            self._advance(cur, nd)
            t += cur
            if t >= frame_time:
                t = 0.0
            dts.append(cur)
            cur = nd
        return dts

    # ---- reporting ---------------------------------------------------------------------------------------------
    @property
    def g2p2g_ms_avg(self):
        return self._t["g2p2g"] / max(self._t["steps"], 1)

    def phase_ms(self):
        return {"g2p2g_ms": self.g2p2g_ms_avg, "halo_blocks_sent": int(sum(self.send_counts)), "halo_particle_blocks": int(self.n_halo_blocks)}

    def block_counts(self):
        c = self.eng.counts()
        return {"particle": c.particle_blocks, "neighbor": c.neighbor_blocks, "exterior": c.exterior_blocks}

    def local_state(self):
        return [self.eng.retrieve_state(m) for m in range(len(self.eng.models))]

    def gather_state(self):
        """All particles of all ranks on every rank (tests): list per model of (xyz, state9, logjp)."""
        out = []
        for xyz, st, lj in self.local_state():
            objs = [None] * self.world
            dist.all_gather_object(objs, (xyz, st, lj))
            out.append(tuple(np.concatenate([o[i] for o in objs]) for i in range(3)))
        return out


class MgspGroupRank:
    """One rank of the C++ MGSP driver (claymore_amd/csrc/mpm_group.inc): the whole substep loop - grid update, halo-first
    G2P2G, collect / ncclSend+ncclRecv / reduce on the comm stream beside the interior G2P2G, rebuild, ncclAllGather of
    the block keys, tagging - runs inside the library with one host synchronisation per substep.  Python only carries
    the 128-byte RCCL unique id from rank 0 to the others (`bootstrap`: callable(bytes or None) -> bytes)."""

    def __init__(self, scene, rank, world, device=0, bootstrap=None, local_group=None, prepartitioned=False, api=None):
        """`scene` is the whole scene (cut here with partition_scene) or, with prepartitioned=True, this rank's share of a
        static particle partition the caller made (any partition is valid: the halo is whatever blocks the shares touch).
        `api`: another build of the HIP library bound with _ffi.bind(..., hip=True) (A/B tools, mutation tests); default: the shipped one."""
        self.rank, self.world = rank, world
        self.api = api if api is not None else _ffi.load_hip()
        local = scene if prepartitioned else partition_scene(scene, rank, world)
        self.n_local = scenes.total_particles(local)
        self.eng = build_engine(local, device=device, api=self.api)
        self.ctx = self.eng.ctx
        self.grp = C.c_void_p()
        if local_group is not None:      # in-process transport: the LocalGroup creates all handles at once
            local_group.register(rank, self)
        else:
            ident = (C.c_char * 128)()
            if rank == 0:
                self._check(self.api.group_unique_id(ident))
            raw = bootstrap(bytes(ident.raw) if rank == 0 else None) if bootstrap else bytes(ident.raw)
            ident = (C.c_char * 128).from_buffer_copy(raw)
            self._check(self.api.group_create(self.ctx, rank, world, ident, C.byref(self.grp)))

    def _check(self, rc):
        if rc != 0:
            msg = self.api.group_last_error(self.grp).decode() if self.grp else self.api.last_error(self.ctx).decode()
            raise EngineError(rc, msg or self.api.last_error(self.ctx).decode())

    def initial_setup(self):
        self._check(self.api.group_initial_setup(self.grp))

    def save_checkpoint(self):
        """This rank's full state at a substep boundary (mpm_checkpoint_save of its context)."""
        return self.eng.save_checkpoint()

    def load_checkpoint(self, buf):
        """Restart: load this rank's checkpoint, then rebuild the tagging with the other ranks (mpm_group_resume: collective)."""
        self.eng.load_checkpoint(buf)
        self._check(self.api.group_resume(self.grp))

    def substep(self, dt, next_dt):
        mv = C.c_float(0)
        self._check(self.api.group_substep(self.grp, dt, next_dt, C.byref(mv)))
        return mv.value

    def run_fixed(self, nsteps, dt):
        self._check(self.api.group_run_fixed(self.grp, int(nsteps), dt))

    def main_loop(self, frames, fps, dt_default):
        n = C.c_int(0)
        self._check(self.api.group_main_loop(self.grp, int(frames), int(fps), dt_default, None, None, C.byref(n)))
        return n.value

    def stats(self):
        sc, nh, ms = (C.c_int * 32)(), C.c_int(0), C.c_float(0)
        self._check(self.api.group_stats(self.grp, sc, C.byref(nh), C.byref(ms)))
        return [int(x) for x in sc], nh.value, ms.value

    @property
    def send_counts(self):
        return self.stats()[0]

    @property
    def n_halo_blocks(self):
        return self.stats()[1]

    @property
    def g2p2g_ms_avg(self):
        return self.stats()[2]

    def phase_ms(self):
        sc, nh, ms = self.stats()
        return {"g2p2g_ms": ms, "halo_blocks_sent": int(sum(sc)), "halo_particle_blocks": int(nh)}

    def block_counts(self):
        c = self.eng.counts()
        return {"particle": c.particle_blocks, "neighbor": c.neighbor_blocks, "exterior": c.exterior_blocks}

    def local_state(self):
        return [self.eng.retrieve_state(m) for m in range(len(self.eng.models))]

    def close(self):
        if self.grp:
            self.api.group_destroy(self.grp)
            self.grp = C.c_void_p()
        self.eng.close()


class LocalGroup:
    """All ranks in one process (one thread each) on the library's in-process transport: contexts may share a GPU."""

    def __init__(self, world):
        self.world, self.ranks = world, [None] * world

    def register(self, rank, r):
        self.ranks[rank] = r

    def create(self):
        api = self.ranks[0].api
        ctxs = (C.c_void_p * self.world)(*[r.ctx for r in self.ranks])
        out = (C.c_void_p * self.world)()
        rc = api.group_create_local(ctxs, self.world, out)
        if rc != 0:
            raise EngineError(rc, "mpm_group_create_local failed")
        for r, g in zip(self.ranks, out):
            r.grp = C.c_void_p(g)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
