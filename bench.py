#!/usr/bin/env python
"""bench.py - BASELINE.json's metric on its own config.

    python bench.py --gpus N --steps K --warmup W

A "step" is one MPM substep (grid update -> fused G2P2G -> sparse partition rebuild [-> halo exchange for
N > 1]) over one synthetic scene.  N = 1 runs BASELINE config 3, the configuration the metric is quoted on:
the 40.1 M-particle Drucker-Prager sand column on a 512^3 sparse grid (`configs[2]`).  N > 1 is MGSP (static particle
partition, one rank per GPU, the substep loop and the halo exchange in the C++ driver on RCCL) on the SAME workload - STRONG
scaling, as BASELINE's metric states ("ms/step at 40 M particles, 512^3, 1/2/4/8 MI355X"; reference docs/benchmark.rst:43-48):
the one C3 column cut into N equal-count slabs along its longest axis.  `--scaling weak` is the other experiment (N touching C3
columns on the same grid, one per rank, N x 40.1 M particles).  Inputs are resident in HBM before the timed region;
value = particles of all ranks * K / time.

Extra objects on the JSON line:
  roofline     - G2P2G (the dominant kernel): algorithmic bytes per launch (144 B/particle
                 for sand) / average kernel duration from HIP events on the engine's compute stream, vs 8 TB/s.  For the default
                 run (N = 1, C3) the HEADLINE roofline.frac is that of a second window INSIDE THE FLOW (20 substeps after
                 --flow-start: the column is collapsing - block churn, plastic return mapping, three Jacobi sweeps); the timed K
                 substeps themselves (the column in free fall, the kernel's cheapest regime) are roofline.rest.  `traffic` and
                 `valu_executed` come from the PMC profile of the MATCHING window (profiles/rNN_pmc.json) and are attached only
                 when that file's stamp is the sha256 of the library this run loaded; physical_frac = those counter bytes /
                 this run's kernel time / 8 TB/s (what the memory system really moved, beside the algorithmic figure).
                 NOTE: `value` and `ms_per_step` are ALWAYS the timed K substeps; when `headline_window` is "flow", roofline.frac /
                 kernel_ms describe ANOTHER window of the same run - the pair that belongs to `value` is roofline.rest
                 (roofline.timed_frac repeats its fraction); roofline.deep is a third window after --deep-start substeps (default 9000: deep in the collapse)
  cpu_baseline - the CPU oracle ("port" of the reference pipeline) timed on this host on the same input, outside the timed
                 region (rank 0, N = 1 only): OpenMP over G2P2G's particle blocks, the grid update and the rebuild's
                 order-independent loops; `cores` = the threads it ran on
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BYTES_PER_PARTICLE = {0: 72, 1: 136, 2: 144, 3: 144}  # BASELINE.md section 4 / SURVEY.md 8(d)
# SURVEY.md 8(d) "ALGORITHMIC FLOPs (secondary)": the reference formulation (27-node G2P ~1.0 k, P2G ~0.75 k, F update
# ~0.1 k, SVD ~0.8 k, stress 0.1-0.2 k); this engine executes fewer (tensor-product gather, eigen-decomposition)
FLOPS_PER_PARTICLE = {0: 1800, 1: 2700, 2: 2700, 3: 2700}
def _pair_kernel(material):
    """Which G2P2G instantiation the library launches for a material (claymore_hip.hip: MPM_PAIR_DEFAULT 0xF - two particles per lane for all
    four materials -, overridden by the environment's MPM_G2P2G_PAIRS)."""
    return bool((int(os.environ.get("MPM_G2P2G_PAIRS", "0xF"), 0) >> int(material)) & 1)


HBM_PEAK_GBS = 8000.0                                  # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3                        # MI355X_MICROARCH.md: peak FP32 (vector)


def make_scene(args):
    from claymore_amd import scenes
    if args.scene == "sand40m":
        sc = scenes.sand_column(9) if args.fraction >= 1.0 else scenes.scaled_sand_column(9, args.fraction)
        name = "C3 sand column collapse (Drucker-Prager), 512^3 sparse grid"
    elif args.scene == "sphere5m":
        sc = scenes.sphere_drop()
        name = "C2 elastic sphere drop (fixed-corotated), 256^3 sparse grid"
    elif args.scene == "spheres50k":
        sc = scenes.two_spheres()
        name = "C1 two elastic spheres, 128^3 grid"
    elif args.scene == "fluid12m":
        sc = scenes.fluid_dam(10, (32, 192, 256))
        name = "C5 per-rank share (1/8): weakly compressible J-fluid slab 32x192x256 cells, 1024^3 sparse grid"
    elif args.scene == "spheres40m":
        sc = scenes.two_spheres_c4()
        name = "C4 on one GPU: two fixed-corotated spheres of R = 84 dx (2 x 19.9 M particles), 512^3 sparse grid"
    elif args.scene == "fluid100m":
        sc = scenes.fluid_dam(10)
        name = "C5 on one GPU: weakly compressible J-fluid box 256x192x256 cells (100.7 M particles), 1024^3 sparse grid"
    else:
        raise SystemExit(f"unknown scene {args.scene}")
    return sc, name


def _free_gb():
    try:
        import psutil
        return psutil.virtual_memory().available / 2 ** 30
    except Exception:  # noqa: BLE001
        pass
    try:
        with open("/proc/meminfo") as f:
            for ln in f:
                if ln.startswith("MemAvailable:"):
                    return int(ln.split()[1]) / 2 ** 20
    except Exception:  # noqa: BLE001
        pass
    return 0.0


def cpu_baseline(args, scene=None):
    """The CPU oracle (a port of the reference pipeline, oracle/mpm_oracle.c) timed on this host, outside the timed region of the GPU run:
      * on the SAME input as the GPU run - for the default scene the full C3 column (40.1 M particles; set-up ~40 s serial, 11 GB, then
        one warm + five timed substeps; OpenMP over particle blocks in G2P2G, over grid blocks in the grid update and over the
        order-independent loops of the rebuild - fills, per-block copies; the inserts that define the block numbering stay serial), and
      * single-threaded on the full C1 scene (BASELINE config 1: the reference's own CPU-runnable case).
    A host with less than 24 GB of free memory falls back to a 1/64-scale column (same aspect ratio, same grid) and says so."""
    import __graft_entry__ as g
    g.build_oracle()
    from claymore_amd import scenes
    from claymore_amd.engine import build_engine
    from oracle_ffi import oracle_api
    api = oracle_api()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(cores, 64))

    def timed(sc, nthreads, steps, warm=1):
        eng = build_engine(sc, api=api)
        t0 = time.perf_counter()
        eng.initial_setup()
        setup = time.perf_counter() - t0
        api.raw.mpmo_set_threads(eng.ctx, nthreads)
        eng.run_fixed(warm, sc["dt"])
        t0 = time.perf_counter()
        eng.run_fixed(steps, sc["dt"])
        dt = time.perf_counter() - t0
        tm = eng.timers()                     # (the oracle's own clocks, last substep)
        eng.close()
        return scenes.total_particles(sc) * steps / dt, dt, setup, {"grid_update_ms": tm.grid_update_ms, "g2p2g_ms": tm.g2p2g_ms, "partition_ms": tm.partition_ms}

    free_gb = _free_gb()
    same = False
    if args.scene == "sand40m":
        full = args.fraction >= 1.0 and free_gb >= 24.0 and scene is not None
        same = full
        sc = scene if full else scenes.scaled_sand_column(9, 1.0 / 64.0)
        steps = 5 if full else 19
        what = (f"the full C3 scene ({scenes.total_particles(sc)} particles, 512^3): the same input as the GPU run" if full else
                f"sand column scaled to {scenes.total_particles(sc)} particles (1/64 of C3, same aspect ratio, 512^3 grid: NOT the GPU run's input; the host has {free_gb:.0f} GB free, the full scene needs 24)")
    elif args.scene == "sphere5m":
        sc, steps = (scene if scene is not None else scenes.sphere_drop()), 5
        same = scene is not None
        what = f"the full C2 scene ({scenes.total_particles(sc)} particles, 256^3)" + (": the same input as the GPU run" if same else "")
    else:
        sc, steps = scenes.two_spheres(), 20
        what = "the full C1 scene"
    rate, dt, setup, last = timed(sc, threads, steps)
    out = {"value": rate, "unit": "particles*steps/s", "cores": threads, "threads": threads, "host_cores": cores, "kind": "port", "same_input": same,
           "timed_substeps": steps, "last_substep_ms": last,
           "sample": f"{what}, {steps} substeps after one warm-up substep, C oracle (oracle/mpm_oracle.c) on {threads} OpenMP threads (G2P2G over particle "
                     f"blocks, grid update over grid blocks, the rebuild's fills and per-block copies; its numbering inserts are serial), {dt:.1f} s "
                     f"(+ {setup:.0f} s of serial set-up, not counted)"}
    c1 = scenes.two_spheres()
    r1, t1, _, _ = timed(c1, 1, 100, warm=2)
    out["c1_single_thread"] = {"value": r1, "unit": "particles*steps/s", "cores": 1,
                               "sample": f"the full C1 scene (BASELINE config 1: two elastic spheres, {scenes.total_particles(c1)} particles, 128^3), 100 substeps, one thread, {t1:.1f} s"}
    return out


def loaded_library_sha16():
    """First 16 hex digits of the sha256 of the engine library this process loads (what tools/stamp.sh writes into every profile)."""
    import hashlib
    from claymore_amd import _ffi
    h = hashlib.sha256()
    with open(_ffi.HIP_LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def attach_counters(dst, pmc, pmc_name, window, lib_sha, kernel_ms, n_rank, bpp):
    """Counter fields of one window (`rest` / `flow`) of the profiled kernel -> dst, ONLY if the profile was taken with the library this run
    loaded (the stamp of profiles/rNN_pmc.json carries its sha256): `traffic` (+ `traffic_range` for the flow window), `physical_frac` =
    those bytes over THIS run's kernel time over the HBM peak, `valu_executed`.  Another build's counters are refused, and said so."""
    import re
    w = pmc.get(window)
    if not w:
        return
    m_ = re.search(r"sha256 ([0-9a-f]{16})", pmc.get("stamp", ""))
    pmc_sha = m_.group(1) if m_ else None
    dst["algorithmic_bytes"] = n_rank * bpp
    if pmc_sha != lib_sha:
        dst["traffic"] = None
        dst["traffic_source"] = (f"REFUSED: {pmc_name} was taken with library sha256 {pmc_sha} ({pmc.get('stamp', 'unstamped')}), this run loaded {lib_sha}: "
                                 "counters of another build are not attached (re-take the PMC passes: tools/gpu_profile_r05.sh)")
        return
    dst["traffic"] = w["traffic_bytes"]
    dst["traffic_source"] = f"{pmc_name}[{window}] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of {pmc.get('kernel', 'the G2P2G kernel')}, corrected as MI355X_MICROARCH.md prescribes; {pmc.get('stamp', 'unstamped')}: the library this run loaded)"
    if kernel_ms > 0:
        # what the memory system moved per launch (counters of the profiled launch) over THIS run's kernel time: the physical HBM rate
        dst["physical_frac"] = w["traffic_bytes"] / (kernel_ms * 1e-3) / (HBM_PEAK_GBS * 1e9)
    if window == "flow" and "traffic_bytes_low" in w:
        # the fetch calibration is that of a streaming kernel; the flow window reads scattered 32-B records, whose requests the counter tallies in full
        dst["traffic_range"] = [w["traffic_bytes_low"], w["traffic_bytes"]]
        if kernel_ms > 0:
            dst["physical_frac_range"] = [w["traffic_bytes_low"] / (kernel_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), dst["physical_frac"]]
    if w.get("valu_insts"):
        # one wave-instruction serves 64 particles: instructions per particle (= per 64-particle iteration of a wave)
        per_particle = w["valu_insts"] * 64.0 / n_rank
        # issue-busy: wave-instructions x the measured cost of this kernel's mix (profiles/r03_energy_model.txt: VOP2 2.07, VOP3 2.16,
        # packed 4.26 cycles; ~2.75 on average) / SIMD cycles available at the sclk the launch runs at (1.97 GHz: it is power-limited)
        busy = w["valu_insts"] * 2.75 / (1024 * kernel_ms * 1e-3 * 1.97e9) if kernel_ms > 0 else 0.0
        dst["valu_executed"] = {"wave_instructions_per_launch": w["valu_insts"], "executed_per_particle": per_particle, "issue_busy": busy,
                                "note": "SQ_INSTS_VALU of the profiled launch; issue_busy prices an instruction at 2.75 cycles of a 1.97 GHz SIMD (measured mix and clock)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # warm-up 10 substeps, time the next 100
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scene", default="sand40m", choices=["sand40m", "sphere5m", "spheres50k", "fluid12m", "spheres40m", "fluid100m"])
    ap.add_argument("--start-step", type=int, default=0,
                    help="untimed substeps before the warm-up: moves the timed window into the flow (the default window of C3 "
                         "starts at rest; after ~3000 substeps the column is collapsing: block churn, mispredicted sort keys)")
    ap.add_argument("--fraction", type=float, default=1.0, help="debug: shrink the sand column")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mgsp", action="store_true", help="debug: drive the multi-GPU code path even with one rank")
    ap.add_argument("--max-ppc", type=int, default=0, help="debug: override the scene's particles-per-cell capacity (reference: 128)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong (BASELINE's metric) = the one C3 column cut N ways, weak = one C3 column per rank (N x 40.1 M particles)")
    ap.add_argument("--partition", default="longest-axis", choices=["longest-axis", "y", "x", "z", "octants", "xz-columns", "y-x"],
                    help="--gpus N > 1, strong scaling: the shape of the static particle partition (claymore_amd.scenes.PARTITION_SHAPES; "
                         "profiles/r06_mgsp_partition.txt: slabs along the longest axis are the best cut while the column stands, x slabs once it has collapsed)")
    ap.add_argument("--equal-count", action="store_true",
                    help="--gpus N > 1, strong scaling: cut the column into exactly equal particle counts instead of moving the cut planes to the nearest "
                         "particle-block faces (default: block-aligned pieces within 10 %% of the equal share - no block is shared by two ranks at the start: "
                         "C3 cut 8 ways along y is 10 890 instead of 11 979 particle blocks on the largest rank)")
    ap.add_argument("--flow-start", type=int, default=3000,
                    help="N = 1, default scene: after the timed window the run goes on to this substep and a second short window is timed "
                         "inside the flow (reported as roofline.flow; 0 = skip)")
    ap.add_argument("--deep-start", type=int, default=9000,
                    help="default run (N = 1, C3): a third 20-substep window after this many substeps, deep in the collapse -> roofline.deep (0 = skip; adds ~15 s)")
    ap.add_argument("--sync-interval", type=int, default=0, help="debug: mpm_config.sync_interval (0 = library default)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="debug: rank r uses device r %% (devices present) instead of device LOCAL_RANK - several ranks per GPU.  RCCL refuses "
                         "that; with a stand-in collective library (MPM_RCCL_LIBRARY, tests/rccl_double/rccl_double_mp.cpp) it is how the "
                         "N-process launch is exercised on a one-GPU box.  Never a measurement.")
    ap.add_argument("--watchdog", type=float, default=900.0,
                    help="seconds after which a run that has not finished prints a JSON error line and exits (a rank stuck in a collective "
                         "would otherwise sit there until the caller's own timeout); 0 = off")
    ap.add_argument("--baseline-timeout", type=float, default=420.0,
                    help="seconds the CPU baseline may take after the GPU measurement is complete; beyond it the record is emitted with "
                         "cpu_baseline.error instead of a value (the finished GPU numbers are never lost to a slow host)")
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON record: libraries that print banners there (gloo's "connected to N peer ranks",
    # RCCL's version block) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # (the host driver of this pool supports dmabuf IPC only: without this RCCL's cross-process buffer sharing fails in hipIpcGetMemHandle;
    #  the harness exports it - kept here for a launch from a bare shell)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    # ---- watchdog: one JSON line whatever happens.  `stage` says where the run was; the thread fires once, from any rank (a rank that
    #      hangs in a collective cannot report for itself: its peers, which wait for it, do)
    import threading
    stage = {"at": "start", "ranks_seen": None}

    def error_line(reason):
        rec = {"metric": "particles*steps/sec", "value": None, "unit": "particles*steps/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
               "error": reason, "stage": stage["at"], "rank": rank, "ranks_seen": stage["ranks_seen"]}
        # stdout carries ONE record: rank 0's.  The other ranks' records go to stderr (with N ranks failing together - which is what an
        # engine error does: every rank returns it in the same substep - stdout would otherwise hold N lines)
        os.write(json_fd if rank == 0 else 2, (json.dumps(rec) + "\n").encode())

    def watchdog():
        if not finished.wait(args.watchdog):
            sys.stderr.write(f"bench.py[rank {rank}]: watchdog after {args.watchdog:.0f} s at stage '{stage['at']}'\n")
            error_line(f"watchdog: no result after {args.watchdog:.0f} s")
            os._exit(3)

    finished = threading.Event()
    if args.watchdog > 0:
        threading.Thread(target=watchdog, daemon=True).start()
    prev_hook = sys.excepthook

    def hook(tp, val, tb):          # an exception (an engine error that every rank returns, a failed self-check) also leaves a JSON line
        finished.set()
        error_line(f"{tp.__name__}: {val}")
        prev_hook(tp, val, tb)

    sys.excepthook = hook
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    if args.oversubscribe:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    use_mgsp = world > 1 or args.mgsp
    if use_mgsp:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
        # host-side rendez-vous only (RCCL unique id, barriers, the max over ranks of the elapsed time): the data path is RCCL
        # inside the library (claymore_amd/csrc/mpm_group.inc), one communicator per rank on the engine's own streams
        stage["at"] = "gloo rendez-vous"
        os.environ.setdefault("MPM_GROUP_VERBOSE", "1")   # the library logs which collective library it loaded and ncclCommCount
        dist.init_process_group("gloo")
        seen = [None] * dist.get_world_size()
        dist.all_gather_object(seen, (rank, local_rank, os.uname().nodename))
        stage["ranks_seen"] = len(seen)
        if rank == 0:
            sys.stderr.write(f"bench.py: gloo rendez-vous complete, {len(seen)} ranks: {seen}\n")

    import __graft_entry__ as g
    if rank == 0:
        g.build_hip()
    if use_mgsp:
        dist.barrier()

    from claymore_amd import scenes
    weak = world > 1 and args.scaling == "weak"
    if weak:
        if args.scene != "sand40m" or args.fraction < 1.0:
            raise SystemExit("--scaling weak is defined for the default scene (one C3 column per rank)")
        sc = scenes.sand_columns_rank(rank, world)          # this rank's column only
        n_total = world * scenes.total_particles(sc)
        workload = f"{world} touching C3 sand columns (Drucker-Prager), one per rank, 512^3 sparse grid"
    else:
        sc, workload = make_scene(args)
        n_total = scenes.total_particles(sc)
    if args.max_ppc:
        sc["config"]["max_ppc"] = args.max_ppc
    if args.sync_interval:
        sc["config"]["sync_interval"] = args.sync_interval
    material = sc["models"][0]["material"]
    dt = sc["dt"]

    if not use_mgsp:
        from claymore_amd.engine import build_engine
        eng = build_engine(sc, device=local_rank)
        eng.initial_setup()
        if args.start_step:
            eng.run_fixed(args.start_step, dt)
        eng.run_fixed(args.warmup, dt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run_fixed(args.steps, dt)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        tm = eng.timers()
        g2p2g_ms = tm.g2p2g_ms
        phases = {"grid_update_ms": tm.grid_update_ms, "g2p2g_ms": tm.g2p2g_ms, "partition_ms": tm.partition_ms,
                  "device_total_ms": tm.total_ms}
        cnt = eng.counts()
        blocks = {"particle": cnt.particle_blocks, "neighbor": cnt.neighbor_blocks, "exterior": cnt.exterior_blocks}
        n_rank = n_total
        # self-check (gmpm_simulator.cuh:617 prints the particle total; the reference loses particles silently): nothing
        # was lost, no P2G contribution was discarded, the grid carries the whole mass
        diag = eng.diagnostics()
        bucketed = sum(cnt.particles[i] for i in range(cnt.model_count))
        totals = eng.grid_totals()
        mass_expected = sum(eng.model_mass(i) * m["xyz"].shape[0] for i, m in enumerate(sc["models"]))
        check = {"particles_bucketed": int(bucketed), "lost_particles": int(diag.lost_particles),
                 "discarded_p2g": int(diag.discarded_p2g), "grid_mass_rel_err": abs(totals[0] - mass_expected) / mass_expected}
        assert bucketed == n_total, check
        assert diag.lost_particles == 0 and diag.discarded_p2g == 0, check
        assert check["grid_mass_rel_err"] < 1e-4 and np.isfinite(totals).all(), check
        flow = None
        done = args.start_step + args.warmup + args.steps
        if args.scene == "sand40m" and args.fraction >= 1.0 and not args.start_step and args.flow_start > done:
            # second window, inside the flow: the column is collapsing (block churn, mispredicted sort keys, plastic return mapping)
            eng.run_fixed(args.flow_start - done, dt)
            eng.run_fixed(5, dt)
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            eng.run_fixed(20, dt)
            torch.cuda.synchronize()
            tf = time.perf_counter() - tf0
            tmf = eng.timers()
            df = eng.diagnostics()
            cf = eng.counts()
            assert sum(cf.particles[i] for i in range(cf.model_count)) == n_total and df.lost_particles == 0, "flow window lost particles"
            flow = {"start_step": args.flow_start + 5, "steps": 20, "kernel_ms": tmf.g2p2g_ms, "ms_per_step": 1e3 * tf / 20,
                    "blocks": {"particle": cf.particle_blocks, "neighbor": cf.neighbor_blocks, "exterior": cf.exterior_blocks}}
            if args.deep_start > args.flow_start + 25:
                # third window, deep in the collapse (the flow window is not the worst of the configured workload): the pile spreads
                # against the walls, single cells hold hundreds of particles
                eng.run_fixed(args.deep_start - (args.flow_start + 25), dt)
                eng.run_fixed(5, dt)
                torch.cuda.synchronize()
                td0 = time.perf_counter()
                eng.run_fixed(20, dt)
                torch.cuda.synchronize()
                td = time.perf_counter() - td0
                tmd, dd, cd = eng.timers(), eng.diagnostics(), eng.counts()
                assert sum(cd.particles[i] for i in range(cd.model_count)) == n_total and dd.lost_particles == 0, "deep window lost particles"
                flow["deep"] = {"start_step": args.deep_start + 5, "steps": 20, "kernel_ms": tmd.g2p2g_ms, "ms_per_step": 1e3 * td / 20,
                                "blocks": {"particle": cd.particle_blocks, "neighbor": cd.neighbor_blocks, "exterior": cd.exterior_blocks}}
        eng.close()
    else:
        from claymore_amd.mgsp import MgspGroupRank

        def bootstrap(raw):
            objs = [raw]
            dist.broadcast_object_list(objs, src=0)
            return objs[0]

        stage["at"] = "ncclCommInitRank"
        from claymore_amd import mgsp as _mgsp
        if args.partition != "longest-axis":
            _mgsp.PARTITION_SHAPE = args.partition
        _mgsp.ALIGN_TO_BLOCKS = not args.equal_count   # (strong scaling: the cut planes are particle-block faces; weak scaling is prepartitioned)
        sim = MgspGroupRank(sc, rank, world, device=local_rank, bootstrap=bootstrap, prepartitioned=weak)
        if rank == 0:
            sys.stderr.write(f"bench.py: RCCL communicator up, world {world} (this rank {rank})\n")
        stage["at"] = "mgsp initial setup (first tagging, first halo exchange)"
        sim.initial_setup()
        stage["at"] = "mgsp substeps"
        if args.start_step:
            sim.run_fixed(args.start_step, dt)
        sim.run_fixed(args.warmup, dt)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.run_fixed(args.steps, dt)
        torch.cuda.synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        g2p2g_ms = sim.g2p2g_ms_avg
        phases = sim.phase_ms()
        blocks = sim.block_counts()
        n_rank = sim.n_local
        # self-check over all ranks: every particle is still bucketed, nothing was lost or discarded
        d = sim.eng.diagnostics()
        c = sim.eng.counts()
        tot = torch.tensor([float(sum(c.particles[i] for i in range(c.model_count))), float(d.lost_particles), float(d.discarded_p2g)], dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        check = {"particles_bucketed": int(tot[0].item()), "lost_particles": int(tot[1].item()), "discarded_p2g": int(tot[2].item())}
        assert check["particles_bucketed"] == n_total and check["lost_particles"] == 0 and check["discarded_p2g"] == 0, check
        flow = None
        sim.close()

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = n_total * args.steps / elapsed
        bpp = BYTES_PER_PARTICLE[material]
        achieved = (n_rank * bpp) / (g2p2g_ms * 1e-3) / 1e9 if g2p2g_ms > 0 else 0.0
        out = {
            "metric": "particles*steps/sec", "value": value, "unit": "particles*steps/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "particles": n_total, "dt": dt,
                       "parallelism": "single GPU" if not use_mgsp else
                       f"mgsp static particle partition x{world} ({'one column per rank' if weak else ('equal-count' if args.equal_count else 'block-aligned') + ' pieces of the one column, shape ' + (args.partition if args.partition != 'longest-axis' else 'slabs along the longest axis (y)')}), C++ driver on RCCL",
                       "blocks": blocks, "phases_ms": phases, **({"oversubscribed": "ranks share GPUs (debug launch, not a measurement)"} if args.oversubscribe else {})},
            "headline_window": "timed",   # which window roofline.frac / kernel_ms describe: "flow" once the flow window has run (below); `value` / ms_per_step are ALWAYS the timed K substeps (= roofline.rest then)
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "g2p2g_pair_kernel" if _pair_kernel(material) else "g2p2g_kernel", "bytes_per_particle": bpp, "particles_per_launch": n_rank,
                         "kernel_ms": g2p2g_ms},
        }
        # the kernel's arithmetic intensity sits at the fp32-vector ridge, so the VALU side is reported too: the
        # algorithmic figure of the reference formulation, and - from the PMC profile of the matching window - what this engine executes
        fpp = FLOPS_PER_PARTICLE[material]
        tfl = (n_rank * fpp) / (g2p2g_ms * 1e-3) / 1e12 if g2p2g_ms > 0 else 0.0
        out["roofline"]["valu"] = {"flops_per_particle": fpp, "achieved_tflops": tfl, "peak": FP32_VECTOR_PEAK_TFLOPS,
                                   "frac": tfl / FP32_VECTOR_PEAK_TFLOPS,
                                   "note": "algorithmic FLOPs of the reference formulation (SURVEY.md 8d), not executed instructions"}
        if args.start_step:
            out["config"]["start_step"] = args.start_step
        if check is not None:
            out["config"]["self_check"] = check
        # measured HBM traffic and executed instructions of the same kernel on the same workload and WINDOW (PMC passes are separate
        # runs, see profiles/): "rest" = the default window, "flow" = a window that starts after >= 2000 substeps.  The counters describe
        # ONE build of the kernel: they are attached only when the file's stamp carries the sha256 of the library this run loaded.
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))      # the newest round's counter passes
        tf = cands[-1] if cands else ""
        pmc_name = os.path.join("profiles", os.path.basename(tf)) if tf else "profiles/(none)"
        pmc = json.load(open(tf)) if tf else {}
        c3 = world == 1 and args.scene == "sand40m" and args.fraction >= 1.0
        lib_sha = loaded_library_sha16()
        out["library_sha256_16"] = lib_sha

        def attach(dst, window, kernel_ms):
            if c3:
                attach_counters(dst, pmc, pmc_name, window, lib_sha, kernel_ms, n_rank, bpp)

        timed_window = "flow" if args.start_step >= 2000 else "rest"
        out["roofline"]["window"] = f"the timed K substeps ({'inside the flow' if timed_window == 'flow' else 'substeps ' + str(args.start_step + args.warmup) + '-' + str(args.start_step + args.warmup + args.steps)})"
        attach(out["roofline"], timed_window, g2p2g_ms)
        if flow:
            # The headline is the FLOW window (the timed K substeps of the default run are the column in free fall - every block
            # settled, no Jacobi sweep, no plastic branch: the kernel's cheapest regime); the timed window moves to roofline.rest.
            rest = dict(out["roofline"])
            fa = (n_rank * bpp) / (flow["kernel_ms"] * 1e-3) / 1e9
            head = {"bound": "hbm", "achieved": fa, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fa / HBM_PEAK_GBS, "traffic": None,
                    "kernel": "g2p2g_pair_kernel" if _pair_kernel(material) else "g2p2g_kernel", "bytes_per_particle": bpp, "particles_per_launch": n_rank, "kernel_ms": flow["kernel_ms"],
                    "window": f"flow: substeps {flow['start_step']}-{flow['start_step'] + flow['steps']} of the same run (the column is collapsing); the timed K substeps are roofline.rest",
                    "start_step": flow["start_step"], "steps": flow["steps"], "ms_per_step": flow["ms_per_step"], "blocks": flow["blocks"]}
            tflf = (n_rank * fpp) / (flow["kernel_ms"] * 1e-3) / 1e12
            head["valu"] = {"flops_per_particle": fpp, "achieved_tflops": tflf, "peak": FP32_VECTOR_PEAK_TFLOPS, "frac": tflf / FP32_VECTOR_PEAK_TFLOPS,
                            "note": "algorithmic FLOPs of the reference formulation (SURVEY.md 8d), not executed instructions"}
            attach(head, "flow", flow["kernel_ms"])
            if "deep" in flow:
                dp = flow["deep"]
                da = (n_rank * bpp) / (dp["kernel_ms"] * 1e-3) / 1e9
                head["deep"] = {"bound": "hbm", "achieved": da, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": da / HBM_PEAK_GBS, "kernel_ms": dp["kernel_ms"],
                                "window": f"deep: substeps {dp['start_step']}-{dp['start_step'] + dp['steps']} of the same run (the pile spreads against the walls)",
                                "ms_per_step": dp["ms_per_step"], "blocks": dp["blocks"]}
            head["rest"] = rest
            head["timed_frac"] = rest["frac"]      # the fraction that belongs to `value` / ms_per_step (a consumer must not pair `value` with roofline.frac of another window)
            out["roofline"] = head
            out["headline_window"] = "flow"

        def emit():
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())

        if world == 1 and not args.no_cpu_baseline and not args.mgsp:
            # The GPU measurement is complete: from here on nothing may lose it.  The baseline runs in a worker with its own time limit; the
            # watchdog is disarmed (it guards the GPU part: a rank stuck in a collective).
            stage["at"] = "cpu baseline (oracle on the host cores)"
            finished.set()
            box = {}

            def work():
                try:
                    box["value"] = cpu_baseline(args, sc)
                except BaseException as e:  # noqa: BLE001
                    box["error"] = f"{type(e).__name__}: {e}"

            th = threading.Thread(target=work, daemon=True)
            th.start()
            th.join(args.baseline_timeout if args.baseline_timeout > 0 else None)
            if "value" in box:
                out["cpu_baseline"] = box["value"]
            else:
                out["cpu_baseline"] = {"value": None, "unit": "particles*steps/s", "kind": "port",
                                       "error": box.get("error", f"not finished after {args.baseline_timeout:.0f} s (--baseline-timeout); the GPU record above is complete")}
            emit()
            if th.is_alive():
                os._exit(0)       # (the worker cannot be cancelled: leave with the record written)
        else:
            emit()
    finished.set()
    if use_mgsp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
