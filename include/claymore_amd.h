/*
 * claymore_amd.h — C ABI of the MI355X-native MPM substep engine (libclaymore_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path of penn-graphics-research/claymore:
 *     grid update -> fused G2P2G -> sparse block-partition rebuild      (Projects/GMPM)
 *     + static-particle-partition halo exchange                          (Projects/MGSP)
 * The reference has no FFI layer: its kernels are C++ templates launched through
 * Cuda::CudaContext::compute_launch (Library/MnSystem/Cuda/Cuda.h:151-184) from the two host classes
 * GmpmSimulator (Projects/GMPM/gmpm_simulator.cuh) and MgspBenchmark (Projects/MGSP/mgsp_benchmark.cuh).
 * Each entry point below replaces one phase of those classes; the file:line it replaces is cited.
 *
 * Conventions
 *   - plain C, POD structs, caller-owned host arrays are copied during the call, the context owns
 *     every device buffer (reference: members of GmpmSimulator, gmpm_simulator.cuh:96-141);
 *   - every function returns an mpm_status (0 = ok); mpm_last_error() gives the text.  The reference
 *     prints and exit()s (Library/MnSystem/Cuda/HostUtils.hpp:33-44) or abort()s on capacity overflow
 *     (gmpm_simulator.cuh:473-476); host drivers map a non-zero status to that behaviour;
 *   - a context is bound to one HIP device and is single-thread-affine (reference: one worker thread
 *     per GPU, mgsp_benchmark.cuh:309-334);
 *   - there is NO CPU fallback behind this ABI: if no HIP device is usable, mpm_create fails.
 *
 * The CPU oracle (oracle/, test infrastructure only) exports the same functions with the prefix
 * mpmo_ instead of mpm_ so that tests can drive both through identical call sequences.
 */
#ifndef CLAYMORE_AMD_H
#define CLAYMORE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mpm_status {
	MPM_OK			   = 0,
	MPM_ERR_INVALID	   = 1, /* bad argument / call order */
	MPM_ERR_DEVICE	   = 2, /* HIP runtime error (reference: check_cuda_errors -> exit) */
	MPM_ERR_CAPACITY   = 3, /* block / bin / cell capacity exceeded (reference: std::abort) */
	MPM_ERR_NONFINITE  = 4, /* inf/NaN grid velocity (reference: stops main loop, gmpm_simulator.cuh:355-358) */
	MPM_ERR_NOT_READY  = 5,
	MPM_ERR_INTERNAL   = 6 /* the library's own books do not balance: particles bucketed + lost + dropped != particles added (the reference
							  only prints the total per frame, gmpm_simulator.cuh:617); the message names the model */
} mpm_status;

/* Projects/GMPM/settings.h:20-26 */
typedef enum mpm_material {
	MPM_J_FLUID			= 0,
	MPM_FIXED_COROTATED = 1,
	MPM_SAND			= 2,
	MPM_NACC			= 3
} mpm_material;

/* Runtime replacement of the compile-time config:: constants (Projects/GMPM/settings.h:33-96). */
typedef struct mpm_config {
	int domain_bits;	 /* grid = 2^bits cells per axis on [0,1)^3; settings.h:59 DOMAIN_BITS (7..10) */
	int max_ppc;		 /* particles per cell capacity; settings.h:75 (reference 128); power of two <= 128 */
	int boundary_blocks; /* slip-wall zone in blocks; settings.h:63 G_BOUNDARY_CONDITION = 2 */
	float gravity;		 /* settings.h:85 (-9.8; MGSP settings.h:108 uses -4.9) */
	float cfl;			 /* utility_funcs.hpp:41 uses 0.5 (MGSP utility_funcs.hpp:39: 0.3) */
	int64_t max_blocks;	 /* (initial) capacity in (exterior) blocks; settings.h:89 G_MAX_ACTIVE_BLOCK; 0 = size from the models */
	int grow;			 /* 1 (default): block / bin capacities grow by 3/2 once 3/4 full, like check_capacity()
							(gmpm_simulator.cuh:283-300); 0: fixed capacities, MPM_ERR_CAPACITY when exceeded */
	int drop_overflow;	 /* 0 (default): a block receiving more than max_ppc * 64 particles is MPM_ERR_CAPACITY; 1: the particles beyond the
							capacity are dropped and counted (mpm_diagnostics.dropped_particles) - what the reference does, silently and
							per cell (particle_buffer.cuh:122-130) */
	int sync_interval;	 /* mpm_run_fixed: substeps enqueued between two host synchronisations (the kernels read their block counts from
							device memory; errors raised in between are reported at the next synchronisation).  0 = default (8), 1 = one
							synchronisation per substep.  The reference synchronises six times per substep (gmpm_simulator.cuh:398-564) */
	int reserved[3];
} mpm_config;

/* Material parameter block (Projects/GMPM/particle_buffer.cuh:141-264).  Unused fields are ignored. */
typedef struct mpm_material_params {
	float rho;			  /* density; mass = rho * volume (particle_buffer.cuh:155-162) */
	float volume;		  /* particle volume */
	float youngs_modulus; /* FC / SAND / NACC */
	float poisson_ratio;
	float bulk;		 /* J_FLUID */
	float gamma;	 /* J_FLUID */
	float viscosity; /* J_FLUID */
	float beta;		 /* SAND (1.0) / NACC (0.5) */
	float xi;		 /* NACC hardening factor */
	float cohesion;	 /* SAND */
	float yield_surface;   /* SAND */
	float msqr;			   /* NACC */
	float log_jp0;		   /* SAND 0, NACC -0.01 */
	int volume_correction; /* SAND */
	int hardening_on;	   /* NACC */
	int reserved[5];
} mpm_material_params;

/* Block / bin counts after a rebuild (gmpm_simulator.cuh:572-575 prints exactly these). */
typedef struct mpm_counts {
	int particle_blocks; /* partition_block_count */
	int neighbor_blocks; /* neighbor_block_count  */
	int exterior_blocks; /* exterior_block_count  */
	int model_count;
	int64_t bins[8];	  /* per model, bincount[i] */
	int64_t particles[8]; /* per model, particles currently bucketed */
} mpm_counts;

/* Per-phase device time of the most recent substep, in milliseconds (reference CudaTimer tags,
 * gmpm_simulator.cuh:346,400,507,526,543,576). */
typedef struct mpm_timers {
	float grid_update_ms;
	float g2p2g_ms;
	float partition_ms;
	float halo_ms;
	float total_ms;
	float reserved[3];
} mpm_timers;

typedef struct mpm_ctx mpm_ctx;

/* Fill cfg with the reference defaults for a grid resolution (settings.h). */
int mpm_default_config(int domain_bits, mpm_config* cfg);
/* Fill p with the reference defaults for a material at a resolution (particle_buffer.cuh:144-264;
 * note the FIXED_COROTATED and SAND default volume carries the reference's 10x factor, :176,:203). */
int mpm_default_material(int material, int domain_bits, mpm_material_params* p);

/* GmpmSimulator::GmpmSimulator + initialize (gmpm_simulator.cuh:121-166), Cuda::Cuda (Cuda.cu:28-137). */
int mpm_create(const mpm_config* cfg, int device, mpm_ctx** out);
void mpm_destroy(mpm_ctx* ctx);
const char* mpm_last_error(const mpm_ctx* ctx);

/* init_model<M> + update_*_parameters (gmpm_simulator.cuh:168-254). xyz = n*3 floats (host). Returns the
 * model index through *model_id. */
int mpm_add_model(mpm_ctx* ctx, int material, const mpm_material_params* params, const float* xyz, size_t n, const float v0[3], int* model_id);

/* initial_setup (gmpm_simulator.cuh:637-781): activate blocks, bucket particles, fill bins, build the
 * neighbor/exterior partition, rasterize the initial grid. */
int mpm_initial_setup(mpm_ctx* ctx);

/* Grid-update phase (gmpm_simulator.cuh:326-347; kernel mgmpm_kernels.cuh:325-420).  *max_vel_sqr receives
 * max |v|^2 over grid nodes (inf if a NaN was seen). */
int mpm_grid_update(mpm_ctx* ctx, float dt, float* max_vel_sqr);
/* compute_dt (utility_funcs.hpp:36-49) with this context's dx and CFL. */
float mpm_compute_dt(const mpm_ctx* ctx, float max_vel, float cur_time, float next_time, float dt_default);
/* G2P2G phase (gmpm_simulator.cuh:364-413; kernel mgmpm_kernels.cuh:665-937). */
int mpm_g2p2g(mpm_ctx* ctx, float dt, float next_dt);
/* Partition rebuild (gmpm_simulator.cuh:415-579), ends with the double-buffer roll. counts may be NULL. */
int mpm_rebuild_partition(mpm_ctx* ctx, mpm_counts* counts);

/* One whole substep = grid update, host dt, g2p2g, rebuild (the body of the loop gmpm_simulator.cuh:324-580).
 * dt is the current step; *next_dt = compute_dt(sqrt(max|v|^2), step_time, frame_time, dt_default). */
int mpm_substep(mpm_ctx* ctx, float dt, float step_time, float frame_time, float dt_default, float* next_dt, float* max_vel);
/* n substeps with a fixed dt (next_dt = dt): the loop of gmpm_simulator.cuh:324-580 without its host logic.  The substeps are enqueued
 * mpm_config.sync_interval at a time; between two host synchronisations every kernel reads its block counts from device memory.  A capacity
 * overflow or a non-finite velocity raised inside a window is reported when the window ends (the context is then in an undefined state, as
 * after any error of a run: load a checkpoint or destroy it).  Timers: per-substep averages over the call. */
int mpm_run_fixed(mpm_ctx* ctx, int nsteps, float dt);

/* output_model (gmpm_simulator.cuh:594-634; kernel mgmpm_kernels.cuh:1087-1122): positions of a model, order
 * unspecified.  *n: in = capacity of xyz in particles, out = particles written. */
int mpm_retrieve_positions(mpm_ctx* ctx, int model, float* xyz, size_t* n);
/* Extension used by the parity tests: also the per-particle state, same order as xyz.
 * state9: for FC/SAND/NACC the left Cauchy-Green tensor b = F F^T (symmetric 3x3, 9 floats) - the state this engine carries
 * instead of the reference's F: every model on the path is isotropic, so positions, grid and log Jp depend on F only through b
 * (claymore_amd/csrc/mpm_device_math.hpp); a negative state9[9*i] marks a reflected F (det F < 0, |.| is b00) -, or J in
 * state9[9*i] for J_FLUID; logjp may be NULL.  The reference itself retrieves positions only (mgmpm_kernels.cuh:1087-1122). */
int mpm_retrieve_state(mpm_ctx* ctx, int model, float* xyz, float* state9, float* logjp, size_t* n);
/* What state9 of mpm_retrieve_state holds for the solid models.  Until ABI 4 this library returned the deformation gradient F there
 * (what the CPU oracle's mpmo_retrieve_state still returns); since ABI 5 it returns b = F F^T, from which F cannot be recovered (the
 * rotation part of F never reaches an output of this path).  A caller written against the older meaning asks here instead of guessing;
 * mpm_build_info() carries the same fact as "state=b" and the ABI number. */
enum { MPM_STATE_F = 0, MPM_STATE_B = 1 };
int mpm_state_kind(void);

int mpm_get_counts(mpm_ctx* ctx, mpm_counts* counts);

/* Full-state checkpoint / restart at a substep boundary (the reference has no restart: its only
 * output is output_model's position dump, gmpm_simulator.cuh:594-634).  The buffer holds both partitions' key lists, the
 * grid, and per model the bins, bin offsets, block sizes and packed advection lists.  mpm_checkpoint_load needs a context
 * with the same configuration and models that has been through mpm_initial_setup; capacities grow as needed.  After a
 * load the MGSP halo tags must be recomputed.  HIP library only. */
int mpm_checkpoint_size(mpm_ctx* ctx, size_t* bytes);
int mpm_checkpoint_save(mpm_ctx* ctx, void* buf, size_t capacity, size_t* written);
int mpm_checkpoint_load(mpm_ctx* ctx, const void* buf, size_t bytes);

/* Level-set collision object of the MGSP grid update (Projects/MGSP/boundary_condition.cuh:25-250, the second
 * update_grid_velocity_query_max overload Projects/MGSP/mgmpm_kernels.cuh:323-399, set up by
 * MgspBenchmark::init_boundary mgsp_benchmark.cuh:257-266).  Field values of SignedDistanceGrid. */
enum { MPM_BOUNDARY_STICKY = 0, MPM_BOUNDARY_SLIP = 1, MPM_BOUNDARY_SEPARATE = 2 }; /* BoundaryT, boundary_condition.cuh:19-23 */
typedef struct mpm_collision_object {
	int type;			/* BoundaryT; default STICKY (:46) */
	float friction;		/* 0.3 (:45) */
	float scale;		/* 1 (:44) */
	float dsdt;			/* 0 (:43) */
	float trans[3];		/* 0 */
	float trans_vel[3]; /* 0 */
	float omega[3];		/* 0 */
	float rot_mat[9];	/* identity; stored as the reference stores vec3x3: element (i, j) at [3 i + j] */
	float time;			/* `current_time` handed to detect_and_resolve_collision; the reference passes 0.f (:364) */
	int reserved[3];
} mpm_collision_object;
int mpm_default_collision_object(mpm_collision_object* obj);
/* Install (obj != NULL) or remove (obj == NULL) the collision object.  sdf / grad_x / grad_y / grad_z: one float per grid
 * NODE of the whole domain, N = 2^domain_bits per axis, node (i, j, k) at [(i N + j) N + k] - the layout of the
 * reference's `<name>_sdf.bin` / `_grad_{0,1,2}.bin` files (boundary_condition.cuh:252-321).  Copied during the call.
 * With an object installed, mpm_grid_update follows the reference's boundary overload, including its max-velocity
 * quirk (|v|^2 is accumulated twice, mgmpm_kernels.cuh:365-373). */
int mpm_set_collision_object(mpm_ctx* ctx, const mpm_collision_object* obj, const float* sdf, const float* grad_x, const float* grad_y, const float* grad_z);

/* Current capacities and the number of times check_capacity() (gmpm_simulator.cuh:283-300) has grown them: blocks
 * (exterior count limit), bins per model (bin_capacity[8]).  HIP library only. */
int mpm_get_capacity(mpm_ctx* ctx, int64_t* block_capacity, int64_t* bin_capacity, int* growth_events);
int mpm_get_timers(mpm_ctx* ctx, mpm_timers* t);
/* What the reference loses silently, counted since initial_setup (as of the last completed rebuild): particles that left
 * the domain or the block neighbourhood (add_advection finds no block, particle_buffer.cuh:105-113) and P2G contributions
 * discarded because a particle moved more than one cell in a substep (mgmpm_kernels.cuh:877-885, a CFL violation).
 * A run that conserves mass has both at zero; bench.py and the full-size tests assert that.  HIP library only. */
typedef struct mpm_diagnostics {
	int64_t lost_particles;
	int64_t discarded_p2g;
	int overflow_flags; /* bit 0: block capacity, bit 1: particles per block (MPM_ERR_CAPACITY, unless bit 1 with drop_overflow) */
	int dropped_particles; /* cumulative; only with mpm_config.drop_overflow */
	int reserved[4];
} mpm_diagnostics;
int mpm_get_diagnostics(mpm_ctx* ctx, mpm_diagnostics* d);
/* Sum over the current grid of {mass, momentum x, y, z} (the reference's sum_grid_mass debug kernel,
 * mgmpm_kernels.cuh:1034-1037, extended to momentum); valid between rebuild and the next grid update. */
int mpm_grid_totals(mpm_ctx* ctx, double out[4]);
/* Table consistency of the current partition (the reference's check_table debug kernel, mgmpm_kernels.cuh:1022-1032): the number of blocks i
 * with query(active_keys[i]) != i plus the number of table entries that belong to no block; 0 = consistent; negative = error. */
int mpm_check_table(mpm_ctx* ctx);
/* Dense dump of the current grid for parity tests: for every neighbor block, key (3 ints) and 256 floats
 * {mass[64], mvx[64], mvy[64], mvz[64]} (grid_buffer.cuh:12-14 layout). *nblocks in = capacity, out = count. */
int mpm_dump_grid(mpm_ctx* ctx, int* keys, float* blocks, size_t* nblocks);
/* How the library was built: "claymore_hip <abi> experiment=<none | list of -D switches>".  A product library reports
 * experiment=none (claymore_amd/csrc/mpm_device_math.hpp: the switches that change what the kernels compute only exist under
 * -DMPM_EXPERIMENT); tests/test_abi.py asserts it for the shipped library.  Needs no device. */
const char* mpm_build_info(void);
/* G2P2G kernel time of the last call as measured with HIP events on the compute stream. */
int mpm_last_g2p2g_ms(mpm_ctx* ctx, float* ms);

/* ---- function-level entry points used by the parity tests (device versions of the per-particle math) ---- */
/* What the constitutive models take from math::svd (Library/MnBase/Math/Matrix/svd.cuh:27-1123; they need U and the singular
 * values only, V cancels): the eigen-decomposition F F^T = U diag(lam) U^T, lam_k = sigma_k^2, U a rotation, columns in no
 * particular order.  F[n*9] column-major -> out[n*12] = U(9) lam(3). */
int mpm_test_eig(const float* F, size_t n, float* out12, int device);
/* compute_stress<M> (Projects/GMPM/constitutive_models.cuh) on deformation gradients F[n*9]: out19 = b'(9) PF(9) logjp'(1), b' the
 * left Cauchy-Green tensor F' F'^T of the model's (possibly projected) F' - what a particle of this engine stores. */
int mpm_test_stress(int material, const mpm_material_params* p, const float* F, const float* logjp, size_t n, float* out19, int device);

/* ---- multi-GPU (MGSP static particle partition, Projects/MGSP/mgsp_benchmark.cuh:661-776) ----
 * One context per rank/GPU owns a fixed particle subset; grids of different ranks overlap only in blocks touched by
 * particles of both.  All buffers named dev_* are DEVICE pointers supplied by the caller (e.g. torch tensors) so that
 * the transport (RCCL all-gather / all-to-all over xGMI) stays outside this library.  The exchange kernels run on the
 * context's comm stream, ordered against the compute stream with events inside the library. */
/* Copy this rank's neighbor-block keys (ivec3, count = neighbor_blocks) into dev_keys: the payload of the all-gather
 * that replaces the pairwise cudaMemcpyAsync of mgsp_benchmark.cuh:681-686.  At most capacity_blocks keys are copied;
 * *count always receives the true number, so a caller with a too small buffer can retry. */
int mpm_halo_keys(mpm_ctx* ctx, int* dev_keys, int capacity_blocks, int* count);
/* reset_overlap_marks / reset counts (mgsp_benchmark.cuh:690-692). */
int mpm_halo_tag_begin(mpm_ctx* ctx);
/* mark_overlapping_blocks (halo_kernels.cuh:21-35): dev_peer_keys = npeer_keys ivec3 of peer `peer` (0..31); marks this
 * rank's neighbor blocks that the peer also owns and builds the send list for that peer. */
int mpm_halo_tag_peer(mpm_ctx* ctx, int peer, const int* dev_peer_keys, int npeer_keys);
/* collect_blockids_for_halo_reduction (halo_kernels.cuh:37-62): split the particle blocks into halo / interior lists and
 * read the per-peer send counts back.  send_counts (host, 32 ints, may be NULL) receives them. */
int mpm_halo_tag_end(mpm_ctx* ctx, int* halo_particle_blocks, int* send_counts);
/* Phases of a multi-GPU substep (mgsp_benchmark.cuh:421-465): clears + g2p2g on the halo list, then on the interior list. */
int mpm_g2p2g_halo(mpm_ctx* ctx, float dt, float next_dt);
int mpm_g2p2g_interior(mpm_ctx* ctx, float dt, float next_dt);
/* collect_grid_blocks (halo_kernels.cuh:64-80): gather the blocks shared with `peer` from grid `gid` (1 = this step's
 * P2G target, 0 = current grid, used once after initial_setup) into dev_keys (n*3 ints) / dev_blocks (n*256 floats). */
int mpm_halo_collect(mpm_ctx* ctx, int peer, int gid, int* dev_keys, float* dev_blocks, int capacity_blocks, int* nsend);
/* reduce_grid_blocks (halo_kernels.cuh:82-97): add nrecv received blocks into grid `gid` (hardware f32 atomics). */
int mpm_halo_reduce(mpm_ctx* ctx, int gid, const int* dev_keys, const float* dev_blocks, int nrecv);
/* ---- fused multi-GPU substep: the same phases with ONE host synchronisation per substep ----
 * (the phase-by-phase calls above synchronise once each, like the reference's issue()/sync() barriers,
 * mgsp_benchmark.cuh:336-356; at 5 M particles per GPU that costs more than the kernels.)
 *   mpm_mgsp_begin        : grid update (no read-back) + clears + G2P2G on the halo list
 *   mpm_halo_collect xN, all-to-all-v, mpm_g2p2g_interior, mpm_halo_reduce xN      (as above)
 *   mpm_mgsp_rebuild_export: partition rebuild launches + padded key export for the all-gather: dev_keys holds
 *                            pad_rows rows of 3 ints, row 0 = {neighbor-block count, 0, 0}, rows 1.. = keys
 *   all-gather of the padded key lists (caller)
 *   mpm_mgsp_tag          : overlap marks / send lists / halo split from the gathered lists (world*pad_rows rows),
 *                            all sizes read on the device
 *   mpm_mgsp_end          : the one synchronisation: counts, send counts (32), halo block count, max |v|^2 of this
 *                            step's grid update; rolls the double buffers.  *max_peer_rows = largest key-list length
 *                            (+1) any rank exported: if it exceeds pad_rows the caller re-tags with a larger pad. */
int mpm_mgsp_begin(mpm_ctx* ctx, float dt, float next_dt);
int mpm_mgsp_rebuild_export(mpm_ctx* ctx, int* dev_keys, int pad_rows);
int mpm_mgsp_tag(mpm_ctx* ctx, const int* dev_all_keys, int pad_rows, int world, int rank);
int mpm_mgsp_end(mpm_ctx* ctx, int* send_counts, int* halo_particle_blocks, int* max_peer_rows, float* max_vel_sqr);

/* HIP stream handles (as void*) so that the caller can order collectives against the engine. */
int mpm_streams(mpm_ctx* ctx, void** compute_stream, void** comm_stream);
int mpm_sync(mpm_ctx* ctx);

/* ---- MGSP group driver: the multi-GPU substep loop in C++ on RCCL (MgspBenchmark::main_loop, mgsp_benchmark.cuh:361-559;
 * worker threads issue()/sync() :309-356; tagging :661-720; halo exchange :723-776, halo_buffer.cuh:54-59).  One group
 * handle per rank, each on its own context.  Transport of mpm_group_create: RCCL (ncclAllGather for the block keys,
 * ncclGroup{ncclSend, ncclRecv} for the symmetric halo exchange on the context's comm stream, ncclAllReduce(max) for dt);
 * the caller only has to carry the 128-byte unique id from rank 0 to the other ranks (MPI, a file, torch.distributed ...).
 * mpm_group_create_local builds `world` handles in ONE process (contexts may share a device) on device-to-device copies:
 * single-GPU boxes, tests, and hosts that drive all GPUs from worker threads; its calls must be made concurrently, one
 * thread per rank.  The environment variable MPM_RCCL_LIBRARY names the collective library to load instead of the system's
 * librccl.so (a differently built RCCL; the tests' in-process double, tests/rccl_double/).  HIP library only. */
typedef struct mpm_group mpm_group;
int mpm_group_unique_id(void* id128);
int mpm_group_create(mpm_ctx* ctx, int rank, int world, const void* id128, mpm_group** out);
int mpm_group_create_local(mpm_ctx* const* ctxs, int world, mpm_group** out /* [world] */);
void mpm_group_destroy(mpm_group* g);
const char* mpm_group_last_error(const mpm_group* g);
/* The transport the group ended up with: "rccl", "peer-direct" (one process, several devices, hipMemcpyPeerAsync behind the peer's events:
 * the reference's own mechanism, halo_buffer.cuh:54-59) or "device-copy" (contexts of one device, or a pair of devices without peer access:
 * synchronous copies behind host barriers).  A caller that asked for peer-direct reports the fall-back with it. */
const char* mpm_group_transport(const mpm_group* g);
/* initial_setup of every rank + first tagging + one exchange that sums the rasterised grids (mgsp_benchmark.cuh:561-659). */
int mpm_group_initial_setup(mpm_group* g);
/* Restart (row f4 for the multi-GPU loop; the reference has none): after EVERY rank has loaded its own checkpoint with
 * mpm_checkpoint_load (taken after a call of this driver returned), rebuild what a checkpoint does not carry - padded key length,
 * overlap marks, halo / interior lists, per-peer send lists.  Collective: every rank calls it. */
int mpm_group_resume(mpm_group* g);
/* One substep, one host synchronisation; *max_vel_sqr = this rank's max |v|^2 (not reduced over ranks). */
int mpm_group_substep(mpm_group* g, float dt, float next_dt, float* max_vel_sqr);
int mpm_group_run_fixed(mpm_group* g, int nsteps, float dt);
/* compute_dt of the MGSP project (Projects/MGSP/utility_funcs.hpp:32-55): CFL 0.3 and the 0.51 frame-remainder rule. */
float mpm_group_compute_dt(const mpm_group* g, float max_vel, float cur_time, float next_time, float dt_default);
/* main_loop with adaptive dt from the maximum grid velocity over all ranks (:410-418); on_frame may be NULL. */
int mpm_group_main_loop(mpm_group* g, int frames, int fps, float dt_default, void (*on_frame)(int frame, void* user), void* user, int* steps_out);
int mpm_group_stats(const mpm_group* g, int* send_counts32, int* halo_particle_blocks, float* g2p2g_ms_avg);

#ifdef __cplusplus
}
#endif
#endif
