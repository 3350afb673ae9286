// claymore_hip.hip — host side of libclaymore_hip.so: the C ABI of include/claymore_amd.h on top of the gfx950
// kernels in mpm_kernels.hpp.  One context = one HIP device, two streams (compute, comm), all buffers owned here.
// Replaces the launch/orchestration part of GmpmSimulator (Projects/GMPM/gmpm_simulator.cuh:283-781) and the thin
// CUDA wrapper it sits on (Library/MnSystem/Cuda/Cuda.h, Cuda.cu).  No CPU fallback: every path needs a GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/claymore_amd.h"
#include "mpm_kernels.hpp"

using namespace mpm;
#define MPM_STR_(x) #x
#define MPM_STR(x) MPM_STR_(x)

namespace {
// ONE device block (and its pinned mirror) holds everything the host reads back at a synchronisation: the status words, the MGSP halo
// counters + the peers' key-list lengths, and the max |v|^2 slots - one device-to-host copy instead of three.
constexpr int kHaloWords	 = 128;// 2 + 32 send counts + 32 peer list lengths + 32 peer status words, padded
constexpr int kStatusBlock = ST_WORDS + kHaloWords + kMaxVelSlots * kMaxVelStride;// in 4-byte words

struct Partition {
	int* table = nullptr;
	int* keys  = nullptr;
	int* count = nullptr;
};

struct Model {
	int material = 0;
	int nch		 = 0;
	mpm_material_params p {};
	MaterialConst mc {};
	size_t n	 = 0;
	float* d_xyz = nullptr;// initial particle array; also the staging buffer of retrieve
	float v0[3]	 = {0, 0, 0};
	float* bins[2]	 = {nullptr, nullptr};
	size_t bin_cap	 = 0;
	int* binoff[2]	 = {nullptr, nullptr};
	int* list[2]	 = {nullptr, nullptr};
	int* size		 = nullptr;
	int* row_of		 = nullptr;
	int* out_count	 = nullptr;
	int* keep		 = nullptr;// per block of the numbering G2P2G ran in: its particle count if every particle stayed with an unchanged sort key, else -1
	int* blockinfo	 = nullptr;// [block][kInfoRow], written by prepare_blocks_kernel
	bool pair		 = false;  // the lists are in the pair layout (mpm_kernels.hpp) and G2P2G carries two particles per lane (mpm_g2p2g_pair.hpp)
	int* pairinfo[2] = {nullptr, nullptr};// pair layout only, [block][kPairChunks]: [0] full pairs per chunk of a block's sorted list (written by the sort, read by G2P2G), [1] what G2P2G hands on to the next sort
	int64_t bincount = 0;
	int64_t bincount_src = 0;// bins in use in bins[rollid] (the source of the next g2p2g): the previous bincount
	int64_t bucketed = 0;// particles currently in the advection lists (device-counted at each rebuild)
	int list_in		 = 0;// which list buffer g2p2g reads next
};

}// namespace

struct mpm_ctx {
	mpm_config cfg {};
	GridCfg g {};
	int device = 0;
	hipStream_t s_compute = nullptr, s_comm = nullptr;
	hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_g0 = nullptr, ev_g1 = nullptr, ev_comm = nullptr, ev_halo = nullptr;
	hipEvent_t ev_tag0 = nullptr, ev_tag1 = nullptr;// group loop: key list exported (compute -> comm) / tagging complete (comm -> compute); created on first use
	// MGSP windowed loop: the status read-back of substep t is waited for AFTER the halo-first G2P2G of substep t + 1 has been enqueued,
	// so the timing events exist twice (index = parity of the substep)
	// lean_events (the windowed group loop in its deferred order): a substep records three events on the compute stream instead of seven - an event between two
	// kernels is a barrier packet of its own, 5-6 us each (profiles/r06_rank_alone_seq.txt) -: no start event (a substep's time is end-to-end of the previous one's),
	// the end event doubles as the read-back's, the halo event is recorded once.  ev_pending: the event the outstanding read-back completes with.
	bool lean_events = false;
	hipEvent_t ev_pending = nullptr;
	hipEvent_t ev_status = nullptr, ev2_a[2] = {nullptr, nullptr}, ev2_b[2] = {nullptr, nullptr}, ev2_g0[2] = {nullptr, nullptr}, ev2_g1[2] = {nullptr, nullptr};
	// grid[0] already holds the velocities of the coming substep (the rebuild's carry-over applied the grid update for this dt):
	// only inside mpm_run_fixed, never when a call returns
	bool grid_preupdated = false;
	float preupdate_dt	 = 0.f;
	float fuse_dt_once	 = 0.f;// set by a driver that knows the next substep's dt (mpm_group_run_fixed): consumed by the next rebuild
	// Between two host synchronisations of mpm_run_fixed the block counts below are ESTIMATES (the values of the last
	// synchronisation): launches are sized by them with a margin, every kernel reads the true counts from the status block.
	bool rebuild_cleared = false;// the rebuild's part of substep_clear_kernel has been issued together with the P2G part
	std::vector<hipEvent_t> ev_ring;// 3 events per substep of a window (substep start, G2P2G start / end; its end is the next one's start) + the window's closing event
	Partition part[2];
	float* grid[2] = {nullptr, nullptr};
	int rollid	   = 0;
	int pbc = 0, nbc = 0, ebc = 0;
	int pbc_prev = 0;// particle blocks of the previous numbering (the one the particle data is laid out in)
	std::vector<Model> models;
	int* d_status		  = nullptr;// ST_WORDS ints
	int* h_status		  = nullptr;// pinned
	unsigned* d_maxvel	  = nullptr;
	float* h_maxvel		  = nullptr;// pinned
	double* d_totals	  = nullptr;
	unsigned long long* d_counter = nullptr;
	bool ready = false;
	int capacity_events = 0;// number of capacity growths so far (check_capacity)
	long long books_bias = 0;// check_books: particles a loaded checkpoint's state lacked beyond this context's own lost / dropped counters
	bool has_collision = false;// level-set collision object of the MGSP grid update
	CollisionObject collision {};
	float4* d_sdf = nullptr;
	mpm_timers timers {};
	float last_g2p2g_ms = 0.f;
	// halo state (MGSP)
	int* d_overlap	   = nullptr;// per neighbour block: bit mask of peers that also own it
	int* d_halo_list   = nullptr;// particle blocks touching an overlap block
	int* d_inner_list  = nullptr;// per particle block: 1 = interior (not touched by the halo-first pass)
	int* d_halo_counts = nullptr;// [0]=halo blocks, [1]=interior blocks, [2+peer]=send count for peer
	int* h_halo_counts = nullptr;// pinned mirror
	int* d_peer_rows   = nullptr;// key-list lengths exported by every rank (fused substep); lives behind d_halo_counts
	int* h_peer_rows   = nullptr;// pinned, behind h_halo_counts
	int mgsp_world	   = 0;
	int mgsp_rank	   = -1;// this rank's number in the last fused tagging (-1: none yet)
	bool halo_tagged   = false;
	int* d_send_ids[32] = {nullptr};
	int n_halo = 0, n_inner = 0;
	int send_count[32] = {0};
	std::string err;
};

// An interrupted call leaves no promise about the grid behind (a checkpoint load restores a canonical one): the three flags that span
// kernels - grid already updated for the next dt, rebuild clear already issued, next dt known - are reset on EVERY error exit (HIP_TRY
// returns included), by scope.
struct FlagGuard {
	mpm_ctx* ctx;
	bool armed = true;
	~FlagGuard() {
		if(armed) {
			ctx->grid_preupdated = false;
			ctx->fuse_dt_once	 = 0.f;
			ctx->rebuild_cleared = false;
		}
	}
};

#define HIP_TRY(expr)                                                                                            \
	do {                                                                                                         \
		hipError_t e_ = (expr);                                                                                  \
		if(e_ != hipSuccess) {                                                                                   \
			ctx->err = std::string(#expr) + " -> " + hipGetErrorString(e_) + " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"; \
			return MPM_ERR_DEVICE;                                                                               \
		}                                                                                                        \
	} while(0)

// the grid update leaves its running maximum in kMaxVelSlots slots (non-negative floats; inf marks a NaN velocity)
static float host_maxvel(const mpm_ctx* ctx) {
	float m = 0.f;
	for(int i = 0; i < kMaxVelSlots; ++i) m = std::max(m, ctx->h_maxvel[i * kMaxVelStride]);
	return m;
}

// Which materials' G2P2G runs two particles per lane (bit m = material m; mpm_g2p2g_pair.hpp).  MPM_PAIR_BUILD: the instantiations compiled in;
// MPM_PAIR_DEFAULT: the ones used unless the environment says otherwise (MPM_G2P2G_PAIRS=<mask>, read once: same-library A/B and tests).
#ifndef MPM_PAIR_BUILD
#define MPM_PAIR_BUILD 0xF
#endif
#ifndef MPM_PAIR_DEFAULT
#define MPM_PAIR_DEFAULT 0xF// all four materials (NACC with the late record fetch: 163 registers, no scratch; -6 % on a 5 M-particle NACC sphere)
#endif
static int pair_mask() {
	static const int mask = [] {
		const char* e = std::getenv("MPM_G2P2G_PAIRS");
		return (e && *e ? (int) std::strtol(e, nullptr, 0) : MPM_PAIR_DEFAULT) & MPM_PAIR_BUILD;
	}();
	return mask;
}

static int launch_prepare(mpm_ctx* ctx, int cur, int prev, bool list_is_out, int binoff_sel, const int* pbc_ptr, int nblocks_est, bool sort, bool publish);

static int fail(mpm_ctx* ctx, int code, const std::string& msg) {
	ctx->err = msg;
	return code;
}

template<typename T>
static hipError_t dalloc(T** p, size_t n) {
	return hipMalloc((void**) p, sizeof(T) * (n ? n : 1));
}

// Scratch device array that is released on every exit path of the function that owns it.
template<typename T>
struct DevScratch {
	T* p = nullptr;
	DevScratch() = default;
	DevScratch(const DevScratch&) = delete;
	DevScratch& operator=(const DevScratch&) = delete;
	~DevScratch() {
		if(p) hipFree(p);
	}
	hipError_t alloc(size_t n) { return dalloc(&p, n); }
};

// Replace *p (old_n elements) by a zero-initialised array of new_n elements holding the old contents.
template<typename T>
static hipError_t regrow(T** p, size_t old_n, size_t new_n, hipStream_t s) {
	T* q		 = nullptr;
	hipError_t e = hipMalloc((void**) &q, sizeof(T) * new_n);
	if(e != hipSuccess) return e;
	if((e = hipMemsetAsync(q, 0, sizeof(T) * new_n, s)) != hipSuccess || (*p && old_n && (e = hipMemcpyAsync(q, *p, sizeof(T) * old_n, hipMemcpyDeviceToDevice, s)) != hipSuccess)
	   || (e = hipStreamSynchronize(s)) != hipSuccess) {
		hipFree(q);// *p is untouched: the caller still owns a valid array of old_n elements
		return e;
	}
	if(*p) hipFree(*p);
	*p = q;
	return hipSuccess;
}

static inline unsigned cdiv(size_t a, size_t b) {
	return (unsigned) ((a + b - 1) / b);
}

extern "C" {

int mpm_default_config(int domain_bits, mpm_config* cfg) {
	if(!cfg || domain_bits < 4 || domain_bits > 10) return MPM_ERR_INVALID;
	memset(cfg, 0, sizeof(*cfg));
	cfg->domain_bits	 = domain_bits;
	cfg->max_ppc		 = 128;	  // settings.h:75
	cfg->boundary_blocks = 2;	  // settings.h:63
	cfg->gravity		 = -9.8f; // settings.h:85
	cfg->cfl			 = 0.5f;  // settings.h:53
	cfg->max_blocks		 = 0;
	cfg->grow			 = 1;
	return MPM_OK;
}

int mpm_default_material(int material, int domain_bits, mpm_material_params* p) {
	if(!p) return MPM_ERR_INVALID;
	memset(p, 0, sizeof(*p));
	const float n	  = (float) (1u << domain_bits);
	p->rho			  = 1e3f;// settings.h:81-83
	p->youngs_modulus = 5e3f;
	p->poisson_ratio  = 0.4f;
	const float vol1  = 1.0f / n / n / n / 8.0f;
	const float vol10 = 10.f / n / n / n / 8.0f;// particle_buffer.cuh:176,:203 (sic)
	switch(material) {
		case MPM_J_FLUID:
			p->volume	 = vol1;
			p->bulk		 = 4e4f;
			p->gamma	 = 7.15f;
			p->viscosity = 0.01f;
			break;
		case MPM_FIXED_COROTATED: p->volume = vol10; break;
		case MPM_SAND:
			p->volume			 = vol10;
			p->cohesion			 = 0.f;
			p->beta				 = 1.f;
			p->yield_surface	 = 0.816496580927726f * 2.f * 0.5f / (3.f - 0.5f);
			p->volume_correction = 1;
			p->log_jp0			 = 0.f;
			break;
		case MPM_NACC:
			p->volume		= vol1;
			p->xi			= 0.8f;
			p->beta			= 0.5f;
			p->msqr			= 3.423772074299613f;
			p->hardening_on = 1;
			p->log_jp0		= -0.01f;
			break;
		default: return MPM_ERR_INVALID;
	}
	return MPM_OK;
}

static MaterialConst make_material_const(const mpm_material_params& p) {
	MaterialConst mc {};
	const float e = p.youngs_modulus, nu = p.poisson_ratio;
	mc.volume			 = p.volume;
	mc.mass				 = p.volume * p.rho;						  // particle_buffer.cuh:158,:186,:252
	mc.lambda			 = e * nu / ((1 + nu) * (1 - 2 * nu));		  // :187
	mc.mu				 = e / (2 * (1 + nu));						  // :188
	mc.bm				 = 2.f / 3.f * (e / (2 * (1 + nu))) + (e * nu / ((1 + nu) * (1 - 2 * nu)));// :255
	mc.bulk				 = p.bulk;
	mc.gamma			 = p.gamma;
	mc.viscosity		 = p.viscosity;
	mc.cohesion			 = p.cohesion;
	mc.beta				 = p.beta;
	mc.yield_surface	 = p.yield_surface;
	mc.xi				 = p.xi;
	mc.msqr				 = p.msqr;
	mc.log_jp0			 = p.log_jp0;
	mc.volume_correction = p.volume_correction;
	mc.hardening_on		 = p.hardening_on;
	return mc;
}

int mpm_create(const mpm_config* cfg, int device, mpm_ctx** out) {
	if(!cfg || !out) return MPM_ERR_INVALID;
	if(cfg->domain_bits < 4 || cfg->domain_bits > 10) return MPM_ERR_INVALID;
	if(cfg->max_ppc < 1 || cfg->max_ppc > 128 || (cfg->max_ppc & (cfg->max_ppc - 1))) return MPM_ERR_INVALID;
	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || device < 0 || device >= ndev) {
		fprintf(stderr, "claymore_hip: no usable HIP device %d (found %d) - there is no CPU fallback\n", device, ndev);
		return MPM_ERR_DEVICE;
	}
	mpm_ctx* ctx = new mpm_ctx();
	ctx->cfg	 = *cfg;
	ctx->device	 = device;
	GridCfg& g	 = ctx->g;
	g.gbits		 = cfg->domain_bits - 2;
	g.G			 = 1 << g.gbits;
	g.ppb		 = cfg->max_ppc * 64;
	g.pid_bits	 = 0;
	while((1 << g.pid_bits) < g.ppb) g.pid_bits++;
	g.boundary = cfg->boundary_blocks;
	g.dx_inv   = (float) (1 << cfg->domain_bits);
	g.dx	   = 1.f / g.dx_inv;
	g.d_inv	   = 4.f * g.dx_inv * g.dx_inv;
	g.gravity  = cfg->gravity;
	g.cap	   = 0;
	// the comm stream gets the highest priority: its small collect / reduce kernels (and RCCL's) are enqueued BEHIND the big
	// interior G2P2G and must not wait for it to drain
	int prio_lo = 0, prio_hi = 0;
	if(hipSetDevice(device) == hipSuccess) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
	if(hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->s_compute, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithPriority(&ctx->s_comm, hipStreamNonBlocking, prio_hi) != hipSuccess) {
		delete ctx;
		return MPM_ERR_DEVICE;
	}
	if(hipEventCreate(&ctx->ev_a) != hipSuccess || hipEventCreate(&ctx->ev_b) != hipSuccess || hipEventCreate(&ctx->ev_g0) != hipSuccess || hipEventCreate(&ctx->ev_g1) != hipSuccess
	   || hipEventCreateWithFlags(&ctx->ev_comm, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_halo, hipEventDisableTiming) != hipSuccess
	   || hipEventCreateWithFlags(&ctx->ev_status, hipEventDisableTiming) != hipSuccess) {
		mpm_destroy(ctx);// frees whatever was created (null handles are skipped)
		return MPM_ERR_DEVICE;
	}
	*out = ctx;
	return MPM_OK;
}

void mpm_destroy(mpm_ctx* ctx) {
	if(!ctx) return;
	hipSetDevice(ctx->device);
	hipDeviceSynchronize();
	hipFree(ctx->d_sdf);
	for(int i = 0; i < 2; ++i) {
		hipFree(ctx->part[i].table);
		hipFree(ctx->part[i].keys);
		hipFree(ctx->part[i].count);
		hipFree(ctx->grid[i]);
	}
	for(auto& m: ctx->models) {
		hipFree(m.d_xyz);
		hipFree(m.blockinfo);
		hipFree(m.pairinfo[0]);
		hipFree(m.pairinfo[1]);
		for(int i = 0; i < 2; ++i) {
			hipFree(m.bins[i]);
			hipFree(m.binoff[i]);
			hipFree(m.list[i]);
		}
		hipFree(m.size);
		hipFree(m.row_of);
		hipFree(m.out_count);
		hipFree(m.keep);
	}
	hipFree(ctx->d_status);
	hipFree(ctx->d_totals);
	hipFree(ctx->d_counter);
	hipFree(ctx->d_overlap);
	hipFree(ctx->d_halo_list);
	hipFree(ctx->d_inner_list);
	for(auto& p: ctx->d_send_ids) hipFree(p);
	if(ctx->h_status) hipHostFree(ctx->h_status);
	if(ctx->ev_a) hipEventDestroy(ctx->ev_a);
	if(ctx->ev_b) hipEventDestroy(ctx->ev_b);
	if(ctx->ev_g0) hipEventDestroy(ctx->ev_g0);
	if(ctx->ev_g1) hipEventDestroy(ctx->ev_g1);
	if(ctx->ev_comm) hipEventDestroy(ctx->ev_comm);
	if(ctx->ev_halo) hipEventDestroy(ctx->ev_halo);
	if(ctx->ev_tag0) hipEventDestroy(ctx->ev_tag0);
	if(ctx->ev_tag1) hipEventDestroy(ctx->ev_tag1);
	if(ctx->ev_status) hipEventDestroy(ctx->ev_status);
	for(int i = 0; i < 2; ++i)
		for(hipEvent_t e: {ctx->ev2_a[i], ctx->ev2_b[i], ctx->ev2_g0[i], ctx->ev2_g1[i]})
			if(e) hipEventDestroy(e);
	for(hipEvent_t e: ctx->ev_ring) hipEventDestroy(e);
	if(ctx->s_compute) hipStreamDestroy(ctx->s_compute);
	if(ctx->s_comm) hipStreamDestroy(ctx->s_comm);
	delete ctx;
}

const char* mpm_last_error(const mpm_ctx* ctx) {
	return ctx ? ctx->err.c_str() : "null context";
}

int mpm_add_model(mpm_ctx* ctx, int material, const mpm_material_params* params, const float* xyz, size_t n, const float v0[3], int* model_id) {
	if(!ctx) return MPM_ERR_INVALID;
	if(!params) return fail(ctx, MPM_ERR_INVALID, "mpm_add_model: material parameters are missing");
	if(ctx->ready) return fail(ctx, MPM_ERR_INVALID, "mpm_add_model: models must be added before mpm_initial_setup");
	if(!xyz && n) return fail(ctx, MPM_ERR_INVALID, "mpm_add_model: positions are missing");
	if(material < 0 || material > 3) return fail(ctx, MPM_ERR_INVALID, "unknown material");
	if((int) ctx->models.size() >= kMaxModels) return fail(ctx, MPM_ERR_CAPACITY, "too many models");
	HIP_TRY(hipSetDevice(ctx->device));
	Model m;
	m.material = material;
	m.p		   = *params;
	m.mc	   = make_material_const(*params);
	m.nch	   = material == MPM_J_FLUID ? 4 : (material == MPM_FIXED_COROTATED ? 9 : 10);// floats per particle in a bin (mpm_g2p2g.hpp: {x, y, z, J}, or a 32-B record {x, y, z, b...} + a row of b21 (+ log Jp))
	m.n		   = n;
	for(int d = 0; d < 3; ++d) m.v0[d] = v0 ? v0[d] : 0.f;
	HIP_TRY(dalloc(&m.d_xyz, 3 * n));
	if(n) {// an empty model is legal (an MGSP rank whose slab of a model is empty, an empty sampled SDF)
		HIP_TRY(hipMemcpyAsync(m.d_xyz, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, ctx->s_compute));
		HIP_TRY(hipStreamSynchronize(ctx->s_compute));
	}
	if(model_id) *model_id = (int) ctx->models.size();
	ctx->models.push_back(m);
	return MPM_OK;
}

// Host synchronisation with a short spin in front of the blocking wait: when the stream is about to drain (the end of a substep whose
// kernels are already running) the blocking wait's wake-up latency (~15-20 us) would sit between two substeps as GPU idle time.
static hipError_t sync_stream(hipStream_t s) {
	const auto t0 = std::chrono::steady_clock::now();
	for(;;) {
		const hipError_t q = hipStreamQuery(s);
		if(q != hipErrorNotReady) return q;
		if(std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) break;
	}
	return hipStreamSynchronize(s);
}
static int read_status(mpm_ctx* ctx) {
	HIP_TRY(hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(int) * kStatusBlock, hipMemcpyDeviceToHost, ctx->s_compute));// status + halo counters + max |v|^2 slots
	HIP_TRY(sync_stream(ctx->s_compute));
	return MPM_OK;
}

static int check_status(mpm_ctx* ctx) {
	const int* st = ctx->h_status;
	if(st[ST_OVERFLOW] & 1) return fail(ctx, MPM_ERR_CAPACITY, "Too much active blocks: block capacity " + std::to_string(ctx->g.cap) + " exceeded");
	if((st[ST_OVERFLOW] & 2) && !ctx->cfg.drop_overflow) return fail(ctx, MPM_ERR_CAPACITY, "particles-per-block capacity exceeded (max_ppc*64 = " + std::to_string(ctx->g.ppb) + ")");
	if(st[ST_OVERFLOW] & 4) return fail(ctx, MPM_ERR_CAPACITY, "bin capacity exceeded");
	return MPM_OK;
}

// initial_setup, gmpm_simulator.cuh:637-781
int mpm_initial_setup(mpm_ctx* ctx) {
	if(!ctx || ctx->ready || ctx->models.empty()) return MPM_ERR_INVALID;
	HIP_TRY(hipSetDevice(ctx->device));
	GridCfg& g		   = ctx->g;
	hipStream_t s	   = ctx->s_compute;
	const size_t table = (size_t) g.G * g.G * g.G;
	const int r = ctx->rollid, n = r ^ 1;
	HIP_TRY(dalloc(&ctx->d_status, kStatusBlock));
	HIP_TRY(hipHostMalloc((void**) &ctx->h_status, sizeof(int) * kStatusBlock));
	memset(ctx->h_status, 0, sizeof(int) * kStatusBlock);
	ctx->d_halo_counts = ctx->d_status + ST_WORDS;
	ctx->h_halo_counts = ctx->h_status + ST_WORDS;
	ctx->d_peer_rows   = ctx->d_halo_counts + 2 + 32;
	ctx->h_peer_rows   = ctx->h_halo_counts + 2 + 32;
	ctx->d_maxvel	   = reinterpret_cast<unsigned*>(ctx->d_status + ST_WORDS + kHaloWords);
	ctx->h_maxvel	   = reinterpret_cast<float*>(ctx->h_status + ST_WORDS + kHaloWords);
	HIP_TRY(dalloc(&ctx->d_totals, 4));
	HIP_TRY(dalloc(&ctx->d_counter, 1));
	HIP_TRY(hipMemsetAsync(ctx->d_status, 0, sizeof(int) * kStatusBlock, s));
	for(int i = 0; i < 2; ++i) {
		HIP_TRY(dalloc(&ctx->part[i].table, table));
		HIP_TRY(dalloc(&ctx->part[i].count, 1));
		HIP_TRY(hipMemsetAsync(ctx->part[i].table, 0xff, sizeof(int) * table, s));
		HIP_TRY(hipMemsetAsync(ctx->part[i].count, 0, sizeof(int), s));
	}
	// Pass 1: activate particle blocks with a provisional capacity = table size bound by the particle count
	size_t total_particles = 0;
	for(auto& m: ctx->models) total_particles += m.n;
	size_t prov = std::min(table, total_particles + 1);
	if(ctx->cfg.max_blocks > 0) prov = std::min<size_t>(table, (size_t) ctx->cfg.max_blocks);
	Partition& P = ctx->part[n];
	HIP_TRY(dalloc(&P.keys, 3 * prov));
	g.cap = (int) prov;
	for(auto& m: ctx->models)
		if(m.n) activate_blocks_kernel<<<cdiv(m.n, 256), 256, 0, s>>>(g, m.n, m.d_xyz, P.table, P.keys, P.count, ctx->d_status);
	int pbc = 0;
	HIP_TRY(hipMemcpyAsync(&pbc, P.count, sizeof(int), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	if(pbc > g.cap) return fail(ctx, MPM_ERR_CAPACITY, "Too much active blocks: " + std::to_string(pbc));
	ctx->pbc = pbc;
	// neighbours, exterior (gmpm_simulator.cuh:706-734) - registered now, still in the provisional key list, so that the
	// capacities below can be sized by the exterior block count that is actually there (a compact body has ~1.15 exterior
	// blocks per particle block, a thin sheet up to 27: sizing by a fixed multiple of pbc wasted 8 GB at C3)
	if(ctx->cfg.max_blocks <= 0 && (size_t) pbc * 27 > prov) {// the provisional list must hold every block that can be registered
		int* keys		  = nullptr;
		const size_t need = std::min(table, (size_t) pbc * 27 + 1);
		HIP_TRY(dalloc(&keys, 3 * need));
		HIP_TRY(hipMemcpyAsync(keys, P.keys, sizeof(int) * 3 * (size_t) pbc, hipMemcpyDeviceToDevice, s));
		HIP_TRY(hipStreamSynchronize(s));
		HIP_TRY(hipFree(P.keys));
		P.keys = keys;
		prov   = need;
		g.cap  = (int) prov;
	}
	// (one-time set-up: the phase counters start from the activation's count; the published counts are formed on the host below)
	HIP_TRY(hipMemcpyAsync(&ctx->d_status[ST_CNT_P], P.count, sizeof(int), hipMemcpyDeviceToDevice, s));
	register_blocks_kernel<0, 1><<<std::max(1u, std::min(4096u, cdiv((size_t) pbc * 8, 256))), 256, 0, s>>>(g, &ctx->d_status[ST_CNT_P], nullptr, &ctx->d_status[ST_CNT_N], &ctx->d_status[ST_PBC], P.table, P.keys, ctx->d_status);
	register_blocks_kernel<-1, 1><<<std::max(1u, std::min(8192u, cdiv((size_t) pbc * 32, 256))), 256, 0, s>>>(g, &ctx->d_status[ST_CNT_P], &ctx->d_status[ST_CNT_N], &ctx->d_status[ST_CNT_E], &ctx->d_status[ST_NBC], P.table, P.keys, ctx->d_status);
	{
		int rc0 = read_status(ctx);
		if(rc0) return rc0;
		rc0 = check_status(ctx);
		if(rc0) return rc0;
		const int ebc0 = ctx->h_status[ST_CNT_P] + ctx->h_status[ST_CNT_N] + ctx->h_status[ST_CNT_E];
		ctx->h_status[ST_EBC] = ebc0;
		ctx->pbc_prev		  = pbc;
		HIP_TRY(hipMemcpyAsync(&ctx->d_status[ST_EBC], &ctx->h_status[ST_EBC], sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(&ctx->d_status[ST_PBCPREV], &ctx->pbc_prev, sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(P.count, &ctx->h_status[ST_EBC], sizeof(int), hipMemcpyHostToDevice, s));
		HIP_TRY(hipStreamSynchronize(s));
	}
	ctx->nbc = ctx->h_status[ST_NBC];
	ctx->ebc = ctx->h_status[ST_EBC];
	// final capacity (reference: compile-time G_MAX_ACTIVE_BLOCK, grown by 1.5x on demand, gmpm_simulator.cuh:283-300):
	// 3/2 of the exterior blocks + slack, i.e. below the 3/4 fill level at which check_capacity() grows it
	size_t cap = ctx->cfg.max_blocks > 0 ? (size_t) ctx->cfg.max_blocks : std::min(table, (size_t) ctx->ebc * 3 / 2 + 4096);
	if(cap < (size_t) ctx->ebc) return fail(ctx, MPM_ERR_CAPACITY, "Too much exterior blocks: " + std::to_string(ctx->ebc) + " (max_blocks " + std::to_string(cap) + ")");
	{
		int* keys = nullptr;
		HIP_TRY(dalloc(&keys, 3 * cap));
		HIP_TRY(hipMemcpyAsync(keys, P.keys, sizeof(int) * 3 * (size_t) ctx->ebc, hipMemcpyDeviceToDevice, s));
		HIP_TRY(hipStreamSynchronize(s));
		HIP_TRY(hipFree(P.keys));
		P.keys = keys;
		HIP_TRY(dalloc(&ctx->part[r].keys, 3 * cap));
	}
	g.cap = (int) cap;
	for(int i = 0; i < 2; ++i) {
		HIP_TRY(dalloc(&ctx->grid[i], cap * 256));
	}
	// per model: bucket particles per block, allocate bins, fill bins + identity advection records
	for(size_t mi = 0; mi < ctx->models.size(); ++mi) {
		Model& m  = ctx->models[mi];
		m.bin_cap = m.n / kBin + cap + 1;
		for(int i = 0; i < 2; ++i) {
			HIP_TRY(dalloc(&m.bins[i], m.bin_cap * m.nch * kBin));
			HIP_TRY(dalloc(&m.binoff[i], cap + 1));
			HIP_TRY(dalloc(&m.list[i], cap * (size_t) g.ppb));
		}
		HIP_TRY(dalloc(&m.size, cap + 1));
		HIP_TRY(dalloc(&m.row_of, cap + 1));
		HIP_TRY(dalloc(&m.out_count, cap + 1));
		HIP_TRY(dalloc(&m.keep, cap + 1));
		HIP_TRY(hipMemsetAsync(m.keep, 0xff, sizeof(int) * (cap + 1), s));
		HIP_TRY(dalloc(&m.blockinfo, cap * (size_t) kInfoRow));
		m.pair = ((pair_mask() >> m.material) & 1) != 0 && g.ppb <= kPairChunks * kListChunk;
		if(m.pair)
			for(int i = 0; i < 2; ++i) HIP_TRY(dalloc(&m.pairinfo[i], (cap + 1) * (size_t) kPairChunks));
		HIP_TRY(hipMemsetAsync(m.out_count, 0, sizeof(int) * (cap + 1), s));
		HIP_TRY(hipMemsetAsync(m.size, 0, sizeof(int) * (cap + 1), s));
		if(m.n) bucket_particles_kernel<<<cdiv(m.n, 256), 256, 0, s>>>(g, m.n, m.d_xyz, P.table, m.out_count, m.list[1], ctx->d_status);
		if(pbc) {
			init_bins_kernel<<<cdiv(pbc, 256), 256, 0, s>>>(pbc, m.out_count, m.size, m.row_of, m.binoff[r], m.binoff[n], &ctx->d_status[ST_BINS0 + mi]);
			fill_bins_kernel<<<pbc, 256, 0, s>>>(g, m.nch, m.mc.log_jp0, m.d_xyz, m.list[1], m.size, m.binoff[r], m.bins[r], m.list[0]);
		}
		m.list_in = 0;
	}
	int rc = read_status(ctx);// bin totals of the models
	if(rc) return rc;
	rc = check_status(ctx);
	if(rc) return rc;
	// a block that STARTS with more particles than it has list slots is a configuration error whatever the overflow policy (the drop
	// policy covers particles that arrive at run time: a dropped particle must not have been rasterised)
	if(ctx->h_status[ST_OVERFLOW] & 2) return fail(ctx, MPM_ERR_CAPACITY, "a block holds more than max_ppc*64 = " + std::to_string(g.ppb) + " particles at set-up");
	if(ctx->h_status[ST_LOST]) return fail(ctx, MPM_ERR_INVALID, "particles outside the domain at setup");
	for(size_t mi = 0; mi < ctx->models.size(); ++mi) {
		ctx->models[mi].bincount = ctx->h_status[ST_BINS0 + mi];
		ctx->models[mi].bincount_src = ctx->models[mi].bincount;
		ctx->models[mi].bucketed = (int64_t) ctx->models[mi].n;
		if((size_t) ctx->models[mi].bincount > ctx->models[mi].bin_cap) return fail(ctx, MPM_ERR_CAPACITY, "bin capacity");
	}
	// copy the partition to the other roll (gmpm_simulator.cuh:745-748): previous numbering == current numbering
	HIP_TRY(hipMemcpyAsync(ctx->part[r].table, P.table, sizeof(int) * table, hipMemcpyDeviceToDevice, s));
	HIP_TRY(hipMemcpyAsync(ctx->part[r].keys, P.keys, sizeof(int) * 3 * (size_t) ctx->ebc, hipMemcpyDeviceToDevice, s));
	HIP_TRY(hipMemcpyAsync(ctx->part[r].count, P.count, sizeof(int), hipMemcpyDeviceToDevice, s));
	// rasterize (gmpm_simulator.cuh:763-771)
	HIP_TRY(hipMemsetAsync(ctx->grid[0], 0, sizeof(float) * 256 * (size_t) ctx->nbc, s));
	for(auto& m: ctx->models)// (m.list[1] still holds the particle ids by block that bucket_particles_kernel wrote; every particle of the model is in a block: ST_LOST was checked above)
		if(m.n && pbc) rasterize_blocks_kernel<<<pbc, 256, 0, s>>>(g, m.d_xyz, m.list[1], m.size, ctx->part[r].keys, ctx->part[r].table, ctx->grid[0], m.mc.mass, m.v0[0], m.v0[1], m.v0[2]);
	// the first G2P2G: current == previous numbering (roll r), lists as filled above
	rc = launch_prepare(ctx, r, n, false, r, &ctx->d_status[ST_PBC], ctx->pbc, true, false);
	if(rc) return rc;
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s));
	ctx->ready = true;
	return MPM_OK;
}

// grid-update phase, gmpm_simulator.cuh:326-347
static int launch_grid_update(mpm_ctx* ctx, float dt) {
	hipStream_t s = ctx->s_compute;
	if(ctx->grid_preupdated) {
		ctx->grid_preupdated = false;
		if(dt != ctx->preupdate_dt) return fail(ctx, MPM_ERR_INVALID, "grid was updated for another dt");
		return MPM_OK;
	}
	HIP_TRY(hipMemsetAsync(ctx->d_maxvel, 0, sizeof(unsigned) * kMaxVelSlots * kMaxVelStride, s));
	if(ctx->nbc) {
		const int est = std::min(ctx->g.cap, ctx->nbc + ctx->nbc / 16 + 64);// (an estimate between two synchronisations of mpm_run_fixed)
		if(ctx->has_collision)
			grid_update_collision_kernel<<<cdiv(est, 4), 256, 0, s>>>(ctx->g, &ctx->d_status[ST_NBC], ctx->grid[0], ctx->part[ctx->rollid].keys, dt, ctx->collision, ctx->d_maxvel);
		else
			grid_update_kernel<<<cdiv(est, 16), 256, 0, s>>>(ctx->g, &ctx->d_status[ST_NBC], ctx->grid[0], ctx->part[ctx->rollid].keys, dt, ctx->d_maxvel);
	}
	return MPM_OK;
}

int mpm_grid_update(mpm_ctx* ctx, float dt, float* max_vel_sqr) {
	if(!ctx || !ctx->ready) return MPM_ERR_NOT_READY;
	HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t s = ctx->s_compute;
	HIP_TRY(hipEventRecord(ctx->ev_a, s));
	int rc = launch_grid_update(ctx, dt);
	if(rc) return rc;
	HIP_TRY(hipEventRecord(ctx->ev_b, s));
	HIP_TRY(hipMemcpyAsync(ctx->h_maxvel, ctx->d_maxvel, sizeof(float) * kMaxVelSlots * kMaxVelStride, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	HIP_TRY(hipEventElapsedTime(&ctx->timers.grid_update_ms, ctx->ev_a, ctx->ev_b));
	if(max_vel_sqr) *max_vel_sqr = host_maxvel(ctx);
	return MPM_OK;
}

float mpm_compute_dt(const mpm_ctx* ctx, float max_vel, float cur_time, float next_time, float dt_default) {
	// utility_funcs.hpp:36-49
	float dt = dt_default;
	if(max_vel > 0.0f) dt = std::min(ctx->g.dx * ctx->cfg.cfl / max_vel, dt);
	return std::min(dt, next_time - cur_time);
}

static ModelView make_view(mpm_ctx* ctx, Model& m) {
	const int r = ctx->rollid, n = r ^ 1;
	ModelView v {};
	v.bins_src	 = m.bins[r];
	v.bins_dst	 = m.bins[n];
	v.binoff_src = m.binoff[r];
	v.binoff_dst = m.binoff[n];
	v.list_in	 = m.list[m.list_in];
	v.list_out	 = m.list[m.list_in ^ 1];
	v.size		 = m.size;
	v.row_of	 = m.row_of;
	v.out_count	 = m.out_count;
	v.keep		 = m.keep;
	v.blockinfo	 = m.blockinfo;
	v.pairinfo_in  = m.pairinfo[0];
	v.pairinfo_out = m.pairinfo[1];
	v.mc		 = m.mc;
	return v;
}

// a launch size for `n` (an estimate that may be a few substeps old) particle blocks: margin for growth, a multiple of 8 (XCDs)
static inline int hint_blocks(const mpm_ctx* ctx, int n) {
	const long long h = (long long) n + n / 16 + 64;
	return (int) ((std::min<long long>(h, std::max(ctx->g.cap, 8)) + 7) & ~7ll);
}

// nblocks_ptr != nullptr: the block count is read from device memory, `nblocks` is the host's estimate of it (launch size);
// otherwise `nblocks` is exact (the halo / interior lists of the MGSP path, whose lengths the host has read back)
static void launch_g2p2g_model(mpm_ctx* ctx, Model& m, const int* block_list, const int* nblocks_ptr, int nblocks, float dt, float next_dt, hipStream_t s, const int* only_flag = nullptr) {
	const int r = ctx->rollid;
	ModelView v = make_view(ctx, m);
	const int* cur_keys	  = ctx->part[r].keys;
	const GridCfg& g = ctx->g;
	StepConst sk;
	sk.dtp	= dt * g.dx_inv;
	sk.dts	= dt * (4.f * g.dx_inv);
	sk.pred = next_dt * g.dx_inv;
	sk.am	= m.mc.mass * g.dx * g.dx * g.d_inv;
	const float cs = -next_dt * g.d_inv * g.dx;// the stress enters the P2G payload as -P F^T vol new_dt D^-1 dx (:850)
	sk.ss	= StressScale {2.f * m.mc.mu * m.mc.volume * cs, m.mc.lambda * m.mc.volume * cs, m.mc.volume * cs};
	sk.refl_lim = sk.dts > 0.f ? (1.f / 3.f) / sk.dts : 3.0e38f;
	sk.jdiv		= g.dx * dt * g.d_inv;
	sk.jvisc	= g.dx * g.d_inv * m.mc.viscosity;
	const int nwg = nblocks_ptr ? hint_blocks(ctx, nblocks) : nblocks;
	// two particles per lane (mpm_g2p2g_pair.hpp) for the materials of pair_mask(); the others one particle per lane
#define MPM_LAUNCH_PAIR(M)                                                                                                                                                                      \
	if constexpr((MPM_PAIR_BUILD >> M) & 1) {                                                                                                                                                   \
		if(m.pair) {                                                                                                                                                                            \
			g2p2g_pair_kernel<M><<<nwg, kG2P2GThreads, 0, s>>>(ctx->g, v, cur_keys, ctx->grid[0], ctx->grid[1], block_list, only_flag, nblocks_ptr, nblocks, dt, next_dt, sk, ctx->d_status); \
			return;                                                                                                                                                                             \
		}                                                                                                                                                                                       \
	}
	switch(m.material) {
		case MPM_J_FLUID: MPM_LAUNCH_PAIR(0) break;
		case MPM_FIXED_COROTATED: MPM_LAUNCH_PAIR(1) break;
		case MPM_SAND: MPM_LAUNCH_PAIR(2) break;
		default: MPM_LAUNCH_PAIR(3) break;
	}
#undef MPM_LAUNCH_PAIR
	switch(m.material) {
		case MPM_J_FLUID: g2p2g_kernel<0><<<nwg, kG2P2GThreads, 0, s>>>(ctx->g, v, cur_keys, ctx->grid[0], ctx->grid[1], block_list, only_flag, nblocks_ptr, nblocks, dt, next_dt, sk, ctx->d_status); break;
		case MPM_FIXED_COROTATED: g2p2g_kernel<1><<<nwg, kG2P2GThreads, 0, s>>>(ctx->g, v, cur_keys, ctx->grid[0], ctx->grid[1], block_list, only_flag, nblocks_ptr, nblocks, dt, next_dt, sk, ctx->d_status); break;
		case MPM_SAND: g2p2g_kernel<2><<<nwg, kG2P2GThreads, 0, s>>>(ctx->g, v, cur_keys, ctx->grid[0], ctx->grid[1], block_list, only_flag, nblocks_ptr, nblocks, dt, next_dt, sk, ctx->d_status); break;
		default: g2p2g_kernel<3><<<nwg, kG2P2GThreads, 0, s>>>(ctx->g, v, cur_keys, ctx->grid[0], ctx->grid[1], block_list, only_flag, nblocks_ptr, nblocks, dt, next_dt, sk, ctx->d_status); break;
	}
}

// substep_clear_kernel: the P2G part precedes G2P2G (clear_grid + the bucket counters, gmpm_simulator.cuh:383,:389), the rebuild
// part precedes the rebuild (reset_table + the counters, :436-446); with_rebuild issues both in one launch
static int launch_clear(mpm_ctx* ctx, int flags) {
	if((flags & kClearRebuild) && !(flags & kClearP2G)) {// called by the rebuild itself
		if(ctx->rebuild_cleared) flags &= ~(kClearRebuild | kClearMaxVel);// (already issued with this substep's P2G part)
		ctx->rebuild_cleared = false;
	}
	if(!(flags & (kClearP2G | kClearRebuild))) return MPM_OK;
	ClearArgs a {};
	a.flags	  = flags;
	a.nmodels = (int) ctx->models.size();
	for(int mi = 0; mi < a.nmodels; ++mi) a.out_count[mi] = ctx->models[mi].out_count, a.keep[mi] = ctx->models[mi].keep;
	a.p2g_grid	   = ctx->grid[1];
	a.status	   = ctx->d_status;
	a.max_vel_bits = ctx->d_maxvel;
	Partition& Pn  = ctx->part[ctx->rollid ^ 1];
	a.old_table	   = Pn.table;
	a.old_keys	   = Pn.keys;
	a.old_count	   = Pn.count;
	substep_clear_kernel<<<1024, 256, 0, ctx->s_compute>>>(ctx->g, a);
	return MPM_OK;
}
// with_rebuild: the rebuild's clear rides along (nobody looks at the old table before the rebuild); fused_update: that rebuild's
// carry-over applies the next grid update, i.e. writes the max |v|^2 slots
static int launch_g2p2g_prologue(mpm_ctx* ctx, bool with_rebuild = false, bool fused_update = false) {
	int rc = launch_clear(ctx, with_rebuild ? (kClearP2G | kClearRebuild | (fused_update ? kClearMaxVel : 0)) : kClearP2G);
	if(rc == MPM_OK && with_rebuild) ctx->rebuild_cleared = true;
	return rc;
}

static int launch_g2p2g(mpm_ctx* ctx, float dt, float next_dt, hipEvent_t e0, hipEvent_t e1, bool with_rebuild_clear = false, bool fused_update = false) {
	hipStream_t s = ctx->s_compute;
	int rc		  = launch_g2p2g_prologue(ctx, with_rebuild_clear, fused_update);
	if(rc) return rc;
	HIP_TRY(hipEventRecord(e0, s));
	if(ctx->pbc)
		for(auto& m: ctx->models) launch_g2p2g_model(ctx, m, nullptr, &ctx->d_status[ST_PBC], ctx->pbc, dt, next_dt, s);
	HIP_TRY(hipEventRecord(e1, s));
	return MPM_OK;
}

int mpm_g2p2g(mpm_ctx* ctx, float dt, float next_dt) {
	if(!ctx || !ctx->ready) return MPM_ERR_NOT_READY;
	HIP_TRY(hipSetDevice(ctx->device));
	int rc = launch_g2p2g(ctx, dt, next_dt, ctx->ev_g0, ctx->ev_g1);
	if(rc) return rc;
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(ctx->s_compute));
	HIP_TRY(hipEventElapsedTime(&ctx->last_g2p2g_ms, ctx->ev_g0, ctx->ev_g1));
	ctx->timers.g2p2g_ms = ctx->last_g2p2g_ms;
	return MPM_OK;
}

// Per-block preparation of the next G2P2G (prepare_blocks_kernel): `cur` is the numbering that G2P2G will run in, `prev`
// the numbering the particle data is laid out in; `list_sel` / `binoff_sel` select the list and bin-offset buffers that
// G2P2G will read (they differ before and after the roll).  nblocks_est: the host's estimate of the particle block count
// *pbc_ptr (the launch size; the kernel walks over what is beyond it).  publish: the rebuild's last kernel publishes the
// exterior block count.
static int launch_prepare(mpm_ctx* ctx, int cur, int prev, bool list_is_out, int binoff_sel, const int* pbc_ptr, int nblocks_est, bool sort, bool publish) {
	if(nblocks_est <= 0 && !publish) return MPM_OK;
	PrepareModels pm {};
	pm.n = (int) ctx->models.size();
	for(int mi = 0; mi < pm.n; ++mi) {
		Model& m		  = ctx->models[mi];
		pm.list[mi]		  = m.list[list_is_out ? (m.list_in ^ 1) : m.list_in];
		pm.size[mi]		  = m.size;
		pm.row_of[mi]	  = m.row_of;
		pm.binoff_src[mi] = m.binoff[binoff_sel];
		pm.blockinfo[mi]  = m.blockinfo;
		pm.keep[mi]		  = sort && list_is_out ? m.keep : nullptr;// (only the rebuild's call sorts lists G2P2G has just written)
		pm.pairinfo[mi]	  = m.pair ? m.pairinfo[0] : nullptr;
		pm.pairhand[mi]	  = m.pairinfo[1];
	}
	const int nwg = std::max(1, std::min(ctx->g.cap, nblocks_est + nblocks_est / 16 + 64));
	int* pub	  = publish ? ctx->d_status : nullptr;
	if(sort)
		prepare_blocks_kernel<true><<<nwg, 64, 0, ctx->s_compute>>>(ctx->g, pm, pbc_ptr, ctx->part[cur].table, ctx->part[cur].keys, ctx->part[prev].table, pub, ctx->part[cur].count);
	else
		prepare_blocks_kernel<false><<<nwg, 64, 0, ctx->s_compute>>>(ctx->g, pm, pbc_ptr, ctx->part[cur].table, ctx->part[cur].keys, ctx->part[prev].table, pub, ctx->part[cur].count);
	return MPM_OK;
}

// partition rebuild, gmpm_simulator.cuh:415-579 (launches only: no host round trip, no runtime fill / copy commands; launch sizes
// from the host's estimates of the block counts, true counts from the status block)
// fuse_dt > 0: the carry-over applies the grid update of the next substep (dt = fuse_dt) as well
// without_prepare: everything but the last kernel (prepare_blocks_kernel: list sort, look-up rows, publication of the exterior block count), which
// launch_rebuild_prepare issues - the group loop puts the tagging's key export in between, so that the key all-gather and the tagging kernels
// run on the comm stream beside it
static int launch_rebuild_prepare(mpm_ctx* ctx) {
	const int r = ctx->rollid, n = r ^ 1;
	const int ebc_est = std::min(ctx->g.cap, ctx->ebc + ctx->ebc / 16 + 1024);
	return launch_prepare(ctx, n, r, true, n, &ctx->d_status[ST_CNT_P], ebc_est, true, true);
}
static int launch_rebuild(mpm_ctx* ctx, float fuse_dt = 0.f, bool without_prepare = false) {
	if(fuse_dt == 0.f) fuse_dt = ctx->fuse_dt_once;
	ctx->fuse_dt_once = 0.f;
	hipStream_t s = ctx->s_compute;
	GridCfg& g	  = ctx->g;
	const int r = ctx->rollid, n = r ^ 1;
	Partition& Pn = ctx->part[n];
	Partition& Pr = ctx->part[r];
	const bool fused = fuse_dt > 0.f && !ctx->has_collision;
	int rc			 = launch_clear(ctx, kClearRebuild | (fused ? kClearMaxVel : 0));// un-insert the old keys of Pn (reset_table, hash_table.cuh:110-112), counters, totals
	if(rc) return rc;
	RebuildModels rm {};
	rm.n = (int) ctx->models.size();
	for(int mi = 0; mi < rm.n; ++mi) {
		Model& m		 = ctx->models[mi];
		rm.out_count[mi] = m.out_count;
		rm.size[mi]		 = m.size;
		rm.row_of[mi]	 = m.row_of;
		rm.binoff[mi]	 = m.binoff[r];// becomes the destination offsets after the roll
		rm.bin_cap[mi]	 = (long long) m.bin_cap;
	}
	int* st			  = ctx->d_status;
	const int ebc_est = std::min(g.cap, ctx->ebc + ctx->ebc / 16 + 1024);
	compact_blocks_kernel<<<std::max(1u, cdiv(ebc_est, 1024)), 1024, 0, s>>>(g, rm, Pr.keys, Pn.keys, Pn.table, st);
	const unsigned rg8 = std::max(1u, std::min(4096u, cdiv((size_t) ebc_est * 8, 256))), rg32 = std::max(1u, std::min(8192u, cdiv((size_t) ebc_est * 32, 256)));
	register_blocks_kernel<0, 1><<<rg8, 256, 0, s>>>(g, &st[ST_CNT_P], nullptr, &st[ST_CNT_N], &st[ST_PBC], Pn.table, Pn.keys, st);
	if(fused) {
		carry_grid_kernel<true><<<2048, 256, 0, s>>>(g, st, Pn.keys, Pr.table, ctx->grid[1], ctx->grid[0], fuse_dt, ctx->d_maxvel);
		ctx->grid_preupdated = true;
		ctx->preupdate_dt	 = fuse_dt;
	} else
		carry_grid_kernel<false><<<2048, 256, 0, s>>>(g, st, Pn.keys, Pr.table, ctx->grid[1], ctx->grid[0], 0.f, nullptr);
	register_blocks_kernel<-1, 1><<<rg32, 256, 0, s>>>(g, &st[ST_CNT_P], &st[ST_CNT_N], &st[ST_CNT_E], &st[ST_NBC], Pn.table, Pn.keys, st);
	// the next G2P2G runs in the new numbering n with the particle data laid out in r: sort its lists (the ones the last
	// G2P2G appended to), look up its blocks' neighbours; this last kernel also publishes the exterior block count
	if(without_prepare) return MPM_OK;
	return launch_prepare(ctx, n, r, true, n, &st[ST_CNT_P], ebc_est, true, true);
}

// check_capacity() (gmpm_simulator.cuh:283-300): once the exterior block count / a model's bin count passes 3/4 of its
// capacity the capacity grows by 3/2.  Runs at a host synchronisation, after a rebuild; every
// array indexed by block number (keys, grids, lists, sizes, bin offsets, halo lists) or by bin is reallocated and
// copied.  The dense table is sized by the domain, not by the capacity.
static int grow_capacity(mpm_ctx* ctx) {
	if(!ctx->cfg.grow) return MPM_OK;
	hipStream_t s	   = ctx->s_compute;
	const size_t table = (size_t) ctx->g.G * ctx->g.G * ctx->g.G;
	const size_t cap   = (size_t) ctx->g.cap;
	if((size_t) ctx->ebc * 4 > cap * 3 && cap < table) {
		const size_t ncap = std::min(table, cap * 3 / 2 + 1);
		HIP_TRY(hipDeviceSynchronize());// both streams idle: buffers are about to be replaced
		for(int i = 0; i < 2; ++i) {
			HIP_TRY(regrow(&ctx->part[i].keys, 3 * cap, 3 * ncap, s));
			HIP_TRY(regrow(&ctx->grid[i], cap * 256, ncap * 256, s));
		}
		for(auto& m: ctx->models) {
			for(int i = 0; i < 2; ++i) {
				HIP_TRY(regrow(&m.binoff[i], cap + 1, ncap + 1, s));
				HIP_TRY(regrow(&m.list[i], cap * (size_t) ctx->g.ppb, ncap * (size_t) ctx->g.ppb, s));
			}
			HIP_TRY(regrow(&m.size, cap + 1, ncap + 1, s));
			HIP_TRY(regrow(&m.row_of, cap + 1, ncap + 1, s));
			HIP_TRY(regrow(&m.out_count, cap + 1, ncap + 1, s));
			HIP_TRY(regrow(&m.keep, cap + 1, ncap + 1, s));
			HIP_TRY(regrow(&m.blockinfo, cap * (size_t) kInfoRow, ncap * (size_t) kInfoRow, s));
			if(m.pair)
				for(int i = 0; i < 2; ++i) HIP_TRY(regrow(&m.pairinfo[i], (cap + 1) * (size_t) kPairChunks, (ncap + 1) * (size_t) kPairChunks, s));
		}
		if(ctx->d_overlap) {
			HIP_TRY(regrow(&ctx->d_overlap, cap + 1, ncap + 1, s));
			HIP_TRY(regrow(&ctx->d_halo_list, cap + 1, ncap + 1, s));
			HIP_TRY(regrow(&ctx->d_inner_list, cap + 1, ncap + 1, s));
		}
		for(int p = 0; p < 32; ++p)
			if(ctx->d_send_ids[p]) HIP_TRY(regrow(&ctx->d_send_ids[p], cap + 1, ncap + 1, s));
		ctx->g.cap = (int) ncap;
		ctx->capacity_events++;
	}
	for(auto& m: ctx->models) {
		// worst case: every block ends with one partially filled bin
		const size_t need = m.n / kBin + (size_t) ctx->g.cap + 1;
		if((size_t) m.bincount * 4 > m.bin_cap * 3 || need > m.bin_cap) {
			const size_t nb = std::max(need, (size_t) m.bincount * 4 > m.bin_cap * 3 ? m.bin_cap * 3 / 2 + 1 : m.bin_cap);
			HIP_TRY(hipDeviceSynchronize());
			for(int i = 0; i < 2; ++i) HIP_TRY(regrow(&m.bins[i], m.bin_cap * m.nch * kBin, nb * m.nch * kBin, s));
			m.bin_cap = nb;
			ctx->capacity_events++;
		}
	}
	return MPM_OK;
}

// the host's half of a rebuild that needs no device data: the roll (gmpm_simulator.cuh:578)
static void roll_partition(mpm_ctx* ctx) {
	for(auto& m: ctx->models) m.list_in ^= 1;
	ctx->rollid ^= 1;
}

// The library checks its own books at every host synchronisation (the reference prints the particle total per frame and leaves the
// reading to the user, gmpm_simulator.cuh:617): every particle that was added is either bucketed in a block of the partition just
// built, or was counted as lost (left the domain / a block nobody registered, particle_buffer.cuh:105-113) or as dropped (overflow
// policy).  The lost / dropped counters are per context, so the per-model statement is made when both are zero - the normal case -
// and the totals are compared otherwise.  A failure is a bug of the engine (round 4's windowed group loop skipped a launch and lost
// the particles of a few blocks with nothing counted as lost), never of the caller: MPM_ERR_INTERNAL.
static int check_books(mpm_ctx* ctx) {
	// (books_bias: what a loaded checkpoint's state was missing beyond this context's own counters - zero otherwise)
	const long long gone = (long long) ctx->h_status[ST_LOST] + (long long) ctx->h_status[ST_DROPPED] + ctx->books_bias;
	long long added = 0, held = 0;
	for(size_t mi = 0; mi < ctx->models.size(); ++mi) {
		const Model& m = ctx->models[mi];
		added += (long long) m.n;
		held += (long long) m.bucketed;
		if(gone == 0 && (long long) m.bucketed != (long long) m.n)
			return fail(ctx, MPM_ERR_INTERNAL,
						"particle books of model " + std::to_string(mi) + " do not balance: " + std::to_string(m.bucketed) + " bucketed of " + std::to_string(m.n) +
							" added, none counted as lost or dropped");
	}
	if(held + gone != added)
		return fail(ctx, MPM_ERR_INTERNAL,
					"particle books do not balance: " + std::to_string(held) + " bucketed + " + std::to_string(ctx->h_status[ST_LOST]) + " lost + " + std::to_string(ctx->h_status[ST_DROPPED]) +
						" dropped != " + std::to_string(added) + " added");
	return MPM_OK;
}

// Host synchronisation after one or more enqueued substeps (every one already rolled on the host): read the status block,
// report what went wrong in the meantime (flags are sticky), take over the counts, grow capacities.
static int apply_status(mpm_ctx* ctx, mpm_counts* counts);
static int sync_counts(mpm_ctx* ctx, mpm_counts* counts) {
	int rc = read_status(ctx);
	if(rc) return rc;
	return apply_status(ctx, counts);
}
// the host's half of a synchronisation, once the status block is in h_status
static int apply_status(mpm_ctx* ctx, mpm_counts* counts) {
	int rc = check_status(ctx);
	if(rc) return rc;
	ctx->pbc = ctx->h_status[ST_PBC];
	ctx->nbc = ctx->h_status[ST_NBC];
	ctx->ebc = ctx->h_status[ST_EBC];
	ctx->pbc_prev = ctx->h_status[ST_PBCPREV];
	if(ctx->ebc > ctx->g.cap) return fail(ctx, MPM_ERR_CAPACITY, "Too much exterior blocks: " + std::to_string(ctx->ebc));
	for(size_t mi = 0; mi < ctx->models.size(); ++mi) {
		Model& m	   = ctx->models[mi];
		m.bincount_src = ctx->h_status[ST_BINSPREV + mi];
		m.bincount	   = ctx->h_status[ST_BINS0 + mi];
		m.bucketed	   = ctx->h_status[ST_PART0 + mi];
		if((size_t) m.bincount > m.bin_cap) return fail(ctx, MPM_ERR_CAPACITY, "bin capacity exceeded");
	}
	if(ctx->h_status[ST_NONFINITE]) return fail(ctx, MPM_ERR_NONFINITE, "Maximum velocity is infinity");// gmpm_simulator.cuh:355-358
	rc = check_books(ctx);
	if(rc) return rc;
	rc = grow_capacity(ctx);
	if(rc) return rc;
	if(counts) return mpm_get_counts(ctx, counts);
	return MPM_OK;
}
static int finish_rebuild(mpm_ctx* ctx, mpm_counts* counts) {
	roll_partition(ctx);
	return sync_counts(ctx, counts);
}

int mpm_rebuild_partition(mpm_ctx* ctx, mpm_counts* counts) {
	if(!ctx || !ctx->ready) return MPM_ERR_NOT_READY;
	HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t s = ctx->s_compute;
	HIP_TRY(hipEventRecord(ctx->ev_a, s));
	int rc = launch_rebuild(ctx);
	if(rc) return rc;
	HIP_TRY(hipEventRecord(ctx->ev_b, s));
	HIP_TRY(hipGetLastError());
	rc = finish_rebuild(ctx, counts);
	if(rc) return rc;
	HIP_TRY(hipEventElapsedTime(&ctx->timers.partition_ms, ctx->ev_a, ctx->ev_b));
	return MPM_OK;
}

int mpm_substep(mpm_ctx* ctx, float dt, float step_time, float frame_time, float dt_default, float* next_dt, float* max_vel) {
	if(!ctx || !ctx->ready) return MPM_ERR_NOT_READY;
	FlagGuard guard {ctx};
	ctx->fuse_dt_once = 0.f;// (this path keeps the grid update a kernel of its own: the rebuild below must not apply one)
	float mv2 = 0.f;
	int rc	  = mpm_grid_update(ctx, dt, &mv2);
	if(rc) return rc;
	if(std::isinf(mv2)) return fail(ctx, MPM_ERR_NONFINITE, "Maximum velocity is infinity");// gmpm_simulator.cuh:355-358
	const float mv = std::sqrt(mv2);																 // :360
	const float nd = mpm_compute_dt(ctx, mv, step_time, frame_time, dt_default);
	if(max_vel) *max_vel = mv;
	if(next_dt) *next_dt = nd;
	rc = launch_g2p2g(ctx, dt, nd, ctx->ev_g0, ctx->ev_g1, true);
	if(rc) return rc;
	hipStream_t s = ctx->s_compute;
	HIP_TRY(hipEventRecord(ctx->ev_a, s));
	rc = launch_rebuild(ctx);
	if(rc) return rc;
	HIP_TRY(hipEventRecord(ctx->ev_b, s));
	HIP_TRY(hipGetLastError());
	rc = finish_rebuild(ctx, nullptr);
	if(rc) return rc;
	HIP_TRY(hipEventElapsedTime(&ctx->last_g2p2g_ms, ctx->ev_g0, ctx->ev_g1));
	HIP_TRY(hipEventElapsedTime(&ctx->timers.partition_ms, ctx->ev_a, ctx->ev_b));
	ctx->timers.g2p2g_ms = ctx->last_g2p2g_ms;
	ctx->timers.total_ms = ctx->timers.grid_update_ms + ctx->timers.g2p2g_ms + ctx->timers.partition_ms;
	guard.armed = false;
	return MPM_OK;
}

// The substep loop with fixed dt.  The host enqueues up to mpm_config.sync_interval substeps (default 8) back to back and only
// then synchronises: every kernel reads its block counts from the status block, launches are sized by the counts of the last
// synchronisation plus a margin (a kernel walks over what is beyond its launch), errors raised in between are sticky flags.
// The reference synchronises the host six times per substep (gmpm_simulator.cuh:398,:470,:503,:518,:541,:564).
int mpm_run_fixed(mpm_ctx* ctx, int nsteps, float dt) {
	if(!ctx || !ctx->ready) return MPM_ERR_NOT_READY;
	HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t s = ctx->s_compute;
	const int K = ctx->cfg.sync_interval > 0 ? std::min(ctx->cfg.sync_interval, 64) : 8;// (clamped to 64: claymore_amd.h)
	FlagGuard guard {ctx};
	while((int) ctx->ev_ring.size() < 3 * K + 1) {
		hipEvent_t e = nullptr;
		HIP_TRY(hipEventCreate(&e));
		ctx->ev_ring.push_back(e);
	}
	{// the counts are exact here: a context that starts a run above the 3/4 mark grows before its first window, not after it
		int rc = grow_capacity(ctx);
		if(rc) return rc;
	}
	double acc_grid = 0, acc_g2p2g = 0, acc_part = 0, acc_total = 0;
	int in_window = 0;
	for(int it = 0; it < nsteps; ++it) {
		// three events per substep - start, G2P2G start, G2P2G end -; its end is the next substep's start (or the window's closing event): an event between two
		// kernels is a barrier packet of its own, 5-6 us on the stream (profiles/r06_rank_alone_seq.txt), a tenth of a 0.2 ms substep when there were four
		hipEvent_t* ev = &ctx->ev_ring[3 * in_window];
		if(in_window == 0) HIP_TRY(hipEventRecord(ev[0], s));
		int rc = launch_grid_update(ctx, dt);
		if(rc) return rc;
		const bool fuse_next = it + 1 < nsteps;// the last substep leaves the canonical state behind
		rc = launch_g2p2g(ctx, dt, dt, ev[1], ev[2], true, fuse_next && !ctx->has_collision);
		if(rc) return rc;
		rc = launch_rebuild(ctx, fuse_next ? dt : 0.f);
		if(rc) return rc;
		HIP_TRY(hipEventRecord(ev[3], s));
		roll_partition(ctx);
		++in_window;
		// Capacities grow only at a synchronisation (by 3/2 per look, at 3/4 fill): a context whose blocks fill more than 7/10 of the capacity
		// (set-up sizes it to 2/3) looks after EVERY substep - what the reference's check_capacity does (gmpm_simulator.cuh:283-300) -,
		// so that a burst of new blocks cannot outrun the growth inside a window
		const bool tight = ctx->cfg.grow && (long long) ctx->ebc * 10 > (long long) ctx->g.cap * 7 &&
						   (size_t) ctx->g.cap < (size_t) ctx->g.G * ctx->g.G * ctx->g.G;// (a capacity that has reached the table size cannot grow: nothing to look for)
		if(in_window == K || it + 1 == nsteps || tight) {
			HIP_TRY(hipGetLastError());
			rc = sync_counts(ctx, nullptr);// (one read-back: status, halo counters, max |v|^2 slots)
			if(rc) return rc;
			if(std::isinf(host_maxvel(ctx))) return fail(ctx, MPM_ERR_NONFINITE, "Maximum velocity is infinity");
			for(int w = 0; w < in_window; ++w) {
				hipEvent_t* e = &ctx->ev_ring[3 * w];
				float t_grid = 0, t_g = 0, t_part = 0, t_tot = 0;
				HIP_TRY(hipEventElapsedTime(&t_grid, e[0], e[1]));
				HIP_TRY(hipEventElapsedTime(&t_g, e[1], e[2]));
				HIP_TRY(hipEventElapsedTime(&t_part, e[2], e[3]));
				HIP_TRY(hipEventElapsedTime(&t_tot, e[0], e[3]));
				acc_grid += t_grid;
				acc_g2p2g += t_g;
				acc_part += t_part;
				acc_total += t_tot;
				ctx->last_g2p2g_ms = t_g;
			}
			in_window = 0;
		}
	}
	if(nsteps > 0) {// per-substep averages over this call, HIP events on the compute stream
		ctx->timers.grid_update_ms = (float) (acc_grid / nsteps);
		ctx->timers.g2p2g_ms	   = (float) (acc_g2p2g / nsteps);
		ctx->timers.partition_ms   = (float) (acc_part / nsteps);
		ctx->timers.total_ms	   = (float) (acc_total / nsteps);
	}
	guard.armed = false;
	return MPM_OK;
}

int mpm_state_kind(void) {
	return MPM_STATE_B;
}

const char* mpm_build_info(void) {
	return "claymore_hip abi7 state=b"
#ifdef MPM_EXPERIMENT
		   " experiment=MPM_EXPERIMENT"
#ifdef MPM_HACK_EDGEWIN
		   ",MPM_HACK_EDGEWIN"
#endif
#ifdef MPM_HACK_NOSHELL
		   ",MPM_HACK_NOSHELL"
#endif
#ifdef MPM_HACK_STALE_INTERIOR
		   ",MPM_HACK_STALE_INTERIOR"
#endif
#ifdef MPM_HACK_NOSERIAL
		   ",MPM_HACK_NOSERIAL"
#endif
#ifdef MPM_HACK_NOWB
		   ",MPM_HACK_NOWB"
#endif
#ifdef MPM_HACK_UNDEF
		   ",MPM_HACK_UNDEF"
#endif
#ifdef MPM_LDS_PAD
		   ",MPM_LDS_PAD"
#endif
#ifdef MPM_G2P2G_STATS
		   ",MPM_G2P2G_STATS"
#endif
#ifdef MPM_G2P2G_WAVES
		   ",MPM_G2P2G_WAVES"
#endif
#ifdef MPM_G2P2G_WAVES_FLUID
		   ",MPM_G2P2G_WAVES_FLUID"
#endif
#ifdef MPM_VARIANT
		   ",MPM_VARIANT=" MPM_STR(MPM_VARIANT)
#endif
#else
		   " experiment=none"
#endif
		;
}

int mpm_last_g2p2g_ms(mpm_ctx* ctx, float* ms) {
	if(!ctx || !ms) return MPM_ERR_INVALID;
	*ms = ctx->last_g2p2g_ms;
	return MPM_OK;
}

int mpm_get_counts(mpm_ctx* ctx, mpm_counts* counts) {
	if(!ctx || !ctx->ready || !counts) return MPM_ERR_NOT_READY;
	memset(counts, 0, sizeof(*counts));
	counts->particle_blocks = ctx->pbc;
	counts->neighbor_blocks = ctx->nbc;
	counts->exterior_blocks = ctx->ebc;
	counts->model_count		= (int) ctx->models.size();
	for(size_t mi = 0; mi < ctx->models.size(); ++mi) {
		counts->bins[mi]	  = ctx->models[mi].bincount;
		counts->particles[mi] = ctx->models[mi].bucketed;
	}
	return MPM_OK;
}

int mpm_default_collision_object(mpm_collision_object* o) {
	if(!o) return MPM_ERR_INVALID;
	memset(o, 0, sizeof(*o));
	o->type		  = MPM_BOUNDARY_STICKY;// boundary_condition.cuh:43-48
	o->friction	  = 0.3f;
	o->scale	  = 1.0f;
	o->dsdt		  = 0.0f;
	o->rot_mat[0] = o->rot_mat[4] = o->rot_mat[8] = 1.f;
	return MPM_OK;
}

int mpm_set_collision_object(mpm_ctx* ctx, const mpm_collision_object* obj, const float* sdf, const float* gx, const float* gy, const float* gz) {
	if(!ctx) return MPM_ERR_INVALID;
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipDeviceSynchronize());
	if(ctx->d_sdf) HIP_TRY(hipFree(ctx->d_sdf));
	ctx->d_sdf		   = nullptr;
	ctx->has_collision = false;
	if(!obj) return MPM_OK;
	if(!sdf || !gx || !gy || !gz) return fail(ctx, MPM_ERR_INVALID, "collision object without a signed distance field");
	if(obj->type < MPM_BOUNDARY_STICKY || obj->type > MPM_BOUNDARY_SEPARATE) return fail(ctx, MPM_ERR_INVALID, "[ERROR] Wrong Boundary Type!");// boundary_condition.cuh:244
	const size_t N = (size_t) 1 << ctx->cfg.domain_bits, n = N * N * N;
	float* stage = nullptr;
	HIP_TRY(dalloc(&stage, 4 * n));
	HIP_TRY(dalloc(&ctx->d_sdf, n));
	const float* src[4] = {sdf, gx, gy, gz};
	for(int c = 0; c < 4; ++c) HIP_TRY(hipMemcpy(stage + c * n, src[c], sizeof(float) * n, hipMemcpyHostToDevice));
	pack_sdf_kernel<<<cdiv(n, 256), 256, 0, ctx->s_compute>>>(n, stage, stage + n, stage + 2 * n, stage + 3 * n, ctx->d_sdf);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(ctx->s_compute));
	HIP_TRY(hipFree(stage));
	CollisionObject& c = ctx->collision;
	c.type	   = obj->type;
	c.friction = obj->friction;
	c.scale	   = obj->scale;
	c.dsdt	   = obj->dsdt;
	for(int d = 0; d < 3; ++d) {
		c.trans[d]	   = obj->trans[d];
		c.trans_vel[d] = obj->trans_vel[d];
		c.omega[d]	   = obj->omega[d];
	}
	for(int i = 0; i < 9; ++i) c.rot[i] = obj->rot_mat[i];
	c.time			   = obj->time;
	c.field			   = ctx->d_sdf;
	ctx->has_collision = true;
	return MPM_OK;
}

int mpm_get_capacity(mpm_ctx* ctx, int64_t* block_capacity, int64_t* bin_capacity, int* growth_events) {
	if(!ctx || !ctx->ready) return MPM_ERR_NOT_READY;
	if(block_capacity) *block_capacity = ctx->g.cap;
	if(bin_capacity)
		for(size_t i = 0; i < ctx->models.size() && i < 8; ++i) bin_capacity[i] = (int64_t) ctx->models[i].bin_cap;
	if(growth_events) *growth_events = ctx->capacity_events;
	return MPM_OK;
}

int mpm_get_diagnostics(mpm_ctx* ctx, mpm_diagnostics* d) {
	if(!ctx || !ctx->ready || !d) return MPM_ERR_NOT_READY;
	memset(d, 0, sizeof(*d));
	d->lost_particles = ctx->h_status[ST_LOST];
	d->discarded_p2g  = ctx->h_status[ST_ARENA];
	d->overflow_flags = ctx->h_status[ST_OVERFLOW];
	d->dropped_particles = ctx->h_status[ST_DROPPED];
#ifdef MPM_G2P2G_STATS
	for(int i = 0; i < 4; ++i) d->reserved[i] = ctx->h_status[24 + i];// cumulative: iterations, loser lanes, edge lanes, iterations with a retry
#endif
	return MPM_OK;
}

int mpm_get_timers(mpm_ctx* ctx, mpm_timers* t) {
	if(!ctx || !t) return MPM_ERR_INVALID;
	*t = ctx->timers;
	return MPM_OK;
}

// output_model, gmpm_simulator.cuh:594-634
int mpm_retrieve_state(mpm_ctx* ctx, int model, float* xyz, float* state9, float* logjp, size_t* n) {
	if(!ctx || !ctx->ready || model < 0 || model >= (int) ctx->models.size() || !n || !xyz) return MPM_ERR_INVALID;
	HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t s = ctx->s_compute;
	Model& m	  = ctx->models[model];
	const int r = ctx->rollid, nn = r ^ 1;
	const size_t cap = std::min(*n, m.n);
	// (an output call, once per frame at most: its two staging arrays are allocated here and released on every exit path)
	DevScratch<float> b_state, b_lj;
	if(state9) HIP_TRY(b_state.alloc(9 * cap));
	if(logjp) HIP_TRY(b_lj.alloc(cap));
	float *d_state = b_state.p, *d_lj = b_lj.p;
	HIP_TRY(hipMemsetAsync(ctx->d_counter, 0, sizeof(unsigned long long), s));
	if(ctx->pbc) retrieve_kernel<<<ctx->pbc, 256, 0, s>>>(ctx->g, m.nch, ctx->part[r].keys, ctx->part[nn].table, m.size, m.row_of, m.list[m.list_in], m.binoff[r], m.bins[r], m.d_xyz, d_state, d_lj, (unsigned long long) cap, ctx->d_counter, m.pair ? 1 : 0);
	unsigned long long count = 0;
	HIP_TRY(hipMemcpyAsync(&count, ctx->d_counter, sizeof(count), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	const size_t got = std::min<size_t>(count, cap);
	HIP_TRY(hipMemcpy(xyz, m.d_xyz, sizeof(float) * 3 * got, hipMemcpyDeviceToHost));
	if(state9) HIP_TRY(hipMemcpy(state9, d_state, sizeof(float) * 9 * got, hipMemcpyDeviceToHost));
	if(logjp) HIP_TRY(hipMemcpy(logjp, d_lj, sizeof(float) * got, hipMemcpyDeviceToHost));
	*n = got;
	if(count > cap) return fail(ctx, MPM_ERR_CAPACITY, "output array too small");
	return MPM_OK;
}

int mpm_retrieve_positions(mpm_ctx* ctx, int model, float* xyz, size_t* n) {
	return mpm_retrieve_state(ctx, model, xyz, nullptr, nullptr, n);
}

int mpm_grid_totals(mpm_ctx* ctx, double out[4]) {
	if(!ctx || !ctx->ready || !out) return MPM_ERR_NOT_READY;
	HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t s = ctx->s_compute;
	HIP_TRY(hipMemsetAsync(ctx->d_totals, 0, sizeof(double) * 4, s));
	if(ctx->nbc) grid_totals_kernel<<<256, 256, 0, s>>>(ctx->nbc, ctx->grid[0], ctx->d_totals);
	HIP_TRY(hipMemcpyAsync(out, ctx->d_totals, sizeof(double) * 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	return MPM_OK;
}

int mpm_check_table(mpm_ctx* ctx) {
	if(!ctx || !ctx->ready) return -1;
	if(hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->s_compute) != hipSuccess) return -1;
	const Partition& P = ctx->part[ctx->rollid];
	const size_t table = (size_t) ctx->g.G * ctx->g.G * ctx->g.G;
	std::vector<int> t(table), k(3 * (size_t) ctx->ebc);
	if(hipMemcpy(t.data(), P.table, sizeof(int) * table, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	if(ctx->ebc && hipMemcpy(k.data(), P.keys, sizeof(int) * k.size(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
	long long bad = 0, set = 0;
	for(int v: t) set += v != -1;
	for(int i = 0; i < ctx->ebc; ++i) {
		const int x = k[3 * i], y = k[3 * i + 1], z = k[3 * i + 2];
		const bool in = x >= 0 && y >= 0 && z >= 0 && x < ctx->g.G && y < ctx->g.G && z < ctx->g.G;
		if(!in || t[((size_t) x << (2 * ctx->g.gbits)) | ((size_t) y << ctx->g.gbits) | (size_t) z] != i) ++bad;
	}
	bad += std::llabs(set - (long long) ctx->ebc);
	return (int) std::min<long long>(bad, 0x7fffffff);
}

int mpm_dump_grid(mpm_ctx* ctx, int* keys, float* blocks, size_t* nblocks) {
	if(!ctx || !ctx->ready || !nblocks) return MPM_ERR_NOT_READY;
	if(*nblocks < (size_t) ctx->nbc) return fail(ctx, MPM_ERR_CAPACITY, "grid dump too small");
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipStreamSynchronize(ctx->s_compute));
	if(keys) HIP_TRY(hipMemcpy(keys, ctx->part[ctx->rollid].keys, sizeof(int) * 3 * (size_t) ctx->nbc, hipMemcpyDeviceToHost));
	if(blocks) HIP_TRY(hipMemcpy(blocks, ctx->grid[0], sizeof(float) * 256 * (size_t) ctx->nbc, hipMemcpyDeviceToHost));
	*nblocks = ctx->nbc;
	return MPM_OK;
}

// ---- function-level tests ---------------------------------------------------------------------------
#define HIP_TRY0(expr)                         \
	do {                                       \
		if((expr) != hipSuccess) return MPM_ERR_DEVICE; \
	} while(0)

int mpm_test_eig(const float* F, size_t n, float* out12, int device) {
	if(!F || !out12 || n == 0) return MPM_ERR_INVALID;
	HIP_TRY0(hipSetDevice(device));
	DevScratch<float> dF, dO;
	HIP_TRY0(dF.alloc(9 * n));
	HIP_TRY0(dO.alloc(12 * n));
	HIP_TRY0(hipMemcpy(dF.p, F, sizeof(float) * 9 * n, hipMemcpyHostToDevice));
	test_eig_kernel<<<cdiv(n, 256), 256>>>(n, dF.p, dO.p);
	HIP_TRY0(hipGetLastError());
	HIP_TRY0(hipMemcpy(out12, dO.p, sizeof(float) * 12 * n, hipMemcpyDeviceToHost));
	return MPM_OK;
}

int mpm_test_stress(int material, const mpm_material_params* p, const float* F, const float* logjp, size_t n, float* out19, int device) {
	if(material < 1 || material > 3 || !p) return MPM_ERR_INVALID;
	HIP_TRY0(hipSetDevice(device));
	DevScratch<float> dF, dL, dO;
	HIP_TRY0(dF.alloc(9 * n));
	HIP_TRY0(dO.alloc(19 * n));
	HIP_TRY0(hipMemcpy(dF.p, F, sizeof(float) * 9 * n, hipMemcpyHostToDevice));
	if(logjp) {
		HIP_TRY0(dL.alloc(n));
		HIP_TRY0(hipMemcpy(dL.p, logjp, sizeof(float) * n, hipMemcpyHostToDevice));
	}
	test_stress_kernel<<<cdiv(n, 256), 256>>>(material, make_material_const(*p), n, dF.p, dL.p, dO.p, nullptr);
	HIP_TRY0(hipGetLastError());
	HIP_TRY0(hipMemcpy(out19, dO.p, sizeof(float) * 19 * n, hipMemcpyDeviceToHost));
	return MPM_OK;
}

int mpm_streams(mpm_ctx* ctx, void** compute_stream, void** comm_stream) {
	if(!ctx) return MPM_ERR_INVALID;
	if(compute_stream) *compute_stream = (void*) ctx->s_compute;
	if(comm_stream) *comm_stream = (void*) ctx->s_comm;
	return MPM_OK;
}

int mpm_sync(mpm_ctx* ctx) {
	if(!ctx) return MPM_ERR_INVALID;
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipStreamSynchronize(ctx->s_compute));
	HIP_TRY(hipStreamSynchronize(ctx->s_comm));
	return MPM_OK;
}

}// extern "C"

#include "mpm_halo.inc"
#include "mpm_checkpoint.inc"
#include "mpm_group.inc"
