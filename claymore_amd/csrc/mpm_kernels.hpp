// mpm_kernels.hpp — hand-written gfx950 kernels of the MPM substep hot path.
//
// Design (MI355X-first, not a translation of the reference's CUDA):
//   * wave64 everywhere: one wave == one 4x4x4 grid block (64 cells) in the grid kernels, one wave == one
//     64-slot AoSoA particle bin in G2P2G, so every bin-stride load/store is a full 256-B row;
//   * G2P2G workgroup = ONE wave = one particle block.  The 8 neighbouring grid blocks are staged through LDS
//     once (float4 {vx,vy,vz,-} per node -> one ds_read_b96 per stencil node).  The P2G scatter is atomic-free:
//     gfx950 serialises ds_add_f32 (193 cycles per wave-instruction, profiles/r01_lds_microbench.txt), so the
//     advection records of a block are counting-sorted (prepare_blocks_kernel, once per substep) into "k-th
//     particle of every key" order, key = predicted stencil base; the 64 lanes of an iteration therefore hold 64
//     distinct bases, and each lane read-modify-writes its 27 float4 nodes {m, px, py, pz} with plain
//     ds_read_b128/ds_write_b128 - a chain of 27 ordered LDS round trips that ScatterChain threads through the
//     NEXT particle's gather / SVD / stress arithmetic (lanes that collide with another lane's stencil base are
//     detected through an LDS owner table and retried).  The arena is written back with one hardware f32 atomic
//     per touched node and channel;
//   * block-level advection lists instead of the reference's cell buckets + compaction passes: a particle
//     appends ONE 4-byte record {direction tag, predicted stencil base, slot} to the list of the block it lands in
//     (wave-aggregated atomic for the particles that stay), and next step's G2P2G consumes that list
//     directly through a row indirection - 8 B/particle of bookkeeping traffic instead of 24 B and three
//     kernels fewer (reference: add_advection + cell_bucket_to_block + update_buckets);
//   * the partition rebuild is ONE compaction kernel with wave-aggregated atomics (no scans, no host
//     round trips): renumber blocks, rebuild the dense table, allocate bins; neighbour / exterior
//     registration and the grid carry-over read their counts from device memory.
//
// Reference kernels replaced (Projects/GMPM/mgmpm_kernels.cuh): update_grid_velocity_query_max :325-420,
// g2p2g :665-937 (+ :422-663), activate_blocks :21-34, build_particle_cell_buckets :36-68,
// cell_bucket_to_block :70-84, compute_bin_capacity :86-94, init_adv_bucket :96-104, clear_grid :106-115,
// register_neighbor_blocks :117-133, register_exterior_blocks :135-151, rasterize :153-219,
// array_to_buffer :221-323, mark_active_* :939-964, update_partition :966-977, update_buckets :979-1000,
// copy_selected_grid_blocks :1002-1020, retrieve_particle_buffer :1087-1122.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "mpm_device_math.hpp"
#include "mpm_collision.hpp"

namespace mpm {

constexpr int kBin		  = 64; // particles per AoSoA bin == wavefront width
constexpr int kG2P2GThreads = 64; // ONE wave per particle block: no cross-wave LDS hazards, no barriers that wait
constexpr int kMaxModels  = 8;
constexpr int kArenaStrideX = 68; // arena x stride in float4 nodes (64 + 4: keeps every b128 lane group on 16 distinct 16-B slots)
constexpr int kArenaNodes	= 544;// >= 7*68 + 7*8 + 7 + 1
constexpr int kSortChunk	= 1024;// advection records staged in LDS per pass of g2p2g (a block with more particles takes several passes)
constexpr int kSortRounds	= 24;  // particles per key (per chunk) that get an exact interleaved position; more -> appended behind
constexpr int kSortKeys		= 216; // sort key = PREDICTED stencil base of the particle in the arena of its block (6^3 values)
constexpr int kKeyBits		= 8;
constexpr int kG2PStrideX	= 52;  // G2P arena holds only nodes 1..6 of the 8^3 arena (the gather never touches 0 and 7):
constexpr int kG2PNodes		= 312; // index (x-1)*52 + (y-1)*8 + (z-1); 52 = 48 + 4 keeps ds_read_b96 lane groups conflict-free
constexpr int kStay		  = 13; // dir_offset(0,0,0), utility_funcs.hpp:25-27

// status block indices (device ints, read back once per substep)
enum { ST_PBC = 0, ST_NBC = 1, ST_EBC = 2, ST_OVERFLOW = 3, ST_LOST = 4, ST_ARENA = 5, ST_BINS0 = 8, ST_PART0 = 16, ST_WORDS = 32 };

struct GridCfg {
	int G;		  // blocks per axis
	int gbits;	  // log2(G)
	int ppb;	  // advection-list capacity per block (max_ppc * 64)
	int pid_bits; // log2(ppb)
	int boundary; // wall zone in blocks
	int cap;	  // block capacity
	float dx, dx_inv, d_inv, gravity;
};

__device__ __forceinline__ bool key_ok(const GridCfg& c, int x, int y, int z) {
	return ((unsigned) x < (unsigned) c.G) & ((unsigned) y < (unsigned) c.G) & ((unsigned) z < (unsigned) c.G);
}
__device__ __forceinline__ size_t key_index(const GridCfg& c, int x, int y, int z) {
	return ((size_t) x << (2 * c.gbits)) | ((size_t) y << c.gbits) | (size_t) z;// row-major, StructuralDeclaration.h:235-251
}
__device__ __forceinline__ int table_query(const GridCfg& c, const int* __restrict__ table, int x, int y, int z) {
	return key_ok(c, x, y, z) ? table[key_index(c, x, y, z)] : -1;
}
// Partition::insert, hash_table.cuh:118-127 (claim with CAS, then append).  Out-of-domain keys are ignored.
__device__ __forceinline__ void table_insert(const GridCfg& c, int* table, int* keys, int* count, int x, int y, int z, int* status) {
	if(!key_ok(c, x, y, z)) return;
	const size_t i = key_index(c, x, y, z);
	if(table[i] != -1) return;// cheap pre-check, most inserts hit an existing block
	if(atomicCAS(&table[i], -1, -2) == -1) {
		const int idx = atomicAdd(count, 1);
		if(idx < c.cap) {
			table[i]		 = idx;
			keys[3 * idx]	 = x;
			keys[3 * idx + 1] = y;
			keys[3 * idx + 2] = z;
		} else {
			table[i] = -1;
			atomicOr(&status[ST_OVERFLOW], 1);
		}
	}
}

// Append to a list through a shared counter with ONE atomic per wave (same-address atomics serialise in L2 at
// ~11 ns each: 84 k single-lane atomics cost ~1 ms).  Returns the slot for lanes with pred, -1 otherwise.
__device__ __forceinline__ int wave_append(int* counter, bool pred) {
	const unsigned long long m = __ballot(pred);
	if(m == 0ull) return -1;
	const int lane	 = threadIdx.x & 63;
	const int leader = __ffsll((long long) m) - 1;
	int base		 = 0;
	if(lane == leader) base = atomicAdd(counter, __popcll(m));
	base = __shfl(base, leader);
	return pred ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

// ------------------------------------------------------------------------------------------------------
constexpr int kMaxVelSlots	 = 64;// the running maximum is kept in 64 slots (by workgroup): same-address atomics serialise in L2 (~11 ns each)
constexpr int kMaxVelStride = 32;// ... one slot per 128-B line: the filtering loads of 20 k waves then spread over the L2 channels
// Grid update: momentum -> velocity, gravity, slip walls, max |v|^2.   One wave per grid block, lane = cell.
// (update_grid_velocity_query_max, mgmpm_kernels.cuh:325-420)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grid_update_kernel(GridCfg cfg, int nblocks, float* __restrict__ grid, const int* __restrict__ keys, float dt, unsigned* __restrict__ max_vel_bits) {
	// 16 lanes per grid block, 4 cells (one float4) per lane and channel: 16-B accesses, 4 blocks per wave
	const int sub	  = threadIdx.x & 15;
	const int blockno = (blockIdx.x * 256 + threadIdx.x) >> 4;
	float vel_sqr	  = 0.f;
	if(blockno < nblocks) {
		const int kx = keys[3 * blockno], ky = keys[3 * blockno + 1], kz = keys[3 * blockno + 2];
		const bool wx = kx < cfg.boundary || kx >= cfg.G - cfg.boundary;
		const bool wy = ky < cfg.boundary || ky >= cfg.G - cfg.boundary;
		const bool wz = kz < cfg.boundary || kz >= cfg.G - cfg.boundary;
		float4* g	   = reinterpret_cast<float4*>(grid + (size_t) blockno * 256) + sub;
		const float4 m = g[0];
		float4 p0 = g[16], p1 = g[32], p2 = g[48];
		const float gdt = cfg.gravity * dt;
#define MPM_CELL(c)                                                                 \
	if(m.c > 0.0f) {                                                                \
		const float mass_inv = 1.f / m.c;                                           \
		const float v0		 = wx ? 0.0f : p0.c * mass_inv;                         \
		const float v1		 = (wy ? 0.0f : p1.c * mass_inv) + gdt;                 \
		const float v2		 = wz ? 0.0f : p2.c * mass_inv;                         \
		p0.c				 = v0;                                                  \
		p1.c				 = v1;                                                  \
		p2.c				 = v2;                                                  \
		float q				 = v0 * v0 + v1 * v1 + v2 * v2;                         \
		if(q != q) q = __builtin_inff(); /* NaN -> inf signals failure (:385-388) */ \
		vel_sqr = fmaxf(vel_sqr, q);                                                \
	}
		MPM_CELL(x)
		MPM_CELL(y)
		MPM_CELL(z)
		MPM_CELL(w)
#undef MPM_CELL
		g[16] = p0;
		g[32] = p1;
		g[48] = p2;
	}
#pragma unroll
	for(int off = 32; off > 0; off >>= 1) vel_sqr = fmaxf(vel_sqr, __shfl_xor(vel_sqr, off));
	// non-negative floats order as uints.  One same-address atomic per wave would serialise in L2 (90 k waves = 1 ms):
	// the running maximum only grows, so a plain (possibly stale) read filters almost all of them out.
	if((threadIdx.x & 63) == 0 && vel_sqr > 0.f) {
		const unsigned bits = __float_as_uint(vel_sqr);
		unsigned* slot		= max_vel_bits + (blockIdx.x & (kMaxVelSlots - 1)) * kMaxVelStride;
		if(bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
	}
}

// Grid update with a level-set collision object (second update_grid_velocity_query_max overload,
// Projects/MGSP/mgmpm_kernels.cuh:323-399): one wave per grid block, lane = cell.  Reports the reference's doubled
// |v|^2 (vel.dot(vel) followed by the three += of the plain overload, :365-373).
__global__ __launch_bounds__(256) void grid_update_collision_kernel(GridCfg cfg, int nblocks, float* __restrict__ grid, const int* __restrict__ keys, float dt, CollisionObject obj, unsigned* __restrict__ max_vel_bits) {
	const int cell	  = threadIdx.x & 63;
	const int blockno = (blockIdx.x * 256 + threadIdx.x) >> 6;
	float vel_sqr	  = 0.f;
	if(blockno < nblocks) {
		const int kx = keys[3 * blockno], ky = keys[3 * blockno + 1], kz = keys[3 * blockno + 2];
		const bool wx = kx < cfg.boundary || kx >= cfg.G - cfg.boundary;
		const bool wy = ky < cfg.boundary || ky >= cfg.G - cfg.boundary;
		const bool wz = kz < cfg.boundary || kz >= cfg.G - cfg.boundary;
		float* g		 = grid + (size_t) blockno * 256 + cell;
		const float mass = g[0];
		if(mass > 0.0f) {
			const float mass_inv = 1.f / mass;
			float vel[3];
			vel[0] = wx ? 0.0f : g[64] * mass_inv;
			vel[1] = (wy ? 0.0f : g[128] * mass_inv) + cfg.gravity * dt;
			vel[2] = wz ? 0.0f : g[192] * mass_inv;
			const int node[3] = {kx * 4 + (cell >> 4), ky * 4 + ((cell >> 2) & 3), kz * 4 + (cell & 3)};
			collision_resolve(obj, node, cfg.dx, cfg.G * 4, (float) cfg.boundary * cfg.dx * 4.f, (float) (cfg.G - cfg.boundary) * 4.f * cfg.dx, vel);
			g[64]	= vel[0];
			g[128]	= vel[1];
			g[192]	= vel[2];
			float q = vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2];
			q += vel[0] * vel[0];
			q += vel[1] * vel[1];
			q += vel[2] * vel[2];
			if(q != q) q = __builtin_inff();
			vel_sqr = q;
		}
	}
#pragma unroll
	for(int off = 32; off > 0; off >>= 1) vel_sqr = fmaxf(vel_sqr, __shfl_xor(vel_sqr, off));
	if((threadIdx.x & 63) == 0 && vel_sqr > 0.f) {
		const unsigned bits = __float_as_uint(vel_sqr);
		unsigned* slot		= max_vel_bits + (blockIdx.x & (kMaxVelSlots - 1)) * kMaxVelStride;
		if(bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
	}
}

// ---- checkpoint helpers (mpm_checkpoint.inc) ----
// advection-list rows <-> one packed array: block b's size[b] records live at packed[offset[b] ...)
__global__ __launch_bounds__(64) void pack_lists_kernel(int ppb, const int* __restrict__ size, const int* __restrict__ row_of, const long long* __restrict__ offset, const int* __restrict__ list, int* __restrict__ packed) {
	const int b = blockIdx.x;
	const int n = size[b];
	const int* row = list + (size_t) row_of[b] * ppb;
	for(int i = threadIdx.x; i < n; i += 64) packed[offset[b] + i] = row[i];
}
__global__ __launch_bounds__(64) void unpack_lists_kernel(int ppb, const int* __restrict__ size, int* __restrict__ row_of, const long long* __restrict__ offset, int* __restrict__ list, const int* __restrict__ packed) {
	const int b = blockIdx.x;
	const int n = size[b];
	int* row	= list + (size_t) b * ppb;// rows are re-seated at their own block number
	for(int i = threadIdx.x; i < n; i += 64) row[i] = packed[offset[b] + i];
	if(threadIdx.x == 0) row_of[b] = b;
}
// dense table from a key list (the inverse of what compact / register build incrementally)
__global__ __launch_bounds__(256) void table_from_keys_kernel(GridCfg cfg, int n, const int* __restrict__ keys, int* __restrict__ table) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if(i < n) table[key_index(cfg, keys[3 * i], keys[3 * i + 1], keys[3 * i + 2])] = i;
}

// node-major {sdis, gx, gy, gz} from the four per-node arrays the ABI receives
__global__ __launch_bounds__(256) void pack_sdf_kernel(size_t n, const float* __restrict__ sd, const float* __restrict__ gx, const float* __restrict__ gy, const float* __restrict__ gz, float4* __restrict__ out) {
	const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
	if(i < n) out[i] = make_float4(sd[i], gx[i], gy[i], gz[i]);
}

// ------------------------------------------------------------------------------------------------------
// G2P2G
// ------------------------------------------------------------------------------------------------------
struct ModelView {
	const float* bins_src;// [bin][nch][64], laid out by the previous block numbering
	float* bins_dst;	  // laid out by the current numbering
	const int* binoff_src;// first bin of a block, previous numbering
	const int* binoff_dst;// current numbering
	const int* list_in;	  // advection records written by the previous step; row = row_of[b]
	int* list_out;		  // records for the next step; row = destination block (current numbering)
	const int* size;	  // particles per current block
	const int* row_of;	  // row of list_in that belongs to current block b
	int* out_count;		  // append counters of list_out
	const int* blockinfo; // [block][kInfoRow]: source bin offsets, destination / grid block numbers (prepare_blocks_kernel)
	MaterialConst mc;
};

template<int MAT>
struct MatTraits;
template<>
struct MatTraits<0> {
	static constexpr int nch = 4;
};
template<>
struct MatTraits<1> {
	static constexpr int nch = 12;
};
template<>
struct MatTraits<2> {
	static constexpr int nch = 13;
};
template<>
struct MatTraits<3> {
	static constexpr int nch = 13;
};

__device__ __forceinline__ void dir_components(int dir, int& dx, int& dy, int& dz) {
	dz = (dir % 3) - 1;
	dy = ((dir / 3) % 3) - 1;
	dx = (dir / 9) - 1;
}

// P2G payload of one particle: everything the scatter needs after the material update.
struct P2GPayload {
	float fd[3];	 // offset from the new stencil base node, in cells
	float mv[3];	 // mass * velocity
	float contrib[9];// (A m - stress new_dt) D^-1 dx   (see :850)
};

// Scatter one particle per active lane into the LDS arena (float4 {mass, px, py, pz} per node) WITHOUT atomics:
// gfx950 executes ds_add_f32 at one lane per ~3 cycles (193 cycles per wave-instruction, tools/lds_microbench),
// a plain ds_read_b128 / 4 v_add / ds_write_b128 costs 17.  Correctness rests on two facts: (1) the caller only
// activates lanes with pairwise distinct stencil bases, so for one stencil offset all lanes touch distinct nodes;
// (2) a workgroup is a single wave, whose LDS operations execute in program order - the compiler barrier keeps the
// read-modify-write of offset o ahead of the read of offset o+1, which may hit the node another lane just wrote.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void p2g_scatter_rmw(float4* __restrict__ node0, const P2GPayload& pl, float mass) {
	float w[3][3];
#pragma unroll
	for(int d = 0; d < 3; ++d) bspline_weight_cells(pl.fd[d], w[d]);
	// packed fp32 (v_pk_fma_f32) on the naturally paired halves of the float4 node: {m, px} and {py, pz}
	const v2f c12 = {pl.contrib[7], pl.contrib[8]};
#pragma unroll
	for(int i = 0; i < 3; ++i) {
		const float px = (float) i - pl.fd[0];
#pragma unroll
		for(int j = 0; j < 3; ++j) {
			const float py	= (float) j - pl.fd[1];
			const float wij = w[0][i] * w[1][j];
			const float b0	= pl.mv[0] + pl.contrib[0] * px + pl.contrib[3] * py;
			v2f b12			= {pl.mv[1] + pl.contrib[1] * px + pl.contrib[4] * py, pl.mv[2] + pl.contrib[2] * px + pl.contrib[5] * py};
#pragma unroll
			for(int k = 0; k < 3; ++k) {
				const float pz = (float) k - pl.fd[2];
				const float W  = wij * w[2][k];
				float4* node   = node0 + i * kArenaStrideX + j * 8 + k;
				float4 acc	   = *node;
				v2f m0		   = {mass, b0 + pl.contrib[6] * pz};
				v2f t12		   = c12 * pz + b12;
				v2f a01		   = {acc.x, acc.y};
				v2f a23		   = {acc.z, acc.w};
				a01			   = m0 * W + a01;
				a23			   = t12 * W + a23;
				*node		   = make_float4(a01.x, a01.y, a23.x, a23.y);
				__asm__ volatile("" ::: "memory");
				__builtin_amdgcn_sched_barrier(0);// rare path: keep the 27 steps' arithmetic from being hoisted (registers)
			}
		}
	}
}

// Tensor-product B-spline gather of one particle per lane (:801-835): vel = sum W v, A = sum W v (x_i - x_p)^T in cell
// units, separable over the three axes (27 x 6 + 9 x 9 + 3 x 12 multiply-adds instead of 27 x 12).  Written on float2
// so that the x and y components, and the (w, w (x_i - x_p)) weight pairs, go through packed fp32 FMAs (v_pk_fma_f32 with
// op_sel broadcasts): 3 packed instructions per node instead of 6 scalar ones.
constexpr int kGatherSites = 9;
template<int BASE, class Hook>
__device__ __forceinline__ void gather_apic(const float4* __restrict__ gbase, const float (&w)[3][3], const float (&fd)[3], float (&vel)[3], float (&A)[9], Hook& hk) {
	v2f wz[3], wy[3], wx[3];// {w, w * (node - particle)} per axis and stencil offset
#pragma unroll
	for(int t = 0; t < 3; ++t) {
		wx[t] = (v2f) {w[0][t], w[0][t] * ((float) t - fd[0])};
		wy[t] = (v2f) {w[1][t], w[1][t] * ((float) t - fd[1])};
		wz[t] = (v2f) {w[2][t], w[2][t] * ((float) t - fd[2])};
	}
	v2f vel_xy = {0.f, 0.f}, A0_xy = {0.f, 0.f}, A3_xy = {0.f, 0.f}, A6_xy = {0.f, 0.f};
	v2f velz_A2 = {0.f, 0.f};
	float A5 = 0.f, A8 = 0.f;
	v2f u0_xy, uy_xy, uz_xy, u0z_uyz;
	float uzz;
#define MPM_GATHER_ROW(i, j)                                                     \
	{                                                                            \
		v2f t0_xy = {0.f, 0.f}, t1_xy = {0.f, 0.f}, t0z_t1z = {0.f, 0.f};        \
		_Pragma("unroll") for(int k = 0; k < 3; ++k) {                           \
			const float4 v = gbase[i * kG2PStrideX + j * 8 + k];                 \
			const v2f vxy  = {v.x, v.y};                                         \
			t0_xy		   = vxy * wz[k].x + t0_xy;                              \
			t1_xy		   = vxy * wz[k].y + t1_xy;                              \
			t0z_t1z		   = wz[k] * v.z + t0z_t1z;                              \
		}                                                                        \
		if(j == 0) {                                                             \
			u0_xy = uy_xy = uz_xy = u0z_uyz = (v2f) {0.f, 0.f};                  \
			uzz											= 0.f;                   \
		}                                                                        \
		u0_xy	= t0_xy * wy[j].x + u0_xy;                                       \
		uy_xy	= t0_xy * wy[j].y + uy_xy;                                       \
		uz_xy	= t1_xy * wy[j].x + uz_xy;                                       \
		u0z_uyz = wy[j] * t0z_t1z.x + u0z_uyz;                                   \
		uzz += wy[j].x * t0z_t1z.y;                                              \
		if(j == 2) {                                                             \
			vel_xy	= u0_xy * wx[i].x + vel_xy;                                  \
			A0_xy	= u0_xy * wx[i].y + A0_xy;                                   \
			A3_xy	= uy_xy * wx[i].x + A3_xy;                                   \
			A6_xy	= uz_xy * wx[i].x + A6_xy;                                   \
			velz_A2 = wx[i] * u0z_uyz.x + velz_A2;                               \
			A5 += wx[i].x * u0z_uyz.y;                                           \
			A8 += wx[i].x * uzz;                                                 \
		}                                                                        \
		hk.template at<BASE + 3 * i + j>();                                      \
	}
	MPM_GATHER_ROW(0, 0)
	MPM_GATHER_ROW(0, 1)
	MPM_GATHER_ROW(0, 2)
	MPM_GATHER_ROW(1, 0)
	MPM_GATHER_ROW(1, 1)
	MPM_GATHER_ROW(1, 2)
	MPM_GATHER_ROW(2, 0)
	MPM_GATHER_ROW(2, 1)
	MPM_GATHER_ROW(2, 2)
#undef MPM_GATHER_ROW
	vel[0] = vel_xy.x;
	vel[1] = vel_xy.y;
	vel[2] = velz_A2.x;
	A[0]   = A0_xy.x;
	A[1]   = A0_xy.y;
	A[2]   = velz_A2.y;
	A[3]   = A3_xy.x;
	A[4]   = A3_xy.y;
	A[5]   = A5;
	A[6]   = A6_xy.x;
	A[7]   = A6_xy.y;
	A[8]   = A8;
}

// The P2G scatter of one particle per lane as a chain of 27 ordered LDS read-modify-write steps that is threaded
// through unrelated register-only arithmetic (the NEXT particle's re-bucketing, F update, SVD and stress).  Each step
// is an LDS round trip (~130-200 cycles under load) and the steps cannot overlap each other - step o+1 may hit the node
// another lane wrote in step o (see p2g_scatter_rmw) - so issued back to back they leave the wave idle; spread over
// NSITES call sites `at<SITE>()` of the host computation (~30 VALU instructions apart) the round trips disappear
// behind it.  Site s completes steps [27 s / NSITES, 27 (s+1) / NSITES): the accumulator of the following step is
// requested right after a step's write and consumed at the next site.  Only lanes with `win` (pairwise distinct
// stencil bases) take part; the others are scattered afterwards by p2g_resolve.
template<int NSITES>
struct ScatterChain {
	float4* node0;
	P2GPayload pp;// element-wise copy: a reference member or a struct copy keeps the payload in scratch memory
	float mass;
	int win;// (int, not bool: a 1-byte member makes the compiler slice its neighbours into bytes)
	float pw[3][3];
	float b0, wij;
	v2f b12, c12;
	float4 acc;
	MPM_DEV ScatterChain(float4* n0, const P2GPayload& p, float m, bool w)
		: node0(n0)
		, mass(m)
		, win(w) {
#pragma unroll
		for(int d = 0; d < 3; ++d) {
			pp.fd[d] = p.fd[d];
			pp.mv[d] = p.mv[d];
		}
#pragma unroll
		for(int d = 0; d < 9; ++d) pp.contrib[d] = p.contrib[d];
#pragma unroll
		for(int d = 0; d < 3; ++d) bspline_weight_cells(pp.fd[d], pw[d]);
		c12 = (v2f) {pp.contrib[7], pp.contrib[8]};
		if(win) acc = node0[0];
	}
	MPM_DEV void step(int o) {// o is a compile-time constant after unrolling
		const int i = o / 9, j = (o / 3) % 3, k = o % 3;
		if(k == 0) {
			const float px = (float) i - pp.fd[0], py = (float) j - pp.fd[1];
			b0	= pp.mv[0] + pp.contrib[0] * px + pp.contrib[3] * py;
			b12 = (v2f) {pp.mv[1] + pp.contrib[1] * px + pp.contrib[4] * py, pp.mv[2] + pp.contrib[2] * px + pp.contrib[5] * py};
			wij = pw[0][i] * pw[1][j];
		}
		const float pz = (float) k - pp.fd[2];
		const float W  = wij * pw[2][k];
		const v2f m0   = {mass, b0 + pp.contrib[6] * pz};
		const v2f t12  = c12 * pz + b12;
		if(win) {
			v2f a01			 = {acc.x, acc.y};
			v2f a23			 = {acc.z, acc.w};
			a01				 = m0 * W + a01;
			a23				 = t12 * W + a23;
			const int off	 = i * kArenaStrideX + j * 8 + k;
			node0[off]		 = make_float4(a01.x, a01.y, a23.x, a23.y);
			__asm__ volatile("" ::: "memory");
			if(o + 1 < 27) {
				const int i1 = (o + 1) / 9, j1 = ((o + 1) / 3) % 3, k1 = (o + 1) % 3;
				acc			 = node0[i1 * kArenaStrideX + j1 * 8 + k1];
			}
		}
	}
	template<int SITE>
	MPM_DEV void at() {
		static_assert(SITE >= 0 && SITE < NSITES, "site out of range");
#pragma unroll
		for(int o = SITE * 27 / NSITES; o < (SITE + 1) * 27 / NSITES; ++o) step(o);
	}
	template<int LO, int HI>
	MPM_DEV void range() {
		if constexpr(LO < HI) {
			at<LO>();
			range<LO + 1, HI>();
		}
	}
};

// Resolve intra-wave conflicts for one batch of payloads: lanes whose stencil base (key) is unique in the wave
// scatter immediately; the others retry.  key < 216 (6^3 possible new cells around a block).
__device__ __forceinline__ void p2g_resolve(float4* __restrict__ arena, unsigned char* __restrict__ owner, bool pending, int key, int nodeoff, const P2GPayload& pl, float mass, int lane) {
	while(__any(pending)) {
		if(pending) owner[key] = (unsigned char) lane;
		__syncthreads();// single-wave workgroup: orders the LDS write before the read-back (and fences the compiler)
		const bool win = pending && (int) owner[key] == lane;
		__syncthreads();
		if(win) p2g_scatter_rmw(arena + nodeoff, pl, mass);
		pending = pending && !win;
	}
}

// ------------------------------------------------------------------------------------------------------
// Per-block preparation for G2P2G, run once per substep after the partition rebuild with one wave per particle block:
//  * the block's advection records are counting-sorted IN PLACE into "k-th particle of every key" order, the key being the
//    stencil base the particle is PREDICTED to have after the coming step (written into the record one step earlier from
//    x + v dt, exact for > 99.9 % of the particles): the 64 lanes of a G2P2G iteration then scatter to 64 distinct
//    stencil bases (replaces cell_bucket_to_block, mgmpm_kernels.cuh:70-84);
//  * the 27 source-block bin offsets, the 27 destination block numbers and the 8 grid-block numbers the block will need are
//    looked up in the two dense tables and written as one 256-B row.
// Both used to be the per-block prologue of g2p2g_kernel.  There they were a chain of dependent LDS / global round trips
// executed at 2 waves per SIMD (the register budget of the main loop); here the same work runs at full occupancy, and
// G2P2G starts its first particle fetch two round trips earlier.
// ------------------------------------------------------------------------------------------------------
constexpr int kInfoRow = 64;// ints per block: [0,27) source bin offsets, [27,54) destination block numbers, [54,62) grid blocks
struct PrepareModels {
	int n;
	int* list[kMaxModels];				// advection lists, sorted in place
	const int* size[kMaxModels];
	const int* row_of[kMaxModels];
	const int* binoff_src[kMaxModels];	// bin offsets in the numbering the particle data is laid out in (previous partition)
	int* blockinfo[kMaxModels];
};
template<bool SORT>
__global__ __launch_bounds__(64) void prepare_blocks_kernel(GridCfg cfg, PrepareModels pm, const int* __restrict__ pbc_ptr, const int* __restrict__ cur_table, const int* __restrict__ cur_keys, const int* __restrict__ prev_table) {
	// records are sorted in chunks of 512 (8 per lane): G2P2G only needs every aligned 64-record slice to hold distinct keys,
	// and the smaller chunk keeps this kernel at 3.9 KB of LDS, i.e. at the hardware's wave limit
	constexpr int kPrepChunk = 512;
	__shared__ int s_sorted[kPrepChunk];
	__shared__ unsigned long long s_mask[kSortRounds][4];// per round: which of the 216 keys still have a k-th particle
	__shared__ int s_round0[kSortRounds + 1];			  // first sorted position of round k
	__shared__ unsigned s_wordoff[kSortRounds];			  // packed prefix of the four mask words' popcounts (3 x 8 bit)
	__shared__ int s_cnt[256 - 32];						  // 216 keys used
	const int lane = threadIdx.x;
	const int b	   = blockIdx.x;
	if(b >= *pbc_ptr) return;
	const int kx = cur_keys[3 * b], ky = cur_keys[3 * b + 1], kz = cur_keys[3 * b + 2];
	// ---- look-ups (model independent except for the bin offsets)
	int srcno = -1, other = -1;
	if(lane < 27) {
		int ox, oy, oz;
		dir_components(lane, ox, oy, oz);
		srcno = table_query(cfg, prev_table, kx + ox, ky + oy, kz + oz);
	} else if(lane < 54) {
		int ox, oy, oz;
		dir_components(lane - 27, ox, oy, oz);
		other = table_query(cfg, cur_table, kx - ox, ky - oy, kz - oz);
	} else if(lane < 62) {
		const int lb = lane - 54;
		other		 = table_query(cfg, cur_table, kx + ((lb >> 2) & 1), ky + ((lb >> 1) & 1), kz + (lb & 1));
	}
	const int key_shift		= cfg.pid_bits;
	const int tag_shift		= cfg.pid_bits + kKeyBits;
	const unsigned rec_mask = (1u << (tag_shift + 5)) - 1u;
	for(int m = 0; m < pm.n; ++m) {
		int info = other;
		if constexpr(SORT) {
		const int size = pm.size[m][b];
		int* list	   = pm.list[m] + (size_t) pm.row_of[m][b] * cfg.ppb;
		constexpr int NIT = kPrepChunk / 64;
		unsigned recs[NIT];
		auto load_chunk = [&](int chunk0, int nrec) {// unconditional, clamped: all loads of a chunk are in flight together
#pragma unroll
			for(int it = 0; it < NIT; ++it) recs[it] = (unsigned) list[chunk0 + min(it * 64 + lane, nrec - 1)];
		};
		if(size > 0) load_chunk(0, min(kPrepChunk, size));
		// the bin offsets of the 27 source blocks: the look-up issued at the top has arrived by now, this dependent load
		// overlaps the LDS-only sort below
		if(lane < 27) info = srcno >= 0 ? pm.binoff_src[m][srcno] : -1;
		auto sort_chunk = [&](int chunk0, int nrec) {
#pragma unroll
			for(int q = 0; q < 4; ++q)
				if(lane + 64 * q < 224) s_cnt[lane + 64 * q] = 0;
			__syncthreads();
			unsigned packed[NIT];
#pragma unroll
			for(int it = 0; it < NIT; ++it) {
				const int idx = it * 64 + lane;
				packed[it]	  = 0u;
				if(idx < nrec) {
					const unsigned rec = recs[it] & rec_mask;
					const int c		   = (rec >> key_shift) & 255;
					const int k		   = atomicAdd(&s_cnt[c], 1);// ds_add_rtn_u32: integer LDS atomics run at full rate
					packed[it]		   = rec | ((unsigned) min(k, kSortRounds) << 26);
				}
			}
			__syncthreads();
			{
				int my[4];
				int maxc = 0;
#pragma unroll
				for(int q = 0; q < 4; ++q) {
					my[q] = lane + 64 * q < 224 ? s_cnt[lane + 64 * q] : 0;
					maxc  = max(maxc, my[q]);
				}
#pragma unroll
				for(int off = 32; off > 0; off >>= 1) maxc = max(maxc, __shfl_xor(maxc, off));
				maxc	= min(maxc, kSortRounds);
				int run = 0;
				for(int k = 0; k < maxc; ++k) {
					const unsigned long long m0 = __ballot(my[0] > k), m1 = __ballot(my[1] > k), m2 = __ballot(my[2] > k), m3 = __ballot(my[3] > k);
					const int p0 = __popcll(m0), p1 = p0 + __popcll(m1), p2 = p1 + __popcll(m2);
					if(lane == 0) {
						s_mask[k][0] = m0;
						s_mask[k][1] = m1;
						s_mask[k][2] = m2;
						s_mask[k][3] = m3;
						s_wordoff[k] = (unsigned) p0 << 8 | (unsigned) p1 << 16 | (unsigned) p2 << 24;
						s_round0[k]	 = run;
					}
					run += p2 + __popcll(m3);
				}
				if(lane == 0) s_round0[kSortRounds] = run;// overflow records (k >= kSortRounds) go behind the sorted ones
				s_cnt[lane] = 0;
			}
			__syncthreads();
#pragma unroll
			for(int it = 0; it < NIT; ++it) {
				if(it * 64 + lane < nrec) {
					const unsigned rec = packed[it] & rec_mask;
					const int k		   = packed[it] >> 26;
					const int c		   = (rec >> key_shift) & 255;
					int pos;
					if(k < kSortRounds) {
						const int w = c >> 6;
						pos			= s_round0[k] + (int) ((s_wordoff[k] >> (8 * w)) & 255u) + __popcll(s_mask[k][w] & ((1ull << (c & 63)) - 1ull));
					} else {
						pos = s_round0[kSortRounds] + atomicAdd(&s_cnt[0], 1);
					}
					s_sorted[pos] = (int) rec;
				}
			}
			__syncthreads();
#pragma unroll
			for(int it = 0; it < NIT; ++it)
				if(it * 64 + lane < nrec) list[chunk0 + it * 64 + lane] = s_sorted[it * 64 + lane];
			__syncthreads();
		};
		for(int chunk0 = 0; chunk0 < size; chunk0 += kPrepChunk) {
			if(chunk0) load_chunk(chunk0, min(kPrepChunk, size - chunk0));
			sort_chunk(chunk0, min(kPrepChunk, size - chunk0));
		}
		}
		if constexpr(!SORT) {
			if(lane < 27) info = srcno >= 0 ? pm.binoff_src[m][srcno] : -1;
		}
		pm.blockinfo[m][(size_t) b * kInfoRow + lane] = info;
	}
}

// Phase timing (ABL & 32, profiling builds only): wall cycles (s_memtime) that wave 0..n spend in each part of the
// iteration, summed over all waves (slot names: claymore_hip.hip, mpm_destroy); profiles/r01_phase_timing.txt.
__device__ unsigned long long g_prof[1024][20];// spread over 1024 rows: same-address atomics serialise
#define MPM_TICK(slot) \
	if constexpr(ABL & 32) { \
		__asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
		const unsigned long long t_now = __builtin_readcyclecounter(); \
		t_acc[slot] += t_now - t_last; \
		t_last = t_now; \
	}

// tools/asm_stats.sh -DMPM_ASM_MARKS: comment markers in the assembly for per-phase instruction counts
#ifdef MPM_ASM_MARKS
#define MPM_MARK(name) __asm__ volatile("; MPM_MARK " name)
#else
#define MPM_MARK(name)
#endif

// ABL: ablation mask for profiling builds (0 in production): 1 skip the P2G scatter, 2 skip the stress (SVD),
// 4 skip the G2P gather, 32 phase timing (below).  Values are kept live with empty asm statements so that the compiler
// cannot delete upstream work.
template<int MAT, int ABL = 0>
__global__ __launch_bounds__(kG2P2GThreads, 2) void g2p2g_kernel(GridCfg cfg, ModelView mv, const int* __restrict__ cur_keys, const float* __restrict__ grid, float* __restrict__ next_grid, const int* __restrict__ block_list, float dt, float new_dt, int* __restrict__ status) {
	constexpr int NCH = MatTraits<MAT>::nch;
	__shared__ float4 g2p[kG2PNodes];// node velocities {vx,vy,vz,-} of arena nodes 1..6 per axis
	__shared__ float4 p2g[kArenaNodes];// {mass, momentum} accumulators
	__shared__ int s_sorted[kSortChunk];// advection records of the current chunk (sorted by prepare_blocks_kernel)
	__shared__ unsigned char s_owner[216];
	__shared__ int s_src_binoff[27], s_dst_no[27], s_nb[8];

	const int lane = threadIdx.x;
	// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Consecutive block numbers are spatial
	// neighbours (they share grid blocks: reads in the set-up, atomics in the write-back), so every XCD gets one contiguous
	// eighth of the block range instead of every eighth block.
	const int nwg = (int) gridDim.x, xcd = (int) (blockIdx.x & 7u), q = (int) (blockIdx.x >> 3);
	const int bid = xcd * (nwg >> 3) + min(xcd, nwg & 7) + q;// XCD r owns (nwg / 8) + (r < nwg % 8) consecutive numbers
	const int b	  = block_list ? block_list[bid] : bid;
	// The per-block set-up is three waves of independent loads (a chain of ~8 dependent round trips of 2-4 us each would
	// be a third of the kernel); sorting the records and the table look-ups happened in prepare_blocks_kernel.
	// ---- round trip 1: everything addressed by the block number alone (scalar loads)
	const int size		 = mv.size[b];
	const int kx = cur_keys[3 * b], ky = cur_keys[3 * b + 1], kz = cur_keys[3 * b + 2];
	const int row		 = mv.row_of[b];
	const int binoff_dst = mv.binoff_dst[b];
	if(size == 0) return;// (:692-697)
	unsigned long long t_acc[20] = {};// [9] iterations whose lanes all won the claim, [10] iterations with losers, [11] mispredicted keys
	unsigned long long t_last	= 0;
	if constexpr(ABL & 32) t_last = __builtin_readcyclecounter();
	__asm__ volatile("" ::"s"(row), "s"(binoff_dst), "s"(kz));
	MPM_TICK(12)
	const int* list		 = mv.list_in + (size_t) row * cfg.ppb;
	const float dx_inv	 = cfg.dx_inv;
	const float scale	 = 4.f * cfg.dx_inv;// dx * D^-1 (settings.h:66): A is accumulated in cell units
	const float mass	 = mv.mc.mass;
	const int key_shift = cfg.pid_bits;
	const int tag_shift = cfg.pid_bits + kKeyBits;
	// ---- round trip 2: the first chunk's (sorted) advection records and the block's row of look-up results
	auto load_records = [&](int chunk0) {
		const int last = min(kSortChunk, size - chunk0) - 1;
		int recs[kSortChunk / 64];
#pragma unroll
		for(int it = 0; it < kSortChunk / 64; ++it) recs[it] = list[chunk0 + min(it * 64 + lane, last)];
#pragma unroll
		for(int it = 0; it < kSortChunk / 64; ++it) s_sorted[it * 64 + lane] = recs[it];
	};
	const int info = mv.blockinfo[(size_t) b * kInfoRow + lane];
	load_records(0);
	if(lane < 27) s_src_binoff[lane] = info;
	else if(lane < 54) s_dst_no[lane - 27] = info;
	else if(lane < 62) s_nb[lane - 54] = info;
	for(int i = lane; i < kArenaNodes; i += 64) p2g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
	__syncthreads();
	MPM_TICK(13)
	// ---- round trip 3: the 8 grid blocks (lane = cell -> 256-B rows per channel, :699-727) and - below, at the top of the
	//      chunk loop - the first 64 particles
	const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;// lane == cell of a 4x4x4 block
	float4 gv[8];
#pragma unroll
	for(int lb = 0; lb < 8; ++lb) {
		const int nb	= s_nb[lb];
		const float* gb = grid + (size_t) (nb < 0 ? 0 : nb) * 256;
		gv[lb].x		= gb[64 + lane];
		gv[lb].y		= gb[128 + lane];
		gv[lb].z		= gb[192 + lane];
		gv[lb].w		= 0.f;
		if(nb < 0) gv[lb].x = gv[lb].y = gv[lb].z = 0.f;
	}
	MPM_TICK(6)
	// Software prefetch: the particle data of iteration i+1 is requested at the top of iteration i (HBM latency under
	// load is 2-4 us and only two waves share a SIMD).  Two details keep the compiler's s_waitcnt insertion from
	// turning this into a wait for everything (it only counts memory operations that are issued unconditionally):
	// the loads are unconditional - lanes past the end of the chunk re-read its last record - and the wait for the
	// data is forced at the END of iteration i (`touch`), in the same straight-line code as the 13 particle stores,
	// where it is an exact `vmcnt(13)`; at the loop header it would be `vmcnt(0)`, i.e. include the stores'
	// acknowledgements and the list-append atomics.
	struct Prefetch {
		float pos[3], st[10];
		int key;// the stencil base this particle was predicted to have after this step (its sort key)
	};
	int nrec   = min(kSortChunk, size);
	auto fetch = [&](int idx0, Prefetch& f) {
		const int rec	 = s_sorted[min(idx0 + lane, nrec - 1)];
		const int tag	 = rec >> tag_shift;
		const int sp	 = rec & (cfg.ppb - 1);
		const int sbin	 = s_src_binoff[tag] + (sp >> 6);
		const float* src = mv.bins_src + (size_t) sbin * (NCH * kBin) + (sp & 63);
		f.key			 = (rec >> key_shift) & 255;
		f.pos[0]		 = src[0];
		f.pos[1]		 = src[kBin];
		f.pos[2]		 = src[2 * kBin];
		if constexpr(MAT == 0) {
			f.st[0] = src[3 * kBin];
		} else {
#pragma unroll
			for(int d = 0; d < 9; ++d) f.st[d] = src[(3 + d) * kBin];
			if constexpr(NCH == 13) f.st[9] = src[12 * kBin];
		}
	};
	auto touch = [&](Prefetch& f) {
#pragma unroll
		for(int d = 0; d < 3; ++d) __asm__ volatile("" : "+v"(f.pos[d]));
#pragma unroll
		for(int d = 0; d < (MAT == 0 ? 1 : (NCH == 13 ? 10 : 9)); ++d) __asm__ volatile("" : "+v"(f.st[d]));
	};
	Prefetch pf;
	fetch(0, pf);// the first 64 particles are in flight while the grid blocks requested above go to LDS
#pragma unroll
	for(int lb = 0; lb < 8; ++lb) {
		const int ax = cx + ((lb & 4) ? 4 : 0) - 1, ay = cy + ((lb & 2) ? 4 : 0) - 1, az = cz + ((lb & 1) ? 4 : 0) - 1;
		if(((unsigned) ax < 6u) & ((unsigned) ay < 6u) & ((unsigned) az < 6u)) g2p[ax * kG2PStrideX + ay * 8 + az] = gv[lb];
	}
	__syncthreads();
	MPM_TICK(17)
	for(int chunk0 = 0;;) {
		touch(pf);
		MPM_TICK(1)
		// Software pipeline: the scatter of iteration i-1 (an ordered chain of 27 LDS round trips) is issued inside
		// the gather of iteration i; `pv` is the payload in flight.
		P2GPayload pv;
		int pv_key = 0, pv_off = 0;
		bool pv_in = false, have_prev = false;
		for(int idx0 = 0; idx0 < nrec; idx0 += 64) {
			MPM_MARK("loop_top");
			const bool active = idx0 + lane < nrec;
			const int pidib	  = chunk0 + idx0 + lane;// slot in the destination bins == position in the sorted order
			// ---- advection record -> source bin (:747-768): data was requested one iteration ago
			float pos[3] = {pf.pos[0], pf.pos[1], pf.pos[2]};
			float st[10];// J, or F[9] (+ logJp)
#pragma unroll
			for(int d = 0; d < 10; ++d) st[d] = pf.st[d];
			const int predicted_key = pf.key;
			if constexpr(ABL & 32) {
#pragma unroll
				for(int d = 0; d < 10; ++d) __asm__ volatile("" ::"v"(st[d]));
#pragma unroll
				for(int d = 0; d < 3; ++d) __asm__ volatile("" ::"v"(pos[d]));
				t_acc[8] += 1;
			}
			MPM_TICK(1)
			fetch(idx0 + 64, pf);
			// ---- stencil base + weights (:774-797) for ALL lanes (idle lanes of a last partial iteration carry a dummy
			//      position inside the block); offsets in cell units (exact: dx is a power of two)
			int base[3], arena[3];
			float fd[3], w[3][3];
#pragma unroll
			for(int d = 0; d < 3; ++d) {
				const float p = pos[d] * dx_inv;
				base[d]		  = lround_pos(p) - 1;
				fd[d]		  = p - (float) base[d];
				bspline_weight_cells(fd[d], w[d]);
				arena[d] = ((base[d] - 1) & 3) + 1;
			}
			float vel[3] = {0.f, 0.f, 0.f};
			float A[9]	 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
			const float4* gbase = g2p + (arena[0] - 1) * kG2PStrideX + (arena[1] - 1) * 8 + (arena[2] - 1);
			MPM_MARK("claim");
			// ---- claim the stencil bases of the payload in flight: lanes whose base is unique in the wave (`win`) scatter
			//      in the chain threaded through this iteration's arithmetic, the others afterwards
			bool win = false;
			if constexpr(!(ABL & 1)) {
				if(have_prev) {
					// one LDS round trip: the workgroup is a single wave, whose LDS operations execute in order, so the read
					// below sees the write above without waiting for it in between
					if(pv_in) s_owner[pv_key] = (unsigned char) lane;
					__asm__ volatile("" ::: "memory");// another lane may have written the same byte: no store-to-load forwarding
					win = pv_in && (int) s_owner[pv_key] == lane;
					if constexpr(ABL & 32) {
						if(__all(win || !pv_in)) t_acc[9] += 1;
						else t_acc[10] += 1;
					}
				}
			}
			MPM_MARK("gather");
			MPM_TICK(2)
			constexpr int kPreSites = kGatherSites + 3, kStressSites = MAT == 0 ? 1 : (MAT == 1 ? kFcSites : (MAT == 2 ? kSandSites : kNaccSites));
			constexpr int kSites	= kPreSites + kStressSites + 1;
			ScatterChain<kSites> chain(p2g + pv_off, pv, mass, win);
			if constexpr(ABL & 4) {
				chain.template range<0, kGatherSites>();
				const float4 v = gbase[0];
				vel[0] = v.x * w[0][0]; vel[1] = v.y * w[1][1]; vel[2] = v.z * w[2][2];
				A[0] = v.x * fd[0]; A[4] = v.y * fd[1]; A[8] = v.z * fd[2];
			} else {
				gather_apic<0>(gbase, w, fd, vel, A, chain);
			}
			if constexpr(ABL & 32) {
#pragma unroll
				for(int d = 0; d < 9; ++d) __asm__ volatile("" ::"v"(A[d]));
			}
			MPM_TICK(3)
			MPM_MARK("rebucket");
			P2GPayload pl;
			// Every lane runs the whole body: the idle lanes of a last partial iteration carry a dummy particle and write
			// it into the padding slots of the block's last bin (slot == pidib < 64 * ceil(size / 64), allocated but never
			// read).  Keeping the 13 stores out of divergent control flow matters: the wait for the NEXT iteration's
			// prefetched data at the loop back-edge is then `vmcnt(13)` instead of `vmcnt(0)` (the compiler can only count
			// unconditional operations), i.e. it no longer includes the store acknowledgements.
			// ---- advect (:838)
#pragma unroll
			for(int d = 0; d < 3; ++d) pos[d] += vel[d] * dt;
			// ---- new base, re-bucket (:852-866, add_advection particle_buffer.cuh:100-135).  The list-append atomics are
			//      issued BEFORE the stress computation, which hides their round trip to L2 (~3 k cycles).
			int nbase[3], narena[3], dirv[3];
			bool in_arena = active;
#pragma unroll
			for(int d = 0; d < 3; ++d) {
				const float p = pos[d] * dx_inv;
				nbase[d]	  = lround_pos(p) - 1;
				pl.fd[d]	  = p - (float) nbase[d];
				pl.mv[d]	  = mass * vel[d];
				dirv[d]		  = ((base[d] - 1) >> 2) - ((nbase[d] - 1) >> 2);
				narena[d]	  = arena[d] + (nbase[d] - base[d]);
				in_arena &= (narena[d] >= 0) & (narena[d] + 2 < 8);
			}
			chain.template at<kGatherSites + 0>();
			const int key	  = narena[0] * 36 + narena[1] * 6 + narena[2];
			const int nodeoff = narena[0] * kArenaStrideX + narena[1] * 8 + narena[2];
			if constexpr(ABL & 32) t_acc[11] += __popcll(__ballot(in_arena && key != predicted_key));
			const bool dir_ok = ((unsigned) (dirv[0] + 1) < 3u) & ((unsigned) (dirv[1] + 1) < 3u) & ((unsigned) (dirv[2] + 1) < 3u);
			const int ntag	  = dir_ok ? (dirv[0] + 1) * 9 + (dirv[1] + 1) * 3 + dirv[2] + 1 : kStay;
			const int dno	  = (active && dir_ok) ? s_dst_no[ntag] : -1;
			// sort key for the NEXT step: predicted stencil base after one more advection with the current velocity,
			// expressed in the arena of the block the particle is in after THIS step (clamped to the 6^3 range)
			int pkey = 0;
#pragma unroll
			for(int d = 0; d < 3; ++d) {
				const int pb = (int) __builtin_rintf((pos[d] + vel[d] * new_dt) * dx_inv) - 1;// a prediction: ties do not matter
				const int nk = min(max(((nbase[d] - 1) & 3) + 1 + (pb - nbase[d]), 0), 5);
				pkey		 = pkey * 6 + nk;
			}
			chain.template at<kGatherSites + 1>();
			const int rec	= (ntag << tag_shift) | (pkey << key_shift) | pidib;
			const bool stay = dno >= 0 && ntag == kStay;
			// particles that stay in this block share one wave-aggregated atomic
			const unsigned long long stay_m = __ballot(stay);
			const int stay_leader			= stay_m ? __ffsll((long long) stay_m) - 1 : 0;
			const int stay_rank				= __popcll(stay_m & ((1ull << lane) - 1ull));
			int raw_stay = 0, raw_move = 0;
			int b_opaque = b;
			__asm__("" : "+v"(b_opaque));// hide the uniform address: the compiler's atomic optimiser would broadcast the
										 // result with v_readfirstlane right here, i.e. wait for the round trip
			if(stay_m != 0ull && lane == stay_leader) raw_stay = atomicAdd(&mv.out_count[b_opaque], __popcll(stay_m));
			if(dno >= 0 && !stay) raw_move = atomicAdd(&mv.out_count[dno], 1);
			if(active) {
				if(dno < 0) atomicAdd(&status[ST_LOST], 1);// reference: particle silently lost (particle_buffer.cuh:105-113)
				if(!in_arena) atomicAdd(&status[ST_ARENA], 1);// (:877-885) contribution discarded
			}
			MPM_MARK("stress");
			// ---- material update, store to the destination bin (coalesced: slot == pidib) (:470-663)
			float* dst = mv.bins_dst + (size_t) (binoff_dst + (pidib >> 6)) * (NCH * kBin) + (pidib & 63);
			dst[0]		  = pos[0];
			dst[kBin]	  = pos[1];
			dst[2 * kBin] = pos[2];
			if constexpr(MAT == 0) {
				float Aw[9];
#pragma unroll
				for(int d = 0; d < 9; ++d) Aw[d] = A[d] * cfg.dx;
				chain.template at<kGatherSites + 2>();
				const float J = stress_jfluid(mv.mc, st[0], Aw, dt, cfg.d_inv, pl.contrib);
				chain.template at<kPreSites>();
				dst[3 * kBin] = J;
			} else {
				float dws[9], Fold[9], F[9];
#pragma unroll
				for(int d = 0; d < 9; ++d) {
					dws[d]	= (A[d] * dt) * scale + ((d & 0x3) != 0 ? 0.f : 1.f);
					Fold[d] = st[d];
				}
				matmul3(dws, Fold, F);
				chain.template at<kGatherSites + 2>();
				if constexpr(ABL & 2) {
#pragma unroll
					for(int d = 0; d < 9; ++d) pl.contrib[d] = F[d] * mv.mc.mu;
					if constexpr(NCH == 13) dst[12 * kBin] = st[9];
					chain.template range<kPreSites, kPreSites + kStressSites>();
				} else if constexpr(MAT == 1) {
					stress_fixed_corotated<kPreSites>(mv.mc, F, pl.contrib, chain);
				} else if constexpr(MAT == 2) {
					float lj = st[9];
					stress_sand<kPreSites>(mv.mc, F, lj, pl.contrib, chain, dst + 3 * kBin, kBin);
					dst[12 * kBin] = lj;
				} else {
					float lj = st[9];
					stress_nacc<kPreSites>(mv.mc, F, lj, pl.contrib, chain);
					dst[12 * kBin] = lj;
				}
				if constexpr(MAT != 2 || (ABL & 2)) {
#pragma unroll
					for(int d = 0; d < 9; ++d) dst[(3 + d) * kBin] = F[d];
				}
			}
			MPM_MARK("tail");
			// (:850) contrib = (A m - contrib new_dt) D^-1, pre-multiplied by dx so that P2G can stay in cell units
			{
				const float am = mass * cfg.dx * cfg.dx * cfg.d_inv;
				const float cs = new_dt * cfg.d_inv * cfg.dx;
#pragma unroll
				for(int d = 0; d < 9; ++d) pl.contrib[d] = A[d] * am - pl.contrib[d] * cs;
			}
			chain.template at<kSites - 1>();
			// ---- the lanes that lost the claim scatter now (rare: a partial or overflowing sort round)
			if constexpr(!(ABL & 1)) {
				if(have_prev) {
					const bool lost = pv_in && !win;
					if(__any(lost)) p2g_resolve(p2g, s_owner, lost, pv_key, pv_off, pv, mass, lane);
				}
			}
			if constexpr(ABL & 32) {
#pragma unroll
				for(int d = 0; d < 9; ++d) __asm__ volatile("" ::"v"(pl.contrib[d]));
			}
			MPM_TICK(4)
			// ---- list append: the atomics' results are in by now
			{
				const int basev = __shfl(raw_stay, stay_leader);
				if(dno >= 0) {
					const int slot = stay ? basev + stay_rank : raw_move;
					if(slot >= cfg.ppb)
						atomicOr(&status[ST_OVERFLOW], 2);// reference drops beyond 128 per cell (:122-130)
					else
						mv.list_out[(size_t) dno * cfg.ppb + slot] = rec;
				}
			}
			MPM_MARK("loop_end");
			touch(pf);// the next iteration's particle data must have arrived by now
			MPM_TICK(5)
			// ---- hand the payload to the next iteration, whose arithmetic its scatter chain is threaded through (:887-905)
			if constexpr(ABL & 1) {
#pragma unroll
				for(int d = 0; d < 9; ++d) __asm__ volatile("" ::"v"(pl.contrib[d]));
#pragma unroll
				for(int d = 0; d < 3; ++d) __asm__ volatile("" ::"v"(pl.fd[d]), "v"(pl.mv[d]));
				__asm__ volatile("" ::"v"(key), "v"(nodeoff));
			}
			pv		  = pl;
			pv_key	  = in_arena ? key : 0;
			pv_off	  = in_arena ? nodeoff : 0;
			pv_in	  = in_arena;
			have_prev = true;
		}
		// drain the pipeline: scatter of the chunk's last iteration
		if constexpr(!(ABL & 1)) {
			if(have_prev) p2g_resolve(p2g, s_owner, pv_in, pv_key, pv_off, pv, mass, lane);
		}
		__syncthreads();
		MPM_TICK(3)
		chunk0 += kSortChunk;
		if(chunk0 >= size) break;
		load_records(chunk0);
		__syncthreads();
		nrec = min(kSortChunk, size - chunk0);
		fetch(0, pf);
		MPM_TICK(0)
	}
	// ---- arena -> next grid: one hardware f32 atomic per touched node, 256-B rows (:907-936)
#pragma unroll
	for(int lb = 0; lb < 8; ++lb) {
		const int nb   = s_nb[lb];
		const float4 v = p2g[(cx + ((lb & 4) ? 4 : 0)) * kArenaStrideX + (cy + ((lb & 2) ? 4 : 0)) * 8 + (cz + ((lb & 1) ? 4 : 0))];
		if(nb >= 0) {
			float* g = next_grid + (size_t) nb * 256 + lane;
			if(v.x != 0.f) unsafeAtomicAdd(g, v.x);
			if(v.y != 0.f) unsafeAtomicAdd(g + 64, v.y);
			if(v.z != 0.f) unsafeAtomicAdd(g + 128, v.z);
			if(v.w != 0.f) unsafeAtomicAdd(g + 192, v.w);
		}
	}
	if constexpr(ABL & 32) {
		__asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
		MPM_TICK(7)
		if(lane == 0) {
#pragma unroll
			for(int i = 0; i < 20; ++i) atomicAdd(&g_prof[blockIdx.x & 1023][i], t_acc[i]);
		}
	}
}

// ------------------------------------------------------------------------------------------------------
// Partition rebuild
// ------------------------------------------------------------------------------------------------------
struct RebuildModels {
	int n;
	const int* out_count[kMaxModels];
	int* size[kMaxModels];
	int* row_of[kMaxModels];
	int* binoff[kMaxModels];// destination bin offsets for the NEXT step (new numbering)
};

// One pass replaces mark_active_particle_blocks + exclusive_scan + exclusive_scan_inverse + update_partition +
// update_buckets + compute_bin_capacity + exclusive_scan (gmpm_simulator.cuh:436-505): blocks that received
// particles get a new number (wave-aggregated atomic), their key goes into the new table, their bins are
// allocated.  Order of the new numbering is arbitrary (as it is in the reference: insert order of atomics).
__global__ __launch_bounds__(256) void compact_blocks_kernel(GridCfg cfg, int ebc, RebuildModels rm, const int* __restrict__ old_keys, int* __restrict__ new_keys, int* __restrict__ new_table, int* __restrict__ new_count, int* __restrict__ status) {
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	int c[kMaxModels];
	bool any = false;
	for(int m = 0; m < rm.n; ++m) {
		c[m] = b < ebc ? rm.out_count[m][b] : 0;
		any |= c[m] > 0;
	}
	// particles per model (the reference's "total number of particles" check, gmpm_simulator.cuh:617): one atomic per wave
	for(int m = 0; m < rm.n; ++m) {
		int t = c[m];
#pragma unroll
		for(int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
		if((threadIdx.x & 63) == 0 && t) atomicAdd(&status[ST_PART0 + m], t);
	}
	if(!any) return;
	const int nb = atomicAdd(new_count, 1);// the compiler turns this into one atomic per wave
	const int kx = old_keys[3 * b], ky = old_keys[3 * b + 1], kz = old_keys[3 * b + 2];
	new_keys[3 * nb]					  = kx;
	new_keys[3 * nb + 1]				  = ky;
	new_keys[3 * nb + 2]				  = kz;
	new_table[key_index(cfg, kx, ky, kz)] = nb;
	for(int m = 0; m < rm.n; ++m) {
		rm.size[m][nb]	 = c[m];
		rm.row_of[m][nb] = b;
		const int nbins	 = (c[m] + kBin - 1) / kBin;
		rm.binoff[m][nb] = nbins ? atomicAdd(&status[ST_BINS0 + m], nbins) : 0;
	}
}

// register_neighbor_blocks / register_exterior_blocks (mgmpm_kernels.cuh:117-151); pbc is read from device memory.
// One LANE per (particle block, offset): the (HI-LO+1)^3 look-ups of a block are independent loads instead of a chain of
// dependent ones in one thread (27 x ~1 us), and the few lanes that really insert share one counter atomic per wave.
template<int LO, int HI>
__global__ __launch_bounds__(256) void register_blocks_kernel(GridCfg cfg, const int* __restrict__ pbc_ptr, int* table, int* keys, int* count, int* status) {
	constexpr int W	   = HI - LO + 1;
	constexpr int NOFF = W * W * W;
	constexpr int LPB  = NOFF <= 8 ? 8 : 32;// lanes per block (27 padded to 32)
	const int pbc	   = *pbc_ptr;
	const long long total = (long long) pbc * LPB;
	for(long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x; t < ((total + 63) & ~63ll); t += (long long) gridDim.x * blockDim.x) {
		const int b = (int) (t / LPB), o = (int) (t % LPB);
		bool claim	= false;
		int x = 0, y = 0, z = 0;
		size_t i = 0;
		if(t < total && o < NOFF) {
			x = keys[3 * b] + LO + o / (W * W);
			y = keys[3 * b + 1] + LO + (o / W) % W;
			z = keys[3 * b + 2] + LO + o % W;
			if(key_ok(cfg, x, y, z)) {
				i = key_index(cfg, x, y, z);
				// Partition::insert, hash_table.cuh:118-127: claim with CAS (cheap pre-check first: most keys exist already)
				if(table[i] == -1) claim = atomicCAS(&table[i], -1, -2) == -1;
			}
		}
		const int idx = wave_append(count, claim);// whole waves reach this point together (the loop bound is wave-aligned)
		if(claim) {
			if(idx < cfg.cap) {
				table[i]		  = idx;
				keys[3 * idx]	  = x;
				keys[3 * idx + 1] = y;
				keys[3 * idx + 2] = z;
			} else {
				table[i] = -1;
				atomicOr(&status[ST_OVERFLOW], 1);
			}
		}
	}
}

// Carry the P2G result (old numbering) into the current grid (new numbering): every NEW neighbour block is written
// exactly once - copied from its old block if it existed, zero otherwise.  Replaces clear_grid +
// mark_active_grid_blocks + copy_selected_grid_blocks (gmpm_simulator.cuh:436-446, :536-541).  One wave per block.
__global__ __launch_bounds__(256) void carry_grid_kernel(GridCfg cfg, const int* __restrict__ new_nbc_ptr, const int* __restrict__ new_keys, const int* __restrict__ old_table, int old_nbc, const float* __restrict__ p2g_grid, float* __restrict__ grid) {
	const int nbc  = min(*new_nbc_ptr, cfg.cap);
	const int lane = threadIdx.x & 63;
	for(int nb = blockIdx.x * 4 + (threadIdx.x >> 6); nb < nbc; nb += gridDim.x * 4) {
		const int old = table_query(cfg, old_table, new_keys[3 * nb], new_keys[3 * nb + 1], new_keys[3 * nb + 2]);
		float4 v	  = {0.f, 0.f, 0.f, 0.f};
		if(old >= 0 && old < old_nbc) {
			const float* s = p2g_grid + (size_t) old * 256;
			v			   = {s[lane], s[64 + lane], s[128 + lane], s[192 + lane]};
		}
		float* d	  = grid + (size_t) nb * 256;
		d[lane]		  = v.x;
		d[64 + lane]  = v.y;
		d[128 + lane] = v.z;
		d[192 + lane] = v.w;
	}
}

// ------------------------------------------------------------------------------------------------------
// Initial setup (gmpm_simulator.cuh:637-781)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void particle_block_key(const GridCfg& cfg, const float* xyz, size_t i, int& bx, int& by, int& bz, int& cellx, int& celly, int& cellz) {
	cellx = node_index(xyz[3 * i], cfg.dx_inv) - 2;
	celly = node_index(xyz[3 * i + 1], cfg.dx_inv) - 2;
	cellz = node_index(xyz[3 * i + 2], cfg.dx_inv) - 2;
	bx	  = cellx / 4;// C++ truncating division as in the reference (Vec.h integer '/')
	by	  = celly / 4;
	bz	  = cellz / 4;
}
// activate_blocks, mgmpm_kernels.cuh:21-34
__global__ void activate_blocks_kernel(GridCfg cfg, size_t n, const float* __restrict__ xyz, int* table, int* keys, int* count, int* status) {
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	int bx, by, bz, cx, cy, cz;
	particle_block_key(cfg, xyz, i, bx, by, bz, cx, cy, cz);
	table_insert(cfg, table, keys, count, bx, by, bz, status);
}
// build_particle_cell_buckets (:36-68) at block granularity: particle id appended to its block's list
__global__ void bucket_particles_kernel(GridCfg cfg, size_t n, const float* __restrict__ xyz, const int* __restrict__ table, int* counts, int* ids, int* status) {
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	int bx, by, bz, cx, cy, cz;
	particle_block_key(cfg, xyz, i, bx, by, bz, cx, cy, cz);
	const int bno = table_query(cfg, table, bx, by, bz);
	if(bno < 0) {
		atomicAdd(&status[ST_LOST], 1);
		return;
	}
	const int slot = atomicAdd(&counts[bno], 1);
	if(slot >= cfg.ppb) {
		atomicOr(&status[ST_OVERFLOW], 2);
		return;
	}
	ids[(size_t) bno * cfg.ppb + slot] = (int) i;
}
// compute_bin_capacity + scan (:86-94) as one atomic allocation per block; also row_of = identity
__global__ void init_bins_kernel(int pbc, const int* __restrict__ counts, int* size, int* row_of, int* binoff_a, int* binoff_b, int* bin_total) {
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	if(b >= pbc) return;
	const int c		= counts[b];
	size[b]			= c;
	row_of[b]		= b;
	const int nbins = (c + kBin - 1) / kBin;
	const int off	= nbins ? atomicAdd(bin_total, nbins) : 0;
	binoff_a[b]		= off;
	binoff_b[b]		= off;
}
// array_to_buffer (:221-323) + init_adv_bucket (:96-104): one workgroup per block
__global__ __launch_bounds__(256) void fill_bins_kernel(GridCfg cfg, int nch, float log_jp0, const float* __restrict__ xyz, const int* __restrict__ ids, const int* __restrict__ size, const int* __restrict__ binoff, float* bins, int* list_in) {
	const int b = blockIdx.x;
	const int n = size[b];
	for(int pidib = threadIdx.x; pidib < n; pidib += blockDim.x) {
		const int pid = ids[(size_t) b * cfg.ppb + pidib];
		float* dst	  = bins + (size_t) (binoff[b] + (pidib >> 6)) * (nch * kBin) + (pidib & 63);
		dst[0]		  = xyz[3 * (size_t) pid];
		dst[kBin]	  = xyz[3 * (size_t) pid + 1];
		dst[2 * kBin] = xyz[3 * (size_t) pid + 2];
		if(nch == 4) {
			dst[3 * kBin] = 1.f;
		} else {
			for(int d = 0; d < 9; ++d) dst[(3 + d) * kBin] = (d % 4 == 0) ? 1.f : 0.f;
			if(nch == 13) dst[12 * kBin] = log_jp0;
		}
		const int cx = node_index(xyz[3 * (size_t) pid], cfg.dx_inv) - 2, cy = node_index(xyz[3 * (size_t) pid + 1], cfg.dx_inv) - 2, cz = node_index(xyz[3 * (size_t) pid + 2], cfg.dx_inv) - 2;
		const int key = (((cx & 3) + 1) * 6 + ((cy & 3) + 1)) * 6 + ((cz & 3) + 1);// stencil base in the block's arena (no motion predicted)
		list_in[(size_t) b * cfg.ppb + pidib] = (kStay << (cfg.pid_bits + kKeyBits)) | (key << cfg.pid_bits) | pidib;
	}
}
// rasterize, mgmpm_kernels.cuh:153-219 (one-time, global atomics)
__global__ void rasterize_kernel(GridCfg cfg, size_t n, const float* __restrict__ xyz, const int* __restrict__ table, float* grid, float mass, float v0x, float v0y, float v0z) {
	const size_t pi = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(pi >= n) return;
	int base[3];
	float w[3][3];
	for(int d = 0; d < 3; ++d) {
		const float p = xyz[3 * pi + d] * cfg.dx_inv;
		base[d]		  = lround_pos(p) - 1;
		bspline_weight_cells(p - (float) base[d], w[d]);
	}
	for(int i = 0; i < 3; ++i)
		for(int j = 0; j < 3; ++j)
			for(int k = 0; k < 3; ++k) {
				const int gx = base[0] + i, gy = base[1] + j, gz = base[2] + k;
				const int bno = table_query(cfg, table, gx >> 2, gy >> 2, gz >> 2);
				if(bno < 0) continue;
				const float wm = mass * (w[0][i] * w[1][j] * w[2][k]);
				float* g	   = grid + (size_t) bno * 256 + (gx & 3) * 16 + (gy & 3) * 4 + (gz & 3);
				unsafeAtomicAdd(g, wm);
				unsafeAtomicAdd(g + 64, wm * v0x);
				unsafeAtomicAdd(g + 128, wm * v0y);
				unsafeAtomicAdd(g + 192, wm * v0z);
			}
}

// ------------------------------------------------------------------------------------------------------
// Output: retrieve_particle_buffer, mgmpm_kernels.cuh:1087-1122 (+ state for the parity tests)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void retrieve_kernel(GridCfg cfg, int nch, const int* __restrict__ cur_keys, const int* __restrict__ prev_table, const int* __restrict__ size, const int* __restrict__ row_of, const int* __restrict__ list_in, const int* __restrict__ binoff_src, const float* __restrict__ bins_src, float* xyz, float* state9, float* logjp, unsigned long long capacity, unsigned long long* counter) {
	const int b = blockIdx.x;
	const int n = size[b];
	if(n == 0) return;
	const int kx = cur_keys[3 * b], ky = cur_keys[3 * b + 1], kz = cur_keys[3 * b + 2];
	const int* list = list_in + (size_t) row_of[b] * cfg.ppb;
	for(int pidib = threadIdx.x; pidib < n; pidib += blockDim.x) {
		const int rec = list[pidib];
		int ox, oy, oz;
		dir_components(rec >> (cfg.pid_bits + kKeyBits), ox, oy, oz);
		const int sp	 = rec & (cfg.ppb - 1);
		const int srcno	 = table_query(cfg, prev_table, kx + ox, ky + oy, kz + oz);
		const float* src = bins_src + (size_t) (binoff_src[srcno] + (sp >> 6)) * (nch * kBin) + (sp & 63);
		const unsigned long long o = atomicAdd(counter, 1ull);
		if(o >= capacity) continue;
		xyz[3 * o]	   = src[0];
		xyz[3 * o + 1] = src[kBin];
		xyz[3 * o + 2] = src[2 * kBin];
		if(state9) {
			if(nch == 4) {
				state9[9 * o] = src[3 * kBin];
				for(int d = 1; d < 9; ++d) state9[9 * o + d] = 0.f;
			} else {
				for(int d = 0; d < 9; ++d) state9[9 * o + d] = src[(3 + d) * kBin];
			}
		}
		if(logjp) logjp[o] = nch == 13 ? src[12 * kBin] : 0.f;
	}
}

__global__ void grid_totals_kernel(int nblocks, const float* __restrict__ grid, double* out) {
	const int lane = threadIdx.x & 63;
	double acc[4]  = {0, 0, 0, 0};
	for(int b = blockIdx.x * 4 + (threadIdx.x >> 6); b < nblocks; b += gridDim.x * 4) {
		const float* g = grid + (size_t) b * 256;
		for(int ch = 0; ch < 4; ++ch) acc[ch] += (double) g[ch * 64 + lane];
	}
	for(int ch = 0; ch < 4; ++ch) {
		double v = acc[ch];
		for(int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
		if(lane == 0) atomicAdd(&out[ch], v);
	}
}

// ------------------------------------------------------------------------------------------------------
// Function-level test kernels (device math vs golden vectors)
// ------------------------------------------------------------------------------------------------------
__global__ void test_svd_kernel(size_t n, const float* __restrict__ Fin, float* __restrict__ out21) {
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	float F[9], U[9], S[3], V[9];
	for(int d = 0; d < 9; ++d) F[d] = Fin[9 * i + d];
	svd3(F, U, S, V);
	for(int d = 0; d < 9; ++d) out21[21 * i + d] = U[d];
	for(int d = 0; d < 3; ++d) out21[21 * i + 9 + d] = S[d];
	for(int d = 0; d < 9; ++d) out21[21 * i + 12 + d] = V[d];
}
__global__ void test_stress_kernel(int material, MaterialConst mc, size_t n, const float* __restrict__ Fin, const float* __restrict__ ljin, float* __restrict__ out19) {
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	float F[9], PF[9];
	for(int d = 0; d < 9; ++d) F[d] = Fin[9 * i + d];
	float lj = ljin ? ljin[i] : 0.f;
	if(material == 1)
		stress_fixed_corotated(mc, F, PF);
	else if(material == 2)
		stress_sand(mc, F, lj, PF);
	else
		stress_nacc(mc, F, lj, PF);
	for(int d = 0; d < 9; ++d) out19[19 * i + d] = F[d];
	for(int d = 0; d < 9; ++d) out19[19 * i + 9 + d] = PF[d];
	out19[19 * i + 18] = lj;
}

}// namespace mpm
