// mpm_kernels.hpp — hand-written gfx950 kernels of the MPM substep hot path.
//
// Design (MI355X-first, not a translation of the reference's CUDA):
//   * wave64 everywhere: one wave == one 4x4x4 grid block (64 cells) in the grid kernels (every channel a 256-B row), one
//     wave == one particle block in G2P2G (mpm_g2p2g.hpp); particles live in 64-slot bins of 16 / 32-B records (+ a row of one or
//     two floats per particle for the solid models, whose state is b = F F^T: mpm_device_math.hpp), one record = one particle = one
//     or two 16-B accesses of its lane;
//   * the P2G scatter is atomic-free: gfx950 serialises ds_add_f32 (193 cycles per wave-instruction,
//     profiles/r01_lds_microbench.txt), so the advection records of a block are counting-sorted by predicted stencil base
//     and dealt to 64-record slices round-robin (prepare_blocks_kernel, once per substep): the 64 lanes of an iteration
//     hold distinct bases, and each lane read-modify-writes its 27 float4 nodes {m, px, py, pz} with plain
//     ds_read_b128 / ds_write_b128 (mpm_g2p2g.hpp); the arena is written back with one hardware f32 atomic per touched
//     node and channel;
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

constexpr int kBin		  = 64; // particle records per bin == wavefront width
constexpr int kG2P2GThreads = 64; // ONE wave per particle block: no cross-wave LDS hazards, no barriers that wait
constexpr int kMaxModels  = 8;
constexpr int kSortKeys		= 216; // sort key = PREDICTED stencil base of the particle in the arena of its block (6^3 values)
constexpr int kKeyBits		= 8;
constexpr int kStay		  = 13; // dir_offset(0,0,0), utility_funcs.hpp:25-27

// status block indices (device ints; read back once per host synchronisation, i.e. once per window of substeps in mpm_run_fixed).
// ST_PBC / ST_NBC / ST_EBC are the PUBLISHED counts of the current partition; during a rebuild the new partition is counted in the
// three phase counters ST_CNT_P (particle blocks), ST_CNT_N (further neighbour blocks), ST_CNT_E (further exterior blocks) - a block
// registered in a later phase gets the number base + counter, so no phase needs a snapshot of a running counter (the rebuild used to
// take three 4-byte device-to-device copies for that) - and each count is published by the first kernel that runs after its phase.
enum {
	ST_PBC = 0, ST_NBC = 1, ST_EBC = 2, ST_OVERFLOW = 3, ST_LOST = 4, ST_ARENA = 5, ST_DROPPED = 6, ST_NONFINITE = 7, ST_BINS0 = 8, ST_PART0 = 16,
	/* 24..28: MPM_G2P2G_STATS */ ST_CNT_P = 29, ST_CNT_N = 30, ST_CNT_E = 31, ST_BINSPREV = 32, ST_PBCPREV = 40, ST_WORDS = 64
};

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


// Inclusive prefix sum over the 64 lanes of a wave on the DPP cross-lane paths - four shifts inside the rows of 16 lanes, then lane 15 of rows 0 / 2 into rows
// 1 / 3 and lane 31 into the upper half (row_bcast15 / row_bcast31: gfx9 only) - six vector instructions; the shuffle version is six ds_bpermute, i.e. six
// dependent LDS round trips (in prepare_blocks_kernel's counting sort: a third of a chunk's exposed latency).  Every lane of the wave must be active.
__device__ __forceinline__ int wave_scan_incl(int v) {
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);// row_shr:1
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);// row_shr:2
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);// row_shr:4
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);// row_shr:8
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);// row_bcast15 into rows 1 and 3
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);// row_bcast31 into rows 2 and 3
	return v;
}

// The same for a whole workgroup (every thread must call it; blockDim.x a multiple of 64, at most 1024): `amount` items per
// thread, ONE global atomic per workgroup.  Returns the thread's first slot (meaningless for amount == 0).
__device__ __forceinline__ int block_append(int* counter, int amount) {
	__shared__ int s_wave[16];
	__shared__ int s_base;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
	const int incl = wave_scan_incl(amount);// inclusive scan over the wave
	__syncthreads();// (s_wave / s_base may still be read by a previous call)
	if(lane == 63) s_wave[wave] = incl;
	__syncthreads();
	if(threadIdx.x == 0) {
		int total = 0;
		for(int w = 0; w < nwaves; ++w) {
			const int c = s_wave[w];
			s_wave[w]	= total;
			total += c;
		}
		s_base = total ? atomicAdd(counter, total) : 0;
	}
	__syncthreads();
	return s_base + s_wave[wave] + incl - amount;
}

// ------------------------------------------------------------------------------------------------------
// Layout of a block's advection list after prepare_blocks_kernel.  The list is cut into chunks of kListChunk slots
// (records [512 c, 512 c + n) of the block); a chunk with n records is laid out as S = ceil(n / 64) slices of 64 slots
// (one G2P2G iteration each), slice s holding n / S + (s < n % S) records in its first slots: records of one sort key go
// to consecutive slices (wrap-around rule), so a slice holds a key twice only if the key has more than S particles.
// The slots behind a slice's records are holes (never read); a block still occupies ceil(size / 64) * 64 slots.
// ------------------------------------------------------------------------------------------------------
constexpr int kListChunk = 512;
// The PAIR layout (round 6; models whose G2P2G carries two particles per lane, mpm_g2p2g_pair.hpp).  A chunk with n records is cut into
// S = ceil(n / 128) slices - one G2P2G iteration each, the fewest there can be - of up to 64 LANE SLOTS; a slot holds a first member A and, most
// of them, a second member B.  With Pf full pairs (two records of one sort key) and n1 = n - 2 Pf singles (the odd record of a key) the slots are,
// in this order:
//   Pf pair slots        in key-major order: both members share their predicted stencil base, G2P2G sums their P2G contributions in registers;
//   X  mismatched slots  X = max(0, Pf + n1 - 64 S): when the singles do not all fit a slot of their own, the last 2 X of them share slots two by
//                        two (different keys: the second member splits from the first and takes the serial path);
//   n1 - 2 X single slots (A only).
// Slot p goes to slice p mod S, lane p / S (the wrap-around rule on slots: a key's pairs land in consecutive slices, a slice holds a key twice only
// if the key has more than S pairs), so in every slice the lanes with a B are the first ones.  A slice is stored DENSE: its A records, then its B
// records; slices one behind the other.  No holes: position == slot in the destination bins == slot in the list, a block occupies exactly `size`
// slots, and the order G2P2G appends the staying particles in IS this layout whatever the shape of the chunk (the sort is skipped for every
// settled block).  The scatter arena of a slot is a bit of its records (kArenaBit): lane parity for the pair slots (a key's two pair slots of one
// slice sit in neighbouring lanes), and for a single the arena its key's pair slot of the same slice does NOT use - a key's single never competes
// with its own pairs.  The one number a reader needs beside the block's size is Pf per chunk: pairinfo[block][chunk], kPairChunks ints per block,
// indexed by the block's number like the look-up row (G2P2G fetches both in its first round trip).
// Chunks of the pair layout: records [512 c, 512 c + 512) like the sliced layout's - but the LAST chunk of a block absorbs a tail of up to kPairTailMerge records
// (a block of 513 .. 768 particles is ONE chunk of 5 or 6 slices).  Mismatched slots arise in FULL chunks only (512 records are 256 slots: every single displaces
// another one), and a column that stood at 512 particles per block is compressed by a few per cent when it flows: 68 % of the blocks of the C3 flow hold more than 512
// particles, most of them 513 - 640, and the few records of their second chunk sat in a slice of their own while the first chunk sent 3.1 % of all particles down the
// serial path as second members of mismatched slots.  With the tail in the same chunk its slice takes the singles: 0.4 % (tools/flow_mismatch.py), the same number of slices.
constexpr int kPairTailMerge = 256;
constexpr int kPairChunkMax	 = 512 + kPairTailMerge;// records of the largest chunk: 6 slices
constexpr int kPairLayoutId	 = 2;					  // (checkpoints: 1 was the layout with plain 512-record chunks)
__host__ __device__ __forceinline__ int pair_chunks(int size) {// chunks of a block with `size` records
	const int nfull = size >> 9, tail = size & 511;
	return nfull + ((tail > 0 && !(nfull >= 1 && tail <= kPairTailMerge)) ? 1 : 0);
}
__host__ __device__ __forceinline__ int pair_chunk_records(int size, int c) {// records of chunk c (which starts at record 512 c)
	return c + 1 < pair_chunks(size) ? 512 : size - 512 * c;
}
constexpr int kPairChunks = 16;// chunks per list row the pair layout supports (ppb <= 8192 = the reference's 128 particles per cell)
constexpr int kArenaBit	  = 30;// (records are {tag 5, key 8, slot <= 13} = 26 bits)
struct PairChunk {// the slices of one chunk in the pair layout (all wave-uniform)
	int n, pf, S, px, L, qb, rb, qa, ra;// px = slots with a B (Pf + X), L = slots; q / r: quotient and remainder of px and L by S
};
__device__ __forceinline__ int div_small(int n, int d) {// n / d for 0 <= n <= 1024, 1 <= d <= 8 (exact, checked exhaustively: tests/test_pair_layout_model.py)
	// ceil(65536 / d) for d = 2..8 as 16-bit fields of two constants: a division by a run-time d, even a wave-uniform one, is ~40 scalar
	// instructions of reciprocal refinement, and G2P2G forms a slice length in every iteration
	const unsigned long long lo = 0x3334400055568000ull, hi = 0x200024932aabull;
	const unsigned inv = d == 1 ? 65536u : (unsigned) ((d < 6 ? lo >> (16 * (d - 2)) : hi >> (16 * (d - 6))) & 0xffffull);
	return (int) (((unsigned) n * inv) >> 16);
}
__device__ __forceinline__ int chunk_records(int size, int chunk) {
	return min(kListChunk, size - chunk * kListChunk);
}
__device__ __forceinline__ PairChunk pair_chunk(int n, int pf) {
	PairChunk c;
	c.n			 = n;
	c.pf		 = pf;
	c.S			 = (n + 127) >> 7;
	const int n1 = n - 2 * pf;
	const int x	 = max(0, pf + n1 - 64 * c.S);
	c.px		 = pf + x;
	c.L			 = pf + n1 - x;
	c.qb		 = div_small(c.px, c.S);
	c.rb		 = c.px - c.qb * c.S;
	c.qa		 = div_small(c.L, c.S);
	c.ra		 = c.L - c.qa * c.S;
	return c;
}
// slice t of the chunk: position of its first record relative to the chunk, lanes with an A, lanes with a B (the first ones; their records sit
// `cnt_a` slots behind the A records)
template<class I>
__device__ __forceinline__ void pair_slice(const PairChunk& c, I t, I& pos, I& cnt_a, I& cnt_b) {
	cnt_a = c.qa + (t < c.ra ? 1 : 0);
	cnt_b = c.qb + (t < c.rb ? 1 : 0);
	pos	  = t * (c.qa + c.qb) + min(t, (I) c.ra) + min(t, (I) c.rb);
}
__device__ __forceinline__ int slice_records(int n, int s) {// records in slice s of a chunk with n records
	const int S = (n + 63) >> 6;
	const int q = div_small(n, S);
	return q + (s < n - q * S ? 1 : 0);
}
// slot of the i-th record (in slice-major order) of a chunk with n records
__device__ __forceinline__ int chunk_slot(int n, int i) {
	const int S = (n + 63) >> 6;
	const int q = div_small(n, S), r = n - q * S;
	const int big = r * (q + 1);
	if(i < big) {
		const int sl = i / (q + 1);
		return sl * 64 + (i - sl * (q + 1));
	}
	const int j = i - big, sl = j / q;
	return (r + sl) * 64 + (j - sl * q);
}
// number of records in the 64-slot slice that starts at slot idx0 of a block with `size` particles
__device__ __forceinline__ int slice_records_at(int size, int idx0) {
	const int chunk = idx0 / kListChunk;
	return slice_records(chunk_records(size, chunk), (idx0 >> 6) & (kListChunk / 64 - 1));
}

// ------------------------------------------------------------------------------------------------------
constexpr int kMaxVelSlots	 = 64;// the running maximum is kept in 64 slots (by workgroup): same-address atomics serialise in L2 (~11 ns each)
constexpr int kMaxVelStride = 32;// ... one slot per 128-B line: the filtering loads of 20 k waves then spread over the L2 channels
// Grid update: momentum -> velocity, gravity, slip walls, max |v|^2.   One wave per grid block, lane = cell.
// (update_grid_velocity_query_max, mgmpm_kernels.cuh:325-420)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grid_update_kernel(GridCfg cfg, const int* __restrict__ nbc_ptr, float* __restrict__ grid, const int* __restrict__ keys, float dt, unsigned* __restrict__ max_vel_bits) {
	// 16 lanes per grid block, 4 cells (one float4) per lane and channel: 16-B accesses, 4 blocks per wave; the neighbour block
	// count is read from device memory (the launch is sized by the host's estimate of it)
	const int sub	  = threadIdx.x & 15;
	const int nblocks = min(*nbc_ptr, cfg.cap);
	float vel_sqr	  = 0.f;
	for(int blockno = (blockIdx.x * 256 + threadIdx.x) >> 4; blockno < nblocks; blockno += gridDim.x * 16) {
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
__global__ __launch_bounds__(256) void grid_update_collision_kernel(GridCfg cfg, const int* __restrict__ nbc_ptr, float* __restrict__ grid, const int* __restrict__ keys, float dt, CollisionObject obj, unsigned* __restrict__ max_vel_bits) {
	const int cell	  = threadIdx.x & 63;
	const int nblocks = min(*nbc_ptr, cfg.cap);
	float vel_sqr	  = 0.f;
	for(int blockno = (blockIdx.x * 256 + threadIdx.x) >> 6; blockno < nblocks; blockno += gridDim.x * 4) {
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
			vel_sqr = fmaxf(vel_sqr, q);
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
// (dense: the pair layout, whose rows have no holes; pairinfo: its per-chunk pair counts, kPairChunks ints per row)
__global__ __launch_bounds__(64) void pack_lists_kernel(int ppb, const int* __restrict__ size, const int* __restrict__ row_of, const long long* __restrict__ offset, const int* __restrict__ list, int* __restrict__ packed, int dense, const int* __restrict__ pairinfo, int* __restrict__ packed_info) {
	const int b = blockIdx.x;
	const int n = size[b];
	const int* row = list + (size_t) row_of[b] * ppb;
	for(int i = threadIdx.x; i < n; i += 64) {
		const int chunk = i / kListChunk;
		packed[offset[b] + i] = dense ? row[i] : row[chunk * kListChunk + chunk_slot(chunk_records(n, chunk), i - chunk * kListChunk)];
	}
	if(dense && threadIdx.x < kPairChunks) packed_info[(size_t) b * kPairChunks + threadIdx.x] = pairinfo[(size_t) b * kPairChunks + threadIdx.x];
}
__global__ __launch_bounds__(64) void unpack_lists_kernel(int ppb, const int* __restrict__ size, int* __restrict__ row_of, const long long* __restrict__ offset, int* __restrict__ list, const int* __restrict__ packed, int dense, int* __restrict__ pairinfo, const int* __restrict__ packed_info) {
	const int b = blockIdx.x;
	const int n = size[b];
	int* row	= list + (size_t) b * ppb;// rows are re-seated at their own block number
	for(int i = threadIdx.x; i < n; i += 64) {
		const int chunk = i / kListChunk;
		if(dense)
			row[i] = packed[offset[b] + i];
		else
			row[chunk * kListChunk + chunk_slot(chunk_records(n, chunk), i - chunk * kListChunk)] = packed[offset[b] + i];
	}
	if(dense && threadIdx.x < kPairChunks) pairinfo[(size_t) b * kPairChunks + threadIdx.x] = packed_info[(size_t) b * kPairChunks + threadIdx.x];
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

__device__ __forceinline__ void dir_components(int dir, int& dx, int& dy, int& dz) {
	dz = (dir % 3) - 1;
	dy = ((dir / 3) % 3) - 1;
	dx = (dir / 9) - 1;
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
	const int* keep[kMaxModels];// (may be null) G2P2G's verdict per block of the PREVIOUS numbering: see prepare_blocks_kernel
	int* pairinfo[kMaxModels];	// non-null: the model's lists are in the pair layout; [block][kPairChunks] full pairs per chunk, written with the sort
	const int* pairhand[kMaxModels];// ... and what G2P2G handed on under the block's PREVIOUS number (a settled block keeps its order and its pair counts)
};
// The launch is sized by a host-side ESTIMATE of the particle block count (the host does not wait for the rebuild's counts,
// mpm_run_fixed); the true count is read from device memory and a workgroup walks over blocks b, b + gridDim.x, ... (one trip
// when the estimate holds).  publish (the status block, or null): the rebuild's exterior phase is over by now - workgroup 0
// publishes the exterior block count (status[ST_EBC], *part_count).
template<bool SORT>
__global__ __launch_bounds__(64) void prepare_blocks_kernel(GridCfg cfg, PrepareModels pm, const int* __restrict__ pbc_ptr, const int* __restrict__ cur_table, const int* __restrict__ cur_keys, const int* __restrict__ prev_table, int* __restrict__ publish, int* __restrict__ part_count) {
	// records are sorted in chunks of 512 (8 per lane): every 64-slot slice of a chunk is one G2P2G iteration
	constexpr int kPrepChunk = kListChunk;
	__shared__ int s_sorted[kPairChunkMax];// (the pair layout's last chunk of a block may hold kPairChunkMax records)
	__shared__ int s_cnt[256];// per key (216 used): count, then first position of the key in key-major order
	__shared__ int s_cur[256];// (pair layout, merged last chunk: the running rank of a key's records in the second pass)
	const int lane = threadIdx.x;
	// (MPM_GROUP_OVERLAP_TAG=1 runs the tagging kernels - halo_mark_all / halo_split_dev - on the comm stream BESIDE this kernel: they read
	//  ST_NBC / ST_PBC only and must never read what is published here, ST_EBC and *part_count)
	if(publish && blockIdx.x == 0 && lane == 0) {
		const int ebc	 = publish[ST_CNT_P] + publish[ST_CNT_N] + publish[ST_CNT_E];
		publish[ST_EBC] = ebc;
		*part_count		 = ebc;
	}
	const int pbc = min(*pbc_ptr, cfg.cap);
	for(int b = blockIdx.x; b < pbc; b += gridDim.x) {
	const int kx = cur_keys[3 * b], ky = cur_keys[3 * b + 1], kz = cur_keys[3 * b + 2];
	// (a settled block's work is two chains of dependent loads - key -> table -> bin offset, and size / row -> keep -> handed-on pair counts -; the first model's
	//  second chain is started HERE, beside the first one, instead of behind it: the kernel is latency-bound at rest)
	int size0 = 0, row0 = 0, keep0 = -2, hand0 = 0;
	if constexpr(SORT) {
		if(pm.n > 0) {
			size0 = pm.size[0][b];
			row0  = pm.row_of[0][b];
			if(pm.keep[0]) keep0 = pm.keep[0][row0];
			if(pm.pairinfo[0]) hand0 = pm.pairhand[0][(size_t) row0 * kPairChunks + (lane & (kPairChunks - 1))];
		}
	}
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
		// (wave-uniform, and said so: the list's address is then a scalar base plus a 32-bit lane offset - one register per load in flight instead of two)
		const int size = __builtin_amdgcn_readfirstlane(m == 0 ? size0 : pm.size[m][b]);
		const int row  = __builtin_amdgcn_readfirstlane(m == 0 ? row0 : pm.row_of[m][b]);
		int* list	   = pm.list[m] + (size_t) row * cfg.ppb;
		// The sort is the identity - and is skipped: 8 B of list traffic per particle and the LDS work - when G2P2G has found that every
		// particle of the block stayed with an unchanged sort key (keep[row] = their number), nobody arrived (the block's new size is that
		// number), and the order G2P2G appended them in IS the sliced layout: slice after slice without holes, i.e. full chunks and a last
		// chunk of full slices or of a single one.  A column at rest is all such blocks but its surface.
		const int tail	= size & (kPrepChunk - 1);
		int* pairinfo	= pm.pairinfo[m] ? pm.pairinfo[m] + (size_t) b * kPairChunks : nullptr;// (pair layout: settled blocks keep their order whatever their shape)
		const int kept	= m == 0 ? keep0 : (pm.keep[m] ? pm.keep[m][row] : -2);
		const bool same = pm.keep[m] && size > 0 && kept == size && (pairinfo || tail <= 64 || (tail & 63) == 0);
		constexpr int NIT = kPrepChunk / 64;// (records per lane of a 512-record group; a merged last chunk of the pair layout is worked off in two groups)
		unsigned recs[NIT];
		auto load_chunk = [&](int chunk0, int nrec) {// unconditional, clamped: all loads of a chunk are in flight together
#pragma unroll
			for(int it = 0; it < NIT; ++it) recs[it] = (unsigned) list[chunk0 + min(it * 64 + lane, nrec - 1)];
		};
		if(size > 0 && !same) load_chunk(0, min(kPrepChunk, size));// (the first 512 records of the first chunk)
		// the bin offsets of the 27 source blocks: the look-up issued at the top has arrived by now, this dependent load
		// overlaps the LDS-only sort below
		if(lane < 27) info = srcno >= 0 ? pm.binoff_src[m][srcno] : -1;
		auto sort_chunk = [&](int chunk0, int nrec) {
#pragma unroll
			for(int q = 0; q < 4; ++q) s_cnt[lane + 64 * q] = 0;
			__syncthreads();
			int rank[NIT];
#pragma unroll
			for(int it = 0; it < NIT; ++it) {
				rank[it] = 0;
				if(it * 64 + lane < nrec) rank[it] = atomicAdd(&s_cnt[(recs[it] >> key_shift) & 255], 1);// ds_add_rtn_u32: integer LDS atomics run at full rate
			}
			__syncthreads();
			{// exclusive prefix sum of the counts over the keys: lane owns keys 4 lane .. 4 lane + 3
				const int c0 = s_cnt[4 * lane], c1 = s_cnt[4 * lane + 1], c2 = s_cnt[4 * lane + 2], c3 = s_cnt[4 * lane + 3];
				const int incl = wave_scan_incl(c0 + c1 + c2 + c3);
				const int excl = incl - (c0 + c1 + c2 + c3);
				__syncthreads();
				s_cnt[4 * lane]		= excl;
				s_cnt[4 * lane + 1] = excl + c0;
				s_cnt[4 * lane + 2] = excl + c0 + c1;
				s_cnt[4 * lane + 3] = excl + c0 + c1 + c2;
			}
			__syncthreads();
			// wrap-around rule: the p-th record in key-major order goes to slice p mod S, position p / S
			const int S = (nrec + 63) >> 6;
			int first[NIT];// (all look-ups in flight together: read one by one at their use they were NIT exposed LDS round trips; the same for the write-back below)
#pragma unroll
			for(int it = 0; it < NIT; ++it) first[it] = s_cnt[(recs[it] >> key_shift) & 255];
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for(int it = 0; it < NIT; ++it) {
				if(it * 64 + lane < nrec) {
					const unsigned rec = recs[it] & rec_mask;
					const int p		   = first[it] + rank[it];
					const int pos	   = div_small(p, S);
					s_sorted[(p - pos * S) * 64 + pos] = (int) rec;
				}
			}
			__syncthreads();
			int out[NIT];
#pragma unroll
			for(int it = 0; it < NIT; ++it) out[it] = s_sorted[it * 64 + lane];
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for(int it = 0; it < NIT; ++it)
				if(it < S && lane < slice_records(nrec, it)) list[chunk0 + it * 64 + lane] = out[it];
			__syncthreads();
		};
		// The pair layout (top of this file): records of one key are dealt out two by two, the odd one of a key goes to the single slices.  nrec <= 512: one group of
		// records, held in registers from the counting pass to the placement.  A merged last chunk (512 < nrec <= kPairChunkMax) is worked off in two groups with the
		// register budget of one: the counting pass only counts, the placement pass reads a group again and draws its ranks from a second set of cursors (s_cur).
		auto sort_chunk_pairs = [&](int chunk0, int nrec) {
			const bool big = nrec > kPrepChunk;// (wave-uniform)
			const int n0 = min(nrec, kPrepChunk), n1g = nrec - n0;// records of group 0 (in recs[]) and of group 1
#pragma unroll
			for(int q = 0; q < 4; ++q) s_cnt[lane + 64 * q] = 0, s_cur[lane + 64 * q] = 0;
			__syncthreads();
			int rank[NIT];
#pragma unroll
			for(int it = 0; it < NIT; ++it) {
				rank[it] = 0;
				if(it * 64 + lane < n0) rank[it] = atomicAdd(&s_cnt[(recs[it] >> key_shift) & 255], 1);
			}
			if(big) {// group 1 (at most kPairTailMerge records) is counted too; its records replace the first ones of group 0 in the registers
#pragma unroll
				for(int it = 0; it < kPairTailMerge / 64; ++it) recs[it] = (unsigned) list[chunk0 + kPrepChunk + min(it * 64 + lane, n1g - 1)];// (only the tail's records: a full load_chunk here costs 14 registers)
#pragma unroll
				for(int it = 0; it < kPairTailMerge / 64; ++it)
					if(it * 64 + lane < n1g) atomicAdd(&s_cnt[(recs[it] >> key_shift) & 255], 1);
			}
			__syncthreads();
			int pf;
			{// per key: its count, and the exclusive prefix sums of the full pairs and the singles before it (one packed scan)
				int c[4], v[4];
#pragma unroll
				for(int i = 0; i < 4; ++i) {
					c[i] = s_cnt[4 * lane + i];
					v[i] = (c[i] >> 1) | ((c[i] & 1) << 16);
				}
				const int incl = wave_scan_incl(v[0] + v[1] + v[2] + v[3]);
				const int tot  = __builtin_amdgcn_readlane(incl, 63);
				pf			  = tot & 0xffff;
				int excl	  = incl - (v[0] + v[1] + v[2] + v[3]);
#pragma unroll
				for(int i = 0; i < 4; ++i) {
					s_cnt[4 * lane + i] = (excl & 0x3ff) | ((excl >> 16) << 10) | (c[i] << 18);// full pairs before the key (<= 384) | singles before it (<= 216) | records of the key (<= 768)
					excl += v[i];
				}
			}
			__syncthreads();
			const PairChunk pc = pair_chunk(nrec, pf);
			const int n1s	   = pc.L - pc.px;// singles with a slot of their own (the last n1 - n1s singles share the mismatched slots two by two)
			auto place = [&](unsigned rec_raw, int r, int e) {
				unsigned rec = rec_raw & rec_mask;
				const int nk = e >> 18, pp = e & 0x3ff;// records of the key, full pairs before it
				int p, member;
				const bool paired = r < (nk & ~1);
				if(paired) {
					p	   = pp + (r >> 1);
					member = r & 1;
				} else {
					const int j = (e >> 10) & 0xff;// singles before this one
					if(j < n1s) {
						p	   = pc.px + j;
						member = 0;
					} else {
						p	   = pc.pf + ((j - n1s) >> 1);
						member = (j - n1s) & 1;
					}
				}
				const int ln = div_small(p, pc.S), d = p - ln * pc.S;
				int pos, ca, cb;
				pair_slice(pc, d, pos, ca, cb);
				// the slot's scatter arena: lane parity - but a single takes the arena its key's pair slot of this slice does not use
				int arena = ln & 1;
				if(!paired) {
					int i0 = d - (pp - div_small(pp, pc.S) * pc.S);// the key's i0-th pair slot is the first one in slice d
					i0 += i0 < 0 ? pc.S : 0;
					if(i0 < (nk >> 1)) arena = 1 ^ (div_small(pp + i0, pc.S) & 1);
				}
				rec |= (unsigned) arena << kArenaBit;
				s_sorted[pos + (member ? ca : 0) + ln] = (int) rec;
			};
			int ent[NIT];// (all look-ups in flight together: read one by one at their use they were NIT exposed LDS round trips; the same for the write-back below)
			if(!big) {
#pragma unroll
				for(int it = 0; it < NIT; ++it) ent[it] = s_cnt[(recs[it] >> key_shift) & 255];
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for(int it = 0; it < NIT; ++it)
					if(it * 64 + lane < n0) place(recs[it], rank[it], ent[it]);
			} else {
				auto place_group = [&](int ng) {// the ng records in recs[]: ranks from the second set of cursors
#pragma unroll
					for(int it = 0; it < NIT; ++it) {
						if(it * 64 + lane < ng) {
							const int key = (recs[it] >> key_shift) & 255;
							const int r	  = atomicAdd(&s_cur[key], 1);
							place(recs[it], r, s_cnt[key]);
						}
					}
				};
				place_group(n1g);// (group 1's records are in the registers)
				load_chunk(chunk0, n0);
				place_group(n0);
			}
			__syncthreads();
			{
				int out[NIT];
#pragma unroll
				for(int it = 0; it < NIT; ++it) out[it] = s_sorted[it * 64 + lane];
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for(int it = 0; it < NIT; ++it)
					if(it * 64 + lane < n0) list[chunk0 + it * 64 + lane] = out[it];
			}
			if(big) {// (the merged tail)
#pragma unroll
				for(int it = 0; it < kPairTailMerge / 64; ++it)
					if(it * 64 + lane < n1g) list[chunk0 + kPrepChunk + it * 64 + lane] = s_sorted[kPrepChunk + it * 64 + lane];
			}
			if(lane == 0) pairinfo[chunk0 / kPrepChunk] = pf;
			__syncthreads();
		};
		if(pairinfo && same && lane < kPairChunks) pairinfo[lane] = m == 0 ? hand0 : pm.pairhand[m][(size_t) row * kPairChunks + lane];
		for(int chunk0 = 0; chunk0 < size && !same; chunk0 += kPrepChunk) {
			const int left = size - chunk0;
			// (pair layout: what is left fits the last chunk - 512 records and a tail of up to kPairTailMerge: pair_chunks)
			const int nrec = (pairinfo && left > kPrepChunk && left <= kPairChunkMax) ? left : min(kPrepChunk, left);
			if(chunk0) load_chunk(chunk0, min(kPrepChunk, nrec));
			if(pairinfo)
				sort_chunk_pairs(chunk0, nrec);
			else
				sort_chunk(chunk0, nrec);
			if(nrec > kPrepChunk) break;
		}
		}
		if constexpr(!SORT) {
			if(lane < 27) info = srcno >= 0 ? pm.binoff_src[m][srcno] : -1;
		}
		pm.blockinfo[m][(size_t) b * kInfoRow + lane] = info;
	}
	}
}

}// namespace mpm
#include "mpm_g2p2g.hpp"
#include "mpm_g2p2g_pair.hpp"
namespace mpm {

// ------------------------------------------------------------------------------------------------------
// Partition rebuild
// ------------------------------------------------------------------------------------------------------
struct RebuildModels {
	int n;
	const int* out_count[kMaxModels];
	int* size[kMaxModels];
	int* row_of[kMaxModels];
	int* binoff[kMaxModels];// destination bin offsets for the NEXT step (new numbering)
	long long bin_cap[kMaxModels];
};

// Everything a substep has to reset, in two kernels instead of the runtime's fill / copy commands (the rebuild alone used to
// enqueue five fills and three 4-byte copies: ~0.05 ms of a 1.8 ms substep, and launch gaps that weigh more the smaller a rank's share):
//   kClearP2G     (before G2P2G):   the P2G accumulation grid and the advection-list append counters;
//   kClearRebuild (before compact): the table of the partition about to be rebuilt - its old keys are UN-INSERTED instead of
//                 an 8 MiB 0xff fill (reset_table, hash_table.cuh:110-112) -, the phase counters, the bin / particle totals (the
//                 previous bin totals are kept: they size the source bins of the next G2P2G), the running max |v|^2 slots of the
//                 fused grid update (an infinite one leaves a sticky flag behind first, gmpm_simulator.cuh:355-358).
// mpm_run_fixed issues both parts in ONE launch before G2P2G (nobody looks at the old table in between).  All sizes are read
// from the status block.
enum { kClearP2G = 1, kClearRebuild = 2, kClearMaxVel = 4 };// kClearMaxVel: the rebuild that follows writes the slots (fused grid update)
struct ClearArgs {
	int flags;
	int nmodels;
	int* out_count[kMaxModels];
	int* keep[kMaxModels];
	float* p2g_grid;
	int* status;
	unsigned* max_vel_bits;
	int* old_table;		  // table of the partition about to be rebuilt
	const int* old_keys;  // its key list
	const int* old_count; // its exterior block count
};
__global__ __launch_bounds__(256) void substep_clear_kernel(GridCfg cfg, ClearArgs a) {
	const int tid = blockIdx.x * 256 + threadIdx.x, nthreads = gridDim.x * 256;
	if(a.flags & kClearP2G) {
		const int nbc = min(a.status[ST_NBC], cfg.cap), ebc = min(a.status[ST_EBC], cfg.cap);
		float4* g	  = reinterpret_cast<float4*>(a.p2g_grid);
		for(int i = tid; i < nbc * 64; i += nthreads) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
		for(int m = 0; m < a.nmodels; ++m)
			for(int i = tid; i <= ebc; i += nthreads) a.out_count[m][i] = 0, a.keep[m][i] = -1;
	}
	if(a.flags & kClearRebuild) {
		const int n = min(*a.old_count, cfg.cap);
		for(int i = tid; i < n; i += nthreads) a.old_table[key_index(cfg, a.old_keys[3 * i], a.old_keys[3 * i + 1], a.old_keys[3 * i + 2])] = -1;
		if(blockIdx.x == gridDim.x - 1) {
			if(threadIdx.x < kMaxModels) {
				a.status[ST_BINSPREV + threadIdx.x] = a.status[ST_BINS0 + threadIdx.x];
				a.status[ST_BINS0 + threadIdx.x]	= 0;
				a.status[ST_PART0 + threadIdx.x]	= 0;
			}
			if(threadIdx.x >= 8 && threadIdx.x < 11) a.status[ST_CNT_P + threadIdx.x - 8] = 0;
			if(threadIdx.x == 11) a.status[ST_PBCPREV] = a.status[ST_PBC];// (particle blocks of the numbering the particle data will be laid out in)
			if(threadIdx.x >= 64 && threadIdx.x < 64 + kMaxVelSlots) {
				unsigned* slot = a.max_vel_bits + (threadIdx.x - 64) * kMaxVelStride;
				if(*slot >= 0x7f800000u) a.status[ST_NONFINITE] = 1;
				if(a.flags & kClearMaxVel) *slot = 0u;// (otherwise the slots keep this substep's grid-update result for the host)
			}
		}
	}
}

// One pass replaces mark_active_particle_blocks + exclusive_scan + exclusive_scan_inverse + update_partition +
// update_buckets + compute_bin_capacity + exclusive_scan (gmpm_simulator.cuh:436-505): blocks that received
// particles get a new number (workgroup-aggregated atomic), their key goes into the new table, their bins are
// allocated.  Order of the new numbering is arbitrary (as it is in the reference: insert order of atomics).
// The old exterior block count is read from the status block; the launch is sized by a host-side estimate of it.
__global__ __launch_bounds__(1024) void compact_blocks_kernel(GridCfg cfg, RebuildModels rm, const int* __restrict__ old_keys, int* __restrict__ new_keys, int* __restrict__ new_table, int* __restrict__ status) {
	// The three counters of this kernel (new block numbers, bins, particle totals) share a cache line: same-line atomics
	// serialise in L2 at ~11 ns each, so they are issued once per 1024-thread workgroup (3 x 95 at C3), not once per wave
	// (3 x 1516 = 33 us, which was the kernel's whole run time).
	__shared__ int s_part[kMaxModels];
	const int ebc = min(status[ST_EBC], cfg.cap);
	for(int b0 = blockIdx.x * blockDim.x; b0 < ebc; b0 += gridDim.x * blockDim.x) {// uniform trip count per workgroup (block_append has barriers)
		const int b = b0 + threadIdx.x;
		int c[kMaxModels];
		bool any = false;
		for(int m = 0; m < rm.n; ++m) {
			c[m] = b < ebc ? rm.out_count[m][b] : 0;
			if(c[m] > cfg.ppb) {// more arrivals than list slots: G2P2G has set the overflow flag and left the surplus records out (drop policy:
								// the block goes on with the ppb particles it has slots for; otherwise the host reports MPM_ERR_CAPACITY)
				atomicAdd(&status[ST_DROPPED], c[m] - cfg.ppb);
				c[m] = cfg.ppb;
			}
			any |= c[m] > 0;
		}
		// particles per model (the reference's "total number of particles" check, gmpm_simulator.cuh:617)
		__syncthreads();
		if(threadIdx.x < kMaxModels) s_part[threadIdx.x] = 0;
		__syncthreads();
		for(int m = 0; m < rm.n; ++m) {
			int t = c[m];
#pragma unroll
			for(int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
			if((threadIdx.x & 63) == 0 && t) atomicAdd(&s_part[m], t);
		}
		__syncthreads();
		if((int) threadIdx.x < rm.n && s_part[threadIdx.x]) atomicAdd(&status[ST_PART0 + threadIdx.x], s_part[threadIdx.x]);
		const int nb = block_append(&status[ST_CNT_P], any ? 1 : 0);
		int kx = 0, ky = 0, kz = 0;
		if(any) {
			kx = old_keys[3 * b], ky = old_keys[3 * b + 1], kz = old_keys[3 * b + 2];
			new_keys[3 * nb]					  = kx;
			new_keys[3 * nb + 1]				  = ky;
			new_keys[3 * nb + 2]				  = kz;
			new_table[key_index(cfg, kx, ky, kz)] = nb;
		}
		for(int m = 0; m < rm.n; ++m) {
			const int nbins = any ? (c[m] + kBin - 1) / kBin : 0;
			const int first = block_append(&status[ST_BINS0 + m], nbins);
			if(any) {
				const bool fits = (long long) first + nbins <= rm.bin_cap[m];// (the host sizes the bins for the worst case and grows them at 3/4: never expected)
				if(!fits) atomicOr(&status[ST_OVERFLOW], 4);
				rm.size[m][nb]	 = fits ? c[m] : 0;
				rm.row_of[m][nb] = b;
				rm.binoff[m][nb] = (nbins && fits) ? first : 0;
			}
		}
	}
}

// register_neighbor_blocks / register_exterior_blocks (mgmpm_kernels.cuh:117-151); pbc is read from device memory.
// One LANE per (particle block, offset): the (HI-LO+1)^3 look-ups of a block are independent loads instead of a chain of
// dependent ones in one thread (27 x ~1 us), and the few lanes that really insert share one counter atomic per workgroup.
// A newly registered block gets the number *pbc_ptr (+ *base2_ptr) + its rank in this phase's own counter; `publish` (may be
// null) receives that base - the count the PREVIOUS phase ended with.
template<int LO, int HI>
__global__ __launch_bounds__(256) void register_blocks_kernel(GridCfg cfg, const int* __restrict__ pbc_ptr, const int* __restrict__ base2_ptr, int* counter, int* publish, int* table, int* keys, int* status) {
	constexpr int W	   = HI - LO + 1;
	constexpr int NOFF = W * W * W;
	constexpr int LPB  = NOFF <= 8 ? 8 : 32;// lanes per block (27 padded to 32)
	const int pbc	   = min(*pbc_ptr, cfg.cap);
	const int base	   = *pbc_ptr + (base2_ptr ? *base2_ptr : 0);
	if(publish && blockIdx.x == 0 && threadIdx.x == 0) *publish = base;
	const long long total = (long long) pbc * LPB;
	const long long bound = (total + blockDim.x - 1) / blockDim.x * blockDim.x;// whole workgroups stay in the loop together (block_append has barriers)
	for(long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x; t < bound; t += (long long) gridDim.x * blockDim.x) {
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
		const int idx = base + block_append(counter, claim ? 1 : 0);// one counter atomic per workgroup (same-address atomics serialise in L2)
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
// UPDATE: the grid update of the NEXT substep (update_grid_velocity_query_max, :325-420; same arithmetic as grid_update_kernel)
// is applied on the way - momentum -> velocity, gravity, slip walls, max |v|^2 - which saves that kernel's own pass over the
// grid.  Used between the substeps of mpm_run_fixed, where the next dt is known and nobody looks at the grid in between.
// Runs between the neighbour and the exterior registration: the new neighbour count is the sum of the first two phase counters, the
// published status[ST_NBC] is still the OLD one (the exterior registration publishes the new one).
template<bool UPDATE>
__global__ __launch_bounds__(256) void carry_grid_kernel(GridCfg cfg, const int* __restrict__ status, const int* __restrict__ new_keys, const int* __restrict__ old_table, const float* __restrict__ p2g_grid, float* __restrict__ grid, float dt, unsigned* __restrict__ max_vel_bits) {
	const int nbc	  = min(status[ST_CNT_P] + status[ST_CNT_N], cfg.cap);
	const int old_nbc = min(status[ST_NBC], cfg.cap);
	const int lane = threadIdx.x & 63;
	float vel_sqr  = 0.f;
	for(int nb = blockIdx.x * 4 + (threadIdx.x >> 6); nb < nbc; nb += gridDim.x * 4) {
		const int kx = new_keys[3 * nb], ky = new_keys[3 * nb + 1], kz = new_keys[3 * nb + 2];
		const int old = table_query(cfg, old_table, kx, ky, kz);
		float4 v	  = {0.f, 0.f, 0.f, 0.f};
		if(old >= 0 && old < old_nbc) {
			const float* s = p2g_grid + (size_t) old * 256;
			v			   = {s[lane], s[64 + lane], s[128 + lane], s[192 + lane]};
		}
		if constexpr(UPDATE) {
			if(v.x > 0.0f) {
				const bool wx = kx < cfg.boundary || kx >= cfg.G - cfg.boundary;
				const bool wy = ky < cfg.boundary || ky >= cfg.G - cfg.boundary;
				const bool wz = kz < cfg.boundary || kz >= cfg.G - cfg.boundary;
				const float mass_inv = 1.f / v.x;
				const float v0		 = wx ? 0.0f : v.y * mass_inv;
				const float v1		 = (wy ? 0.0f : v.z * mass_inv) + cfg.gravity * dt;
				const float v2		 = wz ? 0.0f : v.w * mass_inv;
				v.y					 = v0;
				v.z					 = v1;
				v.w					 = v2;
				float q				 = v0 * v0 + v1 * v1 + v2 * v2;
				if(q != q) q = __builtin_inff();
				vel_sqr = fmaxf(vel_sqr, q);
			}
		}
		float* d	  = grid + (size_t) nb * 256;
		d[lane]		  = v.x;
		d[64 + lane]  = v.y;
		d[128 + lane] = v.z;
		d[192 + lane] = v.w;
	}
	if constexpr(UPDATE) {
#pragma unroll
		for(int off = 32; off > 0; off >>= 1) vel_sqr = fmaxf(vel_sqr, __shfl_xor(vel_sqr, off));
		if(lane == 0 && vel_sqr > 0.f) {
			const unsigned bits = __float_as_uint(vel_sqr);
			unsigned* slot		= max_vel_bits + (blockIdx.x & (kMaxVelSlots - 1)) * kMaxVelStride;
			if(bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
		}
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
// array_to_buffer (:221-323) + init_adv_bucket (:96-104): one workgroup per block.  Record / row layout: mpm_g2p2g.hpp (nch = floats
// per particle in a bin: 4 J-fluid, 9 fixed-corotated, 10 sand / NACC; the solid models start from b = F F^T = I).
__device__ __forceinline__ int rec_floats(int nch) {
	return nch == 4 ? 4 : 8;
}
__global__ __launch_bounds__(256) void fill_bins_kernel(GridCfg cfg, int nch, float log_jp0, const float* __restrict__ xyz, const int* __restrict__ ids, const int* __restrict__ size, const int* __restrict__ binoff, float* bins, int* list_in) {
	const int b = blockIdx.x;
	const int n = size[b];
	for(int pidib = threadIdx.x; pidib < n; pidib += blockDim.x) {
		const int pid = ids[(size_t) b * cfg.ppb + pidib];
		const int rec = rec_floats(nch), row = nch - rec;
		float* bin = bins + (size_t) (binoff[b] + (pidib >> 6)) * (kBin * nch);
		float* dst = bin + (pidib & 63) * rec;
		dst[0]	   = xyz[3 * (size_t) pid] * cfg.dx_inv;// positions live in cell units (x / dx: exact, dx is a power of two) - G2P2G's index and
		dst[1]	   = xyz[3 * (size_t) pid + 1] * cfg.dx_inv;// weight arithmetic is in cell units anyway
		dst[2]	   = xyz[3 * (size_t) pid + 2] * cfg.dx_inv;
		dst[3]	   = 1.f;// J, or b00
		if(nch != 4) {
			dst[4] = 1.f;// b11
			dst[5] = 1.f;// b22
			dst[6] = 0.f;// b10
			dst[7] = 0.f;// b20
			bin[kBin * rec + (pidib & 63) * row] = 0.f;// b21
			if(row == 2) bin[kBin * rec + (pidib & 63) * row + 1] = log_jp0;
		}
		const int cx = node_index(xyz[3 * (size_t) pid], cfg.dx_inv) - 2, cy = node_index(xyz[3 * (size_t) pid + 1], cfg.dx_inv) - 2, cz = node_index(xyz[3 * (size_t) pid + 2], cfg.dx_inv) - 2;
		const int key = (((cy & 3) + 1) * 6 + ((cx & 3) + 1)) * 6 + ((cz & 3) + 1);// stencil base in the block's node cube, y slowest (mpm_g2p2g.hpp); no motion predicted
		list_in[(size_t) b * cfg.ppb + pidib] = (kStay << (cfg.pid_bits + kKeyBits)) | (key << cfg.pid_bits) | pidib;
	}
}
// rasterize, mgmpm_kernels.cuh:153-219, block by block (round 6): one workgroup per particle block - the particles are bucketed by block when this
// runs (ids[block][slot], size[block]) - sums the B-spline masses of the block's particles in an LDS image of the 8^3 node cube its stencils reach
// (nodes 1..6 per axis, like G2P2G's arenas) and writes every touched node back with ONE global atomic per channel; the initial velocity is uniform
// per model, so the three momentum channels are the mass times v0.  The per-particle version below (27 x 4 global atomics per particle) took 98.6 ms
// at 40 M particles and 253.7 ms at 100 M - 40 % of a 110-substep profile run; it stays as the fall-back for a particle outside every block.
__global__ __launch_bounds__(256) void rasterize_blocks_kernel(GridCfg cfg, const float* __restrict__ xyz, const int* __restrict__ ids, const int* __restrict__ size, const int* __restrict__ keys, const int* __restrict__ table, float* grid, float mass, float v0x, float v0y, float v0z) {
	__shared__ float s_m[512];
	const int b = blockIdx.x;
	const int n = size[b];
	if(n == 0) return;
	const int kx = keys[3 * b], ky = keys[3 * b + 1], kz = keys[3 * b + 2];
	for(int i = threadIdx.x; i < 512; i += 256) s_m[i] = 0.f;
	__syncthreads();
	for(int pidib = threadIdx.x; pidib < n; pidib += 256) {
		const size_t pid = (size_t) ids[(size_t) b * cfg.ppb + pidib];
		int base[3];
		float w[3][3];
#pragma unroll
		for(int d = 0; d < 3; ++d) {
			const float p = xyz[3 * pid + d] * cfg.dx_inv;
			base[d]		  = lround_pos(p) - 1;
			bspline_weight_cells(p - (float) base[d], w[d]);
		}
		const int lx = base[0] - 4 * kx, ly = base[1] - 4 * ky, lz = base[2] - 4 * kz;// 1..4: the block owns the cells base - 1
#pragma unroll
		for(int i = 0; i < 3; ++i)
#pragma unroll
			for(int j = 0; j < 3; ++j)
#pragma unroll
				for(int k = 0; k < 3; ++k) atomicAdd(&s_m[((lx + i) << 6) | ((ly + j) << 3) | (lz + k)], mass * (w[0][i] * w[1][j] * w[2][k]));
	}
	__syncthreads();
	for(int i = threadIdx.x; i < 512; i += 256) {
		const float m = s_m[i];
		if(m == 0.f) continue;
		const int x = i >> 6, y = (i >> 3) & 7, z = i & 7;
		const int bno = table_query(cfg, table, kx + (x >> 2), ky + (y >> 2), kz + (z >> 2));
		if(bno < 0) continue;
		float* g = grid + (size_t) bno * 256 + (x & 3) * 16 + (y & 3) * 4 + (z & 3);
		unsafeAtomicAdd(g, m);
		if(v0x != 0.f) unsafeAtomicAdd(g + 64, m * v0x);
		if(v0y != 0.f) unsafeAtomicAdd(g + 128, m * v0y);
		if(v0z != 0.f) unsafeAtomicAdd(g + 192, m * v0z);
	}
}
// (per particle, global atomics: the reference's form)
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
__global__ __launch_bounds__(256) void retrieve_kernel(GridCfg cfg, int nch, const int* __restrict__ cur_keys, const int* __restrict__ prev_table, const int* __restrict__ size, const int* __restrict__ row_of, const int* __restrict__ list_in, const int* __restrict__ binoff_src, const float* __restrict__ bins_src, float* xyz, float* state9, float* logjp, unsigned long long capacity, unsigned long long* counter, int dense) {
	const int b = blockIdx.x;
	const int n = size[b];
	if(n == 0) return;
	const int kx = cur_keys[3 * b], ky = cur_keys[3 * b + 1], kz = cur_keys[3 * b + 2];
	const int* list = list_in + (size_t) row_of[b] * cfg.ppb;
	for(int pidib = threadIdx.x; pidib < ((n + 63) & ~63); pidib += blockDim.x) {
		if(dense ? pidib >= n : (pidib & 63) >= slice_records_at(n, pidib & ~63)) continue;// a hole of the sliced list layout (the pair layout has none)
		const int rec = list[pidib];
		int ox, oy, oz;
		dir_components((rec >> (cfg.pid_bits + kKeyBits)) & 31, ox, oy, oz);
		const int sp	 = rec & (cfg.ppb - 1);
		const int srcno	 = table_query(cfg, prev_table, kx + ox, ky + oy, kz + oz);
		const int recf = rec_floats(nch), rowf = nch - recf;
		const float* bin = bins_src + (size_t) (binoff_src[srcno] + (sp >> 6)) * (kBin * nch);
		const float* src = bin + (sp & 63) * recf;
		const float* row = bin + kBin * recf + (sp & 63) * rowf;
		const unsigned long long o = atomicAdd(counter, 1ull);
		if(o >= capacity) continue;
		xyz[3 * o]	   = src[0] * cfg.dx;// (stored in cell units)
		xyz[3 * o + 1] = src[1] * cfg.dx;
		xyz[3 * o + 2] = src[2] * cfg.dx;
		if(state9) {
			if(nch == 4) {
				state9[9 * o] = src[3];
				for(int d = 1; d < 9; ++d) state9[9 * o + d] = 0.f;
			} else {// b = F F^T as a full symmetric matrix (the sign of b00 marks a reflected F: reported as it is stored)
				const float s6[6] = {src[3], src[4], src[5], src[6], src[7], row[0]};
				float m[9];
				sym_expand(s6, m);
				for(int d = 0; d < 9; ++d) state9[9 * o + d] = m[d];
			}
		}
		if(logjp) logjp[o] = rowf == 2 ? row[1] : 0.f;
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
// (inputs are deformation gradients F: the kernels form b = F F^T and the reflection flag the way a particle would carry them)
__global__ void test_eig_kernel(size_t n, const float* __restrict__ Fin, float* __restrict__ out12) {
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	const size_t j = i < n ? i : n - 1;// every lane runs the decomposition (its sweep count is wave-uniform)
	float F[9], b[6], lam[3], U[9];
	for(int d = 0; d < 9; ++d) F[d] = Fin[9 * j + d];
	left_cauchy_green(F, b);
	NoHook nh;
	bool undeformed;
	sym_eig3<0>(b, lam, U, nh, undeformed);
	if(i >= n) return;
	for(int d = 0; d < 9; ++d) out12[12 * i + d] = U[d];
	for(int d = 0; d < 3; ++d) out12[12 * i + 9 + d] = lam[d];
}
// out19: the updated b (full symmetric matrix), P F^T vol, log Jp; out_refl (may be null): the reflection flag after the update
__global__ void test_stress_kernel(int material, MaterialConst mc, size_t n, const float* __restrict__ Fin, const float* __restrict__ ljin, float* __restrict__ out19, int* __restrict__ out_refl) {
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	const size_t j = i < n ? i : n - 1;// (the stress functions hold wave-level votes: every lane of a wave runs them)
	float F[9], b[6], PF[9], bm[9];
	for(int d = 0; d < 9; ++d) F[d] = Fin[9 * j + d];
	left_cauchy_green(F, b);
	bool refl = det3(F) < 0.f;
	float lj  = ljin ? ljin[j] : 0.f;
	if(material == 1)
		stress_fixed_corotated(mc, b, refl, PF);
	else if(material == 2)
		stress_sand(mc, b, refl, lj, PF);
	else
		stress_nacc(mc, b, refl, lj, PF);
	if(i >= n) return;
	sym_expand(b, bm);
	for(int d = 0; d < 9; ++d) out19[19 * i + d] = bm[d];
	for(int d = 0; d < 9; ++d) out19[19 * i + 9 + d] = PF[d];
	out19[19 * i + 18] = lj;
	if(out_refl) out_refl[i] = refl ? 1 : 0;
}

}// namespace mpm
