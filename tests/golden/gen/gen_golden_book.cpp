// Golden-table generator for the INTEGER BOOKKEEPING of the reference's particle path (G20; runs ONLY in the build container, where
// /root/reference is mounted; gen_golden_book.sh is the build line).
//
// Like G7 / G16-G19 the statements are cut out of the reference's files AS TEXT into scratch includes and compiled here; what this file
// adds is scaffolding only: a serial thread loop in place of a kernel launch (blockIdx / threadIdx as globals, __syncthreads a no-op,
// __shared__ = static), atomicAdd / atomicSub / atomicCAS / atomic_agg_inc as plain operations, a Partition and a particle-buffer
// struct that carry the member NAMES the text uses (index(), active_keys, count, cell_particle_counts, cellbuckets, ...) over plain
// arrays, and the host-side glue of gmpm_simulator.cuh:655-745 / :421-570 (memsets, the two thrust exclusive scans as std loops, the
// launch order).  The table row-major key -> index mapping of the Structural DSL is scaffolding too (pinned separately by G2).
//
// Scene: 216 particles in 5 clumps on the reference's compile-time grid (256^3, dx = 1/256, 128 particles per cell, bins of 32).
//   A  activate_blocks, register_neighbor_blocks, register_exterior_blocks        (initial partition: Partition::insert)
//   B  build_particle_cell_buckets, cell_bucket_to_block, compute_bin_capacity + scan   (initial buckets)
//   C  one add_advection per bucketed particle: its cell moved by a per-particle offset in {-1,0,1}^3, dirtag from dir_offset - what g2p2g
//      hands over (mgmpm_kernels.cuh:852-866)
//   D  the rebuild of gmpm_simulator.cuh:421-570: cell_bucket_to_block, mark_active_particle_blocks, exclusive_scan, exclusive_scan_inverse,
//      reset_table, update_partition (Partition::reinsert), update_buckets, compute_bin_capacity + scan, register_neighbor / exterior
// A serial thread loop fixes ONE of the orders the GPU's atomics may produce; the consumers compare block keys and per-block particle sets
// as SETS and everything order-free (counts, scans, bin offsets, table consistency) exactly.
#include "cuda_host_shim.h"

#include <array>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "utility_funcs.hpp"

struct Dim3 {
	unsigned x = 0, y = 0, z = 0;
};
static Dim3 blockIdx, threadIdx, blockDim, gridDim;
static inline void __syncthreads() {}
#define __shared__ static
static inline int atomicAdd(int* p, int v) {
	const int o = *p;
	*p			= o + v;
	return o;
}
static inline int atomicSub(int* p, int v) {
	const int o = *p;
	*p			= o - v;
	return o;
}
static inline int atomicCAS(int* p, int cmp, int val) {
	const int o = *p;
	if(o == cmp) *p = val;
	return o;
}
#define PRINT_NEGATIVE_BLOGNOS 0
#define PRINT_CELL_OVERFLOW 0

namespace book {// (its own namespace: the reference's Structural DSL already owns the names Instance / Partition in mn)
using namespace mn;
using namespace mn::placeholder;
template<typename T>
T atomic_agg_inc(T* p) {// (DeviceUtils.cuh:182-195: one atomicAdd per coalesced group; serially: a plain post-increment)
	return (*p)++;
}

// ---- scaffolding: the names the cut-out member functions use ---------------------------------------------------------------------
struct block_partition_ {};
template<typename T>
struct Instance {
	int* count;
};
constexpr int kG = config::G_GRID_SIZE;// blocks per axis (64)
template<int Opt>
struct Partition : Instance<block_partition_> {
	using value_t = int;
	using key_t	  = ivec3;
	static constexpr value_t sentinel_v = -1;
	int* index_table;
	key_t* active_keys;
	value_t& index(const key_t& key) const {
		return index_table[((size_t) key[0] * kG + key[1]) * kG + key[2]];
	}
#include "part_methods.inc"
};
struct ParticleBuffer {
	int* cell_particle_counts;
	int* cellbuckets;
	int* particle_bucket_sizes;
	int* blockbuckets;
	int* bin_offsets;
#include "add_advection.inc"
};
struct ParticleArray {
	const float* xyz;
	template<typename I>
	float val(I, uint32_t i) const {
		return xyz[3 * (size_t) i + I::value];
	}
};

#include "kernels.inc"
}// namespace book

using namespace book;

template<typename F>
static void launch(unsigned grid, unsigned block, F&& body) {
	gridDim.x = grid, blockDim.x = block;
	for(unsigned b = 0; b < grid; ++b)
		for(unsigned t = 0; t < block; ++t) {
			blockIdx.x = b, threadIdx.x = t;
			body();
		}
}
static void exclusive_scan_host(int n, const int* in, int* out) {// (gmpm_simulator.cuh: thrust::exclusive_scan)
	int acc = 0;
	for(int i = 0; i < n; ++i) {
		out[i] = acc;
		acc += in[i];
	}
}
static uint32_t hash32(uint32_t x) {
	x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
	return x;
}
template<typename T>
static void dump(const std::string& dir, const char* name, const std::vector<T>& v) {
	FILE* f = fopen((dir + "/" + name).c_str(), "wb");
	if(!f || fwrite(v.data(), sizeof(T), v.size(), f) != v.size()) {
		fprintf(stderr, "cannot write %s\n", name);
		exit(1);
	}
	fclose(f);
}

struct Store {// the arrays behind one Partition + one ParticleBuffer
	std::vector<int> table, count, cellcnt, cellb, sizes, blockb, binoff;
	std::vector<ivec3> keys;
	Partition<1> P {};
	ParticleBuffer B {};
	explicit Store(int cap)
		: table((size_t) kG * kG * kG, -1)
		, count(1, 0)
		, cellcnt((size_t) cap * config::G_BLOCKVOLUME, 0)
		, cellb((size_t) cap * config::G_PARTICLE_NUM_PER_BLOCK, 0)
		, sizes((size_t) cap + 1, 0)
		, blockb((size_t) cap * config::G_PARTICLE_NUM_PER_BLOCK, 0)
		, binoff((size_t) cap + 1, 0)
		, keys((size_t) cap) {
		P.count		  = count.data();
		P.index_table = table.data();
		P.active_keys = keys.data();
		B.cell_particle_counts	= cellcnt.data();
		B.cellbuckets			= cellb.data();
		B.particle_bucket_sizes = sizes.data();
		B.blockbuckets			= blockb.data();
		B.bin_offsets			= binoff.data();
	}
};

int main(int argc, char** argv) {
	const std::string out = argc > 1 ? argv[1] : ".";
	static_assert(config::G_DOMAIN_BITS == 8 && config::G_MAX_PARTICLES_IN_CELL == 128 && config::G_BIN_CAPACITY == 32, "the table is quoted for the reference's compile-time configuration");
	const int cap = 512;
	// ---- the scene: five clumps of a 0.5-cell lattice, one of them straddling a block corner, one a block face; positions exact in float
	std::vector<float> xyz;
	const float dx = 1.f / 256.f;
	const int clumps[5][6] = {{40, 40, 40, 4, 3, 3}, {62, 62, 62, 4, 4, 3}, {90, 41, 60, 6, 2, 2}, {91, 47, 60, 3, 3, 4}, {120, 120, 33, 3, 2, 5}};// first cell, extent in half cells
	for(const auto& c: clumps)
		for(int i = 0; i < c[3]; ++i)
			for(int j = 0; j < c[4]; ++j)
				for(int k = 0; k < c[5]; ++k) {
					xyz.push_back((c[0] + 0.25f + 0.5f * i) * dx);
					xyz.push_back((c[1] + 0.25f + 0.5f * j) * dx);
					xyz.push_back((c[2] + 0.25f + 0.5f * k) * dx);
				}
	const uint32_t n = (uint32_t) (xyz.size() / 3);
	ParticleArray arr {xyz.data()};
	Store cur(cap), nxt(cap);
	std::vector<int> hdr;
	// ---- A: the initial partition (gmpm_simulator.cuh:655-745)
	launch((n + 255) / 256, 256, [&] { activate_blocks(n, arr, cur.P); });
	const int pbc0 = cur.count[0];
	// ---- B: the initial buckets
	launch((n + 255) / 256, 256, [&] { build_particle_cell_buckets(n, arr, cur.B, cur.P); });
	std::fill(cur.sizes.begin(), cur.sizes.begin() + pbc0 + 1, 0);
	launch(pbc0, config::G_BLOCKVOLUME, [&] { cell_bucket_to_block(cur.B.cell_particle_counts, cur.B.cellbuckets, cur.B.particle_bucket_sizes, cur.B.blockbuckets); });
	std::vector<int> bin_sizes((size_t) cap + 1, 0);
	launch(pbc0 / config::G_PARTICLE_BATCH_CAPACITY + 1, config::G_PARTICLE_BATCH_CAPACITY, [&] { compute_bin_capacity((uint32_t) (pbc0 + 1), (const int*) cur.B.particle_bucket_sizes, bin_sizes.data()); });
	exclusive_scan_host(pbc0 + 1, bin_sizes.data(), cur.B.bin_offsets);
	launch((pbc0 + config::G_PARTICLE_BATCH_CAPACITY - 1) / config::G_PARTICLE_BATCH_CAPACITY, config::G_PARTICLE_BATCH_CAPACITY, [&] { register_neighbor_blocks((uint32_t) pbc0, cur.P); });
	const int nbc0 = cur.count[0];
	launch((pbc0 + config::G_PARTICLE_BATCH_CAPACITY - 1) / config::G_PARTICLE_BATCH_CAPACITY, config::G_PARTICLE_BATCH_CAPACITY, [&] { register_exterior_blocks((uint32_t) pbc0, cur.P); });
	const int ebc0 = cur.count[0];
	std::vector<int> keys0, sizes0, buckets0, binoff0;
	for(int b = 0; b < ebc0; ++b)
		for(int d = 0; d < 3; ++d) keys0.push_back(cur.keys[b][d]);
	for(int b = 0; b < pbc0; ++b) {
		sizes0.push_back(cur.sizes[b]);
		for(int i = 0; i < cur.sizes[b]; ++i) buckets0.push_back(cur.blockb[(size_t) b * config::G_PARTICLE_NUM_PER_BLOCK + i]);// particle ids
	}
	for(int b = 0; b <= pbc0; ++b) binoff0.push_back(cur.binoff[b]);
	// ---- C: what g2p2g hands to add_advection (mgmpm_kernels.cuh:852-866): the particle's cell moved by at most one cell per axis
	std::vector<int> delta(3 * (size_t) n), adv;// adv: per call {source block, particle_id_in_block, particle id, new cell x y z, dirtag}
	for(uint32_t p = 0; p < n; ++p)
		for(int d = 0; d < 3; ++d) delta[3 * p + d] = (int) (hash32(3 * p + d + 12345u) % 3u) - 1;
	std::fill(nxt.cellcnt.begin(), nxt.cellcnt.end(), 0);
	for(int b = 0; b < pbc0; ++b)
		for(int pidib = 0; pidib < cur.sizes[b]; ++pidib) {
			const int pid		= cur.blockb[(size_t) b * config::G_PARTICLE_NUM_PER_BLOCK + pidib];
			const ivec3 cell	= get_block_id({xyz[3 * pid], xyz[3 * pid + 1], xyz[3 * pid + 2]}) - 2;// (base_index - 1 of g2p2g: :774-777)
			const ivec3 newcell = cell + ivec3 {delta[3 * pid], delta[3 * pid + 1], delta[3 * pid + 2]};
			const ivec3 bd		= cell / (int) config::G_BLOCKSIZE - newcell / (int) config::G_BLOCKSIZE;
			const int dirtag	= dir_offset({bd[0], bd[1], bd[2]});// (:860-862)
			nxt.B.add_advection(cur.P, newcell, dirtag, pidib);
			const int row[7] = {b, pidib, pid, newcell[0], newcell[1], newcell[2], dirtag};
			adv.insert(adv.end(), row, row + 7);
		}
	// ---- D: the rebuild (gmpm_simulator.cuh:421-570); nxt = particle_bins[(rollid + 1) % 2] + partitions[(rollid + 1) % 2]
	std::fill(nxt.sizes.begin(), nxt.sizes.begin() + ebc0 + 1, 0);
	launch(ebc0, config::G_BLOCKVOLUME, [&] { cell_bucket_to_block(nxt.B.cell_particle_counts, nxt.B.cellbuckets, nxt.B.particle_bucket_sizes, nxt.B.blockbuckets); });
	std::vector<int> sources((size_t) cap + 1, 0), destinations((size_t) cap + 1, 0);
	launch(ebc0 / config::G_PARTICLE_BATCH_CAPACITY + 1, config::G_PARTICLE_BATCH_CAPACITY, [&] { mark_active_particle_blocks((uint32_t) (ebc0 + 1), (const int*) nxt.B.particle_bucket_sizes, sources.data()); });
	std::vector<int> marks(sources.begin(), sources.begin() + ebc0 + 1);
	exclusive_scan_host(ebc0 + 1, sources.data(), destinations.data());
	const int pbc1 = destinations[ebc0];
	nxt.count[0]   = pbc1;
	launch((ebc0 + 255) / 256, 256, [&] { exclusive_scan_inverse(ebc0, (const int*) destinations.data(), sources.data()); });
	std::fill(nxt.table.begin(), nxt.table.end(), -1);// reset_table
	launch((pbc1 + config::G_PARTICLE_BATCH_CAPACITY - 1) / config::G_PARTICLE_BATCH_CAPACITY, config::G_PARTICLE_BATCH_CAPACITY, [&] { update_partition((uint32_t) pbc1, (const int*) sources.data(), cur.P, nxt.P); });
	// (update_buckets copies from the NEXT buffer, where add_advection wrote, into the current one under the new numbering: :481-487)
	launch(pbc1, config::G_PARTICLE_BATCH_CAPACITY, [&] { update_buckets((uint32_t) pbc1, (const int*) sources.data(), nxt.B, cur.B); });
	launch(pbc1 / config::G_PARTICLE_BATCH_CAPACITY + 1, config::G_PARTICLE_BATCH_CAPACITY, [&] { compute_bin_capacity((uint32_t) (pbc1 + 1), (const int*) cur.B.particle_bucket_sizes, bin_sizes.data()); });
	exclusive_scan_host(pbc1 + 1, bin_sizes.data(), cur.B.bin_offsets);
	launch((pbc1 + config::G_PARTICLE_BATCH_CAPACITY - 1) / config::G_PARTICLE_BATCH_CAPACITY, config::G_PARTICLE_BATCH_CAPACITY, [&] { register_neighbor_blocks((uint32_t) pbc1, nxt.P); });
	const int nbc1 = nxt.count[0];
	launch((pbc1 + config::G_PARTICLE_BATCH_CAPACITY - 1) / config::G_PARTICLE_BATCH_CAPACITY, config::G_PARTICLE_BATCH_CAPACITY, [&] { register_exterior_blocks((uint32_t) pbc1, nxt.P); });
	const int ebc1 = nxt.count[0];
	std::vector<int> keys1, sizes1, buckets1, binoff1, src1, dst1;
	for(int b = 0; b < ebc1; ++b)
		for(int d = 0; d < 3; ++d) keys1.push_back(nxt.keys[b][d]);
	for(int b = 0; b < pbc1; ++b) {
		sizes1.push_back(cur.sizes[b]);
		for(int i = 0; i < cur.sizes[b]; ++i) buckets1.push_back(cur.blockb[(size_t) b * config::G_PARTICLE_NUM_PER_BLOCK + i]);// (dirtag * 8192) | particle_id_in_block of the SOURCE block
	}
	for(int b = 0; b <= pbc1; ++b) binoff1.push_back(cur.binoff[b]);
	for(int b = 0; b <= ebc0; ++b) dst1.push_back(destinations[b]);
	for(int b = 0; b < pbc1; ++b) src1.push_back(sources[b]);
	// table consistency of both partitions: query(active_keys[i]) == i, nothing else set
	for(int which = 0; which < 2; ++which) {
		const Store& s = which ? nxt : cur;
		const int cnt  = which ? ebc1 : ebc0;
		size_t set	   = 0;
		for(int v: s.table) set += v != -1;
		if(set != (size_t) cnt) return fprintf(stderr, "table %d: %zu entries for %d keys\n", which, set, cnt), 1;
		for(int i = 0; i < cnt; ++i)
			if(s.P.query(s.keys[i]) != i) return fprintf(stderr, "table %d: query(active_keys[%d]) != %d\n", which, i, i), 1;
	}
	hdr = {(int) n, pbc0, nbc0, ebc0, pbc1, nbc1, ebc1, (int) (adv.size() / 7)};
	dump(out, "g20_hdr.i32", hdr);
	dump(out, "g20_xyz.f32", xyz);
	dump(out, "g20_delta.i32", delta);
	dump(out, "g20_keys0.i32", keys0);
	dump(out, "g20_sizes0.i32", sizes0);
	dump(out, "g20_buckets0.i32", buckets0);
	dump(out, "g20_binoff0.i32", binoff0);
	dump(out, "g20_adv.i32", adv);
	dump(out, "g20_marks.i32", marks);
	dump(out, "g20_scan.i32", dst1);
	dump(out, "g20_scan_inverse.i32", src1);
	dump(out, "g20_keys1.i32", keys1);
	dump(out, "g20_sizes1.i32", sizes1);
	dump(out, "g20_buckets1.i32", buckets1);
	dump(out, "g20_binoff1.i32", binoff1);
	printf("G20: %u particles; blocks %d / %d / %d -> %d / %d / %d; %zu add_advection calls\n", n, pbc0, nbc0, ebc0, pbc1, nbc1, ebc1, adv.size() / 7);
	return 0;
}
