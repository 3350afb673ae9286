// Golden-vector generator for the BODY of the reference's g2p2g kernel and of its grid update (runs ONLY in the build container,
// where /root/reference is mounted; gen_golden_kernel.sh is the build line).
//
// Projects/GMPM/mgmpm_kernels.cuh holds every CUDA kernel of the project and cannot be host-compiled as a whole.  Like G7, the
// statements of interest are cut out of that file AS TEXT by gen_golden_kernel.sh into scratch includes and compiled here between
// locals that carry the names the text uses; everything those statements call (bspline_weight, get_block_id, dir_offset,
// matrix_matrix_multiplication_3d, compute_stress<...>, math::svd, the vec<> arithmetic) is the reference's own code, included from
// where it lies.  What this file adds is scaffolding only: the arenas as plain arrays, a bin accessor with the Structural DSL's
// val(_k, i) spelling, a recorder for add_advection, and atomicAdd as a plain add (one particle per scene: no concurrency).
//
//   G16  gather + advect                       mgmpm_kernels.cuh:772-838   (gather.inc)
//   G17  contrib line, re-bucketing, scatter    mgmpm_kernels.cuh:845-905   (scatter.inc)
//   G18  J-fluid / FC / sand / NACC bodies        mgmpm_kernels.cuh:473-663   (body_jfluid.inc, body_fc.inc, body_sand.inc, body_nacc.inc)
//   G19  grid-update cell arithmetic            mgmpm_kernels.cuh:353-388   (gridcell.inc)
// One row of G16-G18 is ONE particle walked through the kernel's whole per-particle body (the three text blocks in the kernel's own
// order, sharing locals exactly as they do there); the files record the inputs and every intermediate.
#include "cuda_host_shim.h"

#include <array>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "constitutive_models.cuh"
#include "utility_funcs.hpp"

static inline float atomicAdd(float* p, float v) {// (single particle per scene: the kernel's shared-memory atomic is a plain add here)
	const float o = *p;
	*p			  = o + v;
	return o;
}

namespace {
using namespace mn;
using namespace mn::placeholder;// _0 ... _12 (Library/MnBase/Meta/TypeMeta.h:139-151)

// ---- scaffolding: the names the cut-out statements use --------------------------------------------------------------------------
struct CalculateContributionAndStoreParticleDataIntermediate {// mgmpm_kernels.cuh:465-468
	vec3 pos;
	float J;
};
struct BinRef {// what particle_buffer.ch(_0, binno) hands out: 13 channels x G_BIN_CAPACITY slots
	float (*chan)[config::G_BIN_CAPACITY];
	template<typename I>
	float& val(I, int i) const {
		return chan[I::value][i];
	}
};
struct Buffer {// the members of ParticleBuffer<M> the statements read (particle_buffer.cuh:141-264)
	float (*bins)[13][config::G_BIN_CAPACITY];
	const int* bin_offsets;
	float volume, mass, mu, lambda;
	float cohesion, beta, yield_surface;
	bool volume_correction;
	float bm, xi, msqr;
	bool hardening_on;
	float bulk, gamma, viscosity;// J_FLUID (particle_buffer.cuh:148-153)
	template<typename I>
	BinRef ch(I, int binno) const {
		return BinRef {bins[binno]};
	}
};
struct Advection {// recorder for next_particle_buffer.add_advection(partition, cellid, dirtag, particle_id_in_block) (particle_buffer.cuh:100)
	mutable int cell[3], dirtag, pid, calls;
	void add_advection(int /*partition*/, const ivec3& cellid, int tag, int particle_id_in_block) const {
		cell[0] = cellid[0], cell[1] = cellid[1], cell[2] = cellid[2];
		dirtag = tag, pid = particle_id_in_block;
		calls++;
	}
};

// the three per-material bodies of calculate_contribution_and_store_particle_data (the signature is the reference's, :519 / :562 / :613)
void body_jfluid(const Buffer particle_buffer, const Buffer next_particle_buffer, int advection_source_blockno, int source_pidib, int src_blockno, int particle_id_in_block, Duration dt, const std::array<float, 9>& A, std::array<float, 9>& contrib, CalculateContributionAndStoreParticleDataIntermediate& data) {
#include "body_jfluid.inc"
}
void body_fc(const Buffer particle_buffer, const Buffer next_particle_buffer, int advection_source_blockno, int source_pidib, int src_blockno, int particle_id_in_block, Duration dt, const std::array<float, 9>& A, std::array<float, 9>& contrib, CalculateContributionAndStoreParticleDataIntermediate& data) {
#include "body_fc.inc"
}
void body_sand(const Buffer particle_buffer, const Buffer next_particle_buffer, int advection_source_blockno, int source_pidib, int src_blockno, int particle_id_in_block, Duration dt, const std::array<float, 9>& A, std::array<float, 9>& contrib, CalculateContributionAndStoreParticleDataIntermediate& data) {
#include "body_sand.inc"
}
void body_nacc(const Buffer particle_buffer, const Buffer next_particle_buffer, int advection_source_blockno, int source_pidib, int src_blockno, int particle_id_in_block, Duration dt, const std::array<float, 9>& A, std::array<float, 9>& contrib, CalculateContributionAndStoreParticleDataIntermediate& data) {
#include "body_nacc.inc"
}

struct Row {// everything recorded for one particle
	float pos_in[3], F_in[9], logjp_in;
	int base[3], arena[3];		   // base_index, global_base_index after the masking (:774-797)
	float vel[3], A[9], pos_out[3];// G16
	float F_out[9], logjp_out, stress[9];// G18: what the body stored and what compute_stress left in contrib
	float contrib[9];					 // G17: after (:850)
	int adv_cell[3], dirtag, narena[3], discarded;
	float local_pos[3];
	float nodes[27][4];// p2gbuffer[ch][narena + (i, j, k)], (i, j, k) row-major
};

// One particle through the body of g2p2g (:746-905), the cut-out blocks in the kernel's order and scope
void particle(int material, const float (&g2pbuffer)[3][8][8][8], const Buffer& particle_buffer_in, Duration dt, Duration new_dt, Row& r) {
	float p2gbuffer[4][8][8][8];
	memset(p2gbuffer, 0, sizeof(p2gbuffer));
	// bins: one source bin (slot 5 holds the particle), one destination bin
	static float bins_src[1][13][config::G_BIN_CAPACITY], bins_dst[1][13][config::G_BIN_CAPACITY];
	memset(bins_src, 0, sizeof(bins_src));
	memset(bins_dst, 0, sizeof(bins_dst));
	const int zero_offsets[1] = {0};
	Buffer particle_buffer		= particle_buffer_in;
	particle_buffer.bins		= bins_src;
	particle_buffer.bin_offsets = zero_offsets;
	Buffer next_bins			= particle_buffer_in;
	next_bins.bins				= bins_dst;
	next_bins.bin_offsets		= zero_offsets;
	const int source_pidib = 5, advection_source_blockno = 0, src_blockno = 0, particle_id_in_block = 9;
	for(int d = 0; d < 3; ++d) bins_src[0][d][source_pidib] = r.pos_in[d];
	for(int d = 0; d < 9; ++d) bins_src[0][3 + d][source_pidib] = r.F_in[d];
	bins_src[0][12][source_pidib] = r.logjp_in;
	// (:770-772: the kernel fetches pos through fetch_particle_buffer_data, three channel reads)
	vec3 pos {bins_src[0][0][source_pidib], bins_src[0][1][source_pidib], bins_src[0][2][source_pidib]};
	float J = material == 0 ? bins_src[0][3][source_pidib] : 0.f;// (fetch_particle_buffer_data<J_FLUID>: channel 3 is J; for the solid models it is F[0], not used as J)
	r.discarded = 1;
#include "gather.inc"
	for(int d = 0; d < 3; ++d) {
		r.base[d]	 = base_index[d];
		r.arena[d]	 = global_base_index[d];
		r.vel[d]	 = vel[d];
		r.pos_out[d] = pos[d];
	}
	for(int d = 0; d < 9; ++d) r.A[d] = A[d];
	// (:840-846)
	CalculateContributionAndStoreParticleDataIntermediate store_particle_buffer_tmp = {};
	store_particle_buffer_tmp.pos													= pos;
	store_particle_buffer_tmp.J														= J;
	vec9 contrib;
	if(material == 0) body_jfluid(particle_buffer, next_bins, advection_source_blockno, source_pidib, src_blockno, particle_id_in_block, dt, A.data_arr(), contrib.data_arr(), store_particle_buffer_tmp);
	if(material == 1) body_fc(particle_buffer, next_bins, advection_source_blockno, source_pidib, src_blockno, particle_id_in_block, dt, A.data_arr(), contrib.data_arr(), store_particle_buffer_tmp);
	if(material == 2) body_sand(particle_buffer, next_bins, advection_source_blockno, source_pidib, src_blockno, particle_id_in_block, dt, A.data_arr(), contrib.data_arr(), store_particle_buffer_tmp);
	if(material == 3) body_nacc(particle_buffer, next_bins, advection_source_blockno, source_pidib, src_blockno, particle_id_in_block, dt, A.data_arr(), contrib.data_arr(), store_particle_buffer_tmp);
	for(int d = 0; d < 9; ++d) {
		r.F_out[d]	= bins_dst[0][3 + d][particle_id_in_block];
		r.stress[d] = contrib[d];
	}
	r.logjp_out = bins_dst[0][12][particle_id_in_block];
	if(bins_dst[0][0][particle_id_in_block] != pos[0] || bins_dst[0][1][particle_id_in_block] != pos[1] || bins_dst[0][2][particle_id_in_block] != pos[2]) {
		fprintf(stderr, "the body did not store the advected position\n");
		exit(2);
	}
	const Advection next_particle_buffer {};
	const int partition = 0;
	r.dirtag			= -1;
	[&]() {// (the statements `return` out of the kernel when the new stencil base leaves the arena: :877-885)
#include "scatter.inc"
		r.discarded = 0;
		for(int d = 0; d < 3; ++d) r.narena[d] = new_global_base_index[d];
		for(int i = 0; i < 3; ++i)
			for(int j = 0; j < 3; ++j)
				for(int k = 0; k < 3; ++k)
					for(int ch = 0; ch < 4; ++ch) r.nodes[9 * i + 3 * j + k][ch] = p2gbuffer[ch][new_global_base_index[0] + i][new_global_base_index[1] + j][new_global_base_index[2] + k];
	}();
	if(next_particle_buffer.calls != 1) {
		fprintf(stderr, "add_advection was not called exactly once\n");
		exit(2);
	}
	for(int d = 0; d < 9; ++d) r.contrib[d] = contrib[d];
	for(int d = 0; d < 3; ++d) {
		r.adv_cell[d]  = next_particle_buffer.cell[d];
		r.local_pos[d] = local_pos[d];
	}
	r.dirtag = next_particle_buffer.dirtag;
	if(r.discarded) {
		for(int d = 0; d < 3; ++d) r.narena[d] = -99;
		memset(r.nodes, 0, sizeof(r.nodes));
		// nothing may have reached the arena
		for(int i = 0; i < 4 * 512; ++i)
			if((&p2gbuffer[0][0][0][0])[i] != 0.f) {
				fprintf(stderr, "a discarded particle wrote to the arena\n");
				exit(2);
			}
	} else {
		// nothing outside the 27 recorded nodes
		double total = 0, inside = 0;
		for(int i = 0; i < 512; ++i) total += p2gbuffer[0][0][0][i];
		for(int n = 0; n < 27; ++n) inside += r.nodes[n][0];
		if(std::abs(total - inside) > 1e-12 * std::abs(total)) {
			fprintf(stderr, "mass outside the recorded stencil\n");
			exit(2);
		}
	}
}

// G19: the cell arithmetic of update_grid_velocity_query_max (:353-388) for one cell
struct GridBlock {// grid.ch(_0, blockno): val_1d(_channel, cell)
	float v[4][1];
	template<typename I>
	float& val_1d(I, int c) {
		return v[I::value][c];
	}
};
void grid_cell(const float (&cell_in)[4], int is_in_bound, Duration dt, float (&vel_out)[3], float& vel_sqr_out) {
	GridBlock grid_block;
	for(int c = 0; c < 4; ++c) grid_block.v[c][0] = cell_in[c];
	const int cell_id_in_block = 0;
#include "gridcell.inc"
	for(int d = 0; d < 3; ++d) vel_out[d] = grid_block.v[1 + d][0];
	vel_sqr_out = vel_sqr;
}

uint64_t g_state = 0x243F6A8885A308D3ull;
inline uint32_t rnd_u32() {
	g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
	return (uint32_t) (g_state >> 33);
}
inline float rnd01() {
	return (float) (rnd_u32() & 0xFFFFFF) / (float) 0x1000000;
}
inline float rnd_sym() {
	return 2.f * rnd01() - 1.f;
}
std::string g_out;
template<typename T>
void dump(const char* name, const std::vector<T>& v) {
	const std::string fn = g_out + "/" + name;
	FILE* f				 = fopen(fn.c_str(), "wb");
	if(!f) {
		perror(fn.c_str());
		exit(1);
	}
	fwrite(v.data(), sizeof(T), v.size(), f);
	fclose(f);
	printf("wrote %s (%zu elements)\n", fn.c_str(), v.size());
}
}// namespace

int main(int argc, char** argv) {
	g_out = argc > 1 ? argv[1] : ".";
	// ---- parameters (particle_buffer.cuh:141-264 defaults, the volume of an 8-ppc particle at this dx)
	const float dx = config::G_DX, vol = dx * dx * dx / config::MODEL_PPC, rho = 1e3f;
	const float E = 5e3f, nu = 0.4f;
	Buffer pb {};
	pb.volume = vol, pb.mass = rho * vol;
	pb.lambda = E * nu / ((1 + nu) * (1 - 2 * nu)), pb.mu = E / (2 * (1 + nu));
	pb.cohesion = 0.f, pb.beta = 1.f, pb.volume_correction = true;
	{
		pb.yield_surface = 0.816496580927726f * 2.f * 0.5f / (3.f - 0.5f);// particle_buffer.cuh:217
	}
	pb.bm = 2.f / 3.f * pb.mu + pb.lambda, pb.xi = 0.8f, pb.msqr = 3.423772074299613f, pb.hardening_on = true;
	pb.bulk = 4e4f, pb.gamma = 7.15f, pb.viscosity = 0.01f;
	const float dtv = 1e-4f, new_dtv = 7.5e-5f;// (dt != new_dt: the two uses must not be confused)
	const float beta_nacc = 0.5f;// (particle_buffer.cuh:243; sand: 1, :213)
	std::vector<float> par = {(float) config::DOMAIN_BITS, vol, pb.mass, pb.mu, pb.lambda, pb.cohesion, pb.beta, pb.yield_surface, pb.volume_correction ? 1.f : 0.f, pb.bm, pb.xi, pb.msqr, pb.hardening_on ? 1.f : 0.f, dtv, new_dtv, beta_nacc, E, nu, rho, pb.bulk, pb.gamma, pb.viscosity};
	dump("g16_params.f32", par);
	// ---- five velocity arenas (g2pbuffer[3][8][8][8]): rigid translation, shear + noise, fast divergent flow, violent noise, a 90 m/s stream
	const int NA = 5;
	static float arenas[NA][3][8][8][8];
	for(int a = 0; a < NA; ++a)
		for(int x = 0; x < 8; ++x)
			for(int y = 0; y < 8; ++y)
				for(int z = 0; z < 8; ++z) {
					float v[3];
					if(a == 0) v[0] = 0.3f, v[1] = -1.1f, v[2] = 0.05f;
					if(a == 1) v[0] = 0.8f * (float) y * dx * 40.f + 0.02f * rnd_sym(), v[1] = -0.5f + 0.02f * rnd_sym(), v[2] = 0.3f * (float) x * dx * 40.f + 0.02f * rnd_sym();
					if(a == 2) v[0] = 6.f * ((float) x - 3.5f) * dx * 30.f, v[1] = 5.f * ((float) y - 3.5f) * dx * 30.f - 2.f, v[2] = -7.f * ((float) z - 3.5f) * dx * 30.f;
					if(a == 3) v[0] = 70.f * rnd_sym(), v[1] = 70.f * rnd_sym(), v[2] = 70.f * rnd_sym();// (a step far beyond the CFL limit: particles jump a cell, some leave the arena)
					if(a == 4) v[0] = 90.f, v[1] = -3.f, v[2] = 0.f;// (2.3 cells per step in x: about half of the particles leave the arena and are discarded, :877-885)
					for(int c = 0; c < 3; ++c) arenas[a][c][x][y][z] = v[c];
				}
	{
		std::vector<float> av((size_t) NA * 3 * 512);
		memcpy(av.data(), arenas, sizeof(arenas));
		dump("g16_arenas.f32", av);
	}
	// ---- particles: rows of 4 materials (0 = J-fluid) x NA arenas x 48
	const int PER = 48;
	std::vector<float> in, out;
	std::vector<int> iout;
	int discarded = 0, crossed = 0;
	for(int material = 0; material <= 3; ++material)
		for(int a = 0; a < NA; ++a)
			for(int i = 0; i < PER; ++i) {
				Row r {};
				// a particle of the block whose first cell is (64, 32, 96) + 4 (block coordinate 16, 8, 24): positions over the whole block, some on cell faces
				const int bx = 64, by = 32, bz = 96;
				float cellpos[3] = {4.f * rnd01(), 4.f * rnd01(), 4.f * rnd01()};
				if(i % 6 == 0) cellpos[i % 3] = (float) (i % 4) + (i % 12 == 0 ? 0.5f : 0.4999f);// on / next to a rounding boundary of the stencil base
				if(i % 6 == 1) cellpos[(i / 6) % 3] = (i % 12 == 1) ? 0.02f : 3.98f;				 // next to a block face: crosses into the neighbour block
				r.pos_in[0] = ((float) bx + cellpos[0]) * dx;
				r.pos_in[1] = ((float) by + cellpos[1]) * dx;
				r.pos_in[2] = ((float) bz + cellpos[2]) * dx;
				// "plain" rows: F = I and the model's initial log Jp - the state a particle has right after initial_setup, i.e. rows the GPU
				// suite can replay through the public C ABI (one-particle scenes through the real kernel: tests/test_parity_gpu.py)
				const bool plain = (i % 4 == 0) || (i % 12 == 1) || (i % 12 == 7);
				const float amp	 = plain ? 0.f : (i % 4 == 1 ? 0.02f : (i % 4 == 2 ? 0.15f : 0.4f));
				for(int d = 0; d < 9; ++d) r.F_in[d] = ((d & 3) == 0 ? 1.f : 0.f) + amp * rnd_sym();
				if(i % 16 == 15 && !plain) {// compressed: sand inside the cone / NACC hardening
					for(int d = 0; d < 9; ++d) r.F_in[d] = ((d & 3) == 0 ? 0.85f : 0.f) + 0.03f * rnd_sym();
				}
				if(material == 0) {// the J-fluid's state is J (channel 3 = the slot of F[0]); a plain row carries J = 1
					for(int d = 1; d < 9; ++d) r.F_in[d] = 0.f;
					r.F_in[0] = plain ? 1.f : (i % 16 == 15 ? 0.11f : 0.6f + 0.8f * rnd01());
				}
				r.logjp_in = material == 3 ? -0.01f + 0.02f * rnd_sym() : 0.01f * rnd_sym();
				if(plain) r.logjp_in = material == 3 ? -0.01f : 0.f;// LOG_JP_0 (particle_buffer.cuh:210, :242)
				if(material == 0) r.logjp_in = 0.f;
				Buffer pbm = pb;
				if(material == 3) pbm.beta = beta_nacc;
				particle(material, arenas[a], pbm, Duration(dtv), Duration(new_dtv), r);
				discarded += r.discarded;
				crossed += r.dirtag != 13;
				in.push_back((float) material);
				in.push_back((float) a);
				for(float v: r.pos_in) in.push_back(v);
				for(float v: r.F_in) in.push_back(v);
				in.push_back(r.logjp_in);// 15 per row
				for(float v: r.vel) out.push_back(v);
				for(float v: r.A) out.push_back(v);
				for(float v: r.pos_out) out.push_back(v);// 15
				for(float v: r.F_out) out.push_back(v);
				out.push_back(r.logjp_out);
				for(float v: r.stress) out.push_back(v);// 34
				for(float v: r.contrib) out.push_back(v);
				for(float v: r.local_pos) out.push_back(v);// 46
				for(int n = 0; n < 27; ++n)
					for(int ch = 0; ch < 4; ++ch) out.push_back(r.nodes[n][ch]);// 154
				for(int v: r.base) iout.push_back(v);
				for(int v: r.arena) iout.push_back(v);
				for(int v: r.adv_cell) iout.push_back(v);
				iout.push_back(r.dirtag);
				for(int v: r.narena) iout.push_back(v);
				iout.push_back(r.discarded);// 14
			}
	dump("g16_particle_in.f32", in);
	dump("g16_particle_out.f32", out);
	dump("g16_particle_out.i32", iout);
	printf("%d rows, %d crossed into a neighbouring block, %d discarded\n", (int) (in.size() / 15), crossed, discarded);
	// ---- G19: grid cells: rows (mass, mvx, mvy, mvz, is_in_bound) -> (vx, vy, vz, vel_sqr)
	{
		std::vector<float> gin, gout;
		for(int i = 0; i < 512; ++i) {
			float cell[4] = {i % 9 == 0 ? 0.f : (i % 9 == 1 ? -1e-9f : pb.mass * (0.05f + 8.f * rnd01())), 0.f, 0.f, 0.f};
			for(int d = 0; d < 3; ++d) cell[1 + d] = cell[0] * (i % 7 == 3 ? 40.f : 2.f) * rnd_sym() + (i % 9 < 2 ? 1e-6f * rnd_sym() : 0.f);
			if(i % 64 == 61) cell[2] = std::numeric_limits<float>::quiet_NaN();// (is_in_bound 5: y is free)
			if(i % 64 == 26) cell[1] = std::numeric_limits<float>::infinity();// (is_in_bound 2: x is free)
			const int bound = i % 8;
			float vel[3], vs;
			grid_cell(cell, bound, Duration(dtv), vel, vs);
			for(float v: cell) gin.push_back(v);
			gin.push_back((float) bound);
			for(float v: vel) gout.push_back(v);
			gout.push_back(vs);
		}
		dump("g19_gridcell_in.f32", gin);
		dump("g19_gridcell_out.f32", gout);
	}
	return 0;
}
