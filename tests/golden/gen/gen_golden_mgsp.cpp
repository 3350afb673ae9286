// Golden-vector generator for the MGSP project's host-compilable functions (runs ONLY in the build container, where
// /root/reference is mounted; see gen_golden_mgsp.sh for the build line and the scratch copy it makes).
//
// Reference functions exercised (file:line under /root/reference):
//   G12 compute_dt (MGSP form: CFL 0.3, 0.51 frame-remainder rule)       Projects/MGSP/utility_funcs.hpp:32-55
//   G13 SignedDistanceGrid::rot_angle_to_matrix                           Projects/MGSP/boundary_condition.cuh:67-91
//   G14 SignedDistanceGrid::detect_and_resolve_collision                  Projects/MGSP/boundary_condition.cuh:164-248
//       (through query_sdf :141-146, get_signed_distance_and_normal :99-140, get_material_velocity :59-65, vec3_cross_vec3 :93-96)
//   G15 SignedDistanceGrid::get_signed_distance_and_normal / query_sdf    Projects/MGSP/boundary_condition.cuh:99-146
// The signed-distance field is filled through the reference's own fill_signed_distance_field (:254-301, its flat-file index
// rule) from a HASH of the node index (exactly reproducible with integer arithmetic: tests/test_oracle_golden.py rebuilds the
// same 256^3 x 4 field in numpy instead of committing 268 MB); the object is allocated with a host malloc allocator (the
// reference passes its own DeviceAllocator, mgsp_benchmark.cuh:38-49,262).
// The recorded files are data (inputs and the reference's outputs); no reference source is copied.
#include "cuda_host_shim.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "utility_funcs.hpp"
#include "boundary_condition.cuh"

using namespace mn;

static uint64_t g_state = 0xD1B54A32D192ED03ull;
static inline uint32_t rnd_u32() {
	g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
	return (uint32_t) (g_state >> 33);
}
static inline float rnd01() { return (float) (rnd_u32() & 0xFFFFFF) / (float) 0x1000000; }
static inline float rnd_sym() { return 2.f * rnd01() - 1.f; }

static std::string g_out;
template<typename T>
static void dump(const char* name, const std::vector<T>& v) {
	std::string fn = g_out + "/" + name;
	FILE* f		   = fopen(fn.c_str(), "wb");
	if(!f) {
		perror(fn.c_str());
		exit(1);
	}
	fwrite(v.data(), sizeof(T), v.size(), f);
	fclose(f);
	printf("wrote %s (%zu elements)\n", fn.c_str(), v.size());
}

struct HostAllocator {
	void* allocate(std::size_t bytes) { return calloc(1, bytes); }
	void deallocate(void* p, std::size_t) { free(p); }
};

// node (i, j, k), channel c -> a float in [-1, 1) with 24 significant bits (every step exact in float32 and in uint32)
static inline float field_value(uint32_t i, uint32_t j, uint32_t k, uint32_t c) {
	uint32_t h = (i * 73856093u) ^ (j * 19349663u) ^ (k * 83492791u) ^ (c * 0x9E3779B1u);
	h *= 2654435761u;
	h ^= h >> 15;
	h *= 2246822519u;
	h ^= h >> 13;
	return (float) (h >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
}

int main(int argc, char** argv) {
	g_out = argc > 1 ? argv[1] : ".";
	// ---- G12 compute_dt (MGSP): rows (max_vel, cur, next, dt_default) -> dt
	{
		std::vector<float> in, out;
		auto row = [&](float mv, float cur, float next, float dtd) {
			in.insert(in.end(), {mv, cur, next, dtd});
			out.push_back(compute_dt(mv, cur, next, dtd));
		};
		const float spf = 1.f / 48.f;
		row(0.f, 0.f, spf, 1e-4f);
		row(0.f, 0.02f, spf, 1e-4f);
		row(1.f, spf, spf, 1e-4f);		  // frame complete
		row(1.f, 0.03f, spf, 1e-4f);	  // next < cur
		row(100.f, 0.f, spf, 1e-4f);	  // CFL-limited
		row(5.f, spf - 1.5e-4f, spf, 1e-4f);// 0.51 rule
		row(5.f, spf - 0.5e-4f, spf, 1e-4f);// lands on the frame end
		for(int i = 0; i < 400; ++i) {
			const float mv	= (i % 7 == 0) ? 0.f : 200.f * rnd01() * rnd01();
			const float nx	= (i % 3 == 0) ? 1.f / 24.f : ((i % 3 == 1) ? 1.f / 48.f : 1.f / 240.f);
			const float cur = (i % 11 == 0) ? nx * (1.f + 0.1f * rnd01()) : nx * rnd01();
			const float dtd = (i % 2) ? 1e-4f : 3e-5f + 4e-4f * rnd01();
			row(mv, cur, nx, dtd);
		}
		// the last substeps of a frame, walked the way main_loop does (mgsp_benchmark.cuh:375-418)
		for(int k = 0; k < 6; ++k) {
			float cur = 0.f, dt = compute_dt(0.f, 0.f, spf, 1e-4f + 1e-5f * k);
			for(int s = 0; s < 400 && dt > 0.f; ++s) {
				cur += dt;
				const float mv = 3.f + 0.5f * k;
				row(mv, cur, spf, 1e-4f + 1e-5f * k);
				dt = out.back();
			}
		}
		dump("g12_mgsp_dt_in.f32", in);
		dump("g12_mgsp_dt_out.f32", out);
	}
	// ---- G13 rot_angle_to_matrix: (angle, dim) -> 9 floats in the storage order of vec3x3
	{
		std::vector<float> in, out;
		for(int dim = 0; dim < 3; ++dim)
			for(int i = 0; i < 24; ++i) {
				const float a = (i == 0) ? 0.f : 6.5f * rnd_sym();
				in.push_back(a);
				in.push_back((float) dim);
				vec3x3 r = SignedDistanceGrid::rot_angle_to_matrix(a, dim);
				for(int e = 0; e < 9; ++e) out.push_back(r.data_arr()[e]);
			}
		dump("g13_rot_in.f32", in);
		dump("g13_rot_out.f32", out);
	}
	// ---- the collision object with the hashed field
	const int N = config::G_DOMAIN_SIZE;
	SignedDistanceGrid obj {HostAllocator {}};
	{
		std::vector<float> flat((size_t) N * N * N);
		auto fill = [&](uint32_t c) {
			for(int i = 0; i < N; ++i)
				for(int j = 0; j < N; ++j)
					for(int k = 0; k < N; ++k) flat[((size_t) i * N + j) * N + k] = field_value(i, j, k, c);
		};
		fill(0);
		fill_signed_distance_field(_0, flat, obj.self());
		fill(1);
		fill_signed_distance_field(_1, flat, obj.self());
		fill(2);
		fill_signed_distance_field(_2, flat, obj.self());
		fill(3);
		fill_signed_distance_field(_3, flat, obj.self());
	}
	// ---- G15 get_signed_distance_and_normal / query_sdf at points x: (in the wall-free zone?, hit?, sdis, normal)
	{
		std::vector<float> in, out;
		const float lo = config::G_BOUNDARY_CONDITION * config::G_DX * config::G_BLOCKSIZE;
		const float hi = (GridDomain::range(_0) - config::G_BOUNDARY_CONDITION) * config::G_BLOCKSIZE * config::G_DX;
		for(int p = 0; p < 768; ++p) {
			vec3 x;
			for(int d = 0; d < 3; ++d) {
				if(p % 8 == 7)
					x[d] = rnd01();// anywhere in the unit box: most of these fall into the wall zone of some axis
				else
					x[d] = lo + (hi - lo) * rnd01() * 0.999f;
			}
			if(p % 16 == 3) x[p % 3] = lo + (float) (rnd_u32() % 200) * config::G_DX;// exactly on a node plane
			vec3 n;
			n.set(0.f);
			const bool inside = !(x[0] < lo || x[0] >= hi || x[1] < lo || x[1] >= hi || x[2] < lo || x[2] >= hi);
			const bool hit	  = obj.query_sdf(n, x);
			float sd		  = 0.f;
			vec3 n2;
			n2.set(0.f);
			if(inside) sd = obj.get_signed_distance_and_normal(x, n2);
			in.insert(in.end(), {x[0], x[1], x[2]});
			out.insert(out.end(), {inside ? 1.f : 0.f, hit ? 1.f : 0.f, sd, n2[0], n2[1], n2[2]});
		}
		dump("g15_sdf_x.f32", in);
		dump("g15_sdf_out.f32", out);
	}
	// ---- G14 detect_and_resolve_collision: 3 boundary types x friction {0, 0.3} x {object at rest, t = 0 | moving object, t != 0}
	{
		std::vector<float> cfg, vin, vout;
		std::vector<int> nodes;
		const int M = 384;
		for(int p = 0; p < M; ++p) {
			for(int d = 0; d < 3; ++d) nodes.push_back((p % 13 == 5) ? (int) (rnd_u32() % N) : 12 + (int) (rnd_u32() % (N - 24)));
			for(int d = 0; d < 3; ++d) vin.push_back(3.f * rnd_sym());
		}
		for(int type = 0; type < 3; ++type)
			for(int fr = 0; fr < 2; ++fr)
				for(int moving = 0; moving < 2; ++moving) {
					obj.type	 = (BoundaryT) type;
					obj.friction = fr ? 0.3f : 0.f;
					float time	 = 0.f;
					obj.rot_mat.set(0.f);
					obj.rot_mat(0, 0) = obj.rot_mat(1, 1) = obj.rot_mat(2, 2) = 1.f;
					obj.trans.set(0.f);
					obj.trans_vel.set(0.f);
					obj.omega.set(0.f);
					obj.dsdt  = 0.f;
					obj.scale = 1.f;
					if(moving) {
						time  = 0.37f;
						obj.omega[0] = 0.3f, obj.omega[1] = -0.2f, obj.omega[2] = 0.5f;
						obj.trans[0] = 0.01f, obj.trans[1] = -0.02f, obj.trans[2] = 0.015f;
						obj.trans_vel[0] = 0.1f, obj.trans_vel[1] = 0.05f, obj.trans_vel[2] = -0.08f;
						obj.dsdt  = 0.05f;
						obj.scale = 1.1f;
						vec3x3 r  = SignedDistanceGrid::rot_angle_to_matrix(0.4f, 1);// a start orientation that is not the identity
						obj.rot_mat = r;
					}
					cfg.insert(cfg.end(), {(float) type, obj.friction, obj.scale, obj.dsdt});
					for(int d = 0; d < 3; ++d) cfg.push_back(obj.trans[d]);
					for(int d = 0; d < 3; ++d) cfg.push_back(obj.trans_vel[d]);
					for(int d = 0; d < 3; ++d) cfg.push_back(obj.omega[d]);
					for(int e = 0; e < 9; ++e) cfg.push_back(obj.rot_mat.data_arr()[e]);
					cfg.push_back(time);
					for(int p = 0; p < M; ++p) {
						ivec3 node {nodes[3 * p], nodes[3 * p + 1], nodes[3 * p + 2]};
						ivec3 block_id = node / config::G_BLOCKSIZE;
						ivec3 cell_id  = node - block_id * config::G_BLOCKSIZE;
						vec3 vel {vin[3 * p], vin[3 * p + 1], vin[3 * p + 2]};
						obj.detect_and_resolve_collision(block_id, cell_id, time, vel);
						for(int d = 0; d < 3; ++d) vout.push_back(vel[d]);
					}
				}
		dump("g14_col_cfg.f32", cfg);
		dump("g14_col_nodes.i32", nodes);
		dump("g14_col_vel_in.f32", vin);
		dump("g14_col_vel_out.f32", vout);
	}
	return 0;
}
