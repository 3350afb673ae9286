// Golden-vector generator (runs ONLY in the build container, where /root/reference is mounted).
//
// It calls the reference's own pure-arithmetic functions, compiled as host C++ through
// cuda_host_shim.h, on deterministic pseudo-random inputs and records inputs + outputs as raw
// little-endian float32/int32 arrays under tests/golden/.  The recorded files are data (inputs and
// the reference's outputs); no reference source is copied.  See gen_golden.sh for the build line.
//
// Reference functions exercised (file:line under /root/reference):
//   G1 bspline_weight                     Projects/GMPM/utility_funcs.hpp:10-19
//   G2 get_block_id/dir_offset/dir_components  Projects/GMPM/utility_funcs.hpp:21-32
//   G3 math::svd (3x3)                    Library/MnBase/Math/Matrix/svd.cuh:27-1123
//   G4 compute_stress<FIXED_COROTATED>    Projects/GMPM/constitutive_models.cuh:36-73
//   G5 compute_stress<SAND>               Projects/GMPM/constitutive_models.cuh:238-335
//   G6 compute_stress<NACC>               Projects/GMPM/constitutive_models.cuh:77-234
//   G8 compute_dt                         Projects/GMPM/utility_funcs.hpp:36-49
//   G9 matrix_*_3d helpers                Library/MnBase/Math/Matrix/MatrixUtils.h:29-41,147-157,257-286
//   G7 the J-fluid update of calculate_contribution_and_store_particle_data<J_FLUID>   Projects/GMPM/mgmpm_kernels.cuh:473-504
//      (its header cannot be host-compiled - it holds every CUDA kernel of the project -: gen_golden.sh cuts the statements out as text
//       into jfluid_block.inc, which is compiled below inside a function whose locals carry the names the text uses)
#include "cuda_host_shim.h"

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "constitutive_models.cuh"
#include "utility_funcs.hpp"

// G7: the reference's own statements (jfluid_block.inc, cut out by gen_golden.sh) between locals of the names they use
struct JfluidData {
	float J;
};
struct JfluidParams {
	float volume, bulk, gamma, viscosity;
};
static void jfluid_reference_block(JfluidData& data, const JfluidParams& particle_buffer, mn::Duration dt, const std::array<float, 9>& A, std::array<float, 9>& contrib) {
	using namespace mn;
#include "jfluid_block.inc"
}

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
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

// random rotation from a random unit quaternion (column-major 3x3)
static void rand_rot(float* R) {
	float q[4];
	float n = 0;
	for(int i = 0; i < 4; ++i) {
		q[i] = rnd_sym();
		n += q[i] * q[i];
	}
	n = 1.f / std::sqrt(n + 1e-12f);
	for(int i = 0; i < 4; ++i) q[i] *= n;
	float w = q[0], x = q[1], y = q[2], z = q[3];
	R[0] = 1 - 2 * (y * y + z * z);
	R[1] = 2 * (x * y + w * z);
	R[2] = 2 * (x * z - w * y);
	R[3] = 2 * (x * y - w * z);
	R[4] = 1 - 2 * (x * x + z * z);
	R[5] = 2 * (y * z + w * x);
	R[6] = 2 * (x * z + w * y);
	R[7] = 2 * (y * z - w * x);
	R[8] = 1 - 2 * (x * x + y * y);
}
static void mul3(const float* a, const float* b, float* c) {
	for(int j = 0; j < 3; ++j)
		for(int i = 0; i < 3; ++i) c[3 * j + i] = a[i] * b[3 * j] + a[3 + i] * b[3 * j + 1] + a[6 + i] * b[3 * j + 2];
}
// F = U diag(s) V^T with chosen singular values
static void compose(const float* s, float* F) {
	float U[9], V[9], US[9], Vt[9];
	rand_rot(U);
	rand_rot(V);
	for(int j = 0; j < 3; ++j)
		for(int i = 0; i < 3; ++i) {
			US[3 * j + i] = U[3 * j + i] * s[j];
			Vt[3 * j + i] = V[3 * i + j];
		}
	mul3(US, Vt, F);
}

// deformation-gradient test set: classes cycle with the index
static void make_F(int idx, float* F) {
	int cls = idx % 8;
	float s[3];
	switch(cls) {
		case 0:// near identity (what an elastic run sees)
			for(int i = 0; i < 9; ++i) F[i] = ((i & 3) == 0 ? 1.f : 0.f) + 0.05f * rnd_sym();
			break;
		case 1:// moderate strain
			for(int i = 0; i < 9; ++i) F[i] = ((i & 3) == 0 ? 1.f : 0.f) + 0.4f * rnd_sym();
			break;
		case 2:// pure rotation times stretch close to 1
			for(int i = 0; i < 3; ++i) s[i] = 1.f + 0.02f * rnd_sym();
			compose(s, F);
			break;
		case 3:// compression (sand inside cone / NACC case 1)
			for(int i = 0; i < 3; ++i) s[i] = 0.6f + 0.35f * rnd01();
			compose(s, F);
			break;
		case 4:// expansion (sand cone tip / NACC case 2)
			for(int i = 0; i < 3; ++i) s[i] = 1.02f + 0.5f * rnd01();
			compose(s, F);
			break;
		case 5:// shear dominated, volume roughly preserved (sand cone surface)
			s[0] = 1.f + 0.5f * rnd01();
			s[1] = 1.f;
			s[2] = 1.f / s[0] * (0.97f + 0.02f * rnd01());
			compose(s, F);
			break;
		case 6:// near singular
			s[0] = 1.f + rnd01();
			s[1] = 0.5f + 0.5f * rnd01();
			s[2] = 1e-3f * rnd01();
			compose(s, F);
			break;
		default:// reflection (det < 0)
			s[0] = 0.8f + 0.4f * rnd01();
			s[1] = 0.8f + 0.4f * rnd01();
			s[2] = -(0.8f + 0.4f * rnd01());
			compose(s, F);
			break;
	}
	if(idx == 0)
		for(int i = 0; i < 9; ++i) F[i] = ((i & 3) == 0 ? 1.f : 0.f);
	if(idx == 8) {// diagonal
		for(int i = 0; i < 9; ++i) F[i] = 0.f;
		F[0] = 2.f;
		F[4] = 0.5f;
		F[8] = 1.25f;
	}
}

int main(int argc, char** argv) {
	g_out = argc > 1 ? argv[1] : ".";
	using namespace mn;

	// ---- G1 bspline_weight (DOMAIN_BITS = 8 compiled into the reference: dx_inv = 256)
	{
		const int n = 1024;
		std::vector<float> in(n), out(3 * n);
		for(int i = 0; i < n; ++i) {
			// local_pos range in g2p2g is [0.5dx, 1.5dx]; sample a bit wider
			float p = (0.4f + 1.2f * (float) i / (float) (n - 1)) * config::G_DX;
			in[i]	= p;
			vec3 w	= bspline_weight(p);
			out[3 * i + 0] = w[0];
			out[3 * i + 1] = w[1];
			out[3 * i + 2] = w[2];
		}
		dump("g1_bspline_in.f32", in);
		dump("g1_bspline_out.f32", out);
	}
	// ---- G2 index helpers
	{
		std::vector<int> dirs;
		for(int dx = -1; dx <= 1; ++dx)
			for(int dy = -1; dy <= 1; ++dy)
				for(int dz = -1; dz <= 1; ++dz) {
					int tag = dir_offset({dx, dy, dz});
					std::array<int, 3> back {};
					dir_components(tag, back);
					dirs.push_back(dx);
					dirs.push_back(dy);
					dirs.push_back(dz);
					dirs.push_back(tag);
					dirs.push_back(back[0]);
					dirs.push_back(back[1]);
					dirs.push_back(back[2]);
				}
		dump("g2_dirs.i32", dirs);
		const int n = 2048;
		std::vector<float> pos(3 * n);
		std::vector<int> ids(3 * n);
		for(int i = 0; i < n; ++i) {
			for(int d = 0; d < 3; ++d) {
				float x = 0.02f + 0.96f * rnd01();
				if(i % 16 == 0) x = ((float) (rnd_u32() % 254 + 1) + 0.5f) * config::G_DX;// exact half-way cases
				pos[3 * i + d] = x;
			}
			ivec3 id = get_block_id({pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]});
			for(int d = 0; d < 3; ++d) ids[3 * i + d] = id[d];
		}
		dump("g2_node_in.f32", pos);
		dump("g2_node_out.i32", ids);
	}
	// ---- G3 svd
	const int NF = 2048;
	std::vector<float> Fs(9 * NF);
	for(int i = 0; i < NF; ++i) make_F(i, &Fs[9 * i]);
	{
		std::vector<float> out(21 * NF);
		for(int i = 0; i < NF; ++i) {
			const float* F = &Fs[9 * i];
			std::array<float, 9> U {}, V {};
			std::array<float, 3> S {};
			math::svd(F[0], F[3], F[6], F[1], F[4], F[7], F[2], F[5], F[8], U[0], U[3], U[6], U[1], U[4], U[7], U[2], U[5], U[8], S[0], S[1], S[2], V[0], V[3], V[6], V[1], V[4], V[7], V[2], V[5], V[8]);
			for(int k = 0; k < 9; ++k) out[21 * i + k] = U[k];
			for(int k = 0; k < 3; ++k) out[21 * i + 9 + k] = S[k];
			for(int k = 0; k < 9; ++k) out[21 * i + 12 + k] = V[k];
		}
		dump("g3_F_in.f32", Fs);
		dump("g3_svd_out.f32", out);
	}
	// material constants of the reference defaults (settings.h:81-83, particle_buffer.cuh)
	const float E = config::YOUNGS_MODULUS, nu = config::POISSON_RATIO;
	const float lambda = E * nu / ((1 + nu) * (1 - 2 * nu));
	const float mu	   = E / (2 * (1 + nu));
	const float vol	   = 1.f / 256.f / 256.f / 256.f / 8.f;
	{
		std::vector<float> par {vol, mu, lambda};
		dump("g456_params.f32", par);
	}
	// ---- G4 fixed corotated
	{
		std::vector<float> out(9 * NF);
		for(int i = 0; i < NF; ++i) {
			std::array<float, 9> F {}, PF {};
			for(int k = 0; k < 9; ++k) F[k] = Fs[9 * i + k];
			ComputeStressIntermediate<float> d = {};
			compute_stress<float, MaterialE::FIXED_COROTATED>(vol, mu, lambda, F, PF, d);
			for(int k = 0; k < 9; ++k) out[9 * i + k] = PF[k];
		}
		dump("g4_fc_out.f32", out);
	}
	// ---- G5 sand: out = F'(9) PF(9) logjp'(1); input logjp cycles
	{
		std::vector<float> ljp(NF), out(19 * NF);
		const float ys = 0.816496580927726f * 2.f * 0.5f / (3.f - 0.5f);
		for(int i = 0; i < NF; ++i) {
			std::array<float, 9> F {}, PF {};
			for(int k = 0; k < 9; ++k) F[k] = Fs[9 * i + k];
			ComputeStressIntermediate<float> d = {};
			d.cohesion						   = (i % 5 == 4) ? 0.01f : 0.f;
			d.beta							   = 1.f;
			d.yield_surface					   = ys;
			d.volume_correction				   = (i % 7 != 6);
			ljp[i]							   = (i % 3 == 0) ? 0.f : 0.05f * rnd_sym();
			d.log_jp						   = ljp[i];
			compute_stress<float, MaterialE::SAND>(vol, mu, lambda, F, PF, d);
			for(int k = 0; k < 9; ++k) out[19 * i + k] = F[k];
			for(int k = 0; k < 9; ++k) out[19 * i + 9 + k] = PF[k];
			out[19 * i + 18] = d.log_jp;
		}
		dump("g5_sand_logjp_in.f32", ljp);
		dump("g5_sand_out.f32", out);
	}
	// ---- G6 NACC
	{
		std::vector<float> ljp(NF), out(19 * NF);
		const float bm = 2.f / 3.f * mu + lambda;
		for(int i = 0; i < NF; ++i) {
			std::array<float, 9> F {}, PF {};
			for(int k = 0; k < 9; ++k) F[k] = Fs[9 * i + k];
			ComputeStressIntermediate<float> d = {};
			d.bm							   = bm;
			d.xi							   = 0.8f;
			d.beta							   = 0.5f;
			d.msqr							   = 3.423772074299613f;
			d.hardening_on					   = (i % 7 != 6);
			ljp[i]							   = -0.01f + 0.02f * rnd_sym();
			d.log_jp						   = ljp[i];
			compute_stress<float, MaterialE::NACC>(vol, mu, lambda, F, PF, d);
			for(int k = 0; k < 9; ++k) out[19 * i + k] = F[k];
			for(int k = 0; k < 9; ++k) out[19 * i + 9 + k] = PF[k];
			out[19 * i + 18] = d.log_jp;
		}
		dump("g6_nacc_logjp_in.f32", ljp);
		dump("g6_nacc_out.f32", out);
	}
	// ---- G7 J-fluid block: rows (J, A[9]) -> (J', contrib[9]) with the reference's default fluid parameters and dt = 1e-4
	{
		const int n = 512;
		std::vector<float> in(10 * n), out(10 * n), par;
		const JfluidParams pb {vol, 4e4f, 7.15f, 0.01f};// particle_buffer.cuh:148-153 (volume: the generator's, as for G4-G6)
		const float dtv = 1e-4f;
		par = {pb.volume, pb.bulk, pb.gamma, pb.viscosity, dtv, config::G_D_INV};
		for(int i = 0; i < n; ++i) {
			JfluidData d {i < 16 ? 0.1f + 0.01f * (float) i : 0.6f + 0.8f * rnd01()};
			std::array<float, 9> A {}, c {};
			for(int k = 0; k < 9; ++k) A[k] = (i < 16 ? -3e-3f : 1e-3f) * rnd_sym() * (i % 5 == 0 ? 10.f : 1.f);
			if(i < 8) A[0] = A[4] = A[8] = -1e-2f;// drives J below the 0.1 clamp
			in[10 * i] = d.J;
			for(int k = 0; k < 9; ++k) in[10 * i + 1 + k] = A[k];
			jfluid_reference_block(d, pb, Duration(dtv), A, c);
			out[10 * i] = d.J;
			for(int k = 0; k < 9; ++k) out[10 * i + 1 + k] = c[k];
		}
		dump("g7_jfluid_params.f32", par);
		dump("g7_jfluid_in.f32", in);
		dump("g7_jfluid_out.f32", out);
	}
	// ---- G8 compute_dt: rows (max_vel, cur, next, dt_default) -> dt
	{
		std::vector<float> in, out;
		const float mv[]  = {0.f, 1e-3f, 0.5f, 1.f, 3.7f, 25.f, 400.f};
		const float dtd[] = {1e-4f, 5e-6f, 1e-3f};
		const float cur[] = {0.f, 0.01f, 0.0416f};
		for(float a: mv)
			for(float b: dtd)
				for(float c: cur) {
					const float next = 1.f / 24.f;
					Duration r		 = compute_dt(a, Duration(c), Duration(next), Duration(b));
					in.push_back(a);
					in.push_back(c);
					in.push_back(next);
					in.push_back(b);
					out.push_back(r.count());
				}
		dump("g8_dt_in.f32", in);
		dump("g8_dt_out.f32", out);
	}
	// ---- G9 matrix helpers on the first 256 F (pairs i, i+1)
	{
		const int n = 256;
		std::vector<float> out(36 * n);
		for(int i = 0; i < n; ++i) {
			std::array<float, 9> a {}, b {}, c {}, d {}, e {}, f {};
			std::array<float, 3> s {};
			for(int k = 0; k < 9; ++k) {
				a[k] = Fs[9 * i + k];
				b[k] = Fs[9 * (i + 1) + k];
			}
			s = {a[0], b[4], a[8]};
			matrix_matrix_multiplication_3d(a, b, c);
			matmul_mat_diag_mat_t_3d(d, a, s, b);
			matrix_matrix_tranpose_multiplication_3d(a, e);
			matrix_deviatoric_3d(e, f);
			for(int k = 0; k < 9; ++k) {
				out[36 * i + k]		 = c[k];
				out[36 * i + 9 + k]	 = d[k];
				out[36 * i + 18 + k] = e[k];
				out[36 * i + 27 + k] = f[k];
			}
		}
		dump("g9_mat_out.f32", out);
	}
	return 0;
}
