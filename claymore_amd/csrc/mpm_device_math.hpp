// mpm_device_math.hpp — register-resident per-particle math for gfx950 (CDNA4).
//
// Device versions of the arithmetic on claymore's G2P2G path.  Everything stays in VGPRs: 3x3 products, the symmetric
// eigen-decomposition that stands in for the reference's SVD and the four constitutive models (no MFMA: the contractions are 3-wide).  The
// algorithms are the reference's (cited per function, paths relative to /root/reference); the code is
// written for the AMD compiler: selects instead of bit masks (v_cndmask), v_rsq_f32 where the algorithm
// tolerates an approximate reciprocal square root, FMA contraction allowed.
#pragma once
#include <hip/hip_runtime.h>

// ---- experiment switches ------------------------------------------------------------------------------------------
// Every compile-time switch that changes what the kernels COMPUTE (timing hacks that leave a part of the physics out, A/B
// variants of a formulation) is honoured only under -DMPM_EXPERIMENT, which no build of the product sets (__graft_entry__.build(),
// tools/build_variant.sh without it); mpm_build_info() reports the switches a library was built with and tests/test_abi.py checks
// that the shipped one has none.  A stray -DMPM_HACK_* without the guard does not compile.
#if !defined(MPM_EXPERIMENT)
#if defined(MPM_NO_FASTSTAY) || defined(MPM_HACK_STALE_INTERIOR) || defined(MPM_HACK_EDGEWIN) || defined(MPM_HACK_NOSHELL) || defined(MPM_HACK_NOSERIAL) || defined(MPM_HACK_NOWB) || defined(MPM_HACK_UNDEF) || defined(MPM_SCALAR_GS) || defined(MPM_GATHER_B96) || \
	defined(MPM_NT_LOADS) || defined(MPM_LDS_PAD) || defined(MPM_G2P2G_NOLOOP) || defined(MPM_G2P2G_STATS) || defined(MPM_PRE_SITES) || defined(MPM_SERIAL_QUEUE) ||        \
	defined(MPM_G2P2G_WAVES) || defined(MPM_G2P2G_WAVES_FLUID) || defined(MPM_QUEUE_ENTRIES) || defined(MPM_HACK_NOSPLIT) || defined(MPM_PAIR_WAVES) || defined(MPM_PAIR_WAVES_FLUID) ||                   \
	defined(MPM_PAIR_SHARED_GATHER) || defined(MPM_PAIR_LATE_FETCH) || defined(MPM_PAIR_LATE_FETCH_FLUID) || defined(MPM_PAIR_LATE_FETCH_NACC) || defined(MPM_PAIR_DUAL) || defined(MPM_GATHER_ASM) || defined(MPM_CHAIN_ASM)
#error "experiment switch without -DMPM_EXPERIMENT: a product library is built with none of them"
#endif
#endif

namespace mpm {

#define MPM_DEV __device__ __forceinline__
#ifndef MPM_MARK
#ifdef MPM_ASM_MARKS
#define MPM_MARK(name) __asm__ volatile("; MPM_MARK " name)
#else
#define MPM_MARK(name)
#endif
#endif

typedef float v2f_ __attribute__((ext_vector_type(2)));// rows 0 and 1 of a 3-vector / matrix column: packed fp32 (v_pk_*_f32)

// Projects/GMPM/utility_funcs.hpp:10-19 — quadratic B-spline weights; d = offset from the base node in cells
MPM_DEV void bspline_weight_cells(float d, float (&w)[3]) {
	const float a = 1.5f - d;
	w[0]		  = 0.5f * a * a;
	const float b = d - 1.0f;
	w[1]		  = 0.75f - b * b;
	const float c = d - 0.5f;
	w[2]		  = 0.5f * c * c;
}

// lround for p >= 0 (round half away from zero, utility_funcs.hpp:21-23) in 4 instructions: v_rndne_f32 rounds ties to
// even, i.e. differs only for p = k + 0.5 with k even, where p - rndne(p) is exactly +0.5.
MPM_DEV int lround_pos(float p) {
	const float r = __builtin_rintf(p);
	return (int) r + ((p - r) == 0.5f ? 1 : 0);
}
MPM_DEV int node_index(float x, float dx_inv) {
	return lround_pos(x * dx_inv);
}

// Library/MnBase/Math/Matrix/MatrixUtils.h:147-157 (column-major).  Rows 0 and 1 of every column go through packed fp32
// (v_pk_mul_f32 / v_pk_fma_f32 with the scalar of b broadcast by op_sel): 3 x (3 packed + 3 scalar) instead of 27 instructions.
MPM_DEV void matmul3(const float (&a)[9], const float (&b)[9], float (&c)[9]) {
	const v2f_ a0 = {a[0], a[1]}, a1 = {a[3], a[4]}, a2 = {a[6], a[7]};
#pragma unroll
	for(int j = 0; j < 3; ++j) {
		const v2f_ xy = a0 * b[3 * j] + a1 * b[3 * j + 1] + a2 * b[3 * j + 2];
		c[3 * j]	  = xy.x;
		c[3 * j + 1]  = xy.y;
		c[3 * j + 2]  = a[2] * b[3 * j] + a[5] * b[3 * j + 1] + a[8] * b[3 * j + 2];
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Particle state of the three solid models: the LEFT CAUCHY-GREEN tensor b = F F^T, not F.
//
// The reference carries the deformation gradient F (9 floats, particle_buffer.cuh:141-264) and every constitutive model on this
// path starts with its SVD F = U S V^T (constitutive_models.cuh:36-335).  All of them are isotropic - elastic energy and yield
// surface depend on the singular values only, the stress is P F^T = U diag(.) U^T, a projected F is U S_new V^T - so V never
// reaches an output: positions, the grid and log Jp depend on F only through b = F F^T = U S^2 U^T, and b obeys its own update
//     F <- (I + dt grad v) F            ==>   b <- G b G^T,  G = I + dt grad v          (:838-850 of mgmpm_kernels.cuh)
//     F <- U S_new V^T  (projection)    ==>   b <- U diag(S_new^2) U^T
// A particle therefore stores the six entries {b00, b11, b22, b10, b20, b21}: 24 B instead of 36 B, read and written once per
// step in a launch whose time follows its bytes, and the per-particle arithmetic loses F F^T, the matrix product of
// the projection and every branch that existed to recover V.  What b cannot carry is the SIGN of det F (a reflected F, which the
// reference's SVD moves into the smallest singular value, svd.cuh:590-770): it is one bit, kept in the sign of b00 (> 0 otherwise)
// and updated exactly - det(G F) = det G det F, and det G <= 0 needs an entry of dt grad v beyond 1/3 in magnitude (Gershgorin), which
// a wave tests with one comparison per step.  The reference retrieves positions only (retrieve_particle_buffer, :1087-1122).
// ---------------------------------------------------------------------------------------------------------------
// b = F F^T of a column-major F (set-up of a model from given deformation gradients, the function-level tests)
MPM_DEV void left_cauchy_green(const float (&F)[9], float (&b)[6]) {
	b[0] = F[0] * F[0] + F[3] * F[3] + F[6] * F[6];
	b[1] = F[1] * F[1] + F[4] * F[4] + F[7] * F[7];
	b[2] = F[2] * F[2] + F[5] * F[5] + F[8] * F[8];
	b[3] = F[1] * F[0] + F[4] * F[3] + F[7] * F[6];
	b[4] = F[2] * F[0] + F[5] * F[3] + F[8] * F[6];
	b[5] = F[2] * F[1] + F[5] * F[4] + F[8] * F[7];
}
MPM_DEV float det3(const float (&F)[9]) {
	return F[0] * (F[4] * F[8] - F[7] * F[5]) + F[3] * (F[7] * F[2] - F[1] * F[8]) + F[6] * (F[1] * F[5] - F[4] * F[2]);
}
// o = G b G^T for a column-major G and a symmetric b: M = G b (rows 0 / 1 of a column packed), then the six entries of M G^T
// (9 packed + 9 scalar, 3 packed + 12 scalar instructions: what F <- G F followed by F F^T cost before)
MPM_DEV void push_forward(const float (&G)[9], const float (&b)[6], float (&o)[6]) {
	const v2f_ g0 = {G[0], G[1]}, g1 = {G[3], G[4]}, g2 = {G[6], G[7]};
	const v2f_ m0 = g0 * b[0] + g1 * b[3] + g2 * b[4];
	const v2f_ m1 = g0 * b[3] + g1 * b[1] + g2 * b[5];
	const v2f_ m2 = g0 * b[4] + g1 * b[5] + g2 * b[2];
	const float z0 = G[2] * b[0] + G[5] * b[3] + G[8] * b[4];
	const float z1 = G[2] * b[3] + G[5] * b[1] + G[8] * b[5];
	const float z2 = G[2] * b[4] + G[5] * b[5] + G[8] * b[2];
	const v2f_ c0 = m0 * G[0] + m1 * G[3] + m2 * G[6];// (00, 10)
	o[0]		  = c0.x;
	o[3]		  = c0.y;
	o[4]		  = z0 * G[0] + z1 * G[3] + z2 * G[6];
	o[1]		  = m0.y * G[1] + m1.y * G[4] + m2.y * G[7];
	o[5]		  = z0 * G[1] + z1 * G[4] + z2 * G[7];
	o[2]		  = z0 * G[2] + z1 * G[5] + z2 * G[8];
}

// v_rcp_f32 (1 ulp).  The reference binary is built with --use_fast_math (CMake-Utils/setup_cuda.cmake:50), i.e. with
// approximate division itself; IEEE division costs ~10 VALU instructions on gfx950 and G2P2G is VALU-bound.
MPM_DEV float rcp_fast(float x) {
	return __builtin_amdgcn_rcpf(x);
}
// Natural log / exp through the 1-ulp hardware base-2 instructions (v_log_f32, v_exp_f32): 2 VALU instructions instead of
// ~20 / ~15 for the IEEE-careful library versions.  The reference binary uses the same kind of approximation
// (--use_fast_math turns logf / expf into lg2.approx / ex2.approx, CMake-Utils/setup_cuda.cmake:50).  Arguments on this
// path are singular values in [1e-4, ~10] and log-strains of order 1: no denormals, no overflow.
MPM_DEV float log_fast(float x) {
	return __builtin_amdgcn_logf(x) * 0.693147180559945f;
}
MPM_DEV float exp_fast(float x) {
	return __builtin_amdgcn_exp2f(x * 1.442695040888963f);
}
MPM_DEV float rsqrt_approx(float x) {
	return __builtin_amdgcn_rsqf(x);// v_rsq_f32, ~1 ulp; the algorithm re-normalises (svd.cuh:210-215)
}
// Hook: the G2P2G kernel threads an unrelated, latency-bound chain of LDS read-modify-write steps (the P2G scatter of
// the previous particle) through the per-particle arithmetic; `hk.at<SITE>()` is called at evenly spaced points that
// every lane reaches (never inside divergent control flow).  NoHook compiles to nothing.
struct NoHook {
	template<int SITE>
	MPM_DEV void at() {}
};

// ---------------------------------------------------------------------------------------------------------------
// Symmetric eigen-decomposition b = U diag(lam) U^T by cyclic Jacobi with EXACT rotations.
//
// Every constitutive model on this path only needs U and the singular values sigma_k = sqrt(lam_k) of F = U S V^T:
//   * P F^T = U diag(P_hat_k sigma_k) U^T                                  (V cancels: V^T V = I)
//   * a projected b_new = (U S_new V^T)(U S_new V^T)^T = U diag(S_new_k^2) U^T
// so the reference's route (math::svd: Jacobi on F^T F for V, B = F V, Givens QR for U and S, svd.cuh:27-1123) is
// replaced by the left-handed one, which skips B, the QR / column normalisation and the sort.  A rotation annihilates
// s_pq exactly (t = tan(theta) from the stable quadratic root, one v_sqrt / v_rcp / v_rsq) instead of the reference's
// approximate Givens angle (svd.cuh:167-252), which costs half the arithmetic per rotation: the updated diagonal is
// s_pp - t s_pq, s_qq + t s_pq and s_pq := 0.  Sweeps stop for the whole wave once every off-diagonal entry of every
// lane is below 1e-6 of the smallest diagonal entry (at most 4 sweeps, the reference's fixed count).
// Hook: see NoHook.
// ---------------------------------------------------------------------------------------------------------------
MPM_DEV void jacobi_rot(float& spp, float& spq, float& sqq, float& srp, float& srq, float (&up)[3], float (&uq)[3]) {
	const float delta = sqq - spp;
	const float o2	  = spq + spq;
	const float h	  = fmaf(o2, o2, fmaf(delta, delta, 1e-36f));// > 0: no 0/0 for an already diagonal pair
	const float r	  = __builtin_amdgcn_sqrtf(h);
	const float den	  = delta + __builtin_copysignf(r, delta);// |den| >= |delta|: the smaller root, |theta| <= pi/4
	const float t	  = o2 * rcp_fast(den);
	const float c	  = rsqrt_approx(fmaf(t, t, 1.0f));
	const float s	  = t * c;
	spp				  = fmaf(-t, spq, spp);
	sqq				  = fmaf(t, spq, sqq);
	spq				  = 0.f;
	const float x = srp, y = srq;
	srp = c * x - s * y;
	srq = s * x + c * y;
	// U <- U P  (P_pp = P_qq = c, P_pq = s, P_qp = -s)
	const v2f_ p = {up[0], up[1]}, q = {uq[0], uq[1]};
	const v2f_ pn = p * c - q * s, qn = p * s + q * c;
	const float pz = up[2], qz = uq[2];
	up[0] = pn.x;
	up[1] = pn.y;
	up[2] = c * pz - s * qz;
	uq[0] = qn.x;
	uq[1] = qn.y;
	uq[2] = s * pz + c * qz;
}

constexpr float kEigTol = 1e-6f;// sweeps stop once every off-diagonal entry of a lane is below this fraction of its smallest diagonal entry
constexpr int kEigSites = 13;
// b: {b00, b11, b22, b10, b20, b21}.
// undeformed (wave-uniform, out): b is the identity for every lane of the wave - diagonal exactly 1, off-diagonals below the
// tolerance.  Such a particle (free fall, rigid translation: the default window of the C3 bench) has zero stress in every model on
// this path, exactly, and no plastic update: the stress functions return early (for lanes that carry no reflection).
template<int BASE, class Hook>
MPM_DEV void sym_eig3(const float (&b)[6], float (&lam)[3], float (&U)[9], Hook& hk, bool& undeformed) {
	float s11 = b[0], s22 = b[1], s33 = b[2], s21 = b[3], s31 = b[4], s32 = b[5];
	float u1[3] = {1.f, 0.f, 0.f}, u2[3] = {0.f, 1.f, 0.f}, u3[3] = {0.f, 0.f, 1.f};
	MPM_MARK("eig_jacobi");
	hk.template at<BASE + 0>();
	// The convergence test runs BEFORE every sweep (the first one included: a particle in free fall or rigid translation
	// has b diagonal already, and a whole wave of them skips the rotations altogether), not after the last one.  `conv` is the lane's
	// own verdict, `done` the wave's: a sweep runs while any lane needs it (scalar branch), but only the lanes that need it rotate
	// (exec mask), so the number of sweeps a particle gets - and with it its result - depends on the particle alone, not on the other
	// 63 lanes of its wave (i.e. not on the sort order, the block numbering or the partitioning of an MGSP run).
	bool done, conv;
#define MPM_CONVERGED()                                                                 \
	{                                                                                   \
		const float off = fmaxf(fmaxf(fabsf(s21), fabsf(s31)), fabsf(s32));             \
		const float dia = fminf(fminf(fabsf(s11), fabsf(s22)), fabsf(s33));             \
		conv			= off <= kEigTol * dia;                                         \
		done			= __all(conv);                                                  \
	}
#define MPM_SWEEP(IT, LAST)                                                    \
	if(!done) {                                                                \
		if(!conv) jacobi_rot(s11, s21, s22, s31, s32, u1, u2);                 \
	}                                                                          \
	hk.template at<BASE + 1 + 3 * IT>();                                       \
	if(!done) {                                                                \
		if(!conv) jacobi_rot(s22, s32, s33, s21, s31, u2, u3);                 \
	}                                                                          \
	hk.template at<BASE + 2 + 3 * IT>();                                       \
	if(!done) {                                                                \
		if(!conv) jacobi_rot(s33, s31, s11, s32, s21, u3, u1);                 \
		if(!(LAST)) MPM_CONVERGED()                                            \
	}                                                                          \
	hk.template at<BASE + 3 + 3 * IT>();
	undeformed = false;
	MPM_CONVERGED()
	if(done) undeformed = __all((s11 == 1.f) & (s22 == 1.f) & (s33 == 1.f));
#if defined(MPM_EXPERIMENT) && defined(MPM_HACK_UNDEF)// timing experiment only: every wave takes the undeformed exit (wrong physics)
	done = conv = undeformed = true;
#endif
	MPM_SWEEP(0, false)
	MPM_SWEEP(1, false)
	MPM_SWEEP(2, false)
	MPM_SWEEP(3, true)
#undef MPM_SWEEP
#undef MPM_CONVERGED
	MPM_MARK("eig_end");
	lam[0] = s11;
	lam[1] = s22;
	lam[2] = s33;
#pragma unroll
	for(int r = 0; r < 3; ++r) {
		U[r]	 = u1[r];
		U[3 + r] = u2[r];
		U[6 + r] = u3[r];
	}
}

// out = U diag(d) U^T as {00, 11, 22, 10, 20, 21}: packed rows 0 / 1, 18 instructions
MPM_DEV void sym_from_eig(const float (&U)[9], const float (&d)[3], float (&out)[6]) {
	const v2f_ u0 = {U[0], U[1]}, u1 = {U[3], U[4]}, u2 = {U[6], U[7]};
	const v2f_ ud0 = u0 * d[0], ud1 = u1 * d[1], ud2 = u2 * d[2];
	const float udz0 = U[2] * d[0], udz1 = U[5] * d[1], udz2 = U[8] * d[2];
	const v2f_ c0 = ud0 * U[0] + ud1 * U[3] + ud2 * U[6];// (00, 10)
	const v2f_ cz = u0 * udz0 + u1 * udz1 + u2 * udz2;	 // (20, 21)
	out[0] = c0.x;
	out[3] = c0.y;
	out[1] = ud0.y * U[1] + ud1.y * U[4] + ud2.y * U[7];
	out[4] = cz.x;
	out[5] = cz.y;
	out[2] = udz0 * U[2] + udz1 * U[5] + udz2 * U[8];
}
// the symmetric 3 x 3 matrix (column-major 9) of its six entries
MPM_DEV void sym_expand(const float (&s)[6], float (&m)[9]) {
	m[0] = s[0], m[1] = s[3], m[2] = s[4];
	m[3] = s[3], m[4] = s[1], m[5] = s[5];
	m[6] = s[4], m[7] = s[5], m[8] = s[2];
}

// The scale the stress leaves a model with, as uniform factors formed on the host (gfx950 has no scalar float ALU: a uniform product
// formed in the kernel costs a vector instruction per iteration or a vector register for the whole loop).  Function-level tests pass
// {2 mu vol, lambda vol, vol}: P F^T vol as the reference returns it; G2P2G passes the same times -new_dt D^-1 dx, so that the P2G
// payload is one fused multiply-add per entry, A m D^-1 dx^2 + stress (:850 of mgmpm_kernels.cuh), instead of two instructions.
struct StressScale {
	float mu2v, lamv, vol;
};

// Material constants passed by value to the kernels (Projects/GMPM/particle_buffer.cuh:141-264)
struct MaterialConst {
	float mass, volume, mu, lambda;
	float bulk, gamma, viscosity;		 // J_FLUID
	float cohesion, beta, yield_surface;// SAND (beta also NACC)
	float bm, xi, msqr;					 // NACC
	float log_jp0;
	int volume_correction, hardening_on;
};
MPM_DEV StressScale stress_scale(const MaterialConst& mc) {
	return StressScale {2.0f * mc.mu * mc.volume, mc.lambda * mc.volume, mc.volume};
}

// compute_stress<FIXED_COROTATED>, Projects/GMPM/constitutive_models.cuh:36-73.
// P F^T = U diag(P_hat_k sigma_k) U^T with P_hat_k sigma_k = 2 mu (sigma_k - 1) sigma_k + lambda (J - 1) J.
// refl: det F < 0 - the smallest singular value carries the sign (svd.cuh:590-770).
constexpr int kFcSites = kEigSites + 1;
template<int BASE, class Hook>
MPM_DEV void stress_fixed_corotated(const StressScale& ss, const float (&b)[6], bool refl, float (&PF)[9], Hook& hk) {
	float lam[3], U[9];
	bool undeformed;
	sym_eig3<BASE>(b, lam, U, hk, undeformed);
	if(undeformed && !__any(refl)) {// sigma_k = 1, J = 1: P F^T = 0 exactly
#pragma unroll
		for(int d = 0; d < 9; ++d) PF[d] = 0.f;
		hk.template at<BASE + kEigSites>();
		return;
	}
	float sig[3];
#pragma unroll
	for(int k = 0; k < 3; ++k) sig[k] = __builtin_amdgcn_sqrtf(fmaxf(lam[k], 0.f));
	if(refl) {
		const bool m0 = lam[0] <= lam[1] && lam[0] <= lam[2], m1 = !m0 && lam[1] <= lam[2];
		sig[0] = m0 ? -sig[0] : sig[0];
		sig[1] = m1 ? -sig[1] : sig[1];
		sig[2] = (!m0 && !m1) ? -sig[2] : sig[2];
	}
	const float J  = sig[0] * sig[1] * sig[2];
	const float vl = ss.lamv * (J - 1.0f) * J;
	const float vm = ss.mu2v;
	float d[3], pf[6];
#pragma unroll
	for(int k = 0; k < 3; ++k) d[k] = fmaf(vm, lam[k] - sig[k], vl);
	sym_from_eig(U, d, pf);
	sym_expand(pf, PF);
	hk.template at<BASE + kEigSites>();
}
MPM_DEV void stress_fixed_corotated(const MaterialConst& mc, const float (&b)[6], bool refl, float (&PF)[9]) {
	NoHook nh;
	stress_fixed_corotated<0>(stress_scale(mc), b, refl, PF, nh);
}

// compute_stress<SAND>, constitutive_models.cuh:238-335 (Drucker-Prager return mapping, StVK-Hencky) in principal
// log-strains ln sigma_k = 0.5 ln lam_k.  The return mapping moves ln sigma_k to ln S_new_k, so the projected
// b = U diag(exp(2 ln S_new_k)) U^T, and P F^T vol = U diag((2 mu ln S_new_k + lambda tr ln S_new) vol) U^T.
// A reflected F (refl) enters through |S| (:262) and leaves with det > 0 (the projection rebuilds U S_new V^T with S_new > 0).
constexpr int kSandSites = kEigSites + 3;
template<int BASE, class Hook>
MPM_DEV void stress_sand(const MaterialConst& mc, const StressScale& ss, float (&b)[6], bool& refl, float& log_jp, float (&PF)[9], Hook& hk) {
	float lam[3], U[9];
	bool undeformed;
	sym_eig3<BASE>(b, lam, U, hk, undeformed);
	// undeformed, no cohesion, log Jp >= 0: ln sigma = 0 sits at the cone tip with zero strain (:282-289): b and log Jp stay, P F^T = 0.
	// (The three parts are skipped in place, around the hook sites, so that every site exists once in the code.)
#if defined(MPM_EXPERIMENT) && defined(MPM_HACK_UNDEF)
	const bool skip = undeformed;
#else
	const bool skip = undeformed && mc.cohesion == 0.f && __all((log_jp >= 0.f) & !refl);
#endif
	const float scaled_mu = 2.0f * mc.mu;
	float lns[3], epsilon[3], epsilon_hat[3], lnS[3];
	float sum_epsilon = 0.f, trace_epsilon = 0.f, epsilon_hat_norm = 0.f;
	if(!skip) {
#pragma unroll
		for(int i = 0; i < 3; i++) {
			lns[i]	   = 0.5f * log_fast(fmaxf(lam[i], 1e-8f));// ln max(|S|, 1e-4) (:262)
			epsilon[i] = lns[i] - mc.cohesion;
		}
		sum_epsilon	  = epsilon[0] + epsilon[1] + epsilon[2];
		trace_epsilon = sum_epsilon + log_jp;
#pragma unroll
		for(int i = 0; i < 3; i++) epsilon_hat[i] = epsilon[i] - (trace_epsilon * (1.0f / 3.0f));
		epsilon_hat_norm = __builtin_amdgcn_sqrtf(epsilon_hat[0] * epsilon_hat[0] + epsilon_hat[1] * epsilon_hat[1] + epsilon_hat[2] * epsilon_hat[2]);
	}
	hk.template at<BASE + kEigSites>();
	if(!skip) {
		// Return mapping without divergent branches (the three cases of :282-316 as selects; a wave of sand particles usually
		// holds all of them): dl = ln S_new - ln sigma is -epsilon at the cone tip (case II), -r epsilon_hat with
		// r = max(delta_gamma, 0) / |epsilon_hat| otherwise (case III; r = 0 is case I, inside the cone).
		const bool tip	= trace_epsilon >= 0.0f;
		const bool dead = !tip && mc.mu == 0.f;// reference: logf(0) when mu == 0 (:298-300): P is NaN, F is left as it is
		const float delta_gamma = epsilon_hat_norm + (3.0f * mc.lambda + scaled_mu) * rcp_fast(scaled_mu) * trace_epsilon * mc.yield_surface;
		const float r			= fmaxf(delta_gamma, 0.f) * rcp_fast(fmaxf(epsilon_hat_norm, 1e-30f));
		bool moved = false;
#pragma unroll
		for(int i = 0; i < 3; i++) {
			const float dl = tip ? -epsilon[i] : -r * epsilon_hat[i];
			moved |= dl != 0.f;
			lnS[i] = dead ? -__builtin_inff() : lns[i] + dl;
		}
		log_jp = tip ? (mc.volume_correction ? mc.beta * sum_epsilon + log_jp : log_jp) : (dead ? log_jp : 0.f);
		// The projected b = U diag(exp(2 ln S_new)) U^T.  While every particle of the wave is inside the cone (dl = 0 exactly: elastic, the
		// state of a column at rest; or at the tip with zero strain: free fall) the trial b stands and the rebuild is skipped for the
		// whole wave; the reference rebuilds U S V^T there too, which only adds its rounding.  A reflected F is rebuilt whether the
		// strain moved or not (the reference's projection removes the reflection).
		const bool changed = !dead && (moved | refl);
		if(__any(changed)) {
			if(changed) {
				float s2[3];
#pragma unroll
				for(int i = 0; i < 3; i++) s2[i] = exp_fast(lnS[i] + lnS[i]);
				sym_from_eig(U, s2, b);
				refl = false;
			}
		}
	}
	hk.template at<BASE + kEigSites + 1>();
	if(!skip) {
		const float trace_log_S = lnS[0] + lnS[1] + lnS[2];
		float d[3], pf[6];
#pragma unroll
		for(int k = 0; k < 3; ++k) d[k] = ss.mu2v * lnS[k] + ss.lamv * trace_log_S;
		sym_from_eig(U, d, pf);
		sym_expand(pf, PF);
	} else {
#pragma unroll
		for(int d = 0; d < 9; ++d) PF[d] = 0.f;
	}
	hk.template at<BASE + kEigSites + 2>();
}
MPM_DEV void stress_sand(const MaterialConst& mc, float (&b)[6], bool& refl, float& log_jp, float (&PF)[9]) {
	NoHook nh;
	stress_sand<0>(mc, stress_scale(mc), b, refl, log_jp, PF, nh);
}

// compute_stress<NACC>, constitutive_models.cuh:77-234 (USE_JOSH_FRACTURE_PAPER branch) in principal stretches:
// B_hat_k = sigma_k^2 = lam_k, the projections return new squared stretches Bn_k, so the projected b = U diag(Bn) U^T and
// dev(b) and P F^T are diagonal in U.  J^(+-2/3), cube roots and logs go through one v_log_f32 / v_exp_f32 each instead of powf
// (the reference binary's --use_fast_math does the same).  refl = det F < 0.
constexpr int kNaccSites = kEigSites + 2;
template<int BASE, class Hook>
MPM_DEV void stress_nacc(const MaterialConst& mc, const StressScale& ss, float (&b)[6], bool& refl, float& log_jp, float (&PF)[9], Hook& hk) {
	float lam[3], U[9];
	bool undeformed;
	sym_eig3<BASE>(b, lam, U, hk, undeformed);
	// undeformed: p_trial = 0 lies strictly inside (p_min, p0) and y < 0 (:113-151): no projection, no hardening; dev(b) = 0, J = 1: P F^T = 0.
	// (The two parts are skipped in place, around the hook site, so that every site exists once in the code.)
	const bool skip = undeformed && mc.bm > 0.f && mc.beta > 0.f && !__any(refl);
	const float bm = mc.bm;
	bool rebuild = false;
	float p0 = 0.f, p_min = 0.f, lg2J = 0.f, trB3 = 0.f, sh0 = 0.f, sh1 = 0.f, sh2 = 0.f, p_trial = 0.f, y_s_half_coeff = 0.f, y_p_half = 0.f, s_sqrnorm = 0.f, y = 0.f;
	float Bn[3] = {lam[0], lam[1], lam[2]};// new squared stretches
	if(!skip) {
		const float ex = exp_fast(mc.xi * fmaxf(-log_jp, 0.f));
		p0 = bm * (0.00001f + 0.5f * (ex - rcp_fast(ex)));// sinh
		p_min = -mc.beta * p0;
		lg2J	 = 0.5f * __builtin_amdgcn_logf(lam[0] * lam[1] * lam[2]);// log2 |Je_trial|
		const float Je_abs	 = __builtin_amdgcn_exp2f(lg2J);
		const float Je_trial = refl ? -Je_abs : Je_abs;
		trB3	 = (lam[0] + lam[1] + lam[2]) * (1.f / 3.f);
		// a reflected F makes the reference's powf(Je_trial, -2/3) NaN (:96); the same happens here through the sign of Je
		const float Jm23mu = refl ? __builtin_nanf("") : mc.mu * __builtin_amdgcn_exp2f(lg2J * (-2.f / 3.f));
		sh0 = Jm23mu * (lam[0] - trB3), sh1 = Jm23mu * (lam[1] - trB3), sh2 = Jm23mu * (lam[2] - trB3);
		p_trial		   = -bm * 0.5f * (Je_trial - rcp_fast(Je_trial)) * Je_trial;
		y_s_half_coeff = 1.5f * (1.f + 2.f * mc.beta);
		y_p_half	   = mc.msqr * (p_trial - p_min) * (p_trial - p0);
		s_sqrnorm	   = sh0 * sh0 + sh1 * sh1 + sh2 * sh2;
		y			   = y_s_half_coeff * s_sqrnorm + y_p_half;
	}
	hk.template at<BASE + kEigSites>();
	if(!skip) {
		if(p_trial > p0 || p_trial < p_min) {// cases 1, 2: project to a tip of the yield surface (:113-143)
			const float Je_new2 = -2.f * (p_trial > p0 ? p0 : p_min) * rcp_fast(bm) + 1.f;// Je_new^2
			const float l2		= __builtin_amdgcn_logf(Je_new2);						// 2 log2 Je_new
			Bn[0] = Bn[1] = Bn[2] = __builtin_amdgcn_exp2f(l2 * (1.f / 3.f));				// Je_new^(2/3)
			rebuild				  = true;
			if(mc.hardening_on) log_jp += (lg2J - 0.5f * l2) * 0.693147180559945f;
		} else if(y >= 1e-4f) {// case 3: project to the yield surface (:151-203)
			const float B_s_coeff = __builtin_amdgcn_exp2f(lg2J * (2.f / 3.f)) * rcp_fast(mc.mu) * __builtin_amdgcn_sqrtf(-y_p_half * rcp_fast(y_s_half_coeff)) * rsqrt_approx(s_sqrnorm);
			Bn[0]				  = sh0 * B_s_coeff + trB3;
			Bn[1]				  = sh1 * B_s_coeff + trB3;
			Bn[2]				  = sh2 * B_s_coeff + trB3;
			rebuild				  = true;
			if(mc.hardening_on && p0 > 1e-4f && p_trial < p0 - 1e-4f && p_trial > 1e-4f + p_min) {
				// (1 - beta is formed HERE, in the branch: hoisted out of G2P2G's particle loop as a loop invariant it holds a vector register across the loop - gfx950 has no
				//  scalar float ALU - which the pair kernel's NACC instantiation does not have: it was spilled and reloaded inside the loop)
				float beta_here = mc.beta;
				__asm__ volatile("" : "+s"(beta_here));
				const float p_center = (1.0f - beta_here) * p0 * 0.5f;
				const float q_trial	 = __builtin_amdgcn_sqrtf(1.5f * s_sqrnorm);
				float d0 = p_center - p_trial, d1 = -q_trial;
				const float dn = rsqrt_approx(d0 * d0 + d1 * d1);
				d0 *= dn;
				d1 *= dn;
				const float C  = mc.msqr * (p_center - p_min) * (p_center - p0);
				const float B  = mc.msqr * d0 * (2.f * p_center - p0 - p_min);
				const float A  = mc.msqr * d0 * d0 + (1.f + 2.f * mc.beta) * d1 * d1;
				const float sq = __builtin_amdgcn_sqrtf(B * B - 4.f * A * C);
				const float ia = rcp_fast(2.f * A);
				const float p1 = p_center + (-B + sq) * ia * d0;
				const float p2 = p_center + (-B - sq) * ia * d0;
				const float p_fake		= (p_trial - p_center) * (p1 - p_center) > 0.f ? p1 : p2;
				const float Je_new_fake2 = fabsf(-2.f * p_fake * rcp_fast(bm) + 1.f);// Je_new_fake^2
				if(Je_new_fake2 > 1e-8f) log_jp += (lg2J - 0.5f * __builtin_amdgcn_logf(Je_new_fake2)) * 0.693147180559945f;
			}
		}
		const bool neg = refl && !rebuild;
		if(rebuild) {// (the reference rebuilds U S_new V^T with a proper rotation V: the reflection is gone)
			sym_from_eig(U, Bn, b);
			refl = false;
		}
		// elasticity (:206-230): J, dev(b) of the renewed F
		const float lg2Jn	   = 0.5f * __builtin_amdgcn_logf(Bn[0] * Bn[1] * Bn[2]);
		const float Jn_abs	   = __builtin_amdgcn_exp2f(lg2Jn);
		const float J2		   = Jn_abs * Jn_abs;
		const float dev_b_coeff = neg ? __builtin_nanf("") : mc.mu * __builtin_amdgcn_exp2f(lg2Jn * (-2.f / 3.f));
		const float i_coeff	   = bm * .5f * ((J2 - 1.f) * 0.5f - (neg ? __builtin_nanf("") : lg2Jn * 0.693147180559945f));
		const float trBn3	   = (Bn[0] + Bn[1] + Bn[2]) * (1.f / 3.f);
		float d[3], pf[6];
	#pragma unroll
		for(int k = 0; k < 3; ++k) d[k] = (dev_b_coeff * (Bn[k] - trBn3) + i_coeff) * ss.vol;
		sym_from_eig(U, d, pf);
		sym_expand(pf, PF);
	} else {
#pragma unroll
		for(int d = 0; d < 9; ++d) PF[d] = 0.f;
	}
	hk.template at<BASE + kEigSites + 1>();
}
MPM_DEV void stress_nacc(const MaterialConst& mc, float (&b)[6], bool& refl, float& log_jp, float (&PF)[9]) {
	NoHook nh;
	stress_nacc<0>(mc, stress_scale(mc), b, refl, log_jp, PF, nh);
}

// J-fluid (weakly compressible, Tait EOS + Newtonian viscosity), Projects/GMPM/mgmpm_kernels.cuh:476-505
// A: the gathered velocity gradient in CELL units; jdiv = dx dt D^-1 and jvisc = dx D^-1 viscosity are formed on the host (the dx that turns
// A into world units rides in them: nine multiplications less); vol: the model's volume times -new_dt D^-1 dx (StressScale: the stress
// arrives as its P2G term).  The result is symmetric: six entries are computed.
MPM_DEV float stress_jfluid(const MaterialConst& mc, float vol, float jdiv, float jvisc, float J, const float (&A)[9], float (&contrib)[9]) {
	J = fmaf((A[0] + A[4] + A[8]) * jdiv, J, J);
	if(J < 0.1f) J = 0.1f;// reference compares with the double literal 0.1; no float lies in (0.1, 0.1f), so this is identical
	const float voln	 = J * vol;
	const float pressure = mc.bulk * (__builtin_amdgcn_exp2f(-mc.gamma * __builtin_amdgcn_logf(J)) - 1.f);// J^-gamma (J >= 0.1)
	const float kv = jvisc * voln, pv = pressure * voln;
	contrib[0] = fmaf(A[0] + A[0], kv, -pv);
	contrib[4] = fmaf(A[4] + A[4], kv, -pv);
	contrib[8] = fmaf(A[8] + A[8], kv, -pv);
	contrib[1] = contrib[3] = (A[1] + A[3]) * kv;
	contrib[2] = contrib[6] = (A[2] + A[6]) * kv;
	contrib[5] = contrib[7] = (A[5] + A[7]) * kv;
	return J;
}

}// namespace mpm
