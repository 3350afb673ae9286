// mpm_device_math.hpp — register-resident per-particle math for gfx950 (CDNA4).
//
// Device versions of the arithmetic on claymore's G2P2G path.  Everything stays in VGPRs: 3x3 products,
// the McAdams SVD and the four constitutive models (no MFMA: the contractions are 3-wide).  The
// algorithms are the reference's (cited per function, paths relative to /root/reference); the code is
// written for the AMD compiler: selects instead of bit masks (v_cndmask), v_rsq_f32 where the algorithm
// tolerates an approximate reciprocal square root, FMA contraction allowed.
#pragma once
#include <hip/hip_runtime.h>

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
// MatrixUtils.h:29-41: out = m1 * diag * m2^T
MPM_DEV void mat_diag_matT(float (&out)[9], const float (&m1)[9], const float (&dg)[3], const float (&m2)[9]) {
	const v2f_ c0 = {m1[0], m1[1]}, c1 = {m1[3], m1[4]}, c2 = {m1[6], m1[7]};
#pragma unroll
	for(int j = 0; j < 3; ++j) {
		const float t0 = dg[0] * m2[j], t1 = dg[1] * m2[3 + j], t2 = dg[2] * m2[6 + j];
		const v2f_ xy  = c0 * t0 + c1 * t1 + c2 * t2;
		out[3 * j]	   = xy.x;
		out[3 * j + 1] = xy.y;
		out[3 * j + 2] = m1[2] * t0 + m1[5] * t1 + m1[8] * t2;
	}
}
// P F^T * volume (tail of constitutive_models.cuh:63-72)
MPM_DEV void P_Ft_vol(const float (&P)[9], const float (&F)[9], float volume, float (&PF)[9]) {
#pragma unroll
	for(int j = 0; j < 3; ++j) {
#pragma unroll
		for(int i = 0; i < 3; ++i) PF[3 * j + i] = (P[i] * F[j] + P[3 + i] * F[3 + j] + P[6 + i] * F[6 + j]) * volume;
	}
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
#ifdef MPM_EXACT_LOGEXP
	return logf(x);
#else
	return __builtin_amdgcn_logf(x) * 0.693147180559945f;
#endif
}
MPM_DEV float exp_fast(float x) {
#ifdef MPM_EXACT_LOGEXP
	return expf(x);
#else
	return __builtin_amdgcn_exp2f(x * 1.442695040888963f);
#endif
}
MPM_DEV float rsqrt_approx(float x) {
	return __builtin_amdgcn_rsqf(x);// v_rsq_f32, ~1 ulp; the algorithm re-normalises (svd.cuh:210-215)
}
// rsqrt refined by one Newton step, svd.cuh:487-498
MPM_DEV float rsqrt_newton(float x) {
	float t1 = rsqrt_approx(x);
	float t4 = t1 * 0.5f;
	float t3 = t1 * t4;
	t3		 = t1 * t3;
	t3		 = x * t3;
	t1		 = t1 + t4;
	return t1 - t3;
}

// One Jacobi conjugation, Library/MnBase/Math/Matrix/svd.cuh:167-252 (and its two index-permuted copies): same
// approximate Givens angle (rsqrt-normalised (ch, sh), pi/8 fallback).  Two deliberate simplifications, both exact
// in exact arithmetic and ~1e-7 in fp32: (1) columns p, q of V are rotated directly instead of accumulating a
// quaternion that is normalised and expanded afterwards (:236-252, :475-530); (2) the compensation factors
// (sh^2 + ch^2), which equal 1 up to the rsqrt rounding, are dropped (:219-223).
MPM_DEV void jacobi_conj(float& s11, float& s21, float& s22, float& s31, float& s32, float& s33, float (&vp)[3], float (&vq)[3]) {
	float sh   = s21 * 0.5f;
	float tmp5 = s11 - s22;
	float tmp2 = sh * sh;
	bool m	   = tmp2 >= 1.e-20f;
	sh		   = m ? sh : 0.0f;
	float ch   = m ? tmp5 : 1.0f;
	float tmp1 = sh * sh;
	tmp2	   = ch * ch;
	const float tmp4 = rsqrt_approx(tmp1 + tmp2);
	sh		   = tmp4 * sh;
	ch		   = tmp4 * ch;
	m		   = tmp2 <= 5.8284273147583007813f * tmp1;
	sh		   = m ? __uint_as_float(1053028117u) : sh;// sin(pi/8)
	ch		   = m ? __uint_as_float(1064076127u) : ch;// cos(pi/8)
	const float c = ch * ch - sh * sh;
	const float s = 2.f * ch * sh;
	// Givens conjugation of the symmetric matrix
	const float t31 = s * s31, t32 = s * s32;
	s31				= c * s31 + t32;
	s32				= c * s32 - t31;
	const float ss = s * s, cc = c * c, cs = c * s;
	const float n11 = s11 * cc + s22 * ss;
	const float n22 = s22 * cc + s11 * ss;
	const float t2	= (s21 + s21) * cs;
	s21				= s21 * (cc - ss) - tmp5 * cs;
	s11				= n11 + t2;
	s22				= n22 - t2;
	// V <- V G
	const v2f_ p = {vp[0], vp[1]}, q = {vq[0], vq[1]};
	const v2f_ pn = p * c + q * s, qn = q * c - p * s;
	const float pz = vp[2], qz = vq[2];
	vp[0] = pn.x;
	vp[1] = pn.y;
	vp[2] = c * pz + s * qz;
	vq[0] = qn.x;
	vq[1] = qn.y;
	vq[2] = c * qz - s * pz;
}

MPM_DEV void cond_swap(bool c, float& x, float& y) {
	const float t = x;
	x			  = c ? y : x;
	y			  = c ? t : y;
}

// One Givens step of the QR factorisation, svd.cuh:786-880.  (ap*, aq*) are rows p and q of B, (up*, uq*) columns
// p and q of U; apiv / aqpiv are the pivot-column entries (aliases of ap*/aq* elements, read first).
MPM_DEV void qr_givens(float apiv, float aqpiv, float& ap1, float& ap2, float& ap3, float& aq1, float& aq2, float& aq3, float& up1, float& up2, float& up3, float& uq1, float& uq2, float& uq3) {
	float sh   = aqpiv * aqpiv;
	sh		   = (sh >= 1.e-12f) ? aqpiv : 0.0f;
	float ch   = fmaxf(fmaxf(-apiv, apiv), 1.e-12f);
	const bool m = apiv >= 0.f;
	float tmp2 = ch * ch + sh * sh;
	float tmp1 = rsqrt_newton(tmp2) * tmp2;
	ch		   = ch + tmp1;
	{
		const float nch = m ? ch : sh;
		const float nsh = m ? sh : ch;
		ch				= nch;
		sh				= nsh;
	}
	tmp2	= ch * ch + sh * sh;
	tmp1	= rsqrt_newton(tmp2);
	ch		= ch * tmp1;
	sh		= sh * tmp1;
	const float c = ch * ch - sh * sh;
	float s		  = sh * ch;
	s			  = s + s;
#define MPM_ROT(x, y)            \
	{                            \
		const float t1 = s * x;  \
		const float t2 = s * y;  \
		x			   = c * x + t2; \
		y			   = c * y - t1; \
	}
	MPM_ROT(ap1, aq1)
	MPM_ROT(ap2, aq2)
	MPM_ROT(ap3, aq3)
	MPM_ROT(up1, uq1)
	MPM_ROT(up2, uq2)
	MPM_ROT(up3, uq3)
#undef MPM_ROT
}

// math::svd, Library/MnBase/Math/Matrix/svd.cuh:27-1123.  Column-major F, U, V.
// Same algorithm as the reference (4 cyclic Jacobi sweeps on F^T F, columns sorted by norm with V kept a rotation,
// sigma_3 carrying the sign of det F).  U and Sigma come from B = F V: when B is well conditioned its columns are
// sigma_i u_i, so they are normalised directly (sigma_i = |b_i|, sign of sigma_3 from det B); the reference's Givens
// QR (:771-1122) is kept as the path for ill-conditioned B (sigma_3 < 1e-3 sigma_1), where normalising would divide
// by ~0.  Deviation from the reference's QR result is of the order of the Jacobi residual of the reference itself
// (tests/test_parity_gpu.py, tools note in DESIGN.md section 6).
//
// Hook: the G2P2G kernel threads an unrelated, latency-bound chain of LDS read-modify-write steps (the P2G scatter of
// the previous particle) through this arithmetic; `hk.at<SITE>()` is called at kSvdSites evenly spaced points that
// every lane reaches (never inside divergent control flow).  NoHook compiles to nothing.
struct NoHook {
	template<int SITE>
	MPM_DEV void at() {}
};
constexpr int kSvdSites = 16;
// SORTED = false (the stress functions): the columns are left in the order Jacobi produced them.  Every consumer on
// this path is symmetric under a simultaneous permutation of (sigma_i, u_i, v_i), and the three conditional column
// swaps cost ~60 VALU instructions per particle; only the sign convention (a negative determinant goes to the SMALLEST
// singular value) is kept.  SORTED = true is math::svd's contract and is what mpm_test_svd exposes.
template<int BASE, class Hook, bool SORTED = true>
MPM_DEV void svd3(const float (&F)[9], float (&U)[9], float (&S)[3], float (&V)[9], Hook& hk) {
	float s11 = F[0] * F[0] + F[1] * F[1] + F[2] * F[2];
	float s21 = F[3] * F[0] + F[4] * F[1] + F[5] * F[2];
	float s31 = F[6] * F[0] + F[7] * F[1] + F[8] * F[2];
	float s22 = F[3] * F[3] + F[4] * F[4] + F[5] * F[5];
	float s32 = F[6] * F[3] + F[7] * F[4] + F[8] * F[5];
	float s33 = F[6] * F[6] + F[7] * F[7] + F[8] * F[8];
	float v1[3] = {1.f, 0.f, 0.f}, v2[3] = {0.f, 1.f, 0.f}, v3[3] = {0.f, 0.f, 1.f};
	MPM_MARK("svd_jacobi");
	hk.template at<BASE + 0>();
	// The reference always runs 4 sweeps (svd.cuh:167) of its approximate-angle rotations.  Cyclic Jacobi converges
	// quadratically, so once every off-diagonal entry of F^T F of every lane is below 1e-6 of the smallest diagonal
	// entry, U's columns are orthogonal to ~1e-6 (what the reference's four approximate sweeps reach themselves) and
	// the singular values are exact to ~1e-12 relative; the remaining sweeps are skipped for the whole wave then
	// (`done` is wave-uniform: scalar branches).
	bool done = false;
#define MPM_SWEEP(IT)                                                                  \
	if(!done) jacobi_conj(s11, s21, s22, s31, s32, s33, v1, v2);                      \
	hk.template at<BASE + 1 + 3 * IT>();                                              \
	if(!done) jacobi_conj(s22, s32, s33, s21, s31, s11, v2, v3);                      \
	hk.template at<BASE + 2 + 3 * IT>();                                              \
	if(!done) {                                                                       \
		jacobi_conj(s33, s31, s11, s32, s21, s22, v3, v1);                            \
		const float off = fmaxf(fmaxf(fabsf(s21), fabsf(s31)), fabsf(s32));           \
		const float dia = fminf(fminf(fabsf(s11), fabsf(s22)), fabsf(s33));           \
		done			= __all(off <= 1e-6f * dia);                                  \
	}                                                                                 \
	hk.template at<BASE + 3 + 3 * IT>();
	MPM_SWEEP(0)
	MPM_SWEEP(1)
	MPM_SWEEP(2)
	MPM_SWEEP(3)
#undef MPM_SWEEP
	MPM_MARK("svd_post");
	// B = F V (svd.cuh:532-588), columns b1 b2 b3
	float b1[3], b2[3], b3[3];
	{
		const v2f_ f0 = {F[0], F[1]}, f1 = {F[3], F[4]}, f2 = {F[6], F[7]};
#define MPM_FV(bk, vk)                                            \
	{                                                             \
		const v2f_ xy = f0 * vk[0] + f1 * vk[1] + f2 * vk[2];     \
		bk[0]		  = xy.x;                                     \
		bk[1]		  = xy.y;                                     \
		bk[2]		  = F[2] * vk[0] + F[5] * vk[1] + F[8] * vk[2]; \
	}
		MPM_FV(b1, v1)
		MPM_FV(b2, v2)
		MPM_FV(b3, v3)
#undef MPM_FV
	}
	float n1 = b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2];
	float n2 = b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2];
	float n3 = b3[0] * b3[0] + b3[1] * b3[1] + b3[2] * b3[2];
	hk.template at<BASE + 13>();
	// sort columns by squared norm, descending; a swap negates one column so that V stays a rotation (svd.cuh:590-770)
#define MPM_SWAPCOL(c, x, y, nx, ny, neg)                         \
	{                                                             \
		const float sg = (c) ? -1.f : 1.f;                        \
		_Pragma("unroll") for(int r = 0; r < 3; ++r) {            \
			cond_swap(c, b##x[r], b##y[r]);                       \
			cond_swap(c, v##x[r], v##y[r]);                       \
			b##neg[r] *= sg;                                      \
			v##neg[r] *= sg;                                      \
		}                                                         \
		cond_swap(c, nx, ny);                                     \
	}
	if constexpr(SORTED) {
		MPM_SWAPCOL(n1 < n2, 1, 2, n1, n2, 2)
		MPM_SWAPCOL(n1 < n3, 1, 3, n1, n3, 1)
		MPM_SWAPCOL(n2 < n3, 2, 3, n2, n3, 3)
	}
	const float nmin = fminf(fminf(n1, n2), n3), nmax = fmaxf(fmaxf(n1, n2), n3);
	bool well = nmin > 1e-6f * nmax;
	if constexpr(!SORTED) {
		if(!well) {// ill conditioned (rare): sort after all, the Givens QR below wants descending columns
			MPM_SWAPCOL(n1 < n2, 1, 2, n1, n2, 2)
			MPM_SWAPCOL(n1 < n3, 1, 3, n1, n3, 1)
			MPM_SWAPCOL(n2 < n3, 2, 3, n2, n3, 3)
		}
	}
#undef MPM_SWAPCOL
#pragma unroll
	for(int r = 0; r < 3; ++r) {
		V[r]	 = v1[r];
		V[3 + r] = v2[r];
		V[6 + r] = v3[r];
	}
	hk.template at<BASE + 14>();
	if(well) {
		// well conditioned: u_i = b_i / sigma_i
		const float det = b1[0] * (b2[1] * b3[2] - b2[2] * b3[1]) - b2[0] * (b1[1] * b3[2] - b1[2] * b3[1]) + b3[0] * (b1[1] * b2[2] - b1[2] * b2[1]);
		float i1 = rsqrt_approx(n1), i2 = rsqrt_approx(n2);// v_rsq_f32 is 1 ulp: no Newton step needed (svd.cuh:487-498 refines a 12-bit estimate)
		float i3 = rsqrt_approx(n3);
		if(det < 0.f) {// the smallest singular value carries the sign of det F (svd.cuh:590-770 does this through the sort)
			if constexpr(SORTED) {
				i3 = -i3;
			} else {
				const bool m1 = n1 <= n2 && n1 <= n3, m2 = !m1 && n2 <= n3;
				i1 = m1 ? -i1 : i1;
				i2 = m2 ? -i2 : i2;
				i3 = (!m1 && !m2) ? -i3 : i3;
			}
		}
		S[0] = n1 * i1;
		S[1] = n2 * i2;
		S[2] = n3 * i3;
#pragma unroll
		for(int r = 0; r < 3; ++r) {
			U[r]	 = b1[r] * i1;
			U[3 + r] = b2[r] * i2;
			U[6 + r] = b3[r] * i3;
		}
	} else {
		// ill conditioned: the reference's QR by three Givens rotations (svd.cuh:772-1090)
		float a11 = b1[0], a21 = b1[1], a31 = b1[2], a12 = b2[0], a22 = b2[1], a32 = b2[2], a13 = b3[0], a23 = b3[1], a33 = b3[2];
		float u11 = 1.f, u12 = 0.f, u13 = 0.f, u21 = 0.f, u22 = 1.f, u23 = 0.f, u31 = 0.f, u32 = 0.f, u33 = 1.f;
		qr_givens(a11, a21, a11, a12, a13, a21, a22, a23, u11, u21, u31, u12, u22, u32);
		qr_givens(a11, a31, a11, a12, a13, a31, a32, a33, u11, u21, u31, u13, u23, u33);
		qr_givens(a22, a32, a21, a22, a23, a31, a32, a33, u12, u22, u32, u13, u23, u33);
		U[0] = u11;
		U[1] = u21;
		U[2] = u31;
		U[3] = u12;
		U[4] = u22;
		U[5] = u32;
		U[6] = u13;
		U[7] = u23;
		U[8] = u33;
		S[0] = a11;
		S[1] = a22;
		S[2] = a33;
	}
	hk.template at<BASE + 15>();
	MPM_MARK("svd_end");
}
MPM_DEV void svd3(const float (&F)[9], float (&U)[9], float (&S)[3], float (&V)[9]) {
	NoHook nh;
	svd3<0>(F, U, S, V, nh);
}

// Material constants passed by value to the kernels (Projects/GMPM/particle_buffer.cuh:141-264)
struct MaterialConst {
	float mass, volume, mu, lambda;
	float bulk, gamma, viscosity;		 // J_FLUID
	float cohesion, beta, yield_surface;// SAND (beta also NACC)
	float bm, xi, msqr;					 // NACC
	float log_jp0;
	int volume_correction, hardening_on;
};

// compute_stress<FIXED_COROTATED>, Projects/GMPM/constitutive_models.cuh:36-73
constexpr int kFcSites = kSvdSites + 2;
template<int BASE, class Hook>
MPM_DEV void stress_fixed_corotated(const MaterialConst& mc, const float (&F)[9], float (&PF)[9], Hook& hk) {
	float U[9], S[3], V[9];
	svd3<BASE, Hook, false>(F, U, S, V, hk);
	const float J			  = S[0] * S[1] * S[2];
	const float scaled_mu	  = 2.0f * mc.mu;
	const float scaled_lambda = mc.lambda * (J - 1.0f);
	float Ph[3];
	Ph[0] = scaled_mu * (S[0] - 1.f) + scaled_lambda * (S[1] * S[2]);
	Ph[1] = scaled_mu * (S[1] - 1.f) + scaled_lambda * (S[0] * S[2]);
	Ph[2] = scaled_mu * (S[2] - 1.f) + scaled_lambda * (S[0] * S[1]);
	float P[9];
	mat_diag_matT(P, U, Ph, V);
	hk.template at<BASE + kSvdSites>();
	P_Ft_vol(P, F, mc.volume, PF);
	hk.template at<BASE + kSvdSites + 1>();
}
MPM_DEV void stress_fixed_corotated(const MaterialConst& mc, const float (&F)[9], float (&PF)[9]) {
	NoHook nh;
	stress_fixed_corotated<0>(mc, F, PF, nh);
}

// compute_stress<SAND>, constitutive_models.cuh:238-335 (Drucker-Prager return mapping, StVK-Hencky)
constexpr int kSandSites = kSvdSites + 4;
template<int BASE, class Hook>
MPM_DEV void stress_sand(const MaterialConst& mc, float (&F)[9], float& log_jp, float (&PF)[9], Hook& hk, float* __restrict__ Fdst = nullptr, int Fstride = 0) {
	float U[9], S[3], V[9];
	svd3<BASE, Hook, false>(F, U, S, V, hk);
	const float scaled_mu = 2.0f * mc.mu;
	float epsilon[3], New_S[3] = {0.f, 0.f, 0.f};
#pragma unroll
	for(int i = 0; i < 3; i++) {
		const float abs_S = fmaxf(fabsf(S[i]), 1e-4f);
		epsilon[i]		  = log_fast(abs_S) - mc.cohesion;
	}
	const float sum_epsilon	  = epsilon[0] + epsilon[1] + epsilon[2];
	const float trace_epsilon = sum_epsilon + log_jp;
	float epsilon_hat[3];
#pragma unroll
	for(int i = 0; i < 3; i++) epsilon_hat[i] = epsilon[i] - (trace_epsilon * (1.0f / 3.0f));
	const float epsilon_hat_norm = __builtin_amdgcn_sqrtf(epsilon_hat[0] * epsilon_hat[0] + epsilon_hat[1] * epsilon_hat[1] + epsilon_hat[2] * epsilon_hat[2]);
	hk.template at<BASE + kSvdSites>();
	// log(New_S) is needed below; New_S = exp(H), so H itself is used instead of logf(expf(H)) (the reference's
	// round trip, constitutive_models.cuh:311, differs from H by one rounding of expf: ~6e-8 absolute)
	float lnS[3];
	bool rebuild = false;
	if(trace_epsilon >= 0.0f) {// case II: cone tip
		New_S[0] = New_S[1] = New_S[2] = exp_fast(mc.cohesion);
		lnS[0] = lnS[1] = lnS[2] = mc.cohesion;
		rebuild					 = true;
		if(mc.volume_correction) log_jp = mc.beta * sum_epsilon + log_jp;
	} else if(mc.mu != 0.f) {
		log_jp					= 0.f;
		const float delta_gamma = epsilon_hat_norm + (3.0f * mc.lambda + scaled_mu) * rcp_fast(scaled_mu) * trace_epsilon * mc.yield_surface;
		if(delta_gamma <= 0.f) {// case I: inside the cone
#pragma unroll
			for(int i = 0; i < 3; i++) lnS[i] = epsilon[i] + mc.cohesion;
		} else {// case III: project to the cone surface
			const float r = delta_gamma * rcp_fast(epsilon_hat_norm);
#pragma unroll
			for(int i = 0; i < 3; i++) lnS[i] = epsilon[i] - r * epsilon_hat[i] + mc.cohesion;
		}
#pragma unroll
		for(int i = 0; i < 3; i++) New_S[i] = exp_fast(lnS[i]);
		rebuild = true;
	} else {
		lnS[0] = lnS[1] = lnS[2] = -__builtin_inff();// reference: logf(0) when mu == 0 (:298-300)
	}
	hk.template at<BASE + kSvdSites + 1>();
	if(rebuild) mat_diag_matT(F, U, New_S, V);
	if(Fdst) {// the kernel has the projected F written out here: nine registers less while P F^T is formed
#pragma unroll
		for(int d = 0; d < 9; ++d) Fdst[d * Fstride] = F[d];
	}
	hk.template at<BASE + kSvdSites + 2>();
	const float trace_log_S = lnS[0] + lnS[1] + lnS[2];
	// P F^T vol (:318-334).  With F = U New_S V^T as rebuilt above, P F^T = U P_hat V^T V New_S U^T = U diag(P_hat_k New_S_k) U^T
	// with P_hat_k New_S_k = 2 mu ln S_k + lambda tr(ln S): 27 FMAs instead of two 3x3x3 products and three reciprocals
	// (V^T V = I to ~1e-7, the same order as the reference's own SVD residual).
	// (mu == 0 leaves F as it is and ln S = -inf: the reference's P is NaN then, and so is this.)  No branch here on
	// purpose: stores to PF in two arms get merged into one store with a variable offset, which keeps PF in scratch.
	{
		float d[3];
#pragma unroll
		for(int k = 0; k < 3; ++k) d[k] = (scaled_mu * lnS[k] + mc.lambda * trace_log_S) * mc.volume;
		const v2f_ u0 = {U[0], U[1]}, u1 = {U[3], U[4]}, u2 = {U[6], U[7]};
		const v2f_ ud0 = u0 * d[0], ud1 = u1 * d[1], ud2 = u2 * d[2];
		const float udz0 = U[2] * d[0], udz1 = U[5] * d[1], udz2 = U[8] * d[2];
		const v2f_ c0 = ud0 * U[0] + ud1 * U[3] + ud2 * U[6];// (PF00, PF10)
		const v2f_ c1 = ud0 * U[1] + ud1 * U[4] + ud2 * U[7];// (PF01, PF11)
		const v2f_ c2 = ud0 * U[2] + ud1 * U[5] + ud2 * U[8];// (PF02, PF12)
		PF[0] = c0.x;
		PF[1] = c0.y;
		PF[3] = c0.y;
		PF[4] = c1.y;
		PF[6] = c2.x;
		PF[7] = c2.y;
		PF[2] = c2.x;
		PF[5] = c2.y;
		PF[8] = udz0 * U[2] + udz1 * U[5] + udz2 * U[8];
	}
	hk.template at<BASE + kSvdSites + 3>();
}
MPM_DEV void stress_sand(const MaterialConst& mc, float (&F)[9], float& log_jp, float (&PF)[9]) {
	NoHook nh;
	stress_sand<0>(mc, F, log_jp, PF, nh);
}

// compute_stress<NACC>, constitutive_models.cuh:77-234 (USE_JOSH_FRACTURE_PAPER branch)
constexpr int kNaccSites = kSvdSites + 2;
template<int BASE, class Hook>
MPM_DEV void stress_nacc(const MaterialConst& mc, float (&F)[9], float& log_jp, float (&PF)[9], Hook& hk) {
	float U[9], S[3], V[9];
	svd3<BASE, Hook, false>(F, U, S, V, hk);
	const float bm	  = mc.bm;
	const float p0	  = bm * (0.00001f + sinhf(mc.xi * (-log_jp > 0 ? -log_jp : 0)));
	const float p_min = -mc.beta * p0;
	const float Je_trial = S[0] * S[1] * S[2];
	const float B0 = S[0] * S[0], B1 = S[1] * S[1], B2 = S[2] * S[2];
	const float trB3   = (B0 + B1 + B2) / 3.f;
	const float Jm23mu = mc.mu * powf(Je_trial, -2.f / 3.f);
	const float sh0 = Jm23mu * (B0 - trB3), sh1 = Jm23mu * (B1 - trB3), sh2 = Jm23mu * (B2 - trB3);
	const float psi_kappa_partial_J = bm * 0.5f * (Je_trial - 1.f / Je_trial);
	const float p_trial				= -psi_kappa_partial_J * Je_trial;
	const float y_s_half_coeff		= 3.f / 2.f * (1 + 2.f * mc.beta);
	const float y_p_half			= (mc.msqr * (p_trial - p_min) * (p_trial - p0));
	const float s_sqrnorm			= sh0 * sh0 + sh1 * sh1 + sh2 * sh2;
	const float y					= (y_s_half_coeff * s_sqrnorm) + y_p_half;
	if(p_trial > p0) {
		const float Je_new = sqrtf(-2.f * p0 / bm + 1.f);
		S[0] = S[1] = S[2] = powf(Je_new, 1.f / 3.f);
		mat_diag_matT(F, U, S, V);
		if(mc.hardening_on) log_jp += logf(Je_trial / Je_new);
	} else if(p_trial < p_min) {
		const float Je_new = sqrtf(-2.f * p_min / bm + 1.f);
		S[0] = S[1] = S[2] = powf(Je_new, 1.f / 3.f);
		mat_diag_matT(F, U, S, V);
		if(mc.hardening_on) log_jp += logf(Je_trial / Je_new);
	} else if(y >= 1e-4f) {
		const float B_s_coeff = powf(Je_trial, 2.f / 3.f) / mc.mu * sqrtf(-y_p_half / y_s_half_coeff) / sqrtf(s_sqrnorm);
		S[0]				  = sqrtf(sh0 * B_s_coeff + trB3);
		S[1]				  = sqrtf(sh1 * B_s_coeff + trB3);
		S[2]				  = sqrtf(sh2 * B_s_coeff + trB3);
		mat_diag_matT(F, U, S, V);
		if(mc.hardening_on && p0 > 1e-4f && p_trial < p0 - 1e-4f && p_trial > 1e-4f + p_min) {
			const float p_center = (1.0f - mc.beta) * p0 / 2;
			const float q_trial	 = sqrtf(3.f / 2.f * s_sqrnorm);
			float d0 = p_center - p_trial, d1 = -q_trial;
			const float dn = sqrtf(d0 * d0 + d1 * d1);
			d0 /= dn;
			d1 /= dn;
			const float C  = mc.msqr * (p_center - p_min) * (p_center - p0);
			const float B  = mc.msqr * d0 * (2 * p_center - p0 - p_min);
			const float A  = mc.msqr * d0 * d0 + (1 + 2 * mc.beta) * d1 * d1;
			const float sq = sqrtf(B * B - 4 * A * C);
			const float l1 = (-B + sq) / (2 * A);
			const float l2 = (-B - sq) / (2 * A);
			const float p1 = p_center + l1 * d0;
			const float p2 = p_center + l2 * d0;
			const float p_fake		= (p_trial - p_center) * (p1 - p_center) > 0 ? p1 : p2;
			const float tmp_Je_sqr	= (-2 * p_fake / bm + 1);
			const float Je_new_fake = sqrtf(tmp_Je_sqr > 0 ? tmp_Je_sqr : -tmp_Je_sqr);
			if(Je_new_fake > 1e-4f) log_jp += logf(Je_trial / Je_new_fake);
		}
	}
	hk.template at<BASE + kSvdSites>();
	const float J = S[0] * S[1] * S[2];
	float b[9];
#pragma unroll
	for(int j = 0; j < 3; ++j) {
#pragma unroll
		for(int i = 0; i < 3; ++i) b[3 * j + i] = F[i] * F[j] + F[3 + i] * F[3 + j] + F[6 + i] * F[6 + j];
	}
	const float twothirds = (float) (2.0 / 3.0);
	const float bd0		  = b[0] * twothirds - (b[4] + b[8]) / 3.0f;
	const float bd4		  = b[4] * twothirds - (b[0] + b[8]) / 3.0f;
	const float bd8		  = b[8] * twothirds - (b[0] + b[4]) / 3.0f;
	const float dev_b_coeff = mc.mu * powf(J, -2.f / 3.f);
	const float i_coeff		= bm * .5f * ((J * J - 1.f) * 0.5f - logf(J));
	PF[0]					= (dev_b_coeff * bd0 + i_coeff) * mc.volume;
	PF[1]					= (dev_b_coeff * b[1]) * mc.volume;
	PF[2]					= (dev_b_coeff * b[2]) * mc.volume;
	PF[3]					= (dev_b_coeff * b[3]) * mc.volume;
	PF[4]					= (dev_b_coeff * bd4 + i_coeff) * mc.volume;
	PF[5]					= (dev_b_coeff * b[5]) * mc.volume;
	PF[6]					= (dev_b_coeff * b[6]) * mc.volume;
	PF[7]					= (dev_b_coeff * b[7]) * mc.volume;
	PF[8]					= (dev_b_coeff * bd8 + i_coeff) * mc.volume;
	hk.template at<BASE + kSvdSites + 1>();
}
MPM_DEV void stress_nacc(const MaterialConst& mc, float (&F)[9], float& log_jp, float (&PF)[9]) {
	NoHook nh;
	stress_nacc<0>(mc, F, log_jp, PF, nh);
}

// J-fluid (weakly compressible, Tait EOS + Newtonian viscosity), Projects/GMPM/mgmpm_kernels.cuh:476-505
MPM_DEV float stress_jfluid(const MaterialConst& mc, float J, const float (&A)[9], float dt, float d_inv, float (&contrib)[9]) {
	J += (A[0] + A[4] + A[8]) * dt * d_inv * J;
	if(J < 0.1f) J = 0.1f;// reference compares with the double literal 0.1; no float lies in (0.1, 0.1f), so this is identical
	const float voln	 = J * mc.volume;
	const float pressure = mc.bulk * (powf(J, -mc.gamma) - 1.f);
	const float k		 = d_inv * mc.viscosity;
	contrib[0]			 = ((A[0] + A[0]) * k - pressure) * voln;
	contrib[1]			 = (A[1] + A[3]) * k * voln;
	contrib[2]			 = (A[2] + A[6]) * k * voln;
	contrib[3]			 = (A[3] + A[1]) * k * voln;
	contrib[4]			 = ((A[4] + A[4]) * k - pressure) * voln;
	contrib[5]			 = (A[5] + A[7]) * k * voln;
	contrib[6]			 = (A[6] + A[2]) * k * voln;
	contrib[7]			 = (A[7] + A[5]) * k * voln;
	contrib[8]			 = ((A[8] + A[8]) * k - pressure) * voln;
	return J;
}

}// namespace mpm
