/*
 * mpm_oracle_math.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement, in plain C, of the per-particle arithmetic on claymore's substep hot path.
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Arithmetic is IEEE fp32, evaluated in the reference's operation order; compile with
 * -ffp-contract=off so that no multiply-add is fused where the reference forbids it
 * (__fadd_rn/__fsub_rn, svd.cuh:125-157).
 *
 * Pinned against golden vectors generated from the reference's own functions
 * (tests/golden/gen/gen_golden.cpp -> tests/golden/g*.f32), see tests/test_oracle_golden.py.
 */
#ifndef MPM_ORACLE_MATH_H
#define MPM_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

/* ---- Projects/GMPM/utility_funcs.hpp:10-19 bspline_weight: p = local offset in world units ---- */
static inline void orc_bspline_weight(float p, float dx_inv, float dw[3]) {
	float d = p * dx_inv;
	dw[0]	= 0.5f * (1.5f - d) * (1.5f - d);
	d -= 1.0f;
	dw[1] = 0.75f - d * d;
	d	  = 0.5f + d;
	dw[2] = 0.5f * d * d;
}

/* utility_funcs.hpp:21-23 get_block_id: nearest grid node, round half away from zero */
static inline int orc_node_index(float x, float dx_inv) {
	return (int) lroundf(x * dx_inv);
}

/* utility_funcs.hpp:25-32 */
static inline int orc_dir_offset(int dx, int dy, int dz) {
	return (dx + 1) * 9 + (dy + 1) * 3 + dz + 1;
}
static inline void orc_dir_components(int dir, int d[3]) {
	d[2] = (dir % 3) - 1;
	d[1] = ((dir / 3) % 3) - 1;
	d[0] = ((dir / 9) % 3) - 1;
}

/* utility_funcs.hpp:36-49 compute_dt */
static inline float orc_compute_dt(float max_vel, float cur_time, float next_time, float dt_default, float dx, float cfl) {
	float dt = dt_default;
	if(max_vel > 0.0f) {
		const float new_dt = dx * cfl / max_vel;
		dt				   = new_dt < dt ? new_dt : dt;
	}
	const float rem = next_time - cur_time;
	dt				= dt < rem ? dt : rem;
	return dt;
}

/* Projects/MGSP/utility_funcs.hpp:32-55 compute_dt of the MGSP project: CFL 0.3, the frame is never overshot, and a step that
 * does not end the frame is at most 0.51 of what is left of it (pinned by tests/golden/g12_mgsp_dt_*.f32) */
static inline float orc_compute_dt_mgsp(float max_vel, float cur, float next, float dt_default, float dx) {
	if(next < cur) return 0.0f;
	float dt = dt_default;
	if(max_vel > 0.0f) {
		max_vel = dx * 0.3f / max_vel;
		if(max_vel < dt_default) dt = max_vel;
	}
	if(cur + dt >= next) {
		dt = next - cur;
	} else {
		max_vel = (next - cur) * 0.51f;
		if(max_vel < dt) dt = max_vel;
	}
	return dt;
}

/* ---- Library/MnBase/Math/Matrix/MatrixUtils.h (column-major 3x3) ---- */
/* :147-157 */
static inline void orc_matmul3(const float* a, const float* b, float* c) {
	c[0] = a[0] * b[0] + a[3] * b[1] + a[6] * b[2];
	c[1] = a[1] * b[0] + a[4] * b[1] + a[7] * b[2];
	c[2] = a[2] * b[0] + a[5] * b[1] + a[8] * b[2];
	c[3] = a[0] * b[3] + a[3] * b[4] + a[6] * b[5];
	c[4] = a[1] * b[3] + a[4] * b[4] + a[7] * b[5];
	c[5] = a[2] * b[3] + a[5] * b[4] + a[8] * b[5];
	c[6] = a[0] * b[6] + a[3] * b[7] + a[6] * b[8];
	c[7] = a[1] * b[6] + a[4] * b[7] + a[7] * b[8];
	c[8] = a[2] * b[6] + a[5] * b[7] + a[8] * b[8];
}
/* :29-41 out = m1 * diag * m2^T */
static inline void orc_mat_diag_matT(float* out, const float* m1, const float* dg, const float* m2) {
	out[0] = m1[0] * dg[0] * m2[0] + m1[3] * dg[1] * m2[3] + m1[6] * dg[2] * m2[6];
	out[1] = m1[1] * dg[0] * m2[0] + m1[4] * dg[1] * m2[3] + m1[7] * dg[2] * m2[6];
	out[2] = m1[2] * dg[0] * m2[0] + m1[5] * dg[1] * m2[3] + m1[8] * dg[2] * m2[6];
	out[3] = m1[0] * dg[0] * m2[1] + m1[3] * dg[1] * m2[4] + m1[6] * dg[2] * m2[7];
	out[4] = m1[1] * dg[0] * m2[1] + m1[4] * dg[1] * m2[4] + m1[7] * dg[2] * m2[7];
	out[5] = m1[2] * dg[0] * m2[1] + m1[5] * dg[1] * m2[4] + m1[8] * dg[2] * m2[7];
	out[6] = m1[0] * dg[0] * m2[2] + m1[3] * dg[1] * m2[5] + m1[6] * dg[2] * m2[8];
	out[7] = m1[1] * dg[0] * m2[2] + m1[4] * dg[1] * m2[5] + m1[7] * dg[2] * m2[8];
	out[8] = m1[2] * dg[0] * m2[2] + m1[5] * dg[1] * m2[5] + m1[8] * dg[2] * m2[8];
}
/* :257-269 out = in * in^T */
static inline void orc_mat_matT(const float* in, float* out) {
	out[0] = in[0] * in[0] + in[3] * in[3] + in[6] * in[6];
	out[1] = in[1] * in[0] + in[4] * in[3] + in[7] * in[6];
	out[2] = in[2] * in[0] + in[5] * in[3] + in[8] * in[6];
	out[3] = in[0] * in[1] + in[3] * in[4] + in[6] * in[7];
	out[4] = in[1] * in[1] + in[4] * in[4] + in[7] * in[7];
	out[5] = in[2] * in[1] + in[5] * in[4] + in[8] * in[7];
	out[6] = in[0] * in[2] + in[3] * in[5] + in[6] * in[8];
	out[7] = in[1] * in[2] + in[4] * in[5] + in[7] * in[8];
	out[8] = in[2] * in[2] + in[5] * in[5] + in[8] * in[8];
}
/* :272-286 deviatoric part (the reference's rewritten form) */
static inline void orc_deviatoric(const float* in, float* out) {
	out[0] = in[0] * (float) (2.0 / 3.0) - (in[4] + in[8]) / 3.0f;
	out[1] = in[1];
	out[2] = in[2];
	out[3] = in[3];
	out[4] = in[4] * (float) (2.0 / 3.0) - (in[0] + in[8]) / 3.0f;
	out[5] = in[5];
	out[6] = in[6];
	out[7] = in[7];
	out[8] = in[8] * (float) (2.0 / 3.0) - (in[0] + in[4]) / 3.0f;
}

/* ---- Library/MnBase/Math/Matrix/svd.cuh:27-1123: McAdams et al. branch-free 3x3 SVD ---- */
typedef union {
	float f;
	uint32_t u;
} orc_fu;
static inline float orc_sel(int cond, float a, float b) {
	return cond ? a : b;
}
/* __frsqrt_rn: correctly rounded reciprocal square root (svd.cuh:181 and 9 more sites) */
static inline float orc_rsqrt(float x) {
	return (float) (1.0 / sqrt((double) x));
}
#define ORC_FOUR_GAMMA_SQUARED 5.8284273147583007813f /* svd.cuh:16 */
#define ORC_TINY 1.e-20f							  /* svd.cuh:15 */
#define ORC_SMALL 1.e-12f							  /* svd.cuh:14 */
static inline float orc_from_bits(uint32_t u) {
	orc_fu x;
	x.u = u;
	return x.f;
}

/* One Jacobi conjugation (svd.cuh:167-252 for (p,q)=(1,2); :257-345 and :351-441 are the same code with the
 * indices cyclically permuted).  Arguments follow the (1,2) instance: the rotation zeroes s21. */
static inline void orc_jacobi_conj(float* s11, float* s21, float* s22, float* s31, float* s32, float* s33, float* qx, float* qy, float* qz, float* qs) {
	float sh   = *s21 * 0.5f;
	float tmp5 = *s11 - *s22;
	float tmp2 = sh * sh;
	int m	   = tmp2 >= ORC_TINY;
	sh		   = m ? sh : 0.0f;
	float ch   = m ? tmp5 : 1.0f;
	float tmp1 = sh * sh;
	tmp2	   = ch * ch;
	float tmp3 = tmp1 + tmp2;
	float tmp4 = orc_rsqrt(tmp3);
	sh		   = tmp4 * sh;
	ch		   = tmp4 * ch;
	tmp1	   = ORC_FOUR_GAMMA_SQUARED * tmp1;
	m		   = tmp2 <= tmp1;
	sh		   = m ? orc_from_bits(1053028117u) : sh; /* sin(pi/8), svd.cuh:11 */
	ch		   = m ? orc_from_bits(1064076127u) : ch; /* cos(pi/8), svd.cuh:12 */
	tmp1	   = sh * sh;
	tmp2	   = ch * ch;
	float c	   = tmp2 - tmp1;
	float s	   = ch * sh;
	s		   = s + s;
	/* Givens conjugation */
	tmp3 = tmp1 + tmp2;
	*s33 = *s33 * tmp3;
	*s31 = *s31 * tmp3;
	*s32 = *s32 * tmp3;
	*s33 = *s33 * tmp3;
	tmp1 = s * *s31;
	tmp2 = s * *s32;
	*s31 = c * *s31;
	*s32 = c * *s32;
	*s31 = tmp2 + *s31;
	*s32 = *s32 - tmp1;
	tmp2 = s * s;
	tmp1 = *s22 * tmp2;
	tmp3 = *s11 * tmp2;
	tmp4 = c * c;
	*s11 = *s11 * tmp4;
	*s22 = *s22 * tmp4;
	*s11 = *s11 + tmp1;
	*s22 = *s22 + tmp3;
	tmp4 = tmp4 - tmp2;
	tmp2 = *s21 + *s21;
	*s21 = *s21 * tmp4;
	tmp4 = c * s;
	tmp2 = tmp2 * tmp4;
	tmp5 = tmp5 * tmp4;
	*s11 = *s11 + tmp2;
	*s21 = *s21 - tmp5;
	*s22 = *s22 - tmp2;
	/* cumulative rotation as a quaternion */
	tmp1 = sh * *qx;
	tmp2 = sh * *qy;
	tmp3 = sh * *qz;
	sh	 = sh * *qs;
	*qs	 = ch * *qs;
	*qx	 = ch * *qx;
	*qy	 = ch * *qy;
	*qz	 = ch * *qz;
	*qz	 = *qz + sh;
	*qs	 = *qs - tmp3;
	*qx	 = *qx + tmp2;
	*qy	 = *qy - tmp1;
}

/* conditional column swap with negation (svd.cuh:592-650 and the two following copies) */
static inline void orc_cond_swap(int c, float* x, float* y) {
	float t = *x;
	*x		= c ? *y : *x;
	*y		= c ? t : *y;
}

/* rsqrt followed by one Newton step, exactly as written at svd.cuh:487-498 / :796-806 */
static inline float orc_rsqrt_newton(float x) {
	float t1 = orc_rsqrt(x);
	float t4 = t1 * 0.5f;
	float t3 = t1 * t4;
	t3		 = t1 * t3;
	t3		 = x * t3;
	t1		 = t1 + t4;
	t1		 = t1 - t3;
	return t1;
}

/* One Givens step of the QR factorisation (svd.cuh:786-880): zeroes *aq (row q, pivot column) against
 * the pivot *ap.  Rows p and q of A (3 entries each) and columns p and q of U are rotated. */
static inline void orc_qr_givens(float* apiv, float* aq_piv, float* ap[3], float* aq[3], float* up[3], float* uq[3]) {
	float sh   = *aq_piv * *aq_piv;
	sh		   = (sh >= ORC_SMALL) ? *aq_piv : 0.0f;
	float tmp5 = 0.f;
	float ch   = tmp5 - *apiv;
	ch		   = fmaxf(ch, *apiv);
	ch		   = fmaxf(ch, ORC_SMALL);
	int m	   = *apiv >= tmp5;
	float tmp1 = ch * ch;
	float tmp2 = sh * sh;
	tmp2	   = tmp1 + tmp2;
	tmp1	   = orc_rsqrt_newton(tmp2);
	tmp1	   = tmp1 * tmp2;
	ch		   = ch + tmp1;
	{
		float nch = m ? ch : sh;
		float nsh = m ? sh : ch;
		ch		  = nch;
		sh		  = nsh;
	}
	tmp1	= ch * ch;
	tmp2	= sh * sh;
	tmp2	= tmp1 + tmp2;
	tmp1	= orc_rsqrt_newton(tmp2);
	ch		= ch * tmp1;
	sh		= sh * tmp1;
	float c = ch * ch;
	float s = sh * sh;
	c		= c - s;
	s		= sh * ch;
	s		= s + s;
	for(int j = 0; j < 3; ++j) {
		float t1 = s * *ap[j];
		float t2 = s * *aq[j];
		*ap[j]	 = c * *ap[j];
		*aq[j]	 = c * *aq[j];
		*ap[j]	 = *ap[j] + t2;
		*aq[j]	 = *aq[j] - t1;
	}
	for(int i = 0; i < 3; ++i) {
		float t1 = s * *up[i];
		float t2 = s * *uq[i];
		*up[i]	 = c * *up[i];
		*uq[i]	 = c * *uq[i];
		*up[i]	 = *up[i] + t2;
		*uq[i]	 = *uq[i] - t1;
	}
}

/* EXPERIMENT KNOB, not the reference algorithm: a converged double-precision SVD (one-sided Jacobi on F^T F to 1e-30, U
 * from F V, the smallest singular value carries the sign of det F) in place of the reference's approximate four-sweep one.
 * tools/sand_drift_study.py runs the oracle with and without it to measure how far the reference's own SVD residual moves a
 * long plastic run - the yardstick for the HIP engine's distance from the oracle in such runs.  Off by default; the parity
 * tests never switch it on. */
static int orc_exact_svd_enabled = 0;
static inline void orc_svd3_exact(const float* F, float* U, float* S, float* V) {
	double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
	for(int j = 0; j < 3; ++j)
		for(int i = 0; i < 3; ++i) a[i][j] = (double) F[3 * j + i];
	double s[3][3];
	for(int i = 0; i < 3; ++i)
		for(int j = 0; j < 3; ++j) s[i][j] = a[0][i] * a[0][j] + a[1][i] * a[1][j] + a[2][i] * a[2][j];
	for(int sweep = 0; sweep < 30; ++sweep) {
		const double off = fabs(s[0][1]) + fabs(s[0][2]) + fabs(s[1][2]);
		if(off < 1e-30 * (fabs(s[0][0]) + fabs(s[1][1]) + fabs(s[2][2]))) break;
		for(int p = 0; p < 2; ++p)
			for(int q = p + 1; q < 3; ++q) {
				if(s[p][q] == 0.0) continue;
				const double th = (s[q][q] - s[p][p]) / (2.0 * s[p][q]);
				const double t	= (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
				const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
				for(int k = 0; k < 3; ++k) {/* S <- S G */
					const double x = s[k][p], y = s[k][q];
					s[k][p] = c * x - sn * y;
					s[k][q] = sn * x + c * y;
				}
				for(int k = 0; k < 3; ++k) {/* S <- G^T S */
					const double x = s[p][k], y = s[q][k];
					s[p][k] = c * x - sn * y;
					s[q][k] = sn * x + c * y;
				}
				for(int k = 0; k < 3; ++k) {
					const double x = v[k][p], y = v[k][q];
					v[k][p] = c * x - sn * y;
					v[k][q] = sn * x + c * y;
				}
			}
	}
	/* sort descending by eigenvalue, keep V a rotation */
	int ord[3] = {0, 1, 2};
	for(int i = 0; i < 2; ++i)
		for(int j = i + 1; j < 3; ++j)
			if(s[ord[j]][ord[j]] > s[ord[i]][ord[i]]) {
				const int t = ord[i];
				ord[i]		= ord[j];
				ord[j]		= t;
			}
	double vs[3][3];
	for(int k = 0; k < 3; ++k)
		for(int i = 0; i < 3; ++i) vs[i][k] = v[i][ord[k]];
	const double detv = vs[0][0] * (vs[1][1] * vs[2][2] - vs[1][2] * vs[2][1]) - vs[0][1] * (vs[1][0] * vs[2][2] - vs[1][2] * vs[2][0]) + vs[0][2] * (vs[1][0] * vs[2][1] - vs[1][1] * vs[2][0]);
	if(detv < 0)
		for(int i = 0; i < 3; ++i) vs[i][2] = -vs[i][2];
	double b[3][3], sig[3];
	for(int k = 0; k < 3; ++k) {
		for(int i = 0; i < 3; ++i) b[i][k] = a[i][0] * vs[0][k] + a[i][1] * vs[1][k] + a[i][2] * vs[2][k];
		sig[k] = sqrt(b[0][k] * b[0][k] + b[1][k] * b[1][k] + b[2][k] * b[2][k]);
	}
	double u[3][3];
	for(int k = 0; k < 2; ++k)
		for(int i = 0; i < 3; ++i) u[i][k] = b[i][k] / (sig[k] > 1e-300 ? sig[k] : 1e-300);
	u[0][2] = u[1][0] * u[2][1] - u[2][0] * u[1][1];/* third column: cross product, U a rotation */
	u[1][2] = u[2][0] * u[0][1] - u[0][0] * u[2][1];
	u[2][2] = u[0][0] * u[1][1] - u[1][0] * u[0][1];
	sig[2]	= u[0][2] * b[0][2] + u[1][2] * b[1][2] + u[2][2] * b[2][2];/* signed */
	for(int k = 0; k < 3; ++k) {
		S[k] = (float) sig[k];
		for(int i = 0; i < 3; ++i) {
			U[3 * k + i] = (float) u[i][k];
			V[3 * k + i] = (float) vs[i][k];
		}
	}
}

/* math::svd, svd.cuh:27-1123.  F, U, V column-major (F[0]=a11, F[1]=a21, F[3]=a12 ...), as called from
 * compute_stress (constitutive_models.cuh:42). */
static inline void orc_svd3(const float* F, float* U, float* S, float* V) {
	if(orc_exact_svd_enabled) {
		orc_svd3_exact(F, U, S, V);
		return;
	}
	float a11 = F[0], a21 = F[1], a31 = F[2], a12 = F[3], a22 = F[4], a32 = F[5], a13 = F[6], a23 = F[7], a33 = F[8];
	/* normal equations matrix A^T A (svd.cuh:119-157) */
	float s11 = a11 * a11;
	float t	  = a21 * a21;
	s11		  = t + s11;
	t		  = a31 * a31;
	s11		  = t + s11;
	float s21 = a12 * a11;
	t		  = a22 * a21;
	s21		  = t + s21;
	t		  = a32 * a31;
	s21		  = t + s21;
	float s31 = a13 * a11;
	t		  = a23 * a21;
	s31		  = t + s31;
	t		  = a33 * a31;
	s31		  = t + s31;
	float s22 = a12 * a12;
	t		  = a22 * a22;
	s22		  = t + s22;
	t		  = a32 * a32;
	s22		  = t + s22;
	float s32 = a13 * a12;
	t		  = a23 * a22;
	s32		  = t + s32;
	t		  = a33 * a32;
	s32		  = t + s32;
	float s33 = a13 * a13;
	t		  = a23 * a23;
	s33		  = t + s33;
	t		  = a33 * a33;
	s33		  = t + s33;
	float qs = 1.f, qx = 0.f, qy = 0.f, qz = 0.f;
	/* 4 Jacobi sweeps (svd.cuh:167) */
	for(int it = 0; it < 4; ++it) {
		orc_jacobi_conj(&s11, &s21, &s22, &s31, &s32, &s33, &qx, &qy, &qz, &qs); /* (1,2) :168-252 */
		orc_jacobi_conj(&s22, &s32, &s33, &s21, &s31, &s11, &qy, &qz, &qx, &qs); /* (2,3) :257-345 */
		orc_jacobi_conj(&s33, &s31, &s11, &s32, &s21, &s22, &qz, &qx, &qy, &qs); /* (3,1) :351-441 */
	}
	/* normalise quaternion (svd.cuh:475-498) */
	float tmp2 = qs * qs;
	float tmp1 = qx * qx;
	tmp2	   = tmp1 + tmp2;
	tmp1	   = qy * qy;
	tmp2	   = tmp1 + tmp2;
	tmp1	   = qz * qz;
	tmp2	   = tmp1 + tmp2;
	tmp1	   = orc_rsqrt_newton(tmp2);
	qs		   = qs * tmp1;
	qx		   = qx * tmp1;
	qy		   = qy * tmp1;
	qz		   = qz * tmp1;
	/* quaternion -> V (svd.cuh:500-530) */
	tmp1	   = qx * qx;
	tmp2	   = qy * qy;
	float tmp3 = qz * qz;
	float v11  = qs * qs;
	float v22  = v11 - tmp1;
	float v33  = v22 - tmp2;
	v33		   = v33 + tmp3;
	v22		   = v22 + tmp2;
	v22		   = v22 - tmp3;
	v11		   = v11 + tmp1;
	v11		   = v11 - tmp2;
	v11		   = v11 - tmp3;
	tmp1	   = qx + qx;
	tmp2	   = qy + qy;
	tmp3	   = qz + qz;
	float v32  = qs * tmp1;
	float v13  = qs * tmp2;
	float v21  = qs * tmp3;
	tmp1	   = qy * tmp1;
	tmp2	   = qz * tmp2;
	tmp3	   = qx * tmp3;
	float v12  = tmp1 - v21;
	float v23  = tmp2 - v32;
	float v31  = tmp3 - v13;
	v21		   = tmp1 + v21;
	v32		   = tmp2 + v32;
	v13		   = tmp3 + v13;
	/* B = A * V (svd.cuh:532-588), row by row */
	{
		float* rows[3][3] = {{&a11, &a12, &a13}, {&a21, &a22, &a23}, {&a31, &a32, &a33}};
		for(int r = 0; r < 3; ++r) {
			float* x1 = rows[r][0];
			float* x2 = rows[r][1];
			float* x3 = rows[r][2];
			float o2  = *x2;
			float o3  = *x3;
			*x2		  = v12 * *x1;
			*x3		  = v13 * *x1;
			*x1		  = v11 * *x1;
			float u	  = v21 * o2;
			*x1		  = *x1 + u;
			u		  = v31 * o3;
			*x1		  = *x1 + u;
			u		  = v22 * o2;
			*x2		  = *x2 + u;
			u		  = v32 * o3;
			*x2		  = *x2 + u;
			u		  = v23 * o2;
			*x3		  = *x3 + u;
			u		  = v33 * o3;
			*x3		  = *x3 + u;
		}
	}
	/* squared column norms (svd.cuh:594-610) */
	tmp1	   = a11 * a11;
	float tmp4 = a21 * a21;
	tmp1	   = tmp1 + tmp4;
	tmp4	   = a31 * a31;
	tmp1	   = tmp1 + tmp4;
	tmp2	   = a12 * a12;
	tmp4	   = a22 * a22;
	tmp2	   = tmp2 + tmp4;
	tmp4	   = a32 * a32;
	tmp2	   = tmp2 + tmp4;
	tmp3	   = a13 * a13;
	tmp4	   = a23 * a23;
	tmp3	   = tmp3 + tmp4;
	tmp4	   = a33 * a33;
	tmp3	   = tmp3 + tmp4;
	/* sort columns: swap 1-2 (negate col 2), 1-3 (negate col 1), 2-3 (negate col 3); svd.cuh:612-770 */
	{
		int c = tmp1 < tmp2;
		orc_cond_swap(c, &a11, &a12);
		orc_cond_swap(c, &a21, &a22);
		orc_cond_swap(c, &a31, &a32);
		orc_cond_swap(c, &v11, &v12);
		orc_cond_swap(c, &v21, &v22);
		orc_cond_swap(c, &v31, &v32);
		orc_cond_swap(c, &tmp1, &tmp2);
		float neg = 1.f + (c ? -2.f : 0.f);
		a12 *= neg;
		a22 *= neg;
		a32 *= neg;
		v12 *= neg;
		v22 *= neg;
		v32 *= neg;
		c = tmp1 < tmp3;
		orc_cond_swap(c, &a11, &a13);
		orc_cond_swap(c, &a21, &a23);
		orc_cond_swap(c, &a31, &a33);
		orc_cond_swap(c, &v11, &v13);
		orc_cond_swap(c, &v21, &v23);
		orc_cond_swap(c, &v31, &v33);
		orc_cond_swap(c, &tmp1, &tmp3);
		neg = 1.f + (c ? -2.f : 0.f);
		a11 *= neg;
		a21 *= neg;
		a31 *= neg;
		v11 *= neg;
		v21 *= neg;
		v31 *= neg;
		c = tmp2 < tmp3;
		orc_cond_swap(c, &a12, &a13);
		orc_cond_swap(c, &a22, &a23);
		orc_cond_swap(c, &a32, &a33);
		orc_cond_swap(c, &v12, &v13);
		orc_cond_swap(c, &v22, &v23);
		orc_cond_swap(c, &v32, &v33);
		orc_cond_swap(c, &tmp2, &tmp3);
		neg = 1.f + (c ? -2.f : 0.f);
		a13 *= neg;
		a23 *= neg;
		a33 *= neg;
		v13 *= neg;
		v23 *= neg;
		v33 *= neg;
	}
	/* QR of B by three Givens rotations (svd.cuh:772-1090) */
	float u11 = 1.f, u12 = 0.f, u13 = 0.f, u21 = 0.f, u22 = 1.f, u23 = 0.f, u31 = 0.f, u32 = 0.f, u33 = 1.f;
	{
		float* r1[3] = {&a11, &a12, &a13};
		float* r2[3] = {&a21, &a22, &a23};
		float* r3[3] = {&a31, &a32, &a33};
		float* c1[3] = {&u11, &u21, &u31};
		float* c2[3] = {&u12, &u22, &u32};
		float* c3[3] = {&u13, &u23, &u33};
		orc_qr_givens(&a11, &a21, r1, r2, c1, c2); /* zero a21, :786-880 */
		orc_qr_givens(&a11, &a31, r1, r3, c1, c3); /* zero a31, :882-975 */
		orc_qr_givens(&a22, &a32, r2, r3, c2, c3); /* zero a32, :977-1070 */
	}
	U[0] = u11;
	U[1] = u21;
	U[2] = u31;
	U[3] = u12;
	U[4] = u22;
	U[5] = u32;
	U[6] = u13;
	U[7] = u23;
	U[8] = u33;
	V[0] = v11;
	V[1] = v21;
	V[2] = v31;
	V[3] = v12;
	V[4] = v22;
	V[5] = v32;
	V[6] = v13;
	V[7] = v23;
	V[8] = v33;
	S[0] = a11;
	S[1] = a22;
	S[2] = a33;
}

/* PF = P F^T * volume, the common tail of constitutive_models.cuh:63-72 / :324-333 */
static inline void orc_P_Ft_vol(const float* P, const float* F, float volume, float* PF) {
	PF[0] = (P[0] * F[0] + P[3] * F[3] + P[6] * F[6]) * volume;
	PF[1] = (P[1] * F[0] + P[4] * F[3] + P[7] * F[6]) * volume;
	PF[2] = (P[2] * F[0] + P[5] * F[3] + P[8] * F[6]) * volume;
	PF[3] = (P[0] * F[1] + P[3] * F[4] + P[6] * F[7]) * volume;
	PF[4] = (P[1] * F[1] + P[4] * F[4] + P[7] * F[7]) * volume;
	PF[5] = (P[2] * F[1] + P[5] * F[4] + P[8] * F[7]) * volume;
	PF[6] = (P[0] * F[2] + P[3] * F[5] + P[6] * F[8]) * volume;
	PF[7] = (P[1] * F[2] + P[4] * F[5] + P[7] * F[8]) * volume;
	PF[8] = (P[2] * F[2] + P[5] * F[5] + P[8] * F[8]) * volume;
}

/* compute_stress<FIXED_COROTATED>, Projects/GMPM/constitutive_models.cuh:36-73 */
static inline void orc_stress_fixed_corotated(float volume, float mu, float lambda, const float* F, float* PF) {
	float U[9], S[3], V[9];
	orc_svd3(F, U, S, V);
	float J				= S[0] * S[1] * S[2];
	float scaled_mu		= 2.0f * mu;
	float scaled_lambda = lambda * (J - 1.0f);
	float Ph[3];
	Ph[0] = scaled_mu * (S[0] - 1.f) + scaled_lambda * (S[1] * S[2]);
	Ph[1] = scaled_mu * (S[1] - 1.f) + scaled_lambda * (S[0] * S[2]);
	Ph[2] = scaled_mu * (S[2] - 1.f) + scaled_lambda * (S[0] * S[1]);
	float P[9];
	P[0] = Ph[0] * U[0] * V[0] + Ph[1] * U[3] * V[3] + Ph[2] * U[6] * V[6];
	P[1] = Ph[0] * U[1] * V[0] + Ph[1] * U[4] * V[3] + Ph[2] * U[7] * V[6];
	P[2] = Ph[0] * U[2] * V[0] + Ph[1] * U[5] * V[3] + Ph[2] * U[8] * V[6];
	P[3] = Ph[0] * U[0] * V[1] + Ph[1] * U[3] * V[4] + Ph[2] * U[6] * V[7];
	P[4] = Ph[0] * U[1] * V[1] + Ph[1] * U[4] * V[4] + Ph[2] * U[7] * V[7];
	P[5] = Ph[0] * U[2] * V[1] + Ph[1] * U[5] * V[4] + Ph[2] * U[8] * V[7];
	P[6] = Ph[0] * U[0] * V[2] + Ph[1] * U[3] * V[5] + Ph[2] * U[6] * V[8];
	P[7] = Ph[0] * U[1] * V[2] + Ph[1] * U[4] * V[5] + Ph[2] * U[7] * V[8];
	P[8] = Ph[0] * U[2] * V[2] + Ph[1] * U[5] * V[5] + Ph[2] * U[8] * V[8];
	orc_P_Ft_vol(P, F, volume, PF);
}

/* compute_stress<SAND> (Drucker-Prager + StVK-Hencky), constitutive_models.cuh:238-335.  F and *log_jp are
 * rewritten by the plastic projection. */
static inline void orc_stress_sand(float volume, float mu, float lambda, float cohesion, float beta, float yield_surface, int volume_correction, float* F, float* log_jp, float* PF) {
	float U[9], S[3], V[9];
	orc_svd3(F, U, S, V);
	float scaled_mu = 2.0f * mu;
	float epsilon[3], New_S[3] = {0.f, 0.f, 0.f}, New_F[9];
	for(int i = 0; i < 3; i++) {
		float abs_S = S[i] > 0 ? S[i] : -S[i];
		abs_S		= abs_S > 1e-4f ? abs_S : 1e-4f;
		epsilon[i]	= logf(abs_S) - cohesion;
	}
	float sum_epsilon	= epsilon[0] + epsilon[1] + epsilon[2];
	float trace_epsilon = sum_epsilon + *log_jp;
	float epsilon_hat[3];
	for(int i = 0; i < 3; i++) {
		epsilon_hat[i] = epsilon[i] - (trace_epsilon / 3.0f);
	}
	float epsilon_hat_norm = sqrtf(epsilon_hat[0] * epsilon_hat[0] + epsilon_hat[1] * epsilon_hat[1] + epsilon_hat[2] * epsilon_hat[2]);
	if(trace_epsilon >= 0.0f) { /* case II: cone tip */
		New_S[0] = New_S[1] = New_S[2] = expf(cohesion);
		orc_mat_diag_matT(New_F, U, New_S, V);
		for(int i = 0; i < 9; i++) F[i] = New_F[i];
		if(volume_correction) {
			*log_jp = beta * sum_epsilon + *log_jp;
		}
	} else if(mu != 0) {
		*log_jp			  = 0;
		float delta_gamma = epsilon_hat_norm + (3.0f * lambda + scaled_mu) / scaled_mu * trace_epsilon * yield_surface;
		float H[3];
		if(delta_gamma <= 0) { /* case I: inside the cone */
			for(int i = 0; i < 3; i++) H[i] = epsilon[i] + cohesion;
		} else { /* case III: project to the cone surface */
			for(int i = 0; i < 3; i++) H[i] = epsilon[i] - (delta_gamma / epsilon_hat_norm) * epsilon_hat[i] + cohesion;
		}
		for(int i = 0; i < 3; i++) New_S[i] = expf(H[i]);
		orc_mat_diag_matT(New_F, U, New_S, V);
		for(int i = 0; i < 9; i++) F[i] = New_F[i];
	}
	float New_S_log[3] = {logf(New_S[0]), logf(New_S[1]), logf(New_S[2])};
	float P_hat[3];
	float trace_log_S = New_S_log[0] + New_S_log[1] + New_S_log[2];
	for(int i = 0; i < 3; i++) {
		P_hat[i] = (scaled_mu * New_S_log[i] + lambda * trace_log_S) / New_S[i];
	}
	float P[9];
	orc_mat_diag_matT(P, U, P_hat, V);
	orc_P_Ft_vol(P, F, volume, PF);
}

/* compute_stress<NACC>, constitutive_models.cuh:77-234 (USE_JOSH_FRACTURE_PAPER = 1, :11) */
static inline void orc_stress_nacc(float volume, float mu, float lambda, float bm, float xi, float beta, float msqr, int hardening_on, float* F, float* log_jp, float* PF) {
	(void) lambda;
	float U[9], S[3], V[9];
	orc_svd3(F, U, S, V);
	float p0	= bm * (0.00001f + sinhf(xi * (-*log_jp > 0 ? -*log_jp : 0)));
	float p_min = -beta * p0;
	float Je_trial = S[0] * S[1] * S[2];
	float B_hat_trial[3]		   = {S[0] * S[0], S[1] * S[1], S[2] * S[2]};
	float trace_B_hat_trial_divdim = (B_hat_trial[0] + B_hat_trial[1] + B_hat_trial[2]) / 3.f;
	float J_power_neg_2_d_mulmu	   = mu * powf(Je_trial, -2.f / 3.f);
	float s_hat_trial[3]		   = {J_power_neg_2_d_mulmu * (B_hat_trial[0] - trace_B_hat_trial_divdim), J_power_neg_2_d_mulmu * (B_hat_trial[1] - trace_B_hat_trial_divdim), J_power_neg_2_d_mulmu * (B_hat_trial[2] - trace_B_hat_trial_divdim)};
	float psi_kappa_partial_J	   = bm * 0.5f * (Je_trial - 1.f / Je_trial);
	float p_trial				   = -psi_kappa_partial_J * Je_trial;
	float y_s_half_coeff		   = 3.f / 2.f * (1 + 2.f * beta);
	float y_p_half				   = (msqr * (p_trial - p_min) * (p_trial - p0));
	float s_hat_trial_sqrnorm	   = s_hat_trial[0] * s_hat_trial[0] + s_hat_trial[1] * s_hat_trial[1] + s_hat_trial[2] * s_hat_trial[2];
	float y						   = (y_s_half_coeff * s_hat_trial_sqrnorm) + y_p_half;
	float New_F[9];
	if(p_trial > p0) { /* case 1 */
		float Je_new = sqrtf(-2.f * p0 / bm + 1.f);
		S[0] = S[1] = S[2] = powf(Je_new, 1.f / 3.f);
		orc_mat_diag_matT(New_F, U, S, V);
		for(int i = 0; i < 9; i++) F[i] = New_F[i];
		if(hardening_on) *log_jp += logf(Je_trial / Je_new);
	} else if(p_trial < p_min) { /* case 2 */
		float Je_new = sqrtf(-2.f * p_min / bm + 1.f);
		S[0] = S[1] = S[2] = powf(Je_new, 1.f / 3.f);
		orc_mat_diag_matT(New_F, U, S, V);
		for(int i = 0; i < 9; i++) F[i] = New_F[i];
		if(hardening_on) *log_jp += logf(Je_trial / Je_new);
	} else { /* case 3 */
		if(y >= 1e-4) {
			float B_s_coeff = powf(Je_trial, 2.f / 3.f) / mu * sqrtf(-y_p_half / y_s_half_coeff) / sqrtf(s_hat_trial_sqrnorm);
			for(int i = 0; i < 3; i++) S[i] = sqrtf(s_hat_trial[i] * B_s_coeff + trace_B_hat_trial_divdim);
			orc_mat_diag_matT(New_F, U, S, V);
			for(int i = 0; i < 9; i++) F[i] = New_F[i];
			if(hardening_on && p0 > 1e-4 && p_trial < p0 - 1e-4 && p_trial > 1e-4 + p_min) {
				float p_center		 = (1.0f - beta) * p0 / 2;
				float q_trial		 = sqrtf(3.f / 2.f * s_hat_trial_sqrnorm);
				float direction[2]	 = {p_center - p_trial, -q_trial};
				float direction_norm = sqrtf(direction[0] * direction[0] + direction[1] * direction[1]);
				direction[0] /= direction_norm;
				direction[1] /= direction_norm;
				float C	 = msqr * (p_center - p_min) * (p_center - p0);
				float B	 = msqr * direction[0] * (2 * p_center - p0 - p_min);
				float A	 = msqr * direction[0] * direction[0] + (1 + 2 * beta) * direction[1] * direction[1];
				float l1 = (-B + sqrtf(B * B - 4 * A * C)) / (2 * A);
				float l2 = (-B - sqrtf(B * B - 4 * A * C)) / (2 * A);
				float p1 = p_center + l1 * direction[0];
				float p2 = p_center + l2 * direction[0];
				float p_fake	  = (p_trial - p_center) * (p1 - p_center) > 0 ? p1 : p2;
				float tmp_Je_sqr  = (-2 * p_fake / bm + 1);
				float Je_new_fake = sqrtf(tmp_Je_sqr > 0 ? tmp_Je_sqr : -tmp_Je_sqr);
				if(Je_new_fake > 1e-4) *log_jp += logf(Je_trial / Je_new_fake);
			}
		}
	}
	float J = S[0] * S[1] * S[2];
	float b[9], b_dev[9];
	orc_mat_matT(F, b);
	orc_deviatoric(b, b_dev);
	float dev_b_coeff = mu * powf(J, -2.f / 3.f);
	float i_coeff	  = bm * .5f * ((J * J - 1.f) * 0.5f - logf(J));
	PF[0]			  = (dev_b_coeff * b_dev[0] + i_coeff) * volume;
	PF[1]			  = (dev_b_coeff * b_dev[1]) * volume;
	PF[2]			  = (dev_b_coeff * b_dev[2]) * volume;
	PF[3]			  = (dev_b_coeff * b_dev[3]) * volume;
	PF[4]			  = (dev_b_coeff * b_dev[4] + i_coeff) * volume;
	PF[5]			  = (dev_b_coeff * b_dev[5]) * volume;
	PF[6]			  = (dev_b_coeff * b_dev[6]) * volume;
	PF[7]			  = (dev_b_coeff * b_dev[7]) * volume;
	PF[8]			  = (dev_b_coeff * b_dev[8] + i_coeff) * volume;
}

/* J-fluid update + stress, inline in g2p2g: Projects/GMPM/mgmpm_kernels.cuh:476-505.
 * A = un-normalised APIC matrix (column-major), returns the new J. */
static inline float orc_jfluid(float J, const float* A, float dt, float d_inv, float volume, float bulk, float gamma, float viscosity, float* contrib) {
	J += (A[0] + A[4] + A[8]) * dt * d_inv * J;
	if(J < 0.1) J = 0.1; /* double literal in the reference: compares float against 0.1 as double */
	float voln	   = J * volume;
	float pressure = bulk * (powf(J, -gamma) - 1.f);
	contrib[0]	   = ((A[0] + A[0]) * d_inv * viscosity - pressure) * voln;
	contrib[1]	   = (A[1] + A[3]) * d_inv * viscosity * voln;
	contrib[2]	   = (A[2] + A[6]) * d_inv * viscosity * voln;
	contrib[3]	   = (A[3] + A[1]) * d_inv * viscosity * voln;
	contrib[4]	   = ((A[4] + A[4]) * d_inv * viscosity - pressure) * voln;
	contrib[5]	   = (A[5] + A[7]) * d_inv * viscosity * voln;
	contrib[6]	   = (A[6] + A[2]) * d_inv * viscosity * voln;
	contrib[7]	   = (A[7] + A[5]) * d_inv * viscosity * voln;
	contrib[8]	   = ((A[8] + A[8]) * d_inv * viscosity - pressure) * voln;
	return J;
}

#endif
