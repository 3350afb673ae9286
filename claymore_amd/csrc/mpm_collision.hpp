// mpm_collision.hpp — level-set collision object of the MGSP grid update.
//
// Device side of Projects/MGSP/boundary_condition.cuh:25-250 (SignedDistanceGrid::detect_and_resolve_collision and
// helpers).  The signed distance and its gradient live in one float4 {sdis, gx, gy, gz} per grid node of the whole
// domain (N^3 nodes, node (i,j,k) at (i N + j) N + k): one 16-byte load per corner of the trilinear stencil instead of
// four scalar loads from four SoA channels of a 4x4x4-blocked field.  The arithmetic keeps the reference's quirks
// (the "cross product" with plus signs, MatrixUtils.h:53-58; the early return of SEPARATE with a zero normal).
#pragma once
#include "mpm_device_math.hpp"

namespace mpm {

struct CollisionObject {// mpm_collision_object + the field
	int type;
	float friction, scale, dsdt;
	float trans[3], trans_vel[3], omega[3];
	float rot[9];// element (i, j) at [3 i + j] (the reference's vec3x3), handed to the column-major helpers as raw arrays
	float time;
	const float4* field;
};

// rot_angle_to_matrix (:68-91)
MPM_DEV void col_rot_angle_to_matrix(float omega, int dim, float (&res)[9]) {
#pragma unroll
	for(int i = 0; i < 9; ++i) res[i] = 0.f;
	const float c = cosf(omega), s = sinf(omega);
	if(dim == 0) {
		res[0] = 1.f;
		res[4] = res[8] = c;
		res[7]			= s;
		res[5]			= -s;
	} else if(dim == 1) {
		res[4] = 1.f;
		res[0] = res[8] = c;
		res[2]			= s;
		res[6]			= -s;
	} else {
		res[8] = 1.f;
		res[0] = res[4] = c;
		res[3]			= s;
		res[1]			= -s;
	}
}
MPM_DEV void col_cross(float (&out)[3], const float (&a)[3], const float (&b)[3]) {// (sic) plus signs
	out[0] = a[1] * b[2] + a[2] * b[1];
	out[1] = a[2] * b[0] + a[0] * b[2];
	out[2] = a[0] * b[1] + a[1] * b[0];
}

// detect_and_resolve_collision (:164-248); node = integer node coordinates; bc_lo / bc_hi = query_sdf's domain box (:141-146)
MPM_DEV void collision_resolve(const CollisionObject& o, const int (&node)[3], float dx, int N, float bc_lo, float bc_hi, float (&vel)[3]) {
	const float t = o.time;
	float xmt[3], x0[3], x[3];
#pragma unroll
	for(int d = 0; d < 3; ++d) xmt[d] = (float) node[d] * dx - (o.trans[d] + o.trans_vel[d] * t);
	float rot[9];
#pragma unroll
	for(int i = 0; i < 9; ++i) rot[i] = o.rot[i];
	const float inv = 1.f / (1.f + o.dsdt * t);
#pragma unroll
	for(int d = 0; d < 3; ++d) x0[d] = xmt[d] * inv;
	if(t != 0.f) {// at t = 0 the three factors are identities (the only case the reference exercises)
#pragma unroll
		for(int dim = 0; dim < 3; ++dim) {
			float tmp[9], prev[9];
			col_rot_angle_to_matrix(o.omega[dim] * t, dim, tmp);
#pragma unroll
			for(int i = 0; i < 9; ++i) prev[i] = rot[i];
			matmul3(prev, tmp, rot);
		}
	}
	x[0] = rot[0] * x0[0] + rot[1] * x0[1] + rot[2] * x0[2];// mat_t_mul_vec_3d
	x[1] = rot[3] * x0[0] + rot[4] * x0[1] + rot[5] * x0[2];
	x[2] = rot[6] * x0[0] + rot[7] * x0[1] + rot[8] * x0[2];
#pragma unroll
	for(int d = 0; d < 3; ++d) x[d] = x[d] * o.scale + o.trans[d];
	// query_sdf
	if(x[0] < bc_lo || x[0] >= bc_hi || x[1] < bc_lo || x[1] >= bc_hi || x[2] < bc_lo || x[2] >= bc_hi) return;
	int cid[3];
	float w1[3][2];
#pragma unroll
	for(int d = 0; d < 3; ++d) {
		cid[d]			   = (int) (x[d] / dx);
		const float dis_lb = x[d] - ((float) cid[d] * dx);
		w1[d][0]		   = 1.f - dis_lb / dx;
		w1[d][1]		   = dis_lb / dx;
	}
	float sdis = 0.f, n[3] = {0.f, 0.f, 0.f};
#pragma unroll
	for(int i = 0; i < 2; ++i)
#pragma unroll
		for(int j = 0; j < 2; ++j)
#pragma unroll
			for(int k = 0; k < 2; ++k) {
				const float w  = w1[0][i] * w1[1][j] * w1[2][k];
				const float4 v = o.field[((size_t) (cid[0] + i) * N + (size_t) (cid[1] + j)) * N + (size_t) (cid[2] + k)];
				sdis += w * v.x;
				n[0] += w * v.y;
				n[1] += w * v.z;
				n[2] += w * v.w;
			}
	const float nn = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
#pragma unroll
	for(int d = 0; d < 3; ++d) n[d] /= nn;
	if(!(sdis <= 0.f)) return;
	// object velocity in deformation space (:197-204)
	float v_obj[3], radius[3], mat_vel[3];
	col_cross(v_obj, o.omega, xmt);
#pragma unroll
	for(int d = 0; d < 3; ++d) {
		v_obj[d] += xmt[d] * (o.dsdt / o.scale);
		radius[d] = x[d] - o.trans[d];
	}
	col_cross(mat_vel, o.omega, radius);
#pragma unroll
	for(int d = 0; d < 3; ++d) mat_vel[d] += o.trans_vel[d];
	const float rv0 = rot[0] * mat_vel[0] + rot[3] * mat_vel[1] + rot[6] * mat_vel[2];
	const float rv1 = rot[1] * mat_vel[0] + rot[4] * mat_vel[1] + rot[7] * mat_vel[2];
	const float rv2 = rot[2] * mat_vel[0] + rot[5] * mat_vel[1] + rot[8] * mat_vel[2];
	v_obj[0] += rv0 * o.scale + o.trans_vel[0];
	v_obj[1] += rv1 * o.scale + o.trans_vel[1];
	v_obj[2] += rv2 * o.scale + o.trans_vel[2];
#pragma unroll
	for(int d = 0; d < 3; ++d) vel[d] -= v_obj[d];
	if(o.type == 0) {// STICKY
		vel[0] = vel[1] = vel[2] = 0.f;
	} else {
		if(o.type == 2 && n[0] == 0.0f && n[1] == 0.0f && n[2] == 0.0f) {
			vel[0] = vel[1] = vel[2] = 0.f;
			return;// (:227-230) without adding the object velocity back
		}
		float nr[3];
		nr[0] = rot[0] * n[0] + rot[3] * n[1] + rot[6] * n[2];
		nr[1] = rot[1] * n[0] + rot[4] * n[1] + rot[7] * n[2];
		nr[2] = rot[2] * n[0] + rot[5] * n[1] + rot[8] * n[2];
		const float v_dot_n = nr[0] * vel[0] + nr[1] * vel[1] + nr[2] * vel[2];
		const bool project	= o.type == 1 || v_dot_n < 0.f;// SLIP always removes the normal part, SEPARATE only when approaching
		if(project) {
#pragma unroll
			for(int d = 0; d < 3; ++d) vel[d] -= nr[d] * v_dot_n;
			const bool fric = o.type == 1 ? (o.friction > 0.0f && v_dot_n < 0.f) : (o.friction != 0.f);
			if(fric) {
				const float vel_norm = sqrtf(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]);
				if(-v_dot_n * o.friction < vel_norm) {
#pragma unroll
					for(int d = 0; d < 3; ++d) vel[d] += vel[d] / vel_norm * (v_dot_n * o.friction);
				} else {
					vel[0] = vel[1] = vel[2] = 0.f;
				}
			}
		}
	}
#pragma unroll
	for(int d = 0; d < 3; ++d) vel[d] += v_obj[d];
}

}// namespace mpm
