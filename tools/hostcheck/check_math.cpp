// x86 build of the device math (one "lane" at a time): same entry point shape as mpm_test_stress / mpm_test_svd.
#include "mpm_device_math.hpp"
using namespace mpm;
extern "C" int host_test_stress(int material, const MaterialConst* mc, const float* Fin, const float* ljin, size_t n, float* out19) {
	for(size_t i = 0; i < n; ++i) {
		float F[9], PF[9];
		for(int d = 0; d < 9; ++d) F[d] = Fin[9 * i + d];
		float lj = ljin ? ljin[i] : 0.f;
		if(material == 1) stress_fixed_corotated(*mc, F, PF);
		else if(material == 2) stress_sand(*mc, F, lj, PF);
		else stress_nacc(*mc, F, lj, PF);
		for(int d = 0; d < 9; ++d) out19[19 * i + d] = F[d];
		for(int d = 0; d < 9; ++d) out19[19 * i + 9 + d] = PF[d];
		out19[19 * i + 18] = lj;
	}
	return 0;
}
extern "C" int host_test_eig(const float* Fin, size_t n, float* out12) {
	NoHook nh;
	for(size_t i = 0; i < n; ++i) {
		float F[9], lam[3], U[9];
		for(int d = 0; d < 9; ++d) F[d] = Fin[9 * i + d];
		bool und;
		sym_eig3<0>(F, lam, U, nh, und);
		for(int d = 0; d < 9; ++d) out12[12 * i + d] = U[d];
		for(int d = 0; d < 3; ++d) out12[12 * i + 9 + d] = lam[d];
	}
	return 0;
}
extern "C" float host_jfluid(const MaterialConst* mc, float J, const float* A, float dt, float d_inv, float* contrib) {
	float a[9], c[9];
	for(int d = 0; d < 9; ++d) a[d] = A[d];
	const float r = stress_jfluid(*mc, J, a, dt, d_inv, c);
	for(int d = 0; d < 9; ++d) contrib[d] = c[d];
	return r;
}
