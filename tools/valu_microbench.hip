// VALU issue-rate micro-benchmark (gfx950): cycles per wave64 instruction per SIMD for v_fma_f32, v_pk_fma_f32,
// v_mul, v_rsq, v_cndmask at 1, 2 and 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_microbench valu_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));
constexpr int ITERS = 65536;

template<int MODE>
__global__ void k(float* out, float a, float b) {
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	float2_ p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
	float2_ pa = {a, a}, pb = {b, b};
	for(int i = 0; i < ITERS; ++i) {
		if constexpr(MODE == 0) {
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
			F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
		} else if constexpr(MODE == 1) {
#define F(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(pa), "v"(pb));
			F(p0) F(p1) F(p2) F(p3) F(p4) F(p5) F(p6) F(p7)
#undef F
		} else if constexpr(MODE == 2) {
#define F(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
			F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
		} else if constexpr(MODE == 3) {
#define F(x) asm volatile("v_rsq_f32 %0, %0" : "+v"(x));
			F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
		} else if constexpr(MODE == 4) {
#define F(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a));
			F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
		} else if constexpr(MODE == 5) {
#define F(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
			F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
		} else if constexpr(MODE == 6) {// dependent chain on one register
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
			F(x0) F(x0) F(x0) F(x0) F(x0) F(x0) F(x0) F(x0)
#undef F
		} else if constexpr(MODE == 7) {
#define F(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(pa));
			F(p0) F(p1) F(p2) F(p3) F(p4) F(p5) F(p6) F(p7)
#undef F
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
}

template<int MODE>
void run(const char* name, int waves_per_simd) {
	float* d;
	(void) hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
	hipEvent_t a, b;
	(void) hipEventCreate(&a);
	(void) hipEventCreate(&b);
	const int threads = 64 * 4 * waves_per_simd;// one workgroup per CU
	k<MODE><<<256, threads>>>(d, 1.0001f, 0.5f);
	(void) hipEventRecord(a);
	k<MODE><<<256, threads>>>(d, 1.0001f, 0.5f);
	(void) hipEventRecord(b);
	(void) hipEventSynchronize(b);
	float ms;
	(void) hipEventElapsedTime(&ms, a, b);
	const double inst_per_simd = (double) waves_per_simd * ITERS * 8;
	printf("%-22s waves/SIMD=%d : %7.3f ms -> %5.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / inst_per_simd);
	(void) hipFree(d);
}

int main() {
	for(int w: {2, 4}) {
		run<0>("v_fma_f32", w);
		run<5>("v_fmac_f32", w);
		run<2>("v_mul_f32", w);
		run<1>("v_pk_fma_f32", w);
		run<7>("v_pk_mul_f32", w);
		run<3>("v_rsq_f32", w);
		run<4>("v_cndmask_b32", w);
		run<6>("v_fma_f32 dependent", w);
	}
	return 0;
}
