// Ground truth for the VALU issue rate, timed with HIP events (not s_memtime): every SIMD of the chip runs W waves of independent
// v_fma_f32 / v_pk_fma_f32 / v_mul_f32 chains (8 accumulators per lane); prints wave-instructions per SIMD per nominal 2.4 GHz cycle.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template<int KIND>
__global__ __launch_bounds__(64) void rate(int n, float* sink) {
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
	const float b = 1.0001f, c = 0.5f;
	const v2f bb = {b, b}, cc = {c, c};
	for(int i = 0; i < n; ++i) {
		if constexpr(KIND == 0) {
			asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
						 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
						 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
		} else if constexpr(KIND == 1) {
			asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
						 "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
						 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(bb), "v"(cc));
		} else if constexpr(KIND == 2) {
			asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
						 "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
						 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
		} else if constexpr(KIND == 3) {
			asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
						 "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
						 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
		} else if constexpr(KIND == 4) {// VOP2 fused multiply-add (dst is the addend)
			asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
						 "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
						 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
		} else if constexpr(KIND == 5) {
			asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
						 "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
						 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(bb));
		} else if constexpr(KIND == 6) {
			asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
						 "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
						 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
		} else if constexpr(KIND == 7) {// VOP3 integer, three sources
			asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n"
						 "v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9\n"
						 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
		} else {// a mix like the kernel's: 4 VOP2, 2 packed, 2 VOP3
			asm volatile("v_mul_f32 %0, %0, %8\n v_pk_fma_f32 %10, %10, %11, %12\n v_fmac_f32 %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
						 "v_add_f32 %3, %3, %8\n v_pk_fma_f32 %13, %13, %11, %12\n v_mul_f32 %4, %4, %8\n v_fma_f32 %5, %5, %8, %9\n"
						 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(p0), "v"(bb), "v"(cc), "v"(p1));
		}
	}
	sink[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template<int KIND>
static void run(const char* name, float* sink) {
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	const int n = 200000;
	for(int W: {1, 2, 4, 8}) {
		const int blocks = 256 * 4 * W;// one wave per workgroup: W waves per SIMD when they spread evenly
		rate<KIND><<<blocks, 64>>>(1000, sink);
		hipDeviceSynchronize();
		hipEventRecord(a);
		rate<KIND><<<blocks, 64>>>(n, sink);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms;
		hipEventElapsedTime(&ms, a, b);
		const double per_simd = 8.0 * n * W;// wave-instructions each SIMD executed
		printf("%-14s %d waves per SIMD: %.3f ms, %.2f cycles (2.4 GHz) per wave-instruction per SIMD\n", name, W, ms, ms * 1e-3 * 2.4e9 / per_simd);
	}
}
int main() {
	float* sink;
	hipMalloc(&sink, sizeof(float) * 64 * 256 * 4 * 8);
	run<0>("v_fma_f32", sink);
	run<1>("v_pk_fma_f32", sink);
	run<2>("v_mul_f32", sink);
	run<3>("v_rsq_f32", sink);
	run<4>("v_fmac_f32", sink);
	run<5>("v_pk_mul_f32", sink);
	run<6>("v_mov_b32", sink);
	run<7>("v_add3_u32", sink);
	run<8>("mix 4/2/2", sink);
	return 0;
}
