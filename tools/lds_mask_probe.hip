// Does the cost of an LDS read-modify-write pair depend on how many lanes are active?  (12 single-wave workgroups per CU, node = lane, b128.)
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_mask_probe lds_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 2048;
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void probe(float* out, unsigned long long mask) {
	__shared__ v4f s[1024];
	const int lane = threadIdx.x;
	for(int i = lane; i < 1024; i += 64) s[i] = (v4f) {0.f, 0.f, 0.f, 0.f};
	__syncthreads();
	int n = lane;
	v4f v = {(float) lane, 1.f, 2.f, 3.f};
	if((mask >> lane) & 1ull) {
#pragma unroll 1
		for(int it = 0; it < ITERS; ++it) {
#pragma unroll
			for(int o = 0; o < 8; ++o) {
				__asm__ volatile("" : "+v"(n));
				v4f x = s[n];
				x += v;
				s[n] = x;
				__asm__ volatile("" ::: "memory");
			}
		}
	}
	__syncthreads();
	out[blockIdx.x * 64 + lane] = s[lane].x;
}
int main() {
	float* d;
	hipMalloc(&d, 256 * 12 * 64 * 4);
	struct { const char* name; unsigned long long m; } cases[] = {
		{"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"even lanes (32)", 0x5555555555555555ull}, {"lanes 0-15", 0xffffull}, {"every 4th lane (16)", 0x1111111111111111ull},
		{"lanes 0-7", 0xffull}, {"every 8th lane (8)", 0x0101010101010101ull}, {"lanes 0-3", 0xfull}, {"every 16th lane (4)", 0x0001000100010001ull}, {"lane 0", 1ull}};
	for(auto& c: cases) {
		hipEvent_t a, b;
		hipEventCreate(&a);
		hipEventCreate(&b);
		probe<<<256 * 12, 64>>>(d, c.m);
		hipEventRecord(a);
		probe<<<256 * 12, 64>>>(d, c.m);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms;
		hipEventElapsedTime(&ms, a, b);
		printf("%-24s rmw pair %6.2f periods per CU\n", c.name, ms * 1e-3 * 2.4e9 / (12.0 * ITERS * 8));
	}
	return 0;
}
