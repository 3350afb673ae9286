// What does one s_memtime tick mean?  A single wave spins for a fixed number of s_memtime ticks; the host times the kernel
// with HIP events: ticks per second = the counter's frequency (compare with the shader clock rocm-smi reports).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
	const unsigned long long t0 = __builtin_readcyclecounter();// s_memtime
	unsigned long long t = t0, w0 = wall_clock64();
	while(t - t0 < ticks) t = __builtin_readcyclecounter();
	out[0] = t - t0;
	out[1] = wall_clock64() - w0;// constant 100 MHz counter (s_memrealtime)
}
// dependent / independent v_fma chains timed in both counters
__global__ void fma_chain(int n, float* sink, unsigned long long* out) {
	float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f;
	const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
	for(int i = 0; i < n; ++i) {
		a = __builtin_fmaf(a, b, c);
		a = __builtin_fmaf(a, b, d);
		a = __builtin_fmaf(a, b, e);
		a = __builtin_fmaf(a, b, c);
	}
	out[0] = __builtin_readcyclecounter() - t0;
	out[1] = wall_clock64() - w0;
	sink[threadIdx.x] = a;
}
int main() {
	unsigned long long* d;
	float* s;
	hipMalloc(&d, 16);
	hipMalloc(&s, 1024);
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	for(unsigned long long ticks: {100000000ull, 400000000ull}) {
		spin<<<1, 64>>>(1000, d);
		hipDeviceSynchronize();
		hipEventRecord(a);
		spin<<<1, 64>>>(ticks, d);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms;
		hipEventElapsedTime(&ms, a, b);
		unsigned long long h[2];
		hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
		printf("spin %llu s_memtime ticks: %.3f ms by HIP events -> s_memtime runs at %.1f MHz; wall_clock64 delta %llu -> %.1f MHz\n", h[0], ms, h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3));
	}
	const int n = 1 << 22;
	fma_chain<<<1, 64>>>(1000, s, d);
	hipDeviceSynchronize();
	hipEventRecord(a);
	fma_chain<<<1, 64>>>(n, s, d);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	unsigned long long h[2];
	hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
	printf("dependent v_fma chain, one wave: %d x 4 instructions in %.3f ms = %.2f ns per instruction = %.2f s_memtime ticks per instruction\n", n, ms, ms * 1e6 / (4.0 * n), (double) h[0] / (4.0 * n));
	int clk = 0;
	hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
	printf("hipDeviceAttributeClockRate %d kHz -> %.2f shader cycles per dependent instruction\n", clk, ms * 1e-3 / (4.0 * n) * clk * 1e3);
	return 0;
}
