// LDS throughput by access width under the kernel's occupancy (12 single-wave workgroups per CU, all issuing LDS operations): periods of the
// 2.4 GHz clock per wave-instruction per CU for ds_read / ds_write of 4, 8, 12 and 16 bytes per lane, lanes 16 bytes apart (node = lane).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_width_probe lds_width_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITERS = 2048;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v3f __attribute__((ext_vector_type(3)));
typedef float v2f __attribute__((ext_vector_type(2)));

template<int BYTES, bool WRITE>
__global__ __launch_bounds__(64) void probe(float* out, int stride) {
	__shared__ v4f s[1024];
	const int lane = threadIdx.x;
	for(int i = lane; i < 1024; i += 64) s[i] = (v4f) {0.f, 0.f, 0.f, 0.f};
	__syncthreads();
	int n	  = (lane * stride) & 1023;
	float acc = 0.f;
#pragma unroll 1
	for(int it = 0; it < ITERS; ++it) {
#pragma unroll
		for(int o = 0; o < 8; ++o) {
			__asm__ volatile("" : "+v"(n));
			char* p = reinterpret_cast<char*>(&s[n]);
			if constexpr(BYTES == 16) {
				if constexpr(WRITE) *reinterpret_cast<v4f*>(p) = (v4f) {acc, 1.f, 2.f, 3.f};
				else {
					const v4f x = *reinterpret_cast<v4f*>(p);
					acc += x.x + x.y + x.z + x.w;
				}
			} else if constexpr(BYTES == 12) {
				if constexpr(WRITE) *reinterpret_cast<v3f*>(p) = (v3f) {acc, 1.f, 2.f};
				else {
					const v3f x = *reinterpret_cast<v3f*>(p);
					acc += x.x + x.y + x.z;
				}
			} else if constexpr(BYTES == 8) {
				if constexpr(WRITE) *reinterpret_cast<v2f*>(p) = (v2f) {acc, 1.f};
				else {
					const v2f x = *reinterpret_cast<v2f*>(p);
					acc += x.x + x.y;
				}
			} else {
				if constexpr(WRITE) *reinterpret_cast<float*>(p) = acc;
				else acc += *reinterpret_cast<float*>(p);
			}
			__asm__ volatile("" ::: "memory");
		}
	}
	__syncthreads();
	out[blockIdx.x * 64 + lane] = acc + s[lane].x;
}

template<int BYTES, bool WRITE>
static void run(float* d_out, int stride) {
	const int blocks = 256 * 12;
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	probe<BYTES, WRITE><<<blocks, 64>>>(d_out, stride);
	hipEventRecord(a);
	probe<BYTES, WRITE><<<blocks, 64>>>(d_out, stride);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	const double per = ms * 1e-3 * 2.4e9 / (12.0 * ITERS * 8);
	printf("%s b%-3d lanes %2d B apart: %6.2f periods per wave-instruction per CU  (%5.1f B per period)\n", WRITE ? "ds_write" : "ds_read ", BYTES * 8, stride * 16, per, 64.0 * BYTES / per);
}

int main() {
	float* d_out;
	hipMalloc(&d_out, 256 * 12 * 64 * sizeof(float));
	for(int stride: {1}) {
		run<4, false>(d_out, stride);
		run<8, false>(d_out, stride);
		run<12, false>(d_out, stride);
		run<16, false>(d_out, stride);
		run<4, true>(d_out, stride);
		run<8, true>(d_out, stride);
		run<12, true>(d_out, stride);
		run<16, true>(d_out, stride);
	}
	return 0;
}
