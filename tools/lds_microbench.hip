// Micro-benchmark of the LDS operations the P2G reduction could be built from (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o lds_microbench lds_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITERS = 256;
constexpr int OPS	= 32;// LDS ops per iteration

template<int MODE>
__global__ __launch_bounds__(256) void k(float* out, int stride, int same) {
	__shared__ float s[8192];
	const int tid = threadIdx.x;
	for(int i = tid; i < 8192; i += 256) s[i] = 0.f;
	__syncthreads();
	const int lane = tid & 63;
	// address pattern: lane -> distinct word (stride), or groups of `same` lanes share a word
	int base = ((lane / same) * stride + (tid >> 6) * 1024) & 8191;
	float v	 = (float) tid;
	float acc = 0.f;
	for(int it = 0; it < ITERS; ++it) {
#pragma unroll
		for(int o = 0; o < OPS; ++o) {
			const int a = (base + o * 67) & 8191;
			if constexpr(MODE == 0) {
				atomicAdd(&s[a], v);// ds_add_f32
			} else if constexpr(MODE == 1) {
				atomicAdd((int*) &s[a], tid);// ds_add_u32
			} else if constexpr(MODE == 2) {
				s[a] = s[a] + v;// ds_read_b32 + v_add + ds_write_b32
			} else if constexpr(MODE == 3) {
				acc += s[a];// ds_read_b32 only
			} else if constexpr(MODE == 4) {
				s[a] = v;// ds_write_b32 only
			} else if constexpr(MODE == 5) {
				acc += atomicAdd(&s[a], v);// ds_add_rtn_f32
			} else if constexpr(MODE == 6) {
				float4* p = (float4*) &s[(a & ~3)];
				float4 x  = *p;
				x.x += v; x.y += v; x.z += v; x.w += v;
				*p = x;// b128 RMW
			} else if constexpr(MODE == 7) {
				float2* p = (float2*) &s[(a & ~1)];
				float2 x  = *p;
				x.x += v; x.y += v;
				*p = x;// b64 RMW
			}
		}
		v += 1.f;
	}
	__syncthreads();
	out[blockIdx.x * 256 + tid] = s[tid] + acc;
}

template<int MODE>
void run(const char* name, int stride, int same) {
	float* d;
	hipMalloc(&d, 4096 * 256 * 4);
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	const int blocks = 256 * 8;
	k<MODE><<<blocks, 256>>>(d, stride, same);
	hipEventRecord(a);
	k<MODE><<<blocks, 256>>>(d, stride, same);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	// wave-instructions per CU: blocks/256 CUs * 4 waves * ITERS * OPS
	const double winst = (double) blocks / 256 * 4 * ITERS * OPS;
	const double cyc   = ms * 1e-3 * 2.4e9;
	printf("%-28s stride=%3d same=%2d : %8.3f ms  -> %7.2f cycles per wave-instruction per CU\n", name, stride, same, ms, cyc / winst);
	hipFree(d);
}

int main() {
	for(int same: {1, 2, 8}) {
		run<0>("ds_add_f32", 1, same);
		run<1>("ds_add_u32", 1, same);
		run<5>("ds_add_rtn_f32", 1, same);
	}
	run<0>("ds_add_f32", 9, 1);
	run<2>("rmw b32 (read,add,write)", 1, 1);
	run<2>("rmw b32 (read,add,write)", 9, 1);
	run<7>("rmw b64", 2, 1);
	run<6>("rmw b128", 4, 1);
	run<3>("ds_read_b32", 1, 1);
	run<4>("ds_write_b32", 1, 1);
	return 0;
}
