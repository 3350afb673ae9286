// Energy cost model: the G2P2G launch runs the chip at its power limit (1.3 kW, sclk ~1.97 GHz of 2.4: profiles/r03_power.txt), so what
// bounds the kernel is energy per particle, not any one unit's throughput.  This tool runs ONE instruction class at a time on every CU for
// a fixed wall time and prints the achieved rate; tools/gpu_energy.sh samples rocm-smi beside it: (P - P_idle) / rate = energy per operation.
//   usage: energy_microbench <mode> <seconds>
//   modes: 0 idle waves (s_sleep)   1 v_fmac_f32   2 v_pk_fma_f32   3 v_fma_f32 (VOP3)   4 ds_read_b128   5 ds_write_b128
//          6 ds read-modify-write b128 pair   7 global_load_dwordx4 stream (4 GiB, HBM)   8 global_store_dwordx4 stream   9 v_mov_b32
//          10 ds_write_b64 x 2   11 ds_write2_b64   12 ds_read2_b64   13 ds_write_b96
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template<int MODE>
__global__ __launch_bounds__(64) void burn(int n, float* sink) {
	__shared__ float4 lds[64 * 2];
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
	const float b = 1.0001f, c = 0.5f;
	const v2f bb = {b, b}, cc = {c, c};
	lds[threadIdx.x]	  = make_float4(a0, a1, a2, a3);
	lds[64 + threadIdx.x] = make_float4(a0, a1, a2, a3);
	__syncthreads();
	const unsigned addr = threadIdx.x * 16;// linear: conflict-free
	v4f r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
	typedef float v3f __attribute__((ext_vector_type(3)));
	v3f q3 = {a0, a1, a2};
	for(int i = 0; i < n; ++i) {
		if constexpr(MODE == 0) {
			asm volatile(REP8("s_sleep 16\n"));
		} else if constexpr(MODE == 1) {
			asm volatile(REP64("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
							   "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
						 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
						 : "v"(b), "v"(c));
		} else if constexpr(MODE == 2) {
			asm volatile(REP64("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
							   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
						 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
						 : "v"(bb), "v"(cc));
		} else if constexpr(MODE == 3) {
			asm volatile(REP64("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
							   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
						 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
						 : "v"(b), "v"(c));
		} else if constexpr(MODE == 9) {
			asm volatile(REP64("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
							   "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
						 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
		} else if constexpr(MODE == 4) {
			asm volatile(REP8("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4 offset:1024\n"
							  "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4 offset:1024\n s_waitcnt lgkmcnt(0)\n")
						 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
						 : "v"(addr)
						 : "memory");
		} else if constexpr(MODE == 5) {
			asm volatile(REP8("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n"
							  "ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n s_waitcnt lgkmcnt(0)\n")
						 :
						 : "v"(addr), "v"(r0)
						 : "memory");
		} else if constexpr(MODE == 10) {// 16 B per lane as two ds_write_b64
			asm volatile(REP8("ds_write_b64 %0, %1\n ds_write_b64 %0, %2 offset:8\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %2 offset:1032\n"
							  "ds_write_b64 %0, %1\n ds_write_b64 %0, %2 offset:8\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %2 offset:1032\n s_waitcnt lgkmcnt(0)\n")
						 :
						 : "v"(addr), "v"(p0), "v"(p1)
						 : "memory");
		} else if constexpr(MODE == 11) {// 16 B per lane as one ds_write2_b64 (offsets in units of 8 B)
			asm volatile(REP8("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1\n ds_write2_b64 %0, %1, %2 offset0:128 offset1:129\n ds_write2_b64 %0, %1, %2 offset0:0 offset1:1\n ds_write2_b64 %0, %1, %2 offset0:128 offset1:129\n"
							  "ds_write2_b64 %0, %1, %2 offset0:0 offset1:1\n ds_write2_b64 %0, %1, %2 offset0:128 offset1:129\n ds_write2_b64 %0, %1, %2 offset0:0 offset1:1\n ds_write2_b64 %0, %1, %2 offset0:128 offset1:129\n s_waitcnt lgkmcnt(0)\n")
						 :
						 : "v"(addr), "v"(p0), "v"(p1)
						 : "memory");
		} else if constexpr(MODE == 12) {// 16 B per lane as one ds_read2_b64
			asm volatile(REP8("ds_read2_b64 %0, %4 offset0:0 offset1:1\n ds_read2_b64 %1, %4 offset0:128 offset1:129\n ds_read2_b64 %2, %4 offset0:0 offset1:1\n ds_read2_b64 %3, %4 offset0:128 offset1:129\n"
							  "ds_read2_b64 %0, %4 offset0:0 offset1:1\n ds_read2_b64 %1, %4 offset0:128 offset1:129\n ds_read2_b64 %2, %4 offset0:0 offset1:1\n ds_read2_b64 %3, %4 offset0:128 offset1:129\n s_waitcnt lgkmcnt(0)\n")
						 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
						 : "v"(addr)
						 : "memory");
		} else if constexpr(MODE == 13) {// 12 B per lane: ds_write_b96
			asm volatile(REP8("ds_write_b96 %0, %1\n ds_write_b96 %0, %1 offset:1024\n ds_write_b96 %0, %1\n ds_write_b96 %0, %1 offset:1024\n"
							  "ds_write_b96 %0, %1\n ds_write_b96 %0, %1 offset:1024\n ds_write_b96 %0, %1\n ds_write_b96 %0, %1 offset:1024\n s_waitcnt lgkmcnt(0)\n")
						 :
						 : "v"(addr), "v"(q3)
						 : "memory");
		} else if constexpr(MODE == 6) {
			asm volatile(REP8("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_write_b128 %1, %0\n ds_read_b128 %0, %1 offset:1024\n s_waitcnt lgkmcnt(0)\n ds_write_b128 %1, %0 offset:1024\n"
							  "ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_write_b128 %1, %0\n ds_read_b128 %0, %1 offset:1024\n s_waitcnt lgkmcnt(0)\n ds_write_b128 %1, %0 offset:1024\n")
						 : "+v"(r0)
						 : "v"(addr)
						 : "memory");
		}
	}
	sink[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + r0.x + r1.y + r2.z + r3.w;
}

__global__ __launch_bounds__(256) void stream_read(const float4* __restrict__ src, size_t n, float* sink) {
	float4 acc = {0, 0, 0, 0};
	for(size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
		const float4 v = src[i];
		acc.x += v.x;
		acc.y += v.y;
		acc.z += v.z;
		acc.w += v.w;
	}
	if(acc.x + acc.y + acc.z + acc.w == 12345.f) sink[0] = 1.f;
}
__global__ __launch_bounds__(256) void stream_write(float4* __restrict__ dst, size_t n) {
	for(size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template<int MODE>
static void run(double seconds, float* sink, int per_iter, const char* what) {
	const int W		 = 8;
	const int blocks = 256 * 4 * W;
	const int n		 = MODE == 0 ? 2000 : (MODE == 4 || MODE == 5 || MODE == 6 || MODE >= 10 ? 400 : 200);
	burn<MODE><<<blocks, 64>>>(10, sink);
	hipDeviceSynchronize();
	const auto t0 = std::chrono::steady_clock::now();
	long launches = 0;
	double el	  = 0;
	do {
		for(int k = 0; k < 8; ++k) burn<MODE><<<blocks, 64>>>(n, sink);
		hipDeviceSynchronize();
		launches += 8;
		el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	} while(el < seconds);
	const double ops = (double) launches * blocks * (double) n * per_iter;// wave-instructions
	printf("mode %d %-28s %.2f s  %.4e wave-instructions/s (%.3e per SIMD per s)\n", MODE, what, el, ops / el, ops / el / 1024.0);
}

int main(int argc, char** argv) {
	const int mode		 = argc > 1 ? atoi(argv[1]) : 1;
	const double seconds = argc > 2 ? atof(argv[2]) : 2.0;
	float* sink;
	hipMalloc(&sink, sizeof(float) * 64 * 256 * 4 * 8);
	switch(mode) {
		case 0: run<0>(seconds, sink, 8, "idle waves (s_sleep)"); break;
		case 1: run<1>(seconds, sink, 512, "v_fmac_f32"); break;
		case 2: run<2>(seconds, sink, 512, "v_pk_fma_f32"); break;
		case 3: run<3>(seconds, sink, 512, "v_fma_f32 (VOP3)"); break;
		case 9: run<9>(seconds, sink, 512, "v_mov_b32"); break;
		case 4: run<4>(seconds, sink, 64, "ds_read_b128"); break;
		case 5: run<5>(seconds, sink, 64, "ds_write_b128"); break;
		case 6: run<6>(seconds, sink, 64, "ds rmw b128 (read+write = 2)"); break;
		case 10: run<10>(seconds, sink, 64, "ds_write_b64 (2 per 16 B)"); break;
		case 11: run<11>(seconds, sink, 64, "ds_write2_b64 (16 B)"); break;
		case 12: run<12>(seconds, sink, 64, "ds_read2_b64 (16 B)"); break;
		case 13: run<13>(seconds, sink, 64, "ds_write_b96"); break;
		case 7:
		case 8: {
			const size_t n = (size_t) 1 << 28;// 4 GiB of float4
			float4* buf;
			if(hipMalloc(&buf, n * sizeof(float4)) != hipSuccess) return 1;
			hipMemset(buf, 0, n * sizeof(float4));
			hipDeviceSynchronize();
			const auto t0 = std::chrono::steady_clock::now();
			long launches = 0;
			double el	  = 0;
			do {
				for(int k = 0; k < 4; ++k) {
					if(mode == 7)
						stream_read<<<256 * 16, 256>>>(buf, n, sink);
					else
						stream_write<<<256 * 16, 256>>>(buf, n);
				}
				hipDeviceSynchronize();
				launches += 4;
				el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			} while(el < seconds);
			printf("mode %d %-28s %.2f s  %.4e bytes/s\n", mode, mode == 7 ? "global load stream" : "global store stream", el, (double) launches * n * 16 / el);
			break;
		}
		default: return 2;
	}
	return 0;
}
