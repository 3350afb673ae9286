// VALU / LDS issue-cost model for gfx950 (round 2): per-wave s_memtime cycles, W waves per SIMD, one workgroup per CU.
//   part 1: cycles per wave-instruction per SIMD for the instruction forms g2p2g is made of (independent and dependent)
//   part 2: which lanes of a wave share an LDS pass for ds_read_b128 / ds_write_b128 / ds_read_b96 (two active lanes
//           on the same banks, different rows: +1 pass iff the hardware serves them in the same group)
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_microbench2 valu_microbench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v3f __attribute__((ext_vector_type(3)));
constexpr int ITERS = 2048;

#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

enum Mode {
	ADD, MUL, FMAC, FMA4, FMA_S, FMA_AA, PKFMA, PKFMA4, PKMUL, PKADD, PKFMA_OPSEL, MOV, CNDMASK, RSQ, RCP, SQRT, LOG, EXP, MIX_RSQ_FMA, MIX_RSQ_PK,
	CVT, RNDNE, ADDU, LSHLADD64, MADU24, MAXF, DPP_ADD, FMA_DEP, PKFMA_DEP, MUL_DEP, FMA_K, SUB, MIX_FMA_PK, MIX_MUL_FMA, NMODES
};
static const char* kNames[NMODES] = {"v_add_f32", "v_mul_f32", "v_fmac_f32 (d=a*b+d)", "v_fma_f32 4 regs", "v_fma_f32 sgpr src", "v_fma_f32 d,a,a,c", "v_pk_fma_f32 acc", "v_pk_fma_f32 4 regs", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32 op_sel bcast", "v_mov_b32", "v_cndmask_b32 vcc", "v_rsq_f32", "v_rcp_f32", "v_sqrt_f32", "v_log_f32", "v_exp_f32", "1 rsq + 3 fma", "1 rsq + 3 pk_fma",
	"v_cvt_i32_f32", "v_rndne_f32", "v_add_u32", "v_lshl_add_u64", "v_mad_u32_u24", "v_max_f32", "v_add_f32 dpp row_shr:1", "v_fma_f32 dependent", "v_pk_fma_f32 dependent", "v_mul_f32 dependent", "v_fma_f32 inline const", "v_sub_f32", "2 fma + 2 pk_fma", "2 mul + 2 fma"};

template<int MODE>
__global__ void kvalu(float* out, unsigned long long* cyc, float a, float b, float sc) {
	float x[16];
	v2f p[16];
#pragma unroll
	for(int i = 0; i < 16; ++i) {
		x[i] = threadIdx.x + i;
		p[i] = (v2f) {x[i], x[i] + 1.f};
	}
	v2f pa = {a, a + 1e-7f}, pb = {b, b};
	float c = b * 2.f;
	v2f pc  = {c, c};
	unsigned long long u = threadIdx.x;
	int iu = threadIdx.x;
	asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555" ::: "vcc");
	__syncthreads();
	const unsigned long long t0 = __builtin_readcyclecounter();
	for(int it = 0; it < ITERS; ++it) {
#define A1(i) asm volatile("v_add_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define A2(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define A3(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define A4(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "+v"(x[i]) : "v"(a), "v"(b), "v"(c));
#define A5(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "+v"(x[i]) : "v"(a), "s"(sc), "v"(c));
#define A6(i) asm volatile("v_fma_f32 %0, %1, %1, %2" : "+v"(x[i]) : "v"(a), "v"(c));
#define A7(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pa), "v"(pb));
#define A8(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "+v"(p[i]) : "v"(pa), "v"(pb), "v"(pc));
#define A9(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
#define A10(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
#define A11(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pa), "v"(pb));
#define A12(i) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(a));
#define A13(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
#define A14(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
#define A15(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
#define A16(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
#define A17(i) asm volatile("v_log_f32 %0, %0" : "+v"(x[i]));
#define A18(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
#define A19(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[i]));
#define A20(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(x[i]));
#define A21(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(iu));
#define A22(i) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(u) : "v"(u));
#define A23(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x[i]) : "v"(iu));
#define A24(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define A25(i) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(a));
#define A26(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b));
#define A27(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(pa), "v"(pb));
#define A28(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[0]) : "v"(a));
#define A29(i) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x[i]) : "v"(a));
#define A30(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
		if constexpr(MODE == ADD) { REP16(A1) }
		else if constexpr(MODE == MUL) { REP16(A2) }
		else if constexpr(MODE == FMAC) { REP16(A3) }
		else if constexpr(MODE == FMA4) { REP16(A4) }
		else if constexpr(MODE == FMA_S) { REP16(A5) }
		else if constexpr(MODE == FMA_AA) { REP16(A6) }
		else if constexpr(MODE == PKFMA) { REP16(A7) }
		else if constexpr(MODE == PKFMA4) { REP16(A8) }
		else if constexpr(MODE == PKMUL) { REP16(A9) }
		else if constexpr(MODE == PKADD) { REP16(A10) }
		else if constexpr(MODE == PKFMA_OPSEL) { REP16(A11) }
		else if constexpr(MODE == MOV) { REP16(A12) }
		else if constexpr(MODE == CNDMASK) { REP16(A13) }
		else if constexpr(MODE == RSQ) { REP16(A14) }
		else if constexpr(MODE == RCP) { REP16(A15) }
		else if constexpr(MODE == SQRT) { REP16(A16) }
		else if constexpr(MODE == LOG) { REP16(A17) }
		else if constexpr(MODE == EXP) { REP16(A18) }
		else if constexpr(MODE == MIX_RSQ_FMA) { A14(0) A3(1) A3(2) A3(3) A14(4) A3(5) A3(6) A3(7) A14(8) A3(9) A3(10) A3(11) A14(12) A3(13) A3(14) A3(15) }
		else if constexpr(MODE == MIX_RSQ_PK) { A14(0) A7(1) A7(2) A7(3) A14(4) A7(5) A7(6) A7(7) A14(8) A7(9) A7(10) A7(11) A14(12) A7(13) A7(14) A7(15) }
		else if constexpr(MODE == CVT) { REP16(A19) }
		else if constexpr(MODE == RNDNE) { REP16(A20) }
		else if constexpr(MODE == ADDU) { REP16(A21) }
		else if constexpr(MODE == LSHLADD64) { REP16(A22) }
		else if constexpr(MODE == MADU24) { REP16(A23) }
		else if constexpr(MODE == MAXF) { REP16(A24) }
		else if constexpr(MODE == DPP_ADD) { REP16(A25) }
		else if constexpr(MODE == FMA_DEP) { REP16(A26) }
		else if constexpr(MODE == PKFMA_DEP) { REP16(A27) }
		else if constexpr(MODE == MUL_DEP) { REP16(A28) }
		else if constexpr(MODE == FMA_K) { REP16(A29) }
		else if constexpr(MODE == SUB) { REP16(A30) }
		else if constexpr(MODE == MIX_FMA_PK) { A3(0) A3(1) A7(2) A7(3) A3(4) A3(5) A7(6) A7(7) A3(8) A3(9) A7(10) A7(11) A3(12) A3(13) A7(14) A7(15) }
		else if constexpr(MODE == MIX_MUL_FMA) { A2(0) A2(1) A3(2) A3(3) A2(4) A2(5) A3(6) A3(7) A2(8) A2(9) A3(10) A3(11) A2(12) A2(13) A3(14) A3(15) }
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	float s = 0.f;
#pragma unroll
	for(int i = 0; i < 16; ++i) s += x[i] + p[i].x + p[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float) u;
	if((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template<int MODE>
void run_valu(int wps) {
	const int threads = 256 * wps, blocks = 256;
	float* d;
	unsigned long long* c;
	(void) hipMalloc(&d, sizeof(float) * threads * blocks);
	(void) hipMalloc(&c, sizeof(unsigned long long) * blocks * threads / 64);
	kvalu<MODE><<<blocks, threads>>>(d, c, 1.0001f, 0.5f, 0.25f);
	kvalu<MODE><<<blocks, threads>>>(d, c, 1.0001f, 0.5f, 0.25f);
	(void) hipDeviceSynchronize();
	std::vector<unsigned long long> h(blocks * threads / 64);
	(void) hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
	double sum = 0;
	for(auto v: h) sum += (double) v;
	const double per_wave = sum / h.size();
	printf("%-28s waves/SIMD=%d : %6.2f cycles per wave-instruction per SIMD (wave sees %6.2f per own instruction)\n", kNames[MODE], wps, per_wave / (ITERS * 16.0) / wps, per_wave / (ITERS * 16.0));
	(void) hipFree(d);
	(void) hipFree(c);
}

template<int M>
void run_all_valu() {
	if constexpr(M < NMODES) {
		for(int w: {1, 2, 3}) run_valu<M>(w);
		run_all_valu<M + 1>();
	}
}

// ---- part 2: LDS pass groups.  Two lanes (l0, l1) active, same banks, different rows.
template<int OP>
__global__ void klds(unsigned long long* cyc, int l0, int l1, float* out) {
	__shared__ float4 s[1024];
	for(int i = threadIdx.x; i < 1024; i += 64) s[i] = make_float4(i, 0.f, 0.f, 0.f);
	__syncthreads();
	const int lane = threadIdx.x;
	float acc = 0.f;
	unsigned long long t0 = 0, t1 = 0;
	// lane l0 -> row 0, lane l1 -> row 1 (256 B apart: same banks); all other lanes are masked off
	const int idx = lane == l0 ? 0 : 16;
	if(lane == l0 || lane == l1) {
		t0 = __builtin_readcyclecounter();
		for(int it = 0; it < 512; ++it) {
#pragma unroll
			for(int o = 0; o < 16; ++o) {
				if constexpr(OP == 0) {
					v4f v;
					asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned) (idx * 16)), "n"(0));
					asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
					acc += v.x;
				} else if constexpr(OP == 1) {
					v4f v = {acc, 1.f, 2.f, 3.f};
					asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned) (idx * 16)), "v"(v) : "memory");
					asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				} else {
					v3f v;
					asm volatile("ds_read_b96 %0, %1" : "=v"(v) : "v"((unsigned) (idx * 16)));
					asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
					acc += v.x;
				}
			}
		}
		t1 = __builtin_readcyclecounter();
	}
	if(lane == l0) cyc[0] = t1 - t0;
	out[lane] = acc;
}

// throughput of full-wave b128 under a lane -> quad mapping (pattern id), 4 waves per CU
__global__ void klds_pattern(unsigned long long* cyc, int pattern, float* out, int write) {
	__shared__ float4 s[2048];
	for(int i = threadIdx.x; i < 2048; i += blockDim.x) s[i] = make_float4(i, 0.f, 0.f, 0.f);
	__syncthreads();
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	int q, row;
	switch(pattern) {
		case 0: q = lane & 15; row = lane >> 4; break;					  // contiguous groups of 16 conflict-free
		case 1: q = lane >> 2; row = lane & 3; break;					  // lanes 4k..4k+3 share a quad
		case 2: q = (lane & 3) + 4 * (lane >> 4); row = (lane >> 2) & 3; break;
		case 3: q = (lane & 7) * 2 + ((lane >> 5) & 1); row = (lane >> 3) & 3; break;
		case 4: q = lane & 7; row = lane >> 3; break;					  // 2-way conflict within contiguous 16
		case 5: q = 0; row = lane; break;								  // everything on one quad
		default: q = (4 * (lane & 3) + 8 * ((lane >> 2) & 1) + ((lane >> 3) & 3)) & 15; row = lane >> 4; break;// the g2p2g arena hash
	}
	const unsigned addr = (unsigned) ((w * 512 + (row & 31) * 16 + q) * 16);
	float acc = 0.f;
	const unsigned long long t0 = __builtin_readcyclecounter();
	for(int it = 0; it < 512; ++it) {
#pragma unroll
		for(int o = 0; o < 16; ++o) {
			if(write) {
				v4f v = {acc, 1.f, 2.f, 3.f};
				asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
			} else {
				v4f v;
				asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				acc += v.x;
			}
		}
	}
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	const unsigned long long t1 = __builtin_readcyclecounter();
	if(lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
	const bool lds_only = argc > 1 && std::string(argv[1]) == "lds";
	if(!lds_only) run_all_valu<0>();
	unsigned long long* c;
	float* d;
	(void) hipMalloc(&c, 8 * 4096);
	(void) hipMalloc(&d, 4 * 65536);
	const char* opn[3] = {"ds_read_b128", "ds_write_b128", "ds_read_b96"};
	for(int op = 0; op < 3; ++op) {
		for(int l0: {0, 5, 16, 40}) {
			printf("%s lane %2d vs lane j (cycles per op, '*' = shares a pass):", opn[op], l0);
			std::vector<double> v(64, 0.0);
			double mn = 1e30;
			for(int j = 0; j < 64; ++j) {
				if(j == l0) continue;
				if(op == 0) klds<0><<<1, 64>>>(c, l0, j, d);
				else if(op == 1) klds<1><<<1, 64>>>(c, l0, j, d);
				else klds<2><<<1, 64>>>(c, l0, j, d);
				unsigned long long h;
				(void) hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
				v[j] = (double) h / (512.0 * 16.0);
				mn	 = std::min(mn, v[j]);
			}
			printf(" min %.1f\n   ", mn);
			for(int j = 0; j < 64; ++j) printf("%s", j == l0 ? "." : (v[j] > mn + 0.5 ? "*" : "-"));
			printf("\n");
		}
	}
	for(int write = 0; write < 2; ++write)
		for(int p = 0; p < 7; ++p) {
			klds_pattern<<<256, 256>>>(c, p, d, write);
			klds_pattern<<<256, 256>>>(c, p, d, write);
			std::vector<unsigned long long> h(1024);
			(void) hipMemcpy(h.data(), c, 8 * 1024, hipMemcpyDeviceToHost);
			double sum = 0;
			for(auto x: h) sum += (double) x;
			printf("%s pattern %d: %.2f cycles per wave-instruction per CU (4 waves)\n", write ? "ds_write_b128" : "ds_read_b128 ", p, sum / 1024 / (512.0 * 16.0) / 4.0);
		}
	return 0;
}
