// Cost model, part 2 (gfx950): hard-register VALU forms (bank / encoding / SGPR / VCC effects) and LDS throughput of
// b128 / b96 accesses under arbitrary lane -> address maps (the g2p2g arena layouts and candidate replacements).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_microbench3 valu_microbench3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#include <algorithm>
#include <random>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v3f __attribute__((ext_vector_type(3)));
constexpr int ITERS = 2048;

#define S2(x) #x
#define S1(x) S2(x)
// 16 instructions with destination register index 20 + step * i
#define R16(PRE, POST) \
	PRE "20" POST "\n" PRE "21" POST "\n" PRE "22" POST "\n" PRE "23" POST "\n" PRE "24" POST "\n" PRE "25" POST "\n" PRE "26" POST "\n" PRE "27" POST "\n" \
	PRE "28" POST "\n" PRE "29" POST "\n" PRE "30" POST "\n" PRE "31" POST "\n" PRE "32" POST "\n" PRE "33" POST "\n" PRE "34" POST "\n" PRE "35" POST "\n"
#define R16ACC(OP, MID) \
	OP " v20" MID "v20\n" OP " v21" MID "v21\n" OP " v22" MID "v22\n" OP " v23" MID "v23\n" OP " v24" MID "v24\n" OP " v25" MID "v25\n" OP " v26" MID "v26\n" OP " v27" MID "v27\n" \
	OP " v28" MID "v28\n" OP " v29" MID "v29\n" OP " v30" MID "v30\n" OP " v31" MID "v31\n" OP " v32" MID "v32\n" OP " v33" MID "v33\n" OP " v34" MID "v34\n" OP " v35" MID "v35\n"
#define R16DD(OP, TAIL) \
	OP " v20, v20" TAIL "\n" OP " v21, v21" TAIL "\n" OP " v22, v22" TAIL "\n" OP " v23, v23" TAIL "\n" OP " v24, v24" TAIL "\n" OP " v25, v25" TAIL "\n" OP " v26, v26" TAIL "\n" OP " v27, v27" TAIL "\n" \
	OP " v28, v28" TAIL "\n" OP " v29, v29" TAIL "\n" OP " v30, v30" TAIL "\n" OP " v31, v31" TAIL "\n" OP " v32, v32" TAIL "\n" OP " v33, v33" TAIL "\n" OP " v34, v34" TAIL "\n" OP " v35, v35" TAIL "\n"
#define R8PK(OP, MID, TAIL) \
	OP " v[20:21]" MID TAIL "\n" OP " v[22:23]" MID TAIL "\n" OP " v[24:25]" MID TAIL "\n" OP " v[26:27]" MID TAIL "\n" OP " v[28:29]" MID TAIL "\n" OP " v[30:31]" MID TAIL "\n" OP " v[32:33]" MID TAIL "\n" OP " v[34:35]" MID TAIL "\n"
#define R8PKACC(OP, MID) \
	OP " v[20:21]" MID "v[20:21]\n" OP " v[22:23]" MID "v[22:23]\n" OP " v[24:25]" MID "v[24:25]\n" OP " v[26:27]" MID "v[26:27]\n" OP " v[28:29]" MID "v[28:29]\n" OP " v[30:31]" MID "v[30:31]\n" OP " v[32:33]" MID "v[32:33]\n" OP " v[34:35]" MID "v[34:35]\n"
#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "vcc", "s10", "s11"

struct Form {
	const char* name;
	int n;// instructions per block
};
enum {
	F_FMA_B123, F_FMA_B111, F_FMA_ACC, F_FMAC, F_FMA_D0, C_VCC, C_SGPR, C_VCC_D, CMP_CND, CMP_E64, CMP_VCC, P_MUL, P_MUL_D, P_FMA4, P_FMAACC, P_ADD, S_MUL, S_FMA, M_MAX, M_MED3, B_BFI, X_XOR, A_ADD3, R_RFL, MUL_B12, MUL_B11, ADD_LIT, FMA_LIT, FMAAK, NFORMS
};
static const Form kForms[NFORMS] = {
	{"v_fma_f32 d, v1, v2, v3 (banks 1,2,3)", 16}, {"v_fma_f32 d, v1, v5, v9 (bank 1 x3)", 16}, {"v_fma_f32 d, v1, v2, d", 16}, {"v_fmac_f32 d, v1, v2", 16}, {"v_fma_f32 d, d, v1, v2", 16},
	{"v_cndmask_b32 d, v1, v2, vcc", 16}, {"v_cndmask_b32_e64 d, v1, v2, s[10:11]", 16}, {"v_cndmask_b32 d, d, v2, vcc", 16}, {"v_cmp_lt vcc + v_cndmask (8 pairs)", 16}, {"v_cmp_lt_f32_e64 s[10:11], v1, d", 16}, {"v_cmp_lt_f32 vcc, v1, d", 16},
	{"v_pk_mul_f32 d, v[2:3], v[4:5]", 8}, {"v_pk_mul_f32 d, d, v[4:5]", 8}, {"v_pk_fma_f32 d, v[2:3], v[4:5], v[6:7]", 8}, {"v_pk_fma_f32 d, v[2:3], v[4:5], d", 8}, {"v_pk_add_f32 d, v[2:3], v[4:5]", 8},
	{"v_mul_f32 d, s4, v1", 16}, {"v_fma_f32 d, s4, v1, v2", 16}, {"v_max_f32 d, v1, v2", 16}, {"v_med3_f32 d, v1, v2, v3", 16}, {"v_bfi_b32 d, v1, v2, v3", 16}, {"v_xor_b32 d, v1, v2", 16}, {"v_add3_u32 d, v1, v2, v3", 16},
	{"v_readfirstlane_b32 s10, d", 16}, {"v_mul_f32 d, v1, v2 (banks 1,2)", 16}, {"v_mul_f32 d, v1, v5 (bank 1 x2)", 16}, {"v_add_f32 d, 0x3f8ccccd, v1 (literal)", 16}, {"v_fmamk_f32 d, v1, lit, v2", 16}, {"v_fmaak_f32 d, v1, v2, lit", 16}};

template<int FORM>
__global__ void kvalu(float* out, unsigned long long* cyc) {
	asm volatile("v_mov_b32 v1, 1.0\n v_mov_b32 v2, 0.5\n v_mov_b32 v3, 2.0\n v_mov_b32 v4, 1.0\n v_mov_b32 v5, 0.5\n v_mov_b32 v6, 1.0\n v_mov_b32 v7, 1.0\n v_mov_b32 v9, 1.0\n s_mov_b32 s4, 0x3f800000\n"
				 "s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x33333333\n s_mov_b32 s10, 0x55555555\n s_mov_b32 s11, 0x0f0f0f0f\n" R16("v_mov_b32 v", ", 1.0")::
					 : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v9", "s4", CLOB);
	__syncthreads();
	const unsigned long long t0 = __builtin_readcyclecounter();
	for(int it = 0; it < ITERS; ++it) {
		if constexpr(FORM == F_FMA_B123) asm volatile(R16("v_fma_f32 v", ", v1, v2, v3")::: CLOB);
		else if constexpr(FORM == F_FMA_B111) asm volatile(R16("v_fma_f32 v", ", v1, v5, v9")::: CLOB);
		else if constexpr(FORM == F_FMA_ACC) asm volatile(R16ACC("v_fma_f32", ", v1, v2, ")::: CLOB);
		else if constexpr(FORM == F_FMAC) asm volatile(R16("v_fmac_f32 v", ", v1, v2")::: CLOB);
		else if constexpr(FORM == F_FMA_D0) asm volatile(R16DD("v_fma_f32", ", v1, v2")::: CLOB);
		else if constexpr(FORM == C_VCC) asm volatile(R16("v_cndmask_b32 v", ", v1, v2, vcc")::: CLOB);
		else if constexpr(FORM == C_SGPR) asm volatile(R16("v_cndmask_b32_e64 v", ", v1, v2, s[10:11]")::: CLOB);
		else if constexpr(FORM == C_VCC_D) asm volatile(R16DD("v_cndmask_b32", ", v2, vcc")::: CLOB);
		else if constexpr(FORM == CMP_CND)
			asm volatile("v_cmp_lt_f32 vcc, v1, v20\n v_cndmask_b32 v20, v1, v2, vcc\n v_cmp_lt_f32 vcc, v1, v21\n v_cndmask_b32 v21, v1, v2, vcc\n v_cmp_lt_f32 vcc, v1, v22\n v_cndmask_b32 v22, v1, v2, vcc\n v_cmp_lt_f32 vcc, v1, v23\n v_cndmask_b32 v23, v1, v2, vcc\n"
						 "v_cmp_lt_f32 vcc, v1, v24\n v_cndmask_b32 v24, v1, v2, vcc\n v_cmp_lt_f32 vcc, v1, v25\n v_cndmask_b32 v25, v1, v2, vcc\n v_cmp_lt_f32 vcc, v1, v26\n v_cndmask_b32 v26, v1, v2, vcc\n v_cmp_lt_f32 vcc, v1, v27\n v_cndmask_b32 v27, v1, v2, vcc\n" ::
							 : CLOB);
		else if constexpr(FORM == CMP_E64) asm volatile(R16("v_cmp_lt_f32_e64 s[10:11], v1, v", "")::: CLOB);
		else if constexpr(FORM == CMP_VCC) asm volatile(R16("v_cmp_lt_f32 vcc, v1, v", "")::: CLOB);
		else if constexpr(FORM == P_MUL) asm volatile(R8PK("v_pk_mul_f32", ", v[2:3], v[4:5]", "")::: CLOB);
		else if constexpr(FORM == P_MUL_D) asm volatile(R8PKACC("v_pk_mul_f32", ", v[4:5], ")::: CLOB);
		else if constexpr(FORM == P_FMA4) asm volatile(R8PK("v_pk_fma_f32", ", v[2:3], v[4:5], v[6:7]", "")::: CLOB);
		else if constexpr(FORM == P_FMAACC) asm volatile(R8PKACC("v_pk_fma_f32", ", v[2:3], v[4:5], ")::: CLOB);
		else if constexpr(FORM == P_ADD) asm volatile(R8PK("v_pk_add_f32", ", v[2:3], v[4:5]", "")::: CLOB);
		else if constexpr(FORM == S_MUL) asm volatile(R16("v_mul_f32 v", ", s4, v1")::: CLOB);
		else if constexpr(FORM == S_FMA) asm volatile(R16("v_fma_f32 v", ", s4, v1, v2")::: CLOB);
		else if constexpr(FORM == M_MAX) asm volatile(R16("v_max_f32 v", ", v1, v2")::: CLOB);
		else if constexpr(FORM == M_MED3) asm volatile(R16("v_med3_f32 v", ", v1, v2, v3")::: CLOB);
		else if constexpr(FORM == B_BFI) asm volatile(R16("v_bfi_b32 v", ", v1, v2, v3")::: CLOB);
		else if constexpr(FORM == X_XOR) asm volatile(R16("v_xor_b32 v", ", v1, v2")::: CLOB);
		else if constexpr(FORM == A_ADD3) asm volatile(R16("v_add3_u32 v", ", v1, v2, v3")::: CLOB);
		else if constexpr(FORM == R_RFL) asm volatile(R16("v_readfirstlane_b32 s10, v", "")::: CLOB);
		else if constexpr(FORM == MUL_B12) asm volatile(R16("v_mul_f32 v", ", v1, v2")::: CLOB);
		else if constexpr(FORM == MUL_B11) asm volatile(R16("v_mul_f32 v", ", v1, v5")::: CLOB);
		else if constexpr(FORM == ADD_LIT) asm volatile(R16("v_add_f32 v", ", 0x3f8ccccd, v1")::: CLOB);
		else if constexpr(FORM == FMA_LIT) asm volatile(R16("v_fmamk_f32 v", ", v1, 0x3f8ccccd, v2")::: CLOB);
		else if constexpr(FORM == FMAAK) asm volatile(R16("v_fmaak_f32 v", ", v1, v2, 0x3f8ccccd")::: CLOB);
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	float r;
	asm volatile("v_add_f32 %0, v20, v35" : "=v"(r)::CLOB);
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
	if((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template<int FORM>
void run_valu(int wps) {
	const int threads = 256 * wps, blocks = 256;
	float* d;
	unsigned long long* c;
	(void) hipMalloc(&d, sizeof(float) * threads * blocks);
	(void) hipMalloc(&c, sizeof(unsigned long long) * blocks * threads / 64);
	kvalu<FORM><<<blocks, threads>>>(d, c);
	kvalu<FORM><<<blocks, threads>>>(d, c);
	(void) hipDeviceSynchronize();
	std::vector<unsigned long long> h(blocks * threads / 64);
	(void) hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
	double sum = 0;
	for(auto v: h) sum += (double) v;
	const double per_wave = sum / h.size() / ((double) ITERS * kForms[FORM].n);
	printf("%-44s w/SIMD=%d : %6.2f cyc/inst/SIMD (wave sees %6.2f)\n", kForms[FORM].name, wps, per_wave / wps, per_wave);
	(void) hipFree(d);
	(void) hipFree(c);
}
template<int M>
void run_all_valu() {
	if constexpr(M < NFORMS) {
		for(int w: {1, 2, 4}) run_valu<M>(w);
		run_all_valu<M + 1>();
	}
}

// ---- LDS throughput under a lane -> byte-address map.  WPC waves per CU, each issues batches of 9 accesses at immediate
// offsets (0, 16, 32 bytes x three 128-byte steps: the 3 x 3 pencil pattern does not matter for banking beyond the base).
template<int OP>// 0 read b128, 1 write b128, 2 read b96, 3 rmw b128 chain (read, wait, write)
__global__ void klds(const int* __restrict__ map, unsigned long long* cyc, float* out, int region_bytes) {
	extern __shared__ char smem[];
	for(int i = threadIdx.x; i < (int) (region_bytes * (blockDim.x >> 6)) / 4; i += blockDim.x) ((float*) smem)[i] = 0.f;
	__syncthreads();
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const unsigned a = (unsigned) (map[lane] + w * region_bytes);
	v4f acc = {0.f, 0.f, 0.f, 0.f};
	const unsigned long long t0 = __builtin_readcyclecounter();
	for(int it = 0; it < 256; ++it) {
		if constexpr(OP == 0) {
			v4f r0, r1, r2, r3, r4, r5, r6, r7;
			asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:128\n ds_read_b128 %4, %8 offset:144\n ds_read_b128 %5, %8 offset:160\n ds_read_b128 %6, %8 offset:256\n ds_read_b128 %7, %8 offset:272\n s_waitcnt lgkmcnt(0)"
						 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)
						 : "v"(a));
			acc += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
		} else if constexpr(OP == 1) {
			asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:16\n ds_write_b128 %0, %1 offset:32\n ds_write_b128 %0, %1 offset:128\n ds_write_b128 %0, %1 offset:144\n ds_write_b128 %0, %1 offset:160\n ds_write_b128 %0, %1 offset:256\n ds_write_b128 %0, %1 offset:272\n s_waitcnt lgkmcnt(0)" ::"v"(a), "v"(acc)
						 : "memory");
		} else if constexpr(OP == 2) {
			v3f r0, r1, r2, r3, r4, r5, r6, r7;
			asm volatile("ds_read_b96 %0, %8\n ds_read_b96 %1, %8 offset:16\n ds_read_b96 %2, %8 offset:32\n ds_read_b96 %3, %8 offset:128\n ds_read_b96 %4, %8 offset:144\n ds_read_b96 %5, %8 offset:160\n ds_read_b96 %6, %8 offset:256\n ds_read_b96 %7, %8 offset:272\n s_waitcnt lgkmcnt(0)"
						 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)
						 : "v"(a));
			acc.x += r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x;
		} else {
#pragma unroll
			for(int o = 0; o < 8; ++o) {
				v4f r;
				asm volatile("ds_read_b128 %0, %1 offset:%2\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a), "n"(0));
				r += acc;
				asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(r) : "memory");
			}
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	if(lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + w] = t1 - t0;
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

static void run_lds(const char* name, const std::vector<int>& map, int region_bytes, int wpc) {
	int* dm;
	unsigned long long* c;
	float* d;
	(void) hipMalloc(&dm, 256);
	(void) hipMalloc(&c, 8 * 256 * 16);
	(void) hipMalloc(&d, 4 * 256 * 1024);
	(void) hipMemcpy(dm, map.data(), 256, hipMemcpyHostToDevice);
	const char* opn[4] = {"read_b128", "write_b128", "read_b96", "rmw_b128"};
	printf("%-46s", name);
	for(int op = 0; op < 4; ++op) {
		const size_t sh = (size_t) region_bytes * wpc;
		for(int rep = 0; rep < 2; ++rep) {
			if(op == 0) klds<0><<<256, 64 * wpc, sh>>>(dm, c, d, region_bytes);
			else if(op == 1) klds<1><<<256, 64 * wpc, sh>>>(dm, c, d, region_bytes);
			else if(op == 2) klds<2><<<256, 64 * wpc, sh>>>(dm, c, d, region_bytes);
			else klds<3><<<256, 64 * wpc, sh>>>(dm, c, d, region_bytes);
		}
		(void) hipDeviceSynchronize();
		std::vector<unsigned long long> h(256 * wpc);
		(void) hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
		double sum = 0;
		for(auto x: h) sum += (double) x;
		const double n = 256.0 * 8.0 * (op == 3 ? 2.0 : 1.0);
		printf(" %s %6.2f", opn[op], sum / h.size() / n / wpc * 1.0);// cycles per wave-instruction per CU = per-wave time / instrs / waves... (all waves run concurrently)
	}
	printf("   (cycles per LDS instruction per CU, %d waves)\n", wpc);
	(void) hipFree(dm);
	(void) hipFree(c);
	(void) hipFree(d);
}

int main(int argc, char** argv) {
	const bool lds_only = argc > 1 && std::string(argv[1]) == "lds";
	if(!lds_only) run_all_valu<0>();
	const int wpc = 8;
	auto mk = [&](std::function<int(int)> f) {
		std::vector<int> m(64);
		for(int i = 0; i < 64; ++i) m[i] = f(i);
		return m;
	};
	const int R = 16384 - 1024;// per-wave region
	run_lds("linear 16 B per lane", mk([](int l) { return l * 16; }), R, wpc);
	run_lds("stride 32 B", mk([](int l) { return l * 32; }), R, wpc);
	run_lds("stride 64 B", mk([](int l) { return l * 64; }), R, wpc);
	run_lds("stride 128 B", mk([](int l) { return l * 128; }), R, wpc);
	run_lds("stride 256 B (one bank group)", mk([](int l) { return (l * 256) % 12288; }), R, wpc);
	run_lds("stride 48 B (3 nodes)", mk([](int l) { return l * 48; }), R, wpc);
	run_lds("stride 80 B", mk([](int l) { return l * 80; }), R, wpc);
	// the kernel's 8^3 arena (x stride 68 nodes, y stride 8): the 64 populated keys x, y, z in 1..4
	auto arena = [&](int sx, int sy, int order) {
		return mk([=](int l) {
			const int a = l & 3, b = (l >> 2) & 3, c = l >> 4;// a fastest
			int x, y, z;
			if(order == 0) { z = a; y = b; x = c; }		 // key = x*36 + y*6 + z (current)
			else if(order == 1) { z = a; x = b; y = c; }	 // key = y*36 + x*6 + z
			else { x = a; y = b; z = c; }
			return ((x + 1) * sx + (y + 1) * sy + (z + 1)) * 16;
		});
	};
	run_lds("arena 68/8 keys x,y,z (current)", arena(68, 8, 0), R, wpc);
	run_lds("arena 68/8 keys y,x,z", arena(68, 8, 1), R, wpc);
	run_lds("arena 66/8 keys x,y,z", arena(66, 8, 0), R, wpc);
	run_lds("arena 72/8 keys x,y,z", arena(72, 8, 0), R, wpc);
	run_lds("arena 64/8 keys x,y,z (dense 8^3)", arena(64, 8, 0), R, wpc);
	run_lds("arena 36/6 dense 6^3 keys x,y,z", arena(36, 6, 0), R, wpc);
	run_lds("arena 36/6 dense 6^3 keys y,x,z", arena(36, 6, 1), R, wpc);
	run_lds("arena 40/6 keys x,y,z", arena(40, 6, 0), R, wpc);
	run_lds("arena 38/6 keys x,y,z", arena(38, 6, 0), R, wpc);
	run_lds("arena 37/6 keys x,y,z", arena(37, 6, 0), R, wpc);
	run_lds("arena 44/7 keys x,y,z", arena(44, 7, 0), R, wpc);
	run_lds("arena 52/8 keys x,y,z (g2p current)", arena(52, 8, 0), R, wpc);
	run_lds("arena 52/8 keys y,x,z", arena(52, 8, 1), R, wpc);
	{
		std::mt19937 rng(1);
		std::vector<int> nodes(216);
		for(int i = 0; i < 216; ++i) nodes[i] = ((i / 36 + 1) * 68 + ((i / 6) % 6 + 1) * 8 + (i % 6) + 1) * 16;
		std::shuffle(nodes.begin(), nodes.end(), rng);
		std::vector<int> m(nodes.begin(), nodes.begin() + 64);
		run_lds("arena 68/8 random 64 of 216 keys", m, R, wpc);
		std::sort(m.begin(), m.end());
		run_lds("arena 68/8 random 64 of 216 keys, sorted", m, R, wpc);
	}
	// 12-byte nodes (b96 gather arena without padding): dense 6^3
	run_lds("12-B nodes dense 6^3 keys x,y,z", mk([](int l) { return (((l >> 4) + 1) * 36 + (((l >> 2) & 3) + 1) * 6 + (l & 3) + 1) * 12; }), R, wpc);
	return 0;
}
