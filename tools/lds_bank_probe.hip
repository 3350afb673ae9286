// LDS bank-conflict probe for the G2P2G arenas (gfx950): which lanes of a wave64 share a pass of ds_read_b128 / ds_write_b128, and
// what a candidate arena layout costs under the lane -> node patterns the kernel produces.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_bank_probe lds_bank_probe.hip ; run: ./lds_bank_probe
// Single-wave workgroups (like g2p2g_kernel), 12 per CU, every lane's node index (16-byte units) comes from a table in global memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#include <array>
#include <algorithm>

constexpr int ITERS = 2048;

typedef float v4f __attribute__((ext_vector_type(4)));
template<int MODE>// 0: ds_read_b128, 1: ds_write_b128, 2: read-modify-write b128
__global__ __launch_bounds__(64) void probe(const int* __restrict__ idx, float* out) {
	__shared__ v4f s[1024];
	const int lane = threadIdx.x;
	for(int i = lane; i < 1024; i += 64) s[i] = (v4f) {0.f, 0.f, 0.f, 0.f};
	__syncthreads();
	int n	= idx[lane] & 1023;
	v4f acc = {0.f, 0.f, 0.f, 0.f};
	v4f v	= {(float) lane, 1.f, 2.f, 3.f};
#pragma unroll 1
	for(int it = 0; it < ITERS; ++it) {
#pragma unroll
		for(int o = 0; o < 8; ++o) {
			__asm__ volatile("" : "+v"(n));// (the address is opaque: no hoisting, no merging of the eight accesses)
			v4f* p = &s[n];
			if constexpr(MODE == 0) {
				acc += *p;
			} else if constexpr(MODE == 1) {
				*p = v;
			} else {
				v4f x = *p;
				x += v;
				*p = x;
			}
			__asm__ volatile("" ::: "memory");
		}
	}
	__syncthreads();
	out[blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w + s[lane].x;
}

static int* d_idx;
static float* d_out;
template<int MODE>
static double run(const std::vector<int>& idx) {
	hipMemcpy(d_idx, idx.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
	const int blocks = 256 * 12;
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	probe<MODE><<<blocks, 64>>>(d_idx, d_out);
	hipEventRecord(a);
	probe<MODE><<<blocks, 64>>>(d_idx, d_out);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	hipEventDestroy(a);
	hipEventDestroy(b);
	const double winst = 12.0 * ITERS * 8 * (MODE == 2 ? 2 : 1);// wave-instructions per CU
	return ms * 1e-3 * 2.4e9 / winst;							  // 2.4 GHz periods per wave-instruction per CU
}

static void report(const char* name, const std::vector<int>& idx) {
	printf("%-64s read %6.2f  write %6.2f  rmw(pair) %6.2f\n", name, run<0>(idx), run<1>(idx), run<2>(idx) * 2);
}

int main() {
	hipMalloc(&d_idx, 64 * sizeof(int));
	hipMalloc(&d_out, 256 * 12 * 64 * sizeof(float));
	std::vector<int> idx(64);
	auto fill = [&](std::function<int(int)> f) {
		for(int l = 0; l < 64; ++l) idx[l] = f(l);
	};
	printf("# cycles (2.4 GHz periods) per wave-instruction per CU, 12 single-wave workgroups per CU; node = 16-byte unit, 16 nodes = all 64 banks\n");
	fill([](int l) { return l; });
	report("linear (node = lane)", idx);
	fill([](int) { return 5; });
	report("broadcast (all lanes one node)", idx);
	// ---- which lanes share a pass with lane 0 / 8 / 20 / 40?  All lanes read node 0 except lane i (node 16: same banks, other address) and lane j (node 32)
	for(int i: {0, 8, 20, 40}) {
		printf("lanes that share a pass with lane %d (read cost above the no-conflict value):", i);
		std::vector<double> t(64, 0.0);
		double base = 1e9;
		for(int j = 0; j < 64; ++j) {
			if(j == i) continue;
			fill([&](int l) { return l == i ? 16 : (l == j ? 32 : 0); });
			t[j] = run<0>(idx);
			if(t[j] < base) base = t[j];
		}
		for(int j = 0; j < 64; ++j)
			if(j != i && t[j] > base * 1.15) printf(" %d", j);
		printf("   (base %.2f)\n", base);
	}
	// ---- row patterns: 16 distinct bank groups per group of lanes
	fill([](int l) { return (l & 15) + 16 * (l >> 4); });
	report("groups of 16 consecutive lanes, each a full bank row", idx);
	fill([](int l) { return (l & 15) + 16 * 5 * (l >> 4); });
	report("same, rows 80 nodes apart", idx);
	fill([](int l) { return (l & 7) + 8 * ((l >> 5) & 1) + 16 * ((l >> 3) & 3); });
	report("lanes {0-7, 32-39} a full row, {8-15, 40-47} the next ...", idx);
	fill([](int l) { return (l & 7) + 8 * ((l >> 3) & 1) + 16 * (l >> 4); });
	report("(control) = groups of 16 consecutive lanes", idx);
	fill([](int l) { return (l & 31) + 32 * (l >> 5); });
	report("groups of 32 lanes: two rows each, linear", idx);
	fill([](int l) { return ((l & 31) >> 1) + 16 * (l & 1) + 32 * (l >> 5); });
	report("even lanes row A, odd lanes row B (interleaved arenas)", idx);
	// ---- layout search: lane -> (x, y, z) stencil base in the 6^3 node cube under the kernel's patterns - "rest" (one particle per key and
	//      slice, keys y-major, bases 1..4) and flow-like slices (keys y-major with multiplicities 0..2, bases 0 / 5 rare) -, node = layout(x, y, z)
	//      for the gather arena (one copy) and + parity * arena for the two scatter arenas; cost = 27 reads + 27 read-modify-write pairs
	struct Pattern {
		std::vector<std::array<int, 3>> k;
	};
	std::vector<Pattern> pats;
	{
		Pattern r;
		for(int l = 0; l < 64; ++l) r.k.push_back({1 + ((l >> 2) & 3), 1 + (l >> 4), 1 + (l & 3)});
		pats.push_back(r);
	}
	srand(7);
	for(int rep = 0; rep < 6; ++rep) {
		Pattern f;
		const int y0 = rep % 3;
		for(int y = y0; y < 6 && f.k.size() < 64; ++y)
			for(int x = 0; x < 6; ++x)
				for(int z = 0; z < 6; ++z) {
					const bool rare = x == 0 || x == 5 || y == 0 || y == 5 || z == 0 || z == 5;
					int m			= rare ? (rand() % 12 == 0) : (rand() % 3);
					while(m-- && f.k.size() < 64) f.k.push_back({x, y, z});
				}
		while(f.k.size() < 64) f.k.push_back(f.k.back());
		pats.push_back(f);
	}
	// ---- what a bank-aware placement of a slice's records over its 64 lanes would buy (the kernel's layout: node = 36 x + 6 y + z, second
	//      arena 228 nodes further: bank group = node mod 16, + 4 for odd lanes in the scatter).  Greedy: a record takes a lane of a 16-lane
	//      group in which its scatter bank group is still free; the two records of a key that share a slice go to lanes of different parity.
	auto node0 = [](const std::array<int, 3>& k) { return 36 * k[0] + 6 * k[1] + k[2]; };
	auto place = [&](const Pattern& P) {
		std::vector<int> lane_of(64, -1), used_lane(64, 0);
		int used[4] = {0, 0, 0, 0};
		// order: second members of duplicate keys first fixed to the opposite parity of their partner
		std::vector<int> par_pref(64, -1);
		for(int i = 1; i < 64; ++i)
			if(P.k[i] == P.k[i - 1] && par_pref[i - 1] != 1) par_pref[i - 1] = 0, par_pref[i] = 1;
		for(int pass = 0; pass < 2; ++pass)
			for(int i = 0; i < 64; ++i) {
				if((pass == 0) != (par_pref[i] >= 0) || lane_of[i] >= 0) continue;// constrained records first
				const int g0 = node0(P.k[i]) & 15;
				int best	 = -1;
				for(int q = 0; q < 4 && best < 0; ++q)
					for(int par = 0; par < 2 && best < 0; ++par) {
						if(par_pref[i] >= 0 && par != par_pref[i]) continue;
						const int gs = (g0 + 4 * par) & 15;
						if(used[q] & (1 << gs)) continue;
						for(int j = par; j < 16; j += 2)
							if(!used_lane[16 * q + j]) {
								best = 16 * q + j;
								used[q] |= 1 << gs;
								break;
							}
					}
				if(best >= 0) lane_of[i] = best, used_lane[best] = 1;
			}
		for(int i = 0; i < 64; ++i)
			if(lane_of[i] < 0) {// leftovers: any free lane of the right parity, then any
				for(int pass = 0; pass < 2 && lane_of[i] < 0; ++pass)
					for(int l = 0; l < 64 && lane_of[i] < 0; ++l)
						if(!used_lane[l] && (pass || par_pref[i] < 0 || (l & 1) == par_pref[i])) lane_of[i] = l, used_lane[l] = 1;
			}
		Pattern Q;
		Q.k.resize(64);
		for(int i = 0; i < 64; ++i) Q.k[lane_of[i]] = P.k[i];
		return Q;
	};
	printf("# bank-aware placement (node = 36 x + 6 y + z, second arena at 228): gather read / scatter rmw pair, key-major lanes -> placed lanes\n");
	for(size_t pi = 0; pi < pats.size(); ++pi) {
		const Pattern Q = place(pats[pi]);
		double v[4];
		int c = 0;
		for(const Pattern* P: {static_cast<const Pattern*>(&pats[pi]), static_cast<const Pattern*>(&Q)}) {
			fill([&](int l) { return node0(P->k[l]); });
			v[c++] = run<0>(idx);
			fill([&](int l) { return node0(P->k[l]) + (l & 1) * 228; });
			v[c++] = run<2>(idx) * 2;
		}
		printf("pattern %zu (%s): %6.2f %6.2f  ->  %6.2f %6.2f\n", pi, pi ? "flow-like" : "rest", v[0], v[1], v[2], v[3]);
	}
	return 0;
}
