// mpm_g2p2g_pair.hpp — G2P2G with TWO particles of one stencil base per lane (round 6; the structural step DESIGN.md 7 / VERDICT r5 #1 asked for).
//
// Same fused G2P + particle update + P2G as g2p2g_kernel (mpm_g2p2g.hpp; reference g2p2g, Projects/GMPM/mgmpm_kernels.cuh:665-937 with the
// per-material bodies :422-663), same one-wave-per-block structure, same arenas and bin formats.  What changes is the unit of an iteration: a
// lane carries a PAIR - two records of one predicted stencil base, which the sort hands out two by two (the pair layout of the advection lists,
// mpm_kernels.hpp: pair slices, then the odd records of the keys as single slices).  Then
//   * the two P2G contributions are summed in registers and take ONE read-modify-write per node: 27 pairs of ds_read_b128 / ds_write_b128 per 128
//     particles instead of 54;
//   * the 27 gather nodes are read ONCE for both when the whole wave's pairs share their base (J-fluid; the solid models' second set of gather
//     accumulators does not fit 168 registers);
//   * claims, exec brackets, slice arithmetic, list prefetch: once per 128 particles.
// A pair whose members END the step with different bases splits: A takes the chain, B the serial path (mispredictions only: < 1 %).
// The pair scatters at the end of its OWN iteration, 27 steps back to back: no loop-carried payload, no chain state beside the gather, and so
// 147-168 registers = three waves per SIMD.  (Measured, profiles/r06_ab_pairs_phase1.txt: the chain of pair i - 1 threaded through the update of
// pair i, as g2p2g_kernel does for single particles, needs 231 registers = two waves per SIMD and is 8 % SLOWER than one particle per lane at four
// waves; the unthreaded pair at three waves is 11 % faster - J-fluid, fixed-corotated and sand alike.)
#pragma once
#include "mpm_g2p2g.hpp"

namespace mpm {

#if !(defined(MPM_EXPERIMENT) && defined(MPM_PAIR_WAVES))
#define MPM_PAIR_WAVES 3
#endif
// bit m: material m reads the 27 gather nodes once for both particles when the wave's pairs share their bases (else: one gather per particle)
#if !(defined(MPM_EXPERIMENT) && defined(MPM_PAIR_WAVES_FLUID))
#define MPM_PAIR_WAVES_FLUID 4// (J-fluid: with the late record fetch below the instantiation needs 127 registers: four waves per SIMD, -2 % at rest, -3 % in the flow against three)
#endif
#if !(defined(MPM_EXPERIMENT) && defined(MPM_PAIR_DUAL))
#define MPM_PAIR_DUAL 0// 1: a B that cannot ride with its A claims the other arena and scatters in the same chain (ScatterChainDual) instead of the serial path - measured +6 % in the C3 flow (every iteration with such a lane pays two read-modify-writes per step): off, profiles/r06_ab_pairs_phase2.txt
#endif
#if !(defined(MPM_EXPERIMENT) && defined(MPM_PAIR_LATE_FETCH_FLUID))
#define MPM_PAIR_LATE_FETCH_FLUID 1// the same switch for the J-fluid instantiation (16-byte records: the shorter prefetch distance costs nothing, the registers buy the fourth wave)
#endif
#if !(defined(MPM_EXPERIMENT) && defined(MPM_PAIR_LATE_FETCH_NACC))
#define MPM_PAIR_LATE_FETCH_NACC 1// NACC: with the early fetch the instantiation reloads a dozen spilled constants per iteration; with the late one it needs 165 registers and none
#endif
#if !(defined(MPM_EXPERIMENT) && defined(MPM_PAIR_LATE_FETCH))
#define MPM_PAIR_LATE_FETCH 0// 0: the next slice's particle records are requested at the top of the iteration; 1: behind the material update; 2: A's at the top, B's behind the material update
#endif
#if !(defined(MPM_EXPERIMENT) && defined(MPM_PAIR_SHARED_GATHER))
#define MPM_PAIR_SHARED_GATHER 0x0// (the second set of gather accumulators does not fit 168 registers beside the slice bookkeeping: scratch operations inside the loop - every one drains the record prefetch - cost more than 27 LDS reads)
#endif

// Tensor-product APIC gather (gather_apic, mpm_g2p2g.hpp) for two particles that share their stencil base: every node is loaded once.
MPM_DEV void gather_apic_shared(const float4* __restrict__ gbase, const float (&w)[2][3][3], const float (&fd)[2][3], float (&vel)[2][3], float (&A)[2][9]) {
	v2f_ wz[2][3], wy[2][3], wx[2][3];
#pragma unroll
	for(int h = 0; h < 2; ++h)
#pragma unroll
		for(int t = 0; t < 3; ++t) {
			wx[h][t] = (v2f_) {w[h][0][t], w[h][0][t] * ((float) t - fd[h][0])};
			wy[h][t] = (v2f_) {w[h][1][t], w[h][1][t] * ((float) t - fd[h][1])};
			wz[h][t] = (v2f_) {w[h][2][t], w[h][2][t] * ((float) t - fd[h][2])};
		}
	v2f_ vel_xy[2], A0_xy[2], A3_xy[2], A6_xy[2], velz_A2[2];
	float A5[2], A8[2];
#pragma unroll
	for(int h = 0; h < 2; ++h) {
		vel_xy[h] = A0_xy[h] = A3_xy[h] = A6_xy[h] = velz_A2[h] = (v2f_) {0.f, 0.f};
		A5[h] = A8[h] = 0.f;
	}
#pragma unroll
	for(int i = 0; i < 3; ++i) {
		v2f_ u0_xy[2], uy_xy[2], uz_xy[2], u0z_uyz[2];
		float uzz[2];
#pragma unroll
		for(int h = 0; h < 2; ++h) {
			u0_xy[h] = uy_xy[h] = uz_xy[h] = u0z_uyz[h] = (v2f_) {0.f, 0.f};
			uzz[h]										= 0.f;
		}
#pragma unroll
		for(int j = 0; j < 3; ++j) {
			v2f_ t0_xy[2], t1_xy[2], t0z_t1z[2];
#pragma unroll
			for(int h = 0; h < 2; ++h) t0_xy[h] = t1_xy[h] = t0z_t1z[h] = (v2f_) {0.f, 0.f};
#pragma unroll
			for(int k = 0; k < 3; ++k) {
				const float4 v = gbase[i * kG2PStrideX + j * kG2PStrideY + k * kG2PStrideZ];
				const v2f_ vxy = {v.x, v.y}, vzz = {v.z, v.w};
#pragma unroll
				for(int h = 0; h < 2; ++h) {
					t0_xy[h]   = vxy * wz[h][k].x + t0_xy[h];
					t1_xy[h]   = vxy * wz[h][k].y + t1_xy[h];
					t0z_t1z[h] = wz[h][k] * vzz + t0z_t1z[h];
				}
			}
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for(int h = 0; h < 2; ++h) {
				u0_xy[h]   = t0_xy[h] * wy[h][j].x + u0_xy[h];
				uy_xy[h]   = t0_xy[h] * wy[h][j].y + uy_xy[h];
				uz_xy[h]   = t1_xy[h] * wy[h][j].x + uz_xy[h];
				u0z_uyz[h] = wy[h][j] * t0z_t1z[h].x + u0z_uyz[h];
				uzz[h] += wy[h][j].x * t0z_t1z[h].y;
			}
		}
#pragma unroll
		for(int h = 0; h < 2; ++h) {
			vel_xy[h]  = u0_xy[h] * wx[h][i].x + vel_xy[h];
			A0_xy[h]   = u0_xy[h] * wx[h][i].y + A0_xy[h];
			A3_xy[h]   = uy_xy[h] * wx[h][i].x + A3_xy[h];
			A6_xy[h]   = uz_xy[h] * wx[h][i].x + A6_xy[h];
			velz_A2[h] = wx[h][i] * u0z_uyz[h].x + velz_A2[h];
			A5[h] += wx[h][i].x * u0z_uyz[h].y;
			A8[h] += wx[h][i].x * uzz[h];
		}
	}
#pragma unroll
	for(int h = 0; h < 2; ++h) {
		vel[h][0] = vel_xy[h].x;
		vel[h][1] = vel_xy[h].y;
		vel[h][2] = velz_A2[h].x;
		A[h][0]	  = A0_xy[h].x;
		A[h][1]	  = A0_xy[h].y;
		A[h][2]	  = velz_A2[h].y;
		A[h][3]	  = A3_xy[h].x;
		A[h][4]	  = A3_xy[h].y;
		A[h][5]	  = A5[h];
		A[h][6]	  = A6_xy[h].x;
		A[h][7]	  = A6_xy[h].y;
		A[h][8]	  = A8[h];
#pragma unroll
		for(int d = 0; d < 9; ++d) __asm__ volatile("" : "+v"(A[h][d]));
#pragma unroll
		for(int d = 0; d < 3; ++d) __asm__ volatile("" : "+v"(vel[h][d]));
	}
}

// One particle's share of a chain step: B-spline weights of the new position and the incremental walk of the affine momentum term
// (ScatterChain, mpm_g2p2g.hpp).  `on` = false zeroes the x weights: the particle contributes nothing (its pair split).
struct ChainHalf {
	float pw[3][3];
	float cx0, cy0, cz0;
	v2f_ cx12, cy12, cz12;
	float slab0, pen0, wij;
	v2f_ slab12, pen12;
	MPM_DEV void init(const P2GPayload& p, bool on) {
#pragma unroll
		for(int d = 0; d < 3; ++d) bspline_weight_cells(p.fd[d], pw[d]);
		if(!on) pw[0][0] = pw[0][1] = pw[0][2] = 0.f;
		cx0	 = p.contrib[0], cy0 = p.contrib[3], cz0 = p.contrib[6];
		cx12 = (v2f_) {p.contrib[1], p.contrib[2]}, cy12 = (v2f_) {p.contrib[4], p.contrib[5]}, cz12 = (v2f_) {p.contrib[7], p.contrib[8]};
		slab0  = p.mv[0] - cx0 * p.fd[0] - cy0 * p.fd[1] - cz0 * p.fd[2];
		slab12 = (v2f_) {p.mv[1], p.mv[2]} - cx12 * p.fd[0] - cy12 * p.fd[1] - cz12 * p.fd[2];
	}
	// the part of a step that does not need the node's accumulator: weight and the two channel pairs {mass, x}, {y, z} of this particle at stencil offset o
	MPM_DEV void prep(int o, float mass, float& W, v2f_& m0, v2f_& t12) {// o: compile-time after unrolling
		const int i = o / 9, j = (o / 3) % 3, k = o % 3;
		if(k == 0) {
			if(j == 0) {
				if(i != 0) {
					slab0 += cx0;
					slab12 += cx12;
				}
				pen0  = slab0;
				pen12 = slab12;
			} else {
				pen0 += cy0;
				pen12 += cy12;
			}
			wij = pw[0][i] * pw[1][j];
		}
		W = wij * pw[2][k];
		// (k = 2: fma with the constant 2 - the doubled z row of ScatterChain would cost three registers per particle)
		m0	= (v2f_) {mass, k == 0 ? pen0 : (k == 1 ? pen0 + cz0 : fmaf(2.f, cz0, pen0))};
		t12 = k == 0 ? pen12 : (k == 1 ? pen12 + cz12 : cz12 * 2.f + pen12);
	}
	MPM_DEV void add(int o, float mass, v2f_& a01, v2f_& a23) {
		float W;
		v2f_ m0, t12;
		prep(o, mass, W, m0, t12);
		a01 = m0 * W + a01;
		a23 = t12 * W + a23;
	}
};

// The chain for an iteration in which some lane's B could NOT ride with its A (another stencil base: the sort's mismatched slots, a misprediction, an A
// on the cube's edge) but holds a claim of its own in the OTHER arena: such a lane does two read-modify-writes per step, A's node in its arena and B's
// node in the other one (different arenas: the two may be in flight together), instead of sending B down the serial path - which costs five times a
// chain particle (profiles/r06_ab_pairs_phase2.txt: 2 % of the particles of the C3 flow, 0.2 of 1.98 ms).  Lanes whose B rides with A (`merge`) add it
// into A's accumulator as before.  Only instantiated behind a wave-uniform test: an iteration without such a lane runs ScatterChain2.
struct ScatterChainDual {
	float4 *node0, *node1;
	float mass, merge_w, dual_w;// 1 / 0: B's contribution goes into A's node / into its own
	int win, dual;
	ChainHalf h[2];
	float4 acc, acc2;
	MPM_DEV ScatterChainDual(float4* n0, float4* n1, const P2GPayload& pa, const P2GPayload& pb, float m, bool w, bool merge, bool d)
		: node0(n0)
		, node1(n1)
		, mass(m)
		, merge_w(merge ? 1.f : 0.f)
		, dual_w(d ? 1.f : 0.f)
		, win(w)
		, dual(d) {
		h[0].init(pa, w);
		h[1].init(pb, merge || d);
		acc = acc2 = make_float4(0.f, 0.f, 0.f, 0.f);
		if(win) acc = node0[0];
		if(dual) acc2 = node1[0];
	}
	MPM_DEV void run() {
#pragma unroll
		for(int o = 0; o < 27; ++o) {
			v2f_ a01 = {acc.x, acc.y}, a23 = {acc.z, acc.w};
			v2f_ b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
			h[0].add(o, mass, a01, a23);
			h[1].add(o, mass, b01, b23);
			a01 = b01 * merge_w + a01;
			a23 = b23 * merge_w + a23;
			const v2f_ c01 = b01 * dual_w + (v2f_) {acc2.x, acc2.y}, c23 = b23 * dual_w + (v2f_) {acc2.z, acc2.w};
			const int i = o / 9, j = (o / 3) % 3, k = o % 3;
			const int off = i * kP2GStrideX + j * kP2GStrideY + k;
			if(win) node0[off] = make_float4(a01.x, a01.y, a23.x, a23.y);
			if(dual) node1[off] = make_float4(c01.x, c01.y, c23.x, c23.y);
			__asm__ volatile("" ::: "memory");
			if(o + 1 < 27) {
				const int i1 = (o + 1) / 9, j1 = ((o + 1) / 3) % 3, k1 = (o + 1) % 3;
				const int off1 = i1 * kP2GStrideX + j1 * kP2GStrideY + k1;
				if(win) acc = node0[off1];
				if(dual) acc2 = node1[off1];
			}
		}
	}
};

// The scatter chain for a PAIR: per node the two contributions are summed in registers, one read-modify-write (cf. ScatterChain).
#if !defined(MPM_EXPERIMENT) || !defined(MPM_CHAIN_ASM)
#define MPM_CHAIN_ASM 1
#endif
template<int NSITES>
struct ScatterChain2 {
	float4* node0;
	float mass;
	int win;
	ChainHalf h[2];
#if MPM_CHAIN_ASM
	// A step is an LDS round trip: the read of node o can only be issued behind the write of node o - 1 (another lane may have written that very node), and what
	// the wave can do meanwhile is the part of step o that does not need the accumulator - both particles' weights and channel values (prep).  The compiler put only
	// A's half in front of its wait (B's went behind A's multiply-adds); the read is issued and awaited by hand here (lds_issue_b128 / s_waitcnt with the prepared
	// values as inputs, so that they are formed first), as in gather_apic.
	unsigned lds0;
	v4f_ acc;
	MPM_DEV ScatterChain2(float4* n0, const P2GPayload& pa, const P2GPayload& pb, float m, bool w, bool with_b)
		: node0(n0)
		, mass(m)
		, win(w)
		, lds0((unsigned) (size_t) n0) {
		h[0].init(pa, true);
		h[1].init(pb, with_b);
		if(win) acc = lds_issue_b128<0>(lds0);
	}
	template<int O>
	MPM_DEV void step() {
		float wa, wb;
		v2f_ m0a, t12a, m0b, t12b;
		h[0].prep(O, mass, wa, m0a, t12a);
		h[1].prep(O, mass, wb, m0b, t12b);
		__asm__ volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc) : "v"(wa), "v"(m0a), "v"(t12a), "v"(wb), "v"(m0b), "v"(t12b));
		v2f_ a01 = {acc.x, acc.y};
		v2f_ a23 = {acc.z, acc.w};
		a01 = m0a * wa + a01;
		a23 = t12a * wa + a23;
		a01 = m0b * wb + a01;
		a23 = t12b * wb + a23;
		constexpr int i = O / 9, j = (O / 3) % 3, k = O % 3;
		node0[i * kP2GStrideX + j * kP2GStrideY + k] = make_float4(a01.x, a01.y, a23.x, a23.y);
		__asm__ volatile("" ::: "memory");
		if constexpr(O + 1 < 27) {
			constexpr int i1 = (O + 1) / 9, j1 = ((O + 1) / 3) % 3, k1 = (O + 1) % 3;
			acc = lds_issue_b128<(i1 * kP2GStrideX + j1 * kP2GStrideY + k1) * 16>(lds0);
		}
	}
	template<int O, int END>
	MPM_DEV void steps() {
		if constexpr(O < END) {
			step<O>();
			steps<O + 1, END>();
		}
	}
	template<int SITE>
	MPM_DEV void at() {
		static_assert(SITE >= 0 && SITE < NSITES, "site out of range");
		if(win) steps<SITE * 27 / NSITES, (SITE + 1) * 27 / NSITES>();
	}
#else
	float4 acc;
	MPM_DEV ScatterChain2(float4* n0, const P2GPayload& pa, const P2GPayload& pb, float m, bool w, bool with_b)
		: node0(n0)
		, mass(m)
		, win(w) {
		h[0].init(pa, true);
		h[1].init(pb, with_b);
		if(win) acc = node0[0];
	}
	MPM_DEV void step(int o) {
		v2f_ a01 = {acc.x, acc.y};
		v2f_ a23 = {acc.z, acc.w};
		h[0].add(o, mass, a01, a23);
		h[1].add(o, mass, a01, a23);
		const int i = o / 9, j = (o / 3) % 3, k = o % 3;
		node0[i * kP2GStrideX + j * kP2GStrideY + k] = make_float4(a01.x, a01.y, a23.x, a23.y);
		__asm__ volatile("" ::: "memory");
		if(o + 1 < 27) {
			const int i1 = (o + 1) / 9, j1 = ((o + 1) / 3) % 3, k1 = (o + 1) % 3;
			acc			 = node0[i1 * kP2GStrideX + j1 * kP2GStrideY + k1];
		}
	}
	template<int SITE>
	MPM_DEV void at() {
		static_assert(SITE >= 0 && SITE < NSITES, "site out of range");
		if(win) {
#pragma unroll
			for(int o = SITE * 27 / NSITES; o < (SITE + 1) * 27 / NSITES; ++o) step(o);
		}
	}
#endif
};

template<int MAT>
__global__ __launch_bounds__(kG2P2GThreads, MAT == 0 ? MPM_PAIR_WAVES_FLUID : MPM_PAIR_WAVES) void g2p2g_pair_kernel(GridCfg cfg, ModelView mv, const int* __restrict__ cur_keys, const float* __restrict__ grid, float* __restrict__ next_grid, const int* __restrict__ block_list, const int* __restrict__ only_flag, const int* __restrict__ nblocks_ptr, int nblocks, float dt, float new_dt, StepConst sk, int* __restrict__ status) {
	constexpr int NCH = MatTraits<MAT>::nch;
	constexpr int REC = MatTraits<MAT>::rec;
	constexpr bool kSharedGather = ((MPM_PAIR_SHARED_GATHER >> MAT) & 1) != 0;
	constexpr int kLateFetch	 = MAT == 0 ? MPM_PAIR_LATE_FETCH_FLUID : (MAT == 3 ? MPM_PAIR_LATE_FETCH_NACC : MPM_PAIR_LATE_FETCH);
	__shared__ float4 g2p[kG2PNodes];
	__shared__ float4 p2g[kP2GArena2 + kP2GNodes];
	__shared__ unsigned char s_owner[2 * 216];
	constexpr bool kQueue = MAT != 0 && kSerialQueue > 0;
	__shared__ float4 s_queue[kQueue ? 4 * kSerialQueue : 1];

	const int lane0 = threadIdx.x;
	const int total = nblocks_ptr ? min(*nblocks_ptr, cfg.cap) : nblocks;
	const int xcd = (int) (blockIdx.x & 7u), nq = nblocks_ptr ? (int) (gridDim.x >> 3) : 0x40000000;
	const int share = (total >> 3) + (xcd < (total & 7) ? 1 : 0);
	const int first = xcd * (total >> 3) + min(xcd, total & 7);
	for(int q = (int) (blockIdx.x >> 3); q < share; q += nq) {
	int lane = lane0;
	__asm__ volatile("" : "+v"(lane));
	const int bid = first + q;
	const int b	  = block_list ? block_list[bid] : bid;
	const int size		 = mv.size[b];
	const int row		 = mv.row_of[b];
	const int binoff_dst = mv.binoff_dst[b];
	const int flag		 = only_flag ? only_flag[b] : 1;
	if(size == 0 || flag == 0) continue;
	const int* list		= mv.list_in + (size_t) row * cfg.ppb;
	const float mass	= mv.mc.mass;
	const int key_shift = cfg.pid_bits;
	const int tag_shift = cfg.pid_bits + kKeyBits;
	const int info		= mv.blockinfo[(size_t) b * kInfoRow + lane];
	int chunk_nrec;// records of chunk (lane & 15) of the block; 0 beyond its last chunk (the last chunk absorbs a short tail: pair_chunks, mpm_kernels.hpp)
	{
		const int nch = pair_chunks(size), cl = lane & (kPairChunks - 1);
		chunk_nrec	  = cl < nch ? (cl + 1 < nch ? 512 : size - 512 * cl) : 0;
	}
	// full pairs of chunk (lane & 15) | its records << 16 (by block number: the same round trip as the scalars above; the records are formed per lane here, once per
	// block: formed from `size` where they are used they cost scalar registers across the particle loop - spills into vector lanes, +113 v_readlane inside it)
	const int pinfo		= (mv.pairinfo_in[(size_t) b * kPairChunks + (lane & (kPairChunks - 1))] & 0xffff) | (chunk_nrec << 16);// (the entries beyond the last chunk were never written)
	// ---- the slices of the block in the pair layout (mpm_kernels.hpp).  Lane t forms the descriptor of slice t (of slice 64 k + t in the k-th batch of
	//      a block with more than 64 slices) once per block: position of A's first record in the block's list, lanes in use (0: beyond the end), pair
	//      slice or single slice.  The loop reads the descriptor two slices ahead with v_readlane: no scalar cursor to carry (a scalar one took ~100
	//      scalar instructions per iteration and ~25 live scalar registers, i.e. spills into vector registers).
	int d_slice = 0;// position | lanes with an A << 16 | lanes with a B << 24 (one register: the loop is short of them)
	auto form_slices = [&](int first) {// descriptors of slices first .. first + 63
		int before = 0;// slices of the chunks before c (wave-uniform)
		d_slice = 0;
#pragma unroll 1
		for(int c = 0; c < kPairChunks; ++c) {
			const int pe = __builtin_amdgcn_readlane(pinfo, c);
			if((pe >> 16) == 0) break;// (beyond the block's last chunk)
			const PairChunk pc = pair_chunk(pe >> 16, pe & 0xffff);
			const int t		   = first + lane - before;
			if(t >= 0 && t < pc.S) {
				int pos, ca, cb;
				pair_slice(pc, t, pos, ca, cb);
				d_slice = (c * kListChunk + pos) | (ca << 16) | (cb << 24);
			}
			before += pc.S;
		}
	};
	form_slices(0);
	struct Slice {
		int pos, cnt, cnt_b;// position of A's first record in the block's list; lanes with an A (0: beyond the end); lanes with a B (the first ones; B's records `cnt` behind A's)
	};
	auto read_slice = [&](int t, Slice& sl) {// t wave-uniform, inside the current batch
		const int d = __builtin_amdgcn_readlane(d_slice, t & 63);
		sl			= Slice {d & 0xffff, (d >> 16) & 255, d >> 24};
	};
	// (idle lanes re-read the last record of their member's run: same inputs, nothing stored; a slice without B's: the lane re-reads A's record;
	//  beyond the end: the block's first record, never processed)
	auto load_recs = [&](const Slice& sl, int (&rec)[2]) {
		rec[0] = list[sl.pos + min(lane, max(sl.cnt, 1) - 1)];
		rec[1] = list[sl.pos + (sl.cnt_b ? sl.cnt + min(lane, sl.cnt_b - 1) : min(lane, max(sl.cnt, 1) - 1))];
	};
	Slice s_cur, s_next;
	int rec_cur[2], rec_next[2];
	read_slice(0, s_cur);
	read_slice(1, s_next);
	load_recs(s_cur, rec_cur);
	load_recs(s_next, rec_next);
	for(int i = lane; i < kP2GArena2 + kP2GNodes; i += 64) p2g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
	__syncthreads();
	float4 gv[8];
#pragma unroll
	for(int lb = 0; lb < 8; ++lb) {
		const int nb	= __shfl(info, 54 + lb);
		const float* gb = grid + (size_t) (nb < 0 ? 0 : nb) * 256;
		gv[lb].x		= gb[64 + lane];
		gv[lb].y		= gb[128 + lane];
		gv[lb].z		= gb[192 + lane];
		if(nb < 0) gv[lb].x = gv[lb].y = gv[lb].z = 0.f;
		gv[lb].w = gv[lb].z;
	}
	constexpr int ROW = NCH - REC;
	struct Prefetch {
		float4 q[REC / 4];
		float row[ROW ? ROW : 1];
		int key;
	};
	// the source bin of a record: the bin offset of its source block sits in lane `tag` of the block's info row (one ds_bpermute, an LDS round trip)
	auto source_bin = [&](int rec) { return __shfl(info, (rec >> tag_shift) & 31) + ((rec & (cfg.ppb - 1)) >> 6); };
	auto fetch_from = [&](int rec, int sbin, Prefetch& f) {
		const int sp	  = rec & (cfg.ppb - 1);
		const float* bin  = mv.bins_src + (size_t) sbin * (kBin * NCH);
		const float4* src = reinterpret_cast<const float4*>(bin + (sp & 63) * REC);
		f.key			  = ((rec >> key_shift) & 255) | ((rec >> kArenaBit) & 1) << 8;// sort key | the slot's scatter arena (pair layout, mpm_kernels.hpp)
#pragma unroll
		for(int d = 0; d < REC / 4; ++d) f.q[d] = src[d];
		if constexpr(ROW == 1) f.row[0] = bin[kBin * REC + (sp & 63)];
		if constexpr(ROW == 2) {
			const float2 t = reinterpret_cast<const float2*>(bin + kBin * REC)[sp & 63];
			f.row[0]	   = t.x;
			f.row[1]	   = t.y;
		}
	};
	auto fetch = [&](int rec, Prefetch& f) { fetch_from(rec, source_bin(rec), f); };
	// both members of a slot: the two look-ups are in flight together (issued one after the other they were two exposed round trips at the top of every iteration)
	auto fetch2 = [&](const int (&rec)[2], Prefetch (&f)[2]) {
		int sb[2] = {source_bin(rec[0]), source_bin(rec[1])};
		__asm__ volatile("" : "+v"(sb[0]), "+v"(sb[1]));
		fetch_from(rec[0], sb[0], f[0]);
		fetch_from(rec[1], sb[1], f[1]);
	};
	Prefetch pf[2];
	fetch2(rec_cur, pf);
	// (kLateFetch == 0) the source bins of the NEXT slice's records are looked up one phase ahead - in front of the scatter chain, whose 27 round trips cover this one -
	// and carried into the next iteration, whose record loads then start at once
	int sb_next[2] = {0, 0};
	if constexpr(kLateFetch == 0) sb_next[0] = source_bin(rec_next[0]), sb_next[1] = source_bin(rec_next[1]);
#pragma unroll
	for(int lb = 0; lb < 8; ++lb) {
		const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;
		const int ax = cx + ((lb & 4) ? 4 : 0) - 1, ay = cy + ((lb & 2) ? 4 : 0) - 1, az = cz + ((lb & 1) ? 4 : 0) - 1;
		if(((unsigned) ax < 6u) & ((unsigned) ay < 6u) & ((unsigned) az < 6u)) g2p[ax * kG2PStrideX + ay * kG2PStrideY + az * kG2PStrideZ] = gv[lb];
	}
	__syncthreads();
	int qn		 = 0;
	bool settled = true;
#ifdef MPM_G2P2G_STATS
	int st_iter = 0, st_losers = 0, st_edge = 0, st_retry_iters = 0, st_partial = 0, st_split = 0;
#endif
	for(int t_cur = 0;; ++t_cur) {
		MPM_MARK("P_top");
		const bool active[2] = {lane < s_cur.cnt, lane < s_cur.cnt_b};
		// slot in the destination bins == position in the sorted order.  An idle lane carries a copy of the last record of its member's run (of A's, in a
		// slice without B's) and stores the same values to the same slot: the stores stay out of divergent control flow, and A's and B's instruction
		// streams stay in ONE basic block - the scheduler interleaves them (with B's half behind wave-uniform branches the kernel was 8 % slower at rest)
		const int pidib[2] = {s_cur.pos + min(lane, s_cur.cnt - 1), s_cur.pos + (s_cur.cnt_b ? s_cur.cnt + min(lane, s_cur.cnt_b - 1) : min(lane, s_cur.cnt - 1))};
#ifdef MPM_G2P2G_STATS
		st_partial += 128 - s_cur.cnt - s_cur.cnt_b;
#endif
		float pos[2][3], st[2][7];
		int okey[2];
#pragma unroll
		for(int h = 0; h < 2; ++h) {
			pos[h][0] = pf[h].q[0].x, pos[h][1] = pf[h].q[0].y, pos[h][2] = pf[h].q[0].z;
			st[h][0] = pf[h].q[0].w;
			if constexpr(MAT != 0) {
				st[h][1] = pf[h].q[1].x;
				st[h][2] = pf[h].q[1].y;
				st[h][3] = pf[h].q[1].z;
				st[h][4] = pf[h].q[1].w;
				st[h][5] = pf[h].row[0];
				if constexpr(ROW == 2) st[h][6] = pf[h].row[1];
			}
			okey[h] = pf[h].key & 255;
		}
		const int arena_sel = pf[0].key >> 8;// the slot's scatter arena
		// the list records two slices ahead, the particle records one slice ahead
		Slice s_nn;
		int rec_nn[2];
		if(((t_cur + 2) & 63) == 0) form_slices(t_cur + 2);// (a block with more than 64 slices: the next batch of descriptors)
		read_slice(t_cur + 2, s_nn);
		load_recs(s_nn, rec_nn);
		if constexpr(kLateFetch == 0) {
			fetch_from(rec_next[0], sb_next[0], pf[0]);
			fetch_from(rec_next[1], sb_next[1], pf[1]);
		}
		if constexpr(kLateFetch == 2) fetch(rec_next[0], pf[0]);
		MPM_MARK("P_gather");
		// ---- stencil bases + weights (:774-797), gather (:801-835)
		int base[2][3], arena[2][3];
		float vel[2][3], A[2][9];
		{
			float fd[2][3], w[2][3][3];
			bool same = true;
#pragma unroll
			for(int h = 0; h < 2; ++h)
#pragma unroll
				for(int d = 0; d < 3; ++d) {
					const float p = pos[h][d];
					base[h][d]	  = lround_pos(p) - 1;
					fd[h][d]	  = p - (float) base[h][d];
					bspline_weight_cells(fd[h][d], w[h][d]);
					arena[h][d] = ((base[h][d] - 1) & 3) + 1;
				}
#pragma unroll
			for(int d = 0; d < 3; ++d) same &= arena[0][d] == arena[1][d];
			if(kSharedGather && __all(same)) {
				gather_apic_shared(g2p + (arena[0][0] - 1) * kG2PStrideX + (arena[0][1] - 1) * kG2PStrideY + (arena[0][2] - 1) * kG2PStrideZ, w, fd, vel, A);
			} else {
#pragma unroll
				for(int h = 0; h < 2; ++h) gather_apic(g2p + (arena[h][0] - 1) * kG2PStrideX + (arena[h][1] - 1) * kG2PStrideY + (arena[h][2] - 1) * kG2PStrideZ, w[h], fd[h], vel[h], A[h]);
			}
		}
		MPM_MARK("P_rebucket");
		// ---- advect (:838), new base, re-bucket (:852-866, add_advection particle_buffer.cuh:100-135)
		float nfd[2][3];
		int narena[2][3], dirv[2][3], pk[2][3];
		bool in_arena[2];
		const float pred = sk.pred;
#pragma unroll
		for(int h = 0; h < 2; ++h) {
			in_arena[h] = active[h];
#pragma unroll
			for(int d = 0; d < 3; ++d) {
				pos[h][d]		= fmaf(vel[h][d], sk.dtp, pos[h][d]);
				const float p	= pos[h][d];
				const int nbase = lround_pos(p) - 1;
				nfd[h][d]		= p - (float) nbase;
				narena[h][d]	= arena[h][d] + (nbase - base[h][d]);
				in_arena[h] &= (narena[h][d] >= 0) & (narena[h][d] <= 5);
				dirv[h][d]	   = -((narena[h][d] - 1) >> 2);
				const int step = (int) __builtin_rintf(fmaf(vel[h][d], pred, nfd[h][d]));
				pk[h][d]	   = min(max(((narena[h][d] - 1) & 3) + step, 0), 5);
			}
		}
		int ntag[2], dno[2], stay_rank[2], raw_move[2] = {0, 0};
		bool stay[2];
		int stay_leader, raw_stay = 0;
		int b_opaque = b;
		__asm__("" : "+v"(b_opaque));
		if(__all((dirv[0][0] | dirv[0][1] | dirv[0][2] | dirv[1][0] | dirv[1][1] | dirv[1][2]) == 0)) {
			// every particle of the iteration stays in this block: one atomic for all of them (the active lanes are the first lanes)
			const int n_a = s_cur.cnt, n_b = s_cur.cnt_b;
#pragma unroll
			for(int h = 0; h < 2; ++h) {
				ntag[h] = kStay;
				dno[h]	= active[h] ? b : -1;
				stay[h] = active[h];
			}
			stay_leader	 = 0;
			stay_rank[0] = lane;
			stay_rank[1] = n_a + lane;
			if(lane == 0) raw_stay = atomicAdd(&mv.out_count[b_opaque], n_a + n_b);
		} else {
			unsigned long long stay_m[2];
#pragma unroll
			for(int h = 0; h < 2; ++h) {
				const bool dir_ok = ((unsigned) (dirv[h][0] + 1) < 3u) & ((unsigned) (dirv[h][1] + 1) < 3u) & ((unsigned) (dirv[h][2] + 1) < 3u);
				ntag[h]			  = dir_ok ? (dirv[h][0] + 1) * 9 + (dirv[h][1] + 1) * 3 + dirv[h][2] + 1 : kStay;
				const int dno_raw = __shfl(info, 27 + ntag[h]);
				dno[h]			  = (active[h] && dir_ok) ? dno_raw : -1;
				stay[h]			  = dno[h] >= 0 && ntag[h] == kStay;
				stay_m[h]		  = __ballot(stay[h]);
				stay_rank[h]	  = (int) __builtin_amdgcn_mbcnt_hi((unsigned) (stay_m[h] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) stay_m[h], 0u));
			}
			stay_rank[1] += __popcll(stay_m[0]);
			const unsigned long long any_m = stay_m[0] | stay_m[1];
			stay_leader					   = any_m ? __ffsll((long long) any_m) - 1 : 0;
			if(any_m != 0ull && lane == stay_leader) raw_stay = atomicAdd(&mv.out_count[b_opaque], __popcll(stay_m[0]) + __popcll(stay_m[1]));
#pragma unroll
			for(int h = 0; h < 2; ++h) {
				if(dno[h] >= 0 && !stay[h]) raw_move[h] = atomicAdd(&mv.out_count[dno[h]], 1);
				if(active[h]) {
					if(dno[h] < 0) atomicAdd(&status[ST_LOST], 1);// reference: particle silently lost (particle_buffer.cuh:105-113)
					if(!in_arena[h]) atomicAdd(&status[ST_ARENA], 1);// (:877-885) contribution discarded
				}
			}
		}
		int pkey[2], rec[2];
#pragma unroll
		for(int h = 0; h < 2; ++h) {
			pkey[h] = pk[h][1] * 36 + pk[h][0] * 6 + pk[h][2];
			rec[h]	= (ntag[h] << tag_shift) | (pkey[h] << key_shift) | pidib[h] | (arena_sel << kArenaBit);// (the arena bit matters only if the block stays settled: the next sort is then skipped)
		}
		settled = settled && __all((!active[0] || (stay[0] && pkey[0] == okey[0])) && (!active[1] || (stay[1] && pkey[1] == okey[1])));
		MPM_MARK("P_material");
		// ---- material update, store to the destination bin (:470-663)
		float contrib[2][9];// -P F^T vol new_dt D^-1 dx
		NoHook nohook;
		auto material = [&](auto hc) {
			constexpr int H = decltype(hc)::value;
			float* dbin		= mv.bins_dst + (size_t) (binoff_dst + (pidib[H] >> 6)) * (kBin * NCH);
			float4* dst		= reinterpret_cast<float4*>(dbin + (pidib[H] & 63) * REC);
			if constexpr(MAT == 0) {
				const float J = stress_jfluid(mv.mc, sk.ss.vol, sk.jdiv, sk.jvisc, st[H][0], A[H], contrib[H]);
				dst[0] = make_float4(pos[H][0], pos[H][1], pos[H][2], J);
			} else {
				float G[9], bo[6], bn[6];
#pragma unroll
				for(int d = 0; d < 9; ++d) G[d] = A[H][d] * sk.dts + ((d & 0x3) != 0 ? 0.f : 1.f);
				bool refl = (__float_as_uint(st[H][0]) & kReflBit) != 0u;
				bo[0]	  = fabsf(st[H][0]);
#pragma unroll
				for(int d = 1; d < 6; ++d) bo[d] = st[H][d];
				{
					const float amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(fabsf(A[H][0]), fabsf(A[H][1])), fabsf(A[H][2])), __builtin_fmaxf(__builtin_fmaxf(fabsf(A[H][3]), fabsf(A[H][4])), fabsf(A[H][5]))),
													   __builtin_fmaxf(__builtin_fmaxf(fabsf(A[H][6]), fabsf(A[H][7])), fabsf(A[H][8])));
					const bool wild	 = !(amax < sk.refl_lim);
					if(__any(wild)) {
						if(wild) refl ^= det3(G) < 0.f;
					}
				}
				push_forward(G, bo, bn);
				float lj = 0.f;
				if constexpr(MAT == 1) {
					stress_fixed_corotated<0>(sk.ss, bn, refl, contrib[H], nohook);
				} else if constexpr(MAT == 2) {
					lj = st[H][6];
					stress_sand<0>(mv.mc, sk.ss, bn, refl, lj, contrib[H], nohook);
				} else {
					lj = st[H][6];
					stress_nacc<0>(mv.mc, sk.ss, bn, refl, lj, contrib[H], nohook);
				}
				dst[0] = make_float4(pos[H][0], pos[H][1], pos[H][2], refl ? -bn[0] : bn[0]);
				dst[1] = make_float4(bn[1], bn[2], bn[3], bn[4]);
				if constexpr(ROW == 1) dbin[kBin * REC + (pidib[H] & 63)] = bn[5];
				if constexpr(ROW == 2) reinterpret_cast<float2*>(dbin + kBin * REC)[pidib[H] & 63] = make_float2(bn[5], lj);
			}
		};
		material(std::integral_constant<int, 0> {});
		material(std::integral_constant<int, 1> {});
		MPM_MARK("P_append");
		{
			const int basev = __shfl(raw_stay, stay_leader);
#pragma unroll
			for(int h = 0; h < 2; ++h) {
				if(dno[h] >= 0) {
					const int slot = stay[h] ? basev + stay_rank[h] : raw_move[h];
					if(slot >= cfg.ppb)
						atomicOr(&status[ST_OVERFLOW], 2);// reference drops beyond 128 per cell (:122-130)
					else
						mv.list_out[((size_t) dno[h] << cfg.pid_bits) + slot] = rec[h];
				}
			}
		}
		// (kLateFetch) The next slice's particle records are requested HERE, behind the material update, not at the top of the iteration: requested at the top, their
		// 22 destination registers (sand) are live through the gather and the material update, the allocator runs out, parks the loads in registers it
		// needs again and waits for them - s_waitcnt vmcnt(0) in the re-bucketing, i.e. the whole HBM latency exposed in every iteration.  From here the
		// loads have the scatter chain (27 LDS round trips) and the other two waves of the SIMD to arrive in.
		if constexpr(kLateFetch == 1) fetch2(rec_next, pf);
		if constexpr(kLateFetch == 2) fetch(rec_next[1], pf[1]);
		MPM_MARK("P_scatter");
		// ---- the pair scatters now (:887-905): payload ((:850) contrib = (A m - stress new_dt) D^-1, times dx: cell units), claim, 27 steps back to back
		P2GPayload pv[2];
		int pv_code[2];
		bool pv_in[2];
		{
			const float am = sk.am;
#pragma unroll
			for(int h = 0; h < 2; ++h) {
#pragma unroll
				for(int d = 0; d < 3; ++d) {
					pv[h].fd[d] = nfd[h][d];
					pv[h].mv[d] = mass * vel[h][d];
				}
#pragma unroll
				for(int d = 0; d < 9; ++d) pv[h].contrib[d] = fmaf(A[h][d], am, contrib[h][d]);
				pv_code[h] = in_arena[h] ? (narena[h][0] | (narena[h][1] << 4) | (narena[h][2] << 8)) : -1;
				pv_in[h]   = pv_code[h] >= 0;
			}
		}
		// ---- claims: A for the slot's arena; a B with another base than its A (or without a chain-able A) for the OTHER arena, on its own.
		//      (Also measured: claims written and read back in FRONT of the material update, which would cover the round trip: +0.5-1 % for sand, +2.5 % for the J-fluid -
		//       one more register across the update; profiles/r06_ab_pairs_phase2.txt.)
		const bool a_ok	   = pv_in[0] && !code_edge(pv_code[0]);
		const bool b_same  = pv_in[1] && pv_in[0] && pv_code[1] == pv_code[0];
#if MPM_PAIR_DUAL
		const bool b_own = pv_in[1] && !b_same && !code_edge(pv_code[1]);
#else
		const bool b_own = false;
#endif
		const int key_a = (a_ok ? code_key(pv_code[0]) : 0) + arena_sel * 216;// two arenas, two claim tables: the sort says which one a slot uses
		const int key_b = (b_own ? code_key(pv_code[1]) : 0) + (arena_sel ^ 1) * 216;
		// (B's claims are written FIRST: a single wave's LDS stores execute in program order, so an A that claims the same base in the same arena overwrites
		//  them - a B never takes an arena away from a pair, it only uses one that would have stayed idle)
		if(b_own) s_owner[key_b] = (unsigned char) (lane | 64);
		__asm__ volatile("" ::: "memory");
		if(a_ok) s_owner[key_a] = (unsigned char) lane;
		__asm__ volatile("" ::: "memory");// another lane may have written the same byte: no store-to-load forwarding
		const bool win	   = a_ok && (int) s_owner[key_a] == lane;
		const bool dual	   = b_own && (int) s_owner[key_b] == (lane | 64);
		const bool merge_b = win && b_same;
		MPM_MARK("P_serial");
		// ---- what the chain will not take - A without a claim or on the cube's edge, B neither riding with its A nor holding a claim of its own - goes first:
		//      the payloads are dead once the chain is set up (both add into the arenas with plain read-modify-writes; a single wave's LDS operations execute
		//      in program order)
		{
			const bool left_a = pv_in[0] && !win;
#if defined(MPM_EXPERIMENT) && defined(MPM_HACK_NOSPLIT)// timing experiment only: a B that does not ride with its A is dropped (wrong physics)
			const bool left_b = false;
#else
			const bool left_b = pv_in[1] && !merge_b && !dual;
#endif
#ifdef MPM_G2P2G_STATS
			st_iter += 1;
			st_losers += __popcll(__ballot(left_a && !code_edge(pv_code[0]))) + __popcll(__ballot(left_b && !code_edge(pv_code[1])));
			st_edge += __popcll(__ballot(pv_in[0] && code_edge(pv_code[0]))) + __popcll(__ballot(pv_in[1] && code_edge(pv_code[1])));
			st_split += __popcll(__ballot(dual));
			st_retry_iters += __any(left_a || left_b) ? 1 : 0;
#endif
#if defined(MPM_EXPERIMENT) && defined(MPM_HACK_NOSERIAL)// timing experiment only: what the chain does not take is dropped (wrong physics)
			if(false)
#else
			if(__any(left_a))
#endif
			{
				if constexpr(kQueue)
					serial_push(p2g, s_queue, qn, left_a, pv_code[0], pv[0], mass, lane, info, next_grid);
				else
					p2g_serial(p2g, left_a, pv_code[0], pv[0], mass, lane, info, next_grid);
			}
#if defined(MPM_EXPERIMENT) && defined(MPM_HACK_NOSERIAL)
			if(false)
#else
			if(__any(left_b))
#endif
			{
				if constexpr(kQueue)
					serial_push(p2g, s_queue, qn, left_b, pv_code[1], pv[1], mass, lane, info, next_grid);
				else
					p2g_serial(p2g, left_b, pv_code[1], pv[1], mass, lane, info, next_grid);
			}
		}
		MPM_MARK("P_chain");
		if constexpr(kLateFetch == 0) sb_next[0] = source_bin(rec_nn[0]), sb_next[1] = source_bin(rec_nn[1]);// (rec_nn was requested at the top of this iteration)
		float4* const node0 = p2g + (win ? code_off(pv_code[0]) + arena_sel * kP2GArena2 : 0);
		if(__any(dual)) {
			float4* const node1 = p2g + (dual ? code_off(pv_code[1]) + (arena_sel ^ 1) * kP2GArena2 : 0);
			ScatterChainDual chain(node0, node1, pv[0], pv[1], mass, win, merge_b, dual);
			chain.run();
		} else {
			ScatterChain2<1> chain(node0, pv[0], pv[1], mass, win, merge_b);
			chain.template at<0>();
		}
		if(s_next.cnt == 0) break;
		s_cur  = s_next;
		s_next = s_nn;
#pragma unroll
		for(int h = 0; h < 2; ++h) rec_next[h] = rec_nn[h];
	}
#ifdef MPM_G2P2G_STATS
	if(lane == 0) {
		atomicAdd(&status[24], st_iter);
		atomicAdd(&status[25], st_losers);
		atomicAdd(&status[26], st_edge);
		atomicAdd(&status[27], MPM_PAIR_DUAL ? st_split : st_retry_iters);// iterations with a serial entry, as the one-particle kernel reports them (with MPM_PAIR_DUAL: the lanes whose B scattered from the other arena)
		atomicAdd(&status[28], st_partial);
		atomicAdd(&status[41], st_retry_iters);
	}
#endif
	if constexpr(kQueue) {
		if(qn) serial_flush(p2g, s_queue, qn, mass, lane, info, next_grid);
	}
	if(lane == 0) mv.keep[b] = settled ? size : -1;
	// if nobody leaves or arrives, row b of list_out holds the records in THIS layout: hand the pair counts on under the block's number
	if(lane < kPairChunks) mv.pairinfo_out[(size_t) b * kPairChunks + lane] = pinfo & 0xffff;
	__syncthreads();
	// ---- arena -> next grid (:907-936), as in g2p2g_kernel
	int lane_wb = lane;
	__asm__ volatile("" : "+v"(lane_wb));
	const int cx = lane_wb >> 4, cy = (lane_wb >> 2) & 3, cz = lane_wb & 3;
#pragma unroll
	for(int lb = 0; lb < 8; ++lb) {
		int sel = 54 + lb;
		__asm__ volatile("" : "+s"(sel));
		const int nb = __shfl(info, sel);
		const int ax = cx + ((lb & 4) ? 4 : 0) - 1, ay = cy + ((lb & 2) ? 4 : 0) - 1, az = cz + ((lb & 1) ? 4 : 0) - 1;
		const bool in = ((unsigned) ax < 6u) & ((unsigned) ay < 6u) & ((unsigned) az < 6u);
		const int n	  = in ? ax * kP2GStrideX + ay * kP2GStrideY + az : 0;
		const float4 va = p2g[n], vb = p2g[kP2GArena2 + n];
		const float4 v	= make_float4(va.x + vb.x, va.y + vb.y, va.z + vb.z, va.w + vb.w);
		if(in && nb >= 0) {
			float* g = next_grid + (size_t) nb * 256 + lane_wb;
			if(v.x != 0.f) unsafeAtomicAdd(g, v.x);
			if(v.y != 0.f) unsafeAtomicAdd(g + 64, v.y);
			if(v.z != 0.f) unsafeAtomicAdd(g + 128, v.z);
			if(v.w != 0.f) unsafeAtomicAdd(g + 192, v.w);
		}
	}
	__syncthreads();
	}
}

}// namespace mpm
