// mpm_g2p2g.hpp — the fused G2P + particle update + P2G kernel (g2p2g, Projects/GMPM/mgmpm_kernels.cuh:665-937 with the
// per-material bodies :422-663), round-2 design with the round-3 / round-4 refinements.
//
// Cost model (tools/valu_rate.hip, tools/lds_bank_probe.hip, profiles/r04_*): the launch runs at the socket's power
// limit with the vector pipe as its most loaded unit (789 vector instructions per 64 particles at rest, 1 325 in a flow) and the LDS
// pipe right behind it (27 gather reads + 27 scatter read-modify-write pairs per iteration); leave-one-out experiments say it is
// neither bound by bytes (the solid models carry b = F F^T: 100 B of HBM traffic per particle), nor by the LDS pipe alone, nor by
// occupancy.  What has paid is executing less.  The design:
//   * <= 168 VGPRs (three waves per SIMD; the solid instantiations use 147-164): the P2G scatter chain of the previous particle is
//     threaded through the re-bucketing and the material update only (the round-1 pipeline through the gather as well cost ~40 more
//     live registers), the gather keeps one z-pencil of loads in flight, its results are pinned where they are produced;
//   * 12.5 KiB of LDS per wave: the advection records are read straight from global memory (prepare_blocks_kernel left them
//     sorted), the block's look-up row stays in a register (ds_bpermute), and ALL arenas hold only nodes 1..6 of the 8^3
//     node cube around the block.  The gather never touches nodes 0 and 7; the scatter does only for particles that have
//     just crossed into a neighbouring block (2 % in a collapsing column, none at rest): those lanes go through the serial
//     path (an LDS queue worked off once per block), which sends the shell part of a stencil to the grid with global atomics;
//   * the per-particle math needs only U and sigma of b = F F^T (sym_eig3 in mpm_device_math.hpp), not a full SVD, and b is the
//     state a particle carries.
// One workgroup = ONE wave = one particle block: LDS operations of a single wave execute in order, which is what makes the
// atomic-free read-modify-write scatter legal.  Lanes of an iteration hold distinct stencil bases: prepare_blocks_kernel
// sorts the records by predicted base and deals them to the block's slices round-robin (a key's particles land in
// consecutive slices); the two particles of a key that still share a slice sit in neighbouring lanes and scatter into
// separate arenas (lane parity); what is left is decided by an owner table and goes through the serial path.
#pragma once
#include "mpm_device_math.hpp"

namespace mpm {

#if !(defined(MPM_EXPERIMENT) && defined(MPM_G2P2G_WAVES_FLUID))
#define MPM_G2P2G_WAVES_FLUID 4// J-fluid instantiation: 128 VGPRs suffice, and with a material update of a handful of instructions the
								// scatter chain has little to hide behind - occupancy does it (same-box: 0.36 ms against 0.40 at three waves)
#endif
#if !(defined(MPM_EXPERIMENT) && defined(MPM_G2P2G_WAVES))
#define MPM_G2P2G_WAVES 3// waves per SIMD the register allocation is held to (168 VGPRs)
#endif

// LDS arenas: nodes 1..6 per axis of the 8^3 cube spanned by the block's 2x2x2 grid blocks.
// Sort key of a particle = its stencil base in that cube, key = y * 36 + x * 6 + z (y slowest): with the usual population
// (bases 1..4 per axis) a 16-lane group then holds one y-plane, which both layouts below serve without bank conflicts
// beyond the hardware's b128 pass structure (tools/valu_microbench3: 4.0 cycles per ds_read_b128 on the gather arena,
// 8.8 per read-modify-write instruction on the scatter arena, against 4.0 / 8.3 for a linear address pattern).
constexpr int kP2GStrideX = 36, kP2GStrideY = 6, kP2GNodes = 216;// scatter arenas (two): (x-1)*36 + (y-1)*6 + (z-1)
// The second scatter arena starts 228 nodes behind the first (12 nodes = 3/4 of a bank row further than back to back): of 600 linear
// layouts x arena offsets timed on the chip under rest-like and flow-like lane -> node patterns (tools/lds_bank_probe.hip,
// profiles/r04_lds_bank_probe.txt) the dense x-major layout is among the best in both regimes, and this offset is its best (flow-like
// patterns: 27 x (gather read + scatter read-modify-write) 1245 against 1274 cycles back to back; rest: the same).
constexpr int kP2GArena2 = 228;
// Gather arena: the same 216 nodes, x / y / z strides 6 / 1 / 36.  Timed on the chip with all twelve waves of a CU reading
// (tools/lds_bank_probe.hip, profiles/r04_lds_bank_probe.txt), a ds_read_b128 of the rest pattern costs 7.4-8.0 periods per wave-instruction
// per CU with this layout and 9.9 with the scatter arenas' 36 / 6 / 1 (flow-like patterns: 10.5 against 10.8) - and the launch is bound by
// LDS throughput: the gather's 27 reads are a quarter of an iteration's LDS time.
constexpr int kG2PStrideX = 6, kG2PStrideY = 1, kG2PStrideZ = 36, kG2PNodes = 216;

struct ModelView {
	const float* bins_src;// [bin][64 records][nch floats], laid out by the previous block numbering
	float* bins_dst;	  // laid out by the current numbering
	const int* binoff_src;// first bin of a block, previous numbering
	const int* binoff_dst;// current numbering
	const int* list_in;	  // advection records written by the previous step (sorted by prepare_blocks_kernel); row = row_of[b]
	int* list_out;		  // records for the next step; row = destination block (current numbering)
	const int* size;	  // particles per current block
	const int* row_of;	  // row of list_in that belongs to current block b
	int* out_count;		  // append counters of list_out
	int* keep;			  // per block: its size if every particle stayed with an unchanged sort key, else -1 (prepare_blocks_kernel skips the sort)
	const int* blockinfo; // [block][kInfoRow]: source bin offsets, destination / grid block numbers (prepare_blocks_kernel)
	const int* pairinfo_in;// pair layout (mpm_kernels.hpp): full pairs per chunk of a block's list, [block][kPairChunks]; null for the sliced layout
	int* pairinfo_out;	   // a block hands its own on under its number (prepare_blocks_kernel: the next sort is skipped for a settled block)
	MaterialConst mc;
};

// Particle storage: bins of 64 particle RECORDS plus, for the solid models, one ROW of 64 entries behind them.
//   J-fluid:          record {x, y, z, J}                                   16 B, no row                      bin = 1024 B
//   fixed-corotated:  record {x, y, z, b00 | b11, b22, b10, b20}            32 B, row of b21          (4 B)   bin = 2304 B
//   sand, NACC:       the same record,                                            row of {b21, log Jp} (8 B)   bin = 2560 B
// b = F F^T is the state the solid models carry instead of F (mpm_device_math.hpp: 24 B instead of 36 B per particle); the sign
// bit of b00 marks a reflected F.  A record is moved with 16-byte loads and stores and never straddles a 64-B sector; the row
// is a coalesced 256 / 512-B access while the sorted order of a step equals the order the particles were written in.
// History: the reference's (and round 1's) AoSoA bins [channel][slot] are ideal only while that holds; in a flow the 64 lanes of an
// iteration read 64 scattered slots, and with one 256-B row per channel that is 13 x ~16 cache lines per wave-load and a 6x HBM
// read amplification (profiles/r02_moving_window_pmc.txt).  Round 2: 64-B records {x, y, z, F[9], log Jp, pad}; round 3: 48-B
// records + a row of log Jp (-11 % kernel time at rest for 16 % fewer bytes: the launch pays for bytes); round 4:
// b instead of F.  `nch` = floats per particle in a bin, `rec` = floats per record, the rest per row entry.
template<int MAT>
struct MatTraits;
template<>
struct MatTraits<0> {
	static constexpr int nch = 4, rec = 4;
};
template<>
struct MatTraits<1> {
	static constexpr int nch = 9, rec = 8;
};
template<>
struct MatTraits<2> {
	static constexpr int nch = 10, rec = 8;
};
template<>
struct MatTraits<3> {
	static constexpr int nch = 10, rec = 8;
};
constexpr unsigned kReflBit = 0x80000000u;// sign bit of b00 (positive otherwise): det F < 0

// P2G payload of one particle: everything the scatter needs after the material update.
struct P2GPayload {
	float fd[3];	 // offset from the new stencil base node, in cells
	float mv[3];	 // mass * velocity
	float contrib[9];// (A m - stress new_dt) D^-1 dx   (:850)
};

// Tensor-product B-spline gather of one particle per lane (:801-835): vel = sum W v, A = sum W v (x_i - x_p)^T in cell
// units, separable over the three axes (27 x 6 + 9 x 9 + 3 x 12 multiply-adds instead of 27 x 12), on float2 so that the
// x / y components and the (w, w (x_i - x_p)) weight pairs go through packed fp32.  A node is {vx, vy, vz, vz}: the
// duplicate makes the load a full ds_read_b128 (4.0 cycles per wave against 7.1 for ds_read_b96) and gives the z
// accumulators a natural register pair.
#if !defined(MPM_EXPERIMENT) || !defined(MPM_GATHER_ASM)
#define MPM_GATHER_ASM 1
#endif
typedef float v4f_ __attribute__((ext_vector_type(4)));
// One z-pencil of the gather arena (three nodes, 16 B each) with all three reads IN FLIGHT and a wait in front of each use.  Written out because the
// compiler, short of registers in this kernel, loads the three nodes into ONE register quad, one after the other, each behind s_waitcnt lgkmcnt(0): 54 exposed
// LDS round trips per pair of particles.  The waits are safe whatever else the compiler has outstanding: LDS operations complete in order, lgkmcnt(N) says "all but
// the newest N are done", and operations the compiler does not know about only make its own waits longer.  (No scalar loads inside the particle loop: tools/kstat.sh.)
template<int OFF>
MPM_DEV v4f_ lds_issue_b128(unsigned addr) {
	v4f_ v;
	__asm__ volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF) : "memory");
	return v;
}
template<int N>
MPM_DEV void lds_arrived(v4f_& v) {// v was requested when N younger LDS reads were issued after it
	__asm__ volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}
MPM_DEV void gather_apic(const float4* __restrict__ gbase, const float (&w)[3][3], const float (&fd)[3], float (&vel)[3], float (&A)[9]) {
	v2f_ wz[3], wy[3], wx[3];// {w, w * (node - particle)} per axis and stencil offset
#pragma unroll
	for(int t = 0; t < 3; ++t) {
		wx[t] = (v2f_) {w[0][t], w[0][t] * ((float) t - fd[0])};
		wy[t] = (v2f_) {w[1][t], w[1][t] * ((float) t - fd[1])};
		wz[t] = (v2f_) {w[2][t], w[2][t] * ((float) t - fd[2])};
	}
	v2f_ vel_xy = {0.f, 0.f}, A0_xy = {0.f, 0.f}, A3_xy = {0.f, 0.f}, A6_xy = {0.f, 0.f};
	v2f_ velz_A2 = {0.f, 0.f};
	float A5 = 0.f, A8 = 0.f;
#if MPM_GATHER_ASM
	// (Also measured: a rolling window - node q + 3 requested into the registers node q leaves behind - needs two registers more, which the sand and J-fluid
	//  instantiations pay with a reload inside the particle loop: +3 % / +1.5 %, profiles/r06_ab_pairs_phase2.txt.)
	const unsigned lds0 = (unsigned) (size_t) gbase;// (the low half of a generic pointer into LDS is the LDS address)
	v2f_ u0_xy, uy_xy, uz_xy, u0z_uyz;
	float uzz;
	auto pencil = [&](auto ic, auto jc) {
		constexpr int I = decltype(ic)::value, J = decltype(jc)::value;
		constexpr int off = (I * kG2PStrideX + J * kG2PStrideY) * 16;
		v4f_ n0 = lds_issue_b128<off>(lds0), n1 = lds_issue_b128<off + kG2PStrideZ * 16>(lds0), n2 = lds_issue_b128<off + 2 * kG2PStrideZ * 16>(lds0);
		v2f_ t0_xy, t1_xy, t0z_t1z;
		lds_arrived<2>(n0);
		{
			const v2f_ vxy = {n0.x, n0.y}, vzz = {n0.z, n0.w};
			t0_xy = vxy * wz[0].x, t1_xy = vxy * wz[0].y, t0z_t1z = wz[0] * vzz;
		}
		lds_arrived<1>(n1);
		{
			const v2f_ vxy = {n1.x, n1.y}, vzz = {n1.z, n1.w};
			t0_xy = vxy * wz[1].x + t0_xy, t1_xy = vxy * wz[1].y + t1_xy, t0z_t1z = wz[1] * vzz + t0z_t1z;
		}
		lds_arrived<0>(n2);
		{
			const v2f_ vxy = {n2.x, n2.y}, vzz = {n2.z, n2.w};
			t0_xy = vxy * wz[2].x + t0_xy, t1_xy = vxy * wz[2].y + t1_xy, t0z_t1z = wz[2] * vzz + t0z_t1z;
		}
		__builtin_amdgcn_sched_barrier(0);// one pencil at a time: registers are the scarce resource
		if constexpr(J == 0) {
			u0_xy = t0_xy * wy[0].x, uy_xy = t0_xy * wy[0].y, uz_xy = t1_xy * wy[0].x, u0z_uyz = wy[0] * t0z_t1z.x, uzz = wy[0].x * t0z_t1z.y;
		} else {
			u0_xy	= t0_xy * wy[J].x + u0_xy;
			uy_xy	= t0_xy * wy[J].y + uy_xy;
			uz_xy	= t1_xy * wy[J].x + uz_xy;
			u0z_uyz = wy[J] * t0z_t1z.x + u0z_uyz;
			uzz += wy[J].x * t0z_t1z.y;
		}
	};
	auto slab = [&](auto ic) {
		constexpr int I = decltype(ic)::value;
		pencil(ic, std::integral_constant<int, 0> {});
		pencil(ic, std::integral_constant<int, 1> {});
		pencil(ic, std::integral_constant<int, 2> {});
		vel_xy	= u0_xy * wx[I].x + vel_xy;
		A0_xy	= u0_xy * wx[I].y + A0_xy;
		A3_xy	= uy_xy * wx[I].x + A3_xy;
		A6_xy	= uz_xy * wx[I].x + A6_xy;
		velz_A2 = wx[I] * u0z_uyz.x + velz_A2;
		A5 += wx[I].x * u0z_uyz.y;
		A8 += wx[I].x * uzz;
	};
	slab(std::integral_constant<int, 0> {});
	slab(std::integral_constant<int, 1> {});
	slab(std::integral_constant<int, 2> {});
#else
#pragma unroll
	for(int i = 0; i < 3; ++i) {
		v2f_ u0_xy = {0.f, 0.f}, uy_xy = {0.f, 0.f}, uz_xy = {0.f, 0.f}, u0z_uyz = {0.f, 0.f};
		float uzz = 0.f;
#pragma unroll
		for(int j = 0; j < 3; ++j) {
			v2f_ t0_xy = {0.f, 0.f}, t1_xy = {0.f, 0.f}, t0z_t1z = {0.f, 0.f};
#pragma unroll
			for(int k = 0; k < 3; ++k) {
				const float4 v = gbase[i * kG2PStrideX + j * kG2PStrideY + k * kG2PStrideZ];
				const v2f_ vxy = {v.x, v.y}, vzz = {v.z, v.w};
				t0_xy		   = vxy * wz[k].x + t0_xy;
				t1_xy		   = vxy * wz[k].y + t1_xy;
				t0z_t1z		   = wz[k] * vzz + t0z_t1z;
			}
			__builtin_amdgcn_sched_barrier(0);// at most one z-pencil (3 nodes, 12 registers) of loads in flight: the other waves of the SIMD cover the LDS latency, registers are the scarce resource
			u0_xy	= t0_xy * wy[j].x + u0_xy;
			uy_xy	= t0_xy * wy[j].y + uy_xy;
			uz_xy	= t1_xy * wy[j].x + uz_xy;
			u0z_uyz = wy[j] * t0z_t1z.x + u0z_uyz;
			uzz += wy[j].x * t0z_t1z.y;
		}
		vel_xy	= u0_xy * wx[i].x + vel_xy;
		A0_xy	= u0_xy * wx[i].y + A0_xy;
		A3_xy	= uy_xy * wx[i].x + A3_xy;
		A6_xy	= uz_xy * wx[i].x + A6_xy;
		velz_A2 = wx[i] * u0z_uyz.x + velz_A2;
		A5 += wx[i].x * u0z_uyz.y;
		A8 += wx[i].x * uzz;
	}
#endif
	vel[0] = vel_xy.x;
	vel[1] = vel_xy.y;
	vel[2] = velz_A2.x;
	A[0]   = A0_xy.x;
	A[1]   = A0_xy.y;
	A[2]   = velz_A2.y;
	A[3]   = A3_xy.x;
	A[4]   = A3_xy.y;
	A[5]   = A5;
	A[6]   = A6_xy.x;
	A[7]   = A6_xy.y;
	A[8]   = A8;
	// A is first used in the material update, far below: pin the results here, or the compiler sinks two thirds of the
	// accumulation down to that use and keeps the x / y / z components of all 27 loaded nodes alive until then (81 VGPRs)
#pragma unroll
	for(int d = 0; d < 9; ++d) __asm__ volatile("" : "+v"(A[d]));
#pragma unroll
	for(int d = 0; d < 3; ++d) __asm__ volatile("" : "+v"(vel[d]));
}

// P2G of the particles the chain cannot take - lanes that lost the claim of their stencil base (another lane of the
// iteration holds the same base) and edge lanes, whose stencil reaches cube nodes 0 or 7 (cells of the 2x2x2 grid blocks
// outside the LDS arena; the particle has just crossed into a neighbouring block).  They are served TWO PARTICLES AT A TIME,
// each with its 27 stencil nodes spread over 27 lanes of one half of the wave (lanes 0..26 and 32..58): the payloads are
// broadcast with ds_bpermute (the LDS crossbar, no VALU: one instruction serves both halves, where v_readlane would take
// 2 x 17 plus the selects), lane o of a half computes the contribution to node o and adds it - a plain read-modify-write
// for arena nodes (the 27 nodes of one particle are distinct, the two halves use separate arenas, and a single wave's LDS
// operations execute in order), a global float atomic per channel for shell nodes.
// One LDS round trip per PAIR whatever the multiplicity of the bases, no claim rounds, no second 27-step pass.
MPM_DEV void p2g_serial(float4* __restrict__ arena, bool pending, int code, const P2GPayload& pl, float mass, int lane, int info, float* __restrict__ next_grid) {
	__asm__ volatile("" : "+v"(lane));// (recomputed here every time: hoisted out of the main loop these would hold six registers)
	const int l	   = lane & 31;
	const int half = lane >> 5;
	const int oi = l / 9, oj = (l / 3) % 3, ok = l % 3;// this lane's stencil offset (lanes 0..26 of each half)
	const float fi = (float) oi, fj = (float) oj, fk = (float) ok;
	float4* const my_arena = arena + half * kP2GArena2;
	unsigned long long todo = __ballot(pending);
	while(todo) {
		const int src_a = __ffsll((long long) todo) - 1;
		todo &= todo - 1;
		const bool two	= todo != 0ull;
		const int src_b = two ? __ffsll((long long) todo) - 1 : src_a;
		todo &= todo - 1;// (no-op on zero)
		const int src = half ? src_b : src_a;
		float fd[3], mvv[3], c[9];
#pragma unroll
		for(int d = 0; d < 3; ++d) {
			fd[d]  = __shfl(pl.fd[d], src);
			mvv[d] = __shfl(pl.mv[d], src);
		}
#pragma unroll
		for(int d = 0; d < 9; ++d) c[d] = __shfl(pl.contrib[d], src);
		const int cd = __shfl(code, src);
		const int nx = cd & 15, ny = (cd >> 4) & 15, nz = cd >> 8;
		float w[3][3];
#pragma unroll
		for(int d = 0; d < 3; ++d) bspline_weight_cells(fd[d], w[d]);
		const float wx = oi == 0 ? w[0][0] : (oi == 1 ? w[0][1] : w[0][2]);
		const float wy = oj == 0 ? w[1][0] : (oj == 1 ? w[1][1] : w[1][2]);
		const float wz = ok == 0 ? w[2][0] : (ok == 1 ? w[2][1] : w[2][2]);
		const float W  = wx * wy * wz;
		const float px = fi - fd[0], py = fj - fd[1], pz = fk - fd[2];
		const float v0 = mass * W;
		const float v1 = (mvv[0] + c[0] * px + c[3] * py + c[6] * pz) * W;
		const float v2 = (mvv[1] + c[1] * px + c[4] * py + c[7] * pz) * W;
		const float v3 = (mvv[2] + c[2] * px + c[5] * py + c[8] * pz) * W;
		const int gx = nx + oi, gy = ny + oj, gz = nz + ok;// cube coordinates 0..7
		const bool inside = ((unsigned) (gx - 1) < 6u) & ((unsigned) (gy - 1) < 6u) & ((unsigned) (gz - 1) < 6u);
		const int nb	  = __shfl(info, 54 + ((gx >> 2) & 1) * 4 + ((gy >> 2) & 1) * 2 + ((gz >> 2) & 1));// grid block of the node (all lanes active here)
		if(l < 27 && (half == 0 || two)) {
			if(inside) {
				float4* node	 = my_arena + (gx - 1) * kP2GStrideX + (gy - 1) * kP2GStrideY + (gz - 1);
				const float4 acc = *node;
				*node			 = make_float4(acc.x + v0, acc.y + v1, acc.z + v2, acc.w + v3);
			} else {
				if(nb >= 0) {
					float* g = next_grid + (size_t) nb * 256 + (gx & 3) * 16 + (gy & 3) * 4 + (gz & 3);
					unsafeAtomicAdd(g, v0);
					unsafeAtomicAdd(g + 64, v1);
					unsafeAtomicAdd(g + 128, v2);
					unsafeAtomicAdd(g + 192, v3);
				}
			}
		}
		__asm__ volatile("" ::: "memory");// the next pair may hit the same nodes: keep the LDS operations in program order
	}
}

// Deferred variant of p2g_serial (round 3, MPM_SERIAL_QUEUE): the lanes that cannot take the chain park their payload in a small LDS
// queue (16 dwords per particle: no round trip, the stores are fire-and-forget) and the queue is worked off once per block, after
// the particle loop - or when it is full -, two particles per step as above.  Per pair the payload then comes from four broadcast
// ds_read_b128 (all lanes of a half read the same entry) instead of 17 ds_bpermute, and consecutive pairs are independent up to the
// read-modify-write itself, so their loads and arithmetic overlap; issued inside the particle loop every pair was an exposed
// bpermute -> arithmetic -> LDS round trip (0.28 of 2.15 ms in the flow window of C3 for 3.6 % of the particles).
#if defined(MPM_EXPERIMENT) && defined(MPM_QUEUE_ENTRIES)
constexpr int kSerialQueue = MPM_QUEUE_ENTRIES;// (0: no queue, the lanes that cannot take the chain scatter inside the particle loop)
#else
constexpr int kSerialQueue = 28;// entries of 64 B: 10.8 + 1.8 KB of LDS per wave, still 12 single-wave workgroups per CU
#endif
MPM_DEV void serial_flush(float4* __restrict__ arena, const float4* __restrict__ queue, int qn, float mass, int lane, int info, float* __restrict__ next_grid) {
	__asm__ volatile("" : "+v"(lane));
	const int l	   = lane & 31;
	const int half = lane >> 5;
	const int oi = l / 9, oj = (l / 3) % 3, ok = l % 3;
	const float fi = (float) oi, fj = (float) oj, fk = (float) ok;
	float4* const my_arena = arena + half * kP2GArena2;
	for(int e = 0; e < qn; e += 2) {
		const bool live	 = e + half < qn;
		const float4* en = queue + 4 * (live ? e + half : e);
		const float4 q0 = en[0], q1 = en[1], q2 = en[2], q3 = en[3];
		const float fd[3] = {q0.x, q0.y, q0.z};
		const int cd	  = __float_as_int(q0.w);
		const int nx = cd & 15, ny = (cd >> 4) & 15, nz = cd >> 8;
		float w[3][3];
#pragma unroll
		for(int d = 0; d < 3; ++d) bspline_weight_cells(fd[d], w[d]);
		const float wx = oi == 0 ? w[0][0] : (oi == 1 ? w[0][1] : w[0][2]);
		const float wy = oj == 0 ? w[1][0] : (oj == 1 ? w[1][1] : w[1][2]);
		const float wz = ok == 0 ? w[2][0] : (ok == 1 ? w[2][1] : w[2][2]);
		const float W  = wx * wy * wz;
		const float px = fi - fd[0], py = fj - fd[1], pz = fk - fd[2];
		const float v0 = mass * W;
		const float v1 = (q1.x + q1.w * px + q2.z * py + q3.y * pz) * W;// mv[0] + c[0] px + c[3] py + c[6] pz
		const float v2 = (q1.y + q2.x * px + q2.w * py + q3.z * pz) * W;// mv[1] + c[1] px + c[4] py + c[7] pz
		const float v3 = (q1.z + q2.y * px + q3.x * py + q3.w * pz) * W;// mv[2] + c[2] px + c[5] py + c[8] pz
		const int gx = nx + oi, gy = ny + oj, gz = nz + ok;// cube coordinates 0..7
		const bool inside = ((unsigned) (gx - 1) < 6u) & ((unsigned) (gy - 1) < 6u) & ((unsigned) (gz - 1) < 6u);
		const int nb	  = __shfl(info, 54 + ((gx >> 2) & 1) * 4 + ((gy >> 2) & 1) * 2 + ((gz >> 2) & 1));
		if(l < 27 && live) {
			if(inside) {
				float4* node	 = my_arena + (gx - 1) * kP2GStrideX + (gy - 1) * kP2GStrideY + (gz - 1);
				const float4 acc = *node;
				*node			 = make_float4(acc.x + v0, acc.y + v1, acc.z + v2, acc.w + v3);
			} else if(nb >= 0) {
#if defined(MPM_EXPERIMENT) && defined(MPM_HACK_NOSHELL)// timing / traffic experiment only: the shell contributions of the serial path are dropped (wrong physics)
				if(size_t(next_grid) != 1) continue;
#endif
				float* g = next_grid + (size_t) nb * 256 + (gx & 3) * 16 + (gy & 3) * 4 + (gz & 3);
				unsafeAtomicAdd(g, v0);
				unsafeAtomicAdd(g + 64, v1);
				unsafeAtomicAdd(g + 128, v2);
				unsafeAtomicAdd(g + 192, v3);
			}
		}
		__asm__ volatile("" ::: "memory");// the next pair may hit the same nodes: keep the LDS operations in program order
	}
}
// park the payloads of the `pending` lanes; qn (wave-uniform) = entries in the queue
MPM_DEV void serial_push(float4* __restrict__ arena, float4* __restrict__ queue, int& qn, bool pending, int code, const P2GPayload& pl, float mass, int lane, int info, float* __restrict__ next_grid) {
	unsigned long long todo = __ballot(pending);
	while(todo) {
		const int room = kSerialQueue - qn;
		const int rank = (int) __builtin_amdgcn_mbcnt_hi((unsigned) (todo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) todo, 0u));
		const bool mine = ((todo >> lane) & 1ull) != 0ull && rank < room;
		if(mine) {
			float4* en = queue + 4 * (qn + rank);
			en[0]	   = make_float4(pl.fd[0], pl.fd[1], pl.fd[2], __int_as_float(code));
			en[1]	   = make_float4(pl.mv[0], pl.mv[1], pl.mv[2], pl.contrib[0]);
			en[2]	   = make_float4(pl.contrib[1], pl.contrib[2], pl.contrib[3], pl.contrib[4]);
			en[3]	   = make_float4(pl.contrib[5], pl.contrib[6], pl.contrib[7], pl.contrib[8]);
		}
		qn += min(__popcll(todo), room);
		todo &= ~__ballot(mine);
		__asm__ volatile("" ::: "memory");
		if(todo) {// the queue is full: work it off now
			serial_flush(arena, queue, qn, mass, lane, info, next_grid);
			qn = 0;
		}
	}
}

// The P2G scatter of one particle per lane as a chain of 27 ordered LDS read-modify-write steps that is threaded through
// unrelated register-only arithmetic: the NEXT particle's F update, eigen-decomposition and material model, a latency-
// bound dependent VALU stream itself (a dependent VALU instruction issues every ~9 cycles).  Each step is an LDS round
// trip and the steps cannot overlap each other - step o+1 may hit the node another lane wrote in step o - so issued back
// to back they leave the wave waiting 27 times.  Site s of NSITES completes steps [27 s / NSITES, 27 (s+1) / NSITES): the
// accumulator of the following step is requested right after a step's write and consumed at the next site.  Only lanes
// with `win` (pairwise distinct stencil bases, no shell nodes) take part; the others go through p2g_serial.
template<int NSITES>
struct ScatterChain {
	float4* node0;
	float mass;
	int win;// (int, not bool: a 1-byte member makes the compiler slice its neighbours into bytes)
	float pw[3][3];
	// The momentum a node (i, j, k) of the stencil receives is W_ijk (m v + C (x_ijk - x_p)), x_ijk - x_p = (i, j, k) - fd in cells: the
	// affine term is walked INCREMENTALLY - slab = m v - C fd + i C_x, pencil = slab + j C_y, node = pencil + k C_z - with one add per
	// component and move (x component scalar, y / z packed), where forming (i, j, k) - fd and multiplying cost three per node
	float cx0, cy0, cz0, cz0x2;	  // x row of C: contrib[0], [3], [6] (column-major), and twice the last
	v2f_ cx12, cy12, cz12, cz12x2;// y / z rows
	float slab0, pen0, wij;
	v2f_ slab12, pen12;
	float4 acc;
	MPM_DEV ScatterChain(float4* n0, const P2GPayload& p, float m, bool w)
		: node0(n0)
		, mass(m)
		, win(w) {
#pragma unroll
		for(int d = 0; d < 3; ++d) bspline_weight_cells(p.fd[d], pw[d]);
		cx0	 = p.contrib[0], cy0 = p.contrib[3], cz0 = p.contrib[6];
		cx12 = (v2f_) {p.contrib[1], p.contrib[2]}, cy12 = (v2f_) {p.contrib[4], p.contrib[5]}, cz12 = (v2f_) {p.contrib[7], p.contrib[8]};
		slab0	= p.mv[0] - cx0 * p.fd[0] - cy0 * p.fd[1] - cz0 * p.fd[2];
		slab12	= (v2f_) {p.mv[1], p.mv[2]} - cx12 * p.fd[0] - cy12 * p.fd[1] - cz12 * p.fd[2];
		cz0x2	= cz0 + cz0;
		cz12x2	= cz12 + cz12;
		if(win) acc = node0[0];
	}
	MPM_DEV void step(int o) {// o is a compile-time constant after unrolling; called under `win`
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
		const float W  = wij * pw[2][k];
		const v2f_ m0  = {mass, k == 0 ? pen0 : (k == 1 ? pen0 + cz0 : pen0 + cz0x2)};
		const v2f_ t12 = k == 0 ? pen12 : (k == 1 ? pen12 + cz12 : pen12 + cz12x2);
		v2f_ a01 = {acc.x, acc.y};
		v2f_ a23 = {acc.z, acc.w};
		a01		 = m0 * W + a01;
		a23		 = t12 * W + a23;
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
		// one exec-mask bracket per site (a site holds one or two steps).  (Letting the other lanes run the steps on a scratch stencil
		// instead removes the brackets altogether and is slower: +1-4 % sand, +11 % J-fluid.)
		if(win) {
#pragma unroll
			for(int o = SITE * 27 / NSITES; o < (SITE + 1) * 27 / NSITES; ++o) step(o);
		}
	}
};

// Uniform per-launch factors, formed on the host: gfx950 has no scalar float ALU, so a uniform product computed in the kernel lives in
// a vector register for the whole particle loop.
struct StepConst {
	float dtp; // dt / dx: advection in cell units (particle positions are stored as x / dx)
	float dts; // dt * dx * D^-1 = dt * 4 / dx: A (cell units) -> dt grad v
	float pred;// new_dt / dx
	float am;  // mass dx^2 D^-1
	StressScale ss;// {2 mu, lambda, 1} * volume * (-new_dt D^-1 dx): the stress arrives as its P2G term (mpm_device_math.hpp)
	float jdiv, jvisc;// J-fluid: dx dt D^-1, dx D^-1 viscosity (stress_jfluid)
	float refl_lim;// (1/3) / dts: an entry of A beyond it could make det(I + dt grad v) <= 0 (the reflection bit of b, mpm_device_math.hpp)
};

// packed stencil base in the node cube (x | y << 4 | z << 8), -1 = outside (contribution discarded, :877-885)
MPM_DEV int code_key(int c) {
	return ((c >> 4) & 15) * 36 + (c & 15) * 6 + (c >> 8);
}
MPM_DEV bool code_edge(int c) {
	const int x = c & 15, y = (c >> 4) & 15, z = c >> 8;
	return (x == 0) | (x == 5) | (y == 0) | (y == 5) | (z == 0) | (z == 5);
}
MPM_DEV int code_off(int c) {
	return ((c & 15) - 1) * kP2GStrideX + (((c >> 4) & 15) - 1) * kP2GStrideY + ((c >> 8) - 1);
}

template<int MAT>
__global__ __launch_bounds__(kG2P2GThreads, MAT == 0 ? MPM_G2P2G_WAVES_FLUID : MPM_G2P2G_WAVES) void g2p2g_kernel(GridCfg cfg, ModelView mv, const int* __restrict__ cur_keys, const float* __restrict__ grid, float* __restrict__ next_grid, const int* __restrict__ block_list, const int* __restrict__ only_flag, const int* __restrict__ nblocks_ptr, int nblocks, float dt, float new_dt, StepConst sk, int* __restrict__ status) {
	constexpr int NCH = MatTraits<MAT>::nch;// floats per particle in a bin
	constexpr int REC = MatTraits<MAT>::rec;// floats per record (the rest: one 64-float row per channel behind the records)
	// 10.8 KB of LDS per wave: 15 single-wave workgroups per CU.
	__shared__ float4 g2p[kG2PNodes];	   // node velocities {vx, vy, vz, vz} of cube nodes 1..6 per axis
	__shared__ float4 p2g[kP2GArena2 + kP2GNodes];  // {mass, momentum} accumulators of cube nodes 1..6 per axis; even lanes use the first
										   // copy, odd lanes the second: the sort puts two particles of one key that share a slice
										   // into neighbouring lanes, so both scatter in the same pass (summed in the write-back)
	__shared__ unsigned char s_owner[2 * 216];
	// (not for the J-fluid: its instantiation runs at four waves per SIMD and 14 workgroups per CU, which the queue's LDS would cost)
	constexpr bool kQueue = MAT != 0 && kSerialQueue > 0;
	__shared__ float4 s_queue[kQueue ? 4 * kSerialQueue : 1];// payloads of the lanes that could not take the scatter chain (serial_push / serial_flush)
#if defined(MPM_EXPERIMENT) && defined(MPM_LDS_PAD)
	__shared__ float s_pad[MPM_LDS_PAD / 4];// experiment: lower the occupancy without touching the code
	if(size_t(grid) == 1) s_pad[threadIdx.x] = 0.f;
#endif

	const int lane0 = threadIdx.x;
	// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Consecutive block numbers are spatial
	// neighbours (they share grid blocks: reads in the set-up, atomics in the write-back), so every XCD gets one contiguous
	// eighth of the block range instead of every eighth block.
	// The number of blocks is read from device memory when nblocks_ptr is given (the launch is then sized by the host's ESTIMATE
	// of it - mpm_run_fixed does not wait for the rebuild's counts -, a multiple of 8): workgroup (xcd, q) takes the blocks
	// q, q + gridDim.x / 8, ... of its XCD's range; one trip when the estimate holds.
	const int total = nblocks_ptr ? min(*nblocks_ptr, cfg.cap) : nblocks;
	const int xcd = (int) (blockIdx.x & 7u), nq = nblocks_ptr ? (int) (gridDim.x >> 3) : 0x40000000;
	const int share = (total >> 3) + (xcd < (total & 7) ? 1 : 0);// XCD r owns (total / 8) + (r < total % 8) consecutive numbers
	const int first = xcd * (total >> 3) + min(xcd, total & 7);
	for(int q = (int) (blockIdx.x >> 3); q < share; q += nq) {
	int lane = lane0;
	__asm__ volatile("" : "+v"(lane));// (what a block derives from the lane number is formed per block: hoisted out of this loop it would sit in registers through the particle loop)
	const int bid = first + q;
	const int b	  = block_list ? block_list[bid] : bid;
	// ---- round trip 1: everything addressed by the block number alone (scalar loads)
	const int size		 = mv.size[b];
	const int row		 = mv.row_of[b];
	const int binoff_dst = mv.binoff_dst[b];
	// only_flag (MGSP interior pass): the launch covers ALL particle blocks in their own (spatially coherent) order and skips the ones the
	// halo-first pass has done - no indirection through a compacted list, whose atomic append order loses that coherence
	const int flag = only_flag ? only_flag[b] : 1;
	if(size == 0 || flag == 0) continue;// (:692-697)
	const int* list		= mv.list_in + (size_t) row * cfg.ppb;
	const float mass	= mv.mc.mass;
	const int key_shift = cfg.pid_bits;
	const int tag_shift = cfg.pid_bits + kKeyBits;
	// ---- round trip 2: the block's row of look-up results and the records of the first two iterations
	const int info = mv.blockinfo[(size_t) b * kInfoRow + lane];
	// (sliced list layout, mpm_kernels.hpp: a 64-slot slice holds its records in its first slots; idle lanes re-read the last one)
	int cnt_cur	   = slice_records_at(size, 0);
	int cnt_next   = 64 < size ? slice_records_at(size, 64) : 1;
	int rec_cur	   = list[min(lane, cnt_cur - 1)];
	int rec_next   = list[(64 < size ? 64 : 0) + min(lane, cnt_next - 1)];
	// (`info` stays in its register: lane l < 27 holds the bin offset of source block l, lanes 27..53 the destination block
	//  numbers, lanes 54..61 the eight grid blocks; they are read with __shfl = ds_bpermute, which costs no LDS space)
	for(int i = lane; i < kP2GArena2 + kP2GNodes; i += 64) p2g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
	__syncthreads();
	// ---- round trip 3: the 8 grid blocks (lane = cell -> 256-B rows per channel, :699-727) and the first 64 particles
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
	// Software prefetch: the particle data of iteration i+1 is requested at the top of iteration i (HBM latency under load
	// is 2-4 us).  The loads are unconditional - lanes past the end of the block re-read its last record - so that the
	// compiler can count them in s_waitcnt; their arrival is implied by the list-append atomics' results being consumed
	// at the end of the iteration (memory operations return in order).
	constexpr int ROW = NCH - REC;// floats per row entry: 0 (J-fluid), 1 (b21), 2 ({b21, log Jp})
	struct Prefetch {
		float4 q[REC / 4];// the particle record
		float row[ROW ? ROW : 1];
		int key;		  // the stencil base this particle was predicted to have after this step (its sort key)
	};
	auto fetch = [&](int rec, Prefetch& f) {
		const int tag	  = (rec >> tag_shift) & 31;
		const int sp	  = rec & (cfg.ppb - 1);
		const int sbin	  = __shfl(info, tag) + (sp >> 6);
		const float* bin  = mv.bins_src + (size_t) sbin * (kBin * NCH);
		const float4* src = reinterpret_cast<const float4*>(bin + (sp & 63) * REC);
		f.key			  = (rec >> key_shift) & 255;
#pragma unroll
		for(int d = 0; d < REC / 4; ++d) f.q[d] = src[d];
		if constexpr(ROW == 1) f.row[0] = bin[kBin * REC + (sp & 63)];
		if constexpr(ROW == 2) {
			const float2 t = reinterpret_cast<const float2*>(bin + kBin * REC)[sp & 63];
			f.row[0]	   = t.x;
			f.row[1]	   = t.y;
		}
	};
	Prefetch pf;
	fetch(rec_cur, pf);
#pragma unroll
	for(int lb = 0; lb < 8; ++lb) {
		const int cx = lane >> 4, cy = (lane >> 2) & 3, cz = lane & 3;// lane == cell of a 4x4x4 block
		const int ax = cx + ((lb & 4) ? 4 : 0) - 1, ay = cy + ((lb & 2) ? 4 : 0) - 1, az = cz + ((lb & 1) ? 4 : 0) - 1;
		if(((unsigned) ax < 6u) & ((unsigned) ay < 6u) & ((unsigned) az < 6u)) g2p[ax * kG2PStrideX + ay * kG2PStrideY + az * kG2PStrideZ] = gv[lb];
	}
	__syncthreads();
	// Software pipeline: the scatter of iteration i-1 (an ordered chain of 27 LDS round trips) is issued inside the material
	// update of iteration i; `pv` is the payload in flight, pv_code its stencil base (-1: none).
	P2GPayload pv;
	int pv_code = -1;
	int qn = 0;// entries in s_queue (wave-uniform)
	bool settled = true;// (wave-uniform) every particle so far stays in this block with the sort key it came with
#ifdef MPM_G2P2G_STATS
	int st_iter = 0, st_losers = 0, st_edge = 0, st_retry_iters = 0, st_partial = 0;
#endif
	for(int idx0 = 0;; idx0 += 64) {
		// one pass more than there are iterations: the last one only drains the pipeline (scatter of the last payload)
		const bool drain = idx0 >= size;
		bool win		 = false;
		const bool pv_in = pv_code >= 0;
		P2GPayload pl;// (only pl.contrib is used inside the iteration: -P F^T vol new_dt D^-1 dx; the payload itself is formed at the hand-over)
		float vel[3], A[9], nfd[3];
		bool in_arena = false;
		int ncode	  = -1;
		if(!drain) {
		MPM_MARK("L_top");
		const bool active = lane < cnt_cur;
#ifdef MPM_G2P2G_STATS
		st_partial += __popcll(__ballot(!active));
#endif
		const int pidib	  = idx0 + lane;// slot in the destination bins == position in the sorted order
		// ---- advection record -> source bin (:747-768): data was requested one iteration ago
		float pos[3] = {pf.q[0].x, pf.q[0].y, pf.q[0].z};
		float st[7];// J, or b {00, 11, 22, 10, 20, 21} (+ log Jp)
		st[0] = pf.q[0].w;
		if constexpr(MAT != 0) {
			st[1] = pf.q[1].x;
			st[2] = pf.q[1].y;
			st[3] = pf.q[1].z;
			st[4] = pf.q[1].w;
			st[5] = pf.row[0];
			if constexpr(ROW == 2) st[6] = pf.row[1];
		}
		const int okey	  = pf.key;// the sort key this particle was processed under
		const int slot_nn = idx0 + 128 < size ? idx0 + 128 : 0;
		const int cnt_nn  = idx0 + 128 < size ? slice_records_at(size, idx0 + 128) : 1;
		const int rec_nn  = list[slot_nn + min(lane, cnt_nn - 1)];
		// the next iteration's particle record is requested a whole iteration ahead: record loads of 64 scattered 64-B sectors
		// take long to return, and at three waves per SIMD the 16 registers are there
		fetch(rec_next, pf);
		rec_next = rec_nn;
		cnt_cur			   = cnt_next;
		cnt_next		   = cnt_nn;
		MPM_MARK("L_gather");
		// ---- stencil base + weights (:774-797) for ALL lanes (idle lanes of a last partial iteration carry a dummy
		//      position inside the block); offsets in cell units (exact: dx is a power of two)
		int base[3], arena[3];
		{
			float fd[3], w[3][3];
#pragma unroll
			for(int d = 0; d < 3; ++d) {
				const float p = pos[d];// (positions are stored in cell units: x / dx, exact for dx a power of two)
				base[d]		  = lround_pos(p) - 1;
				fd[d]		  = p - (float) base[d];
				bspline_weight_cells(fd[d], w[d]);
				arena[d] = ((base[d] - 1) & 3) + 1;
			}
			gather_apic(g2p + (arena[0] - 1) * kG2PStrideX + (arena[1] - 1) * kG2PStrideY + (arena[2] - 1) * kG2PStrideZ, w, fd, vel, A);
		}
		// Every lane runs the whole body: the idle lanes of a last partial iteration carry a dummy particle and write it into
		// the padding slots of the block's last bin (slot == pidib < 64 * ceil(size / 64), allocated but never read), which
		// keeps the 13 stores out of divergent control flow (countable for s_waitcnt).
		// ---- claim the stencil bases of the payload in flight: lanes whose base is unique in the wave (`win`) scatter in the
		//      chain threaded through this iteration's register-only arithmetic - re-bucketing (kPreSites call sites) and the
		//      material update - the others (and the edge lanes) afterwards
		MPM_MARK("L_claim");
		constexpr int kPreSites	   = 3;// chain sites in the re-bucketing
		constexpr int kStressSites = MAT == 0 ? 1 : (MAT == 1 ? kFcSites : (MAT == 2 ? kSandSites : kNaccSites));
		constexpr int kSites	   = kPreSites + kStressSites + 2;
		// (Round 5 tried a second claim round - a lane that loses in its own arena takes the other one if no first-round winner sits there: the losers
		//  stay at 1.29 % of the particles of the C3 flow and the launch is 1.5 % slower - they are keys with three or more particles in one slice, cells
		//  compressed beyond 2 S particles, for which two arenas are not enough; profiles/r05_ab_not_kept.txt.  Also tried there: persistent workgroups that
		//  pull further blocks from a per-XCD device counter - any data-dependent block loop makes the compiler spill 50-90 dwords in every instantiation.)
		const int pv_key = (pv_in ? code_key(pv_code) : 0) + (lane & 1) * 216;// even / odd lanes: separate arenas, separate claims
		if(pv_in) s_owner[pv_key] = (unsigned char) lane;
		__asm__ volatile("" ::: "memory");// another lane may have written the same byte: no store-to-load forwarding
		win = pv_in && !code_edge(pv_code) && (int) s_owner[pv_key] == lane;
		ScatterChain<kSites> chain(p2g + (win ? code_off(pv_code) + (lane & 1) * kP2GArena2 : 0), pv, mass, win);
		MPM_MARK("L_rebucket");
		// ---- advect (:838)
#pragma unroll
		for(int d = 0; d < 3; ++d) pos[d] = fmaf(vel[d], sk.dtp, pos[d]);// x += v dt, in cell units: the same bits as fma(v, dt, x) scaled by 2^bits
		// ---- new base, re-bucket (:852-866, add_advection particle_buffer.cuh:100-135).  The list-append atomics are
		//      issued BEFORE the stress computation, which hides their round trip to L2.
		int narena[3], dirv[3], pk[3];
		in_arena = active;
		const float pred = sk.pred;
#pragma unroll
		for(int d = 0; d < 3; ++d) {
			const float p	= pos[d];
			const int nbase = lround_pos(p) - 1;
			nfd[d]			= p - (float) nbase;
			narena[d]		= arena[d] + (nbase - base[d]);// new stencil base in the node cube of the block the particle came from
			in_arena &= (narena[d] >= 0) & (narena[d] <= 5);
			// the block the particle lands in: cube coordinates 1..4 belong to this block, 0 / 5.. to the neighbours
			// (== ((base - 1) >> 2) - ((nbase - 1) >> 2), base - 1 and arena - 1 being congruent modulo 4)
			dirv[d] = -((narena[d] - 1) >> 2);
			// sort key for the NEXT step: the stencil base predicted after one more advection with the current velocity, in the
			// cube of the block the particle is in after THIS step, clamped to 0..5 (a prediction: ties do not matter).
			// rint((pos + vel new_dt) / dx) - 1 - nbase == rint(fd + vel new_dt / dx) - 1 because nbase is an integer.
			const int step = (int) __builtin_rintf(fmaf(vel[d], pred, nfd[d]));
			pk[d]		   = min(max(((narena[d] - 1) & 3) + step, 0), 5);
		}
		if constexpr(kPreSites == 3) chain.template at<0>();
#if !(defined(MPM_EXPERIMENT) && defined(MPM_NO_FASTSTAY))// round 5: a wave whose particles all stay in this block skips the look-up of the destination block and the ballot arithmetic (-1.1 % at rest, -1.9 % on C2: profiles/r04_ab_faststay.txt); MPM_NO_FASTSTAY: the general path only (A/B)
		int ntag, dno, stay_leader, stay_rank;
		bool stay;
		int raw_stay = 0, raw_move = 0;
		int b_opaque = b;
		__asm__("" : "+v"(b_opaque));
		if(__all((dirv[0] | dirv[1] | dirv[2]) == 0)) {
			ntag		= kStay;
			dno			= active ? b : -1;
			stay		= active;
			stay_leader = 0;
			stay_rank	= lane;// (the active lanes are the first lanes of the slice)
			const int n_act = __popcll(__ballot(active));
			if(lane == 0) raw_stay = atomicAdd(&mv.out_count[b_opaque], n_act);
		} else {
			const bool dir_ok = ((unsigned) (dirv[0] + 1) < 3u) & ((unsigned) (dirv[1] + 1) < 3u) & ((unsigned) (dirv[2] + 1) < 3u);
			ntag			  = dir_ok ? (dirv[0] + 1) * 9 + (dirv[1] + 1) * 3 + dirv[2] + 1 : kStay;
			const int dno_raw = __shfl(info, 27 + ntag);
			dno				  = (active && dir_ok) ? dno_raw : -1;
			stay			  = dno >= 0 && ntag == kStay;
			const unsigned long long stay_m = __ballot(stay);
			stay_leader						= stay_m ? __ffsll((long long) stay_m) - 1 : 0;
			stay_rank						= (int) __builtin_amdgcn_mbcnt_hi((unsigned) (stay_m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) stay_m, 0u));
			if(stay_m != 0ull && lane == stay_leader) raw_stay = atomicAdd(&mv.out_count[b_opaque], __popcll(stay_m));
			if(dno >= 0 && !stay) raw_move = atomicAdd(&mv.out_count[dno], 1);
			if(active) {
				if(dno < 0) atomicAdd(&status[ST_LOST], 1);
				if(!in_arena) atomicAdd(&status[ST_ARENA], 1);
			}
		}
		if constexpr(kPreSites == 3) chain.template at<1>();
		const int pkey = pk[1] * 36 + pk[0] * 6 + pk[2];
		const int rec  = (ntag << tag_shift) | (pkey << key_shift) | pidib;
		settled		   = settled && __all(!active || (stay && pkey == okey));
#else
		const bool dir_ok = ((unsigned) (dirv[0] + 1) < 3u) & ((unsigned) (dirv[1] + 1) < 3u) & ((unsigned) (dirv[2] + 1) < 3u);
		const int ntag	  = dir_ok ? (dirv[0] + 1) * 9 + (dirv[1] + 1) * 3 + dirv[2] + 1 : kStay;
		const int dno_raw = __shfl(info, 27 + ntag);
		const int dno	  = (active && dir_ok) ? dno_raw : -1;
		if constexpr(kPreSites == 3) chain.template at<1>();
		const int pkey	= pk[1] * 36 + pk[0] * 6 + pk[2];
		const int rec	= (ntag << tag_shift) | (pkey << key_shift) | pidib;
		const bool stay = dno >= 0 && ntag == kStay;
		settled = settled && __all(!active || (stay && pkey == okey));
		// particles that stay in this block share one wave-aggregated atomic
		const unsigned long long stay_m = __ballot(stay);
		const int stay_leader			= stay_m ? __ffsll((long long) stay_m) - 1 : 0;
		const int stay_rank				= (int) __builtin_amdgcn_mbcnt_hi((unsigned) (stay_m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) stay_m, 0u));// set bits below this lane
		int raw_stay = 0, raw_move = 0;
		int b_opaque = b;
		__asm__("" : "+v"(b_opaque));// hide the uniform address: the compiler's atomic optimiser would broadcast the
									 // result with v_readfirstlane right here, i.e. wait for the round trip
		if(stay_m != 0ull && lane == stay_leader) raw_stay = atomicAdd(&mv.out_count[b_opaque], __popcll(stay_m));
		if(dno >= 0 && !stay) raw_move = atomicAdd(&mv.out_count[dno], 1);
		if(active) {
			if(dno < 0) atomicAdd(&status[ST_LOST], 1);// reference: particle silently lost (particle_buffer.cuh:105-113)
			if(!in_arena) atomicAdd(&status[ST_ARENA], 1);// (:877-885) contribution discarded
		}
#endif
		if constexpr(kPreSites == 3) chain.template at<2>();
		MPM_MARK("L_material");
		// ---- material update, store to the destination bin (slot == pidib: consecutive records) (:470-663)
		float* dbin = mv.bins_dst + (size_t) (binoff_dst + (pidib >> 6)) * (kBin * NCH);
		float4* dst = reinterpret_cast<float4*>(dbin + (pidib & 63) * REC);
		if constexpr(MAT == 0) {
			chain.template at<kPreSites + 0>();
			const float J = stress_jfluid(mv.mc, sk.ss.vol, sk.jdiv, sk.jvisc, st[0], A, pl.contrib);
			chain.template at<kPreSites + 1>();
			dst[0] = make_float4(pos[0], pos[1], pos[2], J);
		} else {
			// b <- G b G^T, G = I + dt grad v (:838-850: F <- (I + dt grad v) F); A is accumulated in cell units: dx * D^-1 = 4 / dx, settings.h:66
			float G[9], bo[6], bn[6];
#pragma unroll
			for(int d = 0; d < 9; ++d) G[d] = A[d] * sk.dts + ((d & 0x3) != 0 ? 0.f : 1.f);
			bool refl = (__float_as_uint(st[0]) & kReflBit) != 0u;// det F < 0 (mpm_device_math.hpp)
			bo[0]	  = fabsf(st[0]);
#pragma unroll
			for(int d = 1; d < 6; ++d) bo[d] = st[d];
			{// det G <= 0 needs an entry of dt grad v beyond 1/3 (Gershgorin; a CFL-limited step stays orders of magnitude below)
				// (four v_max3_f32 with |.| source modifiers)
				const float amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(fabsf(A[0]), fabsf(A[1])), fabsf(A[2])), __builtin_fmaxf(__builtin_fmaxf(fabsf(A[3]), fabsf(A[4])), fabsf(A[5]))),
												   __builtin_fmaxf(__builtin_fmaxf(fabsf(A[6]), fabsf(A[7])), fabsf(A[8])));
				const bool wild	 = !(amax < sk.refl_lim);
				if(__any(wild)) {
					if(wild) refl ^= det3(G) < 0.f;
				}
			}
			push_forward(G, bo, bn);
			chain.template at<kPreSites + 0>();
			float lj = 0.f;
			if constexpr(MAT == 1) {
				stress_fixed_corotated<kPreSites + 1>(sk.ss, bn, refl, pl.contrib, chain);
			} else if constexpr(MAT == 2) {
				lj = st[6];
				stress_sand<kPreSites + 1>(mv.mc, sk.ss, bn, refl, lj, pl.contrib, chain);
			} else {
				lj = st[6];
				stress_nacc<kPreSites + 1>(mv.mc, sk.ss, bn, refl, lj, pl.contrib, chain);
			}
			dst[0] = make_float4(pos[0], pos[1], pos[2], refl ? -bn[0] : bn[0]);
			dst[1] = make_float4(bn[1], bn[2], bn[3], bn[4]);
			// (lane == slot: one 256-B / 512-B row per wave)
			if constexpr(ROW == 1) dbin[kBin * REC + (pidib & 63)] = bn[5];
			if constexpr(ROW == 2) reinterpret_cast<float2*>(dbin + kBin * REC)[pidib & 63] = make_float2(bn[5], lj);
		}
		MPM_MARK("L_contrib");
		chain.template at<kSites - 1>();
		ncode = in_arena ? (narena[0] | (narena[1] << 4) | (narena[2] << 8)) : -1;
#if defined(MPM_EXPERIMENT) && defined(MPM_HACK_EDGEWIN)// timing experiment only: stencil bases on the cube's edge are moved inside, so no lane takes the serial path for that reason (wrong physics)
		if(in_arena) ncode = min(max(narena[0], 1), 4) | (min(max(narena[1], 1), 4) << 4) | (min(max(narena[2], 1), 4) << 8);
#endif
		MPM_MARK("L_append");
		// ---- list append: the atomics' results are in by now (and with them the next iteration's particle data)
		{
			const int basev = __shfl(raw_stay, stay_leader);
			if(dno >= 0) {
				const int slot = stay ? basev + stay_rank : raw_move;
				if(slot >= cfg.ppb)
					atomicOr(&status[ST_OVERFLOW], 2);// reference drops beyond 128 per cell (:122-130)
				else
					mv.list_out[((size_t) dno << cfg.pid_bits) + slot] = rec;// (ppb is a power of two: a 64-bit shift where the product took two v_mad_u64_u32)
			}
		}
		} else if(pv_in) {
			// draining pass: the last payload's winners run the 27 steps back to back
			const int pv_key = code_key(pv_code) + (lane & 1) * 216;
			s_owner[pv_key]	 = (unsigned char) lane;
			__asm__ volatile("" ::: "memory");
			win = !code_edge(pv_code) && (int) s_owner[pv_key] == lane;
			ScatterChain<1> chain(p2g + (win ? code_off(pv_code) + (lane & 1) * kP2GArena2 : 0), pv, mass, win);
			chain.template at<0>();
		}
		MPM_MARK("L_serial");
		// ---- the lanes that lost the claim and the edge lanes scatter now (rare); in the draining pass: every lane
		{
			const bool left = pv_in && !win;
#ifdef MPM_G2P2G_STATS
			if(!drain) {
				st_iter += 1;
				st_losers += __popcll(__ballot(left && !code_edge(pv_code)));
				st_edge += __popcll(__ballot(pv_in && code_edge(pv_code)));
				st_retry_iters += __any(left) ? 1 : 0;
			}
#endif
#if defined(MPM_EXPERIMENT) && defined(MPM_HACK_NOSERIAL)// timing experiment only: claim losers and edge lanes are dropped (wrong physics)
			if(false)
#else
			if(__any(left))
#endif
			{
				if constexpr(kQueue)
					serial_push(p2g, s_queue, qn, left, pv_code, pv, mass, lane, info, next_grid);
				else
					p2g_serial(p2g, left, pv_code, pv, mass, lane, info, next_grid);
			}
		}
		if(drain) break;
		MPM_MARK("L_handoff");
		// ---- hand the payload to the next iteration (:887-905).  It is formed here, after the last reader of the previous
		//      payload (chain, p2g_serial), so that it can be computed straight into the loop-carried registers (no copies).
		//      (:850) contrib = (A m - stress new_dt) D^-1, pre-multiplied by dx so that P2G can stay in cell units
		{
			const float am = sk.am;
#pragma unroll
			for(int d = 0; d < 3; ++d) {
				pv.fd[d] = nfd[d];
				pv.mv[d] = mass * vel[d];
			}
#pragma unroll
			for(int d = 0; d < 9; ++d) pv.contrib[d] = fmaf(A[d], am, pl.contrib[d]);// (pl.contrib = -stress new_dt D^-1 dx: StressScale)
		}
		pv_code = ncode;
	}
#ifdef MPM_G2P2G_STATS
	if(lane == 0) {
		atomicAdd(&status[24], st_iter);
		atomicAdd(&status[25], st_losers);
		atomicAdd(&status[26], st_edge);
		atomicAdd(&status[27], st_retry_iters);
		atomicAdd(&status[28], st_partial);
	}
#endif
	if constexpr(kQueue) {
		if(qn) serial_flush(p2g, s_queue, qn, mass, lane, info, next_grid);
	}
	{
		if(lane == 0) mv.keep[b] = settled ? size : -1;
	}
	__syncthreads();
	// ---- arena -> next grid: one hardware f32 atomic per touched node and channel (:907-936).  Lane = cell of one of the
	//      eight grid blocks, like the staging above: every atomic instruction covers (27 cells of) ONE 256-B channel row.
	//      (Walking the 216 arena nodes in arena order instead spreads each instruction over ~21 three-cell runs in up to
	//      eight blocks: five times the L2 atomic requests.)
	int lane_wb = lane;
	__asm__ volatile("" : "+v"(lane_wb));// (the cell coordinates and look-up lanes of the write-back are formed here: shared with the set-up's they would sit in ~15 registers through the particle loop)
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
#if defined(MPM_EXPERIMENT) && defined(MPM_HACK_NOWB)// timing experiment only: no write-back of the arenas (wrong physics)
		if(size_t(next_grid) == 1)
#else
		if(in && nb >= 0)
#endif
		{
			float* g = next_grid + (size_t) nb * 256 + lane_wb;
			if(v.x != 0.f) unsafeAtomicAdd(g, v.x);
			if(v.y != 0.f) unsafeAtomicAdd(g + 64, v.y);
			if(v.z != 0.f) unsafeAtomicAdd(g + 128, v.z);
			if(v.w != 0.f) unsafeAtomicAdd(g + 192, v.w);
		}
	}
	__syncthreads();// (a further block of this workgroup starts by clearing the arenas)
	}
}

}// namespace mpm
