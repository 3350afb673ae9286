// mgsp.cpp — host driver with the surface of claymore's `mgsp` executable (Projects/MGSP/mgsp.cu:84-104):
// hard-coded multi-GPU scenarios (mgsp.cu:34-81, cases 2 and 3: one lattice box per device), one engine context
// per device, MgspBenchmark::main_loop (mgsp_benchmark.cuh:361-559) through the C ABI, frames written as
// `model_dev[d]_frame[f].bgeo` (mgsp_benchmark.cuh:582-585).
//
// This is the in-process variant (the reference's structure: one process, N devices); halo blocks travel with
// hipMemcpyPeerAsync over xGMI between staging buffers, exactly where the reference uses cudaMemcpyPeerAsync
// (halo_buffer.cuh:54-59).  The one-process-per-GPU RCCL variant is claymore_amd/mgsp.py.
//
//   mgsp [--devices N] [--scenario 2|3] [--bits B] [--frames F] [--fps R] [--same-device] [--out DIR]
//        [--boundary PREFIX [--boundary-type sticky|slip|separate] [--friction MU]]
// --boundary reads PREFIX_sdf.bin, PREFIX_grad_{0,1,2}.bin (N^3 floats each) like MgspBenchmark::init_boundary
// (mgsp_benchmark.cuh:257-266, boundary_condition.cuh:292-321) and installs the collision object on every device.
// --same-device puts every context on GPU 0 (functional testing on a single GPU).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/claymore_amd.h"
#include "particle_io.hpp"

#define HIPCHK(e)                                                                              \
	do {                                                                                       \
		hipError_t r_ = (e);                                                                   \
		if(r_ != hipSuccess) {                                                                 \
			std::fprintf(stderr, "%s: %s (%s:%d)\n", #e, hipGetErrorString(r_), __FILE__, __LINE__); \
			std::exit(EXIT_FAILURE);                                                           \
		}                                                                                      \
	} while(0)

struct Dev {
	int gpu		 = 0;
	mpm_ctx* ctx = nullptr;
	hipStream_t compute = nullptr, comm = nullptr;
	size_t n = 0;
	int* keys_out = nullptr;// this device's neighbor keys (staging for the "all-gather")
	std::vector<int*> keys_in;// per peer: that peer's keys, copied here
	std::vector<int*> send_k, recv_k;
	std::vector<float*> send_b, recv_b;
	std::vector<int> send_n;
	size_t cap_blocks = 0;
	hipEvent_t sent;
};

static void check(Dev& d, int rc) {
	if(rc) {
		std::fprintf(stderr, "mgsp[%d]: status %d: %s\n", d.gpu, rc, mpm_last_error(d.ctx));
		std::exit(EXIT_FAILURE);
	}
}

// compute_dt of the MGSP project (Projects/MGSP/utility_funcs.hpp:32-55): CFL 0.3 and the 0.51 frame-remainder rule
static float compute_dt_mgsp(float max_vel, float cur, float next, float dt_default, float dx) {
	if(next < cur) return 0.f;
	float dt = dt_default;
	if(max_vel > 0.f) {
		max_vel = dx * 0.3f / max_vel;
		if(max_vel < dt_default) dt = max_vel;
	}
	if(cur + dt >= next) {
		dt = next - cur;
	} else {
		max_vel = (next - cur) * 0.51f;
		if(max_vel < dt) dt = max_vel;
	}
	return dt;
}

int main(int argc, char** argv) {
	int ndev = 2, scenario = 2, bits = 8, frames = 2, fps = 48;
	bool same = false;
	std::string out = ".", boundary, boundary_type = "sticky";
	float friction = 0.3f;
	for(int i = 1; i < argc; ++i) {
		auto is = [&](const char* s) { return !std::strcmp(argv[i], s) && i + 1 < argc; };
		if(is("--devices")) ndev = std::atoi(argv[++i]);
		else if(is("--scenario")) scenario = std::atoi(argv[++i]);
		else if(is("--bits")) bits = std::atoi(argv[++i]);
		else if(is("--frames")) frames = std::atoi(argv[++i]);
		else if(is("--fps")) fps = std::atoi(argv[++i]);
		else if(is("--out")) out = argv[++i];
		else if(is("--boundary")) boundary = argv[++i];
		else if(is("--boundary-type")) boundary_type = argv[++i];
		else if(is("--friction")) friction = (float) std::atof(argv[++i]);
		else if(!std::strcmp(argv[i], "--same-device")) same = true;
	}
	int ngpu = 0;
	HIPCHK(hipGetDeviceCount(&ngpu));
	if(ndev < 1 || ndev > 32 || (!same && ndev > ngpu)) {
		std::fprintf(stderr, "need %d devices, found %d (use --same-device for functional runs)\n", ndev, ngpu);
		return 1;
	}
	const float dx = 1.f / (float) (1 << bits);
	const int N	   = 1 << bits;
	std::vector<Dev> devs(ndev);
	// ---- init_models, mgsp.cu:51-79
	for(int d = 0; d < ndev; ++d) {
		Dev& D = devs[d];
		D.gpu  = same ? 0 : d;
		mpm_config cfg;
		mpm_default_config(bits, &cfg);
		cfg.gravity = -9.8f * 0.5f;// Projects/MGSP/settings.h:108
		cfg.cfl		= 0.3f;
		cfg.max_ppc = 32;
		check(D, mpm_create(&cfg, D.gpu, &D.ctx));
		int lo[3], hi[3];
		if(scenario == 3) {
			const int LEN = 72 * N / 256, STRIDE = N / 2, o = 18 * N / 256;
			lo[0] = o + ((d & 1) ? STRIDE : 0);
			lo[1] = o;
			lo[2] = o + ((d & 2) ? STRIDE : 0);
			hi[0] = lo[0] + LEN;
			hi[1] = lo[1] + LEN / 3;
			hi[2] = lo[2] + LEN;
		} else {
			const int LEN = 54 * N / 256, STRIDE = 56 * N / 256, o = 18 * N / 256;
			lo[0] = o + ((d & 1) ? STRIDE : 0) + (d >> 1) * 0;
			lo[1] = o + (d >> 1) * STRIDE;
			lo[2] = o;
			for(int k = 0; k < 3; ++k) hi[k] = lo[k] + LEN;
		}
		pio::Points pts = pio::sample_lattice(dx, lo, hi, [](const std::array<float, 3>&) { return true; });
		mpm_material_params p;
		mpm_default_material(MPM_FIXED_COROTATED, bits, &p);
		p.volume	   = dx * dx * dx / 8.f;
		const float v0[3] = {0.f, 0.f, 0.f};
		int id;
		check(D, mpm_add_model(D.ctx, MPM_FIXED_COROTATED, &p, pts[0].data(), pts.size(), v0, &id));
		D.n = pts.size();
		std::printf("init model on device %d (gpu %d) with %zu particles\n", d, D.gpu, D.n);
		pio::write_bgeo(out + "/model_dev[" + std::to_string(d) + "]_frame[0].bgeo", pts[0].data(), pts.size());
	}
	if(!boundary.empty()) {// init_boundary (mgsp_benchmark.cuh:257-266)
		const size_t n = (size_t) N * N * N;
		std::vector<float> field[4];
		const char* suffix[4] = {"_sdf.bin", "_grad_0.bin", "_grad_1.bin", "_grad_2.bin"};
		for(int c = 0; c < 4; ++c) {
			field[c].resize(n);
			const std::string fn = boundary + suffix[c];
			FILE* f				 = std::fopen(fn.c_str(), "rb");
			const size_t got	 = f ? std::fread(field[c].data(), sizeof(float), n, f) : 0;
			if(f) std::fclose(f);
			if(got != n) {
				std::fprintf(stderr, "Error in loading file [%s]: read in %zu entries, should be %zu\n", fn.c_str(), got, n);// boundary_condition.cuh:304
				return 1;
			}
		}
		mpm_collision_object obj;
		mpm_default_collision_object(&obj);
		obj.type	 = boundary_type == "slip" ? MPM_BOUNDARY_SLIP : (boundary_type == "separate" ? MPM_BOUNDARY_SEPARATE : MPM_BOUNDARY_STICKY);
		obj.friction = friction;
		for(auto& D: devs) check(D, mpm_set_collision_object(D.ctx, &obj, field[0].data(), field[1].data(), field[2].data(), field[3].data()));
		std::printf("[Collision Object] %s, %s\n", boundary.c_str(), boundary_type.c_str());
	}
	for(auto& D: devs) check(D, mpm_initial_setup(D.ctx));
	// staging buffers sized by the block capacity implied by the initial counts
	for(auto& D: devs) {
		HIPCHK(hipSetDevice(D.gpu));
		void *cs, *ms;
		mpm_streams(D.ctx, &cs, &ms);
		D.compute = (hipStream_t) cs;
		D.comm	  = (hipStream_t) ms;
		mpm_counts c;
		mpm_get_counts(D.ctx, &c);
		D.cap_blocks = (size_t) c.exterior_blocks * 4 + 4096;
		HIPCHK(hipMalloc((void**) &D.keys_out, sizeof(int) * 3 * D.cap_blocks));
		D.keys_in.assign(ndev, nullptr);
		D.send_k.assign(ndev, nullptr);
		D.recv_k.assign(ndev, nullptr);
		D.send_b.assign(ndev, nullptr);
		D.recv_b.assign(ndev, nullptr);
		D.send_n.assign(ndev, 0);
		HIPCHK(hipEventCreateWithFlags(&D.sent, hipEventDisableTiming));
	}
	size_t cap_all = 0;
	for(auto& D: devs) cap_all = std::max(cap_all, D.cap_blocks);
	for(int d = 0; d < ndev; ++d)
		for(int p = 0; p < ndev; ++p) {
			if(p == d) continue;
			HIPCHK(hipSetDevice(devs[d].gpu));
			HIPCHK(hipMalloc((void**) &devs[d].keys_in[p], sizeof(int) * 3 * cap_all));
			HIPCHK(hipMalloc((void**) &devs[d].send_k[p], sizeof(int) * 3 * cap_all));
			HIPCHK(hipMalloc((void**) &devs[d].recv_k[p], sizeof(int) * 3 * cap_all));
			HIPCHK(hipMalloc((void**) &devs[d].send_b[p], sizeof(float) * 256 * cap_all));
			HIPCHK(hipMalloc((void**) &devs[d].recv_b[p], sizeof(float) * 256 * cap_all));
			if(!same) {
				int can = 0;
				if(hipDeviceCanAccessPeer(&can, devs[d].gpu, devs[p].gpu) != hipSuccess) can = 0;
				if(can) (void) hipDeviceEnablePeerAccess(devs[p].gpu, 0);// Cuda.cu:120-127 (already-enabled is not an error worth reporting)
			}
		}

	// halo_tagging, mgsp_benchmark.cuh:661-720
	auto tag = [&]() {
		std::vector<int> nb(ndev);
		for(int d = 0; d < ndev; ++d) check(devs[d], mpm_halo_keys(devs[d].ctx, devs[d].keys_out, (int) devs[d].cap_blocks, &nb[d]));
		for(int d = 0; d < ndev; ++d) {
			if((size_t) nb[d] > cap_all) {
				std::fprintf(stderr, "halo key buffer too small\n");
				std::exit(EXIT_FAILURE);
			}
			check(devs[d], mpm_sync(devs[d].ctx));
		}
		for(int d = 0; d < ndev; ++d)
			for(int p = 0; p < ndev; ++p)
				if(p != d) HIPCHK(hipMemcpyPeer(devs[d].keys_in[p], devs[d].gpu, devs[p].keys_out, devs[p].gpu, sizeof(int) * 3 * (size_t) nb[p]));
		for(int d = 0; d < ndev; ++d) {
			Dev& D = devs[d];
			check(D, mpm_halo_tag_begin(D.ctx));
			for(int p = 0; p < ndev; ++p)
				if(p != d) check(D, mpm_halo_tag_peer(D.ctx, p, D.keys_in[p], nb[p]));
			int nh, sc[32];
			check(D, mpm_halo_tag_end(D.ctx, &nh, sc));
			for(int p = 0; p < ndev; ++p) D.send_n[p] = p == d ? 0 : sc[p];
		}
	};
	// collect_halo_grid_blocks / reduce_halo_grid_blocks, mgsp_benchmark.cuh:723-776
	auto exchange_begin = [&](int gid) {
		for(int d = 0; d < ndev; ++d) {
			Dev& D = devs[d];
			HIPCHK(hipSetDevice(D.gpu));
			for(int p = 0; p < ndev; ++p) {
				const int n = D.send_n[p];
				if(p == d || n == 0) continue;
				int ns;
				check(D, mpm_halo_collect(D.ctx, p, gid, D.send_k[p], D.send_b[p], (int) cap_all, &ns));
				HIPCHK(hipMemcpyPeerAsync(devs[p].recv_k[d], devs[p].gpu, D.send_k[p], D.gpu, sizeof(int) * 3 * (size_t) n, D.comm));
				HIPCHK(hipMemcpyPeerAsync(devs[p].recv_b[d], devs[p].gpu, D.send_b[p], D.gpu, sizeof(float) * 256 * (size_t) n, D.comm));
			}
			HIPCHK(hipEventRecord(D.sent, D.comm));
		}
	};
	auto exchange_end = [&](int gid) {
		for(int d = 0; d < ndev; ++d) {
			Dev& D = devs[d];
			HIPCHK(hipSetDevice(D.gpu));
			bool any = false;
			for(int p = 0; p < ndev; ++p) {
				if(p == d) continue;
				const int n = devs[p].send_n[d];// what p sent to me (== what I sent to p: the relation is symmetric)
				if(n == 0) continue;
				HIPCHK(hipStreamWaitEvent(D.comm, devs[p].sent, 0));
				check(D, mpm_halo_reduce(D.ctx, gid, D.recv_k[p], D.recv_b[p], n));
				any = true;
			}
			if(!any) check(D, mpm_halo_reduce(D.ctx, gid, nullptr, nullptr, 0));
		}
	};

	tag();
	exchange_begin(0);// initial rasterised grids are summed once (mgsp_benchmark.cuh:653-654)
	exchange_end(0);
	for(auto& D: devs) check(D, mpm_sync(D.ctx));

	// main_loop, mgsp_benchmark.cuh:361-559
	const float dt_default = 1e-4f, spf = 1.f / (float) fps;
	float dt   = compute_dt_mgsp(0.f, 0.f, spf, dt_default, dx);
	long steps = 0;
	std::vector<float> buf;
	pio::AsyncWriter io;
	for(int frame = 1; frame <= frames; ++frame) {
		for(float t = 0.f; t < spf;) {
			float maxv2 = 0.f;
			for(auto& D: devs) {
				float m;
				check(D, mpm_grid_update(D.ctx, dt, &m));
				maxv2 = std::max(maxv2, m);// host max over GPUs, :410-416
			}
			if(std::isinf(maxv2)) {
				std::printf("Maximum velocity is infinity\n");
				return 1;
			}
			const float next_dt = compute_dt_mgsp(std::sqrt(maxv2), t, spf, dt_default, dx);
			for(auto& D: devs) check(D, mpm_g2p2g_halo(D.ctx, dt, next_dt));
			exchange_begin(1);
			for(auto& D: devs) check(D, mpm_g2p2g_interior(D.ctx, dt, next_dt));// overlaps with the peer copies
			exchange_end(1);
			for(auto& D: devs) check(D, mpm_rebuild_partition(D.ctx, nullptr));
			tag();
			t += dt;
			dt = next_dt;
			++steps;
		}
		for(int d = 0; d < ndev; ++d) {// output_model, :565-591
			Dev& D = devs[d];
			buf.resize(3 * D.n);
			size_t n = D.n;
			check(D, mpm_retrieve_positions(D.ctx, 0, buf.data(), &n));
			std::printf("total number of particles %zu\n", n);
			io.write_bgeo_async(out + "/model_dev[" + std::to_string(d) + "]_frame[" + std::to_string(frame) + "].bgeo", buf, n);// IO::insert_job, mgsp_benchmark.cuh:586-590
		}
		std::printf("frame %d done after %ld substeps\n", frame, steps);
	}
	io.flush();
	for(auto& D: devs) mpm_destroy(D.ctx);
	return 0;
}
