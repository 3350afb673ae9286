// mgsp.cpp — host driver with the surface of claymore's `mgsp` executable (Projects/MGSP/mgsp.cu:84-104):
// hard-coded multi-GPU scenarios (mgsp.cu:34-81, cases 2 and 3: one lattice box per device), one engine context
// per device, MgspBenchmark::main_loop (mgsp_benchmark.cuh:361-559) through the C ABI, frames written as
// `model_dev[d]_frame[f].bgeo` (mgsp_benchmark.cuh:582-585).
//
// The reference's structure: one process, N devices, one worker thread per device (mgsp_benchmark.cuh:309-356).  Each
// worker drives the library's MGSP loop (mpm_group_*, claymore_amd/csrc/mpm_group.inc).  Two transports for the halo blocks:
//   peer (default here: one process owns all devices, as in the reference) - hipMemcpyPeerAsync between the devices with peer access
//        enabled, the reference's own mechanism (cudaMemcpyPeerAsync, halo_buffer.cuh:54-59; Cuda.cu:120-127), RCCL not needed;
//   rccl - grouped ncclSend / ncclRecv over xGMI, the transport of the one-process-per-GPU launch (bench.py).
// --transport peer|rccl or MPM_GROUP_TRANSPORT selects.
//
//   mgsp [--devices N] [--scenario 2|3] [--bits B] [--frames F] [--fps R] [--same-device] [--transport peer|rccl] [--out DIR]
//        [--boundary PREFIX [--boundary-type sticky|slip|separate] [--friction MU]]
// --boundary reads PREFIX_sdf.bin, PREFIX_grad_{0,1,2}.bin (N^3 floats each) like MgspBenchmark::init_boundary
// (mgsp_benchmark.cuh:257-266, boundary_condition.cuh:292-321) and installs the collision object on every device.
// --same-device puts every context on GPU 0 (functional testing on a single GPU).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
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
	size_t n	 = 0;
};

static void check(Dev& d, int rc) {
	if(rc) {
		std::fprintf(stderr, "mgsp[%d]: status %d: %s\n", d.gpu, rc, mpm_last_error(d.ctx));
		std::exit(EXIT_FAILURE);
	}
}

int main(int argc, char** argv) {
	int ndev = 2, scenario = 2, bits = 8, frames = 2, fps = 48;
	bool same = false;
	std::string out = ".", boundary, boundary_type = "sticky";
	std::string transport = std::getenv("MPM_GROUP_TRANSPORT") ? std::getenv("MPM_GROUP_TRANSPORT") : "peer";
	float friction = 0.3f;
	for(int i = 1; i < argc; ++i) {
		auto is = [&](const char* s) { return !std::strcmp(argv[i], s) && i + 1 < argc; };
		if(is("--devices")) ndev = std::atoi(argv[++i]);
		else if(is("--scenario")) scenario = std::atoi(argv[++i]);
		else if(is("--bits")) bits = std::atoi(argv[++i]);
		else if(is("--frames")) frames = std::atoi(argv[++i]);
		else if(is("--fps")) fps = std::atoi(argv[++i]);
		else if(is("--out")) out = argv[++i];
		else if(is("--transport")) transport = argv[++i];
		else if(is("--boundary")) boundary = argv[++i];
		else if(is("--boundary-type")) boundary_type = argv[++i];
		else if(is("--friction")) friction = (float) std::atof(argv[++i]);
		else if(!std::strcmp(argv[i], "--same-device")) same = true;
	}
	if(transport != "peer" && transport != "rccl") {
		std::fprintf(stderr, "--transport must be peer or rccl\n");
		return 1;
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
		cfg.max_ppc = 128;// settings.h:75
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
	// One worker thread per device, as the reference (mgsp_benchmark.cuh:309-356), each driving the library's MGSP loop
	// (mpm_group_main_loop: halo-first G2P2G, grouped RCCL send / recv beside the interior G2P2G, all-gather of block keys,
	// adaptive dt from the maximum velocity over all devices).  --same-device puts all contexts on GPU 0 (functional runs on a
	// single-GPU box); RCCL cannot host two ranks on one device, so that always means the in-process transport.
	std::vector<mpm_group*> groups(ndev, nullptr);
	unsigned char ident[128] = {};
	const bool local = same || transport == "peer";
	std::printf("halo transport: %s\n", local ? (same ? "in-process (one device)" : "peer-direct (hipMemcpyPeerAsync)") : "rccl");
	if(local) {
		std::vector<mpm_ctx*> ctxs;
		for(auto& D: devs) ctxs.push_back(D.ctx);
		if(mpm_group_create_local(ctxs.data(), ndev, groups.data()) != MPM_OK) {
			std::fprintf(stderr, "mpm_group_create_local failed\n");
			return 1;
		}
	} else if(mpm_group_unique_id(ident) != MPM_OK) {
		std::fprintf(stderr, "RCCL is not available\n");
		return 1;
	}
	pio::AsyncWriter io;
	std::mutex io_mutex;
	std::atomic<int> failed {0};
	struct FrameCtx {
		Dev* D;
		int d;
		const std::string* out;
		pio::AsyncWriter* io;
		std::mutex* m;
	};
	auto on_frame = [](int frame, void* user) {// output_model, mgsp_benchmark.cuh:565-591
		FrameCtx* F = static_cast<FrameCtx*>(user);
		std::vector<float> buf(3 * F->D->n);
		size_t n = F->D->n;
		if(mpm_retrieve_positions(F->D->ctx, 0, buf.data(), &n) != MPM_OK) return;
		std::lock_guard<std::mutex> lk(*F->m);
		std::printf("total number of particles %zu\n", n);
		F->io->write_bgeo_async(*F->out + "/model_dev[" + std::to_string(F->d) + "]_frame[" + std::to_string(frame) + "].bgeo", buf, n);// IO::insert_job, :586-590
	};
	std::vector<std::thread> workers;
	std::vector<int> steps(ndev, 0);
	for(int d = 0; d < ndev; ++d)
		workers.emplace_back([&, d] {
			Dev& D = devs[d];
			if(!local && mpm_group_create(D.ctx, d, ndev, ident, &groups[d]) != MPM_OK) {
				std::fprintf(stderr, "device %d: %s\n", d, mpm_last_error(D.ctx));
				failed = 1;
				return;
			}
			FrameCtx F {&D, d, &out, &io, &io_mutex};
			int rc = mpm_group_initial_setup(groups[d]);
			if(rc == MPM_OK) rc = mpm_group_main_loop(groups[d], frames, fps, 1e-4f, on_frame, &F, &steps[d]);
			if(rc != MPM_OK) {
				std::fprintf(stderr, "device %d: status %d: %s\n", d, rc, mpm_group_last_error(groups[d]));
				failed = 1;
			}
		});
	for(auto& w: workers) w.join();
	if(failed) return 1;
	std::printf("%d frames done after %d substeps\n", frames, steps[0]);
	io.flush();
	for(auto* g: groups) mpm_group_destroy(g);
	for(auto& D: devs) mpm_destroy(D.ctx);
	return 0;
}
