// gmpm.cpp — host driver with the surface of claymore's `gmpm` executable (Projects/GMPM/gmpm.cu:168-209):
//     gmpm -f scene.json
// parses the same scene schema (gmpm.cu:60-165), samples the models, runs GmpmSimulator::main_loop
// (gmpm_simulator.cuh:303-592) through the C ABI of include/claymore_amd.h and writes position-only BGEO frames
// `model_id[i]_frame[f].bgeo` (gmpm_simulator.cuh:204,626-631).  All kernels live behind libclaymore_hip.so.
//
// Schema additions (the reference hard-codes them at compile time, Projects/GMPM/settings.h): simulation.domain_bits
// (default 8), simulation.max_ppc, simulation.gravity, simulation.output_dir; model.file may be "sphere" or "box"
// (analytic shapes on the reference's 8-per-node lattice, offset = min corner, span = extent) besides "*.sdf"
// (text level set; sampled on the same lattice where phi < 0 - deterministic, unlike the reference's rand()-based
// Poisson sampling, Library/MnSystem/IO/PoissonDisk/SampleGenerator.h:112-176).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/claymore_amd.h"
#include "mini_json.hpp"
#include "particle_io.hpp"

namespace {

struct ModelSpec {
	int material = MPM_FIXED_COROTATED;
	mpm_material_params p {};
	pio::Points pts;
	float v0[3] = {0, 0, 0};
};

[[noreturn]] void die(mpm_ctx* ctx, int rc) {
	// check_cuda_errors / std::abort behaviour of the reference: print and leave (HostUtils.hpp:33-44)
	std::fprintf(stderr, "gmpm: status %d: %s\n", rc, ctx ? mpm_last_error(ctx) : "(no context)");
	std::exit(EXIT_FAILURE);
}

int material_of(const std::string& s) {
	if(s == "jfluid") return MPM_J_FLUID;
	if(s == "fixed_corotated") return MPM_FIXED_COROTATED;
	if(s == "sand") return MPM_SAND;
	if(s == "nacc") return MPM_NACC;
	return -1;
}

float num(const mj::Value& v, const char* k, float dflt) {
	return v.has(k) ? (float) v[k].number() : dflt;
}

pio::Points sample_model(const mj::Value& model, int bits, const std::string& scene_dir) {
	const float dx = 1.f / (float) (1 << bits);
	float offset[3], span[3];
	for(int d = 0; d < 3; ++d) {
		offset[d] = (float) model["offset"][d].number();
		span[d]	  = (float) model["span"][d].number();
	}
	int lo[3], hi[3];
	for(int d = 0; d < 3; ++d) {
		lo[d] = (int) std::floor(offset[d] / dx) - 1;
		hi[d] = (int) std::ceil((offset[d] + span[d]) / dx) + 2;
	}
	const std::string file = model["file"].string();
	if(file == "box") {
		return pio::sample_lattice(dx, lo, hi, [&](const std::array<float, 3>& p) {
			for(int d = 0; d < 3; ++d)
				if(p[d] < offset[d] || p[d] >= offset[d] + span[d]) return false;
			return true;
		});
	}
	if(file == "sphere") {
		const float c[3] = {offset[0] + 0.5f * span[0], offset[1] + 0.5f * span[1], offset[2] + 0.5f * span[2]};
		const float r	 = 0.5f * std::min(span[0], std::min(span[1], span[2]));
		return pio::sample_lattice(dx, lo, hi, [&](const std::array<float, 3>& p) {
			const float a = p[0] - c[0], b = p[1] - c[1], e = p[2] - c[2];
			return a * a + b * b + e * e <= r * r;
		});
	}
	if(file.size() > 4 && file.substr(file.size() - 4) == ".sdf") {
		pio::Sdf sdf;
		std::string path = file;
		if(!sdf.load(path)) path = scene_dir + "/" + file;
		if(sdf.phi.empty() && !sdf.load(path)) {
			std::fprintf(stderr, "cannot read level set %s\n", file.c_str());
			std::exit(EXIT_FAILURE);
		}
		// read_sdf (ParticleIO.hpp:32-70): the level set box is scaled uniformly into `span` and moved to `offset`
		float ext[3], scale = 1e30f;
		for(int d = 0; d < 3; ++d) {
			ext[d] = (float) sdf.n[d] * sdf.dx - sdf.mn[d];
			scale  = std::min(scale, span[d] / ext[d]);
		}
		return pio::sample_lattice(dx, lo, hi, [&](const std::array<float, 3>& p) {
			const float q[3] = {(p[0] - offset[0]) / scale + sdf.mn[0], (p[1] - offset[1]) / scale + sdf.mn[1], (p[2] - offset[2]) / scale + sdf.mn[2]};
			return sdf.sample(q[0], q[1], q[2]) < 0.f;
		});
	}
	std::fprintf(stderr, "unknown model file '%s' (expected *.sdf, sphere or box)\n", file.c_str());
	std::exit(EXIT_FAILURE);
}

}// namespace

int main(int argc, char** argv) {
	std::string scene = "scene.json";
	for(int i = 1; i < argc; ++i) {
		if((!std::strcmp(argv[i], "-f") || !std::strcmp(argv[i], "--file")) && i + 1 < argc) scene = argv[++i];
	}
	std::ifstream in(scene);
	if(!in) {
		std::printf("file not exist %s\n", scene.c_str());
		return 1;
	}
	std::stringstream ss;
	ss << in.rdbuf();
	mj::ValuePtr doc;
	try {
		doc = mj::parse(ss.str());
	} catch(const std::exception& e) {
		std::fprintf(stderr, "%s\n", e.what());
		return 1;
	}
	const std::string scene_dir = scene.find('/') == std::string::npos ? "." : scene.substr(0, scene.rfind('/'));
	std::printf("load the scene file of size %zu\n", ss.str().size());

	const mj::Value& sim = (*doc)["simulation"];
	const int gpuid		 = (int) num(sim, "gpuid", 0);
	const int fps		 = (int) num(sim, "fps", 24);	   // DEFAULT_FPS, gmpm_simulator.cuh:27
	const int frames	 = (int) num(sim, "frames", 60);   // DEFAULT_FRAMES :28
	const float dt_def	 = num(sim, "default_dt", 1e-4f);// DEFAULT_DT :26
	const int bits		 = (int) num(sim, "domain_bits", 8);
	const std::string out_dir = sim.has("output_dir") ? sim["output_dir"].string() : ".";
	std::printf("simulation: gpuid[%d], defaultDt[%g], fps[%d], frames[%d]\n", gpuid, dt_def, fps, frames);

	mpm_config cfg;
	if(mpm_default_config(bits, &cfg)) die(nullptr, MPM_ERR_INVALID);
	cfg.max_ppc = (int) num(sim, "max_ppc", 128);
	cfg.gravity = num(sim, "gravity", cfg.gravity);
	mpm_ctx* ctx = nullptr;
	int rc		 = mpm_create(&cfg, gpuid, &ctx);
	if(rc) die(nullptr, rc);

	const mj::Value& models = (*doc)["models"];
	std::printf("has %zu models\n", models.arr.size());
	std::vector<size_t> counts;
	float max_v0 = 0.f;
	for(size_t mi = 0; mi < models.arr.size(); ++mi) {
		const mj::Value& m = models[mi];
		const int mat	   = material_of(m["constitutive"].string());
		if(mat < 0) {
			std::printf("Unknown constitutive: %s", m["constitutive"].string().c_str());
			continue;
		}
		std::printf("model constitutive[%s], file[%s]\n", m["constitutive"].string().c_str(), m["file"].string().c_str());
		mpm_material_params p;
		mpm_default_material(mat, bits, &p);
		// update_*_parameters, gmpm_simulator.cuh:211-254 (sand has no updater in the reference, gmpm.cu:134-135)
		if(mat != MPM_SAND) {
			p.rho	 = num(m, "rho", p.rho);
			p.volume = num(m, "volume", p.volume);
		}
		if(mat == MPM_FIXED_COROTATED || mat == MPM_NACC) {
			p.youngs_modulus = num(m, "youngs_modulus", p.youngs_modulus);
			p.poisson_ratio	 = num(m, "poisson_ratio", p.poisson_ratio);
		}
		if(mat == MPM_J_FLUID) {
			p.bulk		= num(m, "bulk_modulus", p.bulk);
			p.gamma		= num(m, "gamma", p.gamma);
			p.viscosity = num(m, "viscosity", p.viscosity);
		}
		if(mat == MPM_NACC) {
			p.beta = num(m, "beta", p.beta);
			p.xi   = num(m, "xi", p.xi);
		}
		pio::Points pts = sample_model(m, bits, scene_dir);
		float v0[3];
		for(int d = 0; d < 3; ++d) v0[d] = (float) m["velocity"][d].number();
		max_v0 = std::max(max_v0, std::sqrt(v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2]));
		int id = -1;
		rc	   = mpm_add_model(ctx, mat, &p, pts.empty() ? nullptr : pts[0].data(), pts.size(), v0, &id);
		if(rc) die(ctx, rc);
		std::printf("init %d-th model with %zu particles\n", id, pts.size());
		pio::write_bgeo(out_dir + "/model_id[" + std::to_string(id) + "]_frame[0].bgeo", pts.empty() ? nullptr : pts[0].data(), pts.size());
		counts.push_back(pts.size());
	}

	// main_loop, gmpm_simulator.cuh:303-592
	const float spf = 1.f / (float) fps;
	float dt		= mpm_compute_dt(ctx, max_v0, 0.f, spf, dt_def);
	rc				= mpm_initial_setup(ctx);
	if(rc) die(ctx, rc);
	mpm_counts c;
	mpm_get_counts(ctx, &c);
	std::printf("block count on device %d: %d, %d, %d\n", gpuid, c.particle_blocks, c.neighbor_blocks, c.exterior_blocks);
	float cur_time = 0.f;
	long steps	   = 0;
	std::vector<float> buf;
	pio::AsyncWriter io;
	const auto wall0 = std::chrono::steady_clock::now();
	for(int frame = 1; frame <= frames; ++frame) {
		for(float t = 0.f; t < spf;) {
			float next_dt = dt, max_vel = 0.f;
			rc = mpm_substep(ctx, dt, t, spf, dt_def, &next_dt, &max_vel);
			if(rc == MPM_ERR_NONFINITE) {
				std::cout << "Maximum velocity is infinity" << std::endl;
				frame = frames + 1;
				break;
			}
			if(rc) die(ctx, rc);
			t += dt;
			cur_time += dt;
			dt = next_dt;
			++steps;
		}
		if(frame > frames) break;
		mpm_get_counts(ctx, &c);
		mpm_timers tm;
		mpm_get_timers(ctx, &tm);
		std::printf("frame %d: t %.6f, %ld substeps, blocks %d/%d/%d, last substep: grid %.3f ms g2p2g %.3f ms partition %.3f ms\n", frame, cur_time, steps, c.particle_blocks, c.neighbor_blocks, c.exterior_blocks, tm.grid_update_ms, tm.g2p2g_ms, tm.partition_ms);
		for(size_t mi = 0; mi < counts.size(); ++mi) {// output_model, :594-634
			buf.resize(3 * counts[mi]);
			size_t n = counts[mi];
			rc		 = mpm_retrieve_positions(ctx, (int) mi, buf.data(), &n);
			if(rc) die(ctx, rc);
			std::printf("total number of particles %zu\n", n);
			// IO::insert_job (gmpm_simulator.cuh:626-632): the frame is written by the IO thread while the next frame is computed
			io.write_bgeo_async(out_dir + "/model_id[" + std::to_string(mi) + "]_frame[" + std::to_string(frame) + "].bgeo", buf, n);
		}
	}
	io.flush();// IO::flush() at the end of main_loop (gmpm_simulator.cuh:591)
	const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
	std::printf("done: %ld substeps in %.3f s\n", steps, wall);
	mpm_destroy(ctx);
	return 0;
}
