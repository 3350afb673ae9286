// particle_io.hpp — position-only BGEO frames and model sampling for the host drivers.
//
// write_bgeo: the wire format partio emits for the reference's write_partio (Library/MnSystem/IO/ParticleIO.hpp:14-29,
// Externals/partio/io/BGEO.cpp:311-407): Houdini classic BGEO, big-endian: magic 'Bgeo', 'V', version 5, nPoints,
// nPrims 0, nPointGroups 0, nPrimGroups 0, nPointAttrib 0, nVertexAttrib 0, nPrimAttrib 0, nAttrib 0, then per point
// x y z w(=1) as float32, then the two trailing bytes 0x00 0xff.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace pio {
using Points = std::vector<std::array<float, 3>>;

inline void put_be32(std::vector<unsigned char>& o, uint32_t v) {
	o.push_back((unsigned char) (v >> 24));
	o.push_back((unsigned char) (v >> 16));
	o.push_back((unsigned char) (v >> 8));
	o.push_back((unsigned char) v);
}
inline void put_bef(std::vector<unsigned char>& o, float f) {
	uint32_t u;
	static_assert(sizeof(u) == sizeof(f), "");
	std::memcpy(&u, &f, 4);
	put_be32(o, u);
}
inline bool write_bgeo(const std::string& filename, const float* xyz, size_t n) {
	std::vector<unsigned char> o;
	o.reserve(64 + n * 16);
	put_be32(o, ((((('B' << 8) | 'g') << 8) | 'e') << 8) | 'o');
	o.push_back('V');
	put_be32(o, 5);
	put_be32(o, (uint32_t) n);// nPoints
	for(int k = 0; k < 7; ++k) put_be32(o, 0);// nPrims, nPointGroups, nPrimGroups, nPointAttrib, nVertexAttrib, nPrimAttrib, nAttrib
	for(size_t i = 0; i < n; ++i) {
		put_bef(o, xyz[3 * i]);
		put_bef(o, xyz[3 * i + 1]);
		put_bef(o, xyz[3 * i + 2]);
		put_bef(o, 1.0f);
	}
	o.push_back(0x00);// "beginExtra" / "endExtra" markers partio appends (BGEO.cpp:424-427)
	o.push_back(0xff);
	std::ofstream f(filename, std::ios::binary);
	if(!f) return false;
	f.write((const char*) o.data(), (std::streamsize) o.size());
	return (bool) f;
}

// The reference's lattice rule (Library/MnBase/Geometry/GeometrySampler.h:11-37): 8 particles per grid node at +-0.25 dx.
template<typename Inside>
inline Points sample_lattice(float dx, const int lo[3], const int hi[3], Inside inside) {
	Points out;
	for(int i = lo[0]; i < hi[0]; ++i)
		for(int j = lo[1]; j < hi[1]; ++j)
			for(int k = lo[2]; k < hi[2]; ++k)
				for(int a = -1; a <= 1; a += 2)
					for(int b = -1; b <= 1; b += 2)
						for(int c = -1; c <= 1; c += 2) {
							std::array<float, 3> p = {(float) (i * dx + a * 0.25 * dx), (float) (j * dx + b * 0.25 * dx), (float) (k * dx + c * 0.25 * dx)};
							if(inside(p)) out.push_back(p);
						}
	return out;
}

// Text level set as read by the reference's SampleGenerator::LoadSDF (Library/MnSystem/IO/PoissonDisk/SampleGenerator.h:68-110):
// "ni nj nk / minx miny minz / dx / phi[ni*nj*nk]" with i fastest.
struct Sdf {
	int n[3]	 = {0, 0, 0};
	float mn[3]	 = {0, 0, 0};
	float dx	 = 0;
	std::vector<float> phi;
	bool load(const std::string& fn) {
		std::ifstream f(fn);
		if(!f) return false;
		f >> n[0] >> n[1] >> n[2] >> mn[0] >> mn[1] >> mn[2] >> dx;
		phi.resize((size_t) n[0] * n[1] * n[2]);
		for(auto& v: phi) f >> v;
		return (bool) f;
	}
	float at(int i, int j, int k) const { return phi[(size_t) i + (size_t) n[0] * ((size_t) j + (size_t) n[1] * k)]; }
	float sample(float x, float y, float z) const {// trilinear, +inf outside
		const float g[3] = {(x - mn[0]) / dx, (y - mn[1]) / dx, (z - mn[2]) / dx};
		int c[3];
		float t[3];
		for(int d = 0; d < 3; ++d) {
			c[d] = (int) std::floor(g[d]);
			t[d] = g[d] - (float) c[d];
			if(c[d] < 0 || c[d] + 1 >= n[d]) return 1e30f;
		}
		float r = 0.f;
		for(int a = 0; a < 2; ++a)
			for(int b = 0; b < 2; ++b)
				for(int e = 0; e < 2; ++e) r += (a ? t[0] : 1 - t[0]) * (b ? t[1] : 1 - t[1]) * (e ? t[2] : 1 - t[2]) * at(c[0] + a, c[1] + b, c[2] + e);
		return r;
	}
};

// Asynchronous output: one worker thread drains a queue of jobs so that the simulation does not wait for the file system
// (the reference's IO singleton, Library/MnSystem/IO/IO.h:10-67: insert_job / flush).  The caller hands over the
// position array by value (moved into the job).
class AsyncWriter {
	std::mutex mut_;
	std::condition_variable cv_, idle_;
	std::deque<std::function<void()>> jobs_;
	bool running_ = true;
	int busy_	  = 0;
	std::thread th_;
	void worker() {
		for(;;) {
			std::function<void()> job;
			{
				std::unique_lock<std::mutex> lk(mut_);
				cv_.wait(lk, [this] { return !running_ || !jobs_.empty(); });
				if(jobs_.empty()) return;// !running_ and drained
				job = std::move(jobs_.front());
				jobs_.pop_front();
				busy_ = 1;
			}
			job();
			{
				std::lock_guard<std::mutex> lk(mut_);
				busy_ = 0;
			}
			idle_.notify_all();
		}
	}

public:
	AsyncWriter()
		: th_([this] { worker(); }) {}
	~AsyncWriter() {
		{
			std::lock_guard<std::mutex> lk(mut_);
			running_ = false;
		}
		cv_.notify_all();
		th_.join();// remaining jobs are written before the thread exits
	}
	void insert_job(std::function<void()> job) {
		{
			std::lock_guard<std::mutex> lk(mut_);
			jobs_.push_back(std::move(job));
		}
		cv_.notify_one();
	}
	void write_bgeo_async(std::string fn, std::vector<float> xyz, size_t n) {
		insert_job([fn = std::move(fn), xyz = std::move(xyz), n] { write_bgeo(fn, xyz.data(), n); });
	}
	void flush() {// IO::flush: wait until every queued job has been written
		std::unique_lock<std::mutex> lk(mut_);
		idle_.wait(lk, [this] { return jobs_.empty() && !busy_; });
	}
};
}// namespace pio
