// host_selftest.cpp — exercises the pieces of the host drivers that need no GPU (JSON reader, lattice sampler, BGEO
// writer).  usage: host_selftest <out.bgeo>  -> prints "json_ok <n_models> <dt>", "sphere <count>", "bgeo <bytes>"
#include <cstring>
#include <iostream>

#include "mini_json.hpp"
#include "particle_io.hpp"

int main(int argc, char** argv) {
	if(argc == 4 && !std::strcmp(argv[1], "--bgeo-from")) {// host_selftest --bgeo-from points.f32 out.bgeo: raw float32 xyz -> BGEO frame
		std::ifstream in(argv[2], std::ios::binary | std::ios::ate);
		if(!in) return 3;
		const size_t bytes = (size_t) in.tellg();
		std::vector<float> xyz(bytes / sizeof(float));
		in.seekg(0);
		in.read(reinterpret_cast<char*>(xyz.data()), (std::streamsize) bytes);
		return pio::write_bgeo(argv[3], xyz.data(), xyz.size() / 3) ? 0 : 4;
	}
	const char* text = R"({"simulation": {"gpuid": 0, "fps": 1200, "frames": 2, "default_dt": 5e-6},
	  "models": [{"type": "particle", "file": "two_dragons.sdf", "constitutive": "jfluid", "offset": [0.1, 0.1, 0.1],
	              "span": [1.0, 1.0, 1.0], "velocity": [0.0, -1.0, 0.0], "nested": {"a": [true, false, null, "s\"q"]}},
	             {"file": "sphere", "constitutive": "sand", "offset": [0.2, 0.1, 0.2], "span": [1, 1, 1], "velocity": [0, -1e0, 0]}]})";
	auto doc = mj::parse(text);
	std::cout << "json_ok " << (*doc)["models"].arr.size() << " " << (*doc)["simulation"]["default_dt"].number() << " "
			  << (*doc)["models"][0]["nested"]["a"][3].string() << " " << (*doc)["models"][1]["velocity"][1].number() << "\n";
	bool threw = false;
	try {
		mj::parse("{\"a\": [1, 2}");
	} catch(const std::exception&) {
		threw = true;
	}
	std::cout << "json_bad " << (threw ? "rejected" : "accepted") << "\n";
	const float dx = 1.f / 64.f;
	const int lo[3] = {20, 20, 20}, hi[3] = {44, 44, 44};
	auto pts = pio::sample_lattice(dx, lo, hi, [&](const std::array<float, 3>& p) {
		const float a = p[0] - 0.5f, b = p[1] - 0.5f, c = p[2] - 0.5f;
		return a * a + b * b + c * c <= (5 * dx) * (5 * dx);
	});
	std::cout << "sphere " << pts.size() << "\n";
	if(argc > 1) {
		const float xyz[6] = {0.25f, 0.5f, 0.75f, -1.5f, 2.0f, 3.25f};
		std::cout << "bgeo " << (pio::write_bgeo(argv[1], xyz, 2) ? "written" : "failed") << "\n";
		{// the IO worker: 8 frames queued, flush() returns only when all of them are on disk
			pio::AsyncWriter io;
			for(int f = 0; f < 8; ++f) io.write_bgeo_async(std::string(argv[1]) + ".async" + std::to_string(f), std::vector<float>(xyz, xyz + 6), 2);
			io.flush();
			int ok = 0;
			for(int f = 0; f < 8; ++f) {
				std::ifstream in(std::string(argv[1]) + ".async" + std::to_string(f), std::ios::binary | std::ios::ate);
				ok += in && in.tellg() == std::ifstream::pos_type(41 + 2 * 16 + 2);
			}
			std::cout << "async " << ok << "\n";
		}
	}
	return 0;
}
