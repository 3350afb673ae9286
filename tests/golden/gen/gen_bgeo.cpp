// gen_bgeo.cpp — golden BGEO frame written by the reference's OWN partio (Externals/partio, compiled from its sources
// where they lie; see gen_bgeo.sh).  Follows write_partio<float, 3> (Library/MnSystem/IO/ParticleIO.hpp:14-29) call for call:
// one VECTOR attribute "position", addParticles, dataWrite per point, Partio::write.  The point set is read from a raw
// float32 file so that the test can feed the very same numbers to claymore_amd/host/particle_io.hpp.
#include <Partio.h>

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
	if(argc < 3) return 2;
	std::FILE* f = std::fopen(argv[1], "rb");
	if(!f) return 3;
	std::vector<float> xyz;
	float buf[3];
	while(std::fread(buf, sizeof(float), 3, f) == 3) xyz.insert(xyz.end(), buf, buf + 3);
	std::fclose(f);
	const int n = (int) (xyz.size() / 3);
	Partio::ParticlesDataMutable* parts = Partio::create();
	Partio::ParticleAttribute attrib	= parts->addAttribute("position", Partio::VECTOR, 3);
	parts->addParticles(n);
	for(int idx = 0; idx < n; ++idx) {
		float* val = parts->dataWrite<float>(attrib, idx);
		for(int k = 0; k < 3; k++) val[k] = xyz[3 * idx + k];
	}
	Partio::write(argv[2], *parts);
	parts->release();
	return 0;
}
