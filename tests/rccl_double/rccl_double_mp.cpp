// Test infrastructure, not product code: a MULTI-PROCESS double of the RCCL entry points that claymore_amd/csrc/mpm_group.inc binds.
// RCCL refuses two ranks on one device ("Duplicate GPU detected") and the build box has one GPU, so the product's multi-process launch -
// python -m torch.distributed.run --nproc-per-node N bench.py --gpus N: gloo rendez-vous, the unique id broadcast, one engine context per
// PROCESS, the C++ group loop, the max over ranks, the one JSON line - could never run before the driver's 8-GPU node does.  With this
// library (MPM_RCCL_LIBRARY) and bench.py --oversubscribe the N ranks are N processes that share the one GPU; the data of every call is
// staged through a file-backed shared mapping (device -> host by the sender, host -> device by the receiver) and every call synchronises
// the host: simple, slow and obviously correct - nothing here is timed.  (rccl_double.cpp is the in-process variant: ranks as threads.)
//
// Layout of the mapping: Header | world x world mail boxes | world data slots of slot_bytes (a rank's slot is cut into world + 1 equal lanes:
// lane d < world = what this rank sends to rank d, the last lane = its contribution to a collective - a send may still wait for its
// receiver when the sender enters the next all-gather).  Point-to-point: box[src][dst] carries two counters - sent and
// taken - so a send waits only for ITS receiver and a rank without halo traffic never blocks anybody (as with the real library).
// Every wait gives up after RCCL_DOUBLE_TIMEOUT_S (default 120 s) with ncclSystemError, so a broken run ends instead of hanging the box.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
struct Box {
	std::atomic<unsigned> sent, taken;
	size_t bytes;
};
struct Header {
	std::atomic<int> ready;// rank 0 has sized and initialised the mapping
	std::atomic<int> joined, refs;
	std::atomic<int> bar_count, bar_gen;
	int world;
	size_t slot_bytes;
	size_t cbytes[64];
	float cval[64];
};
struct Comm {
	Header* h	= nullptr;
	Box* box	= nullptr;
	char* data	= nullptr;
	size_t map_bytes = 0;
	int rank = 0, world = 0;
	std::string path;
	char* slot(int r) const { return data + (size_t) r * h->slot_bytes; }
	size_t lane_bytes() const { return (h->slot_bytes / (size_t) (world + 1)) & ~(size_t) 255; }
	char* coll(int r) const { return slot(r) + (size_t) world * lane_bytes(); }
};
struct Op {
	bool send;
	const void* sbuf;
	void* rbuf;
	size_t bytes;
	int peer;
	Comm* comm;
	hipStream_t stream;
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
std::atomic<unsigned> g_next_id {1};

double now_s() {
	timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}
double timeout_s() {
	const char* e = getenv("RCCL_DOUBLE_TIMEOUT_S");
	return e && atof(e) > 0 ? atof(e) : 120.0;
}
template<class F>
bool wait_for(F&& ok) {
	const double t0 = now_s(), lim = timeout_s();
	for(unsigned spin = 0; !ok(); ++spin) {
		if(spin < 200)
			sched_yield();
		else
			usleep(100);
		if((spin & 1023u) == 1023u && now_s() - t0 > lim) return false;
	}
	return true;
}
bool barrier(Comm* c) {
	Header* h	  = c->h;
	const int gen = h->bar_gen.load(std::memory_order_acquire);
	if(h->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
		h->bar_count.store(0, std::memory_order_relaxed);
		h->bar_gen.store(gen + 1, std::memory_order_release);
		return true;
	}
	return wait_for([&] { return h->bar_gen.load(std::memory_order_acquire) != gen; });
}
size_t type_size(ncclDataType_t t) {
	switch(t) {
		case ncclInt8:
		case ncclUint8: return 1;
		case ncclFloat16: return 2;
		case ncclInt32:
		case ncclUint32:
		case ncclFloat32: return 4;
		case ncclInt64:
		case ncclUint64:
		case ncclFloat64: return 8;
		default: return 0;
	}
}
#define HIPOK(e)                                             \
	do {                                                     \
		if((e) != hipSuccess) return ncclUnhandledCudaError; \
	} while(0)

ncclResult_t run_ops(std::vector<Op>& ops) {
	// 1. every send: wait until the receiver has taken the previous message of this lane, stage the data, announce it
	for(Op& o: ops)
		if(o.send) {
			Comm* c = o.comm;
			if(o.bytes > c->lane_bytes()) {
				fprintf(stderr, "rccl double (mp): a message of %zu bytes does not fit its %zu-byte lane (RCCL_DOUBLE_SLOT_MB)\n", o.bytes, c->lane_bytes());
				return ncclInvalidArgument;
			}
			Box& b = c->box[(size_t) c->rank * c->world + o.peer];
			if(!wait_for([&] { return b.taken.load(std::memory_order_acquire) == b.sent.load(std::memory_order_relaxed); })) return ncclSystemError;
			HIPOK(hipStreamSynchronize(o.stream));
			HIPOK(hipMemcpy(c->slot(c->rank) + (size_t) o.peer * c->lane_bytes(), o.sbuf, o.bytes, hipMemcpyDeviceToHost));
			b.bytes = o.bytes;
			b.sent.fetch_add(1, std::memory_order_release);
		}
	// 2. every receive: wait for the matching send, copy, release the lane
	ncclResult_t rc = ncclSuccess;
	for(Op& o: ops)
		if(!o.send) {
			Comm* c = o.comm;
			Box& b	= c->box[(size_t) o.peer * c->world + c->rank];
			if(!wait_for([&] { return b.sent.load(std::memory_order_acquire) != b.taken.load(std::memory_order_relaxed); })) return ncclSystemError;
			if(b.bytes != o.bytes)
				rc = ncclInvalidArgument;// the product's exchange is symmetric by construction: a mismatch is a bug
			else {
				HIPOK(hipStreamSynchronize(o.stream));
				HIPOK(hipMemcpy(o.rbuf, c->slot(o.peer) + (size_t) c->rank * c->lane_bytes(), o.bytes, hipMemcpyHostToDevice));
			}
			b.taken.fetch_add(1, std::memory_order_release);
		}
	return rc;
}
}// namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
	if(!id) return ncclInvalidArgument;
	memset(id, 0, sizeof(*id));
	snprintf(id->internal, sizeof(id->internal), "rcclmp-%d-%u-%ld", (int) getpid(), g_next_id.fetch_add(1), (long) time(nullptr));
	return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
	if(!comm || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
	id.internal[sizeof(id.internal) - 1] = 0;
	const char* dir			= getenv("RCCL_DOUBLE_DIR");
	const std::string path	= std::string(dir && dir[0] ? dir : "/tmp") + "/" + id.internal + ".shm";
	const char* mb			= getenv("RCCL_DOUBLE_SLOT_MB");
	const size_t slot_bytes = (size_t) (mb && atoi(mb) > 0 ? atoi(mb) : 256) << 20;
	const size_t head		= (sizeof(Header) + 4095) & ~(size_t) 4095;
	const size_t boxes		= (sizeof(Box) * (size_t) nranks * nranks + 4095) & ~(size_t) 4095;
	const size_t total		= head + boxes + slot_bytes * (size_t) nranks;
	int fd					= -1;
	if(rank == 0) {
		fd = open(path.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
		if(fd < 0 || ftruncate(fd, (off_t) total) != 0) return ncclSystemError;// (sparse: pages exist once they are touched)
	} else {
		if(!wait_for([&] {
			   fd = open(path.c_str(), O_RDWR);
			   if(fd < 0) return false;
			   struct stat st;
			   if(fstat(fd, &st) == 0 && (size_t) st.st_size == total) return true;
			   close(fd);
			   fd = -1;
			   return false;
		   }))
			return ncclSystemError;
	}
	void* m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if(m == MAP_FAILED) return ncclSystemError;
	Comm* c		 = new Comm();
	c->h		 = static_cast<Header*>(m);
	c->box		 = reinterpret_cast<Box*>(static_cast<char*>(m) + head);
	c->data		 = static_cast<char*>(m) + head + boxes;
	c->map_bytes = total, c->rank = rank, c->world = nranks, c->path = path;
	if(rank == 0) {// (a fresh file is all zeros: counters, boxes and the barrier start at 0)
		c->h->world		 = nranks;
		c->h->slot_bytes = slot_bytes;
		c->h->ready.store(1, std::memory_order_release);
	} else if(!wait_for([&] { return c->h->ready.load(std::memory_order_acquire) == 1; }))
		return ncclSystemError;
	if(c->h->world != nranks) return ncclInvalidArgument;
	c->h->refs.fetch_add(1);
	c->h->joined.fetch_add(1, std::memory_order_acq_rel);
	if(!wait_for([&] { return c->h->joined.load(std::memory_order_acquire) >= nranks; })) return ncclSystemError;
	*comm = reinterpret_cast<ncclComm_t>(c);
	return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
	Comm* c = reinterpret_cast<Comm*>(comm);
	if(!c) return ncclSuccess;
	if(c->h->refs.fetch_sub(1) == 1) unlink(c->path.c_str());
	munmap(c->h, c->map_bytes);
	delete c;
	return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t comm) {
	return ncclCommDestroy(comm);
}
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
	const Comm* c = reinterpret_cast<const Comm*>(comm);
	if(!c || !count) return ncclInvalidArgument;
	*count = c->world;
	return ncclSuccess;
}
ncclResult_t ncclGroupStart() {
	++t_depth;
	return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
	if(t_depth <= 0) return ncclInvalidUsage;
	if(--t_depth > 0) return ncclSuccess;
	std::vector<Op> ops;
	ops.swap(t_ops);
	return run_ops(ops);
}
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
	Comm* c = reinterpret_cast<Comm*>(comm);
	if(!c || peer < 0 || peer >= c->world || peer == c->rank || !type_size(type)) return ncclInvalidArgument;
	t_ops.push_back(Op {true, sendbuff, nullptr, count * type_size(type), peer, c, stream});
	if(t_depth == 0) {
		std::vector<Op> ops;
		ops.swap(t_ops);
		return run_ops(ops);
	}
	return ncclSuccess;
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
	Comm* c = reinterpret_cast<Comm*>(comm);
	if(!c || peer < 0 || peer >= c->world || peer == c->rank || !type_size(type)) return ncclInvalidArgument;
	t_ops.push_back(Op {false, nullptr, recvbuff, count * type_size(type), peer, c, stream});
	if(t_depth == 0) {
		std::vector<Op> ops;
		ops.swap(t_ops);
		return run_ops(ops);
	}
	return ncclSuccess;
}
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
	Comm* c = reinterpret_cast<Comm*>(comm);
	if(!c || !type_size(type)) return ncclInvalidArgument;
	const size_t bytes = sendcount * type_size(type);
	if(bytes > c->lane_bytes()) return ncclInvalidArgument;
	HIPOK(hipStreamSynchronize(stream));
	HIPOK(hipMemcpy(c->coll(c->rank), sendbuff, bytes, hipMemcpyDeviceToHost));
	c->h->cbytes[c->rank] = bytes;
	if(!barrier(c)) return ncclSystemError;
	ncclResult_t rc = ncclSuccess;
	for(int p = 0; p < c->world; ++p) {
		if(c->h->cbytes[p] != bytes)
			rc = ncclInvalidArgument;
		else
			HIPOK(hipMemcpy(static_cast<char*>(recvbuff) + (size_t) p * bytes, c->coll(p), bytes, hipMemcpyHostToDevice));
	}
	if(!barrier(c)) return ncclSystemError;// (everybody has read the slots before anybody writes its next message)
	return rc;
}
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
	Comm* c = reinterpret_cast<Comm*>(comm);
	if(!c || type != ncclFloat32 || count != 1 || (op != ncclMax && op != ncclSum && op != ncclMin)) return ncclInvalidArgument;// (all the product needs)
	HIPOK(hipStreamSynchronize(stream));
	float mine = 0.f;
	HIPOK(hipMemcpy(&mine, sendbuff, sizeof(float), hipMemcpyDeviceToHost));
	c->h->cval[c->rank] = mine;
	if(!barrier(c)) return ncclSystemError;
	float r = c->h->cval[0];
	for(int p = 1; p < c->world; ++p) {
		const float v = c->h->cval[p];
		r			  = op == ncclMax ? (v > r ? v : r) : (op == ncclMin ? (v < r ? v : r) : r + v);
	}
	if(!barrier(c)) return ncclSystemError;
	HIPOK(hipMemcpy(recvbuff, &r, sizeof(float), hipMemcpyHostToDevice));
	return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) {
	switch(r) {
		case ncclSuccess: return "no error";
		case ncclInvalidArgument: return "invalid argument (rccl double, multi-process: size mismatch between a send and its receive, a message beyond its lane, or an unsupported call)";
		case ncclInvalidUsage: return "invalid usage";
		case ncclUnhandledCudaError: return "unhandled HIP error";
		case ncclSystemError: return "system error (rccl double, multi-process: a wait timed out or the shared mapping could not be set up)";
		default: return "error";
	}
}
}
