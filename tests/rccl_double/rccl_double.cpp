// Test infrastructure, not product code: an in-process double of the nine RCCL entry points that claymore_amd/csrc/mpm_group.inc
// binds (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclAllReduce, ncclSend, ncclRecv, ncclGroupStart,
// ncclGroupEnd).  RCCL cannot host two ranks on one device, and the build box has one GPU: with this library (loaded through
// MPM_RCCL_LIBRARY) every rank is a thread of one process with its own engine context on the same GPU, and the RCCL branch of the
// group driver - the grouped send / receive of the halo exchange with its offsets and counts, the padded key all-gather, the
// all-reduce of the maximum velocity, their order on the streams - runs with world sizes > 1.  Data moves with device-to-device copies;
// every call synchronises the host (simple and obviously correct: nothing here is timed).  What it checks beyond moving the bytes:
// a receive whose size differs from the matching send fails, as does a collective entered with different sizes.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Msg {
	const void* ptr = nullptr;
	size_t bytes	= 0;
	bool full		= false;
};
struct Group {
	int n = 0, joined = 0, refs = 0;
	std::mutex m;
	std::condition_variable cv;
	int waiting = 0, gen = 0;
	std::vector<Msg> box;// [src * n + dst]
	const void* cptr[64] = {};
	size_t cbytes[64]	 = {};
	float cval[64]		 = {};
	void barrier() {
		std::unique_lock<std::mutex> lk(m);
		const int g = gen;
		if(++waiting == n) {
			waiting = 0;
			++gen;
			cv.notify_all();
		} else
			cv.wait(lk, [&] { return g != gen; });
	}
};
struct Comm {
	Group* g;
	int rank;
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
std::mutex g_m;
std::map<std::string, Group*> g_groups;
unsigned g_next_id = 1;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

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
#define HIPOK(e)                                     \
	do {                                             \
		if((e) != hipSuccess) return ncclUnhandledCudaError; \
	} while(0)

ncclResult_t run_ops(std::vector<Op>& ops) {
	// 1. every send: data complete, then announced
	for(Op& o: ops)
		if(o.send) {
			HIPOK(hipStreamSynchronize(o.stream));
			Group* g = o.comm->g;
			std::unique_lock<std::mutex> lk(g->m);
			Msg& b = g->box[(size_t) o.comm->rank * g->n + o.peer];
			g->cv.wait(lk, [&] { return !b.full; });
			b.ptr = o.sbuf, b.bytes = o.bytes, b.full = true;
			g->cv.notify_all();
		}
	// 2. every receive: wait for the matching send, copy, release it
	ncclResult_t rc = ncclSuccess;
	for(Op& o: ops)
		if(!o.send) {
			Group* g = o.comm->g;
			const void* src;
			size_t bytes;
			{
				std::unique_lock<std::mutex> lk(g->m);
				Msg& b = g->box[(size_t) o.peer * g->n + o.comm->rank];
				g->cv.wait(lk, [&] { return b.full; });
				src = b.ptr, bytes = b.bytes;
			}
			if(bytes != o.bytes) rc = ncclInvalidArgument;// the product's exchange is symmetric by construction: a mismatch is a bug
			else {
				HIPOK(hipMemcpyAsync(o.rbuf, src, bytes, hipMemcpyDeviceToDevice, o.stream));
				HIPOK(hipStreamSynchronize(o.stream));
			}
			std::unique_lock<std::mutex> lk(g->m);
			g->box[(size_t) o.peer * g->n + o.comm->rank].full = false;
			g->cv.notify_all();
		}
	// 3. every send: the peer has copied (the buffer may be reused)
	for(Op& o: ops)
		if(o.send) {
			Group* g = o.comm->g;
			std::unique_lock<std::mutex> lk(g->m);
			Msg& b = g->box[(size_t) o.comm->rank * g->n + o.peer];
			g->cv.wait(lk, [&] { return !b.full; });
		}
	return rc;
}
}// namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
	if(!id) return ncclInvalidArgument;
	std::lock_guard<std::mutex> lk(g_m);
	memset(id, 0, sizeof(*id));
	snprintf(id->internal, sizeof(id->internal), "rccl-double-%u-%p", g_next_id++, (void*) id);
	return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
	if(!comm || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
	const std::string key(id.internal, sizeof(id.internal));
	Group* g;
	{
		std::lock_guard<std::mutex> lk(g_m);
		Group*& slot = g_groups[key];
		if(!slot) {
			slot	= new Group();
			slot->n = nranks;
			slot->box.resize((size_t) nranks * nranks);
		}
		g = slot;
		if(g->n != nranks) return ncclInvalidArgument;
	}
	{
		std::unique_lock<std::mutex> lk(g->m);
		++g->joined, ++g->refs;
		g->cv.notify_all();
		g->cv.wait(lk, [&] { return g->joined >= g->n; });
	}
	*comm = reinterpret_cast<ncclComm_t>(new Comm {g, rank});
	return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
	Comm* c = reinterpret_cast<Comm*>(comm);
	if(!c) return ncclSuccess;
	bool last;
	{
		std::lock_guard<std::mutex> lk(c->g->m);
		last = --c->g->refs == 0;
	}
	if(last) {
		std::lock_guard<std::mutex> lk(g_m);
		for(auto it = g_groups.begin(); it != g_groups.end(); ++it)
			if(it->second == c->g) {
				g_groups.erase(it);
				break;
			}
		delete c->g;
	}
	delete c;
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
	if(!c || peer < 0 || peer >= c->g->n || peer == c->rank || !type_size(type)) return ncclInvalidArgument;
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
	if(!c || peer < 0 || peer >= c->g->n || peer == c->rank || !type_size(type)) return ncclInvalidArgument;
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
	Group* g		   = c->g;
	const size_t bytes = sendcount * type_size(type);
	HIPOK(hipStreamSynchronize(stream));
	g->cptr[c->rank] = sendbuff, g->cbytes[c->rank] = bytes;
	g->barrier();
	ncclResult_t rc = ncclSuccess;
	for(int p = 0; p < g->n; ++p) {
		if(g->cbytes[p] != bytes) rc = ncclInvalidArgument;
		else
			HIPOK(hipMemcpyAsync(static_cast<char*>(recvbuff) + (size_t) p * bytes, g->cptr[p], bytes, hipMemcpyDeviceToDevice, stream));
	}
	HIPOK(hipStreamSynchronize(stream));
	g->barrier();
	return rc;
}
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
	Comm* c = reinterpret_cast<Comm*>(comm);
	if(!c || type != ncclFloat32 || count != 1 || (op != ncclMax && op != ncclSum && op != ncclMin)) return ncclInvalidArgument;// (all the product needs)
	Group* g = c->g;
	HIPOK(hipStreamSynchronize(stream));
	float mine = 0.f;
	HIPOK(hipMemcpy(&mine, sendbuff, sizeof(float), hipMemcpyDeviceToHost));
	g->cval[c->rank] = mine;
	g->barrier();
	float r = g->cval[0];
	for(int p = 1; p < g->n; ++p) r = op == ncclMax ? (g->cval[p] > r ? g->cval[p] : r) : (op == ncclMin ? (g->cval[p] < r ? g->cval[p] : r) : r + g->cval[p]);
	g->barrier();// (everybody has read the slots before anybody enters the next collective)
	HIPOK(hipMemcpy(recvbuff, &r, sizeof(float), hipMemcpyHostToDevice));
	return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) {
	switch(r) {
		case ncclSuccess: return "no error";
		case ncclInvalidArgument: return "invalid argument (rccl double: size mismatch between a send and its receive, or unsupported call)";
		case ncclInvalidUsage: return "invalid usage";
		case ncclUnhandledCudaError: return "unhandled HIP error";
		default: return "error";
	}
}
}
