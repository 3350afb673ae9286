// Test infrastructure, not product code: an in-process double of the nine RCCL entry points that claymore_amd/csrc/mpm_group.inc
// binds (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclAllReduce, ncclSend, ncclRecv, ncclGroupStart,
// ncclGroupEnd).  RCCL cannot host two ranks on one device, and the build box has one GPU: with this library (loaded through
// MPM_RCCL_LIBRARY) every rank is a thread of one process with its own engine context on the same GPU, and the RCCL branch of the
// group driver - the grouped send / receive of the halo exchange with its offsets and counts, the padded key all-gather, the
// all-reduce of the maximum velocity, their order on the streams - runs with world sizes > 1.  Data moves with device-to-device copies.
// Default mode: STREAM-ORDERED like the real library - a send / receive / all-gather is complete in the order of the stream it was
// given and asynchronous to the host: the copies are enqueued behind the peer's event, nothing waits for the GPU (the ranks' host threads
// only meet to hand over pointers and events).  A dependency the product forgot - a kernel that reads a receive buffer from another stream
// without waiting for the exchange's event, a send buffer overwritten while the peer still copies from it - is then a real race on the
// GPU, as it would be with RCCL.  RCCL_DOUBLE_SYNC=1 restores the old behaviour (every call synchronises the host).  What it checks beyond
// moving the bytes: a receive whose size differs from the matching send fails, as does a collective entered with different sizes.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdlib>
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
	hipEvent_t ready = nullptr;// (stream-ordered mode) recorded by the sender where the data is complete
	hipEvent_t done	 = nullptr;// recorded by the receiver behind its copy
	bool done_posted = false;
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
	hipEvent_t cready[64] = {}, cdone[64] = {};
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
	std::vector<hipEvent_t> ev_send, ev_done;// per peer
	hipEvent_t ev_cready = nullptr, ev_cdone = nullptr;
};
bool sync_mode() {
	static const bool v = [] {
		const char* e = getenv("RCCL_DOUBLE_SYNC");
		return e && e[0] && e[0] != '0';
	}();
	return v;
}
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

ncclResult_t run_ops_stream_ordered(std::vector<Op>& ops) {
	// 1. every send: an event where the data is complete (in stream order), announced with the pointer
	for(Op& o: ops)
		if(o.send) {
			Comm* c = o.comm;
			HIPOK(hipEventRecord(c->ev_send[o.peer], o.stream));
			Group* g = c->g;
			std::unique_lock<std::mutex> lk(g->m);
			Msg& b = g->box[(size_t) c->rank * g->n + o.peer];
			g->cv.wait(lk, [&] { return !b.full; });
			b.ptr = o.sbuf, b.bytes = o.bytes, b.ready = c->ev_send[o.peer], b.done_posted = false, b.full = true;
			g->cv.notify_all();
		}
	// 2. every receive: the copy waits (on the GPU) for the sender's event; an event behind the copy goes back
	ncclResult_t rc = ncclSuccess;
	for(Op& o: ops)
		if(!o.send) {
			Comm* c	 = o.comm;
			Group* g = c->g;
			const void* src;
			size_t bytes;
			hipEvent_t ready;
			{
				std::unique_lock<std::mutex> lk(g->m);
				Msg& b = g->box[(size_t) o.peer * g->n + c->rank];
				g->cv.wait(lk, [&] { return b.full && !b.done_posted; });
				src = b.ptr, bytes = b.bytes, ready = b.ready;
			}
			if(bytes != o.bytes)
				rc = ncclInvalidArgument;
			else {
				HIPOK(hipStreamWaitEvent(o.stream, ready, 0));
				HIPOK(hipMemcpyAsync(o.rbuf, src, bytes, hipMemcpyDeviceToDevice, o.stream));
			}
			HIPOK(hipEventRecord(c->ev_done[o.peer], o.stream));
			std::unique_lock<std::mutex> lk(g->m);
			Msg& b = g->box[(size_t) o.peer * g->n + c->rank];
			b.done = c->ev_done[o.peer], b.done_posted = true;
			g->cv.notify_all();
		}
	// 3. every send: what follows on the sender's stream (it may overwrite the buffer) waits for the peer's copy
	for(Op& o: ops)
		if(o.send) {
			Comm* c	 = o.comm;
			Group* g = c->g;
			hipEvent_t done;
			{
				std::unique_lock<std::mutex> lk(g->m);
				Msg& b = g->box[(size_t) c->rank * g->n + o.peer];
				g->cv.wait(lk, [&] { return b.done_posted; });
				done   = b.done;
				b.full = false, b.done_posted = false;
				g->cv.notify_all();
			}
			HIPOK(hipStreamWaitEvent(o.stream, done, 0));
		}
	return rc;
}

ncclResult_t run_ops(std::vector<Op>& ops) {
	if(!sync_mode()) return run_ops_stream_ordered(ops);
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
	Comm* c = new Comm {g, rank, {}, {}, nullptr, nullptr};
	c->ev_send.resize(nranks, nullptr), c->ev_done.resize(nranks, nullptr);
	for(int p = 0; p < nranks; ++p) {
		HIPOK(hipEventCreateWithFlags(&c->ev_send[p], hipEventDisableTiming));
		HIPOK(hipEventCreateWithFlags(&c->ev_done[p], hipEventDisableTiming));
	}
	HIPOK(hipEventCreateWithFlags(&c->ev_cready, hipEventDisableTiming));
	HIPOK(hipEventCreateWithFlags(&c->ev_cdone, hipEventDisableTiming));
	*comm = reinterpret_cast<ncclComm_t>(c);
	return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
	Comm* c = reinterpret_cast<Comm*>(comm);
	if(!c) return ncclSuccess;
	(void) hipDeviceSynchronize();// (no copy of a peer may still wait on one of these events)
	for(hipEvent_t e: c->ev_send) (void) hipEventDestroy(e);
	for(hipEvent_t e: c->ev_done) (void) hipEventDestroy(e);
	(void) hipEventDestroy(c->ev_cready);
	(void) hipEventDestroy(c->ev_cdone);
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
	if(!sync_mode()) {
		HIPOK(hipEventRecord(c->ev_cready, stream));
		g->cptr[c->rank] = sendbuff, g->cbytes[c->rank] = bytes, g->cready[c->rank] = c->ev_cready;
		g->barrier();
		ncclResult_t rc = ncclSuccess;
		for(int p = 0; p < g->n; ++p) {
			if(g->cbytes[p] != bytes) {
				rc = ncclInvalidArgument;
				continue;
			}
			if(p != c->rank) HIPOK(hipStreamWaitEvent(stream, g->cready[p], 0));
			HIPOK(hipMemcpyAsync(static_cast<char*>(recvbuff) + (size_t) p * bytes, g->cptr[p], bytes, hipMemcpyDeviceToDevice, stream));
		}
		HIPOK(hipEventRecord(c->ev_cdone, stream));
		g->cdone[c->rank] = c->ev_cdone;
		g->barrier();
		for(int p = 0; p < g->n; ++p)
			if(p != c->rank) HIPOK(hipStreamWaitEvent(stream, g->cdone[p], 0));// the send buffer may be rewritten only behind every peer's copy
		g->barrier();// (everybody has taken the events before anybody posts the next collective)
		return rc;
	}
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
