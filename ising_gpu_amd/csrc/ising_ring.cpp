// ising_ring.cpp -- the slab ring: 1-D slabs along Y, periodic, one row of each colour to each neighbour per colour
// half-sweep.  Replaces the reference's multi-GPU mechanism -- one managed allocation, remote loads of the
// two rows outside each slab, cudaDeviceSynchronize on every device after each colour (optimized/main.cu:1599-1658,
// loadTile :413-428, barriers :1779-1784 / :1800-1805) -- with explicit halo rows delivered on a SECOND stream per slab
// while the interior rows are being updated:
//
//   compute stream of slab k                              comm stream of slab k
//   ------------------------                              ---------------------
//   wait: halo rows of colour 1-c are in place
//   update rows 0 and Y-1 of colour c   (tiny launch)
//   record ev_edge[c]  ------------------------------->   wait ev_edge[c]
//   update rows 1 .. Y-2 of colour c    (the launch)       first row -> previous slab's bottom halo row
//                                                          last row  -> next slab's top halo row
//                                                          record ev_sent[c]
//
// Two transports move the rows:
//   RCCL  ncclSend/ncclRecv to the two ring neighbours inside one group (xGMI between GPUs).  Used when every slab has
//         its own device: by a single process driving n devices (ising_ring_*: ncclCommInitAll) and by one process per
//         GPU (ising_rank_*: ncclCommInitRank, the unique id travels through the caller's launcher).  librccl is opened
//         at run time (dlopen), so single-GPU users of libising_hip.so do not load it.
//   COPY  hipMemcpyPeerAsync on the comm stream (single process only): several slabs on one device, or no RCCL.
// Results do not depend on the decomposition or the transport: the Philox stream id uses the global row
// (optimized/main.cu:514).
#include "ising_ctx.hpp"
#include "ising_ipc.hpp"

#include <rccl/rccl.h> // types and prototypes only: the library itself is opened with dlopen
#include <dlfcn.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using ising_host::bind;
using ising_host::fail;

namespace {

// ------------------------------------------------------------------------------------------------ RCCL at run time
struct RcclApi {
	void *handle = nullptr;
	std::string error, path;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommInitAll) CommInitAll = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclCommAbort) CommAbort = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	decltype(&ncclGetVersion) GetVersion = nullptr;
};

RcclApi &rccl_state() {
	static RcclApi api;
	return api;
}

RcclApi *rccl() {
	RcclApi &api = rccl_state();
	static bool tried = false;
	if (tried) return api.handle ? &api : nullptr;
	tried = true;
	// RCCL must sit on the SAME HIP runtime as this library: a process can hold two (torch wheels ship their own
	// libamdhip64.so + librccl.so next to the system's), and streams / events of one mean nothing to the other.  So
	// the first candidates are the librccl files next to the libamdhip64 this library is bound to.
	void *h = nullptr;
	if (const char *env = getenv("ISING_RCCL_LIB")) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
	Dl_info info;
	if (!h && dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
		std::string dir(info.dli_fname);
		const size_t slash = dir.rfind('/');
		if (slash != std::string::npos) {
			dir.resize(slash + 1);
			for (const char *leaf : {"librccl.so.1", "librccl.so"}) {
				if (h) break;
				h = dlopen((dir + leaf).c_str(), RTLD_NOW | RTLD_LOCAL);
				if (h) api.path = dir + leaf;
			}
		}
	}
	const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	for (int i = 0; !h && i < 3; i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
	if (!h) {
		const char *e = dlerror();
		api.error = std::string("cannot open librccl.so.1: ") + (e ? e : "?");
		return nullptr;
	}
	bool ok = true;
	auto sym = [&](const char *name) -> void * {
		void *p = dlsym(h, name);
		if (!p) { ok = false; api.error = std::string("librccl lacks ") + name; }
		return p;
	};
#define RCCL_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(sym(name))
	RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
	RCCL_SYM(CommInitRank, "ncclCommInitRank");
	RCCL_SYM(CommInitAll, "ncclCommInitAll");
	RCCL_SYM(CommDestroy, "ncclCommDestroy");
	RCCL_SYM(CommAbort, "ncclCommAbort");
	RCCL_SYM(GroupStart, "ncclGroupStart");
	RCCL_SYM(GroupEnd, "ncclGroupEnd");
	RCCL_SYM(Send, "ncclSend");
	RCCL_SYM(Recv, "ncclRecv");
	RCCL_SYM(AllReduce, "ncclAllReduce");
	RCCL_SYM(GetErrorString, "ncclGetErrorString");
	RCCL_SYM(GetVersion, "ncclGetVersion");
#undef RCCL_SYM
	if (!ok) return nullptr;
	api.handle = h;
	return &api;
}

#define RCCL_TRY(expr)                                                                                              \
	do {                                                                                                            \
		ncclResult_t r_ = (expr);                                                                                   \
		if (r_ != ncclSuccess) return fail(ISING_E_RCCL, "%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString(r_), __FILE__, __LINE__); \
	} while (0)


// ------------------------------------------------------------------------------------------------ per-slab resources
} // namespace

int ising_host::ring_resources(ising_ctx *c) {
	if (c->comm && c->ev_int[1]) return ISING_OK; // all there (created once per slab)
	if (int rc = bind(c)) return rc;
	if (!c->comm) {
		int least = 0, greatest = 0; // numerically lower = higher priority
		HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
		HIP_TRY(hipStreamCreateWithPriority(&c->comm, hipStreamNonBlocking, c->pol.comm_priority ? greatest : 0)); // (A/B: default priority instead of the highest)
	}
	if (c->ballot && c->ghost() > 1 && !c->d_edge) { // deep exchange overlapped with the launches: two counters, a line apart
		HIP_TRY(hipMalloc((void **)&c->d_edge, 32 * sizeof(uint32_t)));
		HIP_TRY(hipMemset(c->d_edge, 0, 32 * sizeof(uint32_t)));
		c->edge_done_target = c->edge_go_epoch = 0;
		c->go_set = false;
	}
	for (int k = 0; k < 2; k++) {
		if (!c->ev_edge[k]) HIP_TRY(hipEventCreateWithFlags(&c->ev_edge[k], hipEventDisableTiming));
		if (!c->ev_sent[k]) HIP_TRY(hipEventCreateWithFlags(&c->ev_sent[k], hipEventDisableTiming));
		if (!c->ev_int[k]) HIP_TRY(hipEventCreateWithFlags(&c->ev_int[k], hipEventDisableTiming));
	}
	if (!c->ev_go) HIP_TRY(hipEventCreateWithFlags(&c->ev_go, hipEventDisableTiming));
	return ISING_OK;
}

namespace {

// The reference requires and enables all-to-all peer access (optimized/main.cu:1507-1537); a ring only needs the two
// neighbours.  Failure to enable is not fatal: hipMemcpyPeerAsync then stages through the host.
void ring_enable_peers(ising_ctx *c) {
	if (c->peers_enabled || !c->ring_prev) return;
	c->peers_enabled = true;
	if (hipSetDevice(c->cfg.device) != hipSuccess) return;
	const int peers[2] = {c->ring_prev->cfg.device, c->ring_next->cfg.device};
	for (int k = 0; k < 2; k++) {
		if (peers[k] == c->cfg.device || (k == 1 && peers[1] == peers[0])) continue;
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, c->cfg.device, peers[k]) == hipSuccess && can) {
			const hipError_t e = hipDeviceEnablePeerAccess(peers[k], 0);
			if (e != hipSuccess) (void)hipGetLastError(); // already enabled or unsupported: fall back silently
		}
	}
}

int ring_check(ising_ctx **ctxs, int n) {
	if (!ctxs || n < 1) return fail(ISING_E_ARG, "bad ring");
	for (int k = 0; k < n; k++) {
		const ising_ctx *c = ctxs[k];
		if (!c) return fail(ISING_E_ARG, "ring slot %d is null", k);
		if (c->cfg.nslabs != n || c->cfg.slab != k) return fail(ISING_E_ARG, "ring slot %d holds slab %d of %d", k, c->cfg.slab, c->cfg.nslabs);
		if (c->rank_mode) return fail(ISING_E_STATE, "slab %d is attached to a multi-process ring (ising_rank_attach)", k);
		if (c->lld != ctxs[0]->lld || c->cfg.Y != ctxs[0]->cfg.Y || c->cfg.X != ctxs[0]->cfg.X) return fail(ISING_E_ARG, "ring slabs differ in shape");
		// ballot and dense rows have the same size but another bit order: a mixed ring would exchange garbage silently
		if (c->layout() != ctxs[0]->layout()) return fail(ISING_E_ARG, "ring slabs differ in device layout (slab %d: %d, slab 0: %d)", k, c->layout(), ctxs[0]->layout());
		if (c->cfg.use_J != ctxs[0]->cfg.use_J || c->cfg.XSL != ctxs[0]->cfg.XSL || c->cfg.YSL != ctxs[0]->cfg.YSL)
			return fail(ISING_E_ARG, "ring slabs differ in couplings / sub-lattices");
	}
	return ISING_OK;
}

bool devices_distinct(ising_ctx **ctxs, int n) {
	for (int a = 0; a < n; a++)
		for (int b = a + 1; b < n; b++)
			if (ctxs[a]->cfg.device == ctxs[b]->cfg.device) return false;
	return true;
}

int rccl_init_all(ising_ctx **ctxs, int n) {
	RcclApi *api = rccl();
	if (!api) return fail(ISING_E_RCCL, "RCCL is not available: %s", rccl_state().error.c_str());
	std::vector<ncclComm_t> comms(n);
	std::vector<int> devs(n);
	for (int k = 0; k < n; k++) devs[k] = ctxs[k]->cfg.device;
	RCCL_TRY(api->CommInitAll(comms.data(), n, devs.data()));
	for (int k = 0; k < n; k++) {
		ctxs[k]->rccl_comm = comms[k];
		ctxs[k]->rccl_owner = true;
	}
	return ISING_OK;
}

void rccl_drop(ising_ctx *c, bool abort) {
	if (c->rccl_comm && c->rccl_owner) {
		if (RcclApi *api = rccl()) {
			(void)hipSetDevice(c->cfg.device);
			if (abort) (void)api->CommAbort(static_cast<ncclComm_t>(c->rccl_comm));
			else (void)api->CommDestroy(static_cast<ncclComm_t>(c->rccl_comm));
		}
	}
	c->rccl_comm = nullptr;
	c->rccl_owner = false;
}

// COPY transport: which stream a slab's copies travel on, and whether launches store their edge rows straight into the
// neighbours' halo rows.  Both follow from the slabs' devices and STREAMS, which the caller may change between ring calls
// (ising_set_stream, ising_use_private_stream), so they are evaluated at every ring call -- a stale "all slabs share one
// stream" would let launches on different streams write each other's halo rows with nothing ordering them.
void decide_copy_lanes(ising_ctx **ctxs, int n) {
	for (int k = 0; k < n; k++) {
		ising_ctx *c = ctxs[k];
		ring_enable_peers(c);
		c->copy_inline = c->ring_prev->cfg.device == c->cfg.device && c->ring_next->cfg.device == c->cfg.device;
		if (c->pol.ring_inline >= 0) c->copy_inline = c->copy_inline && c->pol.ring_inline != 0; // 0: exercise the two-stream schedule on one device
	}
	// All slabs on one device and one stream (the test configurations; `cuIsing -d N --devmap 0,0,...`): the stream
	// orders everything, so a launch can write rows 0 / Y-1 straight into the neighbours' halo rows.
	bool one = true;
	for (int k = 0; k < n; k++) one = one && ctxs[k]->copy_inline && ctxs[k]->stream == ctxs[0]->stream && ctxs[k]->cfg.device == ctxs[0]->cfg.device;
	// Slabs with ghost rows sweep faster in fused launches that take turns on the device, with copies of the ghost rows in
	// between (8 slabs of 131072 x 16384: 3455 vs 3394 flips/ns with the direct stores, 2 of 65536 x 32768: 3457 vs 3369), so the
	// direct stores are for slabs without them (dense / nibble layouts, caller-owned buffers).  ISING_RING_STORE=0/1 forces.
	bool ghosts = true;
	for (int k = 0; k < n; k++) ghosts = ghosts && ctxs[k]->ballot && ctxs[k]->ghost() > 1 && !ctxs[k]->cfg.XSL;
	if (ctxs[0]->pol.ring_store >= 0) one = one && ctxs[0]->pol.ring_store != 0;
	else one = one && !ghosts;
	for (int k = 0; k < n; k++) ctxs[k]->store_ring = one;
}

// Decides the transport of a single-process ring (once) and wires the neighbours.
int ring_bind(ising_ctx **ctxs, int n, int want = ISING_TRANSPORT_AUTO) {
	if (int rc = ring_check(ctxs, n)) return rc;
	for (int k = 0; k < n; k++) {
		ising_ctx *c = ctxs[k];
		c->ring_prev = ctxs[(k + n - 1) % n];
		c->ring_next = ctxs[(k + 1) % n];
		if (c->wrap) continue; // a single slab that wraps in place needs no transport
		if (int rc = ising_host::ring_resources(c)) return rc;
	}
	if (n == 1 && ctxs[0]->wrap) return ISING_OK;
	if (ctxs[0]->transport && want == ISING_TRANSPORT_AUTO) { // decided earlier
		if (ctxs[0]->transport == ISING_TRANSPORT_COPY) decide_copy_lanes(ctxs, n);
		return ISING_OK;
	}
	if (want == ISING_TRANSPORT_AUTO) want = ctxs[0]->pol.ring_transport; // (ISING_RING_TRANSPORT, as read when slab 0 was created)
	if (want == ISING_TRANSPORT_IPC) return fail(ISING_E_ARG, "the IPC transport serves one slab per process (ising_ipc_export / ising_ipc_attach)");
	int use = want;
	if (want == ISING_TRANSPORT_AUTO) use = (n > 1 && devices_distinct(ctxs, n) && rccl()) ? ISING_TRANSPORT_RCCL : ISING_TRANSPORT_COPY;
	if (use == ISING_TRANSPORT_RCCL) {
		if (!devices_distinct(ctxs, n)) return fail(ISING_E_ARG, "the RCCL transport needs one device per slab (RCCL refuses two ranks on one GPU)");
		bool have = true;
		for (int k = 0; k < n; k++) have = have && ctxs[k]->rccl_comm;
		if (!have) {
			for (int k = 0; k < n; k++) rccl_drop(ctxs[k], false);
			const int rc = rccl_init_all(ctxs, n);
			if (rc != ISING_OK) {
				if (want == ISING_TRANSPORT_RCCL) return rc;
				use = ISING_TRANSPORT_COPY; // AUTO: the copies always work
			}
		}
	}
	if (use == ISING_TRANSPORT_COPY) {
		for (int k = 0; k < n; k++) rccl_drop(ctxs[k], false);
		decide_copy_lanes(ctxs, n);
	} else {
		for (int k = 0; k < n; k++) ctxs[k]->copy_inline = ctxs[k]->store_ring = false;
	}
	for (int k = 0; k < n; k++) ctxs[k]->transport = use;
	return ISING_OK;
}

// ------------------------------------------------------------------------------------------------ the half-sweep stages
// rows of plane `color` (a spin colour or ISING_HAM_BLACK): first row, last row, the halo row above, the halo row below
struct EdgeRows {
	uint64_t *first, *last, *halo_top, *halo_bot;
	size_t bytes;
};
// `depth` rows on either side (1: the halo rows every schedule keeps current; G = the slab's ghost rows: sweep_deep)
EdgeRows edge_rows(const ising_ctx *c, int color, int depth = 1) {
	const size_t ld = (size_t)c->plane_ld(color), d = (size_t)depth;
	uint64_t *base = c->plane(color);
	return {base, base + ((size_t)c->cfg.Y - d) * ld, base - d * ld, base + (size_t)c->cfg.Y * ld, d * ld * sizeof(uint64_t)};
}

// Delivers the first/last rows of `color` of the local slabs on their comm streams; `after_edges`: the comm streams first
// wait for the slabs' edge-row kernels (ev_edge[color]).  Records ev_sent[color] (spin colours only).
int transfer(ising_ctx **ctxs, int n, int color, bool after_edges, int depth = 1) {
	if (ctxs[0]->cfg.XSL) return ISING_OK; // sub-lattices never reach across slabs
	const bool spin = color != ISING_HAM_BLACK;
	if (spin) for (int k = 0; k < n; k++) { ctxs[k]->ghost_depth[color] = depth; ctxs[k]->go_set = false; }
	// Copies between slabs of ONE device go on the slab's compute stream: a second stream buys nothing there (the copy
	// needs the same CUs / DMA engines the kernels hold) and every cross-stream event on a shared device is a bubble.
	auto lane = [](const ising_ctx *c) { return c->copy_inline ? c->stream : c->comm; };
	for (int k = 0; k < n; k++) {
		ising_ctx *c = ctxs[k];
		if (int rc = bind(c)) return rc;
		if (after_edges && spin && !c->copy_inline) HIP_TRY(hipStreamWaitEvent(c->comm, c->ev_edge[color], 0));
	}
	if (ctxs[0]->transport == ISING_TRANSPORT_IPC) {
		// one slab per process: push my first / last rows into the neighbours' mapped halo / ghost rows, then tell them (ising_ipc.cpp)
		if (n != 1) return fail(ISING_E_STATE, "the IPC transport serves one attached slab per process");
		const EdgeRows e = edge_rows(ctxs[0], color, depth);
		if (int rc = ising_ipc::push_rows(ctxs[0], color, depth, e.first, e.last, e.halo_top, e.halo_bot, e.bytes)) return rc;
	} else if (ctxs[0]->transport == ISING_TRANSPORT_RCCL) {
		RcclApi *api = rccl();
		if (!api) return fail(ISING_E_RCCL, "RCCL is not available: %s", rccl_state().error.c_str());
		RCCL_TRY(api->GroupStart());
		for (int k = 0; k < n; k++) {
			ising_ctx *c = ctxs[k];
			ncclComm_t comm = static_cast<ncclComm_t>(c->rccl_comm);
			const int nr = c->cfg.nslabs, me = c->cfg.slab, next = (me + 1) % nr, prev = (me + nr - 1) % nr;
			const EdgeRows e = edge_rows(c, color, depth);
			if (int rc = bind(c)) { (void)api->GroupEnd(); return rc; }
			// With two ranks prev == next: the peer's first receive (its top halo row, "from prev") must match our LAST
			// row, so the last row is sent first; receives are posted in the same order.
			ncclResult_t r = api->Send(e.last, e.bytes, ncclUint8, next, comm, c->comm);
			if (r == ncclSuccess) r = api->Send(e.first, e.bytes, ncclUint8, prev, comm, c->comm);
			if (r == ncclSuccess) r = api->Recv(e.halo_top, e.bytes, ncclUint8, prev, comm, c->comm);
			if (r == ncclSuccess) r = api->Recv(e.halo_bot, e.bytes, ncclUint8, next, comm, c->comm);
			if (r != ncclSuccess) {
				(void)api->GroupEnd();
				return fail(ISING_E_RCCL, "ncclSend/ncclRecv failed: %s", api->GetErrorString(r));
			}
		}
		RCCL_TRY(api->GroupEnd());
	} else {
		for (int k = 0; k < n; k++) {
			ising_ctx *c = ctxs[k], *prev = c->ring_prev, *next = c->ring_next;
			if (!prev || !next) return fail(ISING_E_STATE, "slab %d is not part of a single-process ring", c->cfg.slab);
			const EdgeRows e = edge_rows(c, color, depth), ep = edge_rows(prev, color, depth), en = edge_rows(next, color, depth);
			if (int rc = bind(c)) return rc;
			// next slab's top halo <- my last row ; previous slab's bottom halo <- my first row
			HIP_TRY(hipMemcpyPeerAsync(en.halo_top, next->cfg.device, e.last, c->cfg.device, e.bytes, lane(c)));
			HIP_TRY(hipMemcpyPeerAsync(ep.halo_bot, prev->cfg.device, e.first, c->cfg.device, e.bytes, lane(c)));
		}
	}
	if (spin) {
		for (int k = 0; k < n; k++) {
			ising_ctx *c = ctxs[k];
			if (int rc = bind(c)) return rc;
			HIP_TRY(hipEventRecord(c->ev_sent[color], lane(c)));
		}
	}
	return ISING_OK;
}

int stage_edges(ising_ctx *c, int it, int color) {
	if (int rc = ising_host::halo_ready(c, 1 - color)) return rc; // the edge rows read the other colour's halo rows
	if (int rc = ising_update_edges(c, it, color)) return rc;
	if (c->copy_inline) return ISING_OK;
	if (int rc = bind(c)) return rc;
	HIP_TRY(hipEventRecord(c->ev_edge[color], c->stream));
	return ISING_OK;
}

// A ballot slab turns dense when its temperature has no integer thresholds (ising_capi.cpp: launch_ranges).  In a ring
// that must happen on every slab before the first edge row of the sweep is sent, or neighbours would exchange rows in
// two bit orders.
int settle_layout(ising_ctx **ctxs, int n) {
	bool leave = false;
	for (int k = 0; k < n; k++) leave = leave || (ctxs[k]->ballot && ising_host::needs_generic(ctxs[k]));
	if (!leave) return ISING_OK;
	for (int k = 0; k < n; k++) {
		ising_ctx *c = ctxs[k];
		for (int color = 0; color < 2; color++) if (int rc = ising_host::halo_ready(c, color)) return rc;
		if (int rc = ising_host::ballot_leave(c)) return rc;
	}
	for (int k = 0; k < n; k++) if (int rc = ising_synchronize(ctxs[k])) return rc;
	return ISING_OK;
}

// Slabs with a comm stream of their own run the edge rows THERE, next to the interior rows on the compute stream:
//
//   compute stream                                   comm stream
//   wait ev_edge[1-c]  (rows 0, Y-1 of the source)     wait ev_int[1-c]  (rows 1, Y-2 of the source; and the interior of
//   interior rows 1 .. Y-2 of colour c                                    1-c has finished reading rows 0, Y-1 of c)
//   record ev_int[c]                                   wait: halo rows of 1-c have arrived
//                                                      edge rows 0, Y-1 of colour c (tiny launch) -> record ev_edge[c]
//                                                      deliver them (RCCL / peer copies)          -> record ev_sent[c]
//
// Both kernels start when the interior of the previous half-sweep ends and run side by side, so the ~20 us a two-row
// launch takes (its latency, not its work) is off the critical path: with the edge rows in front of the interior launch
// on one stream a 65536-row slab loses ~3 % per half-sweep.
int sweep_two_streams(ising_ctx **ctxs, int n, int first_it, int nsweeps) {
	for (int k = 0; k < n; k++) { // whatever the compute stream holds so far (initialisation, host writes, an earlier sweep)
		ising_ctx *c = ctxs[k];
		if (int rc = bind(c)) return rc;
		HIP_TRY(hipEventRecord(c->ev_int[ISING_WHITE], c->stream));
	}
	for (int it = first_it; it < first_it + nsweeps; it++) {
		for (int color = 0; color < 2; color++) {
			for (int k = 0; k < n; k++) {
				ising_ctx *c = ctxs[k];
				if (int rc = bind(c)) return rc;
				HIP_TRY(hipStreamWaitEvent(c->comm, c->ev_int[1 - color], 0));
				if (int rc = ising_host::halo_ready_on(c, 1 - color, c->comm)) return rc;
				if (int rc = ising_host::update_edges_on(c, it, color, c->comm, c->ev_edge[color])) return rc;
			}
			if (int rc = transfer(ctxs, n, color, false)) return rc;
			for (int k = 0; k < n; k++) {
				ising_ctx *c = ctxs[k];
				if (int rc = bind(c)) return rc;
				if (it > first_it || color == ISING_WHITE) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_edge[1 - color], 0));
				// (ev_int[c] rides on the launch itself: a record behind it would cost 7 us between two launches)
				if (int rc = ising_host::update_interior(c, it, color, c->ev_int[color])) return rc;
			}
		}
	}
	for (int k = 0; k < n; k++) { // later work on the compute stream (counts, reads, the next sweep) sees every row
		ising_ctx *c = ctxs[k];
		if (int rc = bind(c)) return rc;
		HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_edge[ISING_WHITE], 0));
	}
	return ISING_OK;
}

// Every slab's first / last `depth` rows of `color` to its neighbours' ghost rows, behind whatever the compute streams hold.
int exchange_rows(ising_ctx **ctxs, int n, int color, int depth) {
	for (int k = 0; k < n; k++) {
		ising_ctx *c = ctxs[k];
		if (int rc = bind(c)) return rc;
		if (c->transport == ISING_TRANSPORT_IPC && depth > 1) if (int rc = ising_ipc::release_ghosts(c, color)) return rc;
		HIP_TRY(hipEventRecord(c->ev_edge[color], c->stream));
	}
	if (ctxs[0]->transport == ISING_TRANSPORT_COPY) {
		// peer copies write into the neighbours' ghost rows: their launches, which read those rows, must be done too
		for (int k = 0; k < n; k++) {
			ising_ctx *c = ctxs[k], *prev = c->ring_prev, *next = c->ring_next;
			if (int rc = bind(c)) return rc;
			hipStream_t lane = c->copy_inline ? c->stream : c->comm;
			if (prev && prev != c) HIP_TRY(hipStreamWaitEvent(lane, prev->ev_edge[color], 0));
			if (next && next != c && next != prev) HIP_TRY(hipStreamWaitEvent(lane, next->ev_edge[color], 0));
			if (c->copy_inline) HIP_TRY(hipStreamWaitEvent(lane, c->ev_edge[color], 0));
		}
	}
	return transfer(ctxs, n, color, true, depth);
}

// Ballot slabs with G > 1 ghost rows (ising_ctx::ghost_rows): G rows of both colours travel every G colour half-sweeps, and
// ONE fused launch of G levels runs in between -- over the slab's rows and its ghost rows, whose draws are the ones their
// owners make (the generator is counter-based: global row, column, iteration), so both sides of a cut compute the same
// bits and what a level can no longer know of a ghost row never reaches a row it does.  The ring then costs one
// exchange and one launch boundary per G/2 sweeps instead of two launches, a send/recv pair and their events per colour;
// the bytes per sweep are the same.  The reference exchanges one row per colour half-sweep (optimized/main.cu:1779-1805).
int sweep_deep(ising_ctx **ctxs, int n, int first_it, int nsweeps) {
	const int G = ctxs[0]->ghost();
	bool current = true;
	for (int k = 0; k < n; k++) current = current && ctxs[k]->ghost_depth[0] >= G && ctxs[k]->ghost_depth[1] >= G;
	if (!current)
		for (int color = 0; color < 2; color++) if (int rc = exchange_rows(ctxs, n, color, G)) return rc;
	for (int it = first_it, left = nsweeps; left > 0;) {
		const int ns = std::min(left, G / 2);
		for (int k = 0; k < n; k++) {
			ising_ctx *c = ctxs[k];
			for (int color = 0; color < 2; color++) if (int rc = ising_host::halo_ready(c, color)) return rc;
			if (int rc = ising_host::update_deep(c, it, 2 * ns)) return rc;
		}
		for (int color = 0; color < 2; color++) if (int rc = exchange_rows(ctxs, n, color, G)) return rc;
		it += ns;
		left -= ns;
	}
	return ISING_OK;
}

// The comm stream of slab c, behind the transfers of the exchange that delivered the current ghost rows: wait until the
// neighbours' rows are in place as well (RCCL: the receives are this stream's own; IPC: the neighbours' counters; copies:
// their events), then move the slab's edge_go counter -- the edge strips of the next launch poll it (UpdateParams.edge_go).
int post_go(ising_ctx *c, int xs_slot = -1) { // xs_slot: the exchange-statistics slot this exchange is sampled into
	if (int rc = bind(c)) return rc;
	if (c->transport == ISING_TRANSPORT_IPC) {
		for (int color = 0; color < 2; color++) if (int rc = ising_ipc::wait_plane(c, color, c->comm)) return rc;
	} else if (c->transport == ISING_TRANSPORT_COPY) {
		ising_ctx *prev = c->ring_prev, *next = c->ring_next;
		for (int color = 0; color < 2; color++) {
			if (prev && prev != c) HIP_TRY(hipStreamWaitEvent(c->comm, prev->ev_sent[color], 0));
			if (next && next != c && next != prev) HIP_TRY(hipStreamWaitEvent(c->comm, next->ev_sent[color], 0));
		}
	}
	c->edge_go_epoch++;
	if (int rc = ising_ipc::counter_set_on(c->comm, c->d_edge + 16, c->edge_go_epoch)) return rc;
	HIP_TRY(hipEventRecord(c->ev_go, c->comm));
	if (xs_slot >= 0) HIP_TRY(hipEventRecord(c->xs_ev[4 * xs_slot + 3], c->comm));
	c->go_set = true;
	return ISING_OK;
}

// sweep_deep with the exchange taken off the compute streams (north_star: "overlapped with interior updates on a second HIP
// stream"; the reference synchronises every device after every colour, optimized/main.cu:1779-1805).  The fused launches
// follow each other on the compute stream with nothing in between; the comm stream learns from a counter when the edge
// strips of a launch have finished their LAST level (they are each level's first tickets, so that is a level before the
// launch ends), exchanges while the launch works on the interior, and moves a second counter that the next launch's edge
// strips poll at level 0 -- normally long since set.
//
//   compute stream                                   comm stream
//   launch e   (edge strips, last level: +1 each)      wait: edge_done == all edge strips of launch e
//   [wait ev_go]                                       [IPC: tell the neighbours their pushes may come; wait for their word]
//   launch e+1 (edge strips, level 0: poll edge_go)    first / last G rows of both colours -> neighbours   (record ev_sent)
//   ...                                                wait: the neighbours' rows are here; edge_go = e + 1; record ev_go
//
// The wait for ev_go in front of the next launch is what keeps this free of deadlock whatever the transport needs: a
// persistent launch fills the chip, and a transport kernel that cannot be placed next to it (RCCL's send/recv kernel wants
// more registers than five resident workgroups per CU leave: under rocprofv3 it runs when the launch's workgroups begin to
// retire) would never finish if the NEXT launch took the freed slots while its edge strips wait for that very kernel -- which
// is what happened with a ring of one over RCCL at 65536 x 8192 after ~3000 sweeps (tools/soak_overlap.py).  With the wait the
// exchange still runs in the tail of the launch whose rows it carries, and the gap between two launches is an event, not an
// exchange.  ISING_RING_OVERLAP=2 drops the wait (copies and the IPC transport: their kernels are one wave each).
// Exchange epochs one launch of this slab may carry (round 6).  A launch boundary per exchange costs a ring slab its ramp, its tail and the gap to the next launch
// every G/2 sweeps (65536 x 8192: ~1 % of a 5 ms launch) although nothing forces the launch to end: the exchange only concerns the edge units, which can wait for
// it INSIDE a running launch (UpdateParams.epoch_sh) as they already do at a launch's first level.  That needs a transport whose kernels find room next to a
// chip-filling persistent grid -- the peer (IPC) transport's one-lane kernels and copies do, RCCL's send/recv kernel does not -- and a device that no
// other rank's persistent grid shares (two of them waiting for each other's exchange inside launches that never end would stop both: the next launch's turn is
// what lets a co-resident rank in).  Ghost rows a power of two deep; either form of launch (the split form's word-only tail is per launch: it is the epochs that let ring slabs afford it).
int epochs_per_launch(const ising_ctx *c, int n) {
	const int G = c->ghost();
	if (n != 1 || c->transport != ISING_TRANSPORT_IPC || ising_ipc::sharing(c) != 1 || c->pol.ring_epochs == 1 || (G & (G - 1)) != 0 || G < 4 ||
	    c->pol.overlap == 0) return 1;
	if (c->pol.ring_epochs > 1) return c->pol.ring_epochs;
	const double epoch_ms = (double)c->cfg.X * ((double)c->cfg.Y + G) * (G / 2) / 3.4e9; // (3.4 flips/ns)
	return (int)std::max(1.0, std::min(64.0, 50.0 / std::max(epoch_ms, 1e-3)));
}

int sweep_deep_overlapped(ising_ctx **ctxs, int n, int first_it, int nsweeps) {
	const int G = ctxs[0]->ghost();
	bool current = true;
	for (int k = 0; k < n; k++) current = current && ctxs[k]->ghost_depth[0] >= G && ctxs[k]->ghost_depth[1] >= G;
	if (!current)
		for (int color = 0; color < 2; color++) if (int rc = exchange_rows(ctxs, n, color, G)) return rc;
	for (int k = 0; k < n; k++) if (!ctxs[k]->go_set) if (int rc = post_go(ctxs[k])) return rc;
	const bool copies = ctxs[0]->transport == ISING_TRANSPORT_COPY, ipc = ctxs[0]->transport == ISING_TRANSPORT_IPC;
	const int E = epochs_per_launch(ctxs[0], n);
	std::vector<int> xs(n, -1);
	for (int it = first_it, left = nsweeps; left > 0;) {
		const int ns = std::min(left, E * (G / 2));        // sweeps of this launch ...
		const int nep = (2 * ns + G - 1) / G;               // ... in that many exchange epochs (the last may be shorter)
		for (int k = 0; k < n; k++) {
			ising_ctx *c = ctxs[k];
			if (c->pol.overlap != 2 || c->transport == ISING_TRANSPORT_RCCL) {
				if (int rc = bind(c)) return rc;
				HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_go, 0));
			}
			// exchange statistics: this launch and the exchange that follows it (several epochs: its last) go into slot xs[k] (events on the dispatch packet, no extra packets)
			xs[k] = (c->xs_on && c->xs_n < c->xs_cap) ? c->xs_n++ : -1;
			if (xs[k] >= 0) { c->launch_start_next = c->xs_ev[4 * xs[k]]; c->launch_stop_next = c->xs_ev[4 * xs[k] + 1]; c->xs_epochs += nep; }
			if (c->ring_cnt_every > 0) { // print points inside this launch (ring_sweep_counted below): its sweeps whose iteration is a multiple of `every`
				const int every = c->ring_cnt_every, first = (every - it % every) % every;
				const int m = first < ns ? (ns - 1 - first) / every + 1 : 0;
				c->cnt_first_next = first;
				c->cnt_every_next = m > 0 ? every : 0;
				c->cnt_slot0_next = c->ring_cnt_inflight;
				c->cnt_bonds_next = c->ring_cnt_bonds;
				c->ring_cnt_inflight += m;
			}
			// (the split form where ising_create's rule gives the slab one: launches of several epochs, no print points inside -- their slots are laid out by the fused strips --;
			// which of the two forms is ahead depends on the box, +1.4 .. -0.9 % at 65536 x 8192: the slab's first full launches are timed on their dispatch packets, a warm
			// and a timed one of each form, and the faster stays -- the lone slabs' guard, ising_update.cpp)
			bool as_split = c->split && !c->split_always && E > 1 && nep > 1 && c->ring_cnt_every == 0 && !c->guard.no_split;
			if (as_split && c->pol.guard != 0 && c->guard.form < 3 && nep >= std::min(E, 4) && xs[k] < 0) { // (launches of four epochs and more: rates that compare)
				ising_ctx::ShapeGuard &g = c->guard;
				if (!g.e0) { HIP_TRY(hipEventCreate(&g.e0)); HIP_TRY(hipEventCreate(&g.e1)); }
				if (g.pending) { // the timed launch of the step before: wait for it (twice per slab, once in its life)
					float ms = 0;
					g.pending = false;
					if (hipEventSynchronize(g.e1) != hipSuccess || hipEventElapsedTime(&ms, g.e0, g.e1) != hipSuccess || ms <= 0) { (void)hipGetLastError(); g.form = 3; }
					else if (g.form_pending == 1) { g.split_rate = (float)(g.pending_flips / ((double)ms * 1.0e6)); g.form = 1; g.form_warm = false; }
					else { g.fused_rate = (float)(g.pending_flips / ((double)ms * 1.0e6)); g.form = 3; g.no_split = g.fused_rate > 1.02f * g.split_rate; }
					g.form_pending = 0;
					g.launches++;
				}
				if (g.form < 3) {
					as_split = g.form == 0; // (form 0: the split launches; 1: the fused ones)
					if (!g.form_warm) g.form_warm = true; // (this one runs untimed: the form's first launch of the series)
					else {
						c->launch_start_next = g.e0;
						c->launch_stop_next = g.e1;
						g.pending = true;
						g.pending_flips = (double)c->cfg.X * c->cfg.Y * ns;
						g.form_pending = as_split ? 1 : 2;
					}
				} else as_split = !g.no_split;
			}
			if (int rc = ising_host::update_deep(c, it, 2 * ns, true, nep, as_split)) return rc;
		}
		for (int e = 0; e < nep; e++) { // one exchange per epoch, behind the epoch's edge units, next to the launch
			const bool last = e == nep - 1;
			for (int k = 0; k < n; k++) {
				ising_ctx *c = ctxs[k];
				if (int rc = bind(c)) return rc;
				// (a launch that gave up never brings the counter there: the kernel looks at the slab's abort word)
				c->edge_done_target += c->edge_units_per_epoch;
				if (int rc = ising_ipc::counter_wait_on(c->comm, c->d_edge, c->edge_done_target, c->h_abort)) return rc;
				if (last && xs[k] >= 0) HIP_TRY(hipEventRecord(c->xs_ev[4 * xs[k] + 2], c->comm));
				// this slab's launches are done with its ghost rows: the neighbours may overwrite them
				if (ipc) for (int color = 0; color < 2; color++) if (int rc = ising_ipc::release_ghosts(c, color, c->comm)) return rc;
				if (copies) HIP_TRY(hipEventRecord(c->ev_int[0], c->comm));
			}
			if (copies) {
				for (int k = 0; k < n; k++) {
					ising_ctx *c = ctxs[k], *prev = c->ring_prev, *next = c->ring_next;
					if (int rc = bind(c)) return rc;
					if (prev && prev != c) HIP_TRY(hipStreamWaitEvent(c->comm, prev->ev_int[0], 0));
					if (next && next != c && next != prev) HIP_TRY(hipStreamWaitEvent(c->comm, next->ev_int[0], 0));
				}
			}
			for (int color = 0; color < 2; color++) if (int rc = transfer(ctxs, n, color, false, G)) return rc;
			for (int k = 0; k < n; k++) if (int rc = post_go(ctxs[k], last ? xs[k] : -1)) return rc;
		}
		it += ns;
		left -= ns;
	}
	return ISING_OK;
}

int sweep_local(ising_ctx **ctxs, int n, int first_it, int nsweeps) {
	if (int rc = settle_layout(ctxs, n)) return rc;
	if (ctxs[0]->cfg.XSL) { // sub-lattices never reach across slabs (optimized/main.cu:1423-1462): every slab sweeps on its own
		for (int k = 0; k < n; k++) if (int rc = ising_host::sweep_alone(ctxs[k], first_it, nsweeps)) return rc;
		return ISING_OK;
	}
	{
		bool deep = nsweeps > 0 && !ctxs[0]->store_ring && !ctxs[0]->cfg.XSL && ctxs[0]->ghost() > 1;
		for (int k = 0; k < n; k++) deep = deep && ctxs[k]->ballot && ctxs[k]->ghost() == ctxs[0]->ghost() && !ising_host::needs_generic(ctxs[k]) && !ctxs[k]->store_ring;
		// the exchange overlaps with the launches where the rows travel on the slabs' comm streams (ISING_RING_OVERLAP=0: between launches)
		bool overlap = deep && ctxs[0]->pol.overlap != 0;
		for (int k = 0; k < n; k++) overlap = overlap && !ctxs[k]->copy_inline && ctxs[k]->d_edge && ctxs[k]->comm;
		if (overlap) return sweep_deep_overlapped(ctxs, n, first_it, nsweeps);
		if (deep) return sweep_deep(ctxs, n, first_it, nsweeps);
	}
	if (ctxs[0]->store_ring && !ctxs[0]->cfg.XSL) { // one launch per slab and colour; the stream orders the rest
		for (int it = first_it; it < first_it + nsweeps; it++)
			for (int color = 0; color < 2; color++)
				for (int k = 0; k < n; k++) if (int rc = ising_update_color(ctxs[k], it, color, 0, ctxs[k]->cfg.Y)) return rc;
		return ISING_OK;
	}
	const bool two = nsweeps > 0 && !ctxs[0]->copy_inline && !ctxs[0]->cfg.XSL;
	if (two) return sweep_two_streams(ctxs, n, first_it, nsweeps);
	for (int it = first_it; it < first_it + nsweeps; it++) {
		for (int color = 0; color < 2; color++) {
			for (int k = 0; k < n; k++) if (int rc = stage_edges(ctxs[k], it, color)) return rc;
			if (int rc = transfer(ctxs, n, color, true)) return rc;
			for (int k = 0; k < n; k++) if (int rc = ising_update_color(ctxs[k], it, color, 1, ctxs[k]->cfg.Y - 1)) return rc;
		}
	}
	return ISING_OK;
}

// ising_ring_sweep / ising_rank_sweep with the reference's print points taken INSIDE the deep launches (as ising_sweep_counted does for a lone slab): where the ring
// sweeps its slabs through their ghost rows with the exchange in the launches' tails, the launches of every slab count the up spins of their OWN rows after every
// iteration that is a multiple of `every`; mine[k] = this process's slabs' sum for the k-th of them.  ISING_E_UNSUPPORTED-like answer (1 in *fallback) anywhere else.
// `bonds`: the launches' white levels also leave the equal bonds of their rows (every bond of the lattice has one white end in exactly one slab); mine[2 k] = up spins,
// mine[2 k + 1] = equal bonds (0 without `bonds`) of the k-th print point.
int ring_sweep_counted(ising_ctx **ctxs, int n, int first_it, int nsweeps, int every, bool bonds, std::vector<unsigned long long> &mine, bool *fallback) {
	*fallback = true;
	if (int rc = settle_layout(ctxs, n)) return rc;
	bool deep = !ctxs[0]->cfg.XSL && !ctxs[0]->cfg.use_J && ctxs[0]->ghost() > 1 && ctxs[0]->pol.overlap != 0; // (a call of no sweeps reserves the slots: a warm-up)
	for (int k = 0; k < n; k++)
		deep = deep && ctxs[k]->ballot && ctxs[k]->ghost() == ctxs[0]->ghost() && !ising_host::needs_generic(ctxs[k]) && !ctxs[k]->store_ring && !ctxs[k]->copy_inline &&
		       ctxs[k]->d_edge && ctxs[k]->comm;
	if (ctxs[0]->pol.ring_counted == 0) return ISING_OK;
	if (!deep) return ctxs[0]->pol.ring_counted == 2 ? fail(ISING_E_STATE, "ISING_RING_COUNTED=2: this ring does not sweep through ghost rows with overlapped exchanges") : ISING_OK;
	*fallback = false;
	const int G = ctxs[0]->ghost();
	mine.clear();
	std::vector<size_t> slots(n), n_up(n), chunk(n);
	std::vector<unsigned long long *> d_sum(n);
	size_t per_chunk = 64;
	for (int k = 0; k < n; k++) {
		ising_ctx *c = ctxs[k];
		if (int rc = bind(c)) return rc;
		const size_t strips = ((size_t)c->cfg.Y + 2 * (size_t)G - 2 + (size_t)c->H - 1) / (size_t)c->H; // rows [-(G - 1), Y + G - 1) of a deep launch
		if (int rc = ising_host::cnt_reserve(c, strips, bonds, &slots[k], &n_up[k], &chunk[k], &d_sum[k])) return rc;
		per_chunk = std::min(per_chunk, chunk[k]);
	}
	std::vector<unsigned long long> h(2 * per_chunk);
	int it = first_it, left = nsweeps;
	while (left > 0) {
		// as many sweeps as hold `per_chunk` measurements at most
		int ns = 0, m = 0;
		while (ns < left) {
			if ((it + ns) % every == 0) { if (m == (int)per_chunk) break; m++; }
			ns++;
		}
		for (int k = 0; k < n; k++) {
			ising_ctx *c = ctxs[k];
			if (int rc = bind(c)) return rc;
			if (m) HIP_TRY(hipMemsetAsync(c->d_cnt, 0, (size_t)m * slots[k] * sizeof(uint32_t), c->stream));
			c->ring_cnt_every = every;
			c->ring_cnt_bonds = bonds;
			c->ring_cnt_inflight = 0;
		}
		const int rc = sweep_deep_overlapped(ctxs, n, it, ns);
		for (int k = 0; k < n; k++) { ctxs[k]->ring_cnt_every = 0; ctxs[k]->ring_cnt_bonds = false; }
		if (rc) return rc;
		const size_t base = mine.size();
		mine.resize(base + 2 * (size_t)m, 0ull);
		for (int k = 0; k < n && m; k++) {
			ising_ctx *c = ctxs[k];
			if (int rc2 = bind(c)) return rc2;
			if (c->ring_cnt_inflight != m) return fail(ISING_E_STATE, "counted ring sweep: %d measurements launched, %d expected", c->ring_cnt_inflight, m);
			HIP_TRY(ising::launch_count_fold(c->d_cnt, slots[k], n_up[k], m, d_sum[k], c->stream));
			HIP_TRY(hipMemcpyAsync(h.data(), d_sum[k], (size_t)m * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
			if (int rc2 = ising_host::sync_checked(c)) return rc2;
			for (int q = 0; q < 2 * m; q++) mine[base + (size_t)q] += h[(size_t)q];
		}
		it += ns;
		left -= ns;
	}
	return ISING_OK;
}

int sync_both(ising_ctx *c) {
	if (int rc = bind(c)) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (c->comm) HIP_TRY(hipStreamSynchronize(c->comm));
	return ising_host::check_abort(c);
}

// -J: black couplings everywhere, their edge rows to the neighbours, then the white couplings (which gather from them)
int couplings_local(ising_ctx **ctxs, int n) {
	for (int k = 0; k < n; k++) if (int rc = ising_init_couplings_black(ctxs[k])) return rc;
	if (!ctxs[0]->wrap && !ctxs[0]->cfg.XSL) {
		for (int k = 0; k < n; k++) if (int rc = sync_both(ctxs[k])) return rc;
		if (int rc = transfer(ctxs, n, ISING_HAM_BLACK, false)) return rc;
		for (int k = 0; k < n; k++) if (int rc = sync_both(ctxs[k])) return rc;
		// (IPC: the synchronise above covers this rank's copies; the neighbours' arrive behind their counters)
		if (ctxs[0]->transport == ISING_TRANSPORT_IPC) if (int rc = ising_ipc::wait_plane(ctxs[0], ISING_HAM_BLACK, ctxs[0]->stream)) return rc;
	}
	for (int k = 0; k < n; k++) if (int rc = ising_init_couplings_white(ctxs[k])) return rc;
	return ISING_OK;
}

int rank_check(ising_ctx *c) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (!c->rank_mode || !(c->rccl_comm || ising_ipc::attached(c))) return fail(ISING_E_STATE, "the slab is not attached to a multi-process ring (ising_rank_attach / ising_ipc_attach)");
	return ISING_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------ library-internal
int ising_host::halo_ready(ising_ctx *c, int color) { return halo_ready_on(c, color, c->stream); }

int ising_host::halo_ready_on(ising_ctx *c, int color, hipStream_t s) {
	if (c->wrap || c->cfg.XSL || !c->transport) return ISING_OK;
	if (int rc = bind(c)) return rc;
	if (c->transport == ISING_TRANSPORT_RCCL) {
		// the receives are part of this slab's own group on its comm stream
		if (c->ev_sent[color] && s != c->comm) HIP_TRY(hipStreamWaitEvent(s, c->ev_sent[color], 0));
		return ISING_OK;
	}
	if (c->transport == ISING_TRANSPORT_IPC) {
		// this rank's own copies have read its rows (they may be overwritten), and both neighbours' rows have arrived
		if (c->ev_sent[color] && s != c->comm) HIP_TRY(hipStreamWaitEvent(s, c->ev_sent[color], 0));
		return ising_ipc::wait_plane(c, color, s);
	}
	// (a neighbour that delivers on the very stream this slab computes on is ordered by the stream itself)
	ising_ctx *prev = c->ring_prev, *next = c->ring_next;
	auto same_lane = [&](const ising_ctx *o) { return o->copy_inline && o->stream == s && o->cfg.device == c->cfg.device; };
	if (prev && prev->ev_sent[color] && !same_lane(prev)) HIP_TRY(hipStreamWaitEvent(s, prev->ev_sent[color], 0));
	if (next && next != prev && next->ev_sent[color] && !same_lane(next)) HIP_TRY(hipStreamWaitEvent(s, next->ev_sent[color], 0));
	// this slab's own copies (its rows to the neighbours' halo rows, on its comm stream) must be done before stream `s`
	// overwrites the rows they read -- a fused launch over the whole slab (sweep_deep), a layout change -- unless they
	// travel on `s` itself
	if (c != prev && c != next && c->ev_sent[color] && !same_lane(c) && !(s == c->comm)) HIP_TRY(hipStreamWaitEvent(s, c->ev_sent[color], 0));
	return ISING_OK;
}

void ising_host::ring_abort_drain(ising_ctx *c) {
	if (!c->comm) return;
	ising_ipc::set_abort(c, true); // (the polling kernels of the IPC transport look at a word of their own)
	(void)hipStreamSynchronize(c->comm);
	ising_ipc::set_abort(c, false);
}

static void xs_release(ising_ctx *c) {
	for (int i = 0; c->xs_ev && i < 4 * c->xs_cap; i++) if (c->xs_ev[i]) (void)hipEventDestroy(c->xs_ev[i]);
	delete[] c->xs_ev;
	c->xs_ev = nullptr;
	c->xs_cap = c->xs_n = c->xs_epochs = 0;
	c->xs_on = false;
}

void ising_host::ring_release(ising_ctx *c) {
	// neighbours of a single-process ring must not keep pointing at a slab that is going away
	if (c->ring_prev && c->ring_prev->ring_next == c) { c->ring_prev->ring_next = nullptr; c->ring_prev->store_ring = false; }
	if (c->ring_next && c->ring_next->ring_prev == c) { c->ring_next->ring_prev = nullptr; c->ring_next->store_ring = false; }
	c->ring_prev = c->ring_next = nullptr;
	(void)hipSetDevice(c->cfg.device);
	rccl_drop(c, false);
	ising_ipc::destroy(c);
	if (c->comm) (void)hipStreamDestroy(c->comm);
	for (int k = 0; k < 2; k++) {
		if (c->ev_edge[k]) (void)hipEventDestroy(c->ev_edge[k]);
		if (c->ev_sent[k]) (void)hipEventDestroy(c->ev_sent[k]);
		if (c->ev_int[k]) (void)hipEventDestroy(c->ev_int[k]);
	}
	if (c->ev_go) (void)hipEventDestroy(c->ev_go);
	c->ev_go = nullptr;
	c->comm = nullptr;
	xs_release(c);
}

extern "C" {

// ------------------------------------------------------------------------------------------------ exchange statistics
int ising_exchange_stats_begin(ising_ctx *c, int max_exchanges) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (max_exchanges < 1 || max_exchanges > 4096) return fail(ISING_E_ARG, "max_exchanges %d (1 .. 4096)", max_exchanges);
	if (int rc = bind(c)) return rc;
	if (c->xs_ev) { // (events of an earlier sampling may still be in flight)
		HIP_TRY(hipStreamSynchronize(c->stream));
		if (c->comm) HIP_TRY(hipStreamSynchronize(c->comm));
	}
	xs_release(c);
	c->xs_ev = new hipEvent_t[4 * (size_t)max_exchanges]();
	c->xs_cap = max_exchanges;
	for (int i = 0; i < 4 * max_exchanges; i++) HIP_TRY(hipEventCreate(&c->xs_ev[i])); // (with timing)
	c->xs_on = true;
	return ISING_OK;
}

int ising_exchange_stats_fetch(ising_ctx *c, ising_exchange_stats *out) {
	if (!c || !out) return fail(ISING_E_ARG, "null argument");
	memset(out, 0, sizeof(*out));
	if (!c->xs_ev) return fail(ISING_E_STATE, "ising_exchange_stats_begin first");
	if (int rc = bind(c)) return rc;
	c->xs_on = false;
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (c->comm) HIP_TRY(hipStreamSynchronize(c->comm));
	if (int rc = ising_host::check_abort(c)) return rc;
	// b - a in milliseconds, either sign (hipEventElapsedTime wants them in order on some runtimes)
	auto span = [](hipEvent_t a, hipEvent_t b, float *ms) -> hipError_t {
		hipError_t e = hipEventElapsedTime(ms, a, b);
		if (e != hipSuccess) {
			(void)hipGetLastError();
			e = hipEventElapsedTime(ms, b, a);
			*ms = -*ms;
		}
		return e;
	};
	struct Acc { double sum = 0; float max = 0; int n = 0; void add(float v) { sum += v; max = n ? std::max(max, v) : v; n++; } };
	Acc launch, xchg, late, gap;
	for (int e = 0; e < c->xs_n; e++) {
		hipEvent_t *ev = c->xs_ev + 4 * e;
		float ms = 0;
		HIP_TRY(span(ev[0], ev[1], &ms)); launch.add(ms);
		HIP_TRY(span(ev[2], ev[3], &ms)); xchg.add(ms);
		HIP_TRY(span(ev[1], ev[3], &ms)); late.add(ms);
		if (e + 1 < c->xs_n) { HIP_TRY(span(ev[1], ev[4], &ms)); gap.add(ms); }
	}
	out->exchanges = c->xs_epochs;
	out->launches = c->xs_n;
	auto put = [](const Acc &a, float *mean, float *max) { *mean = a.n ? (float)(a.sum / a.n) : 0.f; *max = a.max; };
	put(launch, &out->launch_ms_mean, &out->launch_ms_max);
	put(xchg, &out->exchange_ms_mean, &out->exchange_ms_max);
	put(late, &out->go_after_end_ms_mean, &out->go_after_end_ms_max);
	put(gap, &out->gap_ms_mean, &out->gap_ms_max);
	c->xs_n = c->xs_epochs = 0;
	return ISING_OK;
}

// ------------------------------------------------------------------------------------------------ single-process ring
int ising_ring_set_transport(ising_ctx **ctxs, int n, int transport) {
	if (transport != ISING_TRANSPORT_AUTO && transport != ISING_TRANSPORT_COPY && transport != ISING_TRANSPORT_RCCL)
		return fail(ISING_E_ARG, "bad transport %d (a single-process ring runs on copies or RCCL)", transport);
	if (int rc = ring_check(ctxs, n)) return rc;
	for (int k = 0; k < n; k++) if (int rc = sync_both(ctxs[k])) return rc;
	for (int k = 0; k < n; k++) ctxs[k]->transport = 0;
	return ring_bind(ctxs, n, transport);
}

int ising_ring_transport(ising_ctx **ctxs, int n, int *transport) {
	if (!transport) return fail(ISING_E_ARG, "null argument");
	if (int rc = ring_bind(ctxs, n)) return rc;
	*transport = ctxs[0]->transport;
	return ISING_OK;
}

int ising_ring_exchange(ising_ctx **ctxs, int n, int color) {
	if (color != ISING_BLACK && color != ISING_WHITE && color != ISING_HAM_BLACK) return fail(ISING_E_ARG, "bad colour %d", color);
	if (int rc = ring_bind(ctxs, n)) return rc;
	if (ctxs[0]->wrap) return ISING_OK;
	// whatever the slabs' streams hold (initialisation, host writes) must be complete before the rows travel
	for (int k = 0; k < n; k++) {
		ising_ctx *c = ctxs[k];
		if (int rc = bind(c)) return rc;
		if (color != ISING_HAM_BLACK) HIP_TRY(hipEventRecord(c->ev_edge[color], c->stream));
		else HIP_TRY(hipStreamSynchronize(c->stream));
	}
	int depth = color == ISING_HAM_BLACK ? 1 : ctxs[0]->ghost(); // spin colours: as deep as the slabs' ghost rows go
	for (int k = 0; k < n; k++) if (ctxs[k]->ghost() != ctxs[0]->ghost()) depth = 1;
	return transfer(ctxs, n, color, true, depth);
}

int ising_ring_sweep(ising_ctx **ctxs, int n, int first_it, int nsweeps) {
	if (int rc = ring_bind(ctxs, n)) return rc;
	if (ctxs[0]->wrap) return ising_sweep(ctxs[0], first_it, nsweeps);
	return sweep_local(ctxs, n, first_it, nsweeps);
}

// ising_ring_sweep with the reference's print points (ising_sweep_counted for rings): ups[k] = the up spins of the WHOLE lattice after the k-th iteration of the call
// that is a multiple of `every`.  Counted inside the deep launches where the ring sweeps through ghost rows with overlapped exchanges; sweeps and counts in turn elsewhere.
int ising_ring_sweep_counted(ising_ctx **ctxs, int n, int first_it, int nsweeps, int every, uint64_t *ups, int64_t *bond_equal, int max_counts, int *ncounts) {
	if (!ups || !ncounts) return fail(ISING_E_ARG, "null argument");
	if (first_it < 0 || nsweeps < 0 || every < 1) return fail(ISING_E_ARG, "bad iteration range or count interval");
	if (int rc = ring_bind(ctxs, n)) return rc;
	if (ctxs[0]->wrap) return ising_sweep_counted(ctxs[0], first_it, nsweeps, every, ups, bond_equal, max_counts, ncounts);
	const long long last = (long long)first_it + nsweeps - 1;
	const long long want = nsweeps > 0 ? last / every - ((long long)first_it - 1) / every : 0;
	*ncounts = 0;
	if (want > max_counts) return fail(ISING_E_ARG, "%lld counts, room for %d", want, max_counts);
	std::vector<unsigned long long> mine;
	bool fallback = true;
	if (int rc = ring_sweep_counted(ctxs, n, first_it, nsweeps, every, bond_equal != nullptr, mine, &fallback)) return rc;
	if (!fallback) {
		for (size_t k = 0; k < mine.size() / 2; k++) {
			ups[k] = mine[2 * k];
			if (bond_equal) bond_equal[k] = (int64_t)mine[2 * k + 1];
		}
		*ncounts = (int)(mine.size() / 2);
		return ISING_OK;
	}
	int it = first_it, k = 0;
	while (it <= last) { // the reference's own order of events
		const long long next = std::min<long long>(last, ((long long)it + every - 1) / every * every);
		if (int rc = sweep_local(ctxs, n, it, (int)(next - it + 1))) return rc;
		it = (int)next + 1;
		if (next % every == 0) {
			uint64_t up = 0, dw = 0;
			if (int rc = ising_ring_count(ctxs, n, &up, &dw)) return rc;
			if (bond_equal) if (int rc = ising_ring_bond_equal(ctxs, n, &bond_equal[k])) return rc;
			ups[k++] = up;
		}
	}
	*ncounts = k;
	return ISING_OK;
}

int ising_ring_init_couplings(ising_ctx **ctxs, int n) {
	if (int rc = ring_bind(ctxs, n)) return rc;
	return couplings_local(ctxs, n);
}

int ising_ring_synchronize(ising_ctx **ctxs, int n) {
	if (int rc = ring_check(ctxs, n)) return rc;
	for (int k = 0; k < n; k++) if (int rc = sync_both(ctxs[k])) return rc;
	return ISING_OK;
}

int ising_ring_count(ising_ctx **ctxs, int n, uint64_t *up, uint64_t *down) {
	if (!up || !down) return fail(ISING_E_ARG, "null argument");
	if (int rc = ring_check(ctxs, n)) return rc;
	*up = *down = 0;
	for (int k = 0; k < n; k++) {
		uint64_t u = 0, d = 0;
		if (int rc = ising_count(ctxs[k], &u, &d)) return rc;
		*up += u;
		*down += d;
	}
	return ISING_OK;
}

int ising_ring_bond_equal(ising_ctx **ctxs, int n, int64_t *A) {
	if (!A) return fail(ISING_E_ARG, "null argument");
	if (int rc = ring_bind(ctxs, n)) return rc;
	*A = 0;
	for (int k = 0; k < n; k++) {
		int64_t a = 0;
		if (int rc = ising_bond_equal(ctxs[k], &a)) return rc; // waits for the white halo rows (halo_ready)
		*A += a;
	}
	return ISING_OK;
}

// ------------------------------------------------------------------------------------------------ one slab per process
int ising_rccl_available(int *version) {
	RcclApi *api = rccl();
	if (!api) return fail(ISING_E_RCCL, "RCCL is not available: %s", rccl_state().error.c_str());
	if (version) {
		int v = 0;
		RCCL_TRY(api->GetVersion(&v));
		*version = v;
	}
	return ISING_OK;
}

int ising_rccl_unique_id(void *id_out) {
	if (!id_out) return fail(ISING_E_ARG, "null argument");
	RcclApi *api = rccl();
	if (!api) return fail(ISING_E_RCCL, "RCCL is not available: %s", rccl_state().error.c_str());
	ncclUniqueId id;
	RCCL_TRY(api->GetUniqueId(&id));
	static_assert(sizeof(id) == ISING_RCCL_ID_BYTES, "ncclUniqueId size");
	memcpy(id_out, &id, sizeof(id));
	return ISING_OK;
}

int ising_rank_attach(ising_ctx *c, const void *id_in) {
	if (!c || !id_in) return fail(ISING_E_ARG, "null argument");
	if (c->wrap) return fail(ISING_E_STATE, "a slab that wraps in place has no halo rows to exchange (nslabs == 1 without ring_halo)");
	if (c->rank_mode || c->rccl_comm) return fail(ISING_E_STATE, "the slab is already attached");
	RcclApi *api = rccl();
	if (!api) return fail(ISING_E_RCCL, "RCCL is not available: %s", rccl_state().error.c_str());
	if (int rc = ising_host::ring_resources(c)) return rc;
	ncclUniqueId id;
	memcpy(&id, id_in, sizeof(id));
	ncclComm_t comm = nullptr;
	RCCL_TRY(api->CommInitRank(&comm, c->cfg.nslabs, id, c->cfg.slab)); // collective over all ranks of the ring
	c->rccl_comm = comm;
	c->rccl_owner = true;
	c->rank_mode = true;
	c->transport = ISING_TRANSPORT_RCCL;
	c->ring_prev = c->ring_next = nullptr;
	return ISING_OK;
}

int ising_rank_detach(ising_ctx *c, int abort_pending) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (!c->rank_mode) return ISING_OK;
	if (c->transport == ISING_TRANSPORT_IPC && c->ipc) {
		// abort_pending: this rank's polling kernels (waiting for a neighbour that will not answer) give up, the streams drain
		if (abort_pending) ising_ipc::set_abort(c, true);
		const int rc = sync_both(c);
		// (the segment goes too: its counters belong to this attachment; a later ising_ipc_export starts from zero)
		ising_ipc::destroy(c);
		c->rank_mode = false;
		c->transport = 0;
		c->ghost_depth[0] = c->ghost_depth[1] = 0;
		return rc;
	}
	if (!abort_pending) if (int rc = sync_both(c)) return rc;
	rccl_drop(c, abort_pending != 0);
	c->rank_mode = false;
	c->transport = 0;
	return ISING_OK;
}

int ising_rank_exchange(ising_ctx *c, int color) {
	if (int rc = rank_check(c)) return rc;
	if (color != ISING_BLACK && color != ISING_WHITE && color != ISING_HAM_BLACK) return fail(ISING_E_ARG, "bad colour %d", color);
	if (int rc = bind(c)) return rc;
	const int depth = color == ISING_HAM_BLACK ? 1 : c->ghost(); // (spin colours: as deep as the ghost rows go)
	if (c->transport == ISING_TRANSPORT_IPC && depth > 1) if (int rc = ising_ipc::release_ghosts(c, color)) return rc;
	if (color != ISING_HAM_BLACK) HIP_TRY(hipEventRecord(c->ev_edge[color], c->stream));
	else HIP_TRY(hipStreamSynchronize(c->stream));
	return transfer(&c, 1, color, true, depth);
}

int ising_rank_sweep(ising_ctx *c, int first_it, int nsweeps) {
	if (int rc = rank_check(c)) return rc;
	return sweep_local(&c, 1, first_it, nsweeps);
}

int ising_rank_init_couplings(ising_ctx *c) {
	if (int rc = rank_check(c)) return rc;
	return couplings_local(&c, 1);
}

int ising_rank_wait(ising_ctx *c, int timeout_ms) {
	if (int rc = rank_check(c)) return rc;
	if (int rc = bind(c)) return rc;
	if (timeout_ms < 0) return sync_both(c);
	const auto t0 = std::chrono::steady_clock::now();
	for (;;) {
		const hipError_t a = hipStreamQuery(c->stream), b = hipStreamQuery(c->comm);
		if (a == hipSuccess && b == hipSuccess) return ISING_OK;
		if ((a != hipSuccess && a != hipErrorNotReady) || (b != hipSuccess && b != hipErrorNotReady))
			return fail(ISING_E_HIP, "stream query failed: %s", hipGetErrorString(a != hipSuccess && a != hipErrorNotReady ? a : b));
		if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms)
			return fail(ISING_E_TIMEOUT, "the ring exchange did not complete within %d ms", timeout_ms);
		std::this_thread::sleep_for(std::chrono::microseconds(200));
	}
}

// totals over all ranks: ncclAllReduce of the slab's counter on the compute stream
static int rank_allreduce_u64(ising_ctx *c, unsigned long long *d_val, unsigned long long *host_out) {
	RcclApi *api = rccl();
	if (!api) return fail(ISING_E_RCCL, "RCCL is not available: %s", rccl_state().error.c_str());
	RCCL_TRY(api->AllReduce(d_val, c->d_acc + 2, 1, ncclUint64, ncclSum, static_cast<ncclComm_t>(c->rccl_comm), c->stream));
	HIP_TRY(hipMemcpyAsync(host_out, c->d_acc + 2, sizeof(*host_out), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return ISING_OK;
}

// ising_rank_sweep with print points: collective (every rank calls it with the same arguments); the ranks' sums travel over the rank transport.
int ising_rank_sweep_counted(ising_ctx *c, int first_it, int nsweeps, int every, uint64_t *ups, int64_t *bond_equal, int max_counts, int *ncounts) {
	if (int rc = rank_check(c)) return rc;
	if (!ups || !ncounts) return fail(ISING_E_ARG, "null argument");
	if (first_it < 0 || nsweeps < 0 || every < 1) return fail(ISING_E_ARG, "bad iteration range or count interval");
	const long long last = (long long)first_it + nsweeps - 1;
	const long long want = nsweeps > 0 ? last / every - ((long long)first_it - 1) / every : 0;
	*ncounts = 0;
	if (want > max_counts) return fail(ISING_E_ARG, "%lld counts, room for %d", want, max_counts);
	std::vector<unsigned long long> mine;
	bool fallback = true;
	// A local failure (an allocation, a launch, "measurements launched != expected") must not leave the other ranks in the reductions below:
	// the outcome travels with the launch form -- low 20 bits: ranks that fell back, above: ranks that failed -- and every rank returns together
	const int local_rc = ring_sweep_counted(&c, 1, first_it, nsweeps, every, bond_equal != nullptr, mine, &fallback);
	const std::string local_err = local_rc ? ising_last_error() : "";
	unsigned long long fb = 0;
	if (int rc = ising_host::rank_sum_u64(c, (fallback ? 1ull : 0ull) + (local_rc ? (1ull << 20) : 0ull), &fb)) return rc;
	if (fb >> 20) return local_rc ? fail(local_rc, "%s", local_err.c_str()) : fail(ISING_E_STATE, "counted rank sweep: %llu other rank(s) failed in their launches", fb >> 20);
	// (every rank takes the same branch: the conditions are the configuration's, which the ranks share -- the sum says so)
	if (fb != 0 && fb != (unsigned long long)c->cfg.nslabs) return fail(ISING_E_STATE, "counted rank sweep: the ranks disagree about the launch form");
	if (!fallback) {
		for (size_t k = 0; k < mine.size() / 2; k++) {
			unsigned long long tot = 0;
			if (int rc = ising_host::rank_sum_u64(c, mine[2 * k], &tot)) return rc;
			ups[k] = tot;
			if (bond_equal) {
				if (int rc = ising_host::rank_sum_u64(c, mine[2 * k + 1], &tot)) return rc;
				bond_equal[k] = (int64_t)tot;
			}
		}
		*ncounts = (int)(mine.size() / 2);
		return ISING_OK;
	}
	int it = first_it, k = 0;
	while (it <= last) {
		const long long next = std::min<long long>(last, ((long long)it + every - 1) / every * every);
		if (int rc = sweep_local(&c, 1, it, (int)(next - it + 1))) return rc;
		it = (int)next + 1;
		if (next % every == 0) {
			uint64_t up = 0, dw = 0;
			if (int rc = ising_rank_count(c, &up, &dw)) return rc;
			if (bond_equal) if (int rc = ising_rank_bond_equal(c, &bond_equal[k])) return rc;
			ups[k++] = up;
		}
	}
	*ncounts = k;
	return ISING_OK;
}

int ising_rank_count(ising_ctx *c, uint64_t *up, uint64_t *down) {
	if (int rc = rank_check(c)) return rc;
	if (!up || !down) return fail(ISING_E_ARG, "null argument");
	uint64_t u = 0, d = 0;
	if (int rc = ising_count(c, &u, &d)) return rc; // leaves the slab's up count in d_acc[0]
	unsigned long long tot = 0;
	if (c->transport == ISING_TRANSPORT_IPC) { if (int rc = ising_ipc::allreduce_u64(c, u, &tot)) return rc; }
	else if (int rc = rank_allreduce_u64(c, c->d_acc, &tot)) return rc;
	*up = tot;
	*down = (uint64_t)c->cfg.X * (uint64_t)c->cfg.Y * (uint64_t)c->cfg.nslabs - tot;
	return ISING_OK;
}

int ising_rank_bond_equal(ising_ctx *c, int64_t *A) {
	if (int rc = rank_check(c)) return rc;
	if (!A) return fail(ISING_E_ARG, "null argument");
	int64_t a = 0;
	if (int rc = ising_bond_equal(c, &a)) return rc; // leaves the slab's sum in d_acc[1]
	unsigned long long tot = 0;
	if (c->transport == ISING_TRANSPORT_IPC) { if (int rc = ising_ipc::allreduce_u64(c, (unsigned long long)a, &tot)) return rc; }
	else if (int rc = rank_allreduce_u64(c, c->d_acc + 1, &tot)) return rc;
	*A = (int64_t)tot;
	return ISING_OK;
}

// ------------------------------------------------------------------------------------------------ correlations
int ising_ring_correlations(ising_ctx **ctxs, int n, int ncorr, int64_t *sums) {
	if (int rc = ring_bind(ctxs, n)) return rc;
	if (!sums || ncorr < 1 || ncorr > 128) return fail(ISING_E_ARG, "ncorr must be in [1,128]");
	for (int k = 0; k < n; k++) {
		ising_ctx *c = ctxs[k];
		if (c->cfg.XSL && c->cfg.YSL < ncorr) return fail(ISING_E_ARG, "sub-lattices need at least %d rows for %d correlation distances", ncorr, ncorr);
		if (c->cfg.Y < ncorr) return fail(ISING_E_ARG, "each slab needs at least %d rows for %d correlation distances", ncorr, ncorr);
		if (int rc = bind(c)) return rc;
		if (!c->d_bits || c->d_bits_extra < ncorr) {
			if (c->d_bits) HIP_TRY(hipFree(c->d_bits));
			c->d_bits = nullptr;
			HIP_TRY(hipMalloc((void **)&c->d_bits, (size_t)(c->cfg.Y + 128) * c->lld_packed * sizeof(uint32_t)));
			c->d_bits_extra = 128;
		}
		if (!c->d_corr) HIP_TRY(hipMalloc((void **)&c->d_corr, 128 * sizeof(long long)));
		// bit matrix: X bits = lld_packed 32-bit words per row
		if (c->ballot) if (int rc = ising_host::ballot_image(c)) return rc;
		if (c->ballot) HIP_TRY(ising::launch_dense_pack_bits(c->tmp(ISING_BLACK), c->tmp(ISING_WHITE), c->gx * 32, c->cfg.Y,
		                                                     (uint32_t)c->cfg.slab * (uint32_t)c->cfg.Y, c->d_bits, c->stream));
		else if (c->dense) HIP_TRY(ising::launch_dense_pack_bits(c->lat(ISING_BLACK), c->lat(ISING_WHITE), c->gx * 32, c->cfg.Y,
		                                                    (uint32_t)c->cfg.slab * (uint32_t)c->cfg.Y, c->d_bits, c->stream));
		else HIP_TRY(ising::launch_pack_bits(c->lat(ISING_BLACK), c->lat(ISING_WHITE), c->lld_packed, c->cfg.Y,
		                                     (uint32_t)c->cfg.slab * (uint32_t)c->cfg.Y, c->d_bits, c->stream));
		HIP_TRY(hipMemsetAsync(c->d_corr, 0, 128 * sizeof(long long), c->stream));
	}
	for (int k = 0; k < n; k++) if (int rc = ising_synchronize(ctxs[k])) return rc;
	// rows that follow slab k (vertical partners of its last ncorr rows): the first ncorr bit-rows of slab k+1
	for (int k = 0; k < n; k++) {
		ising_ctx *c = ctxs[k], *next = ctxs[(k + 1) % n];
		if (int rc = bind(c)) return rc;
		if (!c->cfg.XSL) // sub-lattices never look past their own rows
			HIP_TRY(hipMemcpyPeerAsync(c->d_bits + (size_t)c->cfg.Y * c->lld_packed, c->cfg.device, next->d_bits, next->cfg.device,
			                           (size_t)ncorr * c->lld_packed * sizeof(uint32_t), c->stream));
		HIP_TRY(ising::launch_corr(c->d_bits, c->lld_packed, c->cfg.Y, ncorr, c->cfg.XSL ? c->cfg.XSL / 32 : c->lld_packed,
		                           c->cfg.XSL ? c->cfg.YSL : 0, c->d_corr, c->stream));
	}
	std::vector<long long> h(ncorr);
	for (int j = 0; j < ncorr; j++) sums[j] = 0;
	for (int k = 0; k < n; k++) {
		ising_ctx *c = ctxs[k];
		if (int rc = bind(c)) return rc;
		HIP_TRY(hipMemcpyAsync(h.data(), c->d_corr, (size_t)ncorr * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		for (int j = 0; j < ncorr; j++) sums[j] += h[j];
		ising_host::ballot_tmp_release(c); // (ballot layout: the dense-order image doubled the slab's memory while this ran)
	}
	return ISING_OK;
}

int ising_correlations(ising_ctx *c, int ncorr, int64_t *sums) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (c->cfg.nslabs != 1) return fail(ISING_E_STATE, "ising_correlations needs nslabs == 1; use ising_ring_correlations");
	return ising_ring_correlations(&c, 1, ncorr, sums);
}

} // extern "C"

int ising_host::rank_sum_u64(ising_ctx *c, unsigned long long mine, unsigned long long *sum) {
	if (int rc = rank_check(c)) return rc;
	if (c->transport == ISING_TRANSPORT_IPC) return ising_ipc::allreduce_u64(c, mine, sum);
	if (int rc = bind(c)) return rc;
	HIP_TRY(hipMemcpyAsync(c->d_acc + 3, &mine, sizeof(mine), hipMemcpyHostToDevice, c->stream));
	return rank_allreduce_u64(c, c->d_acc + 3, sum);
}
