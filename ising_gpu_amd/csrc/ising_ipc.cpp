// ising_ipc.cpp -- ISING_TRANSPORT_IPC: one process per slab without RCCL (include/ising_hip.h: ising_ipc_export /
// ising_ipc_attach), and the one-lane counter kernels the slab ring's schedules use.
#include "ising_ipc.hpp"

#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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

// One process per slab without RCCL (include/ising_hip.h: ising_ipc_export / ising_ipc_attach).  The reference reaches the
// rows outside a slab by direct peer access (optimized/main.cu:1496-1537, :1637-1642); across processes that is
// hipIpcMemHandle: a rank maps its two neighbours' spin (and coupling) arrays and PUSHES its first / last rows into their
// halo or ghost rows with device-to-device copies on its comm stream.  Streams and events do not cross processes, so the
// hand-over goes through monotone epoch counters in a 4 KiB POSIX shared-memory segment per rank, mapped by every rank and
// registered with HIP (hipHostRegister: host memory is coherent for every device and process): a one-lane kernel on the
// producer's stream writes the counter behind its copies, a one-lane kernel on the consumer's stream polls it.  Counters,
// not events: a waiter names the epoch it needs, so nobody depends on the order in which the processes' hosts happen to
// enqueue (a cross-process event wait sees whatever record was the latest when the WAIT was enqueued).
//   READY_TOP / READY_BOT (per plane)  in MY segment: epoch of the rows my previous / next neighbour has put into my top /
//                                      bottom halo or ghost rows
//   FREE_PREV / FREE_NEXT (per plane)  in MY segment: epoch up to which my previous / next neighbour's launches are done with
//                                      ITS ghost rows, i.e. I may overwrite them (deep exchange only: with one halo row per
//                                      colour half-sweep the neighbour's READY for the other colour already implies it)
constexpr uint32_t IPC_MAGIC = 0x50495349u; // "ISIP"
constexpr uint32_t IPC_VERSION = 1;
constexpr int IPC_FLAG_STRIDE = 16; // 32-bit words between two counters: a 64-byte line each
constexpr int IPC_PLANES = 3;       // black, white, black couplings
enum { F_READY_TOP = 0, F_READY_BOT = 1, F_FREE_PREV = 2, F_FREE_NEXT = 3, F_KINDS = 4 };
inline int ipc_flag(int plane, int kind) { return (plane * F_KINDS + kind) * IPC_FLAG_STRIDE; }

struct IpcSegment { // created zero-filled by its owner
	uint32_t magic, nslabs, slab, pad0[13];
	uint32_t flags[IPC_PLANES * F_KINDS * IPC_FLAG_STRIDE]; // polled and written by kernels
	uint32_t abort, pad1[15];                               // set by the owner's host: its polling kernels give up
	uint64_t red_seq, red_val[2], pad2[5];                  // host-side all-reduce of one 64-bit value (ising_rank_count)
};
static_assert(sizeof(IpcSegment) <= 4096, "one page");
constexpr size_t IPC_SEG_BYTES = 4096;

struct IpcBlob { // what a rank tells the others about itself; ISING_IPC_BLOB_BYTES
	uint32_t magic, version;
	int32_t nslabs, slab, X, Y, layout, ghost, use_J, pid;
	int32_t pci[3], pad;                   // domain, bus, device of the GPU that holds the slab (ranks sharing a device)
	uint64_t lat_ptr, lat_off, ham_ptr, ham_off; // the arrays in the owner's address space; offset inside the exported allocation
	hipIpcMemHandle_t lat_handle, ham_handle;
	char shm_name[40];
};
static_assert(sizeof(IpcBlob) <= ISING_IPC_BLOB_BYTES, "blob size");

} // namespace

struct ising_ipc_state {
	std::string shm_name;
	IpcSegment *mine = nullptr;
	uint32_t *mine_flags = nullptr;     // device pointer to mine->flags
	uint32_t *mine_abort = nullptr;     // ... to mine->abort
	bool attached = false;
	std::vector<IpcSegment *> seg;      // every rank's segment in this process (seg[slab] == mine)
	uint32_t *nb_flags[2] = {nullptr, nullptr}; // device pointers to the previous / next rank's flags
	char *nb_lat[2] = {nullptr, nullptr};       // their spin arrays (d_lat) and coupling arrays (d_ham) as mapped here
	char *nb_ham[2] = {nullptr, nullptr};
	std::vector<void *> opened;         // hipIpcOpenMemHandle bases to close
	std::vector<void *> registered;     // hipHostRegister'ed segments to unregister
	uint32_t epoch[IPC_PLANES] = {0, 0, 0};      // exchanges of a plane posted so far (every rank counts the same)
	uint32_t free_epoch[IPC_PLANES] = {0, 0, 0}; // deep exchanges of a plane announced so far
	uint32_t seen[IPC_PLANES] = {0, 0, 0};       // epoch the compute stream has already waited for
	uint64_t red_seq = 0;
	int sharing = 1;                    // ranks (this one included) whose slabs live on this rank's device
};

namespace {

// one lane polls two counters (system scope: they live in host memory) until both have reached their epochs
__global__ void __launch_bounds__(64) ipc_wait_k(const uint32_t *a, uint32_t need_a, const uint32_t *b, uint32_t need_b, const uint32_t *abort_flag) {
	if (threadIdx.x == 0) {
		for (unsigned n = 1;; ++n) {
			const uint32_t va = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			const uint32_t vb = __hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			if ((int32_t)(va - need_a) >= 0 && (int32_t)(vb - need_b) >= 0) break;
			if ((n & 15u) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
			__builtin_amdgcn_s_sleep(127);
		}
		__atomic_thread_fence(__ATOMIC_ACQUIRE); // system scope: the rows behind the counters
	}
}

// ... and one lane writes two counters behind everything the stream did before (the copies)
__global__ void __launch_bounds__(64) ipc_set_k(uint32_t *a, uint32_t va, uint32_t *b, uint32_t vb) {
	if (threadIdx.x == 0) {
		__atomic_thread_fence(__ATOMIC_RELEASE);
		__hip_atomic_store(a, va, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(b, vb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// one lane waits until a counter in device memory has reached `need` (written by a running launch of another stream)
__global__ void __launch_bounds__(64) counter_wait_k(const uint32_t *a, uint32_t need, const uint32_t *abort_flag) {
	if (threadIdx.x == 0) {
		for (unsigned n = 1;; ++n) {
			const uint32_t va = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if ((int32_t)(va - need) >= 0) break;
			if (abort_flag && (n & 63u) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
			__builtin_amdgcn_s_sleep(64);
		}
		__atomic_thread_fence(__ATOMIC_ACQUIRE);
	}
}

__global__ void __launch_bounds__(64) counter_set_k(uint32_t *a, uint32_t v) {
	if (threadIdx.x == 0) {
		__atomic_thread_fence(__ATOMIC_RELEASE);
		__hip_atomic_store(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

int ipc_wait_on(ising_ctx *c, hipStream_t s, const uint32_t *a, uint32_t na, const uint32_t *b, uint32_t nb) {
	hipLaunchKernelGGL(ipc_wait_k, dim3(1), dim3(64), 0, s, a, na, b, nb, (const uint32_t *)c->ipc->mine_abort);
	HIP_TRY(hipGetLastError());
	return ISING_OK;
}

int ipc_set_on(hipStream_t s, uint32_t *a, uint32_t va, uint32_t *b, uint32_t vb) {
	hipLaunchKernelGGL(ipc_set_k, dim3(1), dim3(64), 0, s, a, va, b, vb);
	HIP_TRY(hipGetLastError());
	return ISING_OK;
}

// stream `s` waits until both neighbours have delivered the current epoch of `plane` into this slab's halo / ghost rows
int ipc_wait_plane(ising_ctx *c, int plane, hipStream_t s) {
	ising_ipc_state *ipc = c->ipc;
	if (!ipc->epoch[plane]) return ISING_OK; // nothing was ever exchanged
	if (s == c->stream && ipc->seen[plane] == ipc->epoch[plane]) return ISING_OK;
	if (int rc = ipc_wait_on(c, s, ipc->mine_flags + ipc_flag(plane, F_READY_TOP), ipc->epoch[plane], ipc->mine_flags + ipc_flag(plane, F_READY_BOT), ipc->epoch[plane])) return rc;
	if (s == c->stream) ipc->seen[plane] = ipc->epoch[plane];
	return ISING_OK;
}

// Deep exchange: everything this slab's compute stream holds so far is done with the slab's ghost rows of `color` once the
// stream gets here -- the neighbours may overwrite them with the next epoch.
int ipc_release_ghosts(ising_ctx *c, int color, hipStream_t s = nullptr) {
	ising_ipc_state *ipc = c->ipc;
	const uint32_t fe = ++ipc->free_epoch[color];
	// in the PREVIOUS rank's segment I am its next neighbour, in the next rank's its previous one
	return ipc_set_on(s ? s : c->stream, ipc->nb_flags[0] + ipc_flag(color, F_FREE_NEXT), fe, ipc->nb_flags[1] + ipc_flag(color, F_FREE_PREV), fe);
}

void ipc_unmap(ising_ctx *c) {
	ising_ipc_state *ipc = c->ipc;
	if (!ipc) return;
	(void)hipSetDevice(c->cfg.device);
	for (void *b : ipc->opened) (void)hipIpcCloseMemHandle(b);
	ipc->opened.clear();
	for (void *h : ipc->registered) { if (h != ipc->mine) { (void)hipHostUnregister(h); } }
	for (IpcSegment *sg : ipc->seg) if (sg && sg != ipc->mine) (void)munmap(sg, IPC_SEG_BYTES);
	ipc->registered.clear();
	ipc->seg.clear();
	for (int k = 0; k < 2; k++) { ipc->nb_flags[k] = nullptr; ipc->nb_lat[k] = ipc->nb_ham[k] = nullptr; }
	ipc->attached = false;
}

void ipc_destroy(ising_ctx *c) {
	ising_ipc_state *ipc = c->ipc;
	if (!ipc) return;
	ipc_unmap(c);
	if (ipc->mine) {
		(void)hipHostUnregister(ipc->mine);
		(void)munmap(ipc->mine, IPC_SEG_BYTES);
		(void)shm_unlink(ipc->shm_name.c_str());
	}
	delete ipc;
	c->ipc = nullptr;
}

// sum of one 64-bit value over all ranks, by the hosts, through the segments (two slots: a rank can be at most one
// reduction ahead of the slowest -- it cannot finish reduction k + 1 before everybody has published k + 1, i.e. read k)
int ipc_allreduce_u64(ising_ctx *c, unsigned long long mine_val, unsigned long long *out, int timeout_ms = 120000) {
	ising_ipc_state *ipc = c->ipc;
	const uint64_t seq = ++ipc->red_seq;
	ipc->mine->red_val[seq & 1] = mine_val;
	__atomic_store_n(&ipc->mine->red_seq, seq, __ATOMIC_RELEASE);
	unsigned long long tot = 0;
	const auto t0 = std::chrono::steady_clock::now();
	for (size_t r = 0; r < ipc->seg.size(); r++) {
		IpcSegment *sg = ipc->seg[r];
		for (unsigned spin = 0; __atomic_load_n(&sg->red_seq, __ATOMIC_ACQUIRE) < seq; ++spin) {
			if ((spin & 1023u) == 1023u) {
				if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms)
					return fail(ISING_E_TIMEOUT, "rank %zu did not reach all-reduce %llu within %d ms", r, (unsigned long long)seq, timeout_ms);
				std::this_thread::sleep_for(std::chrono::microseconds(50));
			}
		}
		tot += sg->red_val[seq & 1];
	}
	*out = tot;
	return ISING_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------ what the ring uses
bool ising_ipc::attached(const ising_ctx *c) { return c->ipc && c->ipc->attached; }
int ising_ipc::sharing(const ising_ctx *c) { return (c->ipc && c->ipc->attached) ? c->ipc->sharing : 1; }
const uint32_t *ising_ipc::abort_word(const ising_ctx *c) { return c->ipc ? c->ipc->mine_abort : nullptr; }
void ising_ipc::set_abort(ising_ctx *c, bool on) { if (c->ipc && c->ipc->mine) __atomic_store_n(&c->ipc->mine->abort, on ? 1u : 0u, __ATOMIC_RELEASE); }
int ising_ipc::wait_plane(ising_ctx *c, int plane, hipStream_t s) { return ipc_wait_plane(c, plane, s); }
int ising_ipc::release_ghosts(ising_ctx *c, int color, hipStream_t s) { return ipc_release_ghosts(c, color, s); }
int ising_ipc::allreduce_u64(ising_ctx *c, unsigned long long v, unsigned long long *out, int timeout_ms) { return ipc_allreduce_u64(c, v, out, timeout_ms); }
void ising_ipc::destroy(ising_ctx *c) { ipc_destroy(c); }

int ising_ipc::counter_wait_on(hipStream_t s, const uint32_t *counter, uint32_t need, const uint32_t *abort_flag) {
	hipLaunchKernelGGL(counter_wait_k, dim3(1), dim3(64), 0, s, counter, need, abort_flag);
	HIP_TRY(hipGetLastError());
	return ISING_OK;
}

int ising_ipc::counter_set_on(hipStream_t s, uint32_t *counter, uint32_t value) {
	hipLaunchKernelGGL(counter_set_k, dim3(1), dim3(64), 0, s, counter, value);
	HIP_TRY(hipGetLastError());
	return ISING_OK;
}

int ising_ipc::push_rows(ising_ctx *c, int color, int depth, const void *first, const void *last, const void *halo_top, const void *halo_bot, size_t bytes) {
	ising_ipc_state *ipc = c->ipc;
	if (!ipc || !ipc->attached) return fail(ISING_E_STATE, "the IPC transport serves one attached slab per process");
	const bool spin = color != ISING_HAM_BLACK;
	const char *my_base = reinterpret_cast<const char *>(spin ? c->d_lat : c->d_ham);
	char *const *nb = spin ? ipc->nb_lat : ipc->nb_ham; // (the neighbours' arrays have this slab's shape: same offsets)
	const ptrdiff_t off_top = static_cast<const char *>(halo_top) - my_base, off_bot = static_cast<const char *>(halo_bot) - my_base;
	if (int rc = bind(c)) return rc;
	// ghost rows (deep exchange) are read AND written by their owner's launches: wait until it has announced the epoch
	if (depth > 1 && spin)
		if (int rc = ipc_wait_on(c, c->comm, ipc->mine_flags + ipc_flag(color, F_FREE_PREV), ipc->free_epoch[color],
		                         ipc->mine_flags + ipc_flag(color, F_FREE_NEXT), ipc->free_epoch[color])) return rc;
	HIP_TRY(hipMemcpyAsync(nb[1] + off_top, last, bytes, hipMemcpyDeviceToDevice, c->comm));  // next slab's top rows <- my last rows
	HIP_TRY(hipMemcpyAsync(nb[0] + off_bot, first, bytes, hipMemcpyDeviceToDevice, c->comm)); // previous slab's bottom rows <- my first rows
	const uint32_t ep = ++ipc->epoch[color];
	return ipc_set_on(c->comm, ipc->nb_flags[1] + ipc_flag(color, F_READY_TOP), ep, ipc->nb_flags[0] + ipc_flag(color, F_READY_BOT), ep);
}

extern "C" {

int ising_ipc_export(ising_ctx *c, void *blob_out) {
	if (!c || !blob_out) return fail(ISING_E_ARG, "null argument");
	if (c->wrap) return fail(ISING_E_STATE, "a slab that wraps in place has no halo rows to exchange (nslabs == 1 without ring_halo)");
	if (c->rank_mode) return fail(ISING_E_STATE, "the slab is already attached");
	if (int rc = ising_host::ring_resources(c)) return rc;
	if (!c->ipc) c->ipc = new ising_ipc_state();
	ising_ipc_state *ipc = c->ipc;
	if (ipc->mine) {
		// A re-export (the attach behind an earlier export failed or was abandoned): neighbours that did attach may have pushed or
		// released into the old segment already, so its counters are not the zeros a new attachment starts from -- a waiter
		// would take epoch 1 for delivered before the rows arrive.  Every export gets a fresh segment under a fresh name.
		(void)hipStreamSynchronize(c->stream);
		if (c->comm) (void)hipStreamSynchronize(c->comm);
		(void)hipHostUnregister(ipc->mine);
		(void)munmap(ipc->mine, IPC_SEG_BYTES);
		(void)shm_unlink(ipc->shm_name.c_str());
		ipc->mine = nullptr;
		ipc->mine_flags = ipc->mine_abort = nullptr;
		ipc->shm_name.clear();
	}
	{
		static std::atomic<unsigned> serial{0};
		char name[40];
		const unsigned long long stamp = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
		snprintf(name, sizeof(name), "/ising_ipc_%d_%u_%llx", (int)getpid(), serial++, stamp & 0xFFFFFFFFull);
		const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
		if (fd < 0) return fail(ISING_E_IO, "shm_open(%s) failed: %s", name, strerror(errno));
		if (ftruncate(fd, (off_t)IPC_SEG_BYTES) != 0) { (void)close(fd); (void)shm_unlink(name); return fail(ISING_E_IO, "ftruncate(%s) failed: %s", name, strerror(errno)); }
		void *m = mmap(nullptr, IPC_SEG_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		(void)close(fd);
		if (m == MAP_FAILED) { (void)shm_unlink(name); return fail(ISING_E_IO, "mmap(%s) failed: %s", name, strerror(errno)); }
		memset(m, 0, IPC_SEG_BYTES);
		const hipError_t e = hipHostRegister(m, IPC_SEG_BYTES, hipHostRegisterMapped | hipHostRegisterPortable);
		void *dev = nullptr;
		if (e != hipSuccess || hipHostGetDevicePointer(&dev, m, 0) != hipSuccess) {
			(void)hipGetLastError();
			if (e == hipSuccess) (void)hipHostUnregister(m);
			(void)munmap(m, IPC_SEG_BYTES);
			(void)shm_unlink(name);
			return fail(ISING_E_HIP, "cannot register the shared flag segment with HIP: %s", hipGetErrorString(e));
		}
		ipc->shm_name = name;
		ipc->mine = static_cast<IpcSegment *>(m);
		ipc->mine->magic = IPC_MAGIC;
		ipc->mine->nslabs = (uint32_t)c->cfg.nslabs;
		ipc->mine->slab = (uint32_t)c->cfg.slab;
		ipc->mine_flags = reinterpret_cast<uint32_t *>(static_cast<char *>(dev) + offsetof(IpcSegment, flags));
		ipc->mine_abort = reinterpret_cast<uint32_t *>(static_cast<char *>(dev) + offsetof(IpcSegment, abort));
	}
	IpcBlob b;
	memset(&b, 0, sizeof(b));
	b.magic = IPC_MAGIC; b.version = IPC_VERSION;
	b.nslabs = c->cfg.nslabs; b.slab = c->cfg.slab; b.X = c->cfg.X; b.Y = c->cfg.Y; b.layout = c->layout(); b.ghost = c->ghost();
	b.use_J = c->cfg.use_J; b.pid = (int32_t)getpid();
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, c->cfg.device));
	b.pci[0] = prop.pciDomainID; b.pci[1] = prop.pciBusID; b.pci[2] = prop.pciDeviceID;
	auto export_array = [&](void *ptr, uint64_t *raw, uint64_t *off, hipIpcMemHandle_t *h) -> int {
		hipDeviceptr_t base = nullptr;
		size_t size = 0;
		HIP_TRY(hipMemGetAddressRange(&base, &size, ptr)); // (a caller-owned buffer may sit inside a larger allocation)
		*raw = (uint64_t)(uintptr_t)ptr;
		*off = (uint64_t)((char *)ptr - (char *)base);
		HIP_TRY(hipIpcGetMemHandle(h, base));
		return ISING_OK;
	};
	if (int rc = export_array(c->d_lat, &b.lat_ptr, &b.lat_off, &b.lat_handle)) return rc;
	if (c->cfg.use_J) if (int rc = export_array(c->d_ham, &b.ham_ptr, &b.ham_off, &b.ham_handle)) return rc;
	snprintf(b.shm_name, sizeof(b.shm_name), "%s", ipc->shm_name.c_str());
	memset(blob_out, 0, ISING_IPC_BLOB_BYTES);
	memcpy(blob_out, &b, sizeof(b));
	return ISING_OK;
}

int ising_ipc_attach(ising_ctx *c, const void *blobs, int nblobs) {
	if (!c || !blobs) return fail(ISING_E_ARG, "null argument");
	if (c->rank_mode || c->rccl_comm) return fail(ISING_E_STATE, "the slab is already attached");
	ising_ipc_state *ipc = c->ipc;
	if (!ipc || !ipc->mine) return fail(ISING_E_STATE, "ising_ipc_export first");
	const int n = c->cfg.nslabs, me = c->cfg.slab;
	if (nblobs != n) return fail(ISING_E_ARG, "%d blobs for a ring of %d slabs", nblobs, n);
	if (int rc = bind(c)) return rc;
	std::vector<IpcBlob> bl(n);
	for (int r = 0; r < n; r++) {
		memcpy(&bl[r], static_cast<const char *>(blobs) + (size_t)r * ISING_IPC_BLOB_BYTES, sizeof(IpcBlob));
		const IpcBlob &b = bl[r];
		if (b.magic != IPC_MAGIC || b.version != IPC_VERSION) return fail(ISING_E_ARG, "blob %d is not an ising_ipc_export blob of this library", r);
		if (b.nslabs != n || b.slab != r) return fail(ISING_E_ARG, "blob %d describes slab %d of %d", r, b.slab, b.nslabs);
		if (b.X != c->cfg.X || b.Y != c->cfg.Y || b.layout != c->layout() || b.ghost != c->ghost() || b.use_J != c->cfg.use_J)
			return fail(ISING_E_ARG, "slab %d differs in shape, device layout, ghost rows or couplings (a ring exchanges rows of one form)", r);
	}
	if (bl[me].pid != (int32_t)getpid() || ipc->shm_name != bl[me].shm_name) return fail(ISING_E_ARG, "blob %d is not the one this context exported", me);
	ipc->sharing = 0;
	for (int r = 0; r < n; r++) if (!memcmp(bl[r].pci, bl[me].pci, sizeof(bl[r].pci))) ipc->sharing++;
	// every rank's flag segment (hosts: all-reduce); the two neighbours' also for the device
	ipc->seg.assign(n, nullptr);
	auto give_up = [&](int code, const char *what, const char *why) { ipc_unmap(c); return fail(code, "%s: %s", what, why); };
	for (int r = 0; r < n; r++) {
		if (r == me) { ipc->seg[r] = ipc->mine; continue; }
		char nm[sizeof(bl[r].shm_name)];
		memcpy(nm, bl[r].shm_name, sizeof(nm));
		nm[sizeof(nm) - 1] = 0;
		const int fd = shm_open(nm, O_RDWR, 0600);
		if (fd < 0) return give_up(ISING_E_IO, "cannot open a neighbour's flag segment", strerror(errno));
		void *m = mmap(nullptr, IPC_SEG_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		(void)close(fd);
		if (m == MAP_FAILED) return give_up(ISING_E_IO, "cannot map a neighbour's flag segment", strerror(errno));
		ipc->seg[r] = static_cast<IpcSegment *>(m);
		if (ipc->seg[r]->magic != IPC_MAGIC || (int)ipc->seg[r]->slab != r) return give_up(ISING_E_ARG, "flag segment", "not the segment of that slab");
	}
	const int nbr[2] = {(me + n - 1) % n, (me + 1) % n};
	for (int k = 0; k < 2; k++) {
		const int r = nbr[k];
		if (k == 1 && nbr[1] == nbr[0]) { ipc->nb_flags[1] = ipc->nb_flags[0]; ipc->nb_lat[1] = ipc->nb_lat[0]; ipc->nb_ham[1] = ipc->nb_ham[0]; break; }
		if (r == me) { // a ring of one sends to itself
			ipc->nb_flags[k] = ipc->mine_flags;
			ipc->nb_lat[k] = reinterpret_cast<char *>(c->d_lat);
			ipc->nb_ham[k] = reinterpret_cast<char *>(c->d_ham);
			continue;
		}
		hipError_t e = hipHostRegister(ipc->seg[r], IPC_SEG_BYTES, hipHostRegisterMapped | hipHostRegisterPortable);
		void *dev = nullptr;
		if (e == hipSuccess) { ipc->registered.push_back(ipc->seg[r]); e = hipHostGetDevicePointer(&dev, ipc->seg[r], 0); }
		if (e != hipSuccess) { (void)hipGetLastError(); return give_up(ISING_E_HIP, "cannot register a neighbour's flag segment with HIP", hipGetErrorString(e)); }
		ipc->nb_flags[k] = reinterpret_cast<uint32_t *>(static_cast<char *>(dev) + offsetof(IpcSegment, flags));
		auto map_array = [&](const hipIpcMemHandle_t &h, uint64_t raw, uint64_t off, char **out) -> hipError_t {
			if (bl[r].pid == (int32_t)getpid()) { *out = reinterpret_cast<char *>((uintptr_t)raw); return hipSuccess; } // (a handle cannot be opened where it was made)
			void *base = nullptr;
			const hipError_t e2 = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
			if (e2 != hipSuccess) return e2;
			ipc->opened.push_back(base);
			*out = static_cast<char *>(base) + off;
			return hipSuccess;
		};
		e = map_array(bl[r].lat_handle, bl[r].lat_ptr, bl[r].lat_off, &ipc->nb_lat[k]);
		if (e == hipSuccess && c->cfg.use_J) e = map_array(bl[r].ham_handle, bl[r].ham_ptr, bl[r].ham_off, &ipc->nb_ham[k]);
		if (e != hipSuccess) { (void)hipGetLastError(); return give_up(ISING_E_HIP, "hipIpcOpenMemHandle of a neighbour's arrays failed", hipGetErrorString(e)); }
	}
	for (int p = 0; p < IPC_PLANES; p++) ipc->epoch[p] = ipc->free_epoch[p] = ipc->seen[p] = 0;
	// (the counters in the segments start at zero and only ever grow: ising_rank_detach destroys the segment with the attachment)
	ipc->red_seq = 0;
	ipc->attached = true;
	c->rank_mode = true;
	c->transport = ISING_TRANSPORT_IPC;
	c->ring_prev = c->ring_next = nullptr;
	c->copy_inline = c->store_ring = false;
	c->ghost_depth[0] = c->ghost_depth[1] = 0;
	return ISING_OK;
}

} // extern "C"
