// ising_batch.cpp -- a batch of independent lattices of one shape advancing together (include/ising_hip.h: ising_batch_*).
// The reference runs one lattice per process and per temperature (optimized/main.cu:1596-1598); a temperature series
// (BASELINE config 5: 31 lattices of 8192^2) is then 31 runs, each of which fills an MI355X to 70 %.  Here ONE fused launch
// carries a level of every lattice of the batch: a level of 8192^2 alone has 128 tickets at 16-row strips, 31 of them have
// 3968 -- enough for tall strips on a full chip (ising_ballot.hip: ballot_update_k<BATCH>), and one more launch measures
// all of them on the ballot layout itself (ballot_measure_k).  Every lattice's spins are what a run of its own gives.
#include "ising_ctx.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using ising_host::bind;
using ising_host::fail;

struct ising_batch {
	std::vector<ising_ctx *> m;
	int device = 0;
	int H = 1, wg_per_cu = 0, nt = 0, nstrips = 0;
	bool wait_late = true;                                   // units draw their first row before they wait for their parents (UpdateParams.wait_late)
	ising::ReplicaParams *d_rep = nullptr, *h_rep = nullptr; // the lattices' records: device copy and pinned staging
	hipEvent_t ev_upload = nullptr;                          // the last upload has left the staging buffer
	bool upload_pending = false;
	uint32_t *d_ctl = nullptr;                               // ticket words, then nrep x (nstrips + 2) completion counters
	uint32_t done_base = 0;
	unsigned long long ticket_base = 0;
	uint32_t *h_abort = nullptr;                             // pinned, device-visible: a batched launch that gave up raises it (the batch's OWN
	                                                         // word: a member's is cleared by any call on that member, which knows nothing of d_ctl)
	static constexpr int MEAS_CAP = 1024;
	static constexpr size_t MEAS_WORDS = (size_t)ising::BALLOT_MEASURE_SLOTS * 8; // per measurement and lattice: 16 partial (up, bond sum) pairs, a line each
	unsigned long long *d_meas = nullptr, *h_meas = nullptr; // [measurement][lattice][slot][8]
	int meas_pending = 0;
	int n() const { return (int)m.size(); }
	hipStream_t stream() const { return m[0]->stream; }
};

namespace {

int refresh_records(ising_batch *b) {
	bool dirty = false;
	std::vector<ising::ReplicaParams> want(b->n());
	for (int r = 0; r < b->n(); r++) {
		const ising_ctx *c = b->m[r];
		if (!c->ballot || !c->wrap) return fail(ISING_E_STATE, "lattice %d of the batch left the ballot layout", r);
		if (!c->fast_ok || ising_host::needs_generic(c)) return fail(ISING_E_STATE, "lattice %d: temperature %g has no integer accept thresholds (sweep it on its own)", r, (double)c->cfg.temp);
		if (c->stream != b->stream()) return fail(ISING_E_STATE, "the lattices of a batch share one stream (lattice %d has another)", r);
		want[r].lat[0] = c->lat(ISING_BLACK);
		want[r].lat[1] = c->lat(ISING_WHITE);
		want[r].n3 = (uint32_t)c->thr[3];
		want[r].n4 = (uint32_t)c->thr[4];
		want[r].seed_lo = (uint32_t)c->cfg.seed;
		want[r].seed_hi = (uint32_t)(c->cfg.seed >> 32);
		dirty = dirty || memcmp(&want[r], &b->h_rep[r], sizeof(want[r])) != 0;
	}
	if (!dirty) return ISING_OK;
	if (b->upload_pending) HIP_TRY(hipEventSynchronize(b->ev_upload)); // (temperatures change a few times per series)
	memcpy(b->h_rep, want.data(), want.size() * sizeof(want[0]));
	HIP_TRY(hipMemcpyAsync(b->d_rep, b->h_rep, want.size() * sizeof(want[0]), hipMemcpyHostToDevice, b->stream()));
	HIP_TRY(hipEventRecord(b->ev_upload, b->stream()));
	b->upload_pending = true;
	return ISING_OK;
}

// A batched launch gave up (UpdateParams.abort_flag): the batch's tickets and completion counters start over, pending
// measurements are dropped, ISING_E_STATE.  Looked at by every batch call, so that the launches after a fault do not each wait
// ~10 s for counters that are out of step.
int batch_check_abort(ising_batch *b) {
	if (!b->h_abort || !__atomic_load_n(b->h_abort, __ATOMIC_ACQUIRE)) return ISING_OK;
	(void)hipStreamSynchronize(b->stream());
	(void)hipMemset(b->d_ctl, 0, ising_host::SLOTCTL_TICKET_BYTES + (size_t)b->n() * ((size_t)b->nstrips + 2) * sizeof(uint32_t));
	b->done_base = 0;
	b->ticket_base = 0;
	b->meas_pending = 0;
	__atomic_store_n(b->h_abort, 0u, __ATOMIC_RELEASE);
	return fail(ISING_E_STATE, "a batched fused launch gave up: its units' parents never completed; the batch's tickets and counters have been reset, "
	                           "the members' lattices are undefined -- initialise or load them again");
}

} // namespace

extern "C" {

int ising_batch_create(ising_ctx **ctxs, int n, ising_batch **out) {
	if (!ctxs || !out || n < 1) return fail(ISING_E_ARG, "bad batch");
	*out = nullptr;
	for (int r = 0; r < n; r++) {
		const ising_ctx *c = ctxs[r];
		if (!c) return fail(ISING_E_ARG, "batch slot %d is null", r);
		if (!c->wrap || c->cfg.nslabs != 1) return fail(ISING_E_ARG, "a batch holds whole lattices (nslabs == 1 without ring halo rows); slot %d is a ring slab", r);
		if (!c->ballot || c->cfg.XSL || c->cfg.use_J) return fail(ISING_E_ARG, "a batch needs the ballot layout without sub-lattices and couplings (slot %d)", r);
		if (c->cfg.X != ctxs[0]->cfg.X || c->cfg.Y != ctxs[0]->cfg.Y || c->cfg.device != ctxs[0]->cfg.device || c->lld != ctxs[0]->lld)
			return fail(ISING_E_ARG, "the lattices of a batch have one shape and one device (slot %d differs)", r);
		for (int q = 0; q < r; q++) if (ctxs[q] == c) return fail(ISING_E_ARG, "slot %d repeats slot %d", r, q);
	}
	ising_batch *b = new ising_batch();
	b->m.assign(ctxs, ctxs + n);
	const ising_ctx *c0 = ctxs[0];
	b->device = c0->cfg.device;
	// strips as tall as a level of ALL lattices allows (8192^2 x 31: 8 rows, five workgroups per CU; alone: 2 rows, three)
	b->wait_late = c0->pol.fused_wait_late != 0;
	ising_host::fused_shape(c0->nwc(), c0->cfg.Y, (long long)c0->cfg.Y * n, &b->H, &b->wg_per_cu, b->wait_late);
	b->nstrips = c0->cfg.Y / b->H;
	// lattices that together exceed the 256 MB memory-side cache stream through it (ising_capi.cpp: fused_nt)
	b->nt = (long long)c0->cfg.X * c0->cfg.Y * n > (1LL << 31);
	hipError_t e = hipSetDevice(b->device);
	const size_t ctl_bytes = ising_host::SLOTCTL_TICKET_BYTES + (size_t)n * ((size_t)b->nstrips + 2) * sizeof(uint32_t);
	if (e == hipSuccess) e = hipMalloc((void **)&b->d_rep, (size_t)n * sizeof(ising::ReplicaParams));
	if (e == hipSuccess) e = hipHostMalloc((void **)&b->h_rep, (size_t)n * sizeof(ising::ReplicaParams), hipHostMallocDefault);
	if (e == hipSuccess) { memset(b->h_rep, 0, (size_t)n * sizeof(ising::ReplicaParams)); e = hipEventCreateWithFlags(&b->ev_upload, hipEventDisableTiming); }
	if (e == hipSuccess) e = hipMalloc((void **)&b->d_ctl, ctl_bytes);
	if (e == hipSuccess) e = hipMemset(b->d_ctl, 0, ctl_bytes);
	const size_t meas_bytes = (size_t)ising_batch::MEAS_CAP * n * ising_batch::MEAS_WORDS * sizeof(unsigned long long);
	if (e == hipSuccess) e = hipMalloc((void **)&b->d_meas, meas_bytes);
	if (e == hipSuccess) e = hipMemset(b->d_meas, 0, meas_bytes);
	if (e == hipSuccess) e = hipHostMalloc((void **)&b->h_meas, meas_bytes, hipHostMallocDefault);
	if (e == hipSuccess) e = hipHostMalloc((void **)&b->h_abort, 64, hipHostMallocMapped);
	if (e == hipSuccess) memset(b->h_abort, 0, 64);
	if (e != hipSuccess) {
		const int rc = fail(ISING_E_HIP, "batch allocation failed: %s", hipGetErrorString(e));
		ising_batch_destroy(b);
		return rc;
	}
	*out = b;
	return ISING_OK;
}

int ising_batch_destroy(ising_batch *b) {
	if (!b) return ISING_OK;
	(void)hipSetDevice(b->device);
	(void)hipDeviceSynchronize(); // (not the members' stream: a caller may have destroyed them first, against the header's advice)
	if (b->d_rep) (void)hipFree(b->d_rep);
	if (b->h_rep) (void)hipHostFree(b->h_rep);
	if (b->ev_upload) (void)hipEventDestroy(b->ev_upload);
	if (b->d_ctl) (void)hipFree(b->d_ctl);
	if (b->d_meas) (void)hipFree(b->d_meas);
	if (b->h_meas) (void)hipHostFree(b->h_meas);
	if (b->h_abort) (void)hipHostFree(b->h_abort);
	delete b;
	return ISING_OK;
}

int ising_batch_info(ising_batch *b, int *strip_rows, int *wg_per_cu, int *lattices) {
	if (!b) return fail(ISING_E_ARG, "null batch");
	if (strip_rows) *strip_rows = b->H;
	if (wg_per_cu) *wg_per_cu = b->wg_per_cu;
	if (lattices) *lattices = b->n();
	return ISING_OK;
}

int ising_batch_sweep(ising_batch *b, int first_it, int nsweeps) {
	if (!b) return fail(ISING_E_ARG, "null batch");
	if (first_it < 0 || nsweeps < 0) return fail(ISING_E_ARG, "bad iteration range");
	HIP_TRY(hipSetDevice(b->device));
	if (int rc = batch_check_abort(b)) return rc;
	if (int rc = refresh_records(b)) return rc;
	ising_ctx *c0 = b->m[0];
	const int nwc = c0->nwc(), Y = c0->cfg.Y;
	const size_t rowb = (size_t)c0->lld * sizeof(uint64_t);
	const int per_launch = ising_host::fused_sweeps_per_launch(c0->pol, (long long)c0->cfg.X * Y * b->n());
	for (int it = first_it, left = nsweeps; left > 0;) {
		const int ns = std::min(left, per_launch);
		ising::UpdateParams p{};
		p.wrap = 1; // every lattice wraps in place: the launch that writes an edge row also writes its mirror
		p.mir0_bytes = (long long)Y * (long long)rowb;
		p.mirL_bytes = -(long long)Y * (long long)rowb;
		p.it = (uint32_t)it;
		p.color = ISING_BLACK;
		p.gx = c0->gx;
		p.Y = Y;
		p.slV = c0->gx * 32;
		p.H = b->H;
		p.row_lo[0] = 0; p.row_hi[0] = Y;
		p.nreal0 = p.nunits0 = p.nunits = 4 * nwc * b->nstrips; // (units of ONE lattice: the launcher multiplies)
		p.scratch = c0->d_scratch;                               // accept-mask slots belong to the workgroup slot: any member's will do
		p.ticket = reinterpret_cast<unsigned long long *>(b->d_ctl);
		p.nlevels = 2 * ns;
		p.ticket_base2[0] = b->ticket_base;
		p.done = b->d_ctl + ising_host::SLOTCTL_TICKET_BYTES / 4;
		p.done_stride = b->nstrips + 2;
		p.done_base = b->done_base;
		p.wg_per_cu = b->wg_per_cu;
		p.wait_late = b->wait_late ? (c0->pol.fused_wait_late == 1 ? 1 : 2) : 0;
		p.nt_stream = b->nt;
		p.rep = b->d_rep;
		p.nrep = b->n();
		p.cus = c0->cus;
		p.grid_cap = c0->pol.fused_wgs;
		p.abort_flag = b->h_abort;
		p.abort_polls = c0->pol.abort_polls;
		if (b->done_base > (1u << 30)) { // keep the monotone completion counters far from wrapping
			HIP_TRY(hipMemsetAsync(b->d_ctl + ising_host::SLOTCTL_TICKET_BYTES / 4, 0, (size_t)b->n() * p.done_stride * sizeof(uint32_t), b->stream()));
			b->done_base = p.done_base = 0;
		}
		int grid = 0;
		HIP_TRY(ising::launch_ballot_update(p, b->stream(), &grid, nullptr));
		b->done_base += (uint32_t)p.nlevels * (uint32_t)nwc;
		b->ticket_base += (unsigned long long)p.nwg * (unsigned long long)p.nlevels + (unsigned long long)grid; // (every workgroup draws one ticket too many)
		it += ns;
		left -= ns;
	}
	return ISING_OK;
}

// test aid (include/ising_hip_testing.h): the host's record of the batch's completion counters out of step with the device
int ising_batch_debug_fault(ising_batch *b, int polls) {
	if (!b) return fail(ISING_E_ARG, "null batch");
	b->done_base += 1u << 20;
	if (polls > 0) b->m[0]->pol.abort_polls = (uint32_t)polls; // (the bound travels with member 0's policy)
	return ISING_OK;
}

int ising_batch_measure_enqueue(ising_batch *b) {
	if (!b) return fail(ISING_E_ARG, "null batch");
	HIP_TRY(hipSetDevice(b->device));
	if (int rc = batch_check_abort(b)) return rc;
	if (b->meas_pending >= ising_batch::MEAS_CAP) return fail(ISING_E_STATE, "%d measurements pending: ising_batch_measure_fetch first", b->meas_pending);
	if (int rc = refresh_records(b)) return rc;
	const ising_ctx *c0 = b->m[0];
	HIP_TRY(ising::launch_ballot_measure(b->d_rep, b->n(), c0->gx, c0->cfg.Y, b->d_meas + (size_t)b->meas_pending * b->n() * ising_batch::MEAS_WORDS, b->stream()));
	b->meas_pending++;
	return ISING_OK;
}

int ising_batch_measure_fetch(ising_batch *b, uint64_t *up, int64_t *bond_equal, int max_n, int *n) {
	if (!b || !up || !bond_equal || !n || max_n < 0) return fail(ISING_E_ARG, "bad argument");
	HIP_TRY(hipSetDevice(b->device));
	if (b->meas_pending > max_n) return fail(ISING_E_ARG, "%d measurements pending, room for %d", b->meas_pending, max_n);
	const size_t pairs = (size_t)b->meas_pending * b->n(), words = pairs * ising_batch::MEAS_WORDS;
	if (words) {
		HIP_TRY(hipMemcpyAsync(b->h_meas, b->d_meas, words * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream()));
		HIP_TRY(hipMemsetAsync(b->d_meas, 0, words * sizeof(unsigned long long), b->stream())); // the accumulators of the next round
	}
	HIP_TRY(hipStreamSynchronize(b->stream()));
	if (int rc = batch_check_abort(b)) return rc;
	for (size_t i = 0; i < pairs; i++) {
		unsigned long long u = 0, a = 0;
		for (int s = 0; s < ising::BALLOT_MEASURE_SLOTS; s++) { u += b->h_meas[i * ising_batch::MEAS_WORDS + 8 * s]; a += b->h_meas[i * ising_batch::MEAS_WORDS + 8 * s + 1]; }
		up[i] = u;
		bond_equal[i] = (int64_t)a;
	}
	*n = b->meas_pending;
	b->meas_pending = 0;
	return ISING_OK;
}

} // extern "C"
