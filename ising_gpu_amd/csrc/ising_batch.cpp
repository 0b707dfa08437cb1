// ising_batch.cpp -- a batch of independent lattices of one shape advancing together (include/ising_hip.h: ising_batch_*).
// The reference runs one lattice per process and per temperature (optimized/main.cu:1596-1598); a temperature series
// (BASELINE config 5: 31 lattices of 8192^2) is then 31 runs, each of which fills an MI355X to 70 %.  Here ONE fused launch
// carries a level of every lattice of the batch: a level of 8192^2 alone has 128 tickets at 16-row strips, 31 of them have
// 3968 -- enough for tall strips on a full chip (ising_ballot.hip: ballot_update_k<BATCH>), and one more launch measures
// all of them on the ballot layout itself (ballot_measure_k).  Every lattice's spins are what a run of its own gives.
// Round 6: lattices of the quad path (ising_quad.hip: up to ~2^26 spins -- the finite-size end of a temperature series: L = 2048, 4096 x 31 temperatures x chains)
// form batches too: ONE quad_pass_k launch per pass carries the tiles of all of them and the draws of all of them for the pass to come; a lone 2048^2 has 64
// tiles for 256 CUs, thirty-one of them have the many-tiles regime's shape (tall tiles, short passes).  The reference's own many-small-systems mode
// (optimized/main.cu:1423-1457, --xsl/--ysl: one temperature) runs at its big-lattice rate; this is the same for lattices that differ in temperature and seed.
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
	// quad kind: the members sweep on the quad path; tile and pass of the batch's own choosing (many tiles), the members' planes and mask buffers
	bool quad = false;
	int qC = 0, qT = 0, qHG = 0, qwaves = 0;
	ising::QuadRec *d_qrec = nullptr, *h_qrec = nullptr;
	unsigned long long *d_qcnt = nullptr; // print points of a counted call: [point][lattice][8 up + 8 equal bonds]
	size_t qcnt_cap = 0;
	int n() const { return (int)m.size(); }
	hipStream_t stream() const { return m[0]->stream; }
};

namespace {

int refresh_records(ising_batch *b) {
	if (b->quad) {
		bool dirty = false;
		std::vector<ising::QuadRec> want(b->n());
		for (int r = 0; r < b->n(); r++) {
			const ising_ctx *c = b->m[r];
			if (!ising_host::quad_ready(c)) return fail(ISING_E_STATE, "lattice %d of the batch left the quad path (temperature %g has no integer accept thresholds?)", r, (double)c->cfg.temp);
			if (c->stream != b->stream()) return fail(ISING_E_STATE, "the lattices of a batch share one stream (lattice %d has another)", r);
			memset(&want[r], 0, sizeof(want[r]));
			want[r].quad = c->d_quad;
			want[r].masks = c->d_qmasks;
			want[r].dense[0] = reinterpret_cast<uint32_t *>(c->lat(ISING_BLACK));
			want[r].dense[1] = reinterpret_cast<uint32_t *>(c->lat(ISING_WHITE));
			want[r].seed_lo = (uint32_t)c->cfg.seed;
			want[r].seed_hi = (uint32_t)(c->cfg.seed >> 32);
			want[r].n3 = (uint32_t)c->thr[3];
			want[r].n4 = (uint32_t)c->thr[4];
			dirty = dirty || memcmp(&want[r], &b->h_qrec[r], sizeof(want[r])) != 0;
		}
		if (!dirty) return ISING_OK;
		if (b->upload_pending) HIP_TRY(hipEventSynchronize(b->ev_upload));
		memcpy(b->h_qrec, want.data(), want.size() * sizeof(want[0]));
		HIP_TRY(hipMemcpyAsync(b->d_qrec, b->h_qrec, want.size() * sizeof(want[0]), hipMemcpyHostToDevice, b->stream()));
		HIP_TRY(hipEventRecord(b->ev_upload, b->stream()));
		b->upload_pending = true;
		return ISING_OK;
	}
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
	if (b->quad) return ISING_OK; // (nothing waits inside a quad pass)
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

// `nsweeps` sweeps of every lattice on the quad path (ising_update.cpp: sweep_quad is the lone form): launch k = the word pass k of ALL lattices on the masks
// launch k - 1 drew + everybody's draws for pass k + 1.  The members' spins live in their dense planes between calls.  `every` > 0: print points as in
// ising_sweep_counted, the sums of point m and lattice r in d_qcnt[(m nrep + r) 16 ..] (eight partial up counts, eight partial sums of equal bonds).
int batch_sweep_quad(ising_batch *b, int first_it, int nsweeps, int every, bool bonds, int *nmeas) {
	if (int rc = refresh_records(b)) return rc;
	int k = 0;
	const std::vector<ising_host::QuadPass> passes = ising_host::quad_passes(first_it, nsweeps, every, b->qT, &k);
	if (nmeas) *nmeas = k;
	if (passes.empty()) return ISING_OK;
	const ising_ctx *c0 = b->m[0];
	const int NRG = c0->cfg.Y / 4, gx = c0->gx, n = b->n();
	const size_t qw = c0->quad_words(), NI = qw / 64;
	const size_t mask_words = (size_t)(2 * b->qT) * NI * 128; // per buffer (the members' buffers are at least as long: qT <= their own pass)
	hipStream_t st = b->stream();
	if (every > 0) {
		const size_t need = (size_t)k * n * 16;
		if (b->qcnt_cap < need) {
			if (b->d_qcnt) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(b->d_qcnt)); b->d_qcnt = nullptr; b->qcnt_cap = 0; }
			HIP_TRY(hipMalloc((void **)&b->d_qcnt, std::max<size_t>(need, 1024) * sizeof(unsigned long long)));
			b->qcnt_cap = std::max<size_t>(need, 1024);
		}
		if (need) HIP_TRY(hipMemsetAsync(b->d_qcnt, 0, need * sizeof(unsigned long long), st));
	}
	HIP_TRY(ising::launch_quad_convert_batch(b->d_qrec, n, 0, true, gx, NRG, st));
	int cur = 0;
	const int np = (int)passes.size();
	for (int q = -1; q < np; q++) {
		ising::QuadPassParams pp{};
		pp.w.gx = pp.d.gx = gx;
		pp.w.NRG = pp.d.NRG = NRG;
		pp.w.C = b->qC;
		pp.w.HG = b->qHG;
		pp.cus = c0->cus;
		pp.w.rep = pp.d.rep = b->d_qrec;
		pp.w.nrep = pp.d.nrep = n;
		if (q >= 0) {
			for (int color = 0; color < 2; color++) {
				pp.w.src_off[color] = ((size_t)cur * 2 + color) * qw;
				pp.w.dst_off[color] = ((size_t)(cur ^ 1) * 2 + color) * qw;
			}
			pp.w.mask_off = (size_t)(q & 1) * mask_words;
			pp.w.nlev = 2 * passes[q].ns;
			if (passes[q].meas >= 0) {
				pp.w.cnt = b->d_qcnt + (size_t)passes[q].meas * n * 16;
				pp.w.cnt_eq = bonds ? pp.w.cnt + 8 : nullptr;
				pp.w.cnt_stride = 16;
			}
			cur ^= 1;
		}
		if (q + 1 < np) {
			pp.d.mask_off = (size_t)((q + 1) & 1) * mask_words;
			pp.d.it = (uint32_t)passes[q + 1].it;
			pp.d.nlev = 2 * passes[q + 1].ns;
		}
		HIP_TRY(ising::launch_quad_pass(pp, b->qwaves, st));
	}
	HIP_TRY(ising::launch_quad_convert_batch(b->d_qrec, n, cur, false, gx, NRG, st));
	return ISING_OK;
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
		const bool quad_r = !c->ballot && c->quad_C > 0 && c->d_quad && c->d_qmasks;
		if ((!c->ballot && !quad_r) || c->cfg.XSL || c->cfg.use_J)
			return fail(ISING_E_ARG, "a batch needs the ballot layout or a lattice of the quad path, without sub-lattices and couplings (slot %d)", r);
		if (r > 0 && quad_r != (!ctxs[0]->ballot)) return fail(ISING_E_ARG, "the lattices of a batch sweep the same way (slot %d differs)", r);
		if (c->cfg.X != ctxs[0]->cfg.X || c->cfg.Y != ctxs[0]->cfg.Y || c->cfg.device != ctxs[0]->cfg.device || c->lld != ctxs[0]->lld)
			return fail(ISING_E_ARG, "the lattices of a batch have one shape and one device (slot %d differs)", r);
		for (int q = 0; q < r; q++) if (ctxs[q] == c) return fail(ISING_E_ARG, "slot %d repeats slot %d", r, q);
	}
	ising_batch *b = new ising_batch();
	b->m.assign(ctxs, ctxs + n);
	const ising_ctx *c0 = ctxs[0];
	b->device = c0->cfg.device;
	b->quad = !c0->ballot;
	if (b->quad) {
		// the many-tiles shape of n x Y rows, its passes no longer than the members' own (their mask buffers hold 2 x quad_T levels)
		int C = 0, T = 0, W = 0;
		ising_host::quad_shape(c0->gx, (long long)c0->cfg.Y * n, &C, &T, &W);
		if (c0->pol.quad_T) T = c0->pol.quad_T;
		if (c0->pol.quad_C) C = c0->pol.quad_C;
		if (c0->pol.quad_waves) W = c0->pol.quad_waves;
		for (int r = 0; r < n; r++) T = std::min(T, ctxs[r]->quad_T);
		T = std::max(1, std::min(T, 32));
		C = std::max(1, std::min(C, c0->cfg.Y / 4));
		W = std::max(1, std::min(W, 16));
		ising::QuadWordParams qp{};
		qp.gx = c0->gx; qp.NRG = c0->cfg.Y / 4; qp.C = C; qp.HG = (2 * T - 1 + 3) / 4;
		int lds_max = 64 * 1024;
		if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeSharedMemPerBlockOptin, b->device) != hipSuccess || lds_max <= 0) { (void)hipGetLastError(); lds_max = 64 * 1024; }
		if (ising::quad_word_maxi(qp, W) > 0 && ising::quad_pass_lds_bytes(qp, W) + 64 <= (size_t)lds_max) { b->qC = C; b->qT = T; b->qHG = qp.HG; b->qwaves = W; }
		else { b->qC = c0->quad_C; b->qT = c0->quad_T; b->qHG = c0->quad_HG; b->qwaves = c0->quad_waves; for (int r = 0; r < n; r++) b->qT = std::min(b->qT, ctxs[r]->quad_T); b->qHG = (2 * b->qT - 1 + 3) / 4; }
		hipError_t e = hipSetDevice(b->device);
		if (e == hipSuccess) e = hipMalloc((void **)&b->d_qrec, (size_t)n * sizeof(ising::QuadRec));
		if (e == hipSuccess) e = hipHostMalloc((void **)&b->h_qrec, (size_t)n * sizeof(ising::QuadRec), hipHostMallocDefault);
		if (e == hipSuccess) { memset(b->h_qrec, 0, (size_t)n * sizeof(ising::QuadRec)); e = hipEventCreateWithFlags(&b->ev_upload, hipEventDisableTiming); }
		if (e != hipSuccess) {
			const int rc = fail(ISING_E_HIP, "batch allocation failed: %s", hipGetErrorString(e));
			ising_batch_destroy(b);
			return rc;
		}
		*out = b;
		return ISING_OK;
	}
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
	if (b->d_qrec) (void)hipFree(b->d_qrec);
	if (b->h_qrec) (void)hipHostFree(b->h_qrec);
	if (b->d_qcnt) (void)hipFree(b->d_qcnt);
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
	if (strip_rows) *strip_rows = b->quad ? 4 * b->qC : b->H; // (quad kind: the rows of a tile; wg_per_cu: 0 -- ising_batch_quad_info has the rest)
	if (wg_per_cu) *wg_per_cu = b->quad ? 0 : b->wg_per_cu;
	if (lattices) *lattices = b->n();
	return ISING_OK;
}

int ising_batch_sweep(ising_batch *b, int first_it, int nsweeps) {
	if (!b) return fail(ISING_E_ARG, "null batch");
	if (first_it < 0 || nsweeps < 0) return fail(ISING_E_ARG, "bad iteration range");
	HIP_TRY(hipSetDevice(b->device));
	if (b->quad) return batch_sweep_quad(b, first_it, nsweeps, 0, false, nullptr);
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
	if (b->quad) { // the members are dense between calls: each measures itself (ising_batch_sweep_counted takes the print points inside the passes instead)
		for (ising_ctx *c : b->m) if (int rc = ising_measure_enqueue(c)) return rc;
		b->meas_pending++;
		return ISING_OK;
	}
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
	if (b->quad) {
		std::vector<uint64_t> u((size_t)std::max(b->meas_pending, 1));
		std::vector<int64_t> a((size_t)std::max(b->meas_pending, 1));
		for (int r = 0; r < b->n(); r++) {
			int k = 0;
			if (int rc = ising_measure_fetch(b->m[r], u.data(), a.data(), b->meas_pending, &k)) return rc;
			if (k != b->meas_pending) return fail(ISING_E_STATE, "lattice %d holds %d measurements, the batch enqueued %d (somebody measured a member on its own)", r, k, b->meas_pending);
			for (int i = 0; i < k; i++) { up[(size_t)i * b->n() + r] = u[i]; bond_equal[(size_t)i * b->n() + r] = a[i]; }
		}
		*n = b->meas_pending;
		b->meas_pending = 0;
		return ISING_OK;
	}
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

// ising_sweep_counted for a batch: `nsweeps` sweeps of every lattice, the up spins (and, bond_equal not null, ising_bond_equal's sums) after every iteration that
// is a multiple of `every`: up[k n + r] = point k, lattice r.  Quad kind: the print points ride in the passes (no launch of their own, no read-back in between);
// ballot kind: a batched launch per segment and one measuring launch per point (ising_batch_sweep + ising_batch_measure_enqueue).
int ising_batch_sweep_counted(ising_batch *b, int first_it, int nsweeps, int every, uint64_t *up, int64_t *bond_equal, int max_counts, int *ncounts) {
	if (!b || !up || !ncounts) return fail(ISING_E_ARG, "null argument");
	if (first_it < 0 || nsweeps < 0 || every < 1) return fail(ISING_E_ARG, "bad iteration range or count interval");
	const long long last = (long long)first_it + nsweeps - 1;
	const long long npts = nsweeps > 0 ? last / every - ((long long)first_it - 1) / every : 0;
	*ncounts = 0;
	if (npts > max_counts) return fail(ISING_E_ARG, "%lld counts, room for %d", npts, max_counts);
	HIP_TRY(hipSetDevice(b->device));
	if (b->meas_pending) return fail(ISING_E_STATE, "%d measurements pending: ising_batch_measure_fetch first", b->meas_pending);
	const int n = b->n();
	if (b->quad) {
		int k = 0;
		if (int rc = batch_sweep_quad(b, first_it, nsweeps, every, bond_equal != nullptr, &k)) return rc;
		std::vector<unsigned long long> h((size_t)std::max(k, 1) * n * 16);
		if (k) HIP_TRY(hipMemcpyAsync(h.data(), b->d_qcnt, (size_t)k * n * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream()));
		HIP_TRY(hipStreamSynchronize(b->stream()));
		for (size_t i = 0; i < (size_t)k * n; i++) {
			unsigned long long u = 0, e = 0;
			for (int w = 0; w < 8; w++) { u += h[i * 16 + w]; e += h[i * 16 + 8 + w]; }
			up[i] = u;
			if (bond_equal) bond_equal[i] = (int64_t)e;
		}
		*ncounts = k;
		return ISING_OK;
	}
	std::vector<int64_t> spare;
	long long got = 0;
	for (int it = first_it; it <= last;) {
		const long long next = std::min<long long>(last, ((long long)it + every - 1) / every * every);
		if (int rc = ising_batch_sweep(b, it, (int)(next - it + 1))) return rc;
		it = (int)next + 1;
		if (next % every == 0) if (int rc = ising_batch_measure_enqueue(b)) return rc;
		if (b->meas_pending == ising_batch::MEAS_CAP || (it > last && b->meas_pending)) {
			int k = 0;
			if (!bond_equal) spare.resize((size_t)b->meas_pending * n);
			if (int rc = ising_batch_measure_fetch(b, up + got * n, bond_equal ? bond_equal + got * n : spare.data(), b->meas_pending, &k)) return rc;
			got += k;
			*ncounts = (int)got;
		}
	}
	return ISING_OK;
}

int ising_batch_quad_info(ising_batch *b, int *tile_row_groups, int *sweeps_per_pass, int *waves) {
	if (!b) return fail(ISING_E_ARG, "null batch");
	if (tile_row_groups) *tile_row_groups = b->quad ? b->qC : 0;
	if (sweeps_per_pass) *sweeps_per_pass = b->quad ? b->qT : 0;
	if (waves) *waves = b->quad ? b->qwaves : 0;
	return ISING_OK;
}

} // extern "C"
