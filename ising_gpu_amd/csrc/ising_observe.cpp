// ising_observe.cpp -- observables of the C-ABI (include/ising_hip.h): countSpins (optimized/main.cu:831-868), the bond sum
// (energy, build-side), asynchronous measurements, and the ballot layout's dense-order image that the geometry-dependent ones
// borrow.  Host side only.
#include "ising_ctx.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using ising_host::bind;
using ising_host::fail;

// ballot layout: the dense-order image is allocated by the first call that needs one
int ising_host::ballot_tmp(ising_ctx *c) {
	if (!c->d_tmp) HIP_TRY(hipMalloc((void **)&c->d_tmp, c->tmp_words() * sizeof(uint64_t)));
	return ISING_OK;
}

void ising_host::ballot_tmp_release(ising_ctx *c) {
	if (!c->d_tmp) return;
	(void)hipStreamSynchronize(c->stream);
	(void)hipFree(c->d_tmp);
	c->d_tmp = nullptr;
}

// The observables read the ballot words as they are where the geometry is the plain one (no sub-lattices): no dense-order
// image, no second copy of the slab -- a slab that fills the device can still be measured.
bool ising_host::ballot_native_observables(const ising_ctx *c) { return c->ballot && !c->cfg.XSL; }

int ising_host::ballot_measure_into_acc(ising_ctx *c) {
	// (each on its own: a slab that nearly fills the device may get the first allocation and not the second)
	if (!c->d_self) HIP_TRY(hipMalloc((void **)&c->d_self, sizeof(ising::ReplicaParams)));
	if (!c->d_mslots) {
		HIP_TRY(hipMalloc((void **)&c->d_mslots, (size_t)ising::BALLOT_MEASURE_SLOTS * 8 * sizeof(unsigned long long)));
		HIP_TRY(hipMemsetAsync(c->d_mslots, 0, (size_t)ising::BALLOT_MEASURE_SLOTS * 8 * sizeof(unsigned long long), c->stream));
	}
	if (c->self_lat[0] != c->lat(ISING_BLACK) || c->self_lat[1] != c->lat(ISING_WHITE)) {
		ising::ReplicaParams r{};
		r.lat[0] = c->lat(ISING_BLACK);
		r.lat[1] = c->lat(ISING_WHITE);
		HIP_TRY(hipMemcpyAsync(c->d_self, &r, sizeof(r), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream)); // (`r` is on the stack; once per slab)
		c->self_lat[0] = r.lat[0];
		c->self_lat[1] = r.lat[1];
	}
	HIP_TRY(ising::launch_ballot_measure(c->d_self, 1, c->gx, c->cfg.Y, c->d_mslots, c->stream));
	HIP_TRY(ising::launch_measure_fold(c->d_mslots, c->d_acc, c->stream));
	return ISING_OK;
}

// ballot layout: convert rows [row_lo, row_hi) of `color` (rows -1 and Y are the halo rows) between d_lat and d_tmp
int ising_host::ballot_rows(ising_ctx *c, int color, long long row_lo, long long row_hi, bool to_dense) {
	if (int rc = ballot_tmp(c)) return rc;
	uint64_t *lat = c->lat(color) + row_lo * c->lld, *tmp = c->tmp(color) + row_lo * c->lld_dense;
	if (to_dense) HIP_TRY(ising::launch_ballot_to_dense(lat, reinterpret_cast<uint32_t *>(tmp), c->gx, row_hi - row_lo, c->stream));
	else HIP_TRY(ising::launch_dense_to_ballot(reinterpret_cast<const uint32_t *>(tmp), lat, c->gx, row_hi - row_lo, c->stream));
	return ISING_OK;
}

// ballot layout: refresh the dense-order image (both colours, halo rows included)
int ising_host::ballot_image(ising_ctx *c) {
	for (int color = 0; color < 2; color++)
		if (int rc = ballot_rows(c, color, -1, (long long)c->cfg.Y + 1, true)) return rc;
	return ISING_OK;
}

// ballot -> dense for good (a temperature without integer thresholds was requested): the slab keeps its buffer
int ising_host::ballot_leave(ising_ctx *c) {
	if (!c->ballot) return ISING_OK;
	if (int rc = ising_host::ballot_image(c)) return rc;
	// the dense-order image becomes the slab (same buffer; when X is not a multiple of 8192 the rows get shorter and the
	// colour arrays move up: pointers handed out by ising_halo_ptrs / ising_device_ptr before are void)
	HIP_TRY(hipMemcpyAsync(c->d_lat, c->d_tmp, c->tmp_words() * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
	c->lld = c->lld_dense;
	c->color_words = (size_t)c->cfg.Y * c->lld;
	if (c->ham_form == 2) {
		for (int w = 0; w < 2; w++) HIP_TRY(ising::launch_ham_ballot_to_planes(c->ham(w), c->gx, c->cfg.Y, c->stream));
		c->ham_form = 1;
	}
	c->ballot = false;
	ising_host::ballot_tmp_release(c);
	return ISING_OK;
}


extern "C" {

int ising_count(ising_ctx *c, uint64_t *up, uint64_t *down) {
	if (!c || !up || !down) return fail(ISING_E_ARG, "null argument");
	if (int rc = bind(c)) return rc;
	HIP_TRY(hipMemsetAsync(c->d_acc, 0, 2 * sizeof(unsigned long long), c->stream));
	HIP_TRY(ising::launch_popcount(c->lat(ISING_BLACK), c->color_words, c->d_acc, c->stream));
	HIP_TRY(ising::launch_popcount(c->lat(ISING_WHITE), c->color_words, c->d_acc, c->stream));
	unsigned long long h = 0;
	HIP_TRY(hipMemcpyAsync(&h, c->d_acc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
	if (int rc = ising_host::sync_checked(c)) return rc;
	*up = h;
	*down = (uint64_t)c->cfg.X * (uint64_t)c->cfg.Y - h; // SPIN_X_WORD - popc per word, optimized/main.cu:722-723
	return ISING_OK;
}

int ising_bond_equal(ising_ctx *c, int64_t *A) {
	if (!c || !A) return fail(ISING_E_ARG, "null argument");
	if (int rc = bind(c)) return rc;
	if (int rc = ising_host::halo_ready(c, ISING_WHITE)) return rc; // black sites of rows 0 / Y-1 read the white halo rows
	if (ising_host::ballot_native_observables(c)) {
		if (int rc = ising_host::ballot_measure_into_acc(c)) return rc;
		unsigned long long h2 = 0;
		HIP_TRY(hipMemcpyAsync(&h2, c->d_acc + 1, sizeof(h2), hipMemcpyDeviceToHost, c->stream));
		if (int rc = ising_host::sync_checked(c)) return rc;
		*A = (int64_t)h2;
		return ISING_OK;
	}
	if (c->ballot) if (int rc = ising_host::ballot_image(c)) return rc;
	ising::BondParams p{};
	p.black = c->ballot ? c->tmp(ISING_BLACK) : c->lat(ISING_BLACK);
	p.white = c->ballot ? c->tmp(ISING_WHITE) : c->lat(ISING_WHITE);
	p.gx = c->gx;
	p.Y = c->cfg.Y;
	p.row_base = (uint32_t)c->cfg.slab * (uint32_t)c->cfg.Y;
	p.slV = c->cfg.XSL ? c->cfg.XSL / 64 : c->gx * 32;
	p.slY = c->cfg.XSL ? c->cfg.YSL : 0;
	p.acc = c->d_acc + 1;
	HIP_TRY(hipMemsetAsync(c->d_acc + 1, 0, sizeof(unsigned long long), c->stream));
	HIP_TRY(c->dense ? ising::launch_dense_bond_equal(p, c->stream) : ising::launch_bond_equal(p, c->stream));
	unsigned long long h = 0;
	HIP_TRY(hipMemcpyAsync(&h, c->d_acc + 1, sizeof(h), hipMemcpyDeviceToHost, c->stream));
	if (int rc = ising_host::sync_checked(c)) return rc;
	*A = (int64_t)h;
	ising_host::ballot_tmp_release(c);
	return ISING_OK;
}

// Asynchronous measurements: count + bond sum of the state the stream holds at this point, into a pinned host array the
// context owns; nothing waits until ising_measure_fetch.  A series of (sweeps, measurement) pairs then runs without a
// single host round trip in between (cuIsing --tsweep: 100 measurements per temperature point).
int ising_measure_enqueue(ising_ctx *c) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (int rc = bind(c)) return rc;
	if (!c->h_meas) HIP_TRY(hipHostMalloc((void **)&c->h_meas, (size_t)ising_ctx::MEAS_CAP * 2 * sizeof(unsigned long long), hipHostMallocDefault));
	if (c->meas_pending >= ising_ctx::MEAS_CAP) return fail(ISING_E_STATE, "%d measurements pending: ising_measure_fetch first", c->meas_pending);
	if (int rc = ising_host::halo_ready(c, ISING_WHITE)) return rc;
	if (ising_host::ballot_native_observables(c)) { // two launches on the slab's own words
		if (int rc = ising_host::ballot_measure_into_acc(c)) return rc;
		HIP_TRY(hipMemcpyAsync(c->h_meas + 2 * (size_t)c->meas_pending, c->d_acc, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
		c->meas_pending++;
		return ISING_OK;
	}
	HIP_TRY(hipMemsetAsync(c->d_acc, 0, 2 * sizeof(unsigned long long), c->stream));
	HIP_TRY(ising::launch_popcount(c->lat(ISING_BLACK), c->color_words, c->d_acc, c->stream));
	HIP_TRY(ising::launch_popcount(c->lat(ISING_WHITE), c->color_words, c->d_acc, c->stream));
	if (c->ballot) if (int rc = ising_host::ballot_image(c)) return rc;
	ising::BondParams p{};
	p.black = c->ballot ? c->tmp(ISING_BLACK) : c->lat(ISING_BLACK);
	p.white = c->ballot ? c->tmp(ISING_WHITE) : c->lat(ISING_WHITE);
	p.gx = c->gx;
	p.Y = c->cfg.Y;
	p.row_base = (uint32_t)c->cfg.slab * (uint32_t)c->cfg.Y;
	p.slV = c->cfg.XSL ? c->cfg.XSL / 64 : c->gx * 32;
	p.slY = c->cfg.XSL ? c->cfg.YSL : 0;
	p.acc = c->d_acc + 1;
	HIP_TRY(c->dense ? ising::launch_dense_bond_equal(p, c->stream) : ising::launch_bond_equal(p, c->stream));
	HIP_TRY(hipMemcpyAsync(c->h_meas + 2 * (size_t)c->meas_pending, c->d_acc, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
	c->meas_pending++;
	return ISING_OK;
}

int ising_measure_fetch(ising_ctx *c, uint64_t *up, int64_t *bond_equal, int max_n, int *n) {
	if (!c || !up || !bond_equal || !n || max_n < 0) return fail(ISING_E_ARG, "bad argument");
	if (int rc = bind(c)) return rc;
	if (c->meas_pending > max_n) return fail(ISING_E_ARG, "%d measurements pending, room for %d", c->meas_pending, max_n);
	if (int rc = ising_host::sync_checked(c)) { c->meas_pending = 0; return rc; }
	for (int i = 0; i < c->meas_pending; i++) {
		up[i] = c->h_meas[2 * i];
		bond_equal[i] = (int64_t)c->h_meas[2 * i + 1];
	}
	*n = c->meas_pending;
	c->meas_pending = 0;
	return ISING_OK;
}

int ising_layout(ising_ctx *c, int *layout) {
	if (!c || !layout) return fail(ISING_E_ARG, "null argument");
	*layout = c->ballot ? ISING_LAYOUT_BALLOT : (c->dense ? ISING_LAYOUT_DENSE : ISING_LAYOUT_NIBBLE);
	return ISING_OK;
}

int ising_device_ptr(ising_ctx *c, int color, void **ptr, size_t *bytes) {
	if (!c || !ptr) return fail(ISING_E_ARG, "null argument");
	if (color != ISING_BLACK && color != ISING_WHITE) return fail(ISING_E_ARG, "bad colour %d", color);
	*ptr = c->lat(color);
	if (bytes) *bytes = c->color_words * sizeof(uint64_t);
	return ISING_OK;
}

} // extern "C"
