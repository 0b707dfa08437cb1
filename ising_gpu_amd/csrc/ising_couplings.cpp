// ising_couplings.cpp -- the -J coupling arrays of the C-ABI (include/ising_hip.h): hamiltInitB_k / hamiltInitW_k
// (optimized/main.cu:153-331, launches :1729-1742), read / write / swap in the reference's nibble form whatever form the
// device layout keeps.  Host side only.
#include "ising_ctx.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using ising_host::bind;
using ising_host::fail;

namespace {

// Inverse of ham_planes_k (ising_dense.hip), in place: four coupling bit-planes per vector -> 32 nibbles.
void planes_to_nibbles(uint64_t *vecs, size_t nvec) {
	for (size_t v = 0; v < nvec; v++) {
		uint32_t pl[4];
		memcpy(pl, vecs + 2 * v, sizeof(pl));
		uint64_t w[2] = {0, 0};
		for (int s = 0; s < 32; s++) {
			const uint64_t nib = ((pl[0] >> s) & 1u) | (((pl[1] >> s) & 1u) << 1) | (((pl[2] >> s) & 1u) << 2) | (((pl[3] >> s) & 1u) << 3);
			w[s >> 4] |= nib << (4 * (s & 15));
		}
		vecs[2 * v] = w[0];
		vecs[2 * v + 1] = w[1];
	}
}

// Host mirror of ham_ballot_to_planes_k (ising_ballot.hip), in place: groups of 4 planes x 64 ballot-order words ->
// 128 vectors of four 32-bit planes.
void ballot_planes_to_planes(uint64_t *ham, size_t ngroups) {
	std::vector<uint64_t> in(256);
	for (size_t g = 0; g < ngroups; g++) {
		uint64_t *grp = ham + g * 256;
		memcpy(in.data(), grp, 256 * sizeof(uint64_t));
		uint32_t *out = reinterpret_cast<uint32_t *>(grp);
		for (int v = 0; v < 128; v++) {
			const int l = ((v >> 5) << 4) | (v & 15), j = (v >> 4) & 1;
			uint32_t pw[4] = {0, 0, 0, 0};
			for (int s = 0; s < 32; s++) {
				const int m = (s & 15) >> 1, q = ((s & 1) << 1) | (s >> 4);
				const int p = 32 * j + 4 * m + q;
				for (int pl = 0; pl < 4; pl++) pw[pl] |= (uint32_t)((in[64 * pl + p] >> l) & 1ull) << s;
			}
			memcpy(out + 4 * v, pw, sizeof(pw));
		}
	}
}


} // namespace

extern "C" {

// ------------------------------------------------------------------------------------------------ couplings (-J)
int ising_init_couplings_black(ising_ctx *c) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (!c->cfg.use_J) return fail(ISING_E_STATE, "couplings are not enabled (use_J)");
	if (int rc = bind(c)) return rc;
	const float prob = fminf(fmaxf(0.0f, c->cfg.J_prob), 1.0f);  // optimized/main.cu:1370
	const uint64_t seed = c->cfg.seed + 1;                        // "just use a different seed", :1734
	ising::HamInitParams p{};
	p.hamB = c->ham(0);
	p.seed_lo = (uint32_t)seed;
	p.seed_hi = (uint32_t)(seed >> 32);
	p.gx = c->gx;
	p.Y = c->cfg.Y;
	p.row_base = (uint32_t)c->cfg.slab * (uint32_t)c->cfg.Y;
	p.wrap = c->wrap;
	if (c->ham_ghost > 1) { // ring slab with ghost rows: rows [-hg, Y + hg), each with the draws of its global row
		const uint32_t total = (uint32_t)c->cfg.nslabs * (uint32_t)c->cfg.Y, hg = (uint32_t)c->ham_ghost;
		p.hamB = c->ham(0) - (size_t)hg * c->lld_packed;
		p.Y = c->cfg.Y + 2 * (int)hg;
		p.row_base = (p.row_base + total - hg % total) % total;
		p.total_rows = total;
	}
	const uint64_t thr = ising_host::draw_prefix(prob, false); // curand_uniform(x) < tgtProb, :193
	if (thr >= (1ull << 32)) return fail(ISING_E_ARG, "J probability %g sets every bit", (double)prob); // unreachable: u <= 1 and prob <= 1 gives at most 2^32 - 1... see below
	p.thr = (uint32_t)thr;
	HIP_TRY(ising::launch_ham_init_black(p, c->stream));
	c->ham_form = 0; // nibble form until the white couplings have been assembled from it
	return ISING_OK;
}

int ising_init_couplings_white(ising_ctx *c) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (!c->cfg.use_J) return fail(ISING_E_STATE, "couplings are not enabled (use_J)");
	if (int rc = bind(c)) return rc;
	ising::HamWhiteParams p{};
	p.hamB = c->ham(0);
	p.hamW = c->ham(1);
	if (c->ham_form) return fail(ISING_E_STATE, "ising_init_couplings_white needs a fresh ising_init_couplings_black");
	p.lld = c->lld_packed;
	p.Y = c->cfg.Y;
	p.row_base = (uint32_t)c->cfg.slab * (uint32_t)c->cfg.Y;
	p.slW = c->cfg.XSL ? c->cfg.XSL / 32 : c->lld_packed;
	p.slY = c->cfg.XSL ? c->cfg.YSL : 0;
	p.wrap = c->wrap;
	const int g = c->ham_ghost - 1; // coupling rows beyond the slab's own that the update reads (ghost rows of a ring slab)
	if (g > 0) {                    // white rows [-g, Y + g) from black rows [-g - 1, Y + g + 1); only the row parity counts here
		p.hamB -= (size_t)g * c->lld_packed;
		p.hamW -= (size_t)g * c->lld_packed;
		p.Y += 2 * g;
		p.row_base += (uint32_t)(g & 1);
	}
	HIP_TRY(ising::launch_ham_init_white(p, c->stream));
	if (c->ballot) {
		// the ballot update reads four planes of ballot-order coupling words per row and wave column
		for (int w = 0; w < 2; w++) HIP_TRY(ising::launch_ham_to_ballot(c->ham(w) - (size_t)g * c->lld_packed, c->gx, c->cfg.Y + 2 * g, c->stream));
		c->ham_form = 2;
	} else if (c->dense) {
		// the dense update reads four coupling bit-planes per 32-site word: transpose both arrays in place
		for (int w = 0; w < 2; w++) HIP_TRY(ising::launch_ham_planes(c->ham(w), c->ham_words() / 2, c->stream));
		c->ham_form = 1;
	}
	return ISING_OK;
}

int ising_init_couplings(ising_ctx *c) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (!c->wrap && !c->cfg.XSL) return fail(ISING_E_STATE, "ising_init_couplings needs nslabs == 1; use the _black/_white pair around a halo exchange");
	if (int rc = ising_init_couplings_black(c)) return rc;
	return ising_init_couplings_white(c);
}

int ising_read_couplings(ising_ctx *c, int which, int64_t row0, int64_t nrows, uint64_t *dst_host) {
	if (!c || !dst_host) return fail(ISING_E_ARG, "null argument");
	if (which != ISING_BLACK && which != ISING_WHITE) return fail(ISING_E_ARG, "bad coupling array %d", which);
	if (row0 < 0 || nrows < 0 || row0 + nrows > c->cfg.Y) return fail(ISING_E_ARG, "rows [%lld,%lld) outside slab of %d rows", (long long)row0, (long long)(row0 + nrows), c->cfg.Y);
	if (!c->cfg.use_J) return fail(ISING_E_STATE, "couplings are not enabled (use_J)");
	if (int rc = bind(c)) return rc;
	const size_t nw = (size_t)nrows * c->lld_packed;
	HIP_TRY(hipMemcpyAsync(dst_host, c->ham(which) + (size_t)row0 * c->lld_packed, nw * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (c->ham_form == 2) ballot_planes_to_planes(dst_host, nw / 256);
	if (c->ham_form) planes_to_nibbles(dst_host, nw / 2);
	return ISING_OK;
}

int ising_swap_couplings(ising_ctx *c) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (!c->cfg.use_J) return fail(ISING_E_STATE, "couplings are not enabled (use_J)");
	if (int rc = bind(c)) return rc;
	const size_t per_array = c->ham_alloc_words() / 2; // rows [-ghost, Y + ghost) of one array
	HIP_TRY(ising::launch_swap_vectors(c->d_ham, c->d_ham + per_array, per_array / 2, c->stream));
	return ISING_OK;
}

int ising_write_couplings(ising_ctx *c, int which, const uint64_t *src_host) {
	if (!c || !src_host) return fail(ISING_E_ARG, "null argument");
	if (which != ISING_BLACK && which != ISING_WHITE) return fail(ISING_E_ARG, "bad coupling array %d", which);
	if (!c->cfg.use_J) return fail(ISING_E_STATE, "couplings are not enabled (use_J)");
	if (!c->wrap || c->ham_ghost != 1) return fail(ISING_E_STATE, "ising_write_couplings needs a lattice that wraps in place (nslabs == 1 without ring halo rows)");
	if (int rc = bind(c)) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream)); // (launches that still read the array)
	HIP_TRY(hipMemcpy(c->ham(which), src_host, c->ham_words() * sizeof(uint64_t), hipMemcpyHostToDevice));
	// into the form the update kernels of this layout read (ising_init_couplings_white): the array just written -- and, where the
	// arrays were still as generated (no white initialisation yet, or nothing at all: zeros are zeros in every form), the other one
	const int want = c->ham_form ? c->ham_form : (c->ballot ? 2 : (c->dense ? 1 : 0));
	for (int w = 0; w < 2; w++) {
		if (w != which && c->ham_form == want) continue;
		if (want == 2) HIP_TRY(ising::launch_ham_to_ballot(c->ham(w), c->gx, c->cfg.Y, c->stream));
		if (want == 1) HIP_TRY(ising::launch_ham_planes(c->ham(w), c->ham_words() / 2, c->stream));
	}
	c->ham_form = want;
	HIP_TRY(hipStreamSynchronize(c->stream));
	return ISING_OK;
}

} // extern "C"
