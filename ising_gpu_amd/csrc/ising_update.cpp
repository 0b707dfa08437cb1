// ising_update.cpp -- the update side of the C-ABI (include/ising_hip.h): one colour half-sweep over row ranges, the fused
// launches that carry many of them, the sweeps of a slab on its own, and the halo / ghost-row surface for a caller's own
// transport.  Replaces the launch sites of the reference's hot loop, optimized/main.cu:1763-1805.  Host side only: the
// arithmetic on the lattice is in ising_ballot.hip / ising_dense.hip / ising_kernels.hip.
#include "ising_ctx.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using ising_host::bind;
using ising_host::fail;
using ising_host::SLOTCTL_TICKET_BYTES;

extern "C" {

// launches update_k over up to two row ranges
// `nlevels` > 1 (ballot layout only): one fused launch of that many colour half-sweeps over the whole slab, starting with
// `color` at iteration `it`
// `stop` (optional): an event that fires when the launch is done
static int launch_ranges(ising_ctx *c, int it, int color, int lo0, int hi0, int lo1, int hi1, int nlevels = 1, hipEvent_t stop = nullptr) {
	// one-shot requests of the ring schedules for THIS launch (taken here, so that an early return cannot leave them set)
	const bool edge_scratch = c->edge_scratch_next;
	const bool overlap = c->overlap_next;
	hipEvent_t start = c->launch_start_next;
	const int cnt_first = c->cnt_first_next, cnt_every = c->cnt_every_next;
	const int cnt_slot0 = c->cnt_slot0_next;
	const bool cnt_bonds = c->cnt_bonds_next;
	const bool split_asked = c->split_next;
	const int epoch_sh = c->epoch_sh_next;
	c->split_next = false;
	c->epoch_sh_next = 0;
	c->cnt_every_next = 0;
	c->cnt_bonds_next = false;
	if (!stop) stop = c->launch_stop_next;
	c->edge_scratch_next = false;
	c->overlap_next = false;
	c->launch_start_next = c->launch_stop_next = nullptr;
	if (color != ISING_BLACK && color != ISING_WHITE) return fail(ISING_E_ARG, "bad colour %d", color);
	if (it < 0) return fail(ISING_E_ARG, "negative iteration %d", it);
	int mode = c->cfg.kernel == ISING_KERNEL_GENERIC ? 1 : 0; // AUTO, FAST -> 0
	if (mode != 1 && !c->fast_ok) {
		if (c->cfg.kernel != ISING_KERNEL_AUTO) return fail(ISING_E_STATE, "temperature %g does not admit the integer-threshold kernels", (double)c->cfg.temp);
		mode = 1;
	}
	if (int rc = bind(c)) return rc;
	if (c->ballot && mode == 1) if (int rc = ising_host::ballot_leave(c)) return rc; // no integer thresholds at this temperature
	c->ghost_depth[color] = 0; // (the neighbours' copies of this slab's rows are stale from here on -- and theirs here, by symmetry)
	if (nlevels > 1) c->ghost_depth[1 - color] = 0;
	const int other = 1 - color;
	ising::UpdateParams p{};
	p.dst = c->lat(color);
	p.src = c->lat(other);
	// periodic wrap of loadTile (optimized/main.cu:414,:422) through mirrored halo rows: the launch that writes an edge row
	// also writes its mirror -- this slab's own halo rows, or (ring on one device, ising_ring.cpp) the neighbours'
	const size_t rowb = (size_t)c->lld * sizeof(uint64_t);
	if (c->wrap) {
		p.wrap = 1;
		p.mir0_bytes = (long long)c->cfg.Y * (long long)rowb;
		p.mirL_bytes = -(long long)c->cfg.Y * (long long)rowb;
	} else if (c->store_ring && c->ring_prev && c->ring_next && !c->cfg.XSL) {
		const char *row0 = reinterpret_cast<const char *>(c->lat(color)), *rowL = row0 + (size_t)(c->cfg.Y - 1) * rowb;
		p.wrap = 1;
		p.mir0_bytes = reinterpret_cast<const char *>(c->ring_prev->lat(color) + c->ring_prev->color_words) - row0;
		p.mirL_bytes = reinterpret_cast<const char *>(c->ring_next->lat(color)) - (long long)rowb - rowL;
	}
	p.seed_lo = (uint32_t)c->cfg.seed;
	p.seed_hi = (uint32_t)(c->cfg.seed >> 32);
	p.it = (uint32_t)it;
	p.color = (uint32_t)color;
	p.gx = c->gx;
	p.Y = c->cfg.Y;
	p.row_base = (uint32_t)c->cfg.slab * (uint32_t)c->cfg.Y;
	p.slV = c->cfg.XSL ? c->cfg.XSL / 64 : c->gx * 32; // (XSL/2)/SPIN_X_WORD/2, optimized/main.cu:1771
	p.slY = c->cfg.XSL ? c->cfg.YSL : 0;
	// the split form of a fused launch (ising_ballot.hip: ballot_split_k; a launch of several levels covers the whole slab, or a ring slab with its ghost rows): strips of its own
	const bool split = c->ballot && c->split && nlevels > 1 && hi1 == lo1 && color == ISING_BLACK && c->d_split_masks && c->d_split_ctl && (c->split_always || split_asked);
	const int HL = split ? c->H_split : c->H;
	p.H = HL;
	p.row_lo[0] = lo0; p.row_hi[0] = hi0;
	p.row_lo[1] = lo1; p.row_hi[1] = hi1;
	const int ugx = c->ballot ? 4 * c->nwc() : c->gx; // column groups per strip as the kernel counts them (ballot: 4 per wave column)
	// tail strips (ballot layout, plain full-slab launch): the last rows of the slab in strips of H2 rows
	int H2 = 0;
	if (c->ballot && nlevels == 1 && c->tail_rows > 0 && hi1 == lo1 && hi0 - lo0 >= 4 * c->tail_rows) {
		H2 = c->tail_h;
		lo1 = hi0 - c->tail_rows;
		hi1 = hi0;
		hi0 = lo1;
		p.row_hi[0] = hi0; p.row_lo[1] = lo1; p.row_hi[1] = hi1;
	}
	p.H2 = H2;
	p.nreal0 = ugx * ((hi0 - lo0 + HL - 1) / HL);
	p.nunits0 = (c->ballot && H2) ? (p.nreal0 + 15) / 16 * 16 : p.nreal0;
	p.nunits = p.nunits0 + ugx * ((hi1 - lo1 + (H2 ? H2 : HL) - 1) / (H2 ? H2 : HL));
	p.n3 = (uint32_t)c->thr[3];
	p.n4 = (uint32_t)c->thr[4];
	memcpy(p.tab, c->tab, sizeof(p.tab));
	// the reference hands hamW to the BLACK update and hamB to the WHITE one (optimized/main.cu:1774, :1795)
	p.jdst = c->cfg.use_J ? c->ham(other) : nullptr;
	p.scratch = (edge_scratch && c->d_scratch_edge) ? c->d_scratch_edge : c->d_scratch;
	if (c->ballot) {
		p.ticket = reinterpret_cast<unsigned long long *>(c->d_slotctl);
		p.nlevels = nlevels;
		p.cus = c->cus;
		if (nlevels > 1) {
			p.grid_cap = c->guard.grid_override ? 0 : c->pol.fused_wgs; // (a shape the guard chose replaces ISING_FUSED_WGS)
			p.abort_flag = c->h_abort;
			p.abort_polls = c->pol.abort_polls;
			for (int k = 0; k < 4; k++) p.ticket_base2[k] = c->ticket_base2[k]; // (the counters are never reset, ising_ballot.hip)
			p.tickets2 = c->fused_tickets2;
			const size_t done_words = c->ctl_strips + 2 * (size_t)c->ghost_rows + 2; // (per form)
			uint32_t *const done = c->d_slotctl + SLOTCTL_TICKET_BYTES / 4 + (split ? done_words : 0);
			uint32_t &done_base = split ? c->split_done_base : c->done_base;
			if (done_base > (1u << 30)) { // keep the monotone completion counters far from wrapping
				HIP_TRY(hipMemsetAsync(done, 0, done_words * sizeof(uint32_t), c->stream));
				done_base = 0;
			}
			p.lat[0] = c->lat(ISING_BLACK);
			p.lat[1] = c->lat(ISING_WHITE);
			p.jham[0] = c->cfg.use_J ? c->ham(1) : nullptr;
			p.jham[1] = c->cfg.use_J ? c->ham(0) : nullptr;
			p.done = done;
			p.wg_per_cu = split ? c->split_wg_per_cu : c->fused_wg_per_cu;
			p.wait_late = c->fused_wait_late ? (c->pol.fused_wait_late == 1 ? 1 : 2) : 0; // (default 2: behind the second draw phase)
			if (cnt_every > 0 && 2 * cnt_first < nlevels) {
				p.cnt_acc = c->d_cnt; p.cnt_first = cnt_first; p.cnt_every = cnt_every; p.cnt_slot0 = cnt_slot0; p.cnt_bonds = cnt_bonds ? 1 : 0;
				p.cnt_magic = (uint32_t)((0x100000000ull + (unsigned long long)cnt_every - 1) / (unsigned long long)cnt_every);
			}
			if (c->clk_on && c->d_clk) p.clk_out = c->d_clk;
			p.nt_stream = c->fused_nt;
			p.done_base = done_base;
			if (lo0 < 0 || hi0 > c->cfg.Y) { // ghost rows are rows of the neighbouring slabs
				p.total_rows = c->cfg.nslabs * c->cfg.Y;
				p.trapezoid = c->pol.trapezoid ? 1 : 0;
				if (overlap && c->d_edge) {
					// the exchange touches the first / last G rows (read by the sends) and the ghost rows (written by the receives)
					const int G = c->ghost();
					p.edge_lo = G;
					p.edge_hi = c->cfg.Y - G;
					p.edge_go = c->d_edge + 16;
					p.edge_go_need = c->edge_go_epoch;
					p.edge_done = c->d_edge;
					unsigned strips = 0; // strips of this launch that touch such a row
					for (int r0 = lo0; r0 < hi0; r0 += HL) if (r0 < p.edge_lo || std::min(r0 + HL, hi0) > p.edge_hi) strips++;
					// what the launch adds to *edge_done per exchange epoch (the caller keeps edge_done_target: one wait per epoch on the comm stream)
					c->edge_units_per_epoch = strips * (uint32_t)c->nwc();
					// several epochs in this launch (ising_ring.cpp: sweep_deep_overlapped): epochs of 2^epoch_sh levels = the ghost rows' depth
					if (epoch_sh > 0 && nlevels > (1 << epoch_sh)) p.epoch_sh = epoch_sh;
				}
			}
		}
		int grid = 0;
		if (split) { // (its ticket words and slot counts start from zero)
			HIP_TRY(hipMemsetAsync(c->d_split_ctl, 0, c->split_ctl_bytes, c->stream));
			p.sp_ctr = c->d_split_ctl;
			p.sp_flags = reinterpret_cast<uint32_t *>(c->d_split_ctl + 8 * 16);
			p.sp_masks = c->d_split_masks;
			p.sp_ring_sh = c->split_ring_sh;
			p.sp_lead = c->split_lead;
			p.sp_cap = c->split_cap;
		}
		if (const hipError_t le = split ? ising::launch_ballot_split(p, c->stream, &grid, stop, start) : ising::launch_ballot_update(p, c->stream, &grid, stop, start); le != hipSuccess) {
			// nothing ran: tickets and counters are where the launches before left them, but to be safe they start over
			if (nlevels > 1) { __atomic_store_n(c->h_abort, 1u, __ATOMIC_RELEASE); (void)ising_host::check_abort(c); }
			return fail(ISING_E_HIP, "kernel launch failed: %s", hipGetErrorString(le));
		}
		if (nlevels > 1) {
			c->last_launch_split = split;
			(split ? c->split_done_base : c->done_base) += (uint32_t)nlevels * (uint32_t)c->nwc();
			// where the launch leaves the counter(s): its units, and every workgroup drew one ticket too many
			const unsigned long long total = (unsigned long long)p.nwg * (unsigned long long)nlevels;
			if (split) {
				// (its tickets are the split form's own words)
			} else if (p.tickets2 > 1) { // units and workgroups of class k = those numbered k mod K
				const unsigned long long K = (unsigned long long)p.tickets2;
				for (unsigned long long k = 0; k < K; k++) c->ticket_base2[k] += (total + K - 1 - k) / K + ((unsigned long long)grid + K - 1 - k) / K;
			} else {
				c->ticket_base2[0] += total + (unsigned long long)grid;
			}
		}
		return ISING_OK;
	}
	if (c->dense) HIP_TRY(ising::launch_dense_update(p, mode, c->stream));
	else HIP_TRY(ising::launch_update(p, mode, c->stream));
	if (stop) HIP_TRY(hipEventRecord(stop, c->stream)); // (the other layouts' launchers take no event: a packet of its own)
	return ISING_OK;
}

int ising_update_color(ising_ctx *c, int it, int color, int row_lo, int row_hi) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (row_lo < 0 || row_hi > c->cfg.Y || row_lo > row_hi) return fail(ISING_E_ARG, "bad row range [%d,%d) of %d", row_lo, row_hi, c->cfg.Y);
	return launch_ranges(c, it, color, row_lo, row_hi, 0, 0);
}

int ising_update_edges(ising_ctx *c, int it, int color) {
	if (!c) return fail(ISING_E_ARG, "null context");
	return launch_ranges(c, it, color, 0, 1, c->cfg.Y - 1, c->cfg.Y);
}

} // extern "C"

int ising_host::update_edges_on(ising_ctx *c, int it, int color, hipStream_t s, hipEvent_t stop) {
	hipStream_t keep = c->stream; // (a context is driven by one host thread)
	c->stream = s;
	c->edge_scratch_next = s != keep; // on another stream than the slab's own: it may run next to an interior launch
	const int rc = launch_ranges(c, it, color, 0, 1, c->cfg.Y - 1, c->cfg.Y, 1, stop);
	c->stream = keep;
	return rc;
}

int ising_host::update_interior(ising_ctx *c, int it, int color, hipEvent_t stop) {
	return launch_ranges(c, it, color, 1, c->cfg.Y - 1, 0, 0, 1, stop);
}

// true when the ring sweeps this slab through its ghost rows right now (ising_ring.cpp: sweep_local takes the same decision)
static bool ghost_sweeps(const ising_ctx *c) {
	return !c->wrap && c->ballot && c->ghost() > 1 && !c->store_ring && !c->cfg.XSL && !ising_host::needs_generic(c);
}

// Ring slab with G > 1 ghost rows: `nlevels` colour half-sweeps (black first) in one fused launch over rows
// [-(G-1), Y+G-1).  The ghost rows are updated like the slab's own -- their draws are the ones the neighbours make --, and
// what is not valid in them any more (one row per level and side) never reaches a row that is.
// `epochs` > 1 (overlapped exchanges only, G a power of two): the launch carries that many exchange epochs of G levels each (the last may be shorter) -- the
// caller runs one exchange per epoch on the comm stream next to it (UpdateParams.epoch_sh).
int ising_host::update_deep(ising_ctx *c, int it, int nlevels, bool overlapped, int epochs, bool split) {
	const int G = c->ghost();
	if (epochs < 1 || (epochs > 1 && (!overlapped || (G & (G - 1)) != 0)))
		return fail(ISING_E_STATE, "deep launch of %d epochs: needs overlapped exchanges and ghost rows a power of two deep (%d)", epochs, G);
	if (G < 2 || nlevels > epochs * G || nlevels <= (epochs - 1) * G || nlevels < 2 || c->store_ring)
		return fail(ISING_E_STATE, "deep launch of %d levels in %d epoch(s) on a slab with %d ghost rows", nlevels, epochs, G);
	c->overlap_next = overlapped;
	c->split_next = split; // (the split form where the caller asks for it: a launch of several epochs on a slab whose table says so; ISING_SPLIT=1: always)
	c->epoch_sh_next = epochs > 1 ? __builtin_ctz((unsigned)G) : 0;
	return launch_ranges(c, it, ISING_BLACK, -(G - 1), c->cfg.Y + G - 1, 0, 0, nlevels);
}

// fused launches carry this slab's sweeps (ballot layout, integer thresholds; with sub-lattices: strips inside the blocks, no couplings)
static bool sweeps_fused(const ising_ctx *c) {
	if (!c->ballot || !c->fused || ising_host::needs_generic(c)) return false;
	if (c->cfg.XSL) return !c->cfg.use_J && (c->cfg.YSL % c->H) == 0;
	return c->wrap;
}

// The split form costs a launch ~0.14 ms more than the fused form (it ends on word units alone) and runs 1-6 % faster in between: it carries the calls that are
// long enough to earn that back twice over -- ~10 ms of sweeps, 2^35 flips (16384^2: 128 sweeps; break-even measured at 63, profiles/rocprof_r05_config2_split.txt).
// Shorter calls run the fused form at its own shape, as before round 5.  (ISING_SPLIT=1: every call.)
static bool split_pays(const ising_ctx *c, int nsweeps) {
	return c->split && !c->guard.no_split && (c->split_always || (long long)nsweeps * c->cfg.X * c->cfg.Y >= (1LL << 35));
}

extern "C" int ising_sweep(ising_ctx *c, int first_it, int nsweeps) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (!c->wrap) return fail(ISING_E_STATE, "ising_sweep needs a single slab without ring halo rows; drive slabs with ising_ring_sweep / ising_rank_sweep or ising_update_color + halo exchange");
	return ising_host::sweep_alone(c, first_it, nsweeps);
}

// (Lattices too small for fused launches -- under 1.5 * 2^24 spins: the dense layout, one launch per colour -- take ~6 us per launch:
// 4096^2 12.0 us per sweep, 2048^2 9.6 us.  Round 4 replayed the launches from a captured hipGraph, the iteration coming from device
// memory: bit-exact and NOT faster, 0.96 .. 1.02 x (profiles/sweep_graph_probe_r04.txt; the code is in the history at 9f35ffa) -- the
// host keeps up; what a launch costs is the kernel itself at two waves per SIMD plus the device-side boundary between two kernels.)
// Small lattices on the dense layout (ising_dense.hip: dense_tile_k): launches of several sweeps each, every workgroup on its own tile
// + halo, no exchange inside a launch.  A launch reads one buffer and writes the other, so a call issues an EVEN number of them and
// the spins are back in d_lat when it returns (a single sweep takes the two per-colour launches).
// (`fast_ok`: the tile kernel knows integer thresholds only -- KERNEL_FAST at a temperature without them falls through to launch_ranges, which returns ISING_E_STATE as the header says)
static bool sweeps_tiled(const ising_ctx *c, int nsweeps) {
	return c->tile_rows > 0 && nsweeps >= 2 && c->wrap && c->dense && !c->ballot && !c->cfg.use_J && !c->cfg.XSL && c->fast_ok && !ising_host::needs_generic(c);
}

// one tile launch of `ns` sweeps from buffer `from_second ? d_lat2 : d_lat` into the other one; `cnt`: see TileParams
static int launch_tiles(ising_ctx *c, int it, int ns, bool from_second, unsigned long long *cnt) {
	auto plane = [&](uint64_t *base, int color) { return reinterpret_cast<uint32_t *>(base + (c->lat(color) - c->d_lat)); };
	uint64_t *from = from_second ? c->d_lat2 : c->d_lat, *to = from_second ? c->d_lat : c->d_lat2;
	ising::TileParams p{};
	for (int color = 0; color < 2; color++) { p.src[color] = plane(from, color); p.dst[color] = plane(to, color); }
	p.seed_lo = (uint32_t)c->cfg.seed;
	p.seed_hi = (uint32_t)(c->cfg.seed >> 32);
	p.it = (uint32_t)it;
	p.n3 = (uint32_t)c->thr[3];
	p.n4 = (uint32_t)c->thr[4];
	p.ns = ns;
	p.gx = c->gx;
	p.Y = c->cfg.Y;
	p.TR = c->tile_rows;
	p.TWI = c->tile_words;
	p.xcd_rows = c->tile_xcd ? (c->gx * 32 / c->tile_words) * (c->cfg.Y / c->tile_rows) / 8 : 0;
	p.cnt = cnt;
	HIP_TRY(ising::launch_dense_tiles(p, c->tile_threads, c->stream));
	return ISING_OK;
}

static int sweep_tiles(ising_ctx *c, int first_it, int nsweeps) {
	if (int rc = bind(c)) return rc;
	if (!c->d_lat2) return fail(ISING_E_STATE, "tile launches without their second buffer (ising_create allocates it)");
	const int S = c->tile_sweeps;
	const int L = 2 * ((nsweeps + 2 * S - 1) / (2 * S)); // launches: even, none longer than S sweeps
	const int base = nsweeps / L, rem = nsweeps % L;
	int it = first_it;
	for (int l = 0; l < L; l++) {
		const int ns = base + (l < rem ? 1 : 0);
		if (int rc = launch_tiles(c, it, ns, (l & 1) != 0, nullptr)) return rc;
		it += ns;
	}
	return ISING_OK;
}

// ising_sweep_counted in tile launches: every print point is the end of a launch, whose workgroups count what they store.  The launches
// alternate between the two buffers; a call that ends in the second one copies it back (the lattices of this path are a few MB).
static int sweep_tiles_counted(ising_ctx *c, int first_it, int nsweeps, int every, uint64_t *ups, long long n, int *ncounts) {
	if (int rc = bind(c)) return rc;
	if (!c->d_lat2) return fail(ISING_E_STATE, "tile launches without their second buffer (ising_create allocates it)");
	if (c->tile_cnt_cap < (size_t)n) {
		if (c->d_tile_cnt) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_tile_cnt)); c->d_tile_cnt = nullptr; c->tile_cnt_cap = 0; }
		const size_t cap = std::max<size_t>(64, (size_t)n);
		HIP_TRY(hipMalloc((void **)&c->d_tile_cnt, cap * sizeof(unsigned long long)));
		c->tile_cnt_cap = cap;
	}
	if (n) HIP_TRY(hipMemsetAsync(c->d_tile_cnt, 0, (size_t)n * sizeof(unsigned long long), c->stream));
	const long long last = (long long)first_it + nsweeps - 1;
	const int S = c->tile_sweeps;
	bool second = false;
	int it = first_it, k = 0;
	while (it <= last) {
		const long long next = std::min<long long>(last, ((long long)it + every - 1) / every * every); // the next multiple of `every` from `it` on
		const int seg = (int)(next - it + 1), L = (seg + S - 1) / S;
		for (int l = 0; l < L; l++) {
			const int ns = seg / L + (l < seg % L ? 1 : 0);
			const bool measured = l == L - 1 && next % every == 0;
			if (int rc = launch_tiles(c, it, ns, second, measured ? c->d_tile_cnt + k : nullptr)) return rc;
			if (measured) k++;
			second = !second;
			it += ns;
		}
	}
	if (second) HIP_TRY(hipMemcpyAsync(c->d_lat, c->d_lat2, c->alloc_words() * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
	std::vector<unsigned long long> h((size_t)std::max(k, 1));
	if (k) HIP_TRY(hipMemcpyAsync(h.data(), c->d_tile_cnt, (size_t)k * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
	if (int rc = ising_host::sync_checked(c)) return rc;
	for (int q = 0; q < k; q++) ups[q] = h[q];
	*ncounts = k;
	return ISING_OK;
}

// The passes of `nsweeps` sweeps from iteration first_it on, T sweeps a pass at most: a segment between two print points (`every` > 0: the iterations that are
// multiples of it) is cut into passes of equal length, the one that ends on a print point carries its index.
std::vector<ising_host::QuadPass> ising_host::quad_passes(int first_it, int nsweeps, int every, int T, int *nmeas) {
	std::vector<QuadPass> passes;
	const long long last = (long long)first_it + nsweeps - 1;
	int k = 0;
	for (int it = first_it; it <= last;) {
		const long long next = every > 0 ? std::min<long long>(last, ((long long)it + every - 1) / every * every) : last; // the segment's last iteration
		const int seg = (int)(next - it + 1), L = (seg + T - 1) / T;
		for (int l = 0; l < L; l++) {
			const int ns = seg / L + (l < seg % L ? 1 : 0);
			const bool measured = every > 0 && l == L - 1 && next % every == 0;
			passes.push_back({it, ns, measured ? k++ : -1});
			it += ns;
		}
	}
	if (nmeas) *nmeas = k;
	return passes;
}

// Small lattices on the quad layout (ising_quad.hip).  A pass = up to T sweeps; launch k = the word pass k on the masks launch k - 1 drew + the draws of pass
// k + 1 (two mask buffers, one stream, no events: the launches of a call are k = -1 .. passes - 1, the first draws only, the last works on words only).
// The spins live in d_lat (dense layout) between calls: a call converts on its way in and out, so everything else the library does with a dense slab --
// counts, energy, dumps, a temperature change -- finds what it always found.
static bool sweeps_quad(const ising_ctx *c, int nsweeps) {
	return c->quad_C > 0 && nsweeps >= 2 && c->wrap && c->dense && !c->ballot && !c->cfg.use_J && !c->cfg.XSL && c->fast_ok && !ising_host::needs_generic(c);
}

bool ising_host::quad_ready(const ising_ctx *c) { return sweeps_quad(c, 2) && c->d_quad && c->d_qmasks; }

// `every` > 0: the up spins after every iteration that is a multiple of it are added to the eight words d_cnt[8 m ..] (zero on entry) -- `bonds`: to
// d_cnt[16 m ..], the bonds between equal spins at the same point to d_cnt[16 m + 8 ..] --; *nmeas = how many
static int sweep_quad(ising_ctx *c, int first_it, int nsweeps, int every, unsigned long long *d_cnt, int *nmeas, bool bonds = false) {
	if (int rc = bind(c)) return rc;
	if (!c->d_quad || !c->d_qmasks) return fail(ISING_E_STATE, "quad sweeps without their buffers (ising_create allocates them)");
	const int T = c->quad_T;
	int k = 0;
	const std::vector<ising_host::QuadPass> passes = ising_host::quad_passes(first_it, nsweeps, every, T, &k);
	if (nmeas) *nmeas = k;
	const int NRG = c->cfg.Y / 4, gx = c->gx;
	const size_t qw = c->quad_words(), NI = qw / 64;
	const size_t mask_words = (size_t)(2 * T) * NI * 128; // per buffer: the levels of one pass
	auto dense_plane = [&](int color) { return reinterpret_cast<uint32_t *>(c->lat(color)); };
	int cur = 0;
	for (int color = 0; color < 2; color++) HIP_TRY(ising::launch_dense_to_quad(dense_plane(color), c->d_quad + (size_t)color * qw, gx, NRG, c->stream));
	const int np = (int)passes.size();
	for (int q = -1; q < np; q++) {
		ising::QuadPassParams pp{};
		pp.w.gx = pp.d.gx = gx;
		pp.w.NRG = pp.d.NRG = NRG;
		pp.w.C = c->quad_C;
		pp.w.HG = c->quad_HG;
		pp.cus = c->cus;
		if (q >= 0) { // the word pass q: reads lattice buffer `cur` and mask buffer q & 1
			for (int color = 0; color < 2; color++) {
				pp.w.src[color] = c->d_quad + ((size_t)cur * 2 + color) * qw;
				pp.w.dst[color] = c->d_quad + ((size_t)(cur ^ 1) * 2 + color) * qw;
			}
			pp.w.masks = c->d_qmasks + (size_t)(q & 1) * mask_words;
			pp.w.nlev = 2 * passes[q].ns;
			pp.w.cnt = passes[q].meas >= 0 ? d_cnt + (size_t)(bonds ? 16 : 8) * passes[q].meas : nullptr;
			pp.w.cnt_eq = (passes[q].meas >= 0 && bonds) ? d_cnt + (size_t)16 * passes[q].meas + 8 : nullptr;
			cur ^= 1;
		}
		if (q + 1 < np) { // the draws of pass q + 1 into the other mask buffer
			pp.d.masks = c->d_qmasks + (size_t)((q + 1) & 1) * mask_words;
			pp.d.seed_lo = (uint32_t)c->cfg.seed;
			pp.d.seed_hi = (uint32_t)(c->cfg.seed >> 32);
			pp.d.it = (uint32_t)passes[q + 1].it;
			pp.d.n3 = (uint32_t)c->thr[3];
			pp.d.n4 = (uint32_t)c->thr[4];
			pp.d.nlev = 2 * passes[q + 1].ns;
		}
		HIP_TRY(ising::launch_quad_pass(pp, c->quad_waves, c->stream));
	}
	for (int color = 0; color < 2; color++) HIP_TRY(ising::launch_quad_to_dense(c->d_quad + ((size_t)cur * 2 + color) * qw, dense_plane(color), gx, NRG, c->stream));
	return ISING_OK;
}

static int sweep_quad_counted(ising_ctx *c, int first_it, int nsweeps, int every, uint64_t *ups, int64_t *bond_equal, long long n, int *ncounts) {
	if (int rc = bind(c)) return rc;
	const size_t per = bond_equal ? 16 : 8, need = (size_t)n * per;
	if (c->tile_cnt_cap < need) {
		if (c->d_tile_cnt) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_tile_cnt)); c->d_tile_cnt = nullptr; c->tile_cnt_cap = 0; }
		const size_t cap = std::max<size_t>(64, need);
		HIP_TRY(hipMalloc((void **)&c->d_tile_cnt, cap * sizeof(unsigned long long)));
		c->tile_cnt_cap = cap;
	}
	if (need) HIP_TRY(hipMemsetAsync(c->d_tile_cnt, 0, need * sizeof(unsigned long long), c->stream));
	int k = 0;
	if (nsweeps > 0) if (int rc = sweep_quad(c, first_it, nsweeps, every, c->d_tile_cnt, &k, bond_equal != nullptr)) return rc;
	std::vector<unsigned long long> h((size_t)std::max(k, 1) * per);
	if (k) HIP_TRY(hipMemcpyAsync(h.data(), c->d_tile_cnt, (size_t)k * per * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
	if (int rc = ising_host::sync_checked(c)) return rc;
	for (int q = 0; q < k; q++) {
		unsigned long long u = 0, e = 0;
		for (int w = 0; w < 8; w++) { u += h[(size_t)q * per + w]; if (bond_equal) e += h[(size_t)q * per + 8 + w]; }
		ups[q] = u;
		if (bond_equal) bond_equal[q] = (int64_t)e;
	}
	*ncounts = k;
	return ISING_OK;
}

// ---- The run-time guard under the shape table.  The fused launches' strip height and workgroups per CU come from tables fitted on the boxes
// this library was measured on (ising_capi.cpp: choose_fused_strip_rows, fused_wgs_for), and their mid-size entries sit next to cliffs -- 16384^2 at eight-row strips
// and six per CU: 3392 flips/ns on one box, 835 on the next --; tests/test_gpu_policy.py finds a moved cliff only when somebody runs it.  So the first
// launches of a lone slab's fused form are timed on their dispatch packets (the host waits for them: once per context).  On the plateau (4096 tickets a level and
// more) a rate of 0.8 x what a lattice of that size runs at settles it after ONE launch.  Below, or under that rate, the neighbouring shapes get one launch each -- a
// workgroup per CU fewer, one more (and on down the slope while it pays), half the strip height -- and the fastest stays if it is worth 3 %.  A shape never changes
// results (every unit's draws depend on its sites and the iteration alone), only who computes what when.
namespace {

// flips/ns a lone slab of this size runs at in the fused form on a whole MI355X (the plateau of DESIGN 4.1 and the mid-size table of 4.2, rounded down; dead lanes of
// a partly filled last wave column draw for nothing).  ISING_GUARD_EXPECT overrides.
double guard_expected(const ising_ctx *c, bool model_only = false) {
	if (c->pol.guard_expect > 0 && !model_only) return c->pol.guard_expect;
	const long long spins = (long long)c->cfg.X * c->cfg.Y;
	const double full = spins >= (1LL << 28) ? 3300.0 : (spins >= (1LL << 27) ? 3150.0 : (spins >= (1LL << 26) ? 2950.0 : (spins >= (1LL << 25) ? 2400.0 : (spins >= 3 * (1LL << 23) ? 1700.0 : 1400.0))));
	return full * (double)c->gx / (4.0 * c->nwc()) * std::min(1.0, c->cus / 256.0);
}

int guard_wg_now(const ising_ctx *c) {
	if (!c->guard.grid_override && c->pol.fused_wgs > 0) return std::max(1, c->pol.fused_wgs / std::max(1, c->cus));
	return c->fused_wg_per_cu > 0 ? c->fused_wg_per_cu : 6;
}

// the slab's fused launches take shape (H, wg) from the next one on
int guard_apply(ising_ctx *c, int H, int wg) {
	if (H != c->H) { // other strips: their completion counters start over (stream order: behind every launch so far)
		const size_t done_words = c->ctl_strips + 2 * (size_t)c->ghost_rows + 2;
		HIP_TRY(hipMemsetAsync(c->d_slotctl + ising_host::SLOTCTL_TICKET_BYTES / 4, 0, done_words * sizeof(uint32_t), c->stream));
		c->done_base = 0;
		c->H = H;
		c->nstrips = c->cfg.Y / H;
		if (c->pol.fused_tickets2 < 0) c->fused_tickets2 = H == 1 ? 4 : (H == 2 ? 2 : 0);
	}
	c->fused_wg_per_cu = wg;
	c->guard.grid_override = true;
	return ISING_OK;
}

// A timed launch is in flight: wait for it, judge it, move on.
int guard_settle(ising_ctx *c) {
	ising_ctx::ShapeGuard &g = c->guard;
	if (!g.pending) return ISING_OK;
	g.pending = false;
	float ms = 0;
	if (hipEventSynchronize(g.e1) != hipSuccess || hipEventElapsedTime(&ms, g.e0, g.e1) != hipSuccess || ms <= 0) { (void)hipGetLastError(); g.state = 3; g.form = 3; g.form_pending = 0; return ISING_OK; }
	const float rate = (float)(g.pending_flips / ((double)ms * 1.0e6));
	g.launches++;
	if (g.form_pending) { // the form of a long call's launches: a split launch, then a fused one, the faster stays (2 %: a launch's noise)
		if (g.form_pending == 1) { g.split_rate = rate; g.form = 1; }
		else {
			g.fused_rate = rate;
			g.form = 3;
			g.no_split = g.fused_rate > 1.02f * g.split_rate;
		}
		g.form_pending = 0;
		return ISING_OK;
	}
	auto add = [&](int H, int wg) {
		if (g.ncand >= ising_ctx::ShapeGuard::MAXC || wg < 1 || wg > 6) return;
		if (H == g.base_H && wg == g.base_wg) return;
		for (int k = 0; k < g.ncand; k++) if (g.cand_H[k] == H && g.cand_wg[k] == wg) return;
		g.cand_H[g.ncand] = H; g.cand_wg[g.ncand++] = wg;
	};
	if (g.state == 1) {
		g.timed++;
		g.base_rate = std::max(g.base_rate, rate);
		// Where a level has tickets for every workgroup the chip holds several times over (the plateau: 32768^2 and up) no shape of the table sits near a cliff: a rate
		// near the expectation settles it.  Below, the neighbours get a launch each whatever the rate -- a shape one step from a cliff loses a fifth, not three
		// quarters (8192 x 1536: 1662 / 1521 / 1240 flips/ns at two / three / four workgroups per CU), and no table of expectations is that sharp.
		const long long tickets = ((long long)c->nwc() * (c->cfg.Y / std::max(1, g.base_H)) + 3) / 4;
		const bool plateau = tickets >= 4096;
		if (plateau && rate >= 0.8f * g.expected) { g.state = 3; g.best_rate = g.base_rate; return ISING_OK; }
		if (g.timed < 2) return ISING_OK; // (a context's first launch also pays for the code's way into the chip: one more before anything changes)
		g.best_H = g.base_H; g.best_wg = g.base_wg; g.best_rate = g.base_rate;
		g.ncand = 0;
		add(g.base_H, g.base_wg - 1);
		add(g.base_H, g.base_wg + 1);
		if (g.base_H >= 2 && (size_t)(c->cfg.Y / (g.base_H / 2)) <= c->ctl_strips) add(g.base_H / 2, g.base_wg); // (twice the tickets a level, the same grid)
		if (!g.ncand) { g.state = 3; return ISING_OK; }
		g.state = 2;
		g.cand = 0;
		return guard_apply(c, g.cand_H[0], g.cand_wg[0]);
	}
	// state 2: candidate g.cand ran
	const int cH = g.cand_H[g.cand], cw = g.cand_wg[g.cand];
	if (rate > g.best_rate) {
		// (downhill from here? one more step the same way, a launch each, while it pays)
		if (rate > 1.01f * g.best_rate && cH == g.best_H && cw != g.best_wg) add(cH, cw + (cw > g.best_wg ? 1 : -1));
		g.best_rate = rate; g.best_H = cH; g.best_wg = cw;
	}
	if (++g.cand >= g.ncand) {
		g.state = 3;
		// (a neighbour stays when it is worth it: 3 % -- two launches' noise is one --, or anything where the table's shape fell short of the expectation)
		const bool worth = g.best_rate > 1.03f * g.base_rate || (g.base_rate < 0.8f * g.expected && g.best_rate > g.base_rate);
		g.switched = worth && (g.best_H != g.base_H || g.best_wg != g.base_wg);
		if (!g.switched) { // back to the table's shape, ISING_FUSED_WGS included
			g.best_H = g.base_H; g.best_wg = g.base_wg; g.best_rate = g.base_rate;
			const int rc = guard_apply(c, g.base_H, g.base_wg);
			g.grid_override = false;
			return rc;
		}
		return guard_apply(c, g.best_H, g.best_wg);
	}
	return guard_apply(c, g.cand_H[g.cand], g.cand_wg[g.cand]);
}

// a call whose launches the guard does not time (print points inside the launches: their slots are laid out by the strips): what is known so far decides
int guard_finish(ising_ctx *c) {
	ising_ctx::ShapeGuard &g = c->guard;
	if (g.pending) if (int rc = guard_settle(c)) return rc;
	if (g.state != 1 && g.state != 2) return ISING_OK;
	if (int rc = guard_settle(c)) return rc;
	if (g.state != 2) return ISING_OK;
	g.state = 3;
	g.switched = g.best_rate > 1.03f * g.base_rate && (g.best_H != g.base_H || g.best_wg != g.base_wg);
	if (!g.switched) { g.best_H = g.base_H; g.best_wg = g.base_wg; g.best_rate = g.base_rate; }
	if (int rc = guard_apply(c, g.best_H, g.best_wg)) return rc;
	if (!g.switched) g.grid_override = false;
	return ISING_OK;
}

// in front of a plain fused launch of `ns` sweeps: the events it is to be timed with, or nothing
void guard_before(ising_ctx *c, int ns, hipEvent_t *stop) {
	ising_ctx::ShapeGuard &g = c->guard;
	*stop = nullptr;
	if (g.state != 1 && g.state != 2) return;
	if (g.state == 1 && g.timed == 0) { g.expected = (float)guard_expected(c); g.base_H = c->H; g.base_wg = guard_wg_now(c); }
	const double flips = (double)c->cfg.X * c->cfg.Y * ns;
	if (flips < 2.0e6 * guard_expected(c, true)) return; // (under ~2 ms the launch's fixed part shows: not a launch to judge a shape by)
	c->launch_start_next = g.e0;
	*stop = g.e1;
	g.pending = true;
	g.pending_flips = flips;
}

} // namespace

// `nsweeps` sweeps of a slab that needs nothing from its neighbours: a single slab that wraps in place, or a slab of
// sub-lattices (also one of several: nothing crosses slabs, optimized/main.cu:1423-1462)
int ising_host::sweep_alone(ising_ctx *c, int first_it, int nsweeps) {
	// ballot layout: many sweeps per fused launch (32 at 65536^2, more on smaller lattices) -- the chip does not drain between colours
	if (sweeps_fused(c)) {
		const int per_launch = ising_host::fused_sweeps_per_launch(c->pol, (long long)c->cfg.X * c->cfg.Y);
		for (int it = first_it, left = nsweeps; left > 0;) {
			const int ns = std::min(left, per_launch);
			const bool split = split_pays(c, nsweeps); // (per launch: the guard may decide against the form in the middle of a call)
			c->split_next = split;
			if (split && c->guard.e0 && c->guard.form < 3 && !c->split_always && c->wrap) {
				// where the table says "split": one launch of each form on this box, timed, before the call's other launches follow the faster
				// (the split form was 1-6 % ahead on round 5's boxes and 3-5 % behind at 65536 x 8192 and 24576^2 on one of round 6's)
				if (int rc = guard_settle(c)) return rc;
				ising_ctx::ShapeGuard &g = c->guard;
				const bool as_split = g.form == 0 || (g.form == 3 && !g.no_split);
				hipEvent_t stop = nullptr;
				const bool long_enough = (double)c->cfg.X * c->cfg.Y * ns >= 2.0e6 * guard_expected(c, true);
				if (g.form == 0 && !g.form_warm && long_enough) g.form_warm = true; // (this one runs untimed: the form's first launch)
				else if (g.form < 2 && long_enough) {
					c->launch_start_next = g.e0;
					stop = g.e1;
					g.pending = true;
					g.pending_flips = (double)c->cfg.X * c->cfg.Y * ns;
					g.form_pending = as_split ? 1 : 2;
					if (!as_split) g.form = 2;
				}
				c->split_next = as_split;
				if (int rc = launch_ranges(c, it, ISING_BLACK, 0, c->cfg.Y, 0, 0, 2 * ns, stop)) { g.pending = false; g.form_pending = 0; return rc; }
				it += ns;
				left -= ns;
				continue;
			}
			if (c->guard.state == 1 || c->guard.state == 2) {
				if (int rc = guard_settle(c)) return rc;
				hipEvent_t stop = nullptr;
				if (!split && c->wrap) guard_before(c, ns, &stop);
				if (int rc = launch_ranges(c, it, ISING_BLACK, 0, c->cfg.Y, 0, 0, 2 * ns, stop)) { c->guard.pending = false; return rc; }
				it += ns;
				left -= ns;
				continue;
			}
			if (int rc = launch_ranges(c, it, ISING_BLACK, 0, c->cfg.Y, 0, 0, 2 * ns)) return rc;
			it += ns;
			left -= ns;
		}
		return ISING_OK;
	}
	if (sweeps_quad(c, nsweeps)) return sweep_quad(c, first_it, nsweeps, 0, nullptr, nullptr);
	if (sweeps_tiled(c, nsweeps)) return sweep_tiles(c, first_it, nsweeps);
	for (int it = first_it; it < first_it + nsweeps; it++) {
		if (int rc = ising_update_color(c, it, ISING_BLACK, 0, c->cfg.Y)) return rc;
		if (int rc = ising_update_color(c, it, ISING_WHITE, 0, c->cfg.Y)) return rc;
	}
	return ISING_OK;
}

// In-launch counts (ising_ballot.hip: COUNT): the wave slots of a chunk of measurements of fused launches over `strips` strips, and their sums behind them.
// slots: per measurement one per wave of a level and colour; a call is worked off in chunks of measurements whose slots fit 64 MiB.
int ising_host::cnt_reserve(ising_ctx *c, size_t strips, bool bonds, size_t *slots, size_t *n_up, size_t *chunk, unsigned long long **d_sum) {
	const size_t waves = ((size_t)4 * c->nwc() * strips + 15) / 16 * 4; // 4 waves per workgroup unit, as launch_ballot_update counts them
	*n_up = 2 * waves;
	*slots = (bonds ? 3 : 2) * waves;
	*chunk = std::max<size_t>(1, std::min<size_t>(64, ((size_t)64 << 20) / (*slots * sizeof(uint32_t))));
	const size_t words = (*chunk * *slots + 15) / 16 * 16, bytes = words * sizeof(uint32_t) + 2 * *chunk * sizeof(unsigned long long) + 64;
	if (c->cnt_cap < bytes) {
		if (c->d_cnt) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_cnt)); c->d_cnt = nullptr; c->cnt_cap = 0; }
		HIP_TRY(hipMalloc((void **)&c->d_cnt, bytes));
		c->cnt_cap = bytes;
	}
	*d_sum = reinterpret_cast<unsigned long long *>(c->d_cnt + words);
	return ISING_OK;
}

// The reference's loop with its print points (optimized/main.cu:1763-1810: sweep, and countSpins whenever the iteration is a multiple of
// printFreq): `nsweeps` sweeps, the up-spin count after every iteration `it` with it % every == 0 -- and, `bond_equal` not null, ising_bond_equal's sum
// at the same points (north_star's energy series; the reference computes none).  Where ising_sweep issues fused launches (a lone slab on the ballot
// layout, no sub-lattices, no couplings) both are taken INSIDE the launches -- no launch boundary and no read-back between two print points
// (ising_ballot.hip: COUNT) --; elsewhere: sweeps, ising_count and ising_bond_equal in turn.
extern "C" int ising_sweep_counted(ising_ctx *c, int first_it, int nsweeps, int every, uint64_t *ups, int64_t *bond_equal, int max_counts, int *ncounts) {
	if (!c || !ups || !ncounts) return fail(ISING_E_ARG, "null argument");
	if (first_it < 0 || nsweeps < 0 || every < 1) return fail(ISING_E_ARG, "bad iteration range or count interval");
	if (!c->wrap) return fail(ISING_E_STATE, "ising_sweep_counted needs a single slab without ring halo rows (rings: ising_ring_sweep_counted / ising_rank_sweep_counted)");
	const long long last = (long long)first_it + nsweeps - 1;
	const long long n = nsweeps > 0 ? last / every - ((long long)first_it - 1) / every : 0; // iterations in [first_it, last] that are multiples of `every`
	*ncounts = 0;
	if (n > max_counts) return fail(ISING_E_ARG, "%lld counts, room for %d", n, max_counts);
	if (int rc = bind(c)) return rc;
	if (sweeps_quad(c, 2)) return sweep_quad_counted(c, first_it, nsweeps, every, ups, bond_equal, n, ncounts);
	if (sweeps_tiled(c, 2) && !bond_equal) return sweep_tiles_counted(c, first_it, nsweeps, every, ups, n, ncounts);
	const bool inside = sweeps_fused(c) && !c->cfg.XSL && !c->cfg.use_J;
	if (!inside) { // one launch per colour, tiles with the energy, sub-lattices, couplings: the reference's own order of events
		int it = first_it, k = 0;
		while (it <= last) {
			const long long next = std::min<long long>(last, ((long long)it + every - 1) / every * every); // the next multiple of `every` from `it` on
			if (int rc = ising_host::sweep_alone(c, it, (int)(next - it + 1))) return rc;
			it = (int)next + 1;
			if (next % every == 0) {
				uint64_t up = 0, dw = 0;
				if (int rc = ising_count(c, &up, &dw)) return rc;
				if (bond_equal) if (int rc = ising_bond_equal(c, &bond_equal[k])) return rc;
				ups[k++] = up;
				*ncounts = k;
			}
		}
		return ISING_OK;
	}
	if (int rc = guard_finish(c)) return rc;
	size_t slots = 0, n_up = 0, chunk = 0;
	unsigned long long *d_sum = nullptr;
	const bool split = split_pays(c, nsweeps); // (one form of launch per call: the slots of a measurement are laid out by its strips)
	if (int rc = ising_host::cnt_reserve(c, (size_t)(split ? c->cfg.Y / c->H_split : c->nstrips), bond_equal != nullptr, &slots, &n_up, &chunk, &d_sum)) return rc;
	const int per_launch = ising_host::fused_sweeps_per_launch(c->pol, (long long)c->cfg.X * c->cfg.Y); // (a counted launch is as long as any other)
	std::vector<unsigned long long> h(2 * chunk);
	long long got = 0;
	int it = first_it, left = nsweeps;
	while (left > 0) {
		// launches until `chunk` measurements are in flight or the sweeps are done
		if (n > got) HIP_TRY(hipMemsetAsync(c->d_cnt, 0, std::min<size_t>(chunk, (size_t)(n - got)) * slots * sizeof(uint32_t), c->stream)); // (the slots of the measurements still to come, at most a chunk)
		int inflight = 0;
		while (left > 0) {
			int ns = std::min(left, per_launch);
			const int first = (every - it % every) % every; // the launch's first measured sweep: iteration it + first is a multiple of `every`
			int m = first < ns ? (ns - 1 - first) / every + 1 : 0;
			if (inflight + m > (int)chunk) { // (the launch ends in front of the measurement that no longer fits)
				m = (int)chunk - inflight;
				ns = first + m * every;
			}
			if (ns == 0) break;
			c->cnt_first_next = first;
			c->cnt_every_next = m > 0 ? every : 0;
			c->cnt_slot0_next = inflight;
			c->cnt_bonds_next = bond_equal != nullptr;
			c->split_next = split;
			if (int rc = launch_ranges(c, it, ISING_BLACK, 0, c->cfg.Y, 0, 0, 2 * ns)) { *ncounts = (int)got; return rc; } // (the counts of the chunks before are the caller's)
			inflight += m;
			it += ns;
			left -= ns;
		}
		HIP_TRY(ising::launch_count_fold(c->d_cnt, slots, n_up, inflight, d_sum, c->stream));
		if (inflight) HIP_TRY(hipMemcpyAsync(h.data(), d_sum, (size_t)inflight * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
		if (int rc = ising_host::sync_checked(c)) { *ncounts = (int)got; return rc; }
		for (int k = 0; k < inflight; k++, got++) {
			ups[got] = h[2 * (size_t)k];
			if (bond_equal) bond_equal[got] = (int64_t)h[2 * (size_t)k + 1];
		}
	}
	*ncounts = (int)got;
	return ISING_OK;
}

extern "C" {

int ising_sweep_info(ising_ctx *c, int *fused, int *max_sweeps_per_launch) {
	if (!c) return fail(ISING_E_ARG, "null context");
	const bool f = (c->wrap || c->cfg.XSL) && sweeps_fused(c);
	// (a ring slab with ghost rows G deep: the ring's sweeps are fused launches of G/2 sweeps between two exchanges)
	const bool deep = ghost_sweeps(c);
	const bool quad = !f && !deep && sweeps_quad(c, 2);
	const bool tiled = !f && !deep && !quad && sweeps_tiled(c, 2);
	if (fused) *fused = ((f || deep) && c->split && !c->cfg.XSL) ? 3 : ((f || deep) ? 1 : (quad ? 4 : (tiled ? 2 : 0)));
	if (max_sweeps_per_launch) *max_sweeps_per_launch = f ? ising_host::fused_sweeps_per_launch(c->pol, (long long)c->cfg.X * c->cfg.Y) : (deep ? c->ghost() / 2 : (quad ? c->quad_T : (tiled ? c->tile_sweeps : 0)));
	return ISING_OK;
}

int ising_shape_guard_info(ising_ctx *c, ising_guard_info *out) {
	if (!c || !out) return fail(ISING_E_ARG, "null argument");
	if (c->guard.pending) if (int rc = guard_settle(c)) return rc; // (a timed launch in flight: its verdict first -- blocks)
	const ising_ctx::ShapeGuard &g = c->guard;
	memset(out, 0, sizeof(*out));
	out->state = g.state;
	out->switched = g.switched ? 1 : 0;
	out->launches_timed = g.launches;
	out->table_strip_rows = g.base_H; out->table_wg_per_cu = g.base_wg;
	out->strip_rows = c->H; out->wg_per_cu = guard_wg_now(c);
	out->expected_flips_per_ns = g.expected; out->table_flips_per_ns = g.base_rate; out->kept_flips_per_ns = g.best_rate;
	out->form_state = c->split && !c->split_always ? g.form : -1;
	out->split_flips_per_ns = g.split_rate; out->fused_flips_per_ns = g.fused_rate;
	out->split_kept = c->split && !g.no_split ? 1 : 0;
	return ISING_OK;
}

int ising_sweep_form(ising_ctx *c, int nsweeps, int *form, int *strip_rows, int *wg_per_cu) {
	if (!c) return fail(ISING_E_ARG, "null context");
	int f = 0;
	if (int rc = ising_sweep_info(c, &f, nullptr)) return rc;
	const bool fused_any = f == 1 || f == 3;
	const bool split = f == 3 && (ghost_sweeps(c) ? c->split_always : split_pays(c, nsweeps));
	if (form) *form = f == 3 ? (split ? 3 : 1) : f;
	if (strip_rows) *strip_rows = fused_any ? (split ? c->H_split : c->H) : 0;
	if (wg_per_cu) *wg_per_cu = fused_any ? (split ? c->split_wg_per_cu : c->fused_wg_per_cu) : 0;
	return ISING_OK;
}

int ising_sweep_timed(ising_ctx *c, int first_it, int nsweeps, float *elapsed_ms) {
	if (!c || !elapsed_ms) return fail(ISING_E_ARG, "null argument");
	if (int rc = bind(c)) return rc;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	hipError_t e = hipEventCreate(&e0);
	if (e == hipSuccess) e = hipEventCreate(&e1);
	if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
	int rc = e == hipSuccess ? ising_sweep(c, first_it, nsweeps) : fail(ISING_E_HIP, "event timing failed: %s", hipGetErrorString(e));
	if (rc == ISING_OK) {
		e = hipEventRecord(e1, c->stream);
		if (e == hipSuccess) e = hipEventSynchronize(e1);
		if (e == hipSuccess) e = hipEventElapsedTime(elapsed_ms, e0, e1);
		if (e != hipSuccess) rc = fail(ISING_E_HIP, "event timing failed: %s", hipGetErrorString(e));
		else rc = ising_host::check_abort(c);
	}
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	return rc;
}

int ising_halo_ptrs(ising_ctx *c, int color, void **send_top, void **send_bot, void **recv_top, void **recv_bot, size_t *row_bytes) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (color != ISING_BLACK && color != ISING_WHITE && color != ISING_HAM_BLACK) return fail(ISING_E_ARG, "bad colour %d", color);
	if (color == ISING_HAM_BLACK && !c->cfg.use_J) return fail(ISING_E_STATE, "couplings are not enabled (use_J)");
	if (c->wrap) return fail(ISING_E_STATE, "no halo buffers with nslabs == 1 (rows wrap inside the slab)");
	uint64_t *base = c->plane(color);
	const size_t ld = (size_t)c->plane_ld(color);
	if (send_top) *send_top = base;
	if (send_bot) *send_bot = base + (size_t)(c->cfg.Y - 1) * ld;
	if (recv_top) *recv_top = base - ld;
	if (recv_bot) *recv_bot = base + (size_t)c->cfg.Y * ld;
	if (row_bytes) *row_bytes = ld * sizeof(uint64_t);
	return ISING_OK;
}

int ising_ghost_ptrs(ising_ctx *c, int color, int *depth, void **send_top, void **send_bot, void **recv_top, void **recv_bot, size_t *block_bytes) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (color != ISING_BLACK && color != ISING_WHITE) return fail(ISING_E_ARG, "bad colour %d", color);
	if (c->wrap) return fail(ISING_E_STATE, "no halo buffers with nslabs == 1 (rows wrap inside the slab)");
	const size_t ld = (size_t)c->lld, G = ghost_sweeps(c) ? (size_t)c->ghost() : 1;
	uint64_t *base = c->lat(color);
	if (depth) *depth = (int)G;
	if (send_top) *send_top = base;
	if (send_bot) *send_bot = base + ((size_t)c->cfg.Y - G) * ld;
	if (recv_top) *recv_top = base - G * ld;
	if (recv_bot) *recv_bot = base + (size_t)c->cfg.Y * ld;
	if (block_bytes) *block_bytes = G * ld * sizeof(uint64_t);
	return ISING_OK;
}

int ising_ghost_delivered(ising_ctx *c, int color) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (color != ISING_BLACK && color != ISING_WHITE) return fail(ISING_E_ARG, "bad colour %d", color);
	if (c->wrap) return fail(ISING_E_STATE, "no halo buffers with nslabs == 1 (rows wrap inside the slab)");
	c->ghost_depth[color] = ghost_sweeps(c) ? c->ghost() : 1;
	return ISING_OK;
}

int ising_sweep_ghost(ising_ctx *c, int first_it, int nsweeps) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (!ghost_sweeps(c)) return fail(ISING_E_STATE, "the slab does not sweep through ghost rows (ising_ghost_ptrs: depth 1); use ising_update_edges / ising_update_color");
	const int G = c->ghost();
	if (nsweeps < 1 || 2 * nsweeps > G) return fail(ISING_E_ARG, "%d sweeps on ghost rows %d deep (at most %d per exchange)", nsweeps, G, G / 2);
	if (c->ghost_depth[0] < G || c->ghost_depth[1] < G)
		return fail(ISING_E_STATE, "the ghost rows are not current: deliver both colours (ising_ghost_ptrs, ising_ghost_delivered) after whatever changed the spins");
	for (int color = 0; color < 2; color++) if (int rc = ising_host::halo_ready(c, color)) return rc; // (a no-op unless the library's own transport is attached too)
	return ising_host::update_deep(c, first_it, 2 * nsweeps);
}


// Test aids (tests/test_gpu_fused.py).  what = 1 puts the host's idea of the completion counters out of step with the device,
// as a faulted launch would leave it -- the next fused launch's units wait for counts that never come -- and lowers the
// bound after which they give up to `arg` polls (0: keep).  what = 2 ages every monotone counter of the slab, device and
// host record together, as billions of sweeps would: the completion counters stand past the point where the next launch
// starts them over, the exchange's counters (units that have left the edge rows, epochs) a few counts before 2^32.
int ising_debug_launch_shape(ising_ctx *c, int *strip_rows, int *wg_per_cu, int *split_lead) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (strip_rows) *strip_rows = c->split ? c->H_split : c->H;
	if (wg_per_cu) *wg_per_cu = c->split ? c->split_wg_per_cu : c->fused_wg_per_cu;
	if (split_lead) *split_lead = c->split ? c->split_lead : 0;
	return ISING_OK;
}

int ising_debug_fault(ising_ctx *c, int what, int arg) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (what == 2) {
		if (int rc = ising_synchronize(c)) return rc;
		if (c->comm) HIP_TRY(hipStreamSynchronize(c->comm));
		if (c->d_slotctl && c->slotctl_bytes > SLOTCTL_TICKET_BYTES) {
			const uint32_t add = (1u << 30) + 12345u - c->done_base; // (counters of strips that lag a level keep their distance)
			std::vector<uint32_t> h((c->slotctl_bytes - SLOTCTL_TICKET_BYTES) / 4);
			HIP_TRY(hipMemcpy(h.data(), c->d_slotctl + SLOTCTL_TICKET_BYTES / 4, h.size() * 4, hipMemcpyDeviceToHost));
			for (auto &v : h) v += add;
			HIP_TRY(hipMemcpy(c->d_slotctl + SLOTCTL_TICKET_BYTES / 4, h.data(), h.size() * 4, hipMemcpyHostToDevice));
			c->done_base += add;
			c->split_done_base += add; // (the split form's counters are the second half of the same words)
		}
		if (c->d_edge) {
			uint32_t h[32];
			HIP_TRY(hipMemcpy(h, c->d_edge, sizeof(h), hipMemcpyDeviceToHost));
			const uint32_t add_done = 0xFFFFFFF0u - c->edge_done_target, add_go = 0xFFFFFFFDu - c->edge_go_epoch;
			h[0] += add_done;
			h[16] += add_go;
			HIP_TRY(hipMemcpy(c->d_edge, h, sizeof(h), hipMemcpyHostToDevice));
			c->edge_done_target += add_done;
			c->edge_go_epoch += add_go;
		}
		return ISING_OK;
	}
	if (what != 1) return fail(ISING_E_ARG, "unknown fault %d", what);
	c->done_base += 1u << 20;
	c->split_done_base += 1u << 20;
	if (arg > 0) c->pol.abort_polls = (uint32_t)arg;
	return ISING_OK;
}

} // extern "C"
