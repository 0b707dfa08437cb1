// ising_ctx.hpp -- the context object behind the C-ABI handle and the helpers the host-side translation units of
// libising_hip.so share (ising_capi.cpp: slab life cycle and launch-shape policy; ising_update.cpp: updates and sweeps;
// ising_observe.cpp: observables; ising_couplings.cpp: -J; ising_io.cpp: boundary formats; ising_ring.cpp / ising_ipc.cpp: the
// slab ring and its transports; ising_batch.cpp).  Internal to the library.
#pragma once
#include <algorithm>
#include "../../include/ising_hip.h"
#include "../../include/ising_hip_testing.h"
#include "ising_kernels.h"

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <vector>

// A/B and test switches.  The environment is read ONCE per context, in ising_create, into this record; nothing
// in the library calls getenv afterwards (ISING_RCCL_LIB, the path of the RCCL to open, is process-wide and read once).
struct ising_policy {
	int fused = -1;          // ISING_FUSED=0/1: one launch per colour / fused launches (-1: by lattice size)
	int fused_nt = -1;       // ISING_FUSED_NT=0/1: non-temporal lattice words (-1: lattices above 2^31 spins)
	int fused_tickets2 = -1; // ISING_FUSED_TICKETS2=0/2/4: ticket counters (-1: by strip height)
	int fused_wgs = 0;       // ISING_FUSED_WGS=n: persistent grid of n workgroups (0: by tickets per level)
	int guard = -1;          // ISING_GUARD=0/1: the run-time guard under the fused launches' shape table (ising_update.cpp: guard_*; -1: on a whole MI355X)
	float guard_expect = 0;  // ISING_GUARD_EXPECT=flips/ns: what the guard holds the first launches against (0: by lattice size)
	int fused_wait_late = -1; // ISING_FUSED_WAIT_LATE=0/1: units of fused launches draw their first row before they wait for their parents (-1: by tickets per level)
	int fused_max_sweeps = 0; // ISING_FUSED_MAX_SWEEPS=n: sweeps one fused launch of a single slab carries at most (0: ~50 ms worth, 32 .. 4096)
	int tiles = -1;          // ISING_TILES=0/1: small lattices on the dense layout sweep in tile launches of several sweeps (-1: by lattice size)
	int tile_rows = 0, tile_words = 0, tile_sweeps = 0, tile_threads = 0, tile_xcd = -1; // ISING_TILE_ROWS / _WORDS / _SWEEPS / _THREADS / _XCD (0 / -1: by lattice size)
	int quad = -1;           // ISING_QUAD=0/1: small lattices on the dense layout sweep on the quad layout -- draws ahead of the lattice, word passes on tiles (ising_quad.hip; -1: by lattice size)
	int quad_C = 0, quad_T = 0, quad_waves = 0; // ISING_QUAD_C / _T / _WAVES: row groups per tile, sweeps per pass, waves per workgroup (0: by lattice size)
	int split = -1;          // ISING_SPLIT=0/1: fused launches of a lone slab in the split form (draw units / word units, ising_ballot.hip: ballot_split_k; -1: by tickets per level)
	int split_lead = -1;     // ISING_SPLIT_LEAD=n: draw units a workgroup does before its first word unit (-1: 1)
	int ring_counted = -1;   // ISING_RING_COUNTED=0/1/2: print points of rings never / where possible (default) / always (an error where not) inside the deep launches
	int ring_ghost = -1;     // ISING_RING_GHOST=n: ghost rows of ballot ring slabs (-1: 64)
	int ring_epochs = 0;     // ISING_RING_EPOCHS=n: exchange epochs one persistent launch of a ring slab carries at most (0: ~50 ms worth, up to 64; 1: a launch per exchange)
	bool no_ballot = false;  // ISING_NO_BALLOT: layout AUTO never picks the ballot layout
	int tail_rows = -1, tail_h = 1; // ISING_TAIL=rows[,h]: one-row tail strips of one-launch-per-colour launches (-1: automatic, 0: off)
	bool trapezoid = true;   // ISING_RING_TRAPEZOID=0: every ghost row at every level
	int overlap = 1;         // ISING_RING_OVERLAP: 0 = the deep exchange between two launches (round 2); 1 (default) = started by the
	                         // running launch's edge strips and hidden in its tail, the next launch waits for it on the stream;
	                         // 2 = free-running: the next launch starts regardless and only its edge strips wait (copies / IPC
	                         // only: a transport whose kernels cannot be placed next to a chip-filling launch would never finish)
	int ring_inline = -1;    // ISING_RING_INLINE=0/1: peer copies on the comm / the compute stream (-1: by device placement)
	int ring_store = -1;     // ISING_RING_STORE=0/1: never / always store edge rows into the neighbours' halo rows (-1: automatic)
	bool comm_priority = true;                     // ISING_RING_COMM_PRIORITY=0: comm streams at default priority
	int ring_transport = 0;  // ISING_RING_TRANSPORT=copy/rccl/auto -> ISING_TRANSPORT_*
	uint32_t abort_polls = 1u << 22; // ISING_ABORT_POLLS: polls (~2.5 us each) a unit of a fused launch waits for its parents before it
	                                 // gives the launch up (~10 s: nothing legitimate waits a level's time that long)
};

struct ising_ctx {
	ising_config cfg{};
	ising_policy pol{};
	int cus = 256;                 // compute units of the slab's device
	int xccs = 8;                  // ... and its accelerator dies (XCDs): the split form's ticket classes are theirs
	bool last_launch_split = false; // the form of the last fused launch (a split launch that gives up takes the form out of service: check_abort)
	uint32_t *h_abort = nullptr;   // pinned, device-visible: [0] != 0 = a fused launch gave up (or the host wants the polling kernels to);
	                               // checked after every synchronise (ising_host::sync_checked)
	bool dense = false; // 1 bit per spin on the device (ising_dense.hip); false = the reference's nibble layout
	bool ballot = false; // dense, with the bits of a row in wave-ballot order (ising_ballot.hip)
	uint64_t *d_lat2 = nullptr;    // dense layout, tile launches (ising_dense.hip: dense_tile_k): the buffer every other launch writes (shape of d_lat)
	int tile_rows = 0, tile_words = 0, tile_sweeps = 0, tile_threads = 0, tile_xcd = 0; // ... their shape (tile_rows == 0: no tile launches)
	unsigned long long *d_tile_cnt = nullptr; // ... the up-spin sums of a call's print points (ising_sweep_counted)
	size_t tile_cnt_cap = 0;
	// Small lattices on the quad layout (ising_quad.hip; ising_update.cpp: sweep_quad): the spins live in d_lat (dense layout) between calls; a call converts,
	// launches once per pass of T sweeps -- the word pass on its tiles + the draws of the pass to come -- and converts back
	int quad_C = 0, quad_T = 0, quad_HG = 0, quad_waves = 0; // tile: row groups, sweeps per pass, halo row groups per side, waves per workgroup (quad_C == 0: off)
	uint64_t *d_quad = nullptr;    // two lattice buffers of 2 colours x Y/4 row groups x gx blocks x 64 words
	uint64_t *d_qmasks = nullptr;  // two mask buffers of 2 quad_T levels x Y/4 x gx KiB
	size_t quad_words() const { return (size_t)cfg.X / 2048 * ((size_t)cfg.Y / 4) * 64; } // per colour
	uint64_t *d_tmp = nullptr;     // ballot layout: dense-order image of d_lat (same shape) for conversions and observables;
	                               // allocated by the first call that needs it (sweeping and counting never do)
	uint64_t *d_scratch = nullptr; // ballot layout: accept-mask slots of the update kernel (see ising_ballot.hip)
	uint64_t *d_scratch_edge = nullptr; // ... of the ring's edge-row launches, which run on the comm stream NEXT TO an interior launch
	bool edge_scratch_next = false;     // the next launch is such a one (set by update_edges_on, cleared by launch_ranges)
	uint32_t *d_slotctl = nullptr; // ballot layout, fused launches: ticket words (576 bytes) + per-strip completion counters
	size_t slotctl_bytes = 0;
	uint32_t done_base = 0;        // value of every completion counter once everything launched so far has run
	size_t ctl_strips = 0;         // strips the completion counters of ONE form have room for (the guard may halve the strip height: twice ising_create's strips)
	// The run-time guard under the shape table (round 6, ising_update.cpp: guard_before / guard_settle): the first launches of a lone slab's fused form are timed on
	// their dispatch packets; under 0.8 x what the table expects of that lattice the neighbouring shapes get one launch each and the fastest stays.
	struct ShapeGuard {
		int state = 0;             // 0: off, 1: timing the table's shape, 2: trying neighbours, 3: settled
		hipEvent_t e0 = nullptr, e1 = nullptr;
		bool pending = false;      // a timed launch is in flight
		double pending_flips = 0;
		int timed = 0;             // launches timed at the table's shape
		int ncand = 0, cand = 0;
		static constexpr int MAXC = 8;
		int cand_H[MAXC] = {}, cand_wg[MAXC] = {};
		int base_H = 0, base_wg = 0, best_H = 0, best_wg = 0;
		float base_rate = 0, best_rate = 0, expected = 0;
		bool switched = false, grid_override = false; // grid_override: the guard's workgroups per CU replace ISING_FUSED_WGS
		int launches = 0;          // launches it looked at
		// ... and the FORM of a long call's launches where the table says "split" (ballot_split_k): one launch of each is timed, the faster stays
		int form = 0;              // 0: nothing timed yet, 1: a split launch timed, 2: a fused one too, 3: decided
		int form_pending = 0;      // the timed launch in flight is 1: a split, 2: a fused one (0: the shape guard's, or none)
		bool form_warm = false;    // a split launch has run untimed (a context's first launch of a form is 2-3 % slow)
		float split_rate = 0, fused_rate = 0;
		bool no_split = false;     // the fused form won: split_pays answers no from now on
	} guard;
	bool fused = false;            // ising_sweep batches colour half-sweeps into fused launches
	int fused_wg_per_cu = 0;       // ... and this many workgroups per CU (0: as many as the chip holds)
	int fused_nt = 0;              // ... whose lattice words carry the non-temporal hint (lattice larger than the 256 MB memory-side cache)
	unsigned long long ticket_base2[4] = {0, 0, 0, 0}; // fused launches: where the launches so far left the ticket counter(s)
	int fused_tickets2 = 0;                      // ... two counters (small lattices: one cannot hand tickets out fast enough)
	bool split = false;            // ... in the split form (ballot_split_k): draw units and word units with tickets of their own, masks through a ring per XCD
	bool split_always = false;     // ISING_SPLIT=1: every launch of several levels; otherwise (ising_create's own choice) the calls that carry split_min_flips and more
	int H_split = 0;               // strip height of the split launches (>= H, the fused launches' -- a call too short for the split form runs those)
	int split_wg_per_cu = 0;       // their persistent grid
	uint32_t split_done_base = 0;  // their completion counters (behind the fused form's in d_slotctl) start every launch from here
	bool split_next = false;       // one-shot: the caller of launch_ranges decided for the split form (sweep_alone, ising_sweep_counted: by the sweeps of the call)
	int split_ring_sh = 0, split_lead = 1, split_cap = 0; // ring slots per class (log2), lead, workgroups a class serves
	uint64_t *d_split_masks = nullptr;            // 8 x 2^split_ring_sh slots x 4 waves x H rows x 1 KiB
	unsigned long long *d_split_ctl = nullptr;    // 8 x 16 ticket words, then 8 x 2^split_ring_sh x 2 slot counts (32-bit); zeroed in front of every launch
	size_t split_ctl_bytes = 0;
	bool fused_wait_late = false;  // ... whose units draw their first row before they wait for their parents (UpdateParams.wait_late)
	int tail_rows = 0, tail_h = 0; // plain full-slab launches: the last tail_rows rows go in strips of tail_h rows
	uint64_t *d_pack = nullptr;    // staging for device-side conversion to / from the packed boundary format
	size_t pack_words = 0;
	uint64_t *d_conv = nullptr;    // ballot layout: the same chunk of rows in dense order (ising_io.cpp: ballot <-> dense <-> packed)
	size_t conv_words = 0;
	struct ising::ReplicaParams *d_self = nullptr; // ballot layout: this slab's record for the ballot-native observables (ballot_measure_k)
	unsigned long long *d_mslots = nullptr;        // ... and their partial sums (BALLOT_MEASURE_SLOTS pairs, a line each)
	const uint64_t *self_lat[2] = {nullptr, nullptr}; // what d_self holds
	int lld_packed = 0; // 64-bit words per colour row in the reference's packed layout (X/32)
	int lld = 0;      // 64-bit words per colour row in the DEVICE layout: X/32 nibble, X/128 dense, ballot: 64 per wave column
	                  // of 8192 lattice columns, i.e. X/128 rounded up to a multiple of 64 (dead lanes of the last one stay zero)
	int lld_dense = 0; // X/128: row size of the dense layout and of the ballot layout's dense-order image d_tmp
	int gx = 0;       // X/2048
	int H = 0;        // rows per strip
	int nstrips = 0;
	size_t color_words = 0;
	uint64_t *d_lat = nullptr;          // [2 colours][Y + 2 rows][lld]: row -1 and row Y of each colour are halo rows
	uint64_t *d_ham = nullptr;          // -J: [hamB, hamW], [Y + 2 rows][lld_packed] each (4 bits per site in both layouts)
	int ham_form = 0;                   // 0: nibbles as generated; 1: per-vector bit-planes (dense); 2: ballot-order planes
	unsigned long long *d_acc = nullptr; // counters: [0] up spins, [1] bond-equal, [2..3] all-reduce staging
	uint32_t *d_bits = nullptr;          // correlations: (Y + d_bits_extra) x lld words, one bit per spin
	int d_bits_extra = 0;
	long long *d_corr = nullptr;         // correlations: 128 sums
	float tab[10]{};
	uint64_t thr[5]{};
	bool fast_ok = false;
	hipStream_t stream = nullptr;
	hipStream_t own_stream = nullptr; // created by ising_use_private_stream, destroyed with the context
	static constexpr int MEAS_CAP = 4096;
	unsigned long long *h_meas = nullptr; // pinned: (up count, bond sum) of the measurements enqueued and not yet fetched
	int meas_pending = 0;

	// ---- ring state (ising_ring.cpp).  The halo rows of colour c travel on `comm`, a second stream per slab:
	//   compute: [wait: halo rows of 1-c have arrived] edge rows of c -> record ev_edge[c] -> interior rows of c
	//   comm:    wait ev_edge[c] -> deliver my first/last row of c (peer copies, or RCCL send/recv) -> record ev_sent[c]
	bool wrap = true;                            // rows -1 / Y mirror the slab's own edge rows (a single slab that is not a ring)
	hipStream_t comm = nullptr;
	hipEvent_t ev_edge[2] = {nullptr, nullptr};
	hipEvent_t ev_sent[2] = {nullptr, nullptr};
	hipEvent_t ev_int[2] = {nullptr, nullptr};  // compute stream: the interior rows of colour c are done
	hipEvent_t ev_go = nullptr;                 // comm stream: the exchange is complete for this slab and edge_go has moved (sweep_deep_overlapped)
	int transport = 0;                           // ISING_TRANSPORT_* in use (0 = not decided yet)
	ising_ctx *ring_prev = nullptr, *ring_next = nullptr; // neighbours in a single-process ring
	void *rccl_comm = nullptr;                   // ncclComm_t of this slab's rank
	bool rccl_owner = false;                     // the communicator was created for this context (destroy it with the context)
	bool rank_mode = false;                      // one slab per process: the neighbours live in other processes
	struct ising_ipc_state *ipc = nullptr;       // ISING_TRANSPORT_IPC: the neighbours' rows mapped through hipIpcMemHandle, flags in
	                                             // POSIX shared memory (ising_ipc.cpp)
	bool peers_enabled = false;
	// deep exchange overlapped with the launches (ising_ring.cpp: sweep_deep_overlapped; UpdateParams.edge_go / edge_done)
	uint32_t *d_edge = nullptr;                  // [0]: units of the launches' last levels that have left the exchange's rows (a count),
	                                             // [16]: epoch of the last exchange that is complete for this slab (set by the comm stream)
	uint32_t edge_done_target = 0;               // value of d_edge[0] once every exchange epoch whose wait is on the comm stream has run
	uint32_t edge_units_per_epoch = 0;           // what the last overlapped deep launch adds to d_edge[0] per exchange epoch (launch_ranges)
	int epoch_sh_next = 0;                       // one-shot: the next deep launch carries several epochs of 2^epoch_sh_next levels (update_deep)
	uint32_t edge_go_epoch = 0;                  // value d_edge[16] is (being) brought to
	bool go_set = false;                         // ... and it stands for the exchange that delivered the current ghost rows
	bool overlap_next = false;                   // one-shot request to launch_ranges: the next deep launch takes part in the overlap
	hipEvent_t launch_start_next = nullptr, launch_stop_next = nullptr; // one-shot: events on the next launch's dispatch packet
	// In-launch counts (ising_sweep_counted): accumulators of the measurements of one call (64 lines of 64 bytes each), and the one-shot
	// request to launch_ranges: which sweeps of the next fused launch are measured, and the number of its first measurement
	uint32_t *d_cnt = nullptr;     // per measurement 2 x (waves of a level) slots, then (64-bit) one sum per measurement
	size_t cnt_cap = 0;            // bytes d_cnt holds
	int ring_cnt_every = 0, ring_cnt_inflight = 0; // ring slab: the deep launches of the call under way count the sweeps whose iteration is a multiple of this (0: none); measurements so far
	int cnt_first_next = 0, cnt_every_next = 0; // (cnt_every_next = 0: the next launch measures nothing)
	int cnt_slot0_next = 0;
	bool cnt_bonds_next = false;   // ... and the launch's white levels leave the equal bonds of their rows too (a third plane of slots per measurement)
	bool ring_cnt_bonds = false;   // ring slab: ... the call under way asked for the energy
	// Measurement aid (ising_kernel_clock): the marks the first eight workgroups of the last fused launch left (cycle and 100 MHz counters, start and end)
	unsigned long long *d_clk = nullptr;
	bool clk_on = false;
	// Exchange statistics (ising_exchange_stats_begin / _fetch; sweep_deep_overlapped): four events per sampled exchange --
	// [4e] launch e begins, [4e+1] launch e ends (both on its dispatch packet), [4e+2] comm stream: the launch's edge strips have
	// finished their last level (the exchange starts), [4e+3] comm stream: the neighbours' rows are in place and edge_go has moved.
	hipEvent_t *xs_ev = nullptr;
	int xs_cap = 0, xs_n = 0, xs_epochs = 0; // slots, sampled launches, the exchange epochs inside them
	bool xs_on = false;
	bool store_ring = false;                     // ring on ONE device and one stream: every launch writes its edge rows straight into the
	                                             // neighbouring slabs' halo rows (UpdateParams.mir0/mirL_bytes): no edge launch, no copies
	bool copy_inline = false;                    // COPY transport, both neighbours on this slab's device: copies on the compute stream

	// Row 0 of a colour.  The halo rows sit directly above (row -1: global row slab*Y-1) and below (row Y) so the
	// kernels address rows -1..Y uniformly.  With one slab they mirror the slab's own last / first row (periodic
	// wrap, maintained by the kernels that write edge rows); with several slabs the neighbours' rows are delivered
	// into them (ising_halo_ptrs / ising_ring_exchange).
	// Ring slabs on the ballot layout keep `ghost_rows` halo rows on either side (rows -G .. -1 and Y .. Y+G-1): the ring
	// then exchanges G rows of both colours every G colour half-sweeps and runs fused launches in between, which update the
	// ghost rows redundantly (ising_ring.cpp: sweep_deep).  Rows -1 and Y are where they always were relative to row 0.
	int ghost_rows = 1;
	int ghost_depth[2] = {0, 0}; // per colour: how deep the ghost rows hold the neighbours' current rows (0: not even row -1 / Y;
	                             // set by the ring's transfers, cleared by whatever changes spins)
	int ghost() const { return ballot ? ghost_rows : 1; }
	uint64_t *lat(int color) const { return d_lat + (size_t)color * (color_words + 2 * (size_t)ghost() * lld) + (size_t)ghost() * lld; }
	uint64_t *halo(int color, int which) const { return which == 0 ? lat(color) - lld : lat(color) + color_words; }
	size_t alloc_words() const { return 2 * (color_words + 2 * (size_t)std::max(ghost_rows, 1) * (size_t)lld); }
	size_t tmp_words() const { return 2 * ((size_t)cfg.Y + 2) * (size_t)lld_dense; }
	uint64_t *tmp(int color) const { return d_tmp + (size_t)color * ((size_t)cfg.Y + 2) * lld_dense + lld_dense; }
	int nwc() const { return (gx + 3) / 4; } // ballot layout: wave columns per row
	size_t ham_words() const { return (size_t)cfg.Y * lld_packed; } // per coupling array, without its two halo rows
	// Rows each coupling array keeps above row 0 / below row Y-1: the halo row, or -- ring slab with ghost rows -- the couplings
	// of the ghost rows plus one row (they are generated in place like the slab's own, global row around the ring; the white
	// ones gather from the black rows one further out).  Fixed at creation.
	int ham_ghost = 1;
	size_t ham_alloc_words() const { return 2 * (ham_words() + 2 * (size_t)ham_ghost * lld_packed); }
	uint64_t *ham(int which) const { return d_ham + (size_t)which * (ham_words() + 2 * (size_t)ham_ghost * lld_packed) + (size_t)ham_ghost * lld_packed; }
	// "colour" 0/1 = spin arrays, 2 = black couplings; row stride and row count-words of that array
	uint64_t *plane(int kind) const { return kind == ISING_HAM_BLACK ? ham(0) : lat(kind); }
	int plane_ld(int kind) const { return kind == ISING_HAM_BLACK ? lld_packed : lld; }
	int layout() const { return ballot ? ISING_LAYOUT_BALLOT : (dense ? ISING_LAYOUT_DENSE : ISING_LAYOUT_NIBBLE); }
};

namespace ising_host {

// records the message for ising_last_error() and returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                                       \
	do {                                                                                                    \
		hipError_t e_ = (expr);                                                                             \
		if (e_ != hipSuccess) return ising_host::fail(ISING_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

int bind(const ising_ctx *c); // hipSetDevice(c->cfg.device)
// Number of 32-bit draws x for which curand_uniform(x) <= p (le) or < p (!le): the accepted draws form a prefix [0, N)
uint64_t draw_prefix(float p, bool le);

// true when the next update of this slab cannot use the integer-threshold kernels (and a ballot slab turns dense)
bool needs_generic(const ising_ctx *c);
// ballot -> dense for good, on the slab's stream
int ballot_leave(ising_ctx *c);
// ballot layout: refresh the dense-order image d_tmp (both colours, halo rows included)
int ballot_image(ising_ctx *c);
// ballot layout: allocate d_tmp; convert rows [row_lo, row_hi) of `color` between d_lat and d_tmp (rows -1 / Y = halo rows)
int ballot_tmp(ising_ctx *c);
int ballot_rows(ising_ctx *c, int color, long long row_lo, long long row_hi, bool to_dense);
void ballot_tmp_release(ising_ctx *c); // the image is held only while an observable that needs it runs (it doubles the slab's memory)
// ballot layout without sub-lattices: up spins and bond sum into d_acc[0], d_acc[1] on the slab's own words (ballot_measure_k)
bool ballot_native_observables(const ising_ctx *c);
int ballot_measure_into_acc(ising_ctx *c);
// makes the slab's stream (or stream `s`) wait until the halo rows of `color` delivered by the ring are in place
int halo_ready(ising_ctx *c, int color);
int halo_ready_on(ising_ctx *c, int color, hipStream_t s);
// one slab per process: the sum of a host value over all ranks (collective; also the ranks' barrier)
int rank_sum_u64(ising_ctx *c, unsigned long long mine, unsigned long long *sum);
// ising_update_edges on another stream of the slab's device / the interior rows 1 .. Y-2; `stop` fires when the launch is done
int update_edges_on(ising_ctx *c, int it, int color, hipStream_t s, hipEvent_t stop);
int update_interior(ising_ctx *c, int it, int color, hipEvent_t stop);
// ring slab with ghost rows: one fused launch of `nlevels` (even, <= ghost rows) colour half-sweeps, ghost rows included;
// `overlapped`: its edge strips wait for / announce the exchange themselves (ising_ring.cpp: sweep_deep_overlapped)
int update_deep(ising_ctx *c, int it, int nlevels, bool overlapped = false, int epochs = 1, bool split = false);
// called by ising_destroy
void ring_release(ising_ctx *c);
// a ring slab's second stream, events and counters (created once per slab)
int ring_resources(ising_ctx *c);
// a fused launch gave up: the slab's comm stream may hold kernels that wait for counters which will never move -- they see the
// abort word(s) and leave; returns once the comm stream is idle
void ring_abort_drain(ising_ctx *c);
// hipStreamSynchronize(c->stream), then: did a fused launch give up (completion counters that never came: UpdateParams.abort_flag)?
// If so its tickets, counters and their host-side bases start from zero again and ISING_E_STATE is returned.
int sync_checked(ising_ctx *c);
int cnt_reserve(ising_ctx *c, size_t strips, bool bonds, size_t *slots, size_t *n_up, size_t *chunk, unsigned long long **d_sum); // ising_update.cpp: the slots of in-launch counts (`bonds`: a third plane)
int check_abort(ising_ctx *c);
// the switches of DESIGN 8a from the environment (ISING_E_ARG for a value that means nothing)
int read_policy(ising_policy *pol);
// sweeps of a slab that needs nothing from its neighbours (a single slab that wraps in place, a slab of sub-lattices)
int sweep_alone(ising_ctx *c, int first_it, int nsweeps);
// sweeps one fused launch of `spins` spins (all lattices of a batch) carries at most: a launch costs ~60 us whatever it
// carries, so small lattices get long launches (ising_capi.cpp)
int fused_sweeps_per_launch(const ising_policy &pol, long long spins);
// launch shape of fused launches by tickets per level (ising_capi.cpp)
void fused_shape(int nwc, int Y, long long rows, int *H, int *wg_per_cu, bool late = false);
// the quad path's tile and pass by width and rows (ising_capi.cpp)
void quad_shape(int gx, long long rows, int *C, int *T, int *W);
// the slab sweeps on the quad path as things stand (its temperature has integer thresholds, its buffers exist)
bool quad_ready(const ising_ctx *c);
// the passes of a call on the quad path: iteration of the first sweep, sweeps, index of the print point the pass ends on (-1: none)
struct QuadPass { int it, ns, meas; };
std::vector<QuadPass> quad_passes(int first_it, int nsweeps, int every, int T, int *nmeas);
constexpr size_t SLOTCTL_TICKET_BYTES = 9 * 64; // ticket words in front of the completion counters (d_slotctl)

} // namespace ising_host
