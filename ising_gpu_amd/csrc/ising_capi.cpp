// ising_capi.cpp -- the C-ABI of libising_hip.so (see include/ising_hip.h for the reference file:line each entry point
// replaces): the context -- creation, the launch-shape policy, tables, streams, initialisation.  The update calls are in
// ising_update.cpp, the observables in ising_observe.cpp, the -J coupling arrays in ising_couplings.cpp, boundary formats and
// checkpoints in ising_io.cpp, the slab ring in ising_ring.cpp / ising_ipc.cpp, batches in ising_batch.cpp.  Host side only:
// owns device memory, tables and launch order; all arithmetic on the lattice happens in the .hip files.
#include "ising_ctx.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
thread_local std::string g_err;
}

namespace ising_host {

int fail(int code, const char *fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_err = buf;
	return code;
}

int bind(const ising_ctx *c) {
	HIP_TRY(hipSetDevice(c->cfg.device));
	return ISING_OK;
}

// every switch the library takes from the environment, read in one place
int read_policy(ising_policy *pol) {
	auto num = [](const char *name, int *out) { if (const char *e = getenv(name)) { *out = atoi(e); return true; } return false; };
	int v = 0;
	if (num("ISING_FUSED", &v)) pol->fused = v != 0;
	if (num("ISING_FUSED_NT", &v)) pol->fused_nt = v != 0;
	if (num("ISING_FUSED_TICKETS2", &v)) pol->fused_tickets2 = (v == 2 || v == 4) ? v : (v ? 2 : 0);
	if (num("ISING_FUSED_WGS", &v)) pol->fused_wgs = v > 0 ? v : 0;
	if (num("ISING_FUSED_MAX_SWEEPS", &v)) pol->fused_max_sweeps = v > 0 ? v : 0;
	if (num("ISING_GUARD", &v)) pol->guard = v != 0;
	if (const char *e = getenv("ISING_GUARD_EXPECT")) { const double x = atof(e); if (x > 0) pol->guard_expect = (float)x; }
	if (num("ISING_FUSED_WAIT_LATE", &v)) pol->fused_wait_late = v < 0 ? 0 : (v > 2 ? 2 : v);
	if (num("ISING_SPLIT", &v)) pol->split = v != 0;
	if (num("ISING_SPLIT_LEAD", &v)) pol->split_lead = v < 0 ? 0 : (v > 4 ? 4 : v);
	if (num("ISING_RING_GHOST", &v)) pol->ring_ghost = v;
	if (num("ISING_RING_EPOCHS", &v)) pol->ring_epochs = v < 0 ? 0 : (v > 1024 ? 1024 : v);
	if (num("ISING_RING_COUNTED", &v)) pol->ring_counted = v < 0 ? 0 : (v > 2 ? 2 : v);
	if (num("ISING_TILES", &v)) pol->tiles = v != 0;
	if (num("ISING_TILE_ROWS", &v) && v > 0) pol->tile_rows = v;
	if (num("ISING_TILE_WORDS", &v) && v > 0) pol->tile_words = v;
	if (num("ISING_TILE_SWEEPS", &v) && v > 0) pol->tile_sweeps = v;
	if (num("ISING_TILE_THREADS", &v) && v > 0) pol->tile_threads = v;
	if (num("ISING_TILE_XCD", &v)) pol->tile_xcd = v != 0;
	if (num("ISING_QUAD", &v)) pol->quad = v != 0;
	if (num("ISING_QUAD_C", &v) && v > 0) pol->quad_C = v;
	if (num("ISING_QUAD_T", &v) && v > 0) pol->quad_T = v;
	if (num("ISING_QUAD_WAVES", &v) && v > 0) pol->quad_waves = v;
	pol->no_ballot = getenv("ISING_NO_BALLOT") != nullptr;
	if (const char *e = getenv("ISING_TAIL")) {
		int rows = 0, h = 1;
		const int got = sscanf(e, "%d,%d", &rows, &h);
		if (got >= 1 && rows >= 0 && h > 0) { pol->tail_rows = rows; pol->tail_h = h; }
	}
	if (num("ISING_RING_TRAPEZOID", &v)) pol->trapezoid = v != 0;
	if (num("ISING_RING_OVERLAP", &v)) pol->overlap = v < 0 ? 0 : (v > 2 ? 2 : v);
	if (num("ISING_RING_INLINE", &v)) pol->ring_inline = v != 0;
	if (num("ISING_RING_STORE", &v)) pol->ring_store = v != 0;
	if (num("ISING_RING_COMM_PRIORITY", &v)) pol->comm_priority = v != 0;
	if (num("ISING_ABORT_POLLS", &v) && v > 0) pol->abort_polls = (uint32_t)v;
	if (const char *e = getenv("ISING_RING_TRANSPORT")) {
		if (!strcmp(e, "copy")) pol->ring_transport = ISING_TRANSPORT_COPY;
		else if (!strcmp(e, "rccl")) pol->ring_transport = ISING_TRANSPORT_RCCL;
		else if (!strcmp(e, "auto")) pol->ring_transport = ISING_TRANSPORT_AUTO;
		else return fail(ISING_E_ARG, "ISING_RING_TRANSPORT must be auto, copy or rccl (got %s)", e);
	}
	return ISING_OK;
}

// A fused launch that gave up (ising_ballot.hip: UpdateParams.abort_flag) leaves tickets half drawn and counters half
// bumped: everything starts from zero again -- the spins are whatever the launch left, the caller initialises or loads.
int check_abort(ising_ctx *c) {
	if (!c->h_abort || !__atomic_load_n(c->h_abort, __ATOMIC_ACQUIRE)) return ISING_OK;
	(void)hipStreamSynchronize(c->stream);
	ising_host::ring_abort_drain(c); // (while the word is still set: the comm stream's waiting kernels leave)
	if (c->d_slotctl) (void)hipMemset(c->d_slotctl, 0, c->slotctl_bytes);
	if (c->d_edge) (void)hipMemset(c->d_edge, 0, 32 * sizeof(uint32_t));
	for (auto &t : c->ticket_base2) t = 0;
	c->done_base = c->split_done_base = 0;
	c->edge_done_target = c->edge_go_epoch = 0;
	c->go_set = false;
	c->ghost_depth[0] = c->ghost_depth[1] = 0;
	// A split launch whose word units never got their masks: a ticket class that no resident workgroup serves (a device or partition that does not run
	// workgroups on all eight XCC ids, a CU mask) -- the plain fused form carries this slab's sweeps from here on.
	if (c->last_launch_split) { c->split = c->split_always = false; c->last_launch_split = false; }
	__atomic_store_n(c->h_abort, 0u, __ATOMIC_RELEASE);
	return fail(ISING_E_STATE, "a fused launch gave up: its units' parents never completed (completion counters out of step with the device after a faulted "
	                           "launch?); tickets and counters have been reset, the lattice is undefined -- initialise or load it again");
}

int sync_checked(ising_ctx *c) {
	HIP_TRY(hipStreamSynchronize(c->stream));
	return check_abort(c);
}

} // namespace ising_host


// every switch the library and its Python mirror take from the environment -- one table, printed by ising_switch_table, kept as docs/SWITCHES.md, and held
// against the sources by tests/test_switches.py (a getenv the table does not know fails the suite).
namespace {
struct SwitchDoc { const char *name, *values, *meaning; };
const SwitchDoc kSwitches[] = {
	{"ISING_FUSED", "0/1", "`ising_sweep` on the ballot layout: one launch per colour / fused launches (default: fused from 1.5*2^24 spins, and below where a level of one-row units feeds two workgroups per CU)"},
	{"ISING_FUSED_WGS", "n", "size of a fused launch's persistent grid, in workgroups (default: by tickets per level, strip height and wave columns: `fused_wgs_for`)"},
	{"ISING_GUARD", "0/1", "run-time guard under the fused launches' shape table: the first launches of a lone slab are timed; below the plateau (under 4096 tickets a level), or under 0.8 x the expected rate, the neighbouring shapes (one workgroup per CU fewer / more, half the strip height) get a launch each and the fastest stays if it is worth 3 %; where the table gives long calls the split form, one split and one fused launch are timed and the faster form stays (default: on a whole MI355X)"},
	{"ISING_TEST_RING_CROSSCHECK_PERTURB", "1", "test aid (Python mirror, `open_native_ring`): the first rank transport's whole-lattice counts are made to differ, so that the cross-check against the second transport has something to catch"},
	{"ISING_GUARD_EXPECT", "flips/ns", "what the guard expects of the lattice (default: by its size, `guard_expected`)"},
	{"ISING_FUSED_TICKETS2", "0/2/4", "fused launches draw from that many ticket counters (default: four for one-row units, two for two-row units, one above)"},
	{"ISING_FUSED_NT", "0/1", "lattice words of fused launches with the non-temporal hint (default: lattices above 2^31 spins)"},
	{"ISING_FUSED_WAIT_LATE", "0/1/2", "units of fused launches draw that many rows before they wait for their parents (default 2)"},
	{"ISING_FUSED_MAX_SWEEPS", "n", "sweeps one fused launch of a single slab or a batch carries at most (default: ~50 ms worth, 32 .. 4096)"},
	{"ISING_SPLIT", "0/1", "fused launches in the split form (draw units / word units, `ballot_split_k`): never / every launch of several levels wherever the form applies (default: lone slabs whose sixteen-row strips make under 2048 tickets a level, on calls of 2^35 flips and more; eight XCC ids required either way)"},
	{"ISING_SPLIT_LEAD", "0..4", "draw units a workgroup of a split launch does before its first word unit (default 1)"},
	{"ISING_QUAD", "0/1", "lone lattices of one to eight blocks of 2048 columns on the quad path (`quad_pass_k`): never / wherever it applies (default: the rule by width and rows in `ising_create`, on a whole MI355X)"},
	{"ISING_QUAD_C", "n", "quad path: row groups (4 rows each) per tile (default: the shape table in `ising_create`)"},
	{"ISING_QUAD_T", "n", "quad path: sweeps per pass, 1 .. 32 (default: the shape table)"},
	{"ISING_QUAD_WAVES", "n", "quad path: waves per workgroup, 1 .. 16 (default: the shape table)"},
	{"ISING_TILES", "0/1", "lone slabs on the dense layout sweep in tile launches (`dense_tile_k`): never / always (default: up to 2^24 spins where the quad path does not apply)"},
	{"ISING_TILE_ROWS", "n", "tile launches: rows per tile (default: the shape rule of `ising_create`)"},
	{"ISING_TILE_WORDS", "n", "tile launches: 32-bit words per tile row"},
	{"ISING_TILE_SWEEPS", "n", "tile launches: sweeps per launch (default 3 / 4 / 6 by tile rows)"},
	{"ISING_TILE_THREADS", "256/512/1024", "tile launches: workgroup size"},
	{"ISING_TILE_XCD", "0/1", "tile launches: tiles in bands per XCD (default on where the tiles divide by eight)"},
	{"ISING_TAIL", "rows[,h]", "one-row tail strips of the one-launch-per-colour form (`0`: off; default: two thirds of a strip per workgroup slot)"},
	{"ISING_NO_BALLOT", "(set)", "layout AUTO never picks the ballot layout"},
	{"ISING_RING_GHOST", "n", "ghost rows of ballot ring slabs (default 64; `1`: one halo row and the per-colour schedules)"},
	{"ISING_RING_EPOCHS", "n", "exchange epochs (of `ISING_RING_GHOST`/2 sweeps each) that ONE persistent launch of a ring slab carries, the exchanges running next to it on the comm stream (IPC transport, a device per rank, fused form; default: ~50 ms worth, at most 64; `1`: a launch per exchange, as before round 6 and as RCCL and peer copies still do)"},
	{"ISING_RING_COUNTED", "0/1/2", "print points of rings never / where possible (default) / always (an error where not) inside the deep launches"},
	{"ISING_RING_OVERLAP", "0/1/2", "the deep exchange between two launches / in the tail of the running launch, the next one waits for it on the stream (default) / free-running, only the next launch's edge units wait (copies and IPC only)"},
	{"ISING_RING_TRAPEZOID", "0/1", "0: every ghost row at every level (default 1: a level touches only the ghost rows that can still reach the slab)"},
	{"ISING_RING_TRANSPORT", "auto/copy/rccl", "a single-process ring's transport"},
	{"ISING_RCCL_LIB", "path", "the librccl to open (process-wide, read when RCCL is first needed)"},
	{"ISING_RING_STORE", "0/1", "slabs sharing one device: never / always store edge rows straight into the neighbours' halo rows (default: only slabs without ghost rows do)"},
	{"ISING_RING_INLINE", "0/1", "peer copies on the comm streams / on the compute stream (default: by device placement)"},
	{"ISING_RING_COMM_PRIORITY", "0/1", "0: comm streams at default priority (default 1: high priority)"},
	{"ISING_ABORT_POLLS", "n", "polls after which a unit of a fused launch gives its parents up (default 2^22, ~10 s)"},
	{"ISING_LIB", "path", "(Python mirror) another build of libising_hip.so, for measurement builds"},
	{"ISING_HIP_RUNTIME", "auto/system", "(Python mirror) `system`: do not preload torch's HIP runtime before the library"},
	{"ISING_RING_EXCHANGE", "p2p/allgather", "(Python mirror, torch.distributed ring) how the edge rows travel"},
};
} // namespace

extern "C" int ising_switch_table(char *buf, size_t len, size_t *needed) {
	std::string out = "| Variable | Values | Meaning |\n|---|---|---|\n";
	for (const SwitchDoc &d : kSwitches) out += std::string("| `") + d.name + "` | " + d.values + " | " + d.meaning + " |\n";
	if (needed) *needed = out.size() + 1;
	if (buf && len) { const size_t n = std::min(len - 1, out.size()); memcpy(buf, out.data(), n); buf[n] = 0; }
	return (buf && len > out.size()) || !buf ? ISING_OK : ising_host::fail(ISING_E_ARG, "ising_switch_table: %zu bytes needed", out.size() + 1);
}

using ising_host::bind;
using ising_host::fail;

namespace {

using ising_host::SLOTCTL_TICKET_BYTES;

// cuRAND's curand_uniform on the host: x*2^-32 + 2^-33 in FP32 (product exact, one rounding).
inline float u01(uint32_t x) { return (float)x * 0x1p-32f + 0x1p-33f; }

} // namespace

// Number of 32-bit draws x for which curand_uniform(x) <= p (le) or < p (!le).  curand_uniform is monotone
// non-decreasing in x, so the accepted draws form a prefix [0, N) and N is found by bisection over the exact
// FP32 formula.  NaN p accepts nothing, exactly like the FP32 comparison it replaces.
uint64_t ising_host::draw_prefix(float p, bool le) {
	auto ok = [&](uint32_t x) { const float u = u01(x); return le ? (u <= p) : (u < p); };
	if (!ok(0u)) return 0;
	if (ok(0xFFFFFFFFu)) return 1ull << 32;
	uint32_t lo = 0u, hi = 0xFFFFFFFFu; // ok(lo), !ok(hi)
	while (hi - lo > 1u) {
		const uint32_t mid = lo + (hi - lo) / 2u;
		if (ok(mid)) lo = mid; else hi = mid;
	}
	return hi;
}


bool ising_host::needs_generic(const ising_ctx *c) {
	return c->cfg.kernel == ISING_KERNEL_GENERIC || (c->cfg.kernel == ISING_KERNEL_AUTO && !c->fast_ok);
}

namespace {

// exp table exactly as optimized/main.cu:1684-1697 evaluates it (FP32, left to right), then the integer
// thresholds per number of aligned neighbours.
void compute_tables(ising_ctx *c, float temp) {
	for (int i = 0; i < 2; i++) {
		for (int j = 0; j < 5; j++) {
			if (temp > 0) {
				c->tab[i * 5 + j] = expf((i ? -2.0f : 2.0f) * static_cast<float>(j * 2 - 4) * (1.0f / temp));
			} else {
				c->tab[i * 5 + j] = (j == 2) ? 0.5f : (i ? -2.0f : 2.0f) * static_cast<float>(j * 2 - 4);
			}
		}
	}
	bool symmetric = true;
	for (int a = 0; a < 5; a++) {
		const uint64_t up = ising_host::draw_prefix(c->tab[5 + a], true);       // spin up, n = a neighbours up
		const uint64_t dn = ising_host::draw_prefix(c->tab[0 + (4 - a)], true); // spin down, 4-a neighbours up
		c->thr[a] = up;
		if (up != dn) symmetric = false;
	}
	const uint64_t always = 1ull << 32;
	c->fast_ok = symmetric && c->thr[0] == always && c->thr[1] == always && c->thr[2] == always &&
	             c->thr[3] < always && c->thr[4] <= c->thr[3];
	c->cfg.temp = temp;
}

// Fused launches: T = tickets (workgroups) per level for strips of H rows; and the launch shape ising_create picks.
long long fused_tickets(int nwc, int Y, int H) { return ((long long)nwc * ((Y + H - 1) / H) + 3) / 4; }
// Workgroups of 4 waves per CU a level of T tickets carries without its units running into unfinished parents (with the waves'
// rotating priorities, ising_ballot.hip: a level wants 1.33 x the grid at three per CU, 2 x at four, 3.2 x at five, 5.3 x at
// six -- 65536^2, T = 8192: 3512 flips/ns with five, 3533 with six; 131072 x 16384, T = 4096 at H = 16: 3504 with five, 3373
// with six; 16384^2, T = 2048 at H = 4: 3285 with four, 3217 with five; 16384 x 8192, T = 1024: 3038 with three, 2858 with four).
// The sixth workgroup per CU waits for 16384 tickets: at 8192 (65536^2) it is worth 0.6 % (3503 -> 3526) and costs 2.7 % more HBM
// traffic (0.892 -> 0.916 GB per colour half-sweep: 20 % more accept-mask slots in the L2s, more of them written back).
// (round 3: the sixth from 8192 tickets -- the throughput is what counts, and the accept-mask slots' extra traffic is far from any
// limit: 65536^2 3500 -> 3525 flips/ns, --steps 20 --warmup 5 3490 -> 3510, three alternating runs each on one box.  Ring slabs keep
// five below 16384 tickets: their transport's kernels want room next to the launch, RCCL's 132 vector registers per lane.)
// Round 4: units draw their first TWO rows before they wait for their parents (UpdateParams.wait_late = 2), so a parent that is up to two
// rows' time late costs nothing and a level of few tickets feeds more workgroups -- the more the shorter its units are: units of one
// or two rows have drawn everything they will before they wait (tools/wait_late_probe.py, profiles/wait_late_probe_r04.txt and
// wait_late2_probe_r04.txt; flips/ns at the rule below against rounds 2-3's: 8192 x 4096 2820 (H = 2, T = 512) vs 2305 (H = 1), 8192^2
// 3087 vs 2767, 16384 x 8192 3246 vs 3051, 16384^2 3330 vs 3298; from T = 4096 up nothing moves, 65536^2 3532.8 vs 3531.8).
// `late` = false gives the rule of rounds 2-3 (ISING_FUSED_WAIT_LATE=0).
// One-row units (the end of round 4, tools/small_fused_probe.py, profiles/small_fused_probe_r04.txt): a grid of G workgroups per CU runs at its own
// plateau (2: 1720 flips/ns, 3: 2150, 4: 2500) as long as a level has r x G x 256 tickets, and falls off a cliff below (8192 x 1536, T = 384: 1355 with three,
// 1699 with two) -- r grows with the wave columns of a row: 0.54 (1), 0.68 (2), 0.86 (3), 0.90 (4), ~1.9 (8: 65536 x 384 ran 554 with the four
// the rule used to give it, 1451 with two).
double fused_level_ratio(int nwc) {
	static const double R[5] = {0.54, 0.54, 0.68, 0.86, 0.90};
	return nwc <= 4 ? R[nwc < 1 ? 1 : nwc] : 0.90 + 0.25 * (nwc - 4);
}
int fused_wgs_cap(long long T, int nwc) {
	int cap = (int)std::min<long long>(6, (long long)((double)T / (fused_level_ratio(nwc) * 256.0)));
	if (cap >= 5 && nwc >= 3 && T < 256 * cap) cap = 4; // (five and six per CU want a level of their own size on wide rows: 32768 x 1152 2143 with five, 2428 with four)
	return cap;
}
int fused_wgs_for(long long T, int H = 16, bool late = false, int nwc = 0) {
	const int base = T >= 8192 ? 6 : (T >= 4096 ? 5 : (T >= 2048 ? 4 : (T >= 1024 ? 3 : (T >= 684 ? 2 : 1))));
	if (!late || T >= 4096) return base;
	if (H == 1 && nwc > 0) return std::min(T >= 2048 ? 6 : (T >= 1024 ? 5 : 4), std::max(2, fused_wgs_cap(T, nwc)));
	const int v = H <= 2 ? (T >= 2048 ? 6 : (T >= 1024 ? 5 : (T >= 512 ? 4 : base))) : (T >= 1024 ? base + 1 : base);
	if (nwc <= 1) return v;
	// taller units, rows of several wave columns (tools/wide_probe.py, profiles/wide_probe_r04.txt): the same cliff, at r = 0.75 (2 wave columns), 0.95 (3), 1.0 (4 .. 15),
	// 1.33 (16 and more) -- 131072 x 1024 (H = 4, T = 1024) 2145 with four per CU, 2995 with three; 65536 x 1024 (H = 2) 1934 with five, 2295 with four; 32768 x 2048 2715 / 2848
	// (two-row units get by on less: 16384 x 2176, T = 544: 2689 with three per CU; 24576 x 1536 and 32768 x 1152, T = 576: 2626 / 2583 with three, off the cliff with four)
	const double r = H <= 2 ? (nwc >= 16 ? 1.33 : (nwc >= 5 ? 1.1 : (nwc == 4 ? 0.80 : (nwc == 3 ? 0.75 : 0.70))))
	                        : (nwc >= 16 ? 1.33 : (nwc >= 4 ? 1.0 : (nwc == 3 ? 0.95 : 0.75)));
	int cap = (int)((double)T / (r * 256.0));
	if (cap >= 5 && nwc >= 3 && T < 256 * cap) cap = 4; // (as for one-row units: 32768 x 2048, T = 1024: 2848 with four per CU, 2646 with five)
	if (cap >= 5 && nwc >= 16 && 4 * T < 7 * 256 * cap) cap = 4; // (rows of 16 wave columns: 131072 x 2048, T = 2048: 3219 with four, 2923 with five)
	return std::min(v, std::max(2, cap));
}
// flips/ns of strips of H rows at wg workgroups per CU where T is ample (tools/grid_probe2.py on 65536^2 .. 131072^2, 24576^2,
// 32768 x 16384, 16384^2, 8192^2 at the end of round 2).  One- and two-row units draw tickets from several counters.
int fused_score(int H, int wg) {
	static const int S[5][7] = {                // wg:  1     2     3     4     5     6
		/* H = 1  */ {0, 900, 1850, 2300, 2650, 2640, 2560},
		/* H = 2  */ {0, 1200, 2270, 2760, 2690, 2680, 2680},
		/* H = 4  */ {0, 1300, 2500, 3060, 3290, 3375, 3415},
		/* H = 8  */ {0, 1400, 2650, 3200, 3385, 3460, 3498},
		/* H = 16 */ {0, 1450, 2740, 3260, 3440, 3512, 3535}};
	const int h = H >= 16 ? 4 : (H >= 8 ? 3 : (H >= 4 ? 2 : (H >= 2 ? 1 : 0)));
	return S[h][wg < 1 ? 1 : (wg > 6 ? 6 : wg)];
}
// the strip height whose level still feeds the most productive grid (`rows`: the rows a launch covers, ghost rows included)
int choose_fused_strip_rows(int nwc, int Y, int rows) {
	int best = 1, best_score = -1;
	for (int H = 1; H <= 16; H <<= 1) {
		if (Y % H) break;
		const int score = fused_score(H, fused_wgs_for(fused_tickets(nwc, rows, H)));
		if (score >= best_score) { best = H; best_score = score; }
	}
	return best;
}

} // namespace

// A fused launch has a fixed cost (the grid's staggered start, and a tail in which the last units finish unevenly and the chip
// drains before the next launch may start): tools/launch_len_probe.py, 32 against 1024 sweeps per launch: 8192 x 4096 2285 ->
// 2332 flips/ns, 8192^2 2763 -> 2801, 16384 x 8192 3055 -> 3086, 16384^2 3293 -> 3318, 32768^2 3471 -> 3483, 65536^2 3512 -> 3518
// (~20 us per launch).  Single slabs and batches therefore carry ~50 ms of sweeps per launch (at 3.4 flips/ns), between 32 and
// 4096 of them -- a ring slab's launches are tied to its ghost rows (32 sweeps).  The counters a launch moves stay far from
// wrapping: 4096 sweeps x 127 wave columns = 2^20 per strip.
int ising_host::fused_sweeps_per_launch(const ising_policy &pol, long long spins) {
	if (pol.fused_max_sweeps > 0) return std::min(pol.fused_max_sweeps, 4096);
	const double per_sweep_ms = (double)spins / 3.4e9; // (3.4 flips/ns = 3.4e9 per ms)
	const long long n = (long long)(50.0 / std::max(per_sweep_ms, 1e-6));
	return (int)std::min<long long>(4096, std::max<long long>(32, n / 32 * 32));
}

// strip height and workgroups per CU of fused launches over `rows` rows of wave columns in all (ising_batch.cpp: the rows of
// every lattice of a batch), strips dividing Y
void ising_host::fused_shape(int nwc, int Y, long long rows, int *H, int *wg_per_cu, bool late) {
	int best = 1, best_score = -1;
	for (int h = 1; h <= 16; h <<= 1) {
		if (Y % h) break;
		const long long T = ((long long)nwc * ((rows + h - 1) / h) + 3) / 4;
		const int score = fused_score(h, fused_wgs_for(T));
		if (score >= best_score) { best = h; best_score = score; }
	}
	*H = best;
	*wg_per_cu = fused_wgs_for(((long long)nwc * ((rows + best - 1) / best) + 3) / 4, best, late, nwc);
}

namespace {

int choose_strip_rows(int gx, int Y, bool dense, bool ballot = false) {
	// Enough (column-group x strip) units to give every SIMD several waves, while keeping strips tall so the two
	// halo rows per strip stay a small fraction of the source traffic (measured optimum: 32 rows for the nibble
	// layout, 8-16 for the dense one, whose traffic is 4x smaller, 8 for the ballot one).
	const long long want_units = 4LL * 8192; // 4 units per wave, ~8 waves on each of 1024 SIMDs
	// ballot layout (launches end on one-row tail strips): 16 rows while that still makes ~2.5 rounds of workgroups (16
	// units each, 1536 resident), else 8 -- measured +0.6..1.3 % from 2^31 spins up, -0.4..-2 % below (tools/big_probe.py)
	if (ballot && (Y % 16) == 0 && (long long)gx * (Y / 16) >= 61440) return 16;
	int H = ballot ? 8 : (dense ? 16 : 32);
	while (H > 1 && ((Y % H) != 0 || (long long)gx * (Y / H) < want_units)) H >>= 1;
	return H;
}

} // namespace

// The quad path's shape by width and rows (row groups per tile, sweeps per pass, waves per workgroup), by measurement: ising_create's comment at its call.
// `rows`: of the lattice -- of ALL lattices of a batch (ising_batch.cpp): many tiles are many tiles whoever owns them.
void ising_host::quad_shape(int gx, long long rows, int *C, int *T, int *W) {
	int C0 = 4, T0 = 8, W0 = 12;
	if (gx == 1) {
		if (rows < 1024) { C0 = 4; T0 = 8; W0 = 12; } // (2048 x 512: 745 against 670 at sixteen sweeps a pass)
		else if (rows < 2048) { C0 = 4; T0 = 16; W0 = 12; }
		else if (rows < 4096) { C0 = 8; T0 = 16; W0 = 12; }
		else if (rows < 8192) { C0 = 4; T0 = 12; W0 = 8; }
		else { C0 = 8; T0 = 8; W0 = 8; } // (round 6, two sets of masks in flight: 2048 x 8192 2553 against (4, 8, 8) 2340, 2048 x 16384 2601 against (8, 4, 8) 2462)
	} else if (gx == 2) {
		if (rows < 2048) { C0 = 4; T0 = 12; W0 = 16; }
		else if (rows < 8192) { C0 = 4; T0 = 8; W0 = 12; }
		else { C0 = 8; T0 = 4; W0 = 12; }
	} else if (gx == 3) { // (from three blocks on: up to three items a wave of sixteen -- 100 registers, four waves per SIMD: a tile has its CU to itself)
		if (rows < 1024) { C0 = 2; T0 = 8; W0 = 16; }       // 6144 x 512: 1361 against (4, 4, 16) 1076
		else if (rows < 2048) { C0 = 4; T0 = 8; W0 = 16; }  // 6144 x 1024: 1984 against (4, 4, 12) 1670
		else if (rows < 4096) { C0 = 8; T0 = 6; W0 = 16; }  // 6144 x 2048: 2057 against 1951 (round 6: (2, 6, 12) 2100, inside the noise of two boxes)
		else { C0 = 4; T0 = 4; W0 = 12; }
	} else if (gx == 4) {
		if (rows < 1024) { C0 = 2; T0 = 8; W0 = 16; }       // 8192 x 512: 1496 against 1231
		else if (rows < 2048) { C0 = 4; T0 = 8; W0 = 16; }  // 8192 x 1024: 2043 against 1642
		else { C0 = 4; T0 = 4; W0 = 16; }
	} else if (gx <= 6) {
		if (rows < 768) { C0 = 2; T0 = 6; W0 = 16; }        // 10240 x 512: 1582, 12288 x 512: 1807 against (2, 2, 12) 1161 / 1229 (round 6: 12288 x 768 (4, 4, 16) 1797 against 1668)
		else { C0 = 4; T0 = 4; W0 = 16; }                     // 10240 x 1024: 1748, 12288 x 1024: 1904 against 1427 / 1466
	} else {
		C0 = 2; T0 = 4; W0 = 16;                               // 14336 x 512: 1592, 16384 x 512: 1747 against (2, 2, 16) 1293 / 1313
	}
	*C = C0; *T = T0; *W = W0;
}

extern "C" {

const char *ising_last_error(void) { return g_err.c_str(); }

int ising_device_count(int *count) {
	if (!count) return fail(ISING_E_ARG, "count is null");
	hipError_t e = hipGetDeviceCount(count);
	if (e != hipSuccess) { *count = 0; return fail(ISING_E_NOGPU, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
	return ISING_OK;
}

int ising_device_info(int device, char *name, size_t name_len, int *cus, int *max_threads_per_cu, int *major, int *minor) {
	hipDeviceProp_t p;
	HIP_TRY(hipGetDeviceProperties(&p, device));
	if (name && name_len) {
		const char *n = p.name[0] ? p.name : p.gcnArchName;
		snprintf(name, name_len, "%s", n);
	}
	if (cus) *cus = p.multiProcessorCount;
	if (max_threads_per_cu) *max_threads_per_cu = p.maxThreadsPerMultiProcessor;
	if (major) *major = p.major;
	if (minor) *minor = p.minor;
	return ISING_OK;
}

int ising_device_peer_access(int device, int peer, int *can_access) {
	if (!can_access) return fail(ISING_E_ARG, "can_access is null");
	*can_access = 0;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ISING_E_NOGPU, "no HIP device visible");
	if (device < 0 || device >= ndev || peer < 0 || peer >= ndev) return fail(ISING_E_ARG, "devices %d, %d out of range (%d visible)", device, peer, ndev);
	if (device == peer) { *can_access = 1; return ISING_OK; }
	HIP_TRY(hipDeviceCanAccessPeer(can_access, device, peer));
	return ISING_OK;
}

// the clock a kernel ran at from the marks its first workgroups left: {cycles, 100 MHz ticks} x {start, end} per XCD
static void clock_from_marks(const unsigned long long *m, int n, double *mean, double *lo, double *hi) {
	double sum = 0, mn = 0, mx = 0;
	int got = 0;
	for (int k = 0; k < n; k++) {
		const unsigned long long c0 = m[4 * k], r0 = m[4 * k + 1], c1 = m[4 * k + 2], r1 = m[4 * k + 3];
		if (c1 <= c0 || r1 <= r0) continue; // (a workgroup that never ran, or marks of two different launches)
		const double mhz = (double)(c1 - c0) / (double)(r1 - r0) * 100.0;
		sum += mhz;
		mn = got ? std::min(mn, mhz) : mhz;
		mx = got ? std::max(mx, mhz) : mhz;
		got++;
	}
	if (mean) *mean = got ? sum / got : 0.0;
	if (lo) *lo = mn;
	if (hi) *hi = mx;
}

int ising_philox_ceiling_clocked(int device, double min_ms, double *sites_per_ns, double *sclk_mhz) {
	if (!sites_per_ns) return fail(ISING_E_ARG, "null argument");
	HIP_TRY(hipSetDevice(device));
	int cus = 0;
	HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
	const int blocks = cus * 32, nrows = 64; // 8 waves per SIMD over four rounds of blocks; ~8.6e9 sites, ~2 ms per launch
	uint32_t *out = nullptr;
	unsigned long long *d_clk = nullptr;
	HIP_TRY(hipMalloc((void **)&out, (size_t)blocks * 256 * sizeof(uint32_t)));
	hipError_t e = hipMalloc((void **)&d_clk, 32 * sizeof(unsigned long long));
	if (e == hipSuccess) e = hipMemset(d_clk, 0, 32 * sizeof(unsigned long long));
	hipEvent_t e0 = nullptr, e1 = nullptr;
	if (e == hipSuccess) e = hipEventCreate(&e0);
	if (e == hipSuccess) e = hipEventCreate(&e1);
	double rate = 0, clock = 0;
	int n = 2; // two warm launches set the scale; then one timed batch of launches back to back that lasts min_ms
	for (int pass = 0; pass < 2 && e == hipSuccess; pass++) {
		e = hipEventRecord(e0, nullptr);
		for (int k = 0; k < n && e == hipSuccess; k++) e = ising::launch_philox_ceiling(out, blocks, nrows, nullptr, (pass == 1 && k == n - 1) ? d_clk : nullptr);
		if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
		if (e == hipSuccess) e = hipEventSynchronize(e1);
		float ms = 0;
		if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
		if (e != hipSuccess || ms <= 0) break;
		if (pass == 0) n = std::max(2, std::min(4096, (int)std::ceil(min_ms / (ms / n))));
		else rate = (double)n * blocks * 256.0 * nrows * 64.0 / (ms * 1e6);
	}
	unsigned long long marks[32];
	if (e == hipSuccess) e = hipMemcpy(marks, d_clk, sizeof(marks), hipMemcpyDeviceToHost);
	if (e == hipSuccess) clock_from_marks(marks, 8, &clock, nullptr, nullptr);
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	(void)hipFree(out);
	if (d_clk) (void)hipFree(d_clk);
	if (e != hipSuccess) return fail(ISING_E_HIP, "philox ceiling probe failed: %s", hipGetErrorString(e));
	*sites_per_ns = rate;
	if (sclk_mhz) *sclk_mhz = clock;
	return ISING_OK;
}

// (the form of rounds 1-4, kept for its callers: the average over ~20 ms of launches)
int ising_philox_ceiling(int device, double *sites_per_ns) { return ising_philox_ceiling_clocked(device, 20.0, sites_per_ns, nullptr); }

int ising_kernel_clock(ising_ctx *c, int enable) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (int rc = bind(c)) return rc;
	if (enable && !c->d_clk) HIP_TRY(hipMalloc((void **)&c->d_clk, 32 * sizeof(unsigned long long)));
	if (enable) HIP_TRY(hipMemsetAsync(c->d_clk, 0, 32 * sizeof(unsigned long long), c->stream));
	c->clk_on = enable != 0;
	return ISING_OK;
}

int ising_kernel_clock_fetch(ising_ctx *c, double *mean, double *lo, double *hi) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (!c->d_clk) return fail(ISING_E_STATE, "ising_kernel_clock was never enabled on this slab");
	if (int rc = bind(c)) return rc;
	if (int rc = ising_host::sync_checked(c)) return rc;
	unsigned long long marks[32];
	HIP_TRY(hipMemcpy(marks, c->d_clk, sizeof(marks), hipMemcpyDeviceToHost));
	double m = 0;
	clock_from_marks(marks, 8, &m, lo, hi);
	if (m <= 0) return fail(ISING_E_STATE, "no fused launch has left its clock marks (ising_sweep's launch form: ising_sweep_info)");
	if (mean) *mean = m;
	return ISING_OK;
}

size_t ising_required_bytes(int32_t X, int32_t Y) {
	if (X <= 0 || Y <= 0) return 0;
	return 2 * ((size_t)Y + 2) * (size_t)(X / 32) * sizeof(uint64_t);
}

size_t ising_required_bytes_layout(int32_t X, int32_t Y, int32_t layout) {
	const size_t full = ising_required_bytes(X, Y);
	return layout == ISING_LAYOUT_NIBBLE ? full : full / 4; // (caller-owned buffers never hold padded ballot rows)
}

// A caller-owned buffer must be device memory of the slab's device and, when the caller states its size, large enough.
static int check_caller_buffer(const void *ptr, size_t have, size_t need, int device, const char *what) {
	hipPointerAttribute_t attr;
	memset(&attr, 0, sizeof(attr));
	const hipError_t e = hipPointerGetAttributes(&attr, ptr);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		return fail(ISING_E_ARG, "%s (%p) is not a HIP allocation: %s", what, ptr, hipGetErrorString(e));
	}
	if (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged)
		return fail(ISING_E_ARG, "%s (%p) is not device memory", what, ptr);
	if (attr.type == hipMemoryTypeDevice && attr.device != device)
		return fail(ISING_E_ARG, "%s (%p) lives on device %d, the slab on device %d", what, ptr, attr.device, device);
	if (have && have < need) return fail(ISING_E_ARG, "%s holds %zu bytes, the slab needs %zu", what, have, need);
	return ISING_OK;
}

int ising_create(const ising_config *cfg, ising_ctx **out) {
	if (!cfg || !out) return fail(ISING_E_ARG, "null argument");
	*out = nullptr;
	// optimized/main.cu:1412-1421
	if (cfg->X <= 0 || (cfg->X % 2048)) return fail(ISING_E_ARG, "Please specify an X dim multiple of 2048 (got %d)", cfg->X);
	if (cfg->Y <= 0 || (cfg->Y % 16)) return fail(ISING_E_ARG, "Please specify a Y dim multiple of 16 (got %d)", cfg->Y);
	if (cfg->nslabs < 1 || cfg->slab < 0 || cfg->slab >= cfg->nslabs) return fail(ISING_E_ARG, "bad slab %d of %d", cfg->slab, cfg->nslabs);
	if ((long long)cfg->Y * cfg->nslabs >= (1LL << 31)) return fail(ISING_E_ARG, "total rows must be < 2^31");
	if ((cfg->XSL != 0) != (cfg->YSL != 0)) return fail(ISING_E_ARG, "XSL and YSL must be given together");
	if (cfg->XSL) { // optimized/main.cu:1440-1453
		if (cfg->XSL < 0 || (cfg->X % cfg->XSL) || (cfg->XSL % 2048))
			return fail(ISING_E_ARG, "Please specify an X sub-lattice dim multiple of 2048 and divisor of %d", cfg->X);
		if (cfg->YSL < 0 || (cfg->Y % cfg->YSL) || (cfg->YSL % 16))
			return fail(ISING_E_ARG, "Please specify a Y sub-lattice dim multiple of 16 divisor of %d", cfg->Y);
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ISING_E_NOGPU, "no HIP device visible");
	if (cfg->device < 0 || cfg->device >= ndev) return fail(ISING_E_ARG, "device %d out of range (%d visible)", cfg->device, ndev);

	ising_ctx *c = new ising_ctx();
	c->cfg = *cfg;
	if (int rc = ising_host::read_policy(&c->pol)) { delete c; return rc; }
	if (hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess || c->cus < 1) { (void)hipGetLastError(); c->cus = 256; }
	if (hipDeviceGetAttribute(&c->xccs, hipDeviceAttributeNumberOfXccs, cfg->device) != hipSuccess || c->xccs < 1) { (void)hipGetLastError(); c->xccs = c->cus >= 200 ? 8 : 1; }
	const ising_policy &pol = c->pol;
	if (cfg->layout != ISING_LAYOUT_AUTO && cfg->layout != ISING_LAYOUT_NIBBLE && cfg->layout != ISING_LAYOUT_DENSE && cfg->layout != ISING_LAYOUT_BALLOT) {
		delete c;
		return fail(ISING_E_ARG, "bad layout %d", cfg->layout);
	}
	if (cfg->kernel != ISING_KERNEL_AUTO && cfg->kernel != ISING_KERNEL_GENERIC && cfg->kernel != ISING_KERNEL_FAST) {
		delete c;
		return fail(ISING_E_ARG, "bad kernel %d", cfg->kernel);
	}
	c->dense = cfg->layout != ISING_LAYOUT_NIBBLE;
	// the ballot layout covers the integer-threshold update without sub-lattices and couplings, 8192-column granularity
	// (sub-lattice widths: 2048, 4096 or a multiple of 8192 columns)
	// X not a multiple of 8192: the last wave column of a row is partly dead (padded rows).  Not with sub-lattices or -J
	// (their period / plane logic counts whole wave columns), and not in a caller-owned buffer (the halo rows the caller
	// sliced out of it would move if the slab ever had to turn dense).
	const bool whole = (cfg->X % 8192) == 0;
	const bool ballot_ok = (whole ? (!cfg->XSL || cfg->XSL <= 4096 || (cfg->XSL % 8192) == 0) : (!cfg->XSL && !cfg->use_J && !cfg->lattice_mem)) &&
	                       (cfg->kernel == ISING_KERNEL_AUTO || cfg->kernel == ISING_KERNEL_FAST);
	if (cfg->layout == ISING_LAYOUT_BALLOT && !ballot_ok) {
		delete c;
		return fail(ISING_E_ARG, "the ballot layout needs the integer-threshold kernel; with sub-lattices, -J or a caller-owned buffer also X %% 8192 == 0 (sub-lattice widths of 2048, 4096 or a multiple of 8192)");
	}
	c->ballot = cfg->layout == ISING_LAYOUT_BALLOT;
	c->lld_packed = cfg->X / 32;
	c->lld = c->dense ? cfg->X / 128 : cfg->X / 32;
	c->lld_dense = cfg->X / 128;
	c->gx = cfg->X / 2048;
	compute_tables(c, cfg->temp);
	c->wrap = cfg->nslabs == 1 && !cfg->ring_halo;
	// How ising_sweep launches on the ballot layout: fused launches (one launch = up to 32 sweeps, in-order tickets,
	// per-strip completion counters; ising_ballot.hip) from 1.5 * 2^24 spins up, on a slab that wraps in place and has no
	// sub-lattices.  A unit's parents are one level = T tickets back, so a level must hold a few times more tickets than
	// workgroups run, or units find their parents unfinished and hold their slots asleep: the strip height H follows
	// from T = wave rows / (H x waves per workgroup), and small lattices run FEWER workgroups than the chip holds.
	// Measured (tools/grid_probe.py, grid_probe2.py), flips/ns fused vs one launch per colour + tail strips (4-wave workgroups):
	//   8192 x 4096 (2^25)   H = 1, 3 per CU (T = 1024), four ticket counters     2260 vs the dense layout's 1827 (2^24: 1369 vs 1415, so dense)
	//   8192^2               H = 2, 3 per CU (T = 1024), two ticket counters      2755 vs 2125 (dense layout 2150)
	//   16384 x 8192         H = 4, 3 per CU (T = 1024)                           3040 vs 2580
	//   16384^2              H = 4, 4 per CU (T = 2048)                           3285 vs 3048
	//   24576^2              H = 8, 4 per CU (T = 2304)                           3386 vs 3182
	//   32768^2              H = 8, 5 per CU (T = 4096)                           3455 vs 3407
	//   131072 x 16384       H = 16, 5 per CU (T = 4096)                          3504 vs 3410
	//   65536^2              H = 16, 6 per CU (T = 8192)                          3533 vs 3461 (five per CU, rounds 1-2: 3512, 2.7 % less HBM traffic)
	//   131072^2             H = 16, 6 per CU (T = 32768)                         3541 vs 3502
	// (choose_fused_strip_rows / fused_wgs_for above; tools/midsize_probe.py re-measures the neighbours of these choices; before the waves' priorities rotated -- ising_ballot.hip -- the same lattices
	// wanted two to four times as many tickets a level and ran 3082 at 16384^2, 3479 at 65536^2.)
	// ISING_FUSED=0/1, cfg.strip_rows and ISING_FUSED_WGS (grid) override.  (8-wave workgroups, an A/B switch of rounds 2-3, lost
	// at every size once the priorities rotated -- 16384^2 3187 vs 3325, 32768^2 3192 vs 3473 -- and are gone.)
	const long long spins = (long long)cfg->X * cfg->Y;
	// Small and narrow lattices (round 5, ising_quad.hip): one launch per pass of T sweeps = the word pass on tiles + halo next to the draws of the pass to come,
	// made ONCE by the rest of the chip -- on a dense slab (the spins live in the dense layout between calls).  A lone slab that wraps in place, integer
	// thresholds, no sub-lattices, no couplings, up to eight blocks of 2048 columns (a tile is as wide as the lattice and keeps two items a wave).
	// By measurement on a whole MI355X (tools/quad_probe.py, profiles/quad_probe_r05.txt; flips/ns quad / the library before): 2048 x 512 739 / 273, 2048^2 1707 / 877,
	// 2048 x 8192 2226 / 1616, 4096 x 1024 1624 / 875, 4096^2 2306 / 1614, 4096 x 16384 2602 / 2194, 6144 x 2048 1996 / 1630, 6144^2 2341 / 2208, 8192 x 1024 1747 / 1219
	// -- and 8192 x 2048 2037 / 2183: the fused launches' from there on.  Five and six blocks up to 1024 rows (three items a wave of sixteen hold tiles of 48 / gx row groups + halo: passes of four to six sweeps;
	// a row of 10240 columns and more is draws enough per launch): 10240 x 512 1582 / 762, 10240 x 1024 1748 / 1011, 12288 x 512 1807 / 1054, 12288 x 1024 1904 / 1293; seven and
	// eight blocks up to 512 rows (14336 x 512 1592 / 1064, 16384 x 512 1747 / 1214; 16384 x 1024: the fused launches' 1728 against 1647).  ISING_QUAD=1 asks for it wherever it applies, 0 never; tests/test_gpu_policy.py holds the rule
	// against the other path.
	const bool quad_can = c->wrap && !cfg->XSL && !cfg->use_J && c->fast_ok && (cfg->Y % 4) == 0 && c->gx <= 8 && pol.quad != 0 &&
	                      (cfg->layout == ISING_LAYOUT_AUTO || cfg->layout == ISING_LAYOUT_DENSE) && cfg->kernel != ISING_KERNEL_GENERIC;
	const bool quad_pick = quad_can && (pol.quad == 1 || (c->cus >= 200 && (c->gx <= 2 ? spins <= (1LL << 26) : (c->gx == 3 ? cfg->Y <= 6144 : (c->gx <= 6 ? cfg->Y <= 1024 : cfg->Y <= 512)))));
	// (sub-lattices: every XSL x YSL block is a periodic system of its own -- nothing crosses slabs, so ring slabs qualify too --;
	// their strips must not straddle a block, and the fused kernels carry no couplings next to sub-lattices)
	const bool fused_can = cfg->XSL ? !cfg->use_J : c->wrap;

	// (rows of a million columns and more -- 128 wave columns -- run 1-2 % faster one launch per colour, whatever their number:
	// 1048576 x 65536 3461 vs 3429 fused, x 524288 3536 vs 3465, 2097152 x 131072 3464 vs 3397; 524288 x 1048576 3484 vs 3513;
	// tools/huge_probe.py.  At 2^37 spins, 8 sweeps: one launch per colour 3506 .. 3528 at every width; fused 3538 up to 32 wave
	// columns, 3512 at 64, 3445 at 128, 3386 at 256, 3186 at 512: a strip's completion counter takes one atomic per wave column
	// and level, all at about the same time, and three polls per unit of the next)
	// (from 768 tickets a level: 8192 x 3072 1834 vs the dense layout's 1658; 8192 x 2048, 512 tickets: 1330 vs 1416)
	// Below 1.5 * 2^24 spins (the end of round 4): a lone slab whose levels of ONE-row units still feed two workgroups per CU (fused_wgs_cap) runs 1720 flips/ns
	// and more in fused launches where the dense layout does 1000 .. 1585 (tile launches) -- 8192 x 2048 2172 vs 1582, 8192 x 2560 2537 vs 1401, 16384 x 1152 2093
	// vs 1217, 32768 x 640 1687 vs 1250; lattices with fewer tickets a level stay dense (8192 x 1024 1065 vs 1179).
	const bool small_fused = c->wrap && !cfg->XSL && spins < 3 * (1LL << 23) && pol.fused_wait_late != 0 && fused_wgs_cap(fused_tickets(c->nwc(), cfg->Y, 1), c->nwc()) >= 2;
	c->fused = pol.fused >= 0 ? pol.fused != 0 : ((spins >= 3 * (1LL << 23) || small_fused) && c->nwc() < 128);
	// AUTO: the ballot kernel's two-phase row pipeline wins from 2^27 spins per slab up -- from 2^25 where fused launches
	// apply (a slab that wraps in place, a ring slab that can keep ghost rows); below, the dense kernel is ahead.  (A partly dead last wave column wastes its
	// dead lanes' draws: rounds 1-3 accepted a tenth of the row.)
	// (end of round 4, measured per fill: a sixth of dead lanes pays at every size -- 20480 x 4096 2495 vs 2075, 20480^2 2815 vs 2687, 28672^2 (an eighth) 3038 vs 2708 --,
	// a quarter up to 2^27 spins -- 12288 x 1536 1898 vs 1245, 12288 x 8192 2436 vs 2242, 12288 x 16384 2500 vs 2592)
	const bool ballot_pays = whole || 5 * c->gx >= 4 * 4 * c->nwc() || (spins <= (1LL << 27) && 4 * c->gx >= 3 * 4 * c->nwc());
	// (a ring slab that can keep ghost rows sweeps in fused launches as well, see below)
	const bool deep_can = !c->wrap && !cfg->XSL && !(cfg->use_J && cfg->coupling_mem) && !cfg->lattice_mem && cfg->Y >= 4 &&
	                      !(pol.ring_ghost >= 0 && pol.ring_ghost < 2);
	const long long ballot_from = (c->fused && fused_can && small_fused) ? 0 : (((c->fused && fused_can) || deep_can) ? 3 * (1LL << 23) : (1LL << 27));
	if (cfg->layout == ISING_LAYOUT_AUTO && ballot_ok && ballot_pays && c->fast_ok && spins >= ballot_from && !pol.no_ballot && !quad_pick)
		c->ballot = true;
	if (c->ballot) c->lld = c->nwc() * 64;
	const bool fused_shape = c->ballot && c->fused && fused_can;
	// A ring slab on the ballot layout keeps G ghost rows on either side (ising_ctx::ghost_rows; ising_ring.cpp: sweep_deep):
	// G rows of both colours travel every G colour half-sweeps, fused launches of G levels run in between.  Not with
	// sub-lattices (nothing crosses slabs) or a caller-owned buffer (fixed shape).
	if (c->ballot && deep_can) {
		// (64 = the levels one fused launch carries.  Ring of one over RCCL, 32 -> 64 rows: 8192^2 2231 -> 2380 flips/ns,
		// 8192 x 16384 2497 -> 2712, 16384^2 2915 -> 2980, 32768^2 3317 -> 3340, 65536^2 3413 -> 3424: the launch boundary
		// and the exchange come half as often, the redundant rows are 128 of Y)
		int G = 64;
		if (pol.ring_ghost >= 0) G = pol.ring_ghost;
		G = std::min(G, cfg->Y / 2) & ~1;
		c->ghost_rows = G >= 2 ? G : 1;
		if (c->ghost_rows > 1 && cfg->use_J) c->ham_ghost = c->ghost_rows + 1; // -J: the ghost rows' couplings are generated in place
	}
	const bool deep_ring = c->ghost_rows > 1;
	// (its fused launches take the shape a single slab of Y + 2 G rows would)
	const int launch_rows = deep_ring ? cfg->Y + 2 * c->ghost_rows : cfg->Y;
	c->H = cfg->strip_rows > 0 ? cfg->strip_rows
	       : (fused_shape && small_fused) ? 1
	       : ((fused_shape || deep_ring) ? choose_fused_strip_rows(c->nwc(), cfg->XSL ? cfg->YSL : cfg->Y, launch_rows)
	                                     : choose_strip_rows(c->gx, cfg->Y, c->dense, c->ballot));
	// (units that draw before they wait: two-row units with 512 tickets a level beat one-row units with 1024 -- 8192 x 4096 2820 vs 2660)
	if (cfg->strip_rows <= 0 && (fused_shape || deep_ring) && c->H == 1 && pol.fused_wait_late != 0 && (cfg->XSL ? cfg->YSL : cfg->Y) % 2 == 0 &&
	    fused_tickets(c->nwc(), launch_rows, 2) >= 512 && (c->nwc() == 1 || fused_wgs_for(fused_tickets(c->nwc(), launch_rows, 2), 2, true, c->nwc()) >= 3)) c->H = 2;
	// (... where a row is one wave column; rows of several want more tickets a level: 32768 x 1024 ran 745 with two-row units, 2446 with one-row units,
	// 16384 x 2048 2199 / 2575, 65536 x 512 371 / 1523 -- profiles/small_fused_probe_r04.txt)
	// Split launches (round 5; ising_ballot.hip: ballot_split_k) carry the sweeps of a slab whose levels have too few tickets for tall strips in the fused form:
	// a lone slab that wraps in place or a ring slab with ghost rows, no sub-lattices, no couplings, a lattice the memory-side cache holds (the masks travel through
	// memory: 4 bits a site next to the lattice's 3).  ISING_SPLIT=1 asks for them wherever they apply, 0 never; otherwise by measurement (tools/ab_probe.py on two
	// boxes, profiles/split_probe_r05e.txt, split_depth_probe_r05.txt; flips/ns split / fused at the fused form's best shape of that box):
	//   8192^2 3256 / 3067 (H = 4)   8192 x 16384 3419 / 3338 (H = 8)   16384^2 3407 / 3392 (the fused form's 3392 needs eight-row strips at six per CU: 835 on the
	//   next box, where the split form does 3391 -- it has no such cliffs)   65536 x 8192 3403 / 3380   24576^2 3412 / 3395   32768^2 and 65536 x 16384: equal   65536^2: 3481 / 3513
	// -- the split form wins while sixteen-row strips make fewer than 2048 tickets a level; its strips are the tallest that still make 512 tickets (8192 x 4096 at
	// H = 4, 256 tickets: 513 flips/ns) AND 512 strips (65536 x 1024 at H = 4, 256 strips of eight wave columns: 544 against the fused form's 2615; 131072 x 2048 at
	// H = 16, 128 strips: 2473 against 3197 -- a strip's counter waits for ALL its wave columns), four rows at least, rows of eight wave columns at most (what was
	// measured), five workgroups per CU (six: +0..1 % on one box, -3..4 % on another).  tests/test_gpu_policy.py holds the choice against its neighbours and the other form.
	{
		const bool nt = pol.fused_nt >= 0 ? pol.fused_nt != 0 : spins > (1LL << 31);
		// (eight ticket classes, one per XCC id: a device or partition that reports another number of dies runs the plain fused form, ISING_SPLIT=1 or not)
		const bool can = ((fused_shape && c->wrap) || deep_ring) && !cfg->XSL && !cfg->use_J && !nt && pol.split != 0 && c->nwc() < 128 && c->xccs == 8;
		const int Yd = cfg->Y; // strips divide the slab's own rows
		if (can && pol.split == 1) {
			c->split = c->split_always = fused_tickets(c->nwc(), launch_rows, c->H) >= 8;
			c->H_split = c->H;
		} else if (can && deep_ring && !small_fused && c->cus >= 200 && cfg->strip_rows <= 0) {
			// Ring slabs (round 6): their launches are tied to the exchanges -- 32 sweeps each, which do not earn back the split form's word-only tail -- UNLESS one
			// persistent launch carries several exchange epochs (the peer transport with a device to itself, ising_ring.cpp: epochs_per_launch): then the rule of
			// the lone slabs applies, and the ring's sweep asks for the form launch by launch (ising_host::update_deep).  Ring of one, 65536 x 8192 (the slab of
			// 65536^2 on eight GPUs): fused 3339, split at sixteen-row strips with epochs 3368 (ghost rows 32 deep: 3385), without epochs 3260; 65536 x 16384
			// (2064 tickets a level at sixteen rows: not "few"): fused 3444, split 3340 (profiles/ring_depth_probe_r06.txt).
			// More ring slabs, same probe (split / fused): 8192^2 3005 / 2980, 16384^2 3317 / 3279, 16384 x 8192 3250 / 3181, 24576^2 3361 / 3381 -- and slabs of 4096 rows,
			// where the ghost rows are 3 % of a launch's: 32768 x 4096 2994 / 3119, 16384 x 4096 2780 / 2779: from 8192 rows on.
			const bool few = fused_tickets(c->nwc(), launch_rows, 16) < 2048 && c->nwc() <= 8 && cfg->Y >= 8192;
			auto feeds = [&](int h) { return launch_rows / h >= 512 && fused_tickets(c->nwc(), launch_rows, h) >= 512; };
			if (few)
				for (int h = 16; h >= 4 && !c->split; h >>= 1)
					if (Yd % h == 0 && h >= c->H && c->ghost_rows % h == 0 && feeds(h)) { c->split = true; c->H_split = h; }
		} else if (can && !small_fused && !deep_ring && c->cus >= 200) { // (a whole MI355X: eight XCDs that each run their share of the grid -- the classes are theirs)
			// (lone slabs only: their launches carry ~50 ms of sweeps.  A launch in the split form ends on word units alone -- the last (lead + 1) x grid of them, five
			// levels deep at 16384^2, memory round trips with an idle vector ALU: ~0.1 ms more per launch than the fused form (rocprofv3, 16 sweeps of 16384^2 per launch:
			// 1450 against 1346 us, profiles/rocprof_r05_config2_split.txt / _fused.txt) --, which a ring slab's launches of 32 sweeps do not earn back; ISING_SPLIT=1 still
			// puts ring slabs on it)
			const bool few = fused_tickets(c->nwc(), launch_rows, 16) < 2048 && c->nwc() <= 8;
			auto feeds = [&](int h) { return launch_rows / h >= 512 && fused_tickets(c->nwc(), launch_rows, h) >= 512; };
			if (cfg->strip_rows > 0) {
				c->split = few && c->H >= 4 && c->H <= 16 && feeds(c->H);
				c->H_split = c->H;
			} else if (few) {
				// (its strips are taller than the fused form's, which keeps its own shape: a call of few sweeps runs that -- ising_update.cpp: split_pays)
				for (int h = 16; h >= 4 && !c->split; h >>= 1)
					if (Yd % h == 0 && h >= c->H && feeds(h)) { c->split = true; c->H_split = h; }
			}
		}
	}
	if (cfg->Y % c->H) { const int h = c->H; delete c; return fail(ISING_E_ARG, "strip_rows %d does not divide Y %d", h, cfg->Y); }
	c->nstrips = cfg->Y / c->H;
	c->color_words = (size_t)cfg->Y * c->lld;
	// The quad path's shape (quad_pick above).  Tiles of C row groups (4 rows each) x the whole width + HG halo row groups per side, T sweeps a pass, waves per
	// workgroup (tiles and drawing workgroups alike: one launch) -- by measurement (profiles/quad_probe_r05.txt).
	if (quad_pick && c->dense && !c->ballot) {
		const int NRG = cfg->Y / 4;
		// Shape by measurement (tools/quad_probe.py, profiles/quad_shapes_probe_r05.txt; row groups per tile, sweeps per pass, waves).  A launch has a fixed part -- ramp,
		// the tiles' first loads, the tail -- that long passes spread over more sweeps while a pass's halo (two rows per sweep and side) makes the word phase
		// dearer: lattices of few tiles want long passes (2048^2: (4, 8, 8) 1799, (4, 12, 8) 2119, (8, 16, 12) 2210 flips/ns), a thousand tiles and more short
		// ones on tall tiles (throughput: less of the word phase is halo -- 2048 x 32768 (8, 4, 8) 2481 against 2374, 4096 x 16384 (8, 4, 12) 2431 against 2342).
		// A wave keeps at most two items: 80 registers, six waves per SIMD.
		int C0 = 4, T0 = 8, W0 = 12;
		ising_host::quad_shape(c->gx, cfg->Y, &C0, &T0, &W0);
		int T = pol.quad_T ? pol.quad_T : T0;
		T = std::max(1, std::min(T, 32));
		const int HG = (2 * T - 1 + 3) / 4;
		int C = pol.quad_C ? pol.quad_C : C0;
		C = std::max(1, std::min(C, NRG));
		int waves = pol.quad_waves ? pol.quad_waves : W0;
		waves = std::max(1, std::min(waves, 16));
		ising::QuadWordParams qp{};
		qp.gx = c->gx; qp.NRG = NRG; qp.C = C; qp.HG = HG;
		const int mi = ising::quad_word_maxi(qp, waves);
		int lds_max = 64 * 1024;
		if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeSharedMemPerBlockOptin, cfg->device) != hipSuccess || lds_max <= 0) { (void)hipGetLastError(); lds_max = 64 * 1024; }
		if (mi > 0 && ising::quad_pass_lds_bytes(qp, waves) + 64 <= (size_t)lds_max) { // (+ the kernel's static words: a print point's two sums)
			c->quad_C = C; c->quad_T = T; c->quad_HG = HG; c->quad_waves = waves;
		} else if (pol.quad == 1 && (pol.quad_C || pol.quad_T || pol.quad_waves)) {
			delete c;
			return fail(ISING_E_ARG, "ISING_QUAD_*: tiles of %d row groups + 2 x %d at %d waves: too many items a wave, or more LDS than a workgroup's (%d bytes) at X = %d", C, HG, waves, lds_max, cfg->X);
		}
	}
	// Lattices that stay on the dense layout (a lone slab, no couplings, no sub-lattices), up to 2^24 spins: tile launches of several sweeps
	// (ising_dense.hip: dense_tile_k) instead of one launch per colour, which is bound by the launches themselves below ~2^24 spins (9.7 us
	// per sweep whatever the lattice).  tools/tile_probe.py, flips/ns tiles / per colour (profiles/tile_probe_r04.txt): 2048 x 1024 479 / 217,
	// 2048^2 828 / 429, 4096 x 2048 1207 / 867, 4096^2 1595 / 1413.  Shape by measurement: one tile per CU (tile rows 8 .. 64 by 16 words
	// + halo at 2^21 .. 2^24 spins), 3 / 4 / 6 sweeps a launch for 8 / 16-32 / 48+ rows, 512 threads while the first half-sweep's words fit two rounds of them.
	if (c->dense && !c->ballot && c->wrap && !cfg->use_J && !cfg->XSL && pol.tiles != 0 && (pol.tiles == 1 || spins <= (1LL << 24)) && c->quad_C == 0) {
		const int wpr = c->gx * 32;
		auto sweeps_for = [&](int tr) { const int s0 = pol.tile_sweeps ? pol.tile_sweeps : (tr <= 8 ? 3 : (tr >= 48 ? 6 : 4)); return std::max(1, std::min({s0, 16, cfg->Y / 2})); };
		// the shape whose busiest CU has the fewest words to update per sweep: rounds of tiles x (rows + the halo rows' average) x (words + 2)
		int TR = pol.tile_rows, TWI = pol.tile_words;
		long long best = -1;
		for (int tr : {8, 16, 24, 32, 48, 64, 96, 128}) {
			if (pol.tile_rows ? tr != 8 : (cfg->Y % tr) != 0) continue;
			for (int tw : {8, 12, 16, 24, 32, 48, 64}) {
				if (pol.tile_words ? tw != 8 : (wpr % tw) != 0) continue;
				const int r = pol.tile_rows ? pol.tile_rows : tr, w = pol.tile_words ? pol.tile_words : tw;
				if (r > cfg->Y || (cfg->Y % r) != 0 || (wpr % w) != 0) continue;
				const int S = sweeps_for(r);
				const long long tiles = (long long)(wpr / w) * (cfg->Y / r);
				const long long rounds = (tiles + c->cus - 1) / c->cus;
				// (+ a launch's fixed cost, ~7 us, in words' time; + 15 % where a CU takes several tiles in turn: 10240 x 1024 849 vs 1004 flips/ns)
				const long long cost = rounds * (r + 2 * S - 1) * (w + 2) * (rounds > 1 ? 115 : 100) / 100 + 900 / S;
				if (!pol.tile_rows && !pol.tile_words && (long long)(r + 4 * S - 2) * (w + 2) > 4096) continue; // (the rule: four rounds of 1024 threads at most)
				if (best < 0 || cost < best) { best = cost; TR = r; TWI = w; }
			}
		}
		const int S = sweeps_for(TR > 0 ? TR : 8);
		int NT = pol.tile_threads;
		if (!NT) NT = (TR + 4 * S - 2) * (TWI + 2) <= 600 ? 512 : 1024;
		const bool ok = best >= 0 && (NT == 256 || NT == 512 || NT == 1024);
		ising::TileParams tp{};
		tp.TR = TR; tp.TWI = TWI; tp.ns = S;
		// (dense_tile_k's dynamic segment + its static words -- the workgroup's up-spin sum -- against what a workgroup of this device may take)
		int lds_max = 64 * 1024;
		if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device) != hipSuccess || lds_max <= 0) { (void)hipGetLastError(); lds_max = 64 * 1024; }
		if (ok && ising::dense_tiles_lds_bytes(tp) + 64 <= (size_t)std::min(lds_max, 64 * 1024)) {
			c->tile_rows = TR; c->tile_words = TWI; c->tile_sweeps = S; c->tile_threads = NT;
			c->tile_xcd = pol.tile_xcd != 0 && (((long long)(wpr / TWI) * (cfg->Y / TR)) % 8) == 0; // (tiles in bands per XCD)
		} else if (pol.tile_rows || pol.tile_words || pol.tile_threads) {
			delete c;
			return fail(ISING_E_ARG, "ISING_TILE_*: tiles of %d rows x %d words, %d sweeps, %d threads do not fit a %d x %d lattice", TR, TWI, S, NT, cfg->Y, cfg->X);
		}
	}
	// lattices larger than the memory-side cache (256 MB = 2^31 spins at 1 bit per spin) stream through it: their words
	// carry the non-temporal hint, which keeps the accept-mask slots in the L2s (ISING_FUSED_NT=0/1 overrides)
	c->fused_nt = pol.fused_nt >= 0 ? pol.fused_nt : (spins > (1LL << 31));
	if (fused_shape || deep_ring) { // workgroups per CU of a fused launch (the chip holds 6 of 4 waves)
		const long long T = fused_tickets(c->nwc(), launch_rows, c->H);
		c->fused_wait_late = pol.fused_wait_late != 0; // (default on: free where parents are never late, 65536^2 3528.6 vs 3527.4)
		c->fused_wg_per_cu = fused_wgs_for(T, c->H, c->fused_wait_late, c->nwc());
		if (deep_ring && T < 16384) c->fused_wg_per_cu = std::min(c->fused_wg_per_cu, 5);
		// Several ticket counters where 4-wave workgroups draw one- or two-row units (2^26 spins): one counter hands out
		// ~80 tickets per us; 8192^2 with one-row units at 2600 flips/ns needs 159 (four counters), with two-row units
		// 79 (two).  ISING_FUSED_TICKETS2=0/2/4 overrides.
		const long long T0 = fused_tickets(c->nwc(), cfg->Y, c->H); // (without a ring slab's ghost rows)
		// (end of round 4: whatever T0 -- one counter caps one-row units at 1350 flips/ns and two-row units at 2690, the rate at which it hands tickets out:
		// 24576 x 4096, H = 2, T0 = 1536: 2697 with one counter, 2999 with two)
		(void)T0;
		c->fused_tickets2 = c->H == 1 ? 4 : (c->H == 2 ? 2 : 0);
		if (pol.fused_tickets2 >= 0) c->fused_tickets2 = pol.fused_tickets2;
	}

	if (c->split) { // (decided above, with the strip height)
		int wgs = pol.fused_wgs > 0 ? std::max(8, pol.fused_wgs) : 5 * c->cus;
		c->split_lead = pol.split_lead >= 0 ? pol.split_lead : 1;
		c->split_cap = (wgs / 8) * 5 / 4 + 1; // an eighth of the grid and a margin: what a class serves (the rest of an uneven placement leaves)
		int sh = 3;
		while ((1 << sh) < (c->split_lead + 1) * c->split_cap) sh++;
		c->split_ring_sh = sh;
		c->split_wg_per_cu = std::max(1, wgs / c->cus);
	}

	hipError_t e = hipSetDevice(cfg->device);
	if (e == hipSuccess && cfg->lattice_mem) {
		if (int rc = check_caller_buffer(cfg->lattice_mem, cfg->lattice_mem_bytes, c->alloc_words() * sizeof(uint64_t), cfg->device, "lattice_mem")) { delete c; return rc; }
	}
	if (e == hipSuccess && cfg->use_J && cfg->coupling_mem) {
		if (int rc = check_caller_buffer(cfg->coupling_mem, cfg->coupling_mem_bytes, c->ham_alloc_words() * sizeof(uint64_t), cfg->device, "coupling_mem")) { delete c; return rc; }
	}
	if (e == hipSuccess) {
		if (cfg->lattice_mem) c->d_lat = static_cast<uint64_t *>(cfg->lattice_mem);
		else e = hipMalloc((void **)&c->d_lat, c->alloc_words() * sizeof(uint64_t));
	}
	if (e == hipSuccess) e = hipMemset(c->d_lat, 0, c->alloc_words() * sizeof(uint64_t)); // optimized/main.cu:1603
	// tile launches read one buffer and write the other: the second one is the library's own (also next to a caller's lattice_mem), allocated
	// here so that a sweep never allocates and an out-of-memory condition is ising_create's to report
	if (e == hipSuccess && c->tile_rows > 0) e = hipMalloc((void **)&c->d_lat2, c->alloc_words() * sizeof(uint64_t));
	if (e == hipSuccess && c->quad_C > 0) {
		e = hipMalloc((void **)&c->d_quad, 4 * c->quad_words() * sizeof(uint64_t));
		if (e == hipSuccess) e = hipMalloc((void **)&c->d_qmasks, 2 * (size_t)(2 * c->quad_T) * (c->quad_words() / 64) * 1024);
	}
	if (e == hipSuccess) e = hipMalloc((void **)&c->d_acc, 4 * sizeof(unsigned long long));
	if (e == hipSuccess && c->ballot) {
		// accept-mask slots, 2 KiB per wave: of every wave of the largest plain launch (one workgroup per unit), and of
		// every workgroup slot of a fused launch (4 waves each)
		// Tail strips of plain launches (launch_ranges): the last rows of a launch go in strips of one row, so that the
		// launch ends on many short units that fill the gaps the last round of H-row units leaves -- about two thirds of
		// one H-row unit per workgroup slot (6 workgroups per CU), measured: +6 % at 16384^2, +3.5 % at 32768^2, +1 % at
		// 65536^2, nothing at 8192^2 (tools/tail_probe.py).  ISING_TAIL=<rows>[,<strip height>] overrides, 0 disables.
		if (c->H > 1) {
			int cus = 256;
			(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device);
			long long rows = 2LL * (6LL * cus) * c->H / (3LL * c->nwc());
			rows = rows / c->H * c->H;
			c->tail_rows = (int)std::min<long long>(rows, cfg->Y / 4 / c->H * c->H);
			c->tail_h = 1;
		}
		if (pol.tail_rows == 0) c->tail_rows = 0;
		else if (pol.tail_rows > 0 && pol.tail_rows % c->H == 0 && pol.tail_rows % pol.tail_h == 0 && pol.tail_h < c->H && 2 * pol.tail_rows < cfg->Y) { c->tail_rows = pol.tail_rows; c->tail_h = pol.tail_h; }
		const size_t strips = c->tail_rows ? (size_t)(cfg->Y - c->tail_rows) / c->H + (size_t)c->tail_rows / c->tail_h : (size_t)c->nstrips;
		// (+ H: the flag-synchronised ring schedule may turn one more H-row strip into one-row strips, launch_ranges)
		const bool guard_can = c->fused && fused_shape && c->wrap && !cfg->XSL && !cfg->use_J && (pol.guard >= 0 ? pol.guard != 0 : c->cus >= 200);
		if (guard_can) {
			c->guard.state = 1;
			if (hipEventCreate(&c->guard.e0) != hipSuccess || hipEventCreate(&c->guard.e1) != hipSuccess) { (void)hipGetLastError(); c->guard.state = 0; }
		}
		const size_t plain = (size_t)c->nwc() * (strips * (guard_can ? 2 : 1) + 2 + (size_t)c->H) * 2048 + 16 * 2048, fused = (size_t)ising::ballot_max_wgs(c->cus) * 4 * 2048;
		e = hipMalloc((void **)&c->d_scratch, std::max(plain, fused));
		// The ring's edge-row launches (two rows, comm stream) run next to the interior launch of the same colour (compute
		// stream): slots of their own, or the two launches overwrite each other's accept masks -- which they did: the
		// two-stream schedule gave wrong spins at 65536^2 (tools/ring_parity_probe.py), unnoticed at test sizes.
		if (e == hipSuccess) e = hipMalloc((void **)&c->d_scratch_edge, ((size_t)2 * c->nwc() + 8) * 2048);
		// ticket words (chunk counter + 8 queue words, 64 bytes apart) + one completion counter per strip (fused launches)
		// (the split form keeps completion counters of its own behind those: its strips are other strips, and both sets are counts that start a launch from a common base)
		// (the guard may halve a lone slab's strip height once: room for twice the strips, here and in the plain launches' accept-mask slots above)
		c->ctl_strips = (size_t)c->nstrips * (guard_can ? 2 : 1);
		const size_t ctl_bytes = SLOTCTL_TICKET_BYTES + 2 * (c->ctl_strips + 2 * (size_t)c->ghost_rows + 2) * sizeof(uint32_t); // (+ strips of the ghost rows)
		if (e == hipSuccess) e = hipMalloc((void **)&c->d_slotctl, ctl_bytes);
		if (e == hipSuccess) e = hipMemset(c->d_slotctl, 0, ctl_bytes);
		c->slotctl_bytes = ctl_bytes;
		if (e == hipSuccess && c->split) {
			const size_t mask_bytes = ((size_t)8 << c->split_ring_sh) * 4 * (size_t)c->H_split * 1024;
			c->split_ctl_bytes = 8 * 16 * sizeof(unsigned long long) + ((size_t)8 << c->split_ring_sh) * 2 * sizeof(uint32_t);
			e = hipMalloc((void **)&c->d_split_masks, mask_bytes);
			if (e == hipSuccess) e = hipMalloc((void **)&c->d_split_ctl, c->split_ctl_bytes);
		}
		// the word a fused launch that gives up raises (pinned: the host reads it without a copy, the kernel only when it waits)
		if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_abort, 64, hipHostMallocMapped);
		if (e == hipSuccess) memset(c->h_abort, 0, 64);

	}
	if (e == hipSuccess && cfg->use_J) {
		if (cfg->coupling_mem) c->d_ham = static_cast<uint64_t *>(cfg->coupling_mem);
		else e = hipMalloc((void **)&c->d_ham, c->ham_alloc_words() * sizeof(uint64_t));
		if (e == hipSuccess) e = hipMemset(c->d_ham, 0, c->ham_alloc_words() * sizeof(uint64_t)); // optimized/main.cu:1609
	}
	if (e != hipSuccess) {
		const int rc = fail(ISING_E_HIP, "device allocation failed: %s", hipGetErrorString(e));
		ising_destroy(c);
		return rc;
	}
	*out = c;
	return ISING_OK;
}

int ising_destroy(ising_ctx *c) {
#if defined(ISING_FUSED_TRACE)
	if (c && c->ballot && c->fused) { (void)hipDeviceSynchronize(); ising::ballot_trace_dump(); }
#endif
	if (!c) return ISING_OK;
	(void)hipSetDevice(c->cfg.device);
#if defined(ISING_QUAD_TRACE)
	if (c->quad_C > 0) { (void)hipDeviceSynchronize(); ising::quad_trace_dump(); }
#endif
	if (c->own_stream) { (void)hipStreamSynchronize(c->own_stream); (void)hipStreamDestroy(c->own_stream); }
	if (c->h_meas) (void)hipHostFree(c->h_meas);
	if (c->h_abort) (void)hipHostFree(c->h_abort);
	ising_host::ring_release(c);
	if (c->d_lat && !c->cfg.lattice_mem) (void)hipFree(c->d_lat);
	if (c->d_acc) (void)hipFree(c->d_acc);
	if (c->d_tmp) (void)hipFree(c->d_tmp);
	if (c->d_lat2) (void)hipFree(c->d_lat2);
	if (c->d_quad) (void)hipFree(c->d_quad);
	if (c->d_qmasks) (void)hipFree(c->d_qmasks);
	if (c->d_tile_cnt) (void)hipFree(c->d_tile_cnt);
	if (c->d_scratch) (void)hipFree(c->d_scratch);
	if (c->d_ham && !c->cfg.coupling_mem) (void)hipFree(c->d_ham);
	if (c->d_bits) (void)hipFree(c->d_bits);
	if (c->d_corr) (void)hipFree(c->d_corr);
	if (c->guard.e0) (void)hipEventDestroy(c->guard.e0);
	if (c->guard.e1) (void)hipEventDestroy(c->guard.e1);
	if (c->d_slotctl) (void)hipFree(c->d_slotctl);
	if (c->d_edge) (void)hipFree(c->d_edge);
	if (c->d_scratch_edge) (void)hipFree(c->d_scratch_edge);
	if (c->d_pack) (void)hipFree(c->d_pack);
	if (c->d_conv) (void)hipFree(c->d_conv);
	if (c->d_self) (void)hipFree(c->d_self);
	if (c->d_mslots) (void)hipFree(c->d_mslots);
	if (c->d_cnt) (void)hipFree(c->d_cnt);
	if (c->d_clk) (void)hipFree(c->d_clk);
	if (c->d_split_masks) (void)hipFree(c->d_split_masks);
	if (c->d_split_ctl) (void)hipFree(c->d_split_ctl);
	delete c;
	return ISING_OK;
}

int ising_set_stream(ising_ctx *c, void *hip_stream) {
	if (!c) return fail(ISING_E_ARG, "null context");
	c->stream = static_cast<hipStream_t>(hip_stream);
	return ISING_OK;
}

int ising_use_private_stream(ising_ctx *c) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (c->own_stream) { c->stream = c->own_stream; return ISING_OK; }
	if (int rc = bind(c)) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream)); // nothing of this context may still run on the stream it leaves
	HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
	c->stream = c->own_stream;
	return ISING_OK;
}

int ising_synchronize(ising_ctx *c) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (int rc = bind(c)) return rc;
	return ising_host::sync_checked(c);
}

int ising_init_lattice(ising_ctx *c) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (int rc = bind(c)) return rc;
	c->ghost_depth[0] = c->ghost_depth[1] = 0;
	const uint64_t half = ising_host::draw_prefix(0.5f, false); // curand_uniform(x) < 0.5f, optimized/main.cu:133
	for (int color = 0; color < 2; color++) {
		ising::InitParams p{};
		p.dst = c->lat(color);
		p.seed_lo = (uint32_t)c->cfg.seed;
		p.seed_hi = (uint32_t)(c->cfg.seed >> 32);
		p.color = (uint32_t)color;
		p.gx = c->gx;
		p.Y = c->cfg.Y;
		p.row_base = (uint32_t)c->cfg.slab * (uint32_t)c->cfg.Y;
		p.wrap = c->wrap;
		p.thr_half = (uint32_t)half;
		HIP_TRY(c->ballot ? ising::launch_ballot_init(p, c->stream) : (c->dense ? ising::launch_dense_init(p, c->stream) : ising::launch_init(p, c->stream)));
	}
	return ISING_OK;
}

int ising_set_temperature(ising_ctx *c, float temp) {
	if (!c) return fail(ISING_E_ARG, "null context");
	compute_tables(c, temp);
	return ISING_OK;
}

int ising_get_tables(ising_ctx *c, float exp_table[10], uint64_t thr[5]) {
	if (!c) return fail(ISING_E_ARG, "null context");
	if (exp_table) memcpy(exp_table, c->tab, sizeof(c->tab));
	if (thr) memcpy(thr, c->thr, sizeof(c->thr));
	return ISING_OK;
}

int ising_strip_info(ising_ctx *c, int *strip_rows, int *nstrips) {
	if (!c) return fail(ISING_E_ARG, "null context");
	// (a slab whose long calls run split launches: their strips)
	if (strip_rows) *strip_rows = c->split ? c->H_split : c->H;
	if (nstrips) *nstrips = c->split ? c->cfg.Y / c->H_split : c->nstrips;
	return ISING_OK;
}

} // extern "C"
