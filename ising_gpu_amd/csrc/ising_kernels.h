// ising_kernels.h -- launch wrappers of the gfx950 kernels (internal to libising_hip.so).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace ising {

// Philox4x32-10 constants (Salmon et al. SC'11 / Random123; cuRAND's curandStatePhilox4_32_10_t uses the same).
constexpr uint32_t PHILOX_M0 = 0xD2511F53u;
constexpr uint32_t PHILOX_M1 = 0xCD9E8D57u;
constexpr uint32_t PHILOX_W0 = 0x9E3779B9u;
constexpr uint32_t PHILOX_W1 = 0xBB67AE85u;

struct UpdateParams {
	uint64_t *dst;            // colour being updated: pointer to row 0 of [-1..Y][lld] words (rows -1, Y = halo rows)
	const uint64_t *src;      // opposite colour, same shape; its rows -1 and Y must be current
	int32_t wrap;             // copy the updated rows 0 / Y-1 into the halo rows that mirror them:
	long long mir0_bytes;     //   row 0's copy goes mir0_bytes behind row 0 (single slab: its own row Y; ring on one device: the
	long long mirL_bytes;     //   previous slab's row Y), row Y-1's copy mirL_bytes behind row Y-1 (own row -1 / next slab's row -1)
	uint32_t seed_lo, seed_hi;
	uint32_t it;              // reference's 1-based iteration index (0 for init)
	uint32_t color;           // 0 black, 1 white
	int32_t gx;               // X/2048: 32-vector column groups per row (= reference gridDim.x)
	int32_t Y;                // rows in this slab
	uint32_t row_base;        // global row of slab row 0 (slab*Y)
	int32_t slV;              // periodic extent in X in 128-bit vectors (= gx*32 without sub-lattices)
	int32_t slY;              // periodic extent in Y in rows; 0 = no sub-lattices (rows -1 / Y are the halo rows)
	int32_t H;                // rows each lane marches (a "strip"; the last strip of a range may be shorter)
	int32_t row_lo[2], row_hi[2]; // up to two row ranges per launch (the two edge rows of a slab go in one launch)
	int32_t H2;               // ballot layout: rows per strip in range 1 when it differs from H (tail strips); 0 = H
	int32_t nreal0;           // ballot layout: units of range 0 that exist (nunits0 may be padded to whole workgroups)
	int32_t nunits0;          // units (column group x strip) of range 0
	int32_t nunits;           // units of both ranges
	uint32_t n3, n4;          // integer accept thresholds for 3 / 4 aligned neighbours (fast kernel)
	float tab[10];            // exp table exp_h[2][5] (generic kernel)
	const uint64_t *jdst;     // coupling words read for the rows being updated (NULL without -J); same shape as dst
	uint64_t *scratch;        // ballot layout: 2 KiB of accept-mask slots per wave of the GRID (ballot_max_wgs() x 4 waves)
	// ballot layout, persistent launches
	unsigned long long *ticket; // fused: ticket words (chunk counter + 8 queue words, 64 bytes apart), zero when the launch starts
	int32_t nwg;              // workgroup units per level (set by the launcher)
	int32_t wg_per_cu;        // fused: cap of the persistent grid, workgroups per CU (0: what the chip holds; host side only)
	int32_t nt_stream;        // fused: lattice words with the non-temporal hint (lattices larger than the memory-side cache; host side only)
	unsigned long long ticket_base2[4]; // fused: value of the ticket counter(s) when this launch starts (counter k: 64 k bytes on)
	int32_t tickets2;         // fused: 0 / 1 = one counter; 2 or 4: that many (unit numbers by class = blockIdx mod 2 or 4)
	int32_t nlevels;          // colour half-sweeps in this launch.  > 1 = fused: level L updates colour (color + L) & 1 at
	                          // iteration it + (color + L) / 2 over rows [row_lo[0], row_hi[0]) = the whole slab (wrap)
	uint64_t *lat[2];         // fused: row-0 pointers of both colours
	const uint64_t *jham[2];  // fused, -J: coupling words read when colour c is updated
	uint32_t *done;           // fused: completed wave columns per strip (monotone); reads `done_base` when the launch starts
	uint32_t done_base;
	int32_t total_rows;       // rows of the whole lattice when this launch covers ghost rows (their global row wraps around the ring); 0: off
	// Ring slab with ghost rows, fused launch (ising_ring.cpp: sweep_deep).
	// trapezoid: level L of a launch of nlevels only needs the rows within nlevels - 1 - L of the slab's own [0, Y) -- what lies
	// further out could not reach a row of the slab any more -- so strips wholly outside that range do nothing at that level.
	int32_t trapezoid;
	// Exchange overlapped with the launches: units that touch a row outside [edge_lo, edge_hi) -- rows the exchange reads (the
	// slab's first / last G rows) or writes (the ghost rows) -- wait at level 0 until *edge_go has reached edge_go_need (the comm
	// stream sets it when this slab's rows have been sent and the neighbours' have arrived), and at the LAST level each adds 1
	// per wave column to *edge_done once its rows are out: the comm stream waits for that count and starts the next exchange
	// while the launch is still working on the slab's interior.  (edge_go NULL: off)
	const uint32_t *edge_go;
	uint32_t edge_go_need;
	uint32_t *edge_done;
	int32_t edge_lo, edge_hi;
	// Several exchange EPOCHS in one launch (round 6; ballot_update_k, fused form): with epoch_sh = s > 0 the launch's levels fall into epochs of 2^s
	// levels (= the ghost rows' depth) -- the trapezoid starts over with every epoch, edge units wait at each epoch's first level until *edge_go has reached
	// edge_go_need + epoch, and add to *edge_done at each epoch's last level (and at the launch's last): the comm stream runs one exchange per epoch next
	// to ONE persistent launch, which neither drains nor restarts in between.  0: the whole launch is one epoch.
	int32_t epoch_sh;
	// Batched fused launch (ising_batch_*): `nrep` independent lattices of one shape share the launch's tickets -- a level has
	// nrep x nwg_rep of them, workgroup unit u of a level belongs to lattice u / nwg_rep -- so that small lattices (8192^2: a
	// level of 128 .. 1024 tickets) fill the chip with tall strips.  Per lattice: a 32-byte record in device memory (both
	// colours' row-0 pointers, its accept thresholds, its seed) read with one scalar load per unit; completion counters
	// done + r * done_stride; lattice words, mirror offsets, iteration and everything else as in the single launch.
	// fused launches: a unit that has polled its parents' counters abort_polls times sets *abort_flag (pinned host memory) and
	// stops waiting, as does every unit that finds it set (looked at every 64th poll): a launch whose counters can never
	// arrive -- bases out of step after a faulted launch -- runs to its end instead of to the watchdog (NULL: polls for ever)
	uint32_t *abort_flag;
	uint32_t abort_polls;
	int32_t cus;              // compute units of the device: workgroup b of a persistent grid is in dispatch round b / cus (host: 0 = ask)
	int32_t grid_cap;         // fused: explicit size of the persistent grid (tests, A/B; 0: by wg_per_cu; host side only)
	int32_t wait_late;        // fused: units draw their first row (2: their first two rows) before they wait for their parents
	// In-launch counts (fused launches of a lone lattice, or of a ring slab over its own rows, that start with the black colour): the up spins are counted
	// after the launch's sweeps cnt_first, cnt_first + cnt_every, ... (cnt_every = 0: none; cnt_magic = ceil(2^32 / cnt_every), unused for 1); the
	// measurements of a launch are numbered cnt_slot0, cnt_slot0 + 1, ... in sweep order; the wave that works on unit wave w of a level leaves its sum in
	// cnt_acc[(2 m + colour) * (4 nwg) + w] (zero before the launch: waves without rows store nothing); launch_count_fold adds a measurement's slots up.
	// NULL: off.  (Rounds 4's 64-bit mask of measured sweeps capped a counted launch at 64 sweeps.)
	uint32_t *cnt_acc;
	int32_t cnt_first, cnt_every;
	uint32_t cnt_magic;
	int32_t cnt_slot0;
	// ... with the energy: cnt_bonds != 0 = a measurement has a third plane of slots, cnt_acc[(3 m + 2) * (4 nwg) + w] (and its up-spin planes are
	// (3 m + colour)): the white level's waves leave the number of (white site of their rows, black neighbour) pairs with equal spins there --
	// their sum over the lattice is ising_bond_equal's A (every bond has one white end)
	int32_t cnt_bonds;
	// Measurement aid (ising_kernel_clock): 8 x {cycle counter, 100 MHz counter} x {start, end} left by the launch's first eight workgroups; NULL: off
	unsigned long long *clk_out;
	// Split launch (ising_ballot.hip: ballot_split_k): draw units and word units with tickets of their own, eight classes (one per XCD).
	unsigned long long *sp_ctr; // per class 16 words: [0] draw tickets, [4] workgroups registered, [8] word tickets (zero when the launch starts)
	uint32_t *sp_flags;         // per class and ring slot {waves that have drawn, waves that have used} (counts, zero when the launch starts)
	uint64_t *sp_masks;         // per class 2^sp_ring_sh slots of 4 waves x H rows x 1 KiB of accept masks
	int32_t sp_ring_sh, sp_lead, sp_cap; // ring slots per class (log2); draw units a workgroup does before its first word unit; workgroups a class serves
	const struct ReplicaParams *rep;
	int32_t nrep, nwg_rep;
	uint32_t rep_magic;       // ceil(2^32 / nwg_rep)
	int32_t done_stride;
};

struct ReplicaParams { // 32 bytes, 32-byte aligned: one s_load_dwordx8
	uint64_t *lat[2];
	uint32_t n3, n4;
	uint32_t seed_lo, seed_hi;
};

// mode: 0 = integer thresholds via v_cmpx, 1 = generic FP32-table kernel
hipError_t launch_update(const UpdateParams &p, int mode, hipStream_t stream);

struct InitParams {
	uint64_t *dst;
	uint32_t seed_lo, seed_hi;
	uint32_t color;
	int32_t gx, Y;
	uint32_t row_base;
	int32_t wrap;
	uint32_t thr_half; // number of draws x with curand_uniform(x) < 0.5f
};
hipError_t launch_init(const InitParams &p, hipStream_t stream);

// measurement aid: `blocks` x 256 lanes x nrows x 16 Philox blocks (= 64 sites each); out holds blocks * 256 words
// `clk` (optional): the first eight blocks leave {cycle counter, 100 MHz counter} at their start and end there (4 words each)
hipError_t launch_philox_ceiling(uint32_t *out, int blocks, int nrows, hipStream_t stream, unsigned long long *clk = nullptr);

// up-spin count of `nwords` packed words, accumulated into *acc (one 64-bit atomic per block)
hipError_t launch_popcount(const uint64_t *v, size_t nwords, unsigned long long *acc, hipStream_t stream);

struct BondParams {
	const uint64_t *black, *white; // row-0 pointers; white's rows -1 and Y (halo rows) must be current
	int32_t gx, Y;
	uint32_t row_base;
	int32_t slV, slY; // as in UpdateParams
	unsigned long long *acc;
};
hipError_t launch_bond_equal(const BondParams &p, hipStream_t stream);

// -J couplings.  hamiltInitB_k: random bits for the black coupling array (4 per site), generator seed, offset 0.
struct HamInitParams {
	uint64_t *hamB;        // row 0 of [-1..Y][lld]
	uint32_t seed_lo, seed_hi;
	int32_t gx, Y;
	uint32_t row_base;
	int32_t wrap;
	uint32_t thr;          // number of draws x with curand_uniform(x) < prob
	uint32_t total_rows;   // rows of the whole lattice when the range runs around the ring (ghost rows of a ring slab), else 0
};
hipError_t launch_ham_init_black(const HamInitParams &p, hipStream_t stream);
// hamiltInitW_k as a gather: every white coupling word is assembled from the black words at the other ends of its bonds
struct HamWhiteParams {
	const uint64_t *hamB;  // rows -1 and Y must be current
	uint64_t *hamW;
	int32_t lld, Y;
	uint32_t row_base;
	int32_t slW, slY;      // periodic extents: words per row (= lld without sub-lattices), rows (0 = use halo rows)
	int32_t wrap;
};
hipError_t launch_ham_init_white(const HamWhiteParams &p, hipStream_t stream);

// dense layout (1 bit per spin, ising_dense.hip): same parameter blocks, rows of gx*32 32-bit words
hipError_t launch_dense_update(const UpdateParams &p, int mode, hipStream_t stream); // mode as launch_update

// Small lattices on the dense layout: `ns` whole sweeps in ONE launch without any exchange between workgroups (ising_dense.hip:
// dense_tile_k).  A workgroup keeps a tile of both colours in LDS together with 2 ns halo rows above and below and one halo word
// (32 sites) left and right, updates it 2 ns times over a region that shrinks by one row per colour half-sweep, and stores its own
// TR x TWI words to the OTHER buffer (the neighbours read their halos from the source buffer while it does).
struct TileParams {
	const uint32_t *src[2]; // row 0 of the black / white array the launch reads
	uint32_t *dst[2];       // ... and of the one it writes (row -1 and row Y of each are the mirror rows of a lone slab)
	uint32_t seed_lo, seed_hi, it; // `it`: iteration of the launch's first sweep
	uint32_t n3, n4;
	int ns;                 // sweeps in this launch (black first)
	int gx, Y;              // 2048-column blocks per row; rows
	int TR, TWI;            // tile: rows x 32-site words (TR divides Y, TWI divides 32 gx)
	int xcd_rows;           // > 0: tiles per XCD -- the tiles are dealt to the 8 XCDs in bands (neighbouring tiles share an L2)
	unsigned long long *cnt; // not null: the up spins of the state the launch stores are added here (a print point, ising_sweep_counted)
};
hipError_t launch_dense_tiles(const TileParams &p, int threads, hipStream_t stream);
size_t dense_tiles_lds_bytes(const TileParams &p);
hipError_t launch_dense_init(const InitParams &p, hipStream_t stream);
// Small and narrow lattices, round 5 (ising_quad.hip): the "quad" layout -- per colour [Y/4 row groups][gx blocks][64 words], bit 16 r4 + tx of word p = the site
// reference thread 64 (R & 3) + 16 r4 + tx of block (bx, R / 4) draws with Philox block p / 4, output p % 4 -- and its kernel, one launch per pass of T sweeps:
// tiles (the word phases of 2 T levels on C row groups + halo, no exchange) next to the draws of the pass to come (one KiB of accept masks per level, row group
// and block, made once).
// Many small lattices of one shape in ONE launch (round 6, ising_batch.cpp): the tiles of all of them side by side, the drawing workgroups draw for all of
// them.  What differs from lattice to lattice travels in a record the kernel fetches with scalar loads (as ballot_update_k<BATCH> does with ReplicaParams).
struct alignas(64) QuadRec {
	uint64_t *quad;          // the lattice's four planes [buffer][colour][quad words] (ising_ctx::d_quad)
	uint64_t *masks;         // its two mask buffers (ising_ctx::d_qmasks)
	uint32_t *dense[2];      // row 0 of its dense planes (the conversion launches on a call's way in and out)
	uint32_t seed_lo, seed_hi;
	uint32_t n3, n4;
	uint32_t pad_[4];
};
struct QuadDrawParams {
	uint64_t *masks;         // [nlev][NRG * gx][128]: (c3, c4) of word p at 16 p bytes
	uint32_t seed_lo, seed_hi;
	uint32_t it;             // iteration of level 0 (black); level L = colour L & 1 of iteration it + L / 2
	uint32_t n3, n4;
	int gx, NRG;
	int nlev;                // 0: no draws in this launch
	int nwaves;              // (set by the launcher) the drawing waves of the launch: each takes an equal run of the pass's quarter items
	const QuadRec *rep;      // not null: `nrep` lattices, lattice major -- seed and thresholds from the records, masks = rep[r].masks + mask_off
	int nrep;
	size_t mask_off;         // (words)
};
struct QuadWordParams {
	const uint64_t *src[2];  // black / white lattice the pass reads
	uint64_t *dst[2];        // ... and writes (another buffer: the neighbours read their halos from src meanwhile)
	const uint64_t *masks;   // level 0 of this pass
	int gx, NRG;
	int C, HG;               // row groups per tile, halo row groups per side (4 HG >= nlev - 1)
	int nlev;                // levels of the pass (even: whole sweeps, black first; at most 64); 0: no word pass in this launch
	unsigned long long *cnt; // not null: the up spins of the state the pass stores are added to these EIGHT words (tile t to word t mod 8)
	unsigned long long *cnt_eq; // not null (with cnt): ... and the bonds between equal spins (ising_bond_equal's sum: the white sites' equal neighbours) to these eight
	const QuadRec *rep;      // not null: tile t belongs to lattice t / tiles_per_lat; planes = rep[r].quad + src_off / dst_off, masks = rep[r].masks + mask_off,
	int nrep;                // print points: cnt / cnt_eq + r * cnt_stride
	int tiles_per_lat;       // (set by the launcher)
	size_t src_off[2], dst_off[2], mask_off, cnt_stride; // (words)
};
struct QuadPassParams {
	QuadWordParams w;
	QuadDrawParams d;
	int cus;                 // compute units of the device
	int ntiles;              // (set by the launcher) workgroups [0, ntiles) are tiles, the rest draw
};
hipError_t launch_quad_pass(QuadPassParams &p, int waves, hipStream_t stream);
#if defined(ISING_QUAD_TRACE)
void quad_trace_dump(); // measurement builds only (ising_quad.hip)
#endif
size_t quad_pass_lds_bytes(const QuadWordParams &w, int waves);
int quad_word_maxi(const QuadWordParams &p, int waves); // items a wave works on per level at most (0: too many for any instantiation)
hipError_t launch_dense_to_quad(const uint32_t *dense, uint64_t *quad, int gx, int NRG, hipStream_t stream);
hipError_t launch_quad_to_dense(const uint64_t *quad, uint32_t *dense, int gx, int NRG, hipStream_t stream); // (dense: row 0; rows -1 and Y are refreshed too)
// ... of every lattice of a batch, both colours, in one launch: rep[r].dense[colour] <-> rep[r].quad + (2 buffer + colour) * gx * NRG * 64
hipError_t launch_quad_convert_batch(const QuadRec *rep, int nrep, int buffer, bool to_quad, int gx, int NRG, hipStream_t stream);

// in-place nibble -> bit-plane transposition of `nvec` 16-byte coupling vectors (dense layout with -J)
hipError_t launch_ham_planes(uint64_t *ham, size_t nvec, hipStream_t stream);
hipError_t launch_swap_vectors(uint64_t *a, uint64_t *b, size_t nvec, hipStream_t stream); // 16-byte vectors of two arrays change places
hipError_t launch_dense_bond_equal(const BondParams &p, hipStream_t stream);
hipError_t launch_dense_pack_bits(const uint64_t *black, const uint64_t *white, int wpr, int Y, uint32_t row_base, uint32_t *bits,
                                  hipStream_t stream);

// boundary format: `nvec` dense 32-bit words (one reference 128-bit vector each) <-> 2 * nvec packed 64-bit words
hipError_t launch_dense_to_packed(const uint32_t *dense, uint64_t *packed, size_t nvec, hipStream_t stream);
hipError_t launch_packed_to_dense(const uint64_t *packed, uint32_t *dense, size_t nvec, hipStream_t stream);

// ballot layout (1 bit per spin in wave-ballot order, ising_ballot.hip): integer-threshold update, conversions
// `p` is completed by the launcher (nwg); *grid_out = workgroups launched
// `stop` / `start` (optional): events that fire when the launch is done / begins (they ride on the dispatch packet)
#if defined(ISING_FUSED_TRACE)
void ballot_trace_dump(); // measurement builds only (ising_ballot.hip)
#endif
hipError_t launch_ballot_update(UpdateParams &p, hipStream_t stream, int *grid_out, hipEvent_t stop = nullptr, hipEvent_t start = nullptr);
hipError_t launch_ballot_split(UpdateParams &p, hipStream_t stream, int *grid_out, hipEvent_t stop = nullptr, hipEvent_t start = nullptr);
int ballot_max_wgs(int cus); // upper bound of the grid of any ballot launch (scratch sizing)
// up-spin count and black-site bond sum of `nrep` ballot lattices (gx, Y each; reps[r].lat[]), spread over BALLOT_MEASURE_SLOTS
// accumulator pairs per lattice, 64 bytes apart: acc[(r * SLOTS + s) * 8 + {0, 1}]
constexpr int BALLOT_MEASURE_SLOTS = 16;
hipError_t launch_ballot_measure(const ReplicaParams *reps, int nrep, int gx, int Y, unsigned long long *acc, hipStream_t stream);
// one lattice: its slots' sums into out[0] (up spins) and out[1] (bond sum); the slots are zero again afterwards
hipError_t launch_measure_fold(unsigned long long *acc, unsigned long long *out, hipStream_t stream);
// in-launch counts: out[2 m] = the sum of the first n_up of measurement m's per_meas slots (up spins), out[2 m + 1] = the sum of the rest (equal bonds), m = 0 .. nmeas - 1
hipError_t launch_count_fold(const uint32_t *slots, size_t per_meas, size_t n_up, int nmeas, unsigned long long *out, hipStream_t stream);
hipError_t launch_ballot_init(const InitParams &p, hipStream_t stream);
hipError_t launch_ballot_to_dense(const uint64_t *bal, uint32_t *dense, int gx, long long rows, hipStream_t stream);
hipError_t launch_dense_to_ballot(const uint32_t *dense, uint64_t *bal, int gx, long long rows, hipStream_t stream);
// -J coupling rows in place: nibble form -> ballot planes; ballot planes -> the dense layout's per-vector planes
hipError_t launch_ham_to_ballot(uint64_t *ham, int gx, long long rows, hipStream_t stream);
hipError_t launch_ham_ballot_to_planes(uint64_t *ham, int gx, long long rows, hipStream_t stream);

// one-bit-per-spin image of a slab in lattice-column order: bits[Y][lld] 32-bit words
hipError_t launch_pack_bits(const uint64_t *black, const uint64_t *white, int lld, int Y, uint32_t row_base, uint32_t *bits,
                            hipStream_t stream);
// two-point sums for distances 1..ncorr over slab rows 0..Y-1; `bits` must hold Y + ncorr rows
hipError_t launch_corr(const uint32_t *bits, int lld, int Y, int ncorr, int slW, int slY, long long *sums, hipStream_t stream);

} // namespace ising
