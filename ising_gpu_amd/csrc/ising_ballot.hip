// ising_ballot.hip -- the "ballot" device layout: 1 bit per spin, bits ordered the way the wave produces them.
//
// The dense layout (ising_dense.hip) keeps the reference's thread<->vector mapping, so every lane has to collect the
// accept decisions of its own 64 sites into its own registers: 8 v_cmpx + 8 masked ORs per draw block, ~30 % of the
// kernel's VALU cycles.  A plain v_cmp hands the same 64 decisions -- one per lane -- to the scalar unit as a 64-bit
// lane mask ("ballot") for free.  This layout makes that mask THE storage word:
//
//   word (row, wc, p),  p = 32 j + 4 m + q   <->  draw block B = 8 j + m, Philox output q of that block
//   bit l = 16 g + tx of the word             <->  the lane that drew it: vector column (4 wc + g) 32 + 16 j + tx,
//                                                  site s(m, q) = {2m, 16+2m, 2m+1, 17+2m}[q] of that vector
//
// i.e. a row is X/8192 "wave columns" of 64 words of 64 bits (X/16 bytes per colour row, the same as the dense
// layout); the Philox stream assignment (tid, counter, output -> site) is the reference's, untouched.
//
// Update of one row by one wave:
//   draw phase   16 Philox blocks per lane as everywhere else; per block 8 v_cmp (2 thresholds x 4 outputs) whose
//                SGPR-pair results go to a per-wave scratch slot with scalar stores (no VALU, no EXEC games);
//                s_dcache_wb once per row pushes the slot to L2.
//   word phase   one row later (the write-back has long finished), lane p owns word p of the row: it reads its two
//                accept masks back (16 bytes, L1-bypassing load), the words above/below from memory and its side word
//                from another lane (ds_bpermute: site s-1 / s+1 of the same vector is the same bit of another word;
//                only sites 0 / 31 reach into the neighbouring vector = the neighbouring lane bit: those 2 words per
//                row are put together on the scalar unit and dropped in with v_writelane), then the same bit-sliced
//                adder and Metropolis mask as the dense kernel, 64 sites per lane.
//
// Scope: the integer-threshold fast path (with or without -J couplings), X a multiple of 8192, sub-lattice widths of 2048,
// 4096 or a multiple of 8192.  Everything else (generic FP32 kernel, other widths) runs on the dense layout; ising_capi.cpp
// converts with the two kernels at the end of this file and uses the dense kernels for the observables that need
// neighbour geometry.
#include "ising_device.hpp"
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

namespace ising {
namespace {

#if !defined(ISING_BAL_THREADS)
#define ISING_BAL_THREADS 256
#endif
// The 16 scalar registers the draw phase's compares write and its scalar stores read are named in the asm text (inline asm
// cannot name halves of an SGPR tuple operand): s[BASE .. BASE+15], BASE a multiple of 4.
#if !defined(ISING_BAL_SGPR_BASE)
#define ISING_BAL_SGPR_BASE 84
#endif
#define ISING_STR2(x) #x
#define ISING_STR(x) ISING_STR2(x)
#define SG(a, b) "s[" ISING_STR(ISING_BAL_SGPR_BASE) "+" #a ":" ISING_STR(ISING_BAL_SGPR_BASE) "+" #b "]"
#if ISING_BAL_SGPR_BASE == 84
#define BAL_CLOB8 "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91"
#define BAL_CLOB16 BAL_CLOB8, "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"
#elif ISING_BAL_SGPR_BASE == 40
#define BAL_CLOB8 "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47"
#define BAL_CLOB16 BAL_CLOB8, "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"
#else
#error "ISING_BAL_SGPR_BASE: 84 or 40"
#endif
constexpr int BAL_THREADS = ISING_BAL_THREADS; // waves of a workgroup share one scalar-cache write-back per row
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint64_t LANE0 = 0x0001000100010001ull;  // tx = 0 of each 16-lane group
constexpr uint64_t LANE15 = 0x8000800080008000ull; // tx = 15

__device__ __forceinline__ constexpr int word_of(int j, int m, int q) { return 32 * j + 4 * m + q; }

__device__ __forceinline__ uint64_t bperm64(int src_lane, uint64_t v) {
	const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)v);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)(v >> 32));
	return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t flips64(uint64_t me, uint64_t up, uint64_t ct, uint64_t dw, uint64_t sd, uint64_t c3, uint64_t c4) {
	const uint32_t lo = flips32((uint32_t)me, (uint32_t)up, (uint32_t)ct, (uint32_t)dw, (uint32_t)sd, (uint32_t)c3, (uint32_t)c4);
	const uint32_t hi = flips32((uint32_t)(me >> 32), (uint32_t)(up >> 32), (uint32_t)(ct >> 32), (uint32_t)(dw >> 32), (uint32_t)(sd >> 32),
	                            (uint32_t)(c3 >> 32), (uint32_t)(c4 >> 32));
	return ((uint64_t)hi << 32) | lo;
}

// Plain or agent-coherent access to lattice words.  COH (fused launches, where another workgroup of the SAME launch
// wrote the word): relaxed agent-scope atomics = global_load/store ... sc1 -- loads bypass the per-CU vector L1, stores
// are written through, so the word is visible chip-wide once the storing wave's vmcnt has drained.
template <bool COH>
__device__ __forceinline__ uint64_t ld_word(const uint64_t *q) {
	if (COH) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return *q;
}
template <bool COH>
__device__ __forceinline__ void st_word(uint64_t *q, uint64_t v) {
	if (COH) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else *q = v;
}
// The row loop of a fused launch issues its lattice loads and stores as inline assembly and waits for them by hand
// (vmcnt counts in order).  With compiler-tracked atomic loads / stores in that loop the compiler protected registers of
// pending operations with s_waitcnt vmcnt(0) at the top of every row -- a wait for the previous row's write-through
// stores, which have nothing to do with the draw phase that follows.  `base` is wave-uniform, `off` the lane's byte offset.
// STREAM: the lattice words carry the non-temporal hint, so that a lattice larger than the memory-side cache does not push
// the accept-mask slots out of the L2s on its way through (65536^2: 0.930 -> 0.889 GB of HBM traffic per colour half-sweep,
// same speed; a lattice that fits the 256 MB cache is re-read from it every level and must not be marked: -4 % at 2^27).
template <bool STREAM>
__device__ __forceinline__ void ld64_coh_issue(uint64_t &v, const uint64_t *base, int off) {
	if (STREAM) asm volatile("global_load_dwordx2 %0, %1, %2 sc1 nt" : "=&v"(v) : "v"(off), "s"(base) : "memory");
	else asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "=&v"(v) : "v"(off), "s"(base) : "memory");
}
template <bool STREAM>
__device__ __forceinline__ void st64_coh_issue(uint64_t *base, int off, uint64_t v) {
	if (STREAM) asm volatile("global_store_dwordx2 %0, %1, %2 sc1 nt" :: "v"(off), "v"(v), "s"(base) : "memory");
	else asm volatile("global_store_dwordx2 %0, %1, %2 sc1" :: "v"(off), "v"(v), "s"(base) : "memory");
}
// wave-uniform values the compiler may have left in vector registers (ticket arithmetic): pin them to the scalar unit
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
	return ((uint64_t)hi << 32) | lo;
}

// Plain launch (FUSED = false): one colour half-sweep, one workgroup per unit (4 waves = 4 wave columns x one strip of H
// rows), handed out by the hardware dispatcher.
// FUSED: one launch carries `nlevels` colour half-sweeps (black, white, black, ...) and as many workgroups as the chip
// holds, each drawing (level, unit) tickets IN ORDER from one counter until the work is gone.  A strip of level L reads
// the rows of strips s-1, s, s+1 written at level L-1 and overwrites rows those three strips read at level L-1, so it
// waits until their completion counters show level L-1 done.  Tickets are handed out level by level and everything a
// ticket waits for has a lower number, so the lowest unfinished ticket is always being worked on by a running
// workgroup: no deadlock whatever the residency or whatever else runs on the chip.  The accept-mask slots belong to the
// workgroup SLOT (blockIdx): ~12 MB that stay in the L2s, where a plain launch of 65536^2 spreads 134 MB of slots that
// spill to HBM (0.89 vs 1.14 GB of traffic per colour half-sweep, free of charge under a saturated vector ALU but traffic
// all the same).  The chip never drains between colours.  A unit's parents are one level of tickets back: the host picks
// strip height and grid size so that they are done when the unit starts (ising_create); from 1.5 * 2^24 spins up
// this form is what ising_sweep launches.
// When a workgroup draws its next ticket: in a unit's LAST iteration, waited for on the spot (~2 us per unit).  Drawn a unit
// ahead -- during the unit's first row, this kernel's first form -- a ticket sits reserved while its workgroup finishes the
// current unit and later tickets start before it, so units of the next level find their parents unfinished (65536^2, H = 16:
// 12.5 M polls that slept per 2.1 M units -> none, 3360 -> 3463 flips/ns; 16384^2, H = 4: 9.7 M -> 3.5 M, 2892 -> 3047;
// requested one word phase early: no gain.  tools/trace_probe.py, profiles/fused_trace_r02.txt; the variants are in the
// history of this file up to round 2.)
// Wave priorities in fused launches.  The SIMD's arbiter serves its waves oldest first, and the waves of a persistent grid
// keep their age: under a saturated vector ALU the youngest wave of a SIMD gets what the others leave.  Units of the same work
// then take 1x .. 5x as long (a -DISING_FUSED_TRACE -DISING_FUSED_TRACE_COUNTS build: 43 % of the units of 65536^2 in 100-130 k
// cycles, 9 % in more than 490 k), and the slow ones are the parents the next level finds unfinished: 6 % of the units slept
// ~90 polls each at five workgroups per CU, more at six.  Every wave therefore steps through the four priorities row by
// row, offset by its dispatch round (= its rank on the SIMD), so the waves of a SIMD take turns at the front: no sleeping
// polls left at 65536^2 (3479 -> 3515 flips/ns, 3534 with six workgroups per CU), 16384^2 3082 -> 3291, 16384 x 8192 with
// 4-wave workgroups 2709 -> 3041.  (Measured and dropped, round 2: feedback once per unit from the tickets drawn meanwhile --
// the same spread, alternating; four steps per row; a row count that runs on across units; the rotation in
// one-launch-per-colour launches, 3458 -> 3389: their workgroups are not a persistent grid; the fifth and sixth wave of a SIMD
// stepping the other way round.  What remains uneven: the fifth and sixth wave still own all the units that take three times
// the median, 7 % of the units at 65536^2.)
#ifndef ISING_FUSED_WAIT_LATE // 1: fused launches may ask their units to draw before they wait for their parents (UpdateParams.wait_late); 0 compiles the request out
#define ISING_FUSED_WAIT_LATE 1
#endif
#ifndef ISING_POLL_SLEEP // s_sleep units (64 cycles) between two looks of a waiting unit at its parents' counters
#define ISING_POLL_SLEEP 32
#endif
#ifndef ISING_FUSED_STAGGER // s_sleep units (64 cycles each) between the start of successive dispatch rounds of a fused launch
#define ISING_FUSED_STAGGER 100
#endif
// Measurement build (make variant DEFS=-DISING_FUSED_TRACE): wave 0 of every workgroup of a fused launch clocks where its
// time goes (s_memtime between the marks below); ballot_trace_dump() prints the chip-wide sums when the slab is
// destroyed.  Never in the product library.
#if defined(ISING_FUSED_TRACE)
__device__ unsigned long long g_trace[16];
__device__ unsigned long long g_hist[32];
#if defined(ISING_FUSED_TRACE_COUNTS) // counts only (units, sleeping polls): no clock reads, the product kernel's residency
#define TRC(i) do {} while (0)
#else
#define TRC(i) do { if (FUSED && wi == 0) { const long long t_ = clock64(); if (lane == 0) tr[i] += (unsigned long long)(t_ - tlast); tlast = t_; } } while (0)
#endif
#define TRN(i, n) do { if (FUSED && wi == 0 && lane == 0) tr[i] += (unsigned long long)(n); } while (0)
#else
#define TRC(i) do {} while (0)
#define TRN(i, n) do {} while (0)
#endif
#if defined(ISING_BAL_NUM_SGPR) // A/B: cap the scalar registers (80: eight workgroups per CU instead of six)
#define BAL_SGPR_ATTR __attribute__((amdgpu_num_sgpr(ISING_BAL_NUM_SGPR)))
#else
#define BAL_SGPR_ATTR
#endif
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
// (Round 4 tried fused launches WITHOUT tickets -- every workgroup of a wholly resident grid owning units b, b + G, ... of every level,
// no atomics, the verdict's "static strip ownership" -- and measured them 6-10 % behind the ticket form at every size and shape
// (16384^2: 3119 vs 3321 flips/ns at two units per workgroup and level, 2967 vs 3211 at one; 8192^2 2510 vs 2793; 32768^2 3179 vs 3475;
// profiles/static_probe_r04a.txt, _r04b.txt; the code is in the history at 851dc8e): the 6 % of a workgroup's time that tickets cost
// buy a load balance between workgroups of unequal speed that is worth more -- with fixed owners every level waits for its slowest.)
// COUNT (round 4; fused launches of a lone lattice, and of a ring slab over its own rows): the up-spin count of the reference's print points (countSpins every `-p` sweeps,
// optimized/main.cu:1806-1810) is taken INSIDE the launch.  A print point every 16 sweeps otherwise cuts the launches into pieces of 16
// with two count launches and a read-back in between (16384^2: 3057 against 3289 flips/ns, 8192^2 2539 against 3023).  Here every unit
// of a measured sweep's two levels takes the popcount of the words it stores -- two v_bcnt per row -- and leaves its sum in a slot of
// its own (measurement, colour, wave); a small kernel behind the launches adds the slots up.  (Plain stores: the first form added to 64
// accumulators per measurement with atomics -- 32768 waves a level at 65536^2 queued at the L2 for ~0.4 ms a level, -4 %.)
template <bool SUBL, bool USEJ, bool FUSED, int NT = BAL_THREADS, bool STREAM = false, bool BATCH = false, bool COUNT = false>
__global__ void __launch_bounds__(NT) BAL_SGPR_ATTR ballot_update_k(const UpdateParams p) {
	static_assert(!COUNT || (FUSED && !SUBL && !USEJ && !BATCH), "in-launch counts: fused launches of one lattice, no sub-lattices, no couplings");
	static_assert(!BATCH || (FUSED && !SUBL && !USEJ), "batched launches: fused, no sub-lattices, no couplings");
	static_assert(!(FUSED && SUBL && USEJ), "fused launches with sub-lattices: no couplings");
	const int lane = threadIdx.x & 63;
	const int tx = threadIdx.x & (GROUP - 1), g = (threadIdx.x >> 4) & 3;
	const int wi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// A row holds ceil(gx / 4) wave columns of 64 words; when gx is not a multiple of 4 the last one is partly dead: its
	// lanes of column groups >= gx draw and compute like the others, but their bits stay zero in memory.
	const int nwc = (p.gx + 3) >> 2;
	const int gxp = nwc << 2; // column groups including the dead ones
	const int wpr = nwc * 64; // 64-bit words per colour row

	// word phase: this lane owns word p = lane of every row
	const int j = lane >> 5, m = (lane >> 2) & 7, q = lane & 3;
	int backA, fwdA; // lane holding the word with site s-1 / s+1 of the same vectors
	if (q == 2) backA = word_of(j, m, 0);
	else if (q == 3) backA = word_of(j, m, 1);
	else if (q == 0) backA = m ? word_of(j, m - 1, 2) : word_of(j, 7, 3);
	else backA = m ? word_of(j, m - 1, 3) : word_of(j, 7, 2);
	if (q == 0) fwdA = word_of(j, m, 2);
	else if (q == 1) fwdA = word_of(j, m, 3);
	else if (q == 2) fwdA = m < 7 ? word_of(j, m + 1, 0) : word_of(j, 0, 1);
	else fwdA = m < 7 ? word_of(j, m + 1, 1) : word_of(j, 0, 0);
	// Sites 0 (back) / 31 (forward) have their side neighbour in the adjacent vector = the adjacent lane's bit of word
	// (., 7, 3) / (., 0, 0), where lanes tx = 0 / 15 cross into the other j, the next group or the next wave column:
	// two words per row and direction -- (0,0,0), (1,0,0) back, (0,7,3), (1,7,3) forward -- are assembled from two
	// words of this row (A0, A1) and one word of another wave column (C).  That is a handful of 64-bit shifts and
	// masks on wave-uniform data: the scalar unit does it (A0, A1, C as scalars, s_lshl/s_and/s_or) and v_writelane
	// drops the results into the two lanes; every other lane takes its ds_bpermute word as it is.
	// The row is periodic every k = slV/32 column groups (k = gx without sub-lattices, optimized/main.cu:413-459): the
	// first vector of a period takes its back neighbour from the period's last vector -- for k = 1, 2 a bit of this
	// wave's own words (shift 15 / 31, SUBL only), for k = 4n a bit of wave column wc + n - 1 instead of wc - 1 (same
	// C term, other address).  Forward is the mirror image.
	const int k = p.slV >> 5;                       // column groups per period
	const bool inwave = SUBL && k < 4;              // periods shorter than a wave column
	const int wsh = k == 1 ? 15 : 31;
	const int n = SUBL ? max(k >> 2, 1) : nwc; // wave columns per period
	// where copies of row 0 / row Y-1 of the updated colour go, as offsets from those rows: this slab's own halo rows
	// Y / -1 (single slab: periodic wrap) or the neighbouring slabs' halo rows (ring on one device: no copies needed)
	const ptrdiff_t mir0 = (ptrdiff_t)(p.mir0_bytes / 8), mirL = (ptrdiff_t)(p.mirL_bytes / 8);

	// this workgroup slot's scratch: per wave two slots of 64 x (c3, c4) masks
	const uint64_t *slot_v = p.scratch + ((size_t)blockIdx.x * (NT / 64) + wi) * 256;
	const uint32_t slot_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)slot_v);
	const uint32_t slot_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)slot_v >> 32));
	uint64_t *slot = reinterpret_cast<uint64_t *>(((uintptr_t)slot_hi << 32) | slot_lo);

	// The block-constant (wave-uniform) first two Philox rounds of the 16 draw blocks, three values each, live in LDS:
	// as SGPRs they overflow the register file or, recomputed per row, make the XORs that consume them 4-cycle
	// SGPR-operand instructions; from LDS they arrive in VGPRs (2-cycle XORs, +1.5 %).  One private copy per wave.
	__shared__ uint4 blk_const_all[NT / 64][16];
	__shared__ unsigned long long ticket_sh[2];
	uint4 *blk_const = blk_const_all[wi];
	const unsigned long long total = (unsigned long long)p.nlevels * (unsigned long long)p.nwg;

#if defined(ISING_FUSED_TRACE)
	__shared__ unsigned long long tr[16];
#if defined(ISING_FUSED_TRACE_COUNTS)
	__shared__ unsigned int hist_sh[2][16];
	if (threadIdx.x < 32) hist_sh[threadIdx.x >> 4][threadIdx.x & 15] = 0;
#endif
	if (threadIdx.x < 16) tr[threadIdx.x] = 0;
	__syncthreads();
	[[maybe_unused]] long long tlast = clock64();
	const long long tstart = tlast;
#endif
	// Fused launches: thread 0 draws the NEXT ticket in a unit's last iteration and leaves it in LDS before that
	// iteration's barrier, where the workgroup picks it up.  (Built with -amdgpu-atomic-optimizer-strategy=None: the
	// wave-aggregating rewrite of atomicAdd needs the result on the spot.)
	// The counter is never reset: a launch's tickets start at p.ticket_base2[] = where the launches before it left the counter
	// (each of their `grid` workgroups drew exactly one ticket past its launch's last), which the host keeps count of --
	// a memset in front of every launch is a fill kernel of its own and two more dependencies in the stream.
	// Row of the whole lattice behind slab row r.  Ghost rows of a ring slab (r < 0, r >= Y: ising_ring.cpp, sweep_deep) are
	// rows of the neighbouring slabs, around the ring: their draws must be the ones their owners make.
	auto global_row = [&](int r) -> uint32_t {
		int gr = (int)p.row_base + r;
		if (p.total_rows) gr = gr < 0 ? gr + p.total_rows : (gr >= p.total_rows ? gr - p.total_rows : gr);
		return (uint32_t)gr;
	};
	// Small lattices with short units draw tickets faster than one counter hands them out (~80 per us chip-wide): then two
	// counters (p.tickets2, 64 bytes apart) -- workgroups with an even blockIdx hand out the even unit numbers in order, the
	// odd ones the odd numbers.  The lowest unfinished unit is still held by a running workgroup of its class or about to be
	// drawn by one, so nothing waits for ever.  8192^2: 2460 -> 2527 flips/ns; from 2^27 spins up one counter is faster.
	const unsigned ncnt = p.tickets2 > 1 ? (unsigned)p.tickets2 : 1u; // 1, 2 or 4 counters
	const unsigned cls = blockIdx.x & (ncnt - 1u);
	auto draw_ticket = [&]() {
		const unsigned long long t = atomicAdd(p.ticket + 8 * cls, 1ull) - p.ticket_base2[cls];
		return (unsigned long long)ncnt * t + cls;
	};
	unsigned long long tkv = 0; // this workgroup's ticket as read from LDS (every lane the same value)
	// The workgroups of a launch start a fraction of a row apart (by the round of 256 they were dispatched in) instead of
	// in lockstep -- all drawing, then all waiting: +0.3..0.6 % on whole runs, more on short launches (4-wave form;
	// -DISING_FUSED_STAGGER=0 switches it off; 8-wave workgroups measured -2 % with it, plain launches -0.3 %).
	const unsigned dround = uni((int)(blockIdx.x / (unsigned)p.cus)); // dispatch round = this workgroup's rank on its CU (a grid of k x CUs lands k per CU)
	if (FUSED && NT == 256 && ISING_FUSED_STAGGER > 0)
		for (unsigned i = 0; i < dround % 6u; ++i) __builtin_amdgcn_s_sleep(ISING_FUSED_STAGGER);
	// Measurement aid (ising_kernel_clock): the first eight workgroups of a fused launch -- one per XCD -- leave the shader's cycle counter and the
	// constant 100 MHz counter in UpdateParams.clk_out when they start and when they leave: cycles / time = the clock the launch ran at.
	// (The pointer is fetched from the kernel arguments on the spot: no register is kept alive for it.)
	auto clock_mark = [&](int which) {
		if (!FUSED || blockIdx.x >= 8u || wi != 0) return;
		uint64_t cp;
		asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(cp) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(offsetof(UpdateParams, clk_out)) : "memory");
		if (cp == 0) return;
		const unsigned long long cyc = __builtin_readcyclecounter(), ref = __builtin_amdgcn_s_memrealtime();
		if (lane == 0) {
			unsigned long long *o = reinterpret_cast<unsigned long long *>(cp) + 4 * blockIdx.x + 2 * which;
			o[0] = cyc;
			o[1] = ref;
		}
	};
	clock_mark(0);
	if (FUSED) {
		if (threadIdx.x == 0) ticket_sh[0] = draw_ticket();
		__syncthreads();
		tkv = ticket_sh[0];
	}
	// a workgroup's tickets grow, so its level is a running count (no 64-bit division per unit)
	int level = 0;
	unsigned long long level_base = 0;
	const int nwc_sh = (nwc & (nwc - 1)) == 0 ? __builtin_ctz((unsigned)nwc) + 2 : -1; // gxp = 4 nwc as a shift where it is one
	for (int round = 0;; ++round) {
		unsigned long long tk;
		if (FUSED) {
			const uint32_t tk_lo = __builtin_amdgcn_readfirstlane((uint32_t)tkv), tk_hi = __builtin_amdgcn_readfirstlane((uint32_t)(tkv >> 32));
			tk = ((unsigned long long)tk_hi << 32) | tk_lo;
		} else {
			tk = round ? total : (unsigned long long)blockIdx.x; // plain: one unit per workgroup
		}
		if (tk >= total) break;
		TRC(0); // ticket pick-up
#if defined(ISING_FUSED_TRACE) && defined(ISING_FUSED_TRACE_COUNTS)
		const long long t_unit0 = clock64();
		long long t_unit1 = t_unit0;
#endif
		TRN(8, 1);
		if (FUSED) {
			while (tk >= level_base + (unsigned)p.nwg) {
				level_base += (unsigned)p.nwg;
				++level;
			}
		}
		// batched launch: which lattice this unit belongs to, and that lattice's record (scalar cache: invalidated at kernel start)
		int wgi = (int)(tk - level_base), rep = 0;
		u32x8 rb = {0, 0, 0, 0, 0, 0, 0, 0};
		if (BATCH) {
			int q = p.nwg_rep == 1 ? wgi : (int)__umulhi((uint32_t)wgi, p.rep_magic); // (the reciprocal of 1 does not fit 32 bits)
			if (q * p.nwg_rep > wgi) --q; // (the reciprocal rounds up: at most one too many)
			rep = uni(q);                 // (pinned to the scalar unit: the compiler takes the running level count for lane-dependent)
			wgi = uni(wgi - rep * p.nwg_rep);
			const uintptr_t ra = reinterpret_cast<uintptr_t>(p.rep + rep);
			const ReplicaParams *rp = reinterpret_cast<const ReplicaParams *>(((uintptr_t)uni((uint32_t)(ra >> 32)) << 32) | uni((uint32_t)ra));
			asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rb) : "s"(rp) : "memory");
		}
		// (not const: a const integer is not captured by the generic lambda of the draw phase)
		uint32_t seed_lo = BATCH ? rb[6] : p.seed_lo, seed_hi = BATCH ? rb[7] : p.seed_hi;
		uint32_t thr3 = BATCH ? rb[4] : p.n3, thr4 = BATCH ? rb[5] : p.n4;
		const uint32_t k2y = seed_hi + 2u * PHILOX_W1;
		const int wave = uni(wgi * (NT / 64) + wi);
		const int unit0 = wave * 4; // a wave covers 4 consecutive 32-vector column groups of one strip (gx % 4 == 0)
		// Two row ranges per launch: the two edge rows of a ring slab, or (tail strips) the bulk of the slab in strips of H
		// rows followed by its last rows in strips of H2 < H rows -- the hardware dispatches workgroups in index order, so
		// the launch ends on many short units instead of a few long ones.  Range 0 is padded to whole workgroups (nunits0
		// vs nreal0) so that the waves of a workgroup share one strip height.
		const int rng = unit0 >= p.nunits0;
		const bool absent = rng ? unit0 >= p.nunits : unit0 >= p.nreal0; // partly empty workgroups still meet the barriers
		const int u = absent ? 0 : unit0 - (rng ? p.nunits0 : 0);
		const int Hr = (rng && p.H2) ? p.H2 : p.H; // strip height of this range (uniform over the workgroup)
		int pos;
		if (nwc_sh >= 0) pos = u >> nwc_sh;
		else pos = uni(u / gxp);
		const int bx0 = u - pos * gxp;
		const int wc = bx0 >> 2;
		const int bx = bx0 + g;
		// FUSED: strips are taken from both ends of the slab inwards (0, N-1, 1, N-2, ...), so that the periodic
		// neighbours strip 0 and strip N-1 wait for are the first tickets of the previous level, not its last
		const int nstr = (p.row_hi[0] - p.row_lo[0] + p.H - 1) / p.H;
		const int sidx = FUSED ? uni((pos & 1) ? nstr - 1 - (pos >> 1) : (pos >> 1)) : pos;
		const int r0 = p.row_lo[rng] + sidx * Hr;
		const int nrows0 = absent ? 0 : min(Hr, p.row_hi[rng] - r0);
		// ring slab with ghost rows: at this level only rows within `keep` of the slab's own can still reach one of them; a strip
		// wholly further out has nothing to do (it still waits for its parents and counts as complete: its counter is a count)
		// (several exchange epochs per launch, UpdateParams.epoch_sh: the trapezoid starts over with every epoch)
		const int elev = uni(FUSED && p.epoch_sh ? level & ((1 << p.epoch_sh) - 1) : level);   // the level within its epoch
		const int elast = uni(FUSED && p.epoch_sh ? (1 << p.epoch_sh) - 1 : p.nlevels - 1);     // an epoch's last level
		const int keep = uni(elast - elev); // (pinned: the compiler takes the running level count for lane-dependent)
		const bool skip = FUSED && p.trapezoid != 0 && !absent && (r0 + nrows0 <= -keep || r0 >= p.Y + keep);
		const bool idle = absent || skip;
		const int nrows = skip ? 0 : nrows0;
		// ... and, exchange overlapped with the launches: a unit that touches rows the exchange reads or writes
		const bool edge_unit = FUSED && p.edge_go != nullptr && !absent && (r0 < p.edge_lo || r0 + nrows0 > p.edge_hi);
		const uint32_t color = FUSED ? uni((p.color + (uint32_t)level) & 1u) : p.color;
		const uint32_t it = FUSED ? uni(p.it + ((p.color + (uint32_t)level) >> 1)) : p.it;
		uint64_t *const lat0 = BATCH ? reinterpret_cast<uint64_t *>(((uintptr_t)rb[1] << 32) | rb[0]) : p.lat[0];
		uint64_t *const lat1 = BATCH ? reinterpret_cast<uint64_t *>(((uintptr_t)rb[3] << 32) | rb[2]) : p.lat[1];
		const uint64_t *src = FUSED ? (color ? lat0 : lat1) : p.src;
		uint64_t *dst = FUSED ? (color ? lat1 : lat0) : p.dst;

		uint64_t first = 0, last = 0;                    // bit 16g (16g + 15): group g opens (closes) a period
#pragma unroll
		for (int gg = 0; gg < 4; ++gg) { // (without sub-lattices the period is the row: k = gx)
			if (SUBL ? (bx0 + gg) % k == 0 : bx0 + gg == 0) first |= 1ull << (16 * gg);
			if (SUBL ? (bx0 + gg) % k == k - 1 : bx0 + gg == p.gx - 1) last |= 1ull << (16 * gg + 15);
		}
		const uint64_t u_b1 = LANE0 & ~1ull & ~first, u_f1 = LANE15 & ~(1ull << 63) & ~last;
		// lanes of this wave that exist (all of them unless this is the partly dead last wave column), the bit of the row's
		// last vector in a word of THIS wave column and in one of the wave column the back neighbour of lane 0 comes from
		const int alive = min(4, p.gx - bx0);
		const uint64_t live = alive >= 4 ? ~0ull : ((1ull << (16 * alive)) - 1ull);
		const int end_here = 16 * alive - 1;
		const int wcn = SUBL ? wc % n : wc; // wave column within its period (n = nwc without sub-lattices)
		const int src_b = wcn ? wc - 1 : wc + n - 1;
		const int end_src = 16 * min(4, p.gx - 4 * src_b) - 1;
		const uint64_t u_bw = inwave ? first : 0ull, u_fw = inwave ? last : 0ull;
		const int u_cb = (src_b - wc) * 64 + word_of(1, 7, 3); // C, in words from this wave's row start
		const int u_cf = ((wcn == n - 1 ? wc - n + 1 : wc + 1) - wc) * 64 + word_of(0, 0, 0);
		// rows are periodic every slY rows (SUBL; otherwise rows -1 and Y are the halo rows)
		const int slY = SUBL ? p.slY : 0;
		const int r0_in_sl = SUBL ? r0 % p.slY : 1;
		int seam = SUBL ? slY - r0_in_sl : 0x7fffffff; // rows left in the current period, this one included

		const uint64_t *rs = src + ((ptrdiff_t)r0 * wpr + wc * 64); // wave-uniform row pointers, lanes index them
		uint64_t *rd = dst + ((ptrdiff_t)r0 * wpr + wc * 64);
		// -J: per row and wave column four coupling planes {right, left, down, up} of 64 ballot-order words each
		const uint64_t *rj = USEJ ? (FUSED ? (color ? p.jham[1] : p.jham[0]) : p.jdst) + 4 * ((ptrdiff_t)r0 * wpr + wc * 64) : nullptr;

		const uint32_t cx_base = 16u * (2u * it + color);
		const uint32_t seed_lo_cy = seed_lo ^ (uint32_t)((2ull * it + color) >> 28); // see dense_update_k

		// FUSED: wait until strips s-1, s, s+1 (periodic) have completed level - 1, `nwc` wave columns each per level.  The
		// first look at their counters travels while the block constants are made.
		const bool must_wait = FUSED && level > 0 && !absent;
		const uint32_t need = p.done_base + (uint32_t)level * (uint32_t)nwc;
		const uint32_t *dp = nullptr;
		uint32_t seen = need;
		TRC(1); // unit decode
		if (must_wait) {
			int sd = sidx + (lane == 0 ? -1 : (lane == 1 ? 0 : 1));
			if (SUBL) { // rows are periodic every slY rows (a multiple of the strip height): the neighbours wrap inside the sub-lattice's strips
				const int nb = p.slY / p.H, lo = sidx - sidx % nb;
				sd = sd < lo ? sd + nb : (sd >= lo + nb ? sd - nb : sd);
			} else {
				sd = sd < 0 ? sd + nstr : (sd >= nstr ? sd - nstr : sd);
			}
			dp = p.done + (BATCH ? rep * p.done_stride : 0) + sd;
			// (the look itself is taken where the unit waits, load and wait in one statement: wait_parents below)
		}
		if (lane < 16) {
			const PhiloxBlockConst kc = philox_block_const(cx_base + (uint32_t)lane, seed_lo, seed_hi);
			blk_const[lane] = make_uint4(kc.s0, kc.s1, kc.s2, 0u);
		}
		auto wait_parents = [&]() {
			if (!must_wait) return;
			[[maybe_unused]] int nsleep = 0;
			uint32_t npoll = 0;
			// Every look at the counters loads and waits in ONE statement.  The compiler knows nothing of inline-assembly loads: a loop-carried copy of a
			// register whose load is still in flight captures the OLD value -- rounds 2-3 polled with `load ... loop { wait; test }` and the
			// generated code copied the register in front of the loop (`v_mov v0, v63` ... `v_mov v63, v0`, `s_waitcnt`): harmless while
			// the load landed outside those few cycles, wrong whenever it landed between the two copies.
			// (inline assembly like the row loop's loads: a tracked load makes the compiler guard `seen`'s register with vmcnt(0) waits all
			// through the row loop)
			if (lane < 3) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(seen) : "v"(dp) : "memory");
			for (;;) { // few, patient polls: every poll is a trip to memory that competes with the lattice traffic
				if (__all((int32_t)(seen - need) >= 0)) break;
				// Counters that never come (bases out of step with the device after a faulted launch): stop waiting.  The unit
				// that has polled abort_polls times raises a flag in pinned host memory, every waiting unit looks at it every 64th
				// poll, and whoever finds it set goes on as if its parents were done: the launch runs to its end at full speed on
				// a lattice that is garbage from here on, and the host finds the flag at its next synchronise.  (The two
				// parameters are fetched from the kernel arguments here, not at kernel entry: two scalar registers less to keep
				// alive through the row loop, which is short of them.)
				if ((++npoll & 63u) == 0u) {
					uint64_t fp;
					uint32_t bound;
					asm volatile("s_load_dwordx2 %0, %2, %3\n\ts_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(fp), "=&s"(bound)
					             : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(offsetof(UpdateParams, abort_flag)), "n"(offsetof(UpdateParams, abort_polls)) : "memory");
					uint32_t *flag = reinterpret_cast<uint32_t *>(fp);
					if (flag != nullptr) {
						if (npoll >= bound && lane == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
						if (npoll >= bound || uni(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != 0u) break;
					}
				}
				TRN(9, 1);
				TRN(15, nsleep++ == 0);
#if defined(ISING_FUSED_TRACE_COUNTS) // where the sleeping units are: by position in the level's visiting order and by level
				if (nsleep == 1) {
					TRN(0, pos < nstr / 8);
					TRN(1, pos >= nstr - nstr / 8);
					TRN(2, level == 1);
					TRN(3, level == p.nlevels - 1);
					TRN(4, pos < 8);
					TRN(5, (pos & 1) == 0);
				}
				TRN(6, (nsleep == 64));  // units that slept 64 polls and more
				TRN(7, (nsleep == 256));
#endif
				__builtin_amdgcn_s_sleep(ISING_POLL_SLEEP);
				if (lane < 3) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(seen) : "v"(dp) : "memory");
			}
			TRC(2); // completion counters (+ block constants)
			// Several epochs in one launch: an epoch's second level is the last that reads words the exchange wrote (the white rows' own words; from there on
			// every word a unit touches was stored by this launch).  Its units may run on an XCD none of the epoch's first edge units ran on, and with no launch
			// boundary in between nothing has dropped what that XCD's L2 holds of those rows from the epoch before: the first level's acquire here as well --
			// behind the wait for the parents, which waited for the exchange; the unit's rows are requested behind this point.
			if (FUSED && p.epoch_sh && edge_unit && elev == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
		};
		// Where the unit waits for its parents.  A draw phase needs nothing from the lattice -- counter words, seed, thresholds --, so a
		// launch may ask its units to draw BEFORE they wait (UpdateParams.wait_late, round 4): 1 = behind the first row's draw phase,
		// 2 (what the host asks for) = behind the second row's as well; the unit's first two rows and its first row-end word are
		// requested behind the wait.  A parent that is up to two rows' time late then costs nothing -- units of one or two rows have
		// drawn everything they will --, and a level of few tickets feeds more workgroups before its units run into each other
		// (ising_capi.cpp: fused_wgs_for; 8192^2 2767 -> 3087 flips/ns); where parents are never late it is free (65536^2 3532.8 vs 3531.8).
		const int late_mode = (FUSED && ISING_FUSED_WAIT_LATE) ? uni(p.wait_late) : 0;
		const bool wait_late = late_mode != 0;
		if (!wait_late) wait_parents();
		if (edge_unit && elev == 0) {
			// the exchange that follows the previous launch (the previous epoch of this one) has read this slab's first / last rows and filled
			// its ghost rows once the comm stream has moved the counter (usually long ago: the exchange starts when the edge strips finish
			// the last level of their epoch, a level before the launch ends or the next epoch begins)
			const uint32_t go_need = uni(p.edge_go_need + (p.epoch_sh ? (uint32_t)level >> p.epoch_sh : 0u));
			uint32_t got = go_need, ngo = 0;
			for (;;) {
				if (lane == 0) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(got) : "v"(p.edge_go) : "memory");
				if (__all((int32_t)(got - go_need) >= 0)) break;
				// (no bound of its own -- a neighbour may be seconds behind --, but the host can call the launch off)
				// (a neighbour may be seconds behind, so the bound is 16 times the one for a unit's parents -- but there is one:
				// an exchange that never completes ends the launch with an error, not at the watchdog)
				if ((++ngo & 63u) == 0u) {
					uint64_t fp;
					uint32_t bound;
					asm volatile("s_load_dwordx2 %0, %2, %3\n\ts_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(fp), "=&s"(bound)
					             : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(offsetof(UpdateParams, abort_flag)), "n"(offsetof(UpdateParams, abort_polls)) : "memory");
					uint32_t *flag = reinterpret_cast<uint32_t *>(fp);
					if (flag != nullptr) {
						const bool late = (ngo >> 4) >= bound; // (a poll here sleeps four times as long as one for the parents)
						if (late && lane == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
						if (late || uni(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != 0u) break;
					}
				}
				__builtin_amdgcn_s_sleep(127);
			}
			// rows written while this launch ran by another kernel, a copy engine -- or, IPC transport with a device per rank, by a
			// PEER device's copy into this slab's ghost rows: system scope (the lattice loads behind it bypass the vector L1 anyway)
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
		}
#if defined(ISING_FUSED_TRACE) && defined(ISING_FUSED_TRACE_COUNTS)
		t_unit1 = clock64(); // (behind the wait for the parents)
#endif
		__builtin_amdgcn_wave_barrier();
		__threadfence_block();
		uint64_t up = 0, ct = 0;
		// FUSED: the unit's first two rows as requested here; they enter the row window (up, ct) in the first word phase,
		// behind that phase's wait -- values loaded by inline assembly must not be loop-carried before they were waited for
		// (the compiler believes them ready and may copy them early)
		uint64_t up0 = 0, ct0 = 0;
		if (!idle) {
			if (FUSED) {
				// (sub-lattices, fused: strips never straddle a period -- slY is a multiple of H --, so the row above a strip's
				// first row is the only one that may lie a period away, :414)
				if (!wait_late) {
					ld64_coh_issue<STREAM>(up0, rs + ((SUBL && r0_in_sl == 0) ? (ptrdiff_t)(slY - 1) * wpr : -(ptrdiff_t)wpr), lane * 8);
					ld64_coh_issue<STREAM>(ct0, rs, lane * 8);
				}
			} else {
				up = ld_word<false>(rs + lane + ((SUBL && r0_in_sl == 0) ? (ptrdiff_t)(slY - 1) * wpr : -(ptrdiff_t)wpr));
				ct = ld_word<false>(rs + lane);
			}
		}

		// Global row and Philox stream id of the row the draw phase handles next, advanced by hand: consecutive rows are 16
		// stream ids apart, a new block row of 16 starts gx * 256 further on, and a ghost row range may run around the ring
		// once.  (Recomputed from the row number every iteration -- which the wrap forces on the compiler -- the id costs
		// five vector instructions per row: ring slabs ran 0.45 % behind a lone slab for that alone.)
		uint32_t grow_d = global_row(r0), grow_w = grow_d; // grow_w: the row of the word phase (one behind)
		uint32_t tid_d = ((grow_d >> 4) * (uint32_t)p.gx + (uint32_t)bx) * 256u + (grow_d & 15u) * 16u + (uint32_t)tx;
		// one scalar-cache write-back per workgroup and row: every wave runs the same number of iterations and meets at a barrier
		const int rmax = Hr;
		const bool wb_wave = threadIdx.x < 64;
		[[maybe_unused]] uint32_t cnt_up = 0; // COUNT: up spins among the words this lane stores in this unit
		// ... and, where the call asks for the energy too (UpdateParams.cnt_bonds), the bonds of those sites to equal neighbours: taken at the WHITE level of a
		// measured sweep, when both colours are the sweep's final ones -- every bond has exactly one white end, so the white sites' sum is ising_bond_equal's A
		[[maybe_unused]] uint32_t cnt_eq = 0;
		// which measurement of the launch this unit's sweep is (-1: none): sweeps cnt_first, cnt_first + cnt_every, ...
		[[maybe_unused]] int meas_idx = -1;
		if (COUNT && p.cnt_every > 0 && (level >> 1) >= p.cnt_first) {
			const uint32_t d = (uint32_t)((level >> 1) - p.cnt_first), e = (uint32_t)p.cnt_every;
			uint32_t qq = e == 1u ? d : __umulhi(d, p.cnt_magic);
			if (qq * e > d) --qq; // (the reciprocal rounds up: at most one too many)
			if (qq * e == d) meas_idx = uni((int)qq);
		}
		[[maybe_unused]] const bool eq_unit = COUNT && p.cnt_bonds != 0 && (level & 1) != 0 && meas_idx >= 0;
		// (Round 4 also requested the next ticket a row early -- an inline-assembly atomic at the top of the last-but-one iteration, picked up
		// behind that iteration's word-phase wait, its ~2 us under a draw phase: no gain, -1 % at 8192^2 (profiles/ticket_early_probe_r04.txt):
		// a ticket that is reserved while its workgroup still works delays the unit it names, as in round 2.  And, requested in the last iteration
		// as ever but picked up behind the last word phase instead of in front of the barrier -- the atomic under that phase's loads, the
		// ticket handed to the other waves through LDS: -0.1 .. -0.3 % everywhere (profiles/ticket_async_probe_r04.txt).  The 4 % of a
		// workgroup's time that the trace books on "next ticket" is time in which the SIMD's other waves have the vector ALU.)
		for (int r = 0; r <= rmax; ++r) {
			// rotating priorities: the waves that share a SIMD (one per dispatch round) take turns at the front
			if (FUSED) {
				// (round 5 A/B: a rotation as long as the waves of a SIMD are many -- {3,2,2,1,1,0} for six -- instead of modulo 4: 1-3 % slower at every size,
				// profiles/prio_rotation_probe_r05.txt; the code is in the history at 7193df8)
				switch ((r + (int)dround) & 3) {
				case 0: __builtin_amdgcn_s_setprio(0); break;
				case 1: __builtin_amdgcn_s_setprio(1); break;
				case 2: __builtin_amdgcn_s_setprio(2); break;
				default: __builtin_amdgcn_s_setprio(3); break;
				}
			}
			// The two words per row whose side neighbours sit in another vector (sites 0 / 31) are assembled on the scalar
			// unit from three source-colour words of row r0 + r - 1: A0, A1 of this wave's own 64 words, C from the
			// neighbouring wave column.  Plain launches: three scalar loads issued before the draw phase of row r0 + r.
			// Fused launches (the scalar cache may hold the row as it was two levels ago): A0, A1 come out of the lanes'
			// `ct` registers, C is a coherent vector load of one word.
			unsigned long long sA0 = 0, sA1 = 0, sC = 0;
			uint64_t vC = 0;
			const uint32_t grow_p = grow_w; // global row of row r0 + r - 1, this iteration's word phase (the draw phase below moves grow_w on)
			const bool late_here = wait_late && r == 1; // the first row is drawn: now the parents, then the unit's first two rows
			if (late_here && late_mode == 1) {          // (on their way during draw phase 1)
				wait_parents();
				if (!idle) {
					ld64_coh_issue<STREAM>(up0, rs + ((SUBL && r0_in_sl == 0) ? (ptrdiff_t)(slY - 1) * wpr : -(ptrdiff_t)wpr), lane * 8);
					ld64_coh_issue<STREAM>(ct0, rs, lane * 8);
				}
			}
			if (r > 0 && r <= nrows && !(late_here && late_mode == 2)) {
				const uint32_t grow = grow_p;
				const bool back = (color == 0) ? !(grow & 1u) : (grow & 1u);
				const uint64_t *qc = rs + (back ? u_cb : u_cf);
				if (FUSED) {
					ld64_coh_issue<STREAM>(vC, qc, 0);
				} else {
					const uint64_t *q0 = rs + (back ? word_of(0, 7, 3) : word_of(0, 0, 0));
					const uint64_t *q1 = rs + (back ? word_of(1, 7, 3) : word_of(1, 0, 0));
					asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %4, 0x0\n\ts_load_dwordx2 %2, %5, 0x0"
					             : "=&s"(sA0), "=&s"(sA1), "=&s"(sC) : "s"(q0), "s"(q1), "s"(qc) : "memory");
				}
			}
			TRC(3); // row prologue
			if (r < nrows) {
				// ---- draw phase, row r0 + r
				const PhiloxRow pr = philox_row_setup(tid_d, seed_lo_cy, k2y);
				{
					uint32_t inc = (grow_d & 15u) == 15u ? (uint32_t)p.gx * 256u - 240u : 16u, g1 = grow_d + 1u;
					if (p.total_rows && g1 == (uint32_t)p.total_rows) { // around the ring: row 0 follows the lattice's last row
						g1 = 0u;
						inc -= ((uint32_t)p.total_rows >> 4) * (uint32_t)p.gx * 256u;
					}
					grow_w = grow_d;
					grow_d = g1;
					tid_d += inc;
				}
				uint64_t *cur = slot + (r & 1) * 128;
				uint4 kc_next = blk_const[0];
				static_for<16>([&](auto B) {
					uint32_t o0, o1, o2, o3;
					// constants of the next block are fetched from LDS while this block's rounds run, and waited for before
					// this block's scalar stores go out (LDS and scalar memory share one counter)
					const uint4 kc = kc_next;
					if (B.value < 15) kc_next = blk_const[B.value + 1];
					philox_block_pre(pr, PhiloxBlockConst{kc.x, kc.y, kc.z}, seed_lo, seed_hi, o0, o1, o2, o3);
					if (B.value < 15) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kc_next.x), "+v"(kc_next.y), "+v"(kc_next.z) :: "memory");
					// (c3, c4) of one output = four consecutive SGPRs = one 16-byte scalar store: word p = 4B + q of the slot.
					// Fixed registers: inline asm cannot name halves of an SGPR tuple operand.  (Eight 8-byte stores from
					// compiler-allocated pairs: -6 %.)
					const uint64_t *dstp = cur + 8 * B.value;
					const uint32_t t3 = thr3, t4 = thr4; // (named here: asm operands alone do not make the lambda capture them)
					asm volatile("v_cmp_gt_u32_e64 " SG(0, 1) ", %0, %2\n\tv_cmp_gt_u32_e64 " SG(2, 3) ", %1, %2\n\t"
					             "v_cmp_gt_u32_e64 " SG(4, 5) ", %0, %3\n\tv_cmp_gt_u32_e64 " SG(6, 7) ", %1, %3\n\t"
					             "v_cmp_gt_u32_e64 " SG(8, 9) ", %0, %4\n\tv_cmp_gt_u32_e64 " SG(10, 11) ", %1, %4\n\t"
					             "v_cmp_gt_u32_e64 " SG(12, 13) ", %0, %5\n\tv_cmp_gt_u32_e64 " SG(14, 15) ", %1, %5\n\t"
					             "s_store_dwordx4 " SG(0, 3) ", %6, 0x0\n\ts_store_dwordx4 " SG(4, 7) ", %6, 0x10\n\t"
					             "s_store_dwordx4 " SG(8, 11) ", %6, 0x20\n\ts_store_dwordx4 " SG(12, 15) ", %6, 0x30"
					             :: "s"(t3), "s"(t4), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(dstp)
					             : "memory", BAL_CLOB16);
				});
			}
			if (late_here && late_mode == 2) { // ... behind draw phase 1 too: the parents, the first two rows and the row-end word of row 0
				wait_parents();
				if (!idle) {
					ld64_coh_issue<STREAM>(up0, rs + ((SUBL && r0_in_sl == 0) ? (ptrdiff_t)(slY - 1) * wpr : -(ptrdiff_t)wpr), lane * 8);
					ld64_coh_issue<STREAM>(ct0, rs, lane * 8);
					const bool back1 = (color == 0) ? !(grow_p & 1u) : (grow_p & 1u);
					ld64_coh_issue<STREAM>(vC, rs + (back1 ? u_cb : u_cf), 0);
				}
			}
			TRC(4); // draw phase
			if (FUSED && r == rmax && wi == 0) { // wave-uniform branch; read by the workgroup after this unit's last barrier
				if (lane == 0) ticket_sh[(round + 1) & 1] = draw_ticket();
			}
			TRC(5); // next ticket
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
			TRC(6); // barrier (scalar stores, write-back of the previous row, the slowest wave)
			if (r > 0 && r <= nrows) {
				// ---- word phase, row r0 + r - 1; its masks were written back during the draw phase above
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				const int lr = r0 + r - 1;
				const uint32_t grow = grow_p;
				const bool back = (color == 0) ? !(grow & 1u) : (grow & 1u); // readBack, optimized/main.cu:542
				// this lane's two accept masks: 16 bytes at slot + 16 lane, past the (non-coherent) vector L1
				const uint64_t *msk = slot + ((r - 1) & 1) * 128;
				u32x4 mk;
				asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(mk) : "v"(lane * 16), "s"(msk) : "memory");
				const bool sl_last = SUBL && seam == 1; // the row below is the period's first row (:422)
				uint64_t dw, me;
				if (FUSED) {
					ld64_coh_issue<STREAM>(dw, rs + (sl_last ? (ptrdiff_t)(1 - slY) * wpr : (ptrdiff_t)wpr), lane * 8);
					ld64_coh_issue<STREAM>(me, rd, lane * 8);
					// everything older than this phase's three loads: the word from the neighbouring wave column, in a unit's
					// first word phase also its first two rows
					asm volatile("s_waitcnt vmcnt(3)" : "+v"(vC), "+v"(up0), "+v"(ct0) :: "memory");
					if (r == 1) { // (a branch taken once per unit, not four selects per row: the empty asm keeps it one)
						asm volatile("" ::: "memory");
						up = up0;
						ct = ct0;
					}
					sA0 = readlane64(ct, back ? word_of(0, 7, 3) : word_of(0, 0, 0));
					sA1 = readlane64(ct, back ? word_of(1, 7, 3) : word_of(1, 0, 0));
					sC = readlane64(vC, 0);
				} else {
					dw = ld_word<false>(rs + (sl_last ? (ptrdiff_t)(1 - slY) * wpr : (ptrdiff_t)wpr) + lane); // scalar base + lane offset
					me = ld_word<false>(rd + lane);
					asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sA0), "+s"(sA1), "+s"(sC) :: "memory");
				}
				uint64_t w0, w1; // side words of lanes (0,0,0), (1,0,0) [back] / (0,7,3), (1,7,3) [forward]
				if (back) {
					w0 = ((sA0 << 1) & ~LANE0) | ((sA1 << 1) & u_b1) | (inwave ? 0ull : ((sC >> end_src) & 1ull));
					if (SUBL) w0 |= (sA1 >> wsh) & u_bw;
					w1 = ((sA1 << 1) & ~LANE0) | ((sA0 >> 15) & LANE0);
				} else {
					w0 = ((sA0 >> 1) & ~LANE15) | ((sA1 << 15) & LANE15);
					w1 = ((sA1 >> 1) & ~LANE15) | ((sA0 >> 1) & u_f1) | (inwave ? 0ull : ((sC & 1ull) << end_here));
					if (SUBL) w1 |= (sA0 << wsh) & u_fw;
				}
				const uint64_t A = bperm64(back ? backA : fwdA, ct);
				uint32_t sdl = (uint32_t)A, sdh = (uint32_t)(A >> 32);
				// (v_writelane takes its lane from an immediate: one SGPR operand per instruction on gfx9; the readfirstlanes
				// are free -- the words are wave-uniform by construction -- and keep the "s" constraints satisfiable when the
				// compiler's uniformity analysis gives up on the surrounding ticket loop)
				const uint32_t w0l = __builtin_amdgcn_readfirstlane((uint32_t)w0), w0h = __builtin_amdgcn_readfirstlane((uint32_t)(w0 >> 32));
				const uint32_t w1l = __builtin_amdgcn_readfirstlane((uint32_t)w1), w1h = __builtin_amdgcn_readfirstlane((uint32_t)(w1 >> 32));
				if (back) {
					asm("v_writelane_b32 %0, %2, 0\n\tv_writelane_b32 %1, %3, 0" : "+v"(sdl), "+v"(sdh) : "s"(w0l), "s"(w0h));
					asm("v_writelane_b32 %0, %2, 32\n\tv_writelane_b32 %1, %3, 32" : "+v"(sdl), "+v"(sdh) : "s"(w1l), "s"(w1h));
				} else {
					asm("v_writelane_b32 %0, %2, 31\n\tv_writelane_b32 %1, %3, 31" : "+v"(sdl), "+v"(sdh) : "s"(w0l), "s"(w0h));
					asm("v_writelane_b32 %0, %2, 63\n\tv_writelane_b32 %1, %3, 63" : "+v"(sdl), "+v"(sdh) : "s"(w1l), "s"(w1h));
				}
				uint64_t sd = ((uint64_t)sdh << 32) | sdl;
				TRC(7); // word phase up to the wait for its loads
				if (FUSED) asm volatile("s_waitcnt vmcnt(0)" : "+v"(mk), "+v"(dw), "+v"(me) :: "memory");
				else asm volatile("s_waitcnt vmcnt(0)" : "+v"(mk) :: "memory");
				TRC(10); // wait for masks and words
				const uint64_t c3 = ((uint64_t)mk.y << 32) | mk.x, c4 = ((uint64_t)mk.w << 32) | mk.z;
				uint64_t nu = up, nc = ct, nd = dw;
				if (USEJ) { // a set coupling bit flips that neighbour's contribution (optimized/main.cu:575-618)
					const uint64_t jr = rj[lane], jl = rj[64 + lane], jd = rj[128 + lane], ju = rj[192 + lane];
					nu ^= ju; nd ^= jd;
					nc ^= back ? jr : jl; // the same-index word holds the right neighbours when `back`, the side word the left
					sd ^= back ? jl : jr;
					rj += 4 * wpr;
				}
				const uint64_t nw = me ^ (flips64(me, nu, nc, nd, sd, c3, c4) & live);
				if (COUNT && (unsigned)lr < (unsigned)p.Y) { // (dead lanes are zero in memory; a ring slab's ghost rows are its neighbours' to count)
					cnt_up += (uint32_t)__popcll(nw);
					if (eq_unit) cnt_eq += (uint32_t)(__popcll(~(nw ^ up) & live) + __popcll(~(nw ^ ct) & live) + __popcll(~(nw ^ dw) & live) + __popcll(~(nw ^ sd) & live));
				}
				if (FUSED) {
					st64_coh_issue<STREAM>(rd, lane * 8, nw);
					if (p.wrap) { // the halo rows that mirror this colour's edge rows
						if (lr == 0) st64_coh_issue<STREAM>(rd + mir0, lane * 8, nw);
						if (lr == p.Y - 1) st64_coh_issue<STREAM>(rd + mirL, lane * 8, nw);
					}
				} else {
					rd[lane] = nw;
					if (p.wrap) {
						if (lr == 0) rd[mir0 + lane] = nw;
						if (lr == p.Y - 1) rd[mirL + lane] = nw;
					}
				}
				rs += wpr;
				rd += wpr;
				if (sl_last) { // the next row opens a new period: the register window does not slide across the seam
					seam = slY;
					if (r < nrows) { up = (rs + (ptrdiff_t)(slY - 1) * wpr)[lane]; ct = rs[lane]; }
				} else {
					up = ct;
					ct = dw;
					--seam;
				}
			}
			TRC(11); // flips, stores
			if (wb_wave && r < rmax) asm volatile("s_dcache_wb" ::: "memory");
		}
		if (FUSED) tkv = ticket_sh[(round + 1) & 1]; // (on its way while the stores drain)
		if (FUSED && !absent) {
			// publish: this wave's stores were written through (sc1); once they have left the wave the strip's counter
			// may move (every storing wave drains its own stores and signals its own unit)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			TRC(13); // store drain
#if defined(ISING_FUSED_TRACE) && defined(ISING_FUSED_TRACE_COUNTS)
			if (wi == 0 && lane == 0) { // unit durations with and without the wait, in units of 1/8 of the mean (h[10] accumulates the work time)
				const long long te = clock64();
				hist_sh[0][min(15ll, (te - t_unit1) / (16384ll * p.H / 8))]++;
				hist_sh[1][min(15ll, (te - t_unit0) / (16384ll * p.H / 8))]++;
				if ((te - t_unit1) / (16384ll * p.H / 8) >= 12) tr[10 + min(3u, dround)] += 1; // long units by dispatch round (3: fourth and later)
			}
#endif
			if (COUNT) { // level 2j and 2j + 1 are the black and the white half of the launch's sweep j: both add to its measurement
				if (meas_idx >= 0) {
					const unsigned long long tot = wave_sum((unsigned long long)cnt_up);
					const int meas = p.cnt_slot0 + meas_idx;
					const size_t planes = p.cnt_bonds ? 3 : 2, per_plane = (size_t)p.nwg * (NT / 64);
					if (lane == 0) p.cnt_acc[((size_t)meas * planes + (size_t)(level & 1)) * per_plane + (size_t)wave] = (uint32_t)tot;
					if (eq_unit) {
						const unsigned long long eq = wave_sum((unsigned long long)cnt_eq);
						if (lane == 0) p.cnt_acc[((size_t)meas * planes + 2) * per_plane + (size_t)wave] = (uint32_t)eq;
					}
				}
			}
			if (lane == 0) __hip_atomic_fetch_add(p.done + (BATCH ? rep * p.done_stride : 0) + sidx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			// last level: the rows the next exchange sends are final and the ghost rows no longer read
			if (edge_unit && (elev == elast || level == p.nlevels - 1) && lane == 0) __hip_atomic_fetch_add(p.edge_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	clock_mark(1);
#if defined(ISING_FUSED_TRACE)
	if (FUSED) {
		__syncthreads();
		if (threadIdx.x == 0) tr[14] = (unsigned long long)(clock64() - tstart);
		__syncthreads();
		if (threadIdx.x < 16) atomicAdd(&g_trace[threadIdx.x], tr[threadIdx.x]);
#if defined(ISING_FUSED_TRACE_COUNTS)
		if (threadIdx.x < 32) atomicAdd(&g_hist[threadIdx.x], (unsigned long long)hist_sh[threadIdx.x >> 4][threadIdx.x & 15]);
#endif
	}
#endif
}

// ---- Split launches (round 5): the fused launch's work in two kinds of units.  What a level of few tickets costs a fused launch is not the wait for late
// parents (units draw before they wait) but the height of its strips: a unit's fixed cost -- ticket, decode, counters, the extra loop iteration -- is ~9 % more
// vector instructions per site at H = 4 than at H = 16 (16384^2: 778 against 714 per row of a wave), and taller strips need as many tickets a level as
// workgroups run, because a unit cannot start its word phases before its parents have finished theirs.  But 85 % of a unit -- the draws -- needs nothing from
// the lattice.  So here a DRAW unit (wave column x strip of H rows x level) only draws: 16 Philox blocks a row, the compares' lane masks into a ring of mask
// slots in memory, no wait, no barrier, one scalar-cache write-back per unit; and a WORD unit, handed out in the same level-major order through a second
// ticket counter, waits for its masks and its parents and runs the H word phases.  Workgroups alternate: one draw unit, one word unit -- after a lead of
// `sp_lead` draw units each --, so at any time about a fifth of them hold word tickets: a level needs a quarter of the grid in tickets instead of
// all of it, strips are 16 rows tall from 2^26 spins up and six workgroups per CU run at every size.
// The accept masks go from one workgroup to another through the XCD's L2 (scalar stores are written back there, not through), so tickets come in eight
// classes, one per XCD: a workgroup serves the class of the XCD it RUNS on (HW_REG_XCC_ID, not blockIdx: correctness does not rest on the dispatcher's
// placement), class x owns the units 8 k + x of every level and a ring of sp_ring slots of its own.  Deadlock: word ticket w is only handed out behind draw
// ticket w (every workgroup draws first), both go out in order within a class, a draw unit waits for nothing but its slot -- freed by word ticket d - sp_ring,
// which was handed out long ago because the workgroups of a class hold at most (sp_lead + 1) draw tickets more than word tickets each -- and a word unit
// for lower tickets only: the lowest unfinished ticket of every class is always held by a running workgroup.  (A class whose XCD runs no workgroup of the
// launch would never be served: the polls' bound turns that into ISING_E_STATE like any other launch that gives up.)
// Ring slabs (ghost rows, trapezoid, the exchange's edge units) and levels whose tickets are not a multiple of eight run here as well: a class's last ticket
// of a level may name a unit that does not exist (absent: it keeps the order of the slots and counts for nothing).
template <bool COUNT>
__global__ void __launch_bounds__(BAL_THREADS) BAL_SGPR_ATTR ballot_split_k(const UpdateParams p) {
	const int lane = threadIdx.x & 63;
	const int tx = threadIdx.x & (GROUP - 1), g = (threadIdx.x >> 4) & 3;
	const int wi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int nwc = (p.gx + 3) >> 2;
	const int gxp = nwc << 2;
	const int wpr = nwc * 64;
	const int j = lane >> 5, m = (lane >> 2) & 7, q = lane & 3;
	int backA, fwdA; // lane holding the word with site s-1 / s+1 of the same vectors (ballot_update_k)
	if (q == 2) backA = word_of(j, m, 0);
	else if (q == 3) backA = word_of(j, m, 1);
	else if (q == 0) backA = m ? word_of(j, m - 1, 2) : word_of(j, 7, 3);
	else backA = m ? word_of(j, m - 1, 3) : word_of(j, 7, 2);
	if (q == 0) fwdA = word_of(j, m, 2);
	else if (q == 1) fwdA = word_of(j, m, 3);
	else if (q == 2) fwdA = m < 7 ? word_of(j, m + 1, 0) : word_of(j, 0, 1);
	else fwdA = m < 7 ? word_of(j, m + 1, 1) : word_of(j, 0, 0);
	const ptrdiff_t mir0 = (ptrdiff_t)(p.mir0_bytes / 8), mirL = (ptrdiff_t)(p.mirL_bytes / 8);

	__shared__ uint4 blk_const_all[BAL_THREADS / 64][16];
	__shared__ unsigned long long ticket_sh[2];
	uint4 *blk_const = blk_const_all[wi];

	uint32_t xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	xcc &= 7u;
	const int T8 = (p.nwg + 7) >> 3; // tickets per level and class (the last one of a level may name a unit beyond the level's last: absent)
	const unsigned long long total = (unsigned long long)p.nlevels * (unsigned long long)T8;
	unsigned long long *const dtp = p.sp_ctr + 16 * xcc, *const wtp = dtp + 8; // this class's draw / word ticket counters (zero when the launch starts)
	const int ring_sh = p.sp_ring_sh; // slots per class: 2^ring_sh
	uint32_t *const flags = p.sp_flags + ((size_t)xcc << (ring_sh + 1)); // per slot {waves that have drawn, waves that have used} -- counts, zero when the launch starts
	const size_t slot_words = (size_t)4 * (size_t)p.H * 128;               // a slot: 4 waves x H rows x (64 x {c3, c4})
	uint64_t *const masks = p.sp_masks + (((size_t)xcc << ring_sh) * 4 + (size_t)wi) * (size_t)p.H * 128;
	const unsigned dround = uni((int)(blockIdx.x / (unsigned)p.cus));
	const int nstr = (p.row_hi[0] - p.row_lo[0] + p.H - 1) / p.H;
	const uint32_t seed_lo = p.seed_lo, seed_hi = p.seed_hi;
	const uint32_t k2y = seed_hi + 2u * PHILOX_W1;
	const int nwc_sh = (nwc & (nwc - 1)) == 0 ? __builtin_ctz((unsigned)nwc) + 2 : -1; // gxp = 4 nwc as a shift where it is one
	// A class serves sp_cap workgroups at most -- its ring holds (sp_lead + 1) slots for each of them, which is what keeps draw units from ever waiting for a
	// word ticket nobody holds --: whoever registers beyond that (a dispatcher that put more of the grid on one XCD than an eighth and a margin) leaves.
	if (threadIdx.x == 0) ticket_sh[0] = atomicAdd(dtp + 4, 1ull);
	__syncthreads();
	if (ticket_sh[0] >= (unsigned long long)p.sp_cap) return;
	__syncthreads();
	// (ising_kernel_clock: as in ballot_update_k)
	auto clock_mark = [&](int which) {
		if (blockIdx.x >= 8u || wi != 0 || p.clk_out == nullptr) return;
		const unsigned long long cyc = __builtin_readcyclecounter(), ref = __builtin_amdgcn_s_memrealtime();
		if (lane == 0) {
			unsigned long long *o = p.clk_out + 4 * blockIdx.x + 2 * which;
			o[0] = cyc;
			o[1] = ref;
		}
	};
	clock_mark(0);

	// row of the whole lattice behind slab row r: ghost rows of a ring slab are rows of its neighbours, around the ring (ballot_update_k)
	auto global_row = [&](int r) -> uint32_t {
		int gr = (int)p.row_base + r;
		if (p.total_rows) gr = gr < 0 ? gr + p.total_rows : (gr >= p.total_rows ? gr - p.total_rows : gr);
		return (uint32_t)gr;
	};
	// a ticket of this class -> its unit of the level: (strip, wave column) of this wave, rows
	struct Unit { int level, elev, sidx, wc, bx0, r0, nrows; bool absent, edge, elast; uint32_t color, it; };
	auto decode = [&](int k, int level) -> Unit {
		Unit u;
		u.level = level;
		const int wgi = 8 * k + (int)xcc;
		const int wave = uni(wgi * (BAL_THREADS / 64) + wi);
		const int unit0 = wave * 4;
		u.absent = unit0 >= p.nreal0;
		const int uu = u.absent ? 0 : unit0;
		const int pos = nwc_sh >= 0 ? (uu >> nwc_sh) : uni(uu / gxp);
		u.bx0 = uu - pos * gxp;
		u.wc = u.bx0 >> 2;
		u.sidx = uni((pos & 1) ? nstr - 1 - (pos >> 1) : (pos >> 1)); // strips from both ends of the slab inwards, as in the fused launches
		u.r0 = p.row_lo[0] + u.sidx * p.H;
		const int nrows0 = u.absent ? 0 : min(p.H, p.row_hi[0] - u.r0);
		// ring slab with ghost rows: at this level only rows within `keep` of the slab's own can still reach one of them (trapezoid, ballot_update_k)
		// (several exchange epochs per launch, UpdateParams.epoch_sh -- round 6, as in ballot_update_k: the trapezoid starts over with every epoch)
		u.elev = uni(p.epoch_sh ? level & ((1 << p.epoch_sh) - 1) : level);
		const int elast = uni(p.epoch_sh ? (1 << p.epoch_sh) - 1 : p.nlevels - 1);
		u.elast = u.elev == elast || level == p.nlevels - 1;
		const int keep = uni(elast - u.elev);
		const bool skip = p.trapezoid != 0 && !u.absent && (u.r0 + nrows0 <= -keep || u.r0 >= p.Y + keep);
		u.nrows = skip ? 0 : nrows0;
		// ... and, exchange overlapped with the launches: a unit that touches rows the exchange reads or writes
		u.edge = p.edge_go != nullptr && !u.absent && (u.r0 < p.edge_lo || u.r0 + nrows0 > p.edge_hi);
		u.color = uni((p.color + (uint32_t)level) & 1u);
		u.it = uni(p.it + ((p.color + (uint32_t)level) >> 1));
		return u;
	};
	// counters that never come (see ballot_update_k): the unit that has polled `abort_polls` times raises the flag, every waiter looks at it every 64th poll
	auto give_up = [&](uint32_t npoll) -> bool {
		uint64_t fp;
		uint32_t bound;
		asm volatile("s_load_dwordx2 %0, %2, %3\n\ts_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(fp), "=&s"(bound)
		             : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(offsetof(UpdateParams, abort_flag)), "n"(offsetof(UpdateParams, abort_polls)) : "memory");
		uint32_t *flag = reinterpret_cast<uint32_t *>(fp);
		if (flag == nullptr) return false;
		if (npoll >= bound && lane == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		return npoll >= bound || uni(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != 0u;
	};

#if defined(ISING_FUSED_TRACE) // measurement build: wave 0 of every workgroup clocks its roles (ballot_trace_dump)
	__shared__ unsigned long long trs[16];
	if (threadIdx.x < 16) trs[threadIdx.x] = 0;
	__syncthreads();
	long long ts_last = clock64();
	const long long ts_start = ts_last;
#define STRC(i) do { if (wi == 0) { const long long t_ = clock64(); if (lane == 0) trs[i] += (unsigned long long)(t_ - ts_last); ts_last = t_; } } while (0)
#define STRN(i, n) do { if (wi == 0 && lane == 0) trs[i] += (unsigned long long)(n); } while (0)
#else
#define STRC(i) do {} while (0)
#define STRN(i, n) do {} while (0)
#endif
	int lvl_d = 0, lvl_w = 0;
	unsigned long long base_d = 0, base_w = 0;
	int lead = uni(p.sp_lead);
	if (ISING_FUSED_STAGGER > 0)
		for (unsigned i = 0; i < dround % 6u; ++i) __builtin_amdgcn_s_sleep(ISING_FUSED_STAGGER);
	for (;;) {
		// ================================================================ a draw unit
		if (threadIdx.x == 0) ticket_sh[0] = atomicAdd(dtp, 1ull);
		__syncthreads();
		const unsigned long long tkd_v = ticket_sh[0];
		const unsigned long long tkd = ((unsigned long long)uni((uint32_t)(tkd_v >> 32)) << 32) | uni((uint32_t)tkd_v);
		STRC(0); // draw ticket
		if (tkd < total) {
			STRN(8, 1);
			while (tkd >= base_d + (unsigned)T8) { base_d += (unsigned)T8; ++lvl_d; }
			const Unit u = decode((int)(tkd - base_d), lvl_d);
			const uint32_t slot = (uint32_t)tkd & ((1u << ring_sh) - 1u), use = (uint32_t)(tkd >> ring_sh);
			uint32_t *const fl = flags + 2 * slot;
			const uint32_t cx_base = 16u * (2u * u.it + u.color);
			const uint32_t seed_lo_cy = seed_lo ^ (uint32_t)((2ull * u.it + u.color) >> 28); // see dense_update_k
			if (lane < 16) {
				const PhiloxBlockConst kc = philox_block_const(cx_base + (uint32_t)lane, seed_lo, seed_hi);
				blk_const[lane] = make_uint4(kc.s0, kc.s1, kc.s2, 0u);
			}
			if (use > 0) { // the slot's previous masks must have been used: four waves each time it served
				const uint32_t need = 4u * use;
				uint32_t seen = need, npoll = 0;
				for (;;) {
					if (lane == 0) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(seen) : "v"(fl + 1) : "memory");
					if (__all((int32_t)(seen - need) >= 0)) break;
					if ((++npoll & 63u) == 0u && give_up(npoll)) break;
					STRN(10, 1);
					__builtin_amdgcn_s_sleep(ISING_POLL_SLEEP);
				}
			}
			STRC(1); // draw unit: decode, slot
			__builtin_amdgcn_wave_barrier();
			__threadfence_block();
			const int bx = u.bx0 + g;
			uint32_t grow_d = global_row(u.r0);
			uint32_t tid_d = ((grow_d >> 4) * (uint32_t)p.gx + (uint32_t)bx) * 256u + (grow_d & 15u) * 16u + (uint32_t)tx;
			const uint64_t *cur = masks + (size_t)slot * slot_words;
			const uint32_t thr3 = p.n3, thr4 = p.n4;
			for (int r = 0; r < u.nrows; ++r) {
				switch ((r + (int)dround) & 3) { // rotating priorities: the waves of a SIMD take turns at the front (ballot_update_k)
				case 0: __builtin_amdgcn_s_setprio(0); break;
				case 1: __builtin_amdgcn_s_setprio(1); break;
				case 2: __builtin_amdgcn_s_setprio(2); break;
				default: __builtin_amdgcn_s_setprio(3); break;
				}
				const PhiloxRow pr = philox_row_setup(tid_d, seed_lo_cy, k2y);
				{ // the next row's stream id: 16 on, a new block row of 16 every 16 rows, around the ring behind the lattice's last row (ballot_update_k)
					uint32_t inc = (grow_d & 15u) == 15u ? (uint32_t)p.gx * 256u - 240u : 16u, g1 = grow_d + 1u;
					if (p.total_rows && g1 == (uint32_t)p.total_rows) {
						g1 = 0u;
						inc -= ((uint32_t)p.total_rows >> 4) * (uint32_t)p.gx * 256u;
					}
					grow_d = g1;
					tid_d += inc;
				}
				uint4 kc_next = blk_const[0];
				static_for<16>([&](auto B) {
					uint32_t o0, o1, o2, o3;
					const uint4 kc = kc_next;
					if (B.value < 15) kc_next = blk_const[B.value + 1];
					philox_block_pre(pr, PhiloxBlockConst{kc.x, kc.y, kc.z}, seed_lo, seed_hi, o0, o1, o2, o3);
					if (B.value < 15) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kc_next.x), "+v"(kc_next.y), "+v"(kc_next.z) :: "memory");
					const uint64_t *dstp = cur + 8 * B.value;
					const uint32_t t3 = thr3, t4 = thr4;
					asm volatile("v_cmp_gt_u32_e64 " SG(0, 1) ", %0, %2\n\tv_cmp_gt_u32_e64 " SG(2, 3) ", %1, %2\n\t"
					             "v_cmp_gt_u32_e64 " SG(4, 5) ", %0, %3\n\tv_cmp_gt_u32_e64 " SG(6, 7) ", %1, %3\n\t"
					             "v_cmp_gt_u32_e64 " SG(8, 9) ", %0, %4\n\tv_cmp_gt_u32_e64 " SG(10, 11) ", %1, %4\n\t"
					             "v_cmp_gt_u32_e64 " SG(12, 13) ", %0, %5\n\tv_cmp_gt_u32_e64 " SG(14, 15) ", %1, %5\n\t"
					             "s_store_dwordx4 " SG(0, 3) ", %6, 0x0\n\ts_store_dwordx4 " SG(4, 7) ", %6, 0x10\n\t"
					             "s_store_dwordx4 " SG(8, 11) ", %6, 0x20\n\ts_store_dwordx4 " SG(12, 15) ", %6, 0x30"
					             :: "s"(t3), "s"(t4), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(dstp)
					             : "memory", BAL_CLOB16);
				});
				cur += 128;
			}
			// the unit's masks into the XCD's L2, then the slot's count: the word unit of this ticket runs on this XCD
			STRC(2); // draw rows
			asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
			if (lane == 0) __hip_atomic_fetch_add(fl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			STRC(3); // write-back
		}
		if (lead > 0) { // the launch's first units: draws only, so that the word units find their masks long done
			--lead;
			__syncthreads(); // (ticket_sh[0] is rewritten at the top)
			continue;
		}
		// ================================================================ a word unit
		if (threadIdx.x == 0) ticket_sh[1] = atomicAdd(wtp, 1ull);
		__syncthreads();
		const unsigned long long tkw_v = ticket_sh[1];
		const unsigned long long tkw = ((unsigned long long)uni((uint32_t)(tkw_v >> 32)) << 32) | uni((uint32_t)tkw_v);
		STRC(4); // word ticket
		if (tkw >= total) break;
		STRN(9, 1);
		while (tkw >= base_w + (unsigned)T8) { base_w += (unsigned)T8; ++lvl_w; }
		const Unit u = decode((int)(tkw - base_w), lvl_w);
		const int level = u.level, sidx = u.sidx, wc = u.wc, bx0 = u.bx0, r0 = u.r0, nrows = u.nrows;
		const uint32_t color = u.color;
		const uint32_t slot = (uint32_t)tkw & ((1u << ring_sh) - 1u), use = (uint32_t)(tkw >> ring_sh);
		uint32_t *const fl = flags + 2 * slot;
		__builtin_amdgcn_s_setprio(3); // word units are the dependency chain: short, and first in line
		{ // wait: lanes 0..2 for strips s-1, s, s+1 at level - 1 (`nwc` wave columns each per level), lane 3 for the four waves that drew this ticket's masks
			int sd = sidx + (lane == 0 ? -1 : (lane == 1 ? 0 : 1));
			sd = sd < 0 ? sd + nstr : (sd >= nstr ? sd - nstr : sd);
			const bool parents = level > 0 && !u.absent && lane < 3;
			const uint32_t *dp = parents ? p.done + sd : fl;
			const uint32_t need = parents ? p.done_base + (uint32_t)level * (uint32_t)nwc : 4u * (use + 1u);
			uint32_t seen = need, npoll = 0;
			for (;;) {
				if (lane < 4) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(seen) : "v"(dp) : "memory");
				if (__all((int32_t)(seen - need) >= 0)) break;
				if ((++npoll & 63u) == 0u && give_up(npoll)) break;
				STRN(11, 1);
				STRN(12, npoll == 0);
#if defined(ISING_FUSED_TRACE)
				{ const bool late_masks = __builtin_amdgcn_readlane((int)(seen - need), 3) < 0; STRN(13, late_masks); }
#endif
				__builtin_amdgcn_s_sleep(ISING_POLL_SLEEP);
			}
		}
		// (an epoch's second level is the last that reads words the exchange wrote, possibly on an XCD none of the epoch's first edge units ran on: ballot_update_k)
		if (p.epoch_sh && u.edge && u.elev == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
		if (u.edge && u.elev == 0) {
			// the exchange that follows the previous launch (the previous epoch of this one) has read this slab's first / last rows and filled its ghost rows once
			// the comm stream has moved the counter (ballot_update_k: usually long ago; the bound is 16 times a unit's for its parents)
			const uint32_t go_need = uni(p.edge_go_need + (p.epoch_sh ? (uint32_t)level >> p.epoch_sh : 0u));
			uint32_t got = go_need, ngo = 0;
			for (;;) {
				if (lane == 0) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(got) : "v"(p.edge_go) : "memory");
				if (__all((int32_t)(got - go_need) >= 0)) break;
				if ((++ngo & 63u) == 0u && give_up(ngo >> 4)) break;
				__builtin_amdgcn_s_sleep(127);
			}
			// rows written while this launch ran by another kernel, a copy engine or a peer device: system scope (the lattice loads behind it bypass the vector L1 anyway)
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
		}
		STRC(5); // word unit: decode, wait
		[[maybe_unused]] uint32_t cnt_up = 0, cnt_eq = 0;
		[[maybe_unused]] int meas_idx = -1;
		if (COUNT && p.cnt_every > 0 && (level >> 1) >= p.cnt_first) {
			const uint32_t d = (uint32_t)((level >> 1) - p.cnt_first), e = (uint32_t)p.cnt_every;
			uint32_t qq = e == 1u ? d : __umulhi(d, p.cnt_magic);
			if (qq * e > d) --qq;
			if (qq * e == d) meas_idx = uni((int)qq);
		}
		[[maybe_unused]] const bool eq_unit = COUNT && p.cnt_bonds != 0 && (level & 1) != 0 && meas_idx >= 0;
		if (nrows > 0) {
			uint64_t *const lat0 = p.lat[0], *const lat1 = p.lat[1];
			const uint64_t *const rs0 = (color ? lat0 : lat1) + ((ptrdiff_t)r0 * wpr + wc * 64);
			uint64_t *const rd0 = (color ? lat1 : lat0) + ((ptrdiff_t)r0 * wpr + wc * 64);
			uint64_t first = 0, last = 0; // bit 16g (16g + 15): group g opens (closes) the row
#pragma unroll
			for (int gg = 0; gg < 4; ++gg) {
				if (bx0 + gg == 0) first |= 1ull << (16 * gg);
				if (bx0 + gg == p.gx - 1) last |= 1ull << (16 * gg + 15);
			}
			const uint64_t u_b1 = LANE0 & ~1ull & ~first, u_f1 = LANE15 & ~(1ull << 63) & ~last;
			const int alive = min(4, p.gx - bx0);
			const uint64_t live = alive >= 4 ? ~0ull : ((1ull << (16 * alive)) - 1ull);
			const int end_here = 16 * alive - 1;
			const int src_b = wc ? wc - 1 : nwc - 1;
			const int end_src = 16 * min(4, p.gx - 4 * src_b) - 1;
			const int u_cb = (src_b - wc) * 64 + word_of(1, 7, 3);
			const int u_cf = ((wc == nwc - 1 ? 0 : wc + 1) - wc) * 64 + word_of(0, 0, 0);
			const uint64_t *const msk0 = masks + (size_t)slot * slot_words;
			const uint32_t grow0 = global_row(r0);
			// A row's flips need the word of the neighbouring wave column, the lane's two masks, the word below and the word itself: four loads, one wait, the flips,
			// one store -- a memory round trip per row, next to workgroups that draw.  (Round 5 also kept 1 / 2 / 4 rows of loads in flight across the stores of
			// the rows before -- straight-line code, hand-counted vmcnt(n) per row --: the word rows' share of a workgroup's time halved, 15.7 -> 8.0 %, and the
			// launches ran 1-2 % SLOWER at every shape on one box (16384^2, H = 16: 3391 -> 3343 / 3334 at one / four rows in flight; 65536 x 8192: 3392 -> 3345 / 3312;
			// profiles/split_depth_probe_r05.txt): the vector ALU is what binds, the chain of word units is not, and the pipeline's instructions are not free.)
			uint32_t grow = grow0;
			const uint64_t *rs = rs0, *msk = msk0;
			uint64_t *rd = rd0;
			uint64_t up, ct;
			ld64_coh_issue<false>(up, rs - (ptrdiff_t)wpr, lane * 8);
			ld64_coh_issue<false>(ct, rs, lane * 8);
			// (waited for HERE: values loaded by inline assembly must not be loop-carried before their wait -- the compiler may copy them at the loop's head)
			asm volatile("s_waitcnt vmcnt(0)" : "+v"(up), "+v"(ct) :: "memory");
			for (int r = 0; r < nrows; ++r) {
				const int lr = r0 + r;
				const bool back = (color == 0) ? !(grow & 1u) : (grow & 1u) != 0u; // readBack, optimized/main.cu:542
				uint64_t vC, dw, me;
				u32x4 mk;
				ld64_coh_issue<false>(vC, rs + (back ? u_cb : u_cf), 0);
				asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=&v"(mk) : "v"(lane * 16), "s"(msk) : "memory");
				ld64_coh_issue<false>(dw, rs + (ptrdiff_t)wpr, lane * 8);
				ld64_coh_issue<false>(me, rd, lane * 8);
				asm volatile("s_waitcnt vmcnt(0)" : "+v"(vC), "+v"(mk), "+v"(dw), "+v"(me) :: "memory");
				const unsigned long long sA0 = readlane64(ct, back ? word_of(0, 7, 3) : word_of(0, 0, 0));
				const unsigned long long sA1 = readlane64(ct, back ? word_of(1, 7, 3) : word_of(1, 0, 0));
				const unsigned long long sC = readlane64(vC, 0);
				uint64_t w0, w1; // side words of lanes (0,0,0), (1,0,0) [back] / (0,7,3), (1,7,3) [forward]
				if (back) {
					w0 = ((sA0 << 1) & ~LANE0) | ((sA1 << 1) & u_b1) | ((sC >> end_src) & 1ull);
					w1 = ((sA1 << 1) & ~LANE0) | ((sA0 >> 15) & LANE0);
				} else {
					w0 = ((sA0 >> 1) & ~LANE15) | ((sA1 << 15) & LANE15);
					w1 = ((sA1 >> 1) & ~LANE15) | ((sA0 >> 1) & u_f1) | ((sC & 1ull) << end_here);
				}
				const uint64_t A = bperm64(back ? backA : fwdA, ct);
				uint32_t sdl = (uint32_t)A, sdh = (uint32_t)(A >> 32);
				const uint32_t w0l = __builtin_amdgcn_readfirstlane((uint32_t)w0), w0h = __builtin_amdgcn_readfirstlane((uint32_t)(w0 >> 32));
				const uint32_t w1l = __builtin_amdgcn_readfirstlane((uint32_t)w1), w1h = __builtin_amdgcn_readfirstlane((uint32_t)(w1 >> 32));
				if (back) {
					asm("v_writelane_b32 %0, %2, 0\n\tv_writelane_b32 %1, %3, 0" : "+v"(sdl), "+v"(sdh) : "s"(w0l), "s"(w0h));
					asm("v_writelane_b32 %0, %2, 32\n\tv_writelane_b32 %1, %3, 32" : "+v"(sdl), "+v"(sdh) : "s"(w1l), "s"(w1h));
				} else {
					asm("v_writelane_b32 %0, %2, 31\n\tv_writelane_b32 %1, %3, 31" : "+v"(sdl), "+v"(sdh) : "s"(w0l), "s"(w0h));
					asm("v_writelane_b32 %0, %2, 63\n\tv_writelane_b32 %1, %3, 63" : "+v"(sdl), "+v"(sdh) : "s"(w1l), "s"(w1h));
				}
				const uint64_t sd = ((uint64_t)sdh << 32) | sdl;
				const uint64_t c3 = ((uint64_t)mk.y << 32) | mk.x, c4 = ((uint64_t)mk.w << 32) | mk.z;
				const uint64_t nw = me ^ (flips64(me, up, ct, dw, sd, c3, c4) & live);
				if (COUNT && (unsigned)lr < (unsigned)p.Y) { // (a ring slab's ghost rows are its neighbours' to count)
					cnt_up += (uint32_t)__popcll(nw);
					if (eq_unit) cnt_eq += (uint32_t)(__popcll(~(nw ^ up) & live) + __popcll(~(nw ^ ct) & live) + __popcll(~(nw ^ dw) & live) + __popcll(~(nw ^ sd) & live));
				}
				st64_coh_issue<false>(rd, lane * 8, nw);
				if (p.wrap) { // the halo rows that mirror this colour's edge rows
					if (lr == 0) st64_coh_issue<false>(rd + mir0, lane * 8, nw);
					if (lr == p.Y - 1) st64_coh_issue<false>(rd + mirL, lane * 8, nw);
				}
				rs += wpr;
				rd += wpr;
				msk += 128;
				up = ct;
				ct = dw;
				grow += 1u;
				if (p.total_rows && grow == (uint32_t)p.total_rows) grow = 0u; // around the ring: row 0 follows the lattice's last row
			}
		}
		STRC(6); // word rows
		// publish: the stores were written through (sc1); once they have left the wave the strip's counter may move, and the slot is free again
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		STRC(7); // drain
		__builtin_amdgcn_s_setprio(0);
		if (COUNT && meas_idx >= 0 && !u.absent) {
			const int wave = uni((8 * (int)(tkw - base_w) + (int)xcc) * (BAL_THREADS / 64) + wi);
			const unsigned long long tot = wave_sum((unsigned long long)cnt_up);
			const int meas = p.cnt_slot0 + meas_idx;
			const size_t planes = p.cnt_bonds ? 3 : 2, per_plane = (size_t)p.nwg * (BAL_THREADS / 64);
			if (lane == 0) p.cnt_acc[((size_t)meas * planes + (size_t)(level & 1)) * per_plane + (size_t)wave] = (uint32_t)tot;
			if (eq_unit) {
				const unsigned long long eq = wave_sum((unsigned long long)cnt_eq);
				if (lane == 0) p.cnt_acc[((size_t)meas * planes + 2) * per_plane + (size_t)wave] = (uint32_t)eq;
			}
		}
		if (lane == 0) {
			if (!u.absent) __hip_atomic_fetch_add(p.done + sidx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_fetch_add(fl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			// last level: the rows the next exchange sends are final and the ghost rows no longer read
			if (u.edge && u.elast) __hip_atomic_fetch_add(p.edge_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	clock_mark(1);
#if defined(ISING_FUSED_TRACE)
	__syncthreads();
	if (threadIdx.x == 0) trs[14] = (unsigned long long)(clock64() - ts_start);
	__syncthreads();
	if (threadIdx.x < 16) atomicAdd(&g_trace[threadIdx.x], trs[threadIdx.x]);
#endif
#undef STRC
#undef STRN
}

// ---- init (latticeInit_k, optimized/main.cu:92-151): one wave per (row, wave column).  A spin starts up where
// curand_uniform(x) < 0.5f, i.e. x < thr_half: the lane mask of that compare for output q of draw block B *is* word
// 4B + q of the row, so the lattice is written by the scalar unit alone -- 16 blocks x (4 v_cmp + two 16-byte scalar
// stores) per row and wave column, no vector store, no word phase.
__global__ void __launch_bounds__(THREADS) ballot_init_k(const InitParams p) {
	const int tx = threadIdx.x & (GROUP - 1), g = (threadIdx.x >> 4) & 3;
	const int nwc = (p.gx + 3) >> 2;
	const long long wave_ll = flat_block() * (THREADS / 64) + (threadIdx.x >> 6);
	if (wave_ll >= (long long)nwc * p.Y) return;
	const int wave = __builtin_amdgcn_readfirstlane((int)wave_ll);
	const int lr = wave / nwc, wc = wave - lr * nwc;
	const int alive = min(4, p.gx - 4 * wc); // column groups of this wave column that exist
	const unsigned long long live = alive >= 4 ? ~0ull : ((1ull << (16 * alive)) - 1ull);
	const uint32_t grow = p.row_base + (uint32_t)lr;
	const uint32_t tid = ((grow >> 4) * (uint32_t)p.gx + (uint32_t)(4 * wc + g)) * 256u + (grow & 15u) * 16u + (uint32_t)tx;
	const PhiloxRow pr = philox_row_setup(tid, p.seed_lo, p.seed_hi + 2u * PHILOX_W1);
	const uint32_t cx_base = 16u * p.color; // it = 0, optimized/main.cu:116
	const ptrdiff_t wpr = (ptrdiff_t)nwc * 64;
	uint64_t *row = p.dst + ((ptrdiff_t)lr * wpr + wc * 64);
	// single slab: rows 0 and Y - 1 are mirrored into the halo rows Y and -1 (periodic wrap)
	uint64_t *mirror = !p.wrap ? nullptr : (lr == 0 ? row + (ptrdiff_t)p.Y * wpr : (lr == p.Y - 1 ? row - (ptrdiff_t)p.Y * wpr : nullptr));
	static_for<16>([&](auto B) {
		uint32_t o0, o1, o2, o3;
		uint32_t cx = cx_base + (uint32_t)B.value;
		asm volatile("" : "+s"(cx));
		philox_block(pr, cx, p.seed_lo, p.seed_hi, o0, o1, o2, o3);
		const uint64_t *dstp = row + 4 * B.value;
		if (alive < 4) { // the partly dead last wave column: dead lanes stay zero in memory
			const uint64_t *dstm = mirror ? mirror + 4 * B.value : dstp;
			asm volatile("v_cmp_gt_u32_e64 " SG(0, 1) ", %0, %1\n\tv_cmp_gt_u32_e64 " SG(2, 3) ", %0, %2\n\t"
			             "v_cmp_gt_u32_e64 " SG(4, 5) ", %0, %3\n\tv_cmp_gt_u32_e64 " SG(6, 7) ", %0, %4\n\t"
			             "s_and_b64 " SG(0, 1) ", " SG(0, 1) ", %7\n\ts_and_b64 " SG(2, 3) ", " SG(2, 3) ", %7\n\t"
			             "s_and_b64 " SG(4, 5) ", " SG(4, 5) ", %7\n\ts_and_b64 " SG(6, 7) ", " SG(6, 7) ", %7\n\t"
			             "s_store_dwordx4 " SG(0, 3) ", %5, 0x0\n\ts_store_dwordx4 " SG(4, 7) ", %5, 0x10\n\t"
			             "s_store_dwordx4 " SG(0, 3) ", %6, 0x0\n\ts_store_dwordx4 " SG(4, 7) ", %6, 0x10"
			             :: "s"(p.thr_half), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(dstp), "s"(dstm), "s"(live)
			             : "memory", "scc", BAL_CLOB8);
		} else if (!mirror) {
			asm volatile("v_cmp_gt_u32_e64 " SG(0, 1) ", %0, %1\n\tv_cmp_gt_u32_e64 " SG(2, 3) ", %0, %2\n\t"
			             "v_cmp_gt_u32_e64 " SG(4, 5) ", %0, %3\n\tv_cmp_gt_u32_e64 " SG(6, 7) ", %0, %4\n\t"
			             "s_store_dwordx4 " SG(0, 3) ", %5, 0x0\n\ts_store_dwordx4 " SG(4, 7) ", %5, 0x10"
			             :: "s"(p.thr_half), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(dstp)
			             : "memory", BAL_CLOB8);
		} else {
			const uint64_t *dstm = mirror + 4 * B.value;
			asm volatile("v_cmp_gt_u32_e64 " SG(0, 1) ", %0, %1\n\tv_cmp_gt_u32_e64 " SG(2, 3) ", %0, %2\n\t"
			             "v_cmp_gt_u32_e64 " SG(4, 5) ", %0, %3\n\tv_cmp_gt_u32_e64 " SG(6, 7) ", %0, %4\n\t"
			             "s_store_dwordx4 " SG(0, 3) ", %5, 0x0\n\ts_store_dwordx4 " SG(4, 7) ", %5, 0x10\n\t"
			             "s_store_dwordx4 " SG(0, 3) ", %6, 0x0\n\ts_store_dwordx4 " SG(4, 7) ", %6, 0x10"
			             :: "s"(p.thr_half), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(dstp), "s"(dstm)
			             : "memory", BAL_CLOB8);
		}
	});
	asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// ---- observables on the ballot layout itself, for a batch of lattices in ONE launch (ising_batch_measure_enqueue): up spins
// (countSpins, optimized/main.cu:831-868) and A = number of (black site, white neighbour) pairs with equal spins
// (ising_bond_equal) of every lattice of the batch.  One wave per (lattice, strip of R rows, wave column); lane p owns word p
// of a row and meets its four neighbour words exactly as the update kernel's word phase does (above / same index / below
// from memory, the side word from another lane, the two row-end words put together from three wave-uniform words); then
// 4 x 64 bonds are one XOR + popcount each.  Rows -1 and Y of the white array must mirror rows Y-1 and 0 (single slabs).
constexpr int MEASURE_SLOTS = BALLOT_MEASURE_SLOTS;
__global__ void __launch_bounds__(THREADS) ballot_measure_k(const ReplicaParams *__restrict__ reps, int nrep, int gx, int Y, int R, unsigned long long *__restrict__ acc) {
	const int lane = threadIdx.x & 63;
	const int nwc = (gx + 3) >> 2;
	const ptrdiff_t wpr = (ptrdiff_t)nwc * 64;
	const int strips = (Y + R - 1) / R;
	const long long per_rep = (long long)strips * nwc;
	const long long wave_ll = flat_block() * (THREADS / 64) + (threadIdx.x >> 6);
	if (wave_ll >= per_rep * nrep) return; // (whole waves leave; nobody meets at a barrier)
	const int rep = uni((int)(wave_ll / per_rep));
	const int u = uni((int)(wave_ll - (long long)rep * per_rep));
	const int strip = u / nwc, wc = u - strip * nwc;
	const uint64_t *black = reps[rep].lat[0], *white = reps[rep].lat[1];
	// lane geometry of the word phase (ballot_update_k)
	const int j = lane >> 5, m = (lane >> 2) & 7, q = lane & 3;
	int backA, fwdA;
	if (q == 2) backA = word_of(j, m, 0);
	else if (q == 3) backA = word_of(j, m, 1);
	else if (q == 0) backA = m ? word_of(j, m - 1, 2) : word_of(j, 7, 3);
	else backA = m ? word_of(j, m - 1, 3) : word_of(j, 7, 2);
	if (q == 0) fwdA = word_of(j, m, 2);
	else if (q == 1) fwdA = word_of(j, m, 3);
	else if (q == 2) fwdA = m < 7 ? word_of(j, m + 1, 0) : word_of(j, 0, 1);
	else fwdA = m < 7 ? word_of(j, m + 1, 1) : word_of(j, 0, 0);
	const int bx0 = 4 * wc;
	uint64_t first = 0, last = 0;
#pragma unroll
	for (int gg = 0; gg < 4; ++gg) {
		if (bx0 + gg == 0) first |= 1ull << (16 * gg);
		if (bx0 + gg == gx - 1) last |= 1ull << (16 * gg + 15);
	}
	const uint64_t u_b1 = LANE0 & ~1ull & ~first, u_f1 = LANE15 & ~(1ull << 63) & ~last;
	const int alive = min(4, gx - bx0);
	const uint64_t live = alive >= 4 ? ~0ull : ((1ull << (16 * alive)) - 1ull);
	const int end_here = 16 * alive - 1;
	const int src_b = wc ? wc - 1 : nwc - 1;
	const int end_src = 16 * min(4, gx - 4 * src_b) - 1;
	const int u_cb = (src_b - wc) * 64 + word_of(1, 7, 3);
	const int u_cf = ((wc == nwc - 1 ? 0 : wc + 1) - wc) * 64 + word_of(0, 0, 0);

	const int r0 = strip * R, nrows = min(R, Y - r0);
	const uint64_t *rw = white + ((ptrdiff_t)r0 * wpr + wc * 64), *rb = black + ((ptrdiff_t)r0 * wpr + wc * 64);
	uint64_t up = rw[lane - wpr], ct = rw[lane];
	unsigned long long n_up = 0, n_eq = 0;
	for (int i = 0; i < nrows; ++i) {
		const bool back = !((r0 + i) & 1); // black sites of even rows have their side neighbour one white site back (readBack, optimized/main.cu:542)
		const uint64_t dw = rw[lane + wpr], me = rb[lane];
		const uint64_t vC = rw[back ? u_cb : u_cf]; // (one word, the same for every lane)
		const uint64_t sA0 = readlane64(ct, back ? word_of(0, 7, 3) : word_of(0, 0, 0));
		const uint64_t sA1 = readlane64(ct, back ? word_of(1, 7, 3) : word_of(1, 0, 0));
		const uint64_t sC = readlane64(vC, 0);
		uint64_t w0, w1;
		if (back) {
			w0 = ((sA0 << 1) & ~LANE0) | ((sA1 << 1) & u_b1) | ((sC >> end_src) & 1ull);
			w1 = ((sA1 << 1) & ~LANE0) | ((sA0 >> 15) & LANE0);
		} else {
			w0 = ((sA0 >> 1) & ~LANE15) | ((sA1 << 15) & LANE15);
			w1 = ((sA1 >> 1) & ~LANE15) | ((sA0 >> 1) & u_f1) | ((sC & 1ull) << end_here);
		}
		const uint64_t A = bperm64(back ? backA : fwdA, ct);
		uint32_t sdl = (uint32_t)A, sdh = (uint32_t)(A >> 32);
		const uint32_t w0l = __builtin_amdgcn_readfirstlane((uint32_t)w0), w0h = __builtin_amdgcn_readfirstlane((uint32_t)(w0 >> 32));
		const uint32_t w1l = __builtin_amdgcn_readfirstlane((uint32_t)w1), w1h = __builtin_amdgcn_readfirstlane((uint32_t)(w1 >> 32));
		if (back) {
			asm("v_writelane_b32 %0, %2, 0\n\tv_writelane_b32 %1, %3, 0" : "+v"(sdl), "+v"(sdh) : "s"(w0l), "s"(w0h));
			asm("v_writelane_b32 %0, %2, 32\n\tv_writelane_b32 %1, %3, 32" : "+v"(sdl), "+v"(sdh) : "s"(w1l), "s"(w1h));
		} else {
			asm("v_writelane_b32 %0, %2, 31\n\tv_writelane_b32 %1, %3, 31" : "+v"(sdl), "+v"(sdh) : "s"(w0l), "s"(w0h));
			asm("v_writelane_b32 %0, %2, 63\n\tv_writelane_b32 %1, %3, 63" : "+v"(sdl), "+v"(sdh) : "s"(w1l), "s"(w1h));
		}
		const uint64_t sd = ((uint64_t)sdh << 32) | sdl;
		n_eq += (unsigned)(__popcll(~(me ^ up) & live) + __popcll(~(me ^ ct) & live) + __popcll(~(me ^ dw) & live) + __popcll(~(me ^ sd) & live));
		n_up += (unsigned)(__popcll(me) + __popcll(ct));
		rw += wpr;
		rb += wpr;
		up = ct;
		ct = dw;
	}
	n_up = wave_sum(n_up);
	n_eq = wave_sum(n_eq);
	// MEASURE_SLOTS accumulator pairs per lattice, a line apart (the host adds them up): a thousand waves adding to one
	// address queue up at the L2 for longer than they spend counting
	if (lane == 0) {
		unsigned long long *a = acc + ((size_t)rep * MEASURE_SLOTS + (size_t)(u & (MEASURE_SLOTS - 1))) * 8;
		atomicAdd(a, n_up);
		atomicAdd(a + 1, n_eq);
	}
}

// ---- layout conversion, one wave per (row, wave column): 64 ballot words <-> 128 dense 32-bit words
__device__ __forceinline__ int word_of_site(int j, int s) {
	const int m = (s & 15) >> 1;
	const int q = ((s & 1) << 1) | (s >> 4); // s = 2m, 16+2m, 2m+1, 17+2m  ->  q = 0, 1, 2, 3
	return word_of(j, m, q);
}

// (ballot rows: nwc = ceil(gx / 4) wave columns of 64 words; dense rows: gx * 32 words -- the vectors of the dead column
// groups of a partly dead last wave column do not exist there)
__global__ void __launch_bounds__(THREADS) ballot_to_dense_k(const uint64_t *__restrict__ bal, uint32_t *__restrict__ dense, long long ngroups, int gx) {
	__shared__ uint64_t sh[THREADS / 64][64];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int nwc = (gx + 3) >> 2;
	for (long long grp = (long long)blockIdx.x * (THREADS / 64) + wv; grp < ngroups; grp += (long long)gridDim.x * (THREADS / 64)) {
		const long long row = grp / nwc;
		const int wc = (int)(grp - row * nwc), nvec = min(4, gx - 4 * wc) * 32;
		uint32_t *drow = dense + row * gx * 32 + wc * 128;
		sh[wv][lane] = bal[grp * 64 + lane];
		__builtin_amdgcn_wave_barrier();
		__threadfence_block();
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int v = lane + 64 * h; // vector within the wave column: 32 g + 16 j + tx
			const int l = ((v >> 5) << 4) | (v & 15), j = (v >> 4) & 1;
			uint32_t d = 0;
#pragma unroll
			for (int s = 0; s < 32; ++s) d |= (uint32_t)((sh[wv][word_of_site(j, s)] >> l) & 1ull) << s;
			if (v < nvec) drow[v] = d;
		}
		__builtin_amdgcn_wave_barrier();
		__threadfence_block();
	}
}

__global__ void __launch_bounds__(THREADS) dense_to_ballot_k(const uint32_t *__restrict__ dense, uint64_t *__restrict__ bal, long long ngroups, int gx) {
	__shared__ uint32_t sh[THREADS / 64][128];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int j = lane >> 5, m = (lane >> 2) & 7, q = lane & 3;
	const int s = (q & 1) * 16 + 2 * m + (q >> 1);
	const int nwc = (gx + 3) >> 2;
	for (long long grp = (long long)blockIdx.x * (THREADS / 64) + wv; grp < ngroups; grp += (long long)gridDim.x * (THREADS / 64)) {
		const long long row = grp / nwc;
		const int wc = (int)(grp - row * nwc), nvec = min(4, gx - 4 * wc) * 32;
		const uint32_t *drow = dense + row * gx * 32 + wc * 128;
		sh[wv][lane] = lane < nvec ? drow[lane] : 0u;
		sh[wv][lane + 64] = lane + 64 < nvec ? drow[lane + 64] : 0u;
		__builtin_amdgcn_wave_barrier();
		__threadfence_block();
		uint64_t w = 0;
#pragma unroll 8
		for (int l = 0; l < 64; ++l) {
			const int v = ((l >> 4) << 5) | (j << 4) | (l & 15);
			w |= (uint64_t)((sh[wv][v] >> s) & 1u) << l;
		}
		bal[grp * 64 + lane] = w;
		__builtin_amdgcn_wave_barrier();
		__threadfence_block();
	}
}

// ---- -J coupling arrays, in place, one wave per (row, wave column) = 128 vectors = 2 KiB.
// nibble form (32 nibbles <right, left, down, up> = bits 0..3 per 16-byte vector, as ham_init_*_k write them) ->
// four planes of 64 ballot-order words: [plane][word p], bit l of word p = coupling bit of the site that bit l of spin
// word p holds.
__global__ void __launch_bounds__(THREADS) ham_nibbles_to_ballot_k(uint64_t *__restrict__ ham, long long ngroups) {
	__shared__ uint32_t sh[THREADS / 64][128 * 4];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int j = lane >> 5, m = (lane >> 2) & 7, q = lane & 3;
	const int s = (q & 1) * 16 + 2 * m + (q >> 1);
	for (long long grp = (long long)blockIdx.x * (THREADS / 64) + wv; grp < ngroups; grp += (long long)gridDim.x * (THREADS / 64)) {
		uint4 *g4 = reinterpret_cast<uint4 *>(ham + grp * 256);
		const uint4 a = g4[lane], b = g4[lane + 64];
		uint32_t *d = sh[wv];
		d[4 * lane] = a.x; d[4 * lane + 1] = a.y; d[4 * lane + 2] = a.z; d[4 * lane + 3] = a.w;
		d[4 * (lane + 64)] = b.x; d[4 * (lane + 64) + 1] = b.y; d[4 * (lane + 64) + 2] = b.z; d[4 * (lane + 64) + 3] = b.w;
		__builtin_amdgcn_wave_barrier();
		__threadfence_block();
		uint64_t w[4] = {0, 0, 0, 0};
#pragma unroll 4
		for (int l = 0; l < 64; ++l) {
			const int v = ((l >> 4) << 5) | (j << 4) | (l & 15);
			const uint32_t nib = (d[4 * v + (s >> 3)] >> (4 * (s & 7))) & 0xFu;
#pragma unroll
			for (int pl = 0; pl < 4; ++pl) w[pl] |= (uint64_t)((nib >> pl) & 1u) << l;
		}
		__builtin_amdgcn_wave_barrier();
		__threadfence_block();
#pragma unroll
		for (int pl = 0; pl < 4; ++pl) ham[grp * 256 + 64 * pl + lane] = w[pl];
	}
}

// ballot planes -> the dense layout's per-vector planes (uint4 {right, left, down, up}, ham_planes_k's output)
__global__ void __launch_bounds__(THREADS) ham_ballot_to_planes_k(uint64_t *__restrict__ ham, long long ngroups) {
	__shared__ uint64_t sh[THREADS / 64][256];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (long long grp = (long long)blockIdx.x * (THREADS / 64) + wv; grp < ngroups; grp += (long long)gridDim.x * (THREADS / 64)) {
#pragma unroll
		for (int pl = 0; pl < 4; ++pl) sh[wv][64 * pl + lane] = ham[grp * 256 + 64 * pl + lane];
		__builtin_amdgcn_wave_barrier();
		__threadfence_block();
		uint4 out[2];
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int v = lane + 64 * h;
			const int l = ((v >> 5) << 4) | (v & 15), j = (v >> 4) & 1;
			uint32_t pw[4] = {0, 0, 0, 0};
#pragma unroll 4
			for (int s = 0; s < 32; ++s) {
				const int p = word_of_site(j, s);
#pragma unroll
				for (int pl = 0; pl < 4; ++pl) pw[pl] |= (uint32_t)((sh[wv][64 * pl + p] >> l) & 1ull) << s;
			}
			out[h] = make_uint4(pw[0], pw[1], pw[2], pw[3]);
		}
		__builtin_amdgcn_wave_barrier();
		__threadfence_block();
		uint4 *g4 = reinterpret_cast<uint4 *>(ham + grp * 256);
		g4[lane] = out[0];
		g4[lane + 64] = out[1];
	}
}

} // namespace

// Workgroups the chip holds at once for kernel variant `v` on the current device (occupancy x compute units).
// (contexts may be driven from several host threads, one each: the cache is filled under a lock)
static int ballot_resident_wgs(int v, const void *fn, int threads, int cus) {
	static std::mutex mu;
	static int cache[16][128];
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
	std::lock_guard<std::mutex> lock(mu);
	if (!cache[dev][v]) {
		int per_cu = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
		cache[dev][v] = per_cu;
	}
	return cache[dev][v] * cus;
}

int ballot_max_wgs(int cus) { return (cus > 0 ? cus : 256) * 8; } // 8 workgroups of 4 waves per CU at most

template <int NT>
static hipError_t launch_ballot_update_nt(UpdateParams &p, hipStream_t stream, int *grid_out, hipEvent_t stop, hipEvent_t start) {
	p.nwg = (p.nunits + NT / GROUP - 1) / (NT / GROUP); // (p.nunits counts column groups incl. the dead ones: 4 per wave)
	const bool fused = p.nlevels > 1;
	const bool usej = fused ? p.jham[0] != nullptr : p.jdst != nullptr;
	const bool subl = p.slY != 0;
	const bool streamed = fused && p.nt_stream;
	const bool batch = fused && p.nrep > 0;
	const bool count = p.cnt_acc != nullptr;
	if (count && (!fused || batch || usej || subl || p.color != 0)) return hipErrorInvalidValue; // (ising_update.cpp asks only where it applies)
	if (batch && (usej || subl || NT != BAL_THREADS)) return hipErrorInvalidValue;
	if (fused && subl && (usej || NT != BAL_THREADS || p.slY % p.H != 0)) return hipErrorInvalidValue; // (ising_capi.cpp keeps those on one launch per colour)
	if (batch) { // a level = the units of all lattices
		p.nwg_rep = p.nwg;
		p.nwg = p.nwg_rep * p.nrep;
		p.rep_magic = (uint32_t)((0x100000000ull + (unsigned long long)p.nwg_rep - 1) / (unsigned long long)p.nwg_rep);
	}
	// kernel instance: bit 0 couplings, 1 sub-lattices, 2 fused, 3 non-temporal lattice words, 4 batched, 5 in-launch counts
	const int v = (usej ? 1 : 0) | (subl ? 2 : 0) | (fused ? 4 : 0) | (streamed ? 8 : 0) | (batch ? 16 : 0) | (count ? 32 : 0);
	// (the launch is `LAUNCH(instance)`: hipExtLaunchKernelGGL needs the template arguments as written)
#define BAL_INSTANCES(X)                                                                                              \
	X(0, (ballot_update_k<false, false, false, NT>))                                                                   \
	X(1, (ballot_update_k<false, true, false, NT>))                                                                    \
	X(2, (ballot_update_k<true, false, false, NT>))                                                                    \
	X(3, (ballot_update_k<true, true, false, NT>))                                                                     \
	X(4, (ballot_update_k<false, false, true, NT>))                                                                    \
	X(5, (ballot_update_k<false, true, true, NT>))                                                                     \
	X(6, (ballot_update_k<true, false, true, BAL_THREADS>))                                                            \
	X(12, (ballot_update_k<false, false, true, BAL_THREADS, true>))                                                    \
	X(13, (ballot_update_k<false, true, true, BAL_THREADS, true>))                                                     \
	X(14, (ballot_update_k<true, false, true, BAL_THREADS, true>))                                                     \
	X(20, (ballot_update_k<false, false, true, BAL_THREADS, false, true>))                                             \
	X(28, (ballot_update_k<false, false, true, BAL_THREADS, true, true>))                                              \
	X(36, (ballot_update_k<false, false, true, BAL_THREADS, false, false, true>))                                      \
	X(44, (ballot_update_k<false, false, true, BAL_THREADS, true, false, true>))
	const void *fn = nullptr;
	switch (v) {
#define BAL_FN(code, inst) case code: fn = (const void *)inst; break;
	BAL_INSTANCES(BAL_FN)
#undef BAL_FN
	default: return hipErrorInvalidValue;
	}
	// Plain launches: one workgroup per unit, handed out by the hardware dispatcher (a persistent grid striding over the
	// units runs all workgroups in lockstep -- every wave in its draw phase, then every wave in its word phase -- and
	// measured 11 % slower).  Fused launches: as many workgroups as the chip holds; a few more are harmless (they find
	// the tickets gone), so the occupancy query need not be exact.
	const long long total = (long long)p.nwg * p.nlevels;
	if (p.cus <= 0) { // (callers pass the slab's device's count; a bare UpdateParams asks)
		int dev = 0, n = 0;
		if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
		p.cus = n;
	}
	const int cus = p.cus;
	long long grid = fused ? std::min<long long>(std::min(ballot_resident_wgs(v + (NT == 256 ? 0 : 64), fn, NT, cus), ballot_max_wgs(cus) * 256 / NT), total) : total;
	if (fused) {
		// Fewer workgroups than the chip holds when a level has few tickets: a unit's parents are one level = p.nwg tickets
		// back, and a workgroup that finds them unfinished holds its slot asleep (ising_create picks wg_per_cu)
		if (p.grid_cap > 0) grid = std::min<long long>(grid, p.grid_cap); // (ISING_FUSED_WGS, read when the context was created)
		else if (p.wg_per_cu > 0) grid = std::min<long long>(grid, (long long)p.wg_per_cu * cus);
	}
	if (grid < 1) grid = 1;
	if (grid < p.tickets2 || total < p.tickets2) p.tickets2 = 0; // (k counters need a workgroup of every class)
	const dim3 g((unsigned)grid), block(NT);
	// `stop`: an event that fires when this launch is done, hung on the dispatch packet itself (hipExtLaunchKernelGGL) --
	// a hipEventRecord behind the launch is a packet of its own that drains the queue: 7 us between two 650 us launches
	// (`start`: the same for the moment the launch begins -- the exchange statistics, ising_exchange_stats_begin)
	switch (v) {
#define BAL_LAUNCH(code, inst) case code: hipExtLaunchKernelGGL(inst, g, block, 0, stream, start, stop, 0, p); break;
	BAL_INSTANCES(BAL_LAUNCH)
#undef BAL_LAUNCH
	default: return hipErrorInvalidValue;
	}
#undef BAL_INSTANCES
	if (grid_out) *grid_out = (int)grid;
	return hipGetLastError();
}

#if defined(ISING_FUSED_TRACE)
static void ballot_trace_is_split();
#endif
// Split launch (ballot_split_k): p as for a fused launch of the whole slab; the caller has zeroed sp_ctr and sp_flags on the stream.
hipError_t launch_ballot_split(UpdateParams &p, hipStream_t stream, int *grid_out, hipEvent_t stop, hipEvent_t start) {
	if (grid_out) *grid_out = 0;
	p.nwg = (p.nunits + BAL_THREADS / GROUP - 1) / (BAL_THREADS / GROUP);
	if (p.nlevels < 2 || p.nwg < 8 || p.slY != 0 || p.jham[0] != nullptr || p.nrep > 0 || p.color != 0 || p.nt_stream || !p.sp_masks || !p.sp_ctr || !p.sp_flags) return hipErrorInvalidValue;
	if (p.cus <= 0) p.cus = 256;
	const bool count = p.cnt_acc != nullptr;
	const long long grid = std::max<long long>(8, std::min<long long>((long long)p.sp_cap * 8, (long long)(p.wg_per_cu > 0 ? p.wg_per_cu : 6) * p.cus));
	const dim3 g((unsigned)grid), block(BAL_THREADS);
	if (count) hipExtLaunchKernelGGL((ballot_split_k<true>), g, block, 0, stream, start, stop, 0, p);
	else hipExtLaunchKernelGGL((ballot_split_k<false>), g, block, 0, stream, start, stop, 0, p);
	if (grid_out) *grid_out = (int)grid;
#if defined(ISING_FUSED_TRACE)
	ballot_trace_is_split();
#endif
	return hipGetLastError();
}

#if defined(ISING_FUSED_TRACE)
static bool g_trace_split = false;
static void ballot_trace_is_split() { g_trace_split = true; }
void ballot_trace_dump() {
	if (g_trace_split) {
		static const char *sname[16] = {"draw ticket", "draw: decode + slot", "draw rows", "draw: write-back", "word ticket", "word: decode + wait", "word rows", "word: drain",
		                                "", "", "", "", "", "", "TOTAL", ""};
		unsigned long long h[16];
		if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trace), sizeof(h)) != hipSuccess) return;
		fprintf(stderr, "split trace (wave 0 of every workgroup): %llu draw units, %llu word units; polls that slept: %llu for a slot, %llu for parents / masks in %llu units (%llu of them with the masks late)\n",
		        h[8], h[9], h[10], h[11], h[12], h[13]);
		for (int i = 0; i < 15; ++i) {
			if (!sname[i][0]) continue;
			fprintf(stderr, "  %-22s %6.2f %% of workgroup time, %8.1f cycles per unit\n", sname[i], 100.0 * (double)h[i] / (double)h[14], (double)h[i] / (double)(h[9] ? h[9] : 1));
		}
		for (auto &v : h) v = 0;
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), h, sizeof(h));
		return;
	}
	static const char *name[16] = {"ticket pick-up", "decode", "completion counters", "row prologue", "draw", "next ticket", "barrier", "word: issue",
	                               "", "", "word: wait loads", "word: flips+store", "", "drain", "TOTAL", ""};
	unsigned long long h[16];
	if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trace), sizeof(h)) != hipSuccess) return;
	fprintf(stderr, "fused trace (wave 0 of every workgroup): %llu units, %llu polls that slept in %llu units\n", h[8], h[9], h[15]);
#if defined(ISING_FUSED_TRACE_COUNTS)
	fprintf(stderr, "  sleeping units: %llu in the first eighth of a level's visiting order, %llu in the last; %llu at level 1, %llu at the last level; %llu at positions 0..7; "
	                "%llu on even positions; %llu slept 64 polls or more, %llu 256 or more\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
	{
		unsigned long long g[32];
		if (hipMemcpyFromSymbol(g, HIP_SYMBOL(g_hist), sizeof(g)) != hipSuccess) return;
		fprintf(stderr, "  units of 12 bins and more by dispatch round: %llu %llu %llu, fourth and later %llu\n", h[10], h[11], h[12], h[13]);
		for (int k = 0; k < 2; k++) {
			fprintf(stderr, "  unit durations %s, bins of 2048 cycles per row:", k ? "from ticket pick-up" : "behind the wait for the parents");
			for (int i = 0; i < 16; i++) fprintf(stderr, " %llu", g[16 * k + i]);
			fprintf(stderr, "\n");
		}
		for (auto &v : g) v = 0;
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_hist), g, sizeof(g));
		for (auto &v : h) v = 0;
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), h, sizeof(h));
	}
	return;
#endif
	for (int i = 0; i < 15; ++i) {
		if (!name[i][0]) continue;
		fprintf(stderr, "  %-22s %6.2f %% of workgroup time, %8.1f cycles per unit\n", name[i], 100.0 * (double)h[i] / (double)h[14], (double)h[i] / (double)(h[8] ? h[8] : 1));
	}
	for (auto &v : h) v = 0;
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), h, sizeof(h));
}
#endif
hipError_t launch_ballot_update(UpdateParams &p, hipStream_t stream, int *grid_out, hipEvent_t stop, hipEvent_t start) {
	if (grid_out) *grid_out = 0;
	if (p.nunits <= 0) {
		if (start) if (const hipError_t e = hipEventRecord(start, stream); e != hipSuccess) return e;
		return stop ? hipEventRecord(stop, stream) : hipSuccess;
	}
	if (p.nlevels < 1) p.nlevels = 1;
	return launch_ballot_update_nt<BAL_THREADS>(p, stream, grid_out, stop, start);
}

// the BALLOT_MEASURE_SLOTS partial sums of lattice 0 into out[0] (up spins), out[1] (bond sum), and the slots back to zero
__global__ void __launch_bounds__(64) measure_fold_k(unsigned long long *__restrict__ acc, unsigned long long *__restrict__ out) {
	const int lane = threadIdx.x;
	unsigned long long u = 0, a = 0;
	if (lane < MEASURE_SLOTS) {
		u = acc[lane * 8];
		a = acc[lane * 8 + 1];
		acc[lane * 8] = 0;
		acc[lane * 8 + 1] = 0;
	}
	u = wave_sum(u);
	a = wave_sum(a);
	if (lane == 0) {
		out[0] = u;
		out[1] = a;
	}
}

// in-launch counts: measurement m = the sum of its slots -- out[2 m] the up spins (the first `n_up` slots: both colours' waves), out[2 m + 1] the
// equal bonds (the rest: the white level's waves; none when the call did not ask for the energy)
__global__ void __launch_bounds__(THREADS) count_fold_k(const uint32_t *__restrict__ slots, size_t per_meas, size_t n_up, unsigned long long *__restrict__ out) {
	const uint32_t *s = slots + (size_t)blockIdx.x * per_meas;
	unsigned long long v = 0, b = 0;
	for (size_t i = threadIdx.x; i < per_meas; i += THREADS) { if (i < n_up) v += s[i]; else b += s[i]; }
	v = wave_sum(v);
	b = wave_sum(b);
	__shared__ unsigned long long part[2][THREADS / 64];
	if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = v; part[1][threadIdx.x >> 6] = b; }
	__syncthreads();
	if (threadIdx.x < 2) { unsigned long long t = 0; for (int k = 0; k < THREADS / 64; k++) t += part[threadIdx.x][k]; out[2 * blockIdx.x + threadIdx.x] = t; }
}

hipError_t launch_count_fold(const uint32_t *slots, size_t per_meas, size_t n_up, int nmeas, unsigned long long *out, hipStream_t stream) {
	if (nmeas <= 0) return hipSuccess;
	hipLaunchKernelGGL(count_fold_k, dim3((unsigned)nmeas), dim3(THREADS), 0, stream, slots, per_meas, n_up, out);
	return hipGetLastError();
}

hipError_t launch_measure_fold(unsigned long long *acc, unsigned long long *out, hipStream_t stream) {
	hipLaunchKernelGGL(measure_fold_k, dim3(1), dim3(64), 0, stream, acc, out);
	return hipGetLastError();
}

hipError_t launch_ballot_measure(const ReplicaParams *reps, int nrep, int gx, int Y, unsigned long long *acc, hipStream_t stream) {
	const int R = Y >= 4096 ? 16 : 8; // rows a wave marches: the (above, same, below) window slides, two loads per row instead of four
	const long long waves = (long long)nrep * ((gx + 3) / 4) * ((Y + R - 1) / R);
	hipLaunchKernelGGL(ballot_measure_k, flat_grid((waves + THREADS / 64 - 1) / (THREADS / 64)), dim3(THREADS), 0, stream, reps, nrep, gx, Y, R, acc);
	return hipGetLastError();
}

hipError_t launch_ballot_init(const InitParams &p, hipStream_t stream) {
	const long long waves = (long long)((p.gx + 3) / 4) * p.Y;
	hipLaunchKernelGGL(ballot_init_k, flat_grid((waves + THREADS / 64 - 1) / (THREADS / 64)), dim3(THREADS), 0, stream, p);
	return hipGetLastError();
}

// -J coupling rows (X/4 bytes each) between the nibble / dense-plane forms and the ballot planes, in place
hipError_t launch_ham_to_ballot(uint64_t *ham, int gx, long long rows, hipStream_t stream) {
	const long long ngroups = rows * (gx / 4);
	if (ngroups <= 0) return hipSuccess;
	long long blocks = (ngroups + THREADS / 64 - 1) / (THREADS / 64);
	if (blocks > 16384) blocks = 16384;
	hipLaunchKernelGGL(ham_nibbles_to_ballot_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, ham, ngroups);
	return hipGetLastError();
}

hipError_t launch_ham_ballot_to_planes(uint64_t *ham, int gx, long long rows, hipStream_t stream) {
	const long long ngroups = rows * (gx / 4);
	if (ngroups <= 0) return hipSuccess;
	long long blocks = (ngroups + THREADS / 64 - 1) / (THREADS / 64);
	if (blocks > 16384) blocks = 16384;
	hipLaunchKernelGGL(ham_ballot_to_planes_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, ham, ngroups);
	return hipGetLastError();
}

// rows x (gx/4) groups of 64 ballot words <-> 128 dense words; both buffers hold `rows` rows of gx*128 bytes
hipError_t launch_ballot_to_dense(const uint64_t *bal, uint32_t *dense, int gx, long long rows, hipStream_t stream) {
	const long long ngroups = rows * ((gx + 3) / 4);
	if (ngroups <= 0) return hipSuccess;
	long long blocks = (ngroups + THREADS / 64 - 1) / (THREADS / 64);
	if (blocks > 16384) blocks = 16384;
	hipLaunchKernelGGL(ballot_to_dense_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, bal, dense, ngroups, gx);
	return hipGetLastError();
}

hipError_t launch_dense_to_ballot(const uint32_t *dense, uint64_t *bal, int gx, long long rows, hipStream_t stream) {
	const long long ngroups = rows * ((gx + 3) / 4);
	if (ngroups <= 0) return hipSuccess;
	long long blocks = (ngroups + THREADS / 64 - 1) / (THREADS / 64);
	if (blocks > 16384) blocks = 16384;
	hipLaunchKernelGGL(dense_to_ballot_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, dense, bal, ngroups, gx);
	return hipGetLastError();
}

} // namespace ising
