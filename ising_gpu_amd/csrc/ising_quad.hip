// ising_quad.hip -- small and narrow lattices (round 5): every draw made ONCE, ahead of the lattice; the cheap part repeated instead.
//
// Below ~2^25 spins a colour half-sweep is a microsecond of arithmetic, and whatever exchanges rows between workgroups once
// per half-sweep (a launch boundary, a counter in memory) costs several.  The tile launches of round 4 (ising_dense.hip:
// dense_tile_k) buy S sweeps without an exchange by repeating the neighbours' updates of a halo -- draws included, x 1.5 of the
// part that is 94 % of the instructions.  But a draw depends on (seed, site, iteration) alone: the accept masks of a pass of T
// sweeps can be made ahead by code that knows nothing of the lattice, at the rate of the large lattices' draw phase and
// without redundancy; what has to run in order -- the word phase, ~75 of ~780 vector instructions per 4096 sites -- can then
// afford the halo.
//
// Layout ("quad": the wave of the reference's thread block as it is, 4 rows x 16 lanes).  Per colour [Y/4 row groups][X/2048
// blocks][64 words]:
//   word (R, bx, p),  p = 32 j + 4 m + q   <->  draw block B = 8 j + m, Philox output q (as in the ballot layout)
//   bit l = 16 r4 + tx of the word         <->  reference thread 64 (R & 3) + l of block (bx, R / 4): row 4 R + r4, vector
//                                               32 bx + 16 j + tx, site s(m, q) = {2m, 16+2m, 2m+1, 17+2m}[q]
// so the lane mask of v_cmp(draw, threshold) is a storage word here too -- for every X the reference accepts (a multiple of
// 2048), where the ballot layout's wave columns want 8192.  Vertical neighbours are the word itself shifted by 16 bits (+ 16
// bits of the row group above / below), horizontal ones the same bit of another word (sites 0 / 31: the neighbouring lane bit).
//
// quad_pass_k = ONE launch per pass of T sweeps:
//   tiles     workgroup t < ntiles: C row groups x the whole width of both colours + HG halo row groups per side in LDS; 2 T
//             levels over a region that shrinks a row per level; the masks of the next level on their way into
//             accumulation registers; reads one lattice buffer, writes its tile to the other (quad_word_part)
//   drawing   the workgroups behind them: the masks of the NEXT pass -- (level, row group, block) -> 1 KiB (c3, c4 per word),
//             scalar stores straight into the mask buffer; no lattice, no barrier, no order (quad_draw_part)
// ising_update.cpp (sweep_quad): launch k = word pass k on the masks launch k - 1 drew + the draws of pass k + 1; one stream,
// two mask buffers.  DESIGN 4.3; what was measured and dropped on the way: LAB_NOTES 12.
#include "ising_device.hpp"
#include <algorithm>
#include <mutex>
#include <cstdio>

namespace ising {
namespace {

constexpr uint64_t Q_LANE0 = 0x0001000100010001ull;  // tx = 0 of each row
constexpr uint64_t Q_LANE15 = 0x8000800080008000ull; // tx = 15
constexpr uint64_t Q_EVEN = 0x0000FFFF0000FFFFull;   // rows 4 R, 4 R + 2
// Levels of masks in flight per wave.  Round 5 kept three (this level's, the next two on their way); round 6 measured two against it (A/B builds, 2 .. 5): the
// latency of a level's 1 KiB per item is covered by ONE level of word work, and what the third set cost -- eight to twelve accumulation registers, a level loop
// unrolled six times instead of twice -- was worth 3-6 % wherever the tiles are many (31 x 2048^2 2420 -> 2572 flips/ns, 4096 x 16384 2359 -> 2480, 6144^2 2278 ->
// 2364, 12288 x 768 at three items a wave 1496 -> 1664: 88 registers instead of 100) and nothing where they are few (2048^2 2227 / 2243); four and five: slower
// everywhere (profiles/quad_depth_probe_r06.txt).
#ifndef ISING_QUAD_DEPTH
#define ISING_QUAD_DEPTH 2
#endif
constexpr int Q_DEPTH = ISING_QUAD_DEPTH;            // levels of masks in flight per wave (sets of accumulation registers)
constexpr int Q_UNROLL = (Q_DEPTH % 2) ? 2 * Q_DEPTH : Q_DEPTH; // levels a turn of the level loop: colour and mask set of each are compile-time

__device__ __forceinline__ constexpr int qword(int j, int m, int q) { return 32 * j + 4 * m + q; }
__device__ __forceinline__ int qword_of_site(int j, int s) {
	const int w = s >> 4, z = s & 15;
	return qword(j, z >> 1, 2 * (z & 1) + w);
}

// The compares' results live in fixed scalar registers (inline asm cannot name halves of an SGPR tuple operand)
#define QSG(a, b) "s[84+" #a ":84+" #b "]"
#define Q_CLOB16 "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"

// Measurement build (make variant NAME=qtrace DEFS=-DISING_QUAD_TRACE): wave 0 of every word workgroup clocks where its time goes; quad_trace_dump()
// prints the sums when the slab is destroyed.  Never in the product library.
#if defined(ISING_QUAD_TRACE)
__device__ unsigned long long g_qtrace[16];
#define QTRC(i) do { if (wi == 0) { const long long t_ = wall_clock64(); if (lane == 0) qtr[i] += (unsigned long long)(t_ - qt_last); qt_last = t_; } } while (0)
#else
#define QTRC(i) do {} while (0)
#endif

__device__ __forceinline__ uint32_t lds_addr(const void *q) {
	return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)q;
}

// ---- draws.  The drawing workgroups of a launch are as many as find room next to its tiles, each wave of them takes an EQUAL share of the pass's draws --
// a run of consecutive quarter items (4 Philox blocks = 256 B of masks each; level major, item, quarter) -- so the launch's draws end together: workgroups
// of one item a wave, dealt by the dispatcher, ended a launch of 2048^2 on a second, half-empty round (21.6 us a launch where the draws are 10.4 us of the chip).
__device__ __forceinline__ void quad_draw_part(const QuadDrawParams &p, long long wg, int wi, int lane, int NW, uint64_t *lds) {
	uint4 *blk_const = reinterpret_cast<uint4 *>(lds) + wi * 16; // (private to the wave: 256 B)
	const int NI = p.NRG * p.gx;
	// a batch (p.rep): the lattices one after another, each level major -- a wave's run crosses a lattice now and then and picks up its record there
	const int nrep = p.rep ? p.nrep : 1;
	const unsigned per_level = (unsigned)NI * 4u;
	const unsigned long long per_lat = (unsigned long long)p.nlev * per_level;
	const unsigned long long Q = per_lat * (unsigned)nrep, W = (unsigned long long)p.nwaves;
	const unsigned long long w = (unsigned long long)wg * (unsigned)NW + (unsigned)wi;
	if (w >= W) return;
	unsigned long long q_lo = Q * w / W, q_hi = Q * (w + 1) / W;
	{ // (wave-uniform: the scalar unit's)
		const uint32_t a = __builtin_amdgcn_readfirstlane((uint32_t)q_lo), b = __builtin_amdgcn_readfirstlane((uint32_t)(q_lo >> 32));
		const uint32_t c = __builtin_amdgcn_readfirstlane((uint32_t)q_hi), d = __builtin_amdgcn_readfirstlane((uint32_t)(q_hi >> 32));
		q_lo = ((unsigned long long)b << 32) | a;
		q_hi = ((unsigned long long)d << 32) | c;
	}
	if (q_lo >= q_hi) return;
	int r = p.rep ? (int)(q_lo / per_lat) : 0;
	const unsigned rem_lat = (unsigned)(q_lo - (unsigned long long)r * per_lat);
	int level = (int)(rem_lat / per_level);
	unsigned rem = rem_lat - (unsigned)level * per_level;
	int n = (int)(rem >> 2), qq = (int)(rem & 3u);
	int R = n / p.gx, bx = n - R * p.gx;
	uint32_t seed_lo = p.seed_lo, seed_hi = p.seed_hi;
	uint32_t thr3 = p.n3, thr4 = p.n4;
	uint64_t *masks = p.masks;
	auto pick_up = [&](int rr) { // lattice rr's record (scalar loads: rr is wave-uniform)
		const QuadRec *rc = p.rep + rr;
		seed_lo = __builtin_amdgcn_readfirstlane(rc->seed_lo);
		seed_hi = __builtin_amdgcn_readfirstlane(rc->seed_hi);
		thr3 = __builtin_amdgcn_readfirstlane(rc->n3);
		thr4 = __builtin_amdgcn_readfirstlane(rc->n4);
		const uintptr_t m = (uintptr_t)(rc->masks + p.mask_off);
		const uint32_t m_lo = __builtin_amdgcn_readfirstlane((uint32_t)m), m_hi = __builtin_amdgcn_readfirstlane((uint32_t)(m >> 32));
		masks = reinterpret_cast<uint64_t *>(((uintptr_t)m_hi << 32) | m_lo);
	};
	if (p.rep) pick_up(r);
	uint32_t k2y = seed_hi + 2u * PHILOX_W1;
	bool new_level = true;
	uint32_t seed_lo_cy = 0;
	for (unsigned long long q = q_lo; q < q_hi;) { // item by item: the quarters [qq, qe) of item n of `level` (a wave's first and last item may be partial)
		const int qe = (int)min((unsigned long long)4, (unsigned long long)qq + (q_hi - q));
		if (new_level) {
			const uint32_t color = (uint32_t)level & 1u, it = p.it + ((uint32_t)level >> 1);
			const uint32_t cx_base = 16u * (2u * it + color);
			seed_lo_cy = seed_lo ^ (uint32_t)((2ull * it + color) >> 28); // counter word 1 enters round 1 next to the key (dense_update_k)
			__builtin_amdgcn_wave_barrier();
			if (lane < 16) {
				const PhiloxBlockConst kc = philox_block_const(cx_base + (uint32_t)lane, seed_lo, seed_hi);
				blk_const[lane] = make_uint4(kc.s0, kc.s1, kc.s2, 0u);
			}
			__builtin_amdgcn_wave_barrier();
			__threadfence_block();
			new_level = false;
		}
		const uint32_t tid = (((uint32_t)R >> 2) * (uint32_t)p.gx + (uint32_t)bx) * 256u + ((uint32_t)R & 3u) * 64u + (uint32_t)lane;
		const PhiloxRow pr = philox_row_setup(tid, seed_lo_cy, k2y);
		const uint64_t *dst0 = masks + ((size_t)level * (size_t)NI + (size_t)n) * 128;
		const uint32_t d_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)dst0), d_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)dst0 >> 32));
		const uint64_t *dst = reinterpret_cast<const uint64_t *>(((uintptr_t)d_hi << 32) | d_lo);
		for (int qi = qq; qi < qe; ++qi) {
			// (the quarter's four block constants at once: LDS and scalar memory share one counter, so every wait for a constant is also a wait for the scalar
			// stores before it -- one wait a quarter instead of one a block)
			const uint4 *kcs = blk_const + 4 * qi;
			uint4 kq[4] = {kcs[0], kcs[1], kcs[2], kcs[3]};
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kq[0].x), "+v"(kq[0].y), "+v"(kq[0].z), "+v"(kq[1].x), "+v"(kq[1].y), "+v"(kq[1].z), "+v"(kq[2].x), "+v"(kq[2].y), "+v"(kq[2].z), "+v"(kq[3].x), "+v"(kq[3].y), "+v"(kq[3].z) :: "memory");
			static_for<4>([&](auto B) {
				uint32_t o0, o1, o2, o3;
				const uint4 kc = kq[B.value];
				philox_block_pre(pr, PhiloxBlockConst{kc.x, kc.y, kc.z}, seed_lo, seed_hi, o0, o1, o2, o3);
				const uint64_t *dstp = dst + 32 * qi + 8 * B.value; // (c3, c4) of output o of block 4 qi + B = word p = 16 qi + 4 B + o: 16 bytes at 16 p
				const uint32_t t3 = thr3, t4 = thr4;
				asm volatile("v_cmp_gt_u32_e64 " QSG(0, 1) ", %0, %2\n\tv_cmp_gt_u32_e64 " QSG(2, 3) ", %1, %2\n\t"
				             "v_cmp_gt_u32_e64 " QSG(4, 5) ", %0, %3\n\tv_cmp_gt_u32_e64 " QSG(6, 7) ", %1, %3\n\t"
				             "v_cmp_gt_u32_e64 " QSG(8, 9) ", %0, %4\n\tv_cmp_gt_u32_e64 " QSG(10, 11) ", %1, %4\n\t"
				             "v_cmp_gt_u32_e64 " QSG(12, 13) ", %0, %5\n\tv_cmp_gt_u32_e64 " QSG(14, 15) ", %1, %5\n\t"
				             "s_store_dwordx4 " QSG(0, 3) ", %6, 0x0\n\ts_store_dwordx4 " QSG(4, 7) ", %6, 0x10\n\t"
				             "s_store_dwordx4 " QSG(8, 11) ", %6, 0x20\n\ts_store_dwordx4 " QSG(12, 15) ", %6, 0x30"
				             :: "s"(t3), "s"(t4), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(dstp)
				             : "memory", Q_CLOB16);
			});
		}
		q += (unsigned)(qe - qq);
		qq = 0; // (whatever follows starts an item: a wave's share is one run)
		++n;
		if (++bx == p.gx) { bx = 0; ++R; }
		if (n == NI) {
			n = 0; R = 0; bx = 0; ++level; new_level = true;
			if (level == p.nlev && p.rep) { // the next lattice of the batch
				level = 0;
				if (++r < nrep) { pick_up(r); k2y = seed_hi + 2u * PHILOX_W1; }
			}
		}
	}
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// The accept masks of a word wave's items wait in ACCUMULATION registers: a[4 n .. 4 n + 3], n = set * MAXI + item (Q_DEPTH sets = levels in flight: two), and
// one more quad that takes the loads of items a level does not have (every level issues exactly MAXI loads: vmcnt counts them).  Loaded and read by inline
// assembly alone -- the compiler has no use for these registers in a kernel without MFMA and without spills, knows nothing of loads in flight, and so cannot
// copy one early (tracked loads into vector registers were fenced with vmcnt(0) on either side: no prefetch left).  The build checks the ISA: no accumulation
// register outside these statements (tools/check_asm_loads.py --quad).
#define QM_CASES(X) X(0, 0, 1, 2, 3) X(1, 4, 5, 6, 7) X(2, 8, 9, 10, 11) X(3, 12, 13, 14, 15) X(4, 16, 17, 18, 19) X(5, 20, 21, 22, 23) X(6, 24, 25, 26, 27) \
	X(7, 28, 29, 30, 31) X(8, 32, 33, 34, 35) X(9, 36, 37, 38, 39) X(10, 40, 41, 42, 43) X(11, 44, 45, 46, 47) X(12, 48, 49, 50, 51) \
	X(13, 52, 53, 54, 55) X(14, 56, 57, 58, 59) X(15, 60, 61, 62, 63) X(16, 64, 65, 66, 67)
template <int N>
__device__ __forceinline__ void qm_load(int off, const void *base) {
#define QM_L(I, A, B, C, D) if constexpr (N == I) asm volatile("global_load_dwordx4 a[" #A ":" #D "], %0, %1" :: "v"(off), "s"(base) : "memory", "a" #A, "a" #B, "a" #C, "a" #D);
	QM_CASES(QM_L)
#undef QM_L
}
template <int N>
__device__ __forceinline__ void qm_read(uint32_t &x0, uint32_t &x1, uint32_t &x2, uint32_t &x3) {
#define QM_R(I, A, B, C, D) if constexpr (N == I) asm volatile("v_accvgpr_read_b32 %0, a" #A "\n\tv_accvgpr_read_b32 %1, a" #B "\n\tv_accvgpr_read_b32 %2, a" #C "\n\tv_accvgpr_read_b32 %3, a" #D : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) :: "memory");
	QM_CASES(QM_R)
#undef QM_R
}

// ---- words: `nlev` levels (black first) of tile `tile`.  Everything that indexes is wave-uniform and lives on the scalar unit (a lone wave issues an
// instruction every four or five cycles whatever its kind); a wave works on its one to three items in turn (pairs with all LDS reads first cost the registers
// of a sixth wave per SIMD: ISING_QUAD_PAIR); the masks of the next level are on their way into accumulation registers (two sets -- Q_DEPTH --, the level loop
// unrolled by two = colours x sets).
template <int MAXI>
__device__ __forceinline__ void quad_word_part(const QuadWordParams &p, int tile, int wi, int lane, int NW, uint64_t *q_lds) {
	const int gx = p.gx, NRG = p.NRG, HG = p.HG;
	const int NG = p.C + 2 * HG;
	// whose tile: a batch's tiles are dealt lattice by lattice; the lattice's planes, masks and print-point words by its record (scalar loads)
	const uint64_t *srcp[2] = {p.src[0], p.src[1]};
	uint64_t *dstp[2] = {p.dst[0], p.dst[1]};
	const uint64_t *masks = p.masks;
	unsigned long long *cnt = p.cnt, *cnt_eq = p.cnt_eq;
	if (p.rep) {
		const int r = __builtin_amdgcn_readfirstlane(tile / p.tiles_per_lat);
		tile -= r * p.tiles_per_lat;
		const QuadRec *rc = p.rep + r;
		uint64_t *const planes = rc->quad;
		srcp[0] = planes + p.src_off[0]; srcp[1] = planes + p.src_off[1];
		dstp[0] = planes + p.dst_off[0]; dstp[1] = planes + p.dst_off[1];
		masks = rc->masks + p.mask_off;
		if (cnt) cnt += (size_t)r * p.cnt_stride;
		if (cnt_eq) cnt_eq += (size_t)r * p.cnt_stride;
	}
	const int A = tile * p.C;
	const int Cc = min(p.C, NRG - A);
	const int gw = gx * 64;                 // words per row group
	const int plane = NG * gw;              // words per colour in LDS
	uint64_t *lat = q_lds;                  // [2][NG][gx][64]
#if defined(ISING_QUAD_TRACE)
	__shared__ unsigned long long qtr[16];
	if (threadIdx.x < 16) qtr[threadIdx.x] = 0;
	__syncthreads();
	long long qt_last = wall_clock64();
	const long long qt_start = qt_last;
#endif
#if !defined(ISING_QUAD_ABL_NOPRIO)
	__builtin_amdgcn_s_setprio(3);
#endif
	// (the draws of the pass to come share the chip: a word pass is a chain of short levels, theirs is throughput)
	auto wrapR = [&](int g) { // row group of the lattice behind local group g (no division: a halo wraps around a short lattice a few times at most)
		int R = A - HG + g;
		while (R < 0) R += NRG;
		while (R >= NRG) R -= NRG;
		return R;
	};
	// item k of this wave at any level: the (wi + k NW)-th (row group, block) of the level's active range, row group major
	int it_g[MAXI], it_b[MAXI];
#pragma unroll
	for (int k = 0; k < MAXI; ++k) {
		const int n = wi + k * NW;
		it_g[k] = __builtin_amdgcn_readfirstlane(n / gx);
		it_b[k] = __builtin_amdgcn_readfirstlane(n - (n / gx) * gx);
	}
	const int lane16 = lane * 16;
	const uint32_t gw_b = (uint32_t)gw * 8, plane_b = (uint32_t)plane * 8;
	// What indexes a level is computed ONCE, by the lanes in parallel -- lane L of rb_tab[k] / mk_tab[k] = where item k of level L sits in a colour plane (bytes) /
	// in the pass's masks (KiB), ~0 where the level's active range does not reach it -- and the level loop picks its scalars up with v_readlane.
	constexpr uint32_t ABSENT = 0xFFFFFFFFu;
	uint32_t rb_tab[MAXI], mk_tab[MAXI];
	{
		const int L = lane;
		const int e = p.nlev - 1 - L, ge = (e + 3) >> 2;
		const int lo = HG - ge, nrg = L < p.nlev ? Cc + 2 * ge : 0;
#pragma unroll
		for (int k = 0; k < MAXI; ++k) {
			const bool on = it_g[k] < nrg;
			const int g = lo + it_g[k];
			rb_tab[k] = on ? (uint32_t)g * gw_b + (uint32_t)it_b[k] * 512u : ABSENT;
			mk_tab[k] = on ? ((uint32_t)L * (uint32_t)NRG + (uint32_t)wrapR(g)) * (uint32_t)gx + (uint32_t)it_b[k] : ABSENT;
		}
	}
	// the masks of level Lp into set SLOT: exactly MAXI loads (an item the level does not have, or a level past the pass's last: the spare quad)
	auto fetch = [&](int Lp, auto SLOT) {
#if defined(ISING_QUAD_ABL_NOFETCH) // (ablation builds: timing only, results wrong by design -- never in the product library)
		return;
#endif
		static_for<MAXI>([&](auto K) {
			const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)mk_tab[K.value], Lp & 63);
			if (idx != ABSENT && Lp < p.nlev) qm_load<SLOT.value * MAXI + K.value>(lane16, reinterpret_cast<const char *>(masks) + (size_t)idx * 1024);
			else qm_load<Q_DEPTH * MAXI>(lane16, masks);
		});
	};
	static_for<Q_DEPTH - 1>([&](auto D) { fetch(D.value, std::integral_constant<int, D.value>{}); });
	// the tile and its halo row groups, both colours
	for (int cg = wi; cg < 2 * NG; cg += NW) { // (row group, colour) by wave, its gx x 64 words by lane
		const int c = cg >= NG, g = cg - c * NG;
		const uint64_t *from = (c ? srcp[1] : srcp[0]) + (size_t)wrapR(g) * gw;
		for (int w = lane; w < gw; w += 64) lat[c * plane + g * gw + w] = from[w];
	}
	__syncthreads();
	QTRC(0); // tile load

	// lane p owns word p of every item it works on
	const int j = lane >> 5, m = (lane >> 2) & 7, q = lane & 3;
	int backA, fwdA; // word with site s - 1 / s + 1 of the same vectors (sites 0 / 31: of the neighbouring vector, bit shifted below)
	if (q == 2) backA = qword(j, m, 0);
	else if (q == 3) backA = qword(j, m, 1);
	else if (q == 0) backA = m ? qword(j, m - 1, 2) : qword(j, 7, 3);
	else backA = m ? qword(j, m - 1, 3) : qword(j, 7, 2);
	if (q == 0) fwdA = qword(j, m, 2);
	else if (q == 1) fwdA = qword(j, m, 3);
	else if (q == 2) fwdA = m < 7 ? qword(j, m + 1, 0) : qword(j, 0, 1);
	else fwdA = m < 7 ? qword(j, m + 1, 1) : qword(j, 0, 0);
	const bool specB = m == 0 && q == 0, specF = m == 7 && q == 3;
	const int shB = specB ? 1 : 0, shF = specF ? 1 : 0;
	// byte offsets inside a row group that do not depend on the level: own word, the two side words, and per item the words across the vector seam
	const uint32_t o_me = (uint32_t)lane * 8, o_b1 = (uint32_t)backA * 8, o_f1 = (uint32_t)fwdA * 8;
	uint32_t o_b2[MAXI], o_f2[MAXI]; // (relative to the item's own block)
#pragma unroll
	for (int k = 0; k < MAXI; ++k) {
		const int bx = it_b[k], bxm = bx ? bx - 1 : gx - 1, bxp = bx + 1 < gx ? bx + 1 : 0;
		o_b2[k] = (uint32_t)(((j ? bx : bxm) - bx) * 512 + qword(j ^ 1, 7, 3) * 8);
		o_f2[k] = (uint32_t)(((j ? bxp : bx) - bx) * 512 + qword(j ^ 1, 0, 0) * 8);
	}
	const uint32_t lat_w = lds_addr(lat);
	typedef __attribute__((address_space(3))) const uint64_t *lds_cp;
	typedef __attribute__((address_space(3))) uint64_t *lds_p;
	auto ld = [](uint32_t a) { return *(lds_cp)(uintptr_t)a; };

	// one level; COL = the colour it updates, SLOT = the set its masks are in (level L: colour L & 1, set L mod 3)
	// (MEAS: no update -- the level's sites as they stand, each with its four neighbours of the other colour: the equal bonds of a print point with the energy)
	unsigned long long eq_acc = 0;
	auto level = [&](auto COL, auto SLOT, auto MEAS, int L) {
		constexpr int c = COL.value;
		constexpr bool meas = MEAS.value;
		if (!meas) {
			fetch(L + Q_DEPTH - 1, std::integral_constant<int, (SLOT.value + Q_DEPTH - 1) % Q_DEPTH>{});
			QTRC(1); // prefetch issue
			asm volatile("s_waitcnt vmcnt(%0)" :: "n"((Q_DEPTH - 1) * MAXI) : "memory"); // this level's masks have landed (the level(s) behind them may still be out)
			QTRC(2); // wait for the masks
		}
		const uint32_t S_w = lat_w + (c ? 0u : plane_b), D_w = lat_w + (c ? plane_b : 0u);
		// rows whose side neighbour is site s - 1 (readBack, optimized/main.cu:542): the even rows of a black level, the odd rows of a white one; the lanes
		// whose side word crosses the vector seam take one bit per row from the word across it (made here from two flags: eight registers less to keep)
		constexpr uint64_t BK = c ? ~Q_EVEN : Q_EVEN, FW = ~BK;
		const uint64_t m1 = specB ? (~Q_LANE0 & BK) : BK, m2 = specB ? (Q_LANE0 & BK) : 0ull;
		const uint64_t m3 = specF ? (~Q_LANE15 & FW) : FW, m4 = specF ? (Q_LANE15 & FW) : 0ull;
#ifndef ISING_QUAD_PAIR // items a wave works on at a time: 2 = all LDS reads of a pair first (one exposed round trip per pair, ~25 more registers), 1 = one by one
#define ISING_QUAD_PAIR 1
#endif
		constexpr int STEP = ISING_QUAD_PAIR;
		static_for<(MAXI + STEP - 1) / STEP>([&](auto KP) {
			constexpr int k0 = STEP * KP.value, k1 = (STEP > 1 && k0 + 1 < MAXI) ? k0 + 1 : k0; // (an odd MAXI's last item stands alone)
			constexpr int NK = k1 > k0 ? 2 : 1;
			const uint32_t rb0 = (uint32_t)__builtin_amdgcn_readlane((int)rb_tab[k0], L);
			if (rb0 == ABSENT) return; // (items are dealt row group major: the pair's first is its lowest)
			uint64_t ct[2], upc[2], dnc[2], me[2], x1B[2], x1F[2], x2B[2], x2F[2], c3[2], c4[2];
			uint32_t dst[2];
			bool on[2];
#pragma unroll
			for (int t = 0; t < NK; ++t) {
				constexpr int kk[2] = {k0, k1};
				const int k = kk[t];
				const uint32_t rbk = t ? (uint32_t)__builtin_amdgcn_readlane((int)rb_tab[k], L) : rb0;
				on[t] = rbk != ABSENT;
				const uint32_t rb = on[t] ? rbk : rb0; // (an absent partner reads what its pair's first reads)
				const uint32_t ub = rb >= gw_b ? rb - gw_b : rb, db = rb + gw_b < plane_b ? rb + gw_b : rb; // (a tile's outermost rows: rows nobody needs any more)
				const uint32_t ob2 = on[t] ? o_b2[k] : o_b2[k0], of2 = on[t] ? o_f2[k] : o_f2[k0];
				ct[t] = ld(S_w + rb + o_me);
				x1B[t] = ld(S_w + rb + o_b1);
				x1F[t] = ld(S_w + rb + o_f1);
				x2B[t] = ld(S_w + rb + ob2);
				x2F[t] = ld(S_w + rb + of2);
				upc[t] = ld(S_w + ub + o_me);
				dnc[t] = ld(S_w + db + o_me);
				dst[t] = D_w + rb + o_me;
				me[t] = ld(dst[t]);
				uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0; // (an absent item's set holds whatever it held: its result is not stored)
				if (!meas) {
					if (t == 0) qm_read<SLOT.value * MAXI + k0>(a0, a1, a2, a3);
					else qm_read<SLOT.value * MAXI + k1>(a0, a1, a2, a3);
				}
				c3[t] = ((uint64_t)a1 << 32) | a0;
				c4[t] = ((uint64_t)a3 << 32) | a2;
			}
#pragma unroll
			for (int t = 0; t < NK; ++t) {
				const uint64_t sd = ((x1B[t] << shB) & m1) | ((x2B[t] >> 15) & m2) | ((x1F[t] >> shF) & m3) | ((x2F[t] << 15) & m4);
				const uint64_t up = (ct[t] << 16) | (upc[t] >> 48), dw = (ct[t] >> 16) | (dnc[t] << 48);
				if (meas) {
					if (on[t]) eq_acc += (unsigned long long)(__popcll(~(me[t] ^ up)) + __popcll(~(me[t] ^ ct[t])) + __popcll(~(me[t] ^ dw)) + __popcll(~(me[t] ^ sd)));
					continue;
				}
				const uint32_t flo = flips32((uint32_t)me[t], (uint32_t)up, (uint32_t)ct[t], (uint32_t)dw, (uint32_t)sd, (uint32_t)c3[t], (uint32_t)c4[t]);
				const uint32_t fhi = flips32((uint32_t)(me[t] >> 32), (uint32_t)(up >> 32), (uint32_t)(ct[t] >> 32), (uint32_t)(dw >> 32), (uint32_t)(sd >> 32),
				                             (uint32_t)(c3[t] >> 32), (uint32_t)(c4[t] >> 32));
				if (on[t]) *(lds_p)(uintptr_t)dst[t] = me[t] ^ (((uint64_t)fhi << 32) | flo);
			}
		});
		QTRC(3); // items
#if !defined(ISING_QUAD_ABL_NOBARRIER)
		if (!meas) __syncthreads();
#endif
		QTRC(4); // barrier
	};
	for (int L0 = 0; L0 < p.nlev; L0 += Q_UNROLL) { // (two levels a turn at two sets: the colour and the mask set of each are compile-time)
		static_for<Q_UNROLL>([&](auto I) {
			if (L0 + I.value < p.nlev) level(std::integral_constant<int, I.value & 1>{}, std::integral_constant<int, I.value % Q_DEPTH>{}, std::false_type{}, L0 + I.value);
		});
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the spare quad's loads behind the last level)
	// A print point with the energy: every bond has one white end, so the white sites' equal neighbours are ising_bond_equal's sum.  The pass's last level was
	// white on exactly the tile's own rows; the black rows one beyond either edge were final a level earlier: the same items once more, looking instead of flipping.
	// (the sums of a print point: per wave, then per workgroup through two LDS words, then ONE add per tile into one of eight slots -- a thousand waves adding to
	// one word cost a measured pass of 2048^2 7.5 us, with the energy 20, of a 19 us launch)
	__shared__ unsigned long long q_sums[2];
	if (cnt && threadIdx.x < 2) q_sums[threadIdx.x] = 0;
	if (cnt_eq) level(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::true_type{}, p.nlev - 1);
	// the tile itself into the other buffer; a print point: the up spins of what is stored
	unsigned long long ups = 0;
	for (int cg = wi; cg < 2 * Cc; cg += NW) {
		const int c = cg >= Cc, g = cg - c * Cc;
		uint64_t *to = (c ? dstp[1] : dstp[0]) + (size_t)(A + g) * gw;
		for (int w = lane; w < gw; w += 64) {
			const uint64_t v = lat[c * plane + (HG + g) * gw + w];
			to[w] = v;
			ups += (unsigned long long)__popcll(v);
		}
	}
	if (cnt) {
		__syncthreads(); // (q_sums is zero)
		ups = wave_sum(ups);
		const unsigned long long eq = wave_sum(eq_acc);
		if (lane == 0) {
			atomicAdd(&q_sums[0], ups);
			if (cnt_eq) atomicAdd(&q_sums[1], eq);
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			atomicAdd(cnt + (tile & 7), q_sums[0]);
			if (cnt_eq) atomicAdd(cnt_eq + (tile & 7), q_sums[1]);
		}
	}
#if defined(ISING_QUAD_TRACE)
	QTRC(5); // store
	__syncthreads();
	if (threadIdx.x == 0) { qtr[6] = (unsigned long long)(wall_clock64() - qt_start); qtr[7] = 1; qtr[8] = (unsigned long long)p.nlev; }
	__syncthreads();
	if (threadIdx.x < 16) atomicAdd(&g_qtrace[threadIdx.x], qtr[threadIdx.x]);
#endif
}

// One launch = the word pass of T sweeps on the masks the launch before made + the draws of the pass to come: workgroups [0, ntiles) take a tile each (they
// are dispatched first), the rest draw.  One stream, no events: nothing rests on two hardware queues running side by side (the first form of this path --
// draws on a second stream -- ran 1950 flips/ns at 2048^2 in a fresh process and 660 in one whose earlier contexts had created high-priority streams).
// (one and two items a wave: six waves per SIMD -- two workgroups of twelve waves per CU, a tile next to a drawing workgroup -- are worth 80 registers a lane;
// three items: 128, four waves per SIMD -- a tile of sixteen waves has its CU to itself; four: eight waves a workgroup at most)
#ifndef ISING_QUAD_WPE // waves per SIMD the one- and two-item instantiations are compiled for (A/B: 5 = 96 registers, workgroups of ten waves)
#define ISING_QUAD_WPE 6
#endif
template <int MAXI>
__global__ void __launch_bounds__(MAXI <= 3 ? 1024 : 512) __attribute__((amdgpu_waves_per_eu(MAXI <= 2 ? ISING_QUAD_WPE : (MAXI == 3 ? 4 : 2)))) quad_pass_k(const QuadPassParams p) {
	extern __shared__ __attribute__((aligned(16))) uint64_t q_lds[];
	const int lane = threadIdx.x & 63;
	const int wi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int NW = __builtin_amdgcn_readfirstlane((int)(blockDim.x >> 6));
	// Who does what: the tiles first (the dispatcher starts workgroups in index order; a word pass is a chain of short levels, the draws are throughput),
	// the drawing workgroups behind them.  (Measured and dropped: a tile at every n-th index among the drawing workgroups -- 4096^2 2157 against 2326 flips/ns.)
	const unsigned b = blockIdx.x;
	const int tile = b < (unsigned)p.ntiles ? (int)b : -1;
	const long long draw = (long long)b - p.ntiles;
	if (tile >= 0) quad_word_part<MAXI>(p.w, tile, wi, lane, NW, q_lds);
	else quad_draw_part(p.d, draw, wi, lane, NW, q_lds);
}

// ---- dense <-> quad, one wave per (row group, block); dense rows are gx * 32 words of 32 sites
__device__ __forceinline__ void dense_to_quad_body(const uint32_t *__restrict__ dense, uint64_t *__restrict__ quad, int gx, int NRG) {
	__shared__ uint32_t sh[4][128];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int j = lane >> 5, m = (lane >> 2) & 7, q = lane & 3;
	const int s = (q & 1) * 16 + 2 * m + (q >> 1);
	const long long n = (long long)blockIdx.x * 4 + wv;
	if (n >= (long long)NRG * gx) return;
	const int R = (int)(n / gx), bx = (int)(n - (long long)R * gx);
	for (int h = 0; h < 2; ++h) {
		const int idx = lane + 64 * h, r4 = idx >> 5, vv = idx & 31;
		sh[wv][idx] = dense[((size_t)4 * R + r4) * ((size_t)gx * 32) + 32 * bx + vv];
	}
	__builtin_amdgcn_wave_barrier();
	__threadfence_block();
	uint64_t w = 0;
#pragma unroll 8
	for (int l = 0; l < 64; ++l) w |= (uint64_t)((sh[wv][(l >> 4) * 32 + 16 * j + (l & 15)] >> s) & 1u) << l;
	quad[n * 64 + lane] = w;
}

__global__ void __launch_bounds__(256) dense_to_quad_k(const uint32_t *__restrict__ dense, uint64_t *__restrict__ quad, int gx, int NRG) {
	dense_to_quad_body(dense, quad, gx, NRG);
}

// (also refreshes the dense layout's mirror rows -1 and Y: dense points at row 0 of an array with a row above and Y + 1 rows below)
__device__ __forceinline__ void quad_to_dense_body(const uint64_t *__restrict__ quad, uint32_t *__restrict__ dense, int gx, int NRG) {
	__shared__ uint64_t sh[4][64];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const long long n = (long long)blockIdx.x * 4 + wv;
	if (n >= (long long)NRG * gx) return;
	const int R = (int)(n / gx), bx = (int)(n - (long long)R * gx);
	sh[wv][lane] = quad[n * 64 + lane];
	__builtin_amdgcn_wave_barrier();
	__threadfence_block();
	const ptrdiff_t ld = (ptrdiff_t)gx * 32;
	const int Y = 4 * NRG;
	for (int h = 0; h < 2; ++h) {
		const int idx = lane + 64 * h, r4 = idx >> 5, vv = idx & 31;
		const int jj = vv >> 4, l = 16 * r4 + (vv & 15);
		uint32_t d = 0;
#pragma unroll
		for (int s = 0; s < 32; ++s) d |= (uint32_t)((sh[wv][qword_of_site(jj, s)] >> l) & 1ull) << s;
		const int row = 4 * R + r4;
		dense[(ptrdiff_t)row * ld + 32 * bx + vv] = d;
		if (row == 0) dense[(ptrdiff_t)Y * ld + 32 * bx + vv] = d;
		if (row == Y - 1) dense[-ld + 32 * bx + vv] = d;
	}
}
__global__ void __launch_bounds__(256) quad_to_dense_k(const uint64_t *__restrict__ quad, uint32_t *__restrict__ dense, int gx, int NRG) {
	quad_to_dense_body(quad, dense, gx, NRG);
}

// ... of a batch: blockIdx.y = 2 r + colour
template <bool TO_QUAD>
__global__ void __launch_bounds__(256) quad_convert_batch_k(const QuadRec *__restrict__ rep, int buffer, int gx, int NRG) {
	const QuadRec *rc = rep + (blockIdx.y >> 1);
	const int color = blockIdx.y & 1;
	uint64_t *quad = rc->quad + (size_t)(2 * buffer + color) * (size_t)gx * (size_t)NRG * 64;
	if (TO_QUAD) dense_to_quad_body(rc->dense[color], quad, gx, NRG);
	else quad_to_dense_body(quad, rc->dense[color], gx, NRG);
}

} // namespace

#if defined(ISING_QUAD_TRACE)
void quad_trace_dump() {
	unsigned long long h[16] = {};
	if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_qtrace), sizeof(h)) != hipSuccess || !h[7]) return;
	static const char *names[6] = {"tile load", "prefetch issue", "mask wait", "items", "barrier", "store"};
	fprintf(stderr, "quad word passes: %llu workgroups, %.1f levels each, %.2f us each (100 MHz clock)\n", h[7], (double)h[8] / h[7], (double)h[6] / h[7] / 100.0);
	for (int i = 0; i < 6; i++) fprintf(stderr, "  %-15s %6.2f us per workgroup  %5.1f %%\n", names[i], (double)h[i] / h[7] / 100.0, 100.0 * h[i] / h[6]);
	unsigned long long z[16] = {};
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_qtrace), z, sizeof(z));
}
#endif

size_t quad_pass_lds_bytes(const QuadWordParams &w, int waves) {
	const size_t NG = (size_t)w.C + 2 * (size_t)w.HG;
	return std::max<size_t>(2 * NG * (size_t)w.gx * 64 * sizeof(uint64_t), (size_t)waves * 256); // the tile + halo, both colours; a drawing workgroup's block constants
}

int quad_word_maxi(const QuadWordParams &p, int waves) {
	const int items = (p.C + 2 * p.HG) * p.gx, need = (items + waves - 1) / waves;
	for (int mi : {1, 2, 3, 4}) if (mi >= need) return (mi > 3 && waves > 8) ? 0 : mi; // (four items a wave: the registers of eight waves at most)
	return 0;
}

template <int MAXI>
static hipError_t launch_pass_t(const QuadPassParams &p, int waves, long long grid, size_t lds, hipStream_t stream) {
	// tiles past 64 KiB: the limit is raised per device (hipFuncSetAttribute applies to the current one) and remembered per device, under a lock -- contexts on
	// several GPUs, driven by several host threads, share this function
	if (lds > 64 * 1024) {
		static std::mutex mu;
		static size_t allowed[64] = {};
		int dev = 0;
		if (const hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
		std::lock_guard<std::mutex> lock(mu);
		if (dev < 0 || dev >= 64 || lds > allowed[dev]) {
			const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&quad_pass_k<MAXI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			if (e != hipSuccess) return e;
			if (dev >= 0 && dev < 64) allowed[dev] = lds;
		}
	}
	hipLaunchKernelGGL(quad_pass_k<MAXI>, dim3((unsigned)grid), dim3((unsigned)waves * 64), lds, stream, p);
	return hipGetLastError();
}

// workgroups of `waves` waves a CU holds (hipOccupancyMaxActiveBlocksPerMultiprocessor, remembered per instantiation, size and LDS segment)
static int quad_pass_occupancy(int mi, int waves, size_t lds) {
	static std::mutex mu;
	static struct { int mi, waves; size_t lds; int n; } memo[16];
	static int nmemo = 0;
	std::lock_guard<std::mutex> lock(mu);
	for (int k = 0; k < nmemo; k++) if (memo[k].mi == mi && memo[k].waves == waves && memo[k].lds == lds) return memo[k].n;
	int n = 0;
	hipError_t e = hipErrorInvalidValue;
	switch (mi) {
	case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, quad_pass_k<1>, waves * 64, lds); break;
	case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, quad_pass_k<2>, waves * 64, lds); break;
	case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, quad_pass_k<3>, waves * 64, lds); break;
	case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, quad_pass_k<4>, waves * 64, lds); break;
	default: break;
	}
	if (e != hipSuccess || n < 1) { // (the rule of thumb of round 5)
		(void)hipGetLastError();
		const int regs = std::max(1, ((mi <= 1 ? 8 : (mi <= 2 ? 6 : (mi == 3 ? 4 : 3))) * 4) / waves);
		n = std::max(1, std::min<int>(regs, (int)((160 * 1024) / std::max<size_t>(lds, 1))));
	}
	if (nmemo < 16) memo[nmemo++] = {mi, waves, lds, n};
	return n;
}

// `p.w.nlev` = 0: draws only (the first launch of a call); `p.d.nlev` = 0: words only (its last)
hipError_t launch_quad_pass(QuadPassParams &p, int waves, hipStream_t stream) {
	const int mi0 = quad_word_maxi(p.w, waves);
	const int nrep = p.w.rep ? p.w.nrep : (p.d.rep ? p.d.nrep : 1);
	p.w.tiles_per_lat = (p.w.NRG + p.w.C - 1) / p.w.C;
	const long long all_tiles = (long long)p.w.tiles_per_lat * nrep;
	if (all_tiles > 0x3fffffffLL) return hipErrorInvalidValue;
	p.ntiles = p.w.nlev > 0 ? (int)all_tiles : 0;
	// workgroup slots of the chip: what the runtime says a CU holds of this instantiation at this workgroup size and LDS segment (registers -- eight waves per SIMD
	// at one item a wave, six at two, four at three -- and the tiles' LDS segment, which the drawing workgroups of the grid reserve as well: one launch, one size)
	const int per_cu = std::max(1, quad_pass_occupancy(mi0, waves, quad_pass_lds_bytes(p.w, waves)));
	const int cap = std::max(4, p.cus * per_cu) & ~3;
	long long draw_wgs = 0;
	if (p.d.nlev > 0) {
		const long long items = (long long)p.d.NRG * p.d.gx * p.d.nlev * nrep;
		// Few tiles (up to a quarter of the slots: a small lattice, its word pass the launch's critical path): as many drawing workgroups as find room next to
		// them, every wave an equal share of the draws -- they end together (2048^2 1705 -> 1792 flips/ns).  More tiles: drawing workgroups of one item a wave
		// that come and go, the dispatcher balancing (drawing workgroups that stay for the whole launch keep the slots the later tiles need: 6144^2 1508 against 2343).
		if (p.ntiles > cap / 4) {
			draw_wgs = (items + waves - 1) / waves;
			p.d.nwaves = (int)std::min<long long>(items, 0x7fffffff);
		} else { // (a quarter of an item a wave at least)
			draw_wgs = std::max<long long>(1, std::min<long long>(std::max(cap - p.ntiles, cap / 2), (4 * items + waves - 1) / waves));
			p.d.nwaves = (int)(draw_wgs * waves);
		}
	}
	const long long grid = p.ntiles + draw_wgs;
	if (grid <= 0) return hipSuccess;
	if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
	const int mi = quad_word_maxi(p.w, waves);
	const size_t lds = quad_pass_lds_bytes(p.w, waves);
	switch (mi) {
	case 1: return launch_pass_t<1>(p, waves, grid, lds, stream);
	case 2: return launch_pass_t<2>(p, waves, grid, lds, stream);
	case 3: return launch_pass_t<3>(p, waves, grid, lds, stream);
	case 4: return launch_pass_t<4>(p, waves, grid, lds, stream);
	default: return hipErrorInvalidValue;
	}
}

hipError_t launch_dense_to_quad(const uint32_t *dense, uint64_t *quad, int gx, int NRG, hipStream_t stream) {
	const long long n = (long long)NRG * gx;
	hipLaunchKernelGGL(dense_to_quad_k, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, dense, quad, gx, NRG);
	return hipGetLastError();
}

hipError_t launch_quad_convert_batch(const QuadRec *rep, int nrep, int buffer, bool to_quad, int gx, int NRG, hipStream_t stream) {
	const long long n = (long long)NRG * gx;
	if (nrep < 1 || 2 * nrep > 65535) return hipErrorInvalidValue;
	const dim3 grid((unsigned)((n + 3) / 4), (unsigned)(2 * nrep));
	if (to_quad) hipLaunchKernelGGL(quad_convert_batch_k<true>, grid, dim3(256), 0, stream, rep, buffer, gx, NRG);
	else hipLaunchKernelGGL(quad_convert_batch_k<false>, grid, dim3(256), 0, stream, rep, buffer, gx, NRG);
	return hipGetLastError();
}

hipError_t launch_quad_to_dense(const uint64_t *quad, uint32_t *dense, int gx, int NRG, hipStream_t stream) {
	const long long n = (long long)NRG * gx;
	hipLaunchKernelGGL(quad_to_dense_k, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, quad, dense, gx, NRG);
	return hipGetLastError();
}

} // namespace ising
