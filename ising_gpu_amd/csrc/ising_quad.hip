// ising_quad.hip -- small lattices (round 5): every draw made ONCE, by the whole chip; the cheap part repeated instead.
//
// Below ~2^25 spins a colour half-sweep is a microsecond of arithmetic, and whatever exchanges rows between workgroups once
// per half-sweep (a launch boundary, a counter in memory) costs several.  The tile launches of round 4 (ising_dense.hip:
// dense_tile_k) buy S sweeps without an exchange by repeating the neighbours' updates of a halo -- draws included, x 1.5 of the
// part that is 94 % of the instructions.  But a draw depends on (seed, site, iteration) alone: the accept decisions of any
// number of sweeps can be made ahead by a kernel that knows nothing of the lattice, at the rate of the large lattices' draw
// phase, with every SIMD of the chip busy and no redundancy; what has to run in order -- the word phase, 45 of 720 vector
// instructions per 4096 sites -- can then afford the halo.
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
//   quad_draw_k  (level, row group, block) -> 1 KiB of accept masks (c3, c4 per word), scalar stores straight into the mask
//                buffer; no lattice, no barrier, no order.
//   quad_word_k  one workgroup per tile of C row groups x the whole width, both colours + HG halo row groups in LDS; 2 T
//                levels over a region that shrinks a row per level; masks prefetched three levels ahead by LDS-direct loads;
//                reads one lattice buffer, writes its tile to the other.
// ising_update.cpp (sweep_quad) runs the draws of batch k + 1 on a second stream next to the word passes of batch k.
#include "ising_device.hpp"
#include <cstdio>

namespace ising {
namespace {

constexpr uint64_t Q_LANE0 = 0x0001000100010001ull;  // tx = 0 of each row
constexpr uint64_t Q_LANE15 = 0x8000800080008000ull; // tx = 15
constexpr uint64_t Q_EVEN = 0x0000FFFF0000FFFFull;   // rows 4 R, 4 R + 2
constexpr int Q_DEPTH = 3;                           // levels of masks in flight per wave (ring slots)

__device__ __forceinline__ constexpr int qword(int j, int m, int q) { return 32 * j + 4 * m + q; }
__device__ __forceinline__ int qword_of_site(int j, int s) {
	const int w = s >> 4, z = s & 15;
	return qword(j, z >> 1, 2 * (z & 1) + w);
}

// The compares' results live in fixed scalar registers (inline asm cannot name halves of an SGPR tuple operand)
#define QSG(a, b) "s[84+" #a ":84+" #b "]"
#define Q_CLOB16 "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"

// ---- draws: unit = (level, chunk of `chunk` consecutive (row group, block) items), one wave each
// Two forms of the same kernel: at most four waves per SIMD (lattices up to 2^24 spins: a word pass's workgroups -- twelve waves and half the LDS -- find
// room on every CU the moment they are dispatched; 2048^2 1870 -> 2030 flips/ns, 4096 x 2048 2421 -> 2600) and as many as fit (larger lattices: the draws'
// own throughput counts; 4096 x 16384 2876 against 2767).
__device__ __forceinline__ void quad_draw_body(const QuadDrawParams &p);
__global__ void __launch_bounds__(256) quad_draw_k(const QuadDrawParams p) { quad_draw_body(p); }
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4))) quad_draw4_k(const QuadDrawParams p) { quad_draw_body(p); }
__device__ __forceinline__ void quad_draw_body(const QuadDrawParams &p) {
	const int lane = threadIdx.x & 63;
	const int wi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	__shared__ uint4 blk_const_all[4][16];
	uint4 *blk_const = blk_const_all[wi];
	const int NI = p.NRG * p.gx;
	const int upl = (NI + p.chunk - 1) / p.chunk;
	const long long unit = (long long)blockIdx.x * 4 + wi;
	if (unit >= (long long)upl * p.nlev) return;
	const int level = __builtin_amdgcn_readfirstlane((int)(unit / upl));
	const int ch = __builtin_amdgcn_readfirstlane((int)(unit - (long long)level * upl));
	const uint32_t color = (uint32_t)level & 1u, it = p.it + ((uint32_t)level >> 1);
	uint32_t seed_lo = p.seed_lo, seed_hi = p.seed_hi;
	const uint32_t cx_base = 16u * (2u * it + color);
	const uint32_t seed_lo_cy = seed_lo ^ (uint32_t)((2ull * it + color) >> 28); // counter word 1 enters round 1 next to the key (dense_update_k)
	const uint32_t k2y = seed_hi + 2u * PHILOX_W1;
	if (lane < 16) {
		const PhiloxBlockConst kc = philox_block_const(cx_base + (uint32_t)lane, seed_lo, seed_hi);
		blk_const[lane] = make_uint4(kc.s0, kc.s1, kc.s2, 0u);
	}
	__builtin_amdgcn_wave_barrier();
	__threadfence_block();
	const int n0 = ch * p.chunk, n1 = min(NI, n0 + p.chunk);
	uint32_t thr3 = p.n3, thr4 = p.n4;
	for (int n = n0; n < n1; ++n) {
		const int R = n / p.gx, bx = n - R * p.gx;
		const uint32_t tid = (((uint32_t)R >> 2) * (uint32_t)p.gx + (uint32_t)bx) * 256u + ((uint32_t)R & 3u) * 64u + (uint32_t)lane;
		const PhiloxRow pr = philox_row_setup(tid, seed_lo_cy, k2y);
#if defined(ISING_QUAD_DRAW_TEST) // measurement builds (wrong masks by design): 1 = every wave stores into one slot of its own, over and over
		const uint64_t *dst0 = p.masks + ((size_t)(blockIdx.x % 2048u) * 4 + wi) * 128;
#else
		const uint64_t *dst0 = p.masks + ((size_t)level * (size_t)NI + (size_t)n) * 128;
#endif
		const uint32_t d_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)dst0), d_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)dst0 >> 32));
		const uint64_t *dst = reinterpret_cast<const uint64_t *>(((uintptr_t)d_hi << 32) | d_lo);
		uint4 kc_next = blk_const[0];
		static_for<16>([&](auto B) {
			uint32_t o0, o1, o2, o3;
			const uint4 kc = kc_next;
			if (B.value < 15) kc_next = blk_const[B.value + 1];
			philox_block_pre(pr, PhiloxBlockConst{kc.x, kc.y, kc.z}, seed_lo, seed_hi, o0, o1, o2, o3);
			if (B.value < 15) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kc_next.x), "+v"(kc_next.y), "+v"(kc_next.z) :: "memory");
			const uint64_t *dstp = dst + 8 * B.value;
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
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// one LDS-direct load of 1 KiB: lane l's 16 bytes at `base` + 16 l land at LDS byte address `lds` + 16 l (tracked by vmcnt, unknown to the compiler)
__device__ __forceinline__ void mask_fetch(const uint64_t *base, uint32_t lds, int lane16) {
	uint32_t keep; // (m0 is the compiler's: handed back as found)
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(lds), "v"(lane16), "s"(base) : "memory");
}

__device__ __forceinline__ uint32_t lds_addr(const void *q) {
	return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)q;
}

// Measurement build (make variant NAME=qtrace DEFS=-DISING_QUAD_TRACE): wave 0 of every word workgroup clocks where its time goes; quad_trace_dump()
// prints the sums when the slab is destroyed.  Never in the product library.
#if defined(ISING_QUAD_TRACE)
__device__ unsigned long long g_qtrace[16];
#define QTRC(i) do { if (wi == 0) { const long long t_ = wall_clock64(); if (lane == 0) qtr[i] += (unsigned long long)(t_ - qt_last); qt_last = t_; } } while (0)
#else
#define QTRC(i) do {} while (0)
#endif

// ---- words: `nlev` levels (black first) of tile blockIdx.x.  Everything that indexes is wave-uniform and lives on the scalar unit; a wave works on
// its items two at a time (all LDS reads of both first, one exposed round trip per pair).
template <int MAXI>
__global__ void __launch_bounds__(1024) quad_word_k(const QuadWordParams p) {
	extern __shared__ __attribute__((aligned(16))) uint64_t q_lds[];
	const int lane = threadIdx.x & 63;
	const int wi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int NW = __builtin_amdgcn_readfirstlane((int)(blockDim.x >> 6));
	const int gx = p.gx, NRG = p.NRG, HG = p.HG;
	const int NG = p.C + 2 * HG;
	const int A = (int)blockIdx.x * p.C;
	const int Cc = min(p.C, NRG - A);
	const int gw = gx * 64;                 // words per row group
	const int plane = NG * gw;              // words per colour in LDS
	uint64_t *lat = q_lds;                                 // [2][NG][gx][64]
	uint64_t *ring = q_lds + 2 * plane;                    // [NW][Q_DEPTH][MAXI][128]
	uint64_t *dummy = ring + (size_t)NW * Q_DEPTH * MAXI * 128; // [NW][128]
#if defined(ISING_QUAD_TRACE)
	__shared__ unsigned long long qtr[16];
	if (threadIdx.x < 16) qtr[threadIdx.x] = 0;
	__syncthreads();
	long long qt_last = wall_clock64();
	const long long qt_start = qt_last;
#endif
	__builtin_amdgcn_s_setprio(3); // (the draws of the batches to come share the chip: a word pass is a chain of short levels, theirs is throughput)
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
	const uint32_t ring_w = lds_addr(ring) + (uint32_t)wi * (Q_DEPTH * MAXI * 1024), dummy_w = lds_addr(dummy) + (uint32_t)wi * 1024;
	const int lane16 = lane * 16;
	const uint32_t gw_b = (uint32_t)gw * 8, plane_b = (uint32_t)plane * 8;
	// A lone wave issues an instruction every four or five cycles whatever its kind, and a level is a few hundred of them: what indexes a level is computed
	// ONCE, by the lanes in parallel -- lane L of rb_tab[k] / mk_tab[k] = where item k of level L sits in a colour plane (bytes) / in the pass's masks (KiB),
	// ~0 where the level's active range does not reach it -- and the level loop picks its scalars up with v_readlane.
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
	// masks of level Lp, this wave's items: exactly MAXI loads (absent items land in the wave's dummy slot)
	auto prefetch = [&](int Lp, uint32_t slot) {
#pragma unroll
		for (int k = 0; k < MAXI; ++k) {
			const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)mk_tab[k], Lp & 63);
			if (idx != ABSENT && Lp < p.nlev) mask_fetch(p.masks + (size_t)idx * 128, slot + (uint32_t)k * 1024, lane16);
			else mask_fetch(p.masks, dummy_w, lane16);
		}
	};
	constexpr uint32_t SLOT_B = MAXI * 1024;
	for (int Lp = 0; Lp < Q_DEPTH - 1; ++Lp) prefetch(Lp, ring_w + (uint32_t)Lp * SLOT_B);
	// the tile and its halo row groups, both colours
	for (int cg = wi; cg < 2 * NG; cg += NW) { // (row group, colour) by wave, its gx x 64 words by lane
		const int c = cg >= NG, g = cg - c * NG;
		const uint64_t *from = p.src[c] + (size_t)wrapR(g) * gw;
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
	const uint64_t mAB = specB ? ~Q_LANE0 : ~0ull, mBB = specB ? Q_LANE0 : 0ull;
	const uint64_t mAF = specF ? ~Q_LANE15 : ~0ull, mBF = specF ? Q_LANE15 : 0ull;
	// rows whose side neighbour is site s - 1 (readBack, optimized/main.cu:542): the even rows of a black level, the odd rows of a white one
	const uint64_t m1c[2] = {mAB & Q_EVEN, mAB & ~Q_EVEN}, m2c[2] = {mBB & Q_EVEN, mBB & ~Q_EVEN};
	const uint64_t m3c[2] = {mAF & ~Q_EVEN, mAF & Q_EVEN}, m4c[2] = {mBF & ~Q_EVEN, mBF & Q_EVEN};
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

	// one level; COL = the colour it updates (levels alternate from black: the loop below is unrolled by two)
	auto level = [&](auto COL, int L, uint32_t slot, uint32_t slot_ahead) {
		constexpr int c = COL.value;
		prefetch(L + Q_DEPTH - 1, slot_ahead);
		QTRC(1); // prefetch issue
		asm volatile("s_waitcnt vmcnt(%0)" :: "n"((Q_DEPTH - 1) * MAXI) : "memory");
		QTRC(2); // wait for this level's masks
		const uint32_t S_w = lat_w + (c ? 0u : plane_b), D_w = lat_w + (c ? plane_b : 0u);
		const uint64_t m1 = m1c[c], m2 = m2c[c], m3 = m3c[c], m4 = m4c[c];
		static_for<(MAXI + 1) / 2>([&](auto KP) {
			constexpr int k0 = 2 * KP.value, k1 = k0 + 1 < MAXI ? k0 + 1 : k0; // (an odd MAXI's last item stands alone)
			constexpr int NK = k1 > k0 ? 2 : 1;
			constexpr int ks[2] = {k0, k1};
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
				const uint32_t mk = slot + (uint32_t)k * 1024 + (uint32_t)lane16;
				c3[t] = ld(mk);
				c4[t] = ld(mk + 8);
			}
			(void)ks;
#pragma unroll
			for (int t = 0; t < NK; ++t) {
				const uint64_t sd = ((x1B[t] << shB) & m1) | ((x2B[t] >> 15) & m2) | ((x1F[t] >> shF) & m3) | ((x2F[t] << 15) & m4);
				const uint64_t up = (ct[t] << 16) | (upc[t] >> 48), dw = (ct[t] >> 16) | (dnc[t] << 48);
				const uint32_t flo = flips32((uint32_t)me[t], (uint32_t)up, (uint32_t)ct[t], (uint32_t)dw, (uint32_t)sd, (uint32_t)c3[t], (uint32_t)c4[t]);
				const uint32_t fhi = flips32((uint32_t)(me[t] >> 32), (uint32_t)(up >> 32), (uint32_t)(ct[t] >> 32), (uint32_t)(dw >> 32), (uint32_t)(sd >> 32),
				                             (uint32_t)(c3[t] >> 32), (uint32_t)(c4[t] >> 32));
				if (on[t]) *(lds_p)(uintptr_t)dst[t] = me[t] ^ (((uint64_t)fhi << 32) | flo);
			}
		});
		QTRC(3); // items
		__syncthreads();
		QTRC(4); // barrier
	};
	// (ring slot of level L = L mod Q_DEPTH, kept as two running offsets)
	uint32_t s_now = 0, s_ahead = (Q_DEPTH - 1) * SLOT_B;
	auto advance = [&]() {
		s_now = s_now + SLOT_B == Q_DEPTH * SLOT_B ? 0u : s_now + SLOT_B;
		s_ahead = s_ahead + SLOT_B == Q_DEPTH * SLOT_B ? 0u : s_ahead + SLOT_B;
	};
	for (int L = 0; L < p.nlev; L += 2) { // (whole sweeps: nlev is even)
		level(std::integral_constant<int, 0>{}, L, ring_w + s_now, ring_w + s_ahead);
		advance();
		level(std::integral_constant<int, 1>{}, L + 1, ring_w + s_now, ring_w + s_ahead);
		advance();
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the dummy loads behind the last level)
	// the tile itself into the other buffer; a print point: the up spins of what is stored
	unsigned long long ups = 0;
	for (int cg = wi; cg < 2 * Cc; cg += NW) {
		const int c = cg >= Cc, g = cg - c * Cc;
		uint64_t *to = p.dst[c] + (size_t)(A + g) * gw;
		for (int w = lane; w < gw; w += 64) {
			const uint64_t v = lat[c * plane + (HG + g) * gw + w];
			to[w] = v;
			ups += (unsigned long long)__popcll(v);
		}
	}
	if (p.cnt) {
		ups = wave_sum(ups);
		if (lane == 0) atomicAdd(p.cnt, ups);
	}
#if defined(ISING_QUAD_TRACE)
	QTRC(5); // store
	__syncthreads();
	if (threadIdx.x == 0) { qtr[6] = (unsigned long long)(wall_clock64() - qt_start); qtr[7] = 1; qtr[8] = (unsigned long long)p.nlev; }
	__syncthreads();
	if (threadIdx.x < 16) atomicAdd(&g_qtrace[threadIdx.x], qtr[threadIdx.x]);
#endif
}

// ---- dense <-> quad, one wave per (row group, block); dense rows are gx * 32 words of 32 sites
__global__ void __launch_bounds__(256) dense_to_quad_k(const uint32_t *__restrict__ dense, uint64_t *__restrict__ quad, int gx, int NRG) {
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

// (also refreshes the dense layout's mirror rows -1 and Y: dense points at row 0 of an array with a row above and Y + 1 rows below)
__global__ void __launch_bounds__(256) quad_to_dense_k(const uint64_t *__restrict__ quad, uint32_t *__restrict__ dense, int gx, int NRG) {
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

hipError_t launch_quad_draw(const QuadDrawParams &p, hipStream_t stream) {
	const long long NI = (long long)p.NRG * p.gx, units = (NI + p.chunk - 1) / p.chunk * p.nlev;
	if (p.few_waves) hipLaunchKernelGGL(quad_draw4_k, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, stream, p);
	else hipLaunchKernelGGL(quad_draw_k, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, stream, p);
	return hipGetLastError();
}

size_t quad_word_lds_bytes(const QuadWordParams &p, int waves, int maxi) {
	const size_t NG = (size_t)p.C + 2 * (size_t)p.HG;
	return (2 * NG * (size_t)p.gx * 64 + (size_t)waves * Q_DEPTH * (size_t)maxi * 128 + (size_t)waves * 128) * sizeof(uint64_t);
}

int quad_word_maxi(const QuadWordParams &p, int waves) {
	const int items = (p.C + 2 * p.HG) * p.gx, need = (items + waves - 1) / waves;
	for (int mi : {1, 2, 3, 4, 6, 8, 12, 16}) if (mi >= need) return mi;
	return 0;
}

template <int MAXI>
static hipError_t launch_word_t(const QuadWordParams &p, int waves, size_t lds, hipStream_t stream) {
	static size_t allowed = 0; // (per instantiation; contexts of one process share a device class)
	if (lds > 64 * 1024 && lds > allowed) {
		const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&quad_word_k<MAXI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return e;
		allowed = lds;
	}
	const int tiles = (p.NRG + p.C - 1) / p.C;
	hipLaunchKernelGGL(quad_word_k<MAXI>, dim3((unsigned)tiles), dim3((unsigned)waves * 64), lds, stream, p);
	return hipGetLastError();
}

hipError_t launch_quad_word(const QuadWordParams &p, int waves, hipStream_t stream) {
	const int mi = quad_word_maxi(p, waves);
	const size_t lds = quad_word_lds_bytes(p, waves, mi);
	switch (mi) {
	case 1: return launch_word_t<1>(p, waves, lds, stream);
	case 2: return launch_word_t<2>(p, waves, lds, stream);
	case 3: return launch_word_t<3>(p, waves, lds, stream);
	case 4: return launch_word_t<4>(p, waves, lds, stream);
	case 6: return launch_word_t<6>(p, waves, lds, stream);
	case 8: return launch_word_t<8>(p, waves, lds, stream);
	case 12: return launch_word_t<12>(p, waves, lds, stream);
	case 16: return launch_word_t<16>(p, waves, lds, stream);
	default: return hipErrorInvalidValue;
	}
}

hipError_t launch_dense_to_quad(const uint32_t *dense, uint64_t *quad, int gx, int NRG, hipStream_t stream) {
	const long long n = (long long)NRG * gx;
	hipLaunchKernelGGL(dense_to_quad_k, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, dense, quad, gx, NRG);
	return hipGetLastError();
}

hipError_t launch_quad_to_dense(const uint64_t *quad, uint32_t *dense, int gx, int NRG, hipStream_t stream) {
	const long long n = (long long)NRG * gx;
	hipLaunchKernelGGL(quad_to_dense_k, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, quad, dense, gx, NRG);
	return hipGetLastError();
}

} // namespace ising
