// ising_dense.hip -- the same hot path on a dense device layout: 1 bit per spin instead of the reference's 4.
//
// Why a second layout.  The packed-nibble kernel (ising_kernels.hip) is VALU-bound AND runs against the board power
// limit: with 4.2 TB/s of HBM traffic next to a saturated vector ALU the package sits at its cap and DVFS drops the
// shader clock from 2.40 to 2.21 GHz.  The reference needs 4 bits per spin only so that four
// neighbour words can be added nibble-wise; on a bit-sliced adder (a handful of 3-input bit operations per 32 sites)
// the sum costs no more, and the lattice shrinks 4x: 0.375 instead of 1.5 bytes of HBM traffic per flip.  What comes
// out is identical: the C-ABI converts to the reference's packed layout at its boundary (read/write/dump), and the
// Philox stream mapping is untouched because one reference 128-bit vector (32 spins) is exactly one 32-bit word here:
//   bit k of word c  <->  nibble k of word x (k < 16) / nibble k-16 of word y (k >= 16) of reference vector c.
// Throughput is still reported against the reference's 1.5 B/flip accounting.
//
// Not in this layout (the nibble layout is selected automatically): -J couplings.
#include "ising_device.hpp"

namespace ising {
namespace {

// The four draws of Philox block m of a vector belong to spins 2m, 16+2m, 2m+1, 17+2m.  Each draw sets
// its spin's bit in c3 / c4 when it is below the threshold for 3 / 4 aligned neighbours.  v_cmpx writes the lane
// predicate into EXEC and a 2-cycle VOP2 OR with a literal runs under it (see accept_rank in ising_kernels.hip).
template <int M>
__device__ __forceinline__ void accept_bits(uint32_t &c3, uint32_t &c4, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3,
                                            uint32_t n3, uint32_t n4) {
	constexpr uint32_t K0 = 1u << (2 * M), K1 = 1u << (16 + 2 * M), K2 = 1u << (2 * M + 1), K3 = 1u << (17 + 2 * M);
	unsigned long long saved;
	asm volatile(
	    "s_mov_b64 %[sv], exec\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n3], %[o0]\n\t"
	    "v_or_b32_e32 %[c3], %[k0], %[c3]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n4], %[o0]\n\t"
	    "v_or_b32_e32 %[c4], %[k0], %[c4]\n\t"
	    "s_mov_b64 exec, %[sv]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n3], %[o1]\n\t"
	    "v_or_b32_e32 %[c3], %[k1], %[c3]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n4], %[o1]\n\t"
	    "v_or_b32_e32 %[c4], %[k1], %[c4]\n\t"
	    "s_mov_b64 exec, %[sv]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n3], %[o2]\n\t"
	    "v_or_b32_e32 %[c3], %[k2], %[c3]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n4], %[o2]\n\t"
	    "v_or_b32_e32 %[c4], %[k2], %[c4]\n\t"
	    "s_mov_b64 exec, %[sv]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n3], %[o3]\n\t"
	    "v_or_b32_e32 %[c3], %[k3], %[c3]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n4], %[o3]\n\t"
	    "v_or_b32_e32 %[c4], %[k3], %[c4]\n\t"
	    "s_mov_b64 exec, %[sv]"
	    : [c3] "+v"(c3), [c4] "+v"(c4), [sv] "=&s"(saved)
	    : [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2), [o3] "v"(o3), [n3] "s"(n3), [n4] "s"(n4), [k0] "n"(K0), [k1] "n"(K1),
	      [k2] "n"(K2), [k3] "n"(K3)
	    : "vcc");
}

// Bit-sliced count of up neighbours of 32 spins: (n2 n1 n0) = up + down + centre + side, each a 0/1 plane.
// USEJ: J = coupling planes {x: right, y: left, z: down, w: up} of the destination sites; a set bit flips that
// neighbour's contribution (optimized/main.cu:575-618).
template <bool USEJ = false>
__device__ __forceinline__ void neighbour_planes(uint32_t up, uint32_t ct, uint32_t dw, uint32_t side_word, bool back,
                                                 uint32_t &n0, uint32_t &n1, uint32_t &n2, const uint4 &J = uint4()) {
	// the side neighbour of spin k is spin k-1 (back) or k+1 of the same row; the bit that falls off the word comes from
	// the adjacent word (optimized/main.cu:546-573 does the same with nibbles)
	uint32_t sd = back ? __builtin_amdgcn_alignbit(ct, side_word, 31) : __builtin_amdgcn_alignbit(side_word, ct, 1);
	if (USEJ) {
		// the same-index word holds the right neighbours when `back` (the shifted one the left), and vice versa
		up ^= J.w; dw ^= J.z;
		ct ^= back ? J.x : J.y;
		sd ^= back ? J.y : J.x;
	}
	const uint32_t x = up ^ dw, y = up & dw, z = ct ^ sd, w = ct & sd;
	const uint32_t c1 = x & z;
	n0 = x ^ z;
	n1 = y ^ w ^ c1;
	n2 = (y & w) | (c1 & (y ^ w));
}

// The integer-threshold forms go straight from the four neighbour words to the flips (flips32, ising_device.hpp).
template <bool USEJ = false>
__device__ __forceinline__ uint32_t word_flips(uint32_t me, uint32_t up, uint32_t ct, uint32_t dw, uint32_t side_word, bool back,
                                               uint32_t c3, uint32_t c4, const uint4 &J = uint4()) {
	uint32_t sd = back ? __builtin_amdgcn_alignbit(ct, side_word, 31) : __builtin_amdgcn_alignbit(side_word, ct, 1);
	if (USEJ) { // as in neighbour_planes
		up ^= J.w; dw ^= J.z;
		ct ^= back ? J.x : J.y;
		sd ^= back ? J.y : J.x;
	}
	return flips32(me, up, ct, dw, sd, c3, c4);
}

// cuRAND's curand_uniform: x*2^-32 + 2^-33 in FP32, one rounding (the product is exact).
__device__ __forceinline__ float u01(uint32_t x) {
	return __fmaf_rn(__uint2float_rn(x), 0x1p-32f, 0x1p-33f);
}

// ---------------------------------------------------------------------------------------------- update
// GENERIC = false: integer thresholds (needs table >= 1 for <= 2 aligned neighbours).  GENERIC = true: the reference's
// literal per-site FP32 compare against exp_h[spin][n] (optimized/main.cu:637-660) for temperatures that do not admit
// the integer form (T <= 0, saturated tables).
// MODE 0: integer thresholds, v_cmpx accept.  MODE 1 (GENERIC): FP32 table.  (A third form -- integer thresholds through a
// 64 KiB LDS table indexed by the top 16 bits of a draw -- measured equal or -2 % in round 1 and was removed in round 4.)
template <int MODE, bool SUBL = false, bool USEJ = false>
__global__ void __launch_bounds__(THREADS) dense_update_k(const UpdateParams p) {
	constexpr bool GENERIC = MODE == 1;
	__shared__ float sh_tab[10];
	if (GENERIC) {
		if (threadIdx.x < 10) sh_tab[threadIdx.x] = p.tab[threadIdx.x];
		__syncthreads();
	}
	const int tx = threadIdx.x & (GROUP - 1);
	const int unit = blockIdx.x * (THREADS / GROUP) + (threadIdx.x >> 4);
	if (unit >= p.nunits) return;
	const int rng = unit >= p.nunits0;
	const int u = unit - (rng ? p.nunits0 : 0);
	const int sidx = u / p.gx;
	const int bx = u - sidx * p.gx;
	const int r0 = p.row_lo[rng] + sidx * p.H;
	const int nrows = min(p.H, p.row_hi[rng] - r0);
	const int wpr = p.gx * 32; // 32-bit words per colour row; word index == reference vector index
	const int col0 = bx * 32 + tx;
	// adjacent words that supply the side carry bit, as word offsets from this lane's own word (periodic in the row)
	// (with sub-lattices, --xsl, the row is periodic every slV words instead; rows every slY, see `seam` below)
	const int slV = SUBL ? p.slV : wpr;
	const int offL0 = (col0 % slV) == 0 ? slV - 1 : -1, offL1 = GROUP - 1;
	const int offR0 = 1, offR1 = ((col0 + GROUP + 1) % slV) == 0 ? GROUP + 1 - slV : GROUP + 1;
	const int slY = SUBL ? p.slY : 0; // SUBL is a template switch so the common case carries none of the seam logic
	const int r0_in_sl = slY ? r0 % slY : 1;
	int seam = slY ? slY - r0_in_sl : 0x7fffffff; // rows left in the current sub-lattice

	const uint32_t *pc = reinterpret_cast<const uint32_t *>(p.src) + ((ptrdiff_t)r0 * wpr + col0);
	uint32_t *pm = reinterpret_cast<uint32_t *>(p.dst) + ((ptrdiff_t)r0 * wpr + col0);
	// -J: four coupling bit-planes per destination word, {right, left, down, up} (ham_planes_k)
	const uint4 *pj = USEJ ? reinterpret_cast<const uint4 *>(p.jdst) + ((ptrdiff_t)r0 * wpr + col0) : nullptr;
	const ptrdiff_t mir0 = (ptrdiff_t)(p.mir0_bytes / 4), mirL = (ptrdiff_t)(p.mirL_bytes / 4); // see ballot_update_k

	const uint32_t k2y = p.seed_hi + 2u * PHILOX_W1;
	const uint32_t cx_base = 16u * (2u * p.it + p.color);
	// The draw-block counter 16(2 it + colour) + b is 64 bits wide in cuRAND; its high word (non-zero from iteration
	// 2^27 on) enters round 1 next to key word 0, so it folds into the seed operand of the per-row setup.
	const uint32_t seed_lo_cy = p.seed_lo ^ (uint32_t)((2ull * p.it + p.color) >> 28);

	const ptrdiff_t uo = (slY && r0_in_sl == 0) ? (ptrdiff_t)(slY - 1) * wpr : -(ptrdiff_t)wpr; // loadTile wrap, :414
	uint32_t up0 = pc[uo], up1 = pc[uo + GROUP];
	uint32_t ct0 = pc[0], ct1 = pc[GROUP];

	for (int r = 0; r < nrows; ++r) {
		const int lr = r0 + r;
		const uint32_t grow = p.row_base + (uint32_t)lr;
		const bool back = (p.color == 0) ? !(grow & 1u) : (grow & 1u); // readBack, optimized/main.cu:542
		const bool sl_last = SUBL && seam == 1; // last row of its sub-lattice: the row below is the sub-lattice's first row (:422)
		const ptrdiff_t dwo = sl_last ? (ptrdiff_t)(1 - slY) * wpr : (ptrdiff_t)wpr;
		const uint32_t dw0 = pc[dwo], dw1 = pc[dwo + GROUP];
		const uint32_t side0 = pc[back ? offL0 : offR0];
		const uint32_t side1 = pc[back ? offL1 : offR1];
		uint32_t me0 = pm[0], me1 = pm[GROUP];
		uint4 j0 = uint4(), j1 = uint4();
		if (USEJ) { j0 = pj[0]; j1 = pj[GROUP]; pj += wpr; }

		const uint32_t tid = ((grow >> 4) * (uint32_t)p.gx + (uint32_t)bx) * 256u + (grow & 15u) * 16u + (uint32_t)tx;
		const PhiloxRow pr = philox_row_setup(tid, seed_lo_cy, k2y);

		if (!GENERIC) {
			uint32_t c3[2] = {0u, 0u}, c4[2] = {0u, 0u};
			static_for<16>([&](auto B) {
				constexpr int j = B.value >> 3, m = B.value & 7;
				uint32_t o[4];
				uint32_t cx = cx_base + (uint32_t)B.value;
				asm volatile("" : "+s"(cx)); // recompute the scalar rounds per row instead of spilling SGPRs (an LDS table, as in ising_ballot.hip, is slower here)
				philox_block(pr, cx, p.seed_lo, p.seed_hi, o[0], o[1], o[2], o[3]);
				accept_bits<m>(c3[j], c4[j], o[0], o[1], o[2], o[3], p.n3, p.n4);
			});
			me0 ^= word_flips<USEJ>(me0, up0, ct0, dw0, side0, back, c3[0], c4[0], j0);
			me1 ^= word_flips<USEJ>(me1, up1, ct1, dw1, side1, back, c3[1], c4[1], j1);
		} else {
			uint32_t me[2] = {me0, me1}, n0[2], n1[2], n2[2], flip[2] = {0u, 0u};
			neighbour_planes<USEJ>(up0, ct0, dw0, side0, back, n0[0], n1[0], n2[0], j0);
			neighbour_planes<USEJ>(up1, ct1, dw1, side1, back, n0[1], n1[1], n2[1], j1);
#pragma unroll
			for (int j = 0; j < 2; ++j) {
#pragma unroll
				for (int m = 0; m < 8; ++m) {
					uint32_t o[4];
					philox_block(pr, cx_base + 8u * j + m, p.seed_lo, p.seed_hi, o[0], o[1], o[2], o[3]);
					const int bit[4] = {2 * m, 16 + 2 * m, 2 * m + 1, 17 + 2 * m};
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						const uint32_t sp = (me[j] >> bit[q]) & 1u;
						const uint32_t n = ((n0[j] >> bit[q]) & 1u) | (((n1[j] >> bit[q]) & 1u) << 1) | (((n2[j] >> bit[q]) & 1u) << 2);
						if (u01(o[q]) <= sh_tab[sp * 5 + n]) flip[j] |= 1u << bit[q];
					}
				}
			}
			me0 ^= flip[0];
			me1 ^= flip[1];
		}

		pm[0] = me0;
		pm[GROUP] = me1;
		if (p.wrap) { // the halo rows that mirror this colour's edge rows
			if (lr == 0) { pm[mir0] = me0; pm[mir0 + GROUP] = me1; }
			if (lr == p.Y - 1) { pm[mirL] = me0; pm[mirL + GROUP] = me1; }
		}
		pc += wpr;
		pm += wpr;
		if (sl_last) {
			// the next row opens a new sub-lattice: the register window does not slide across the seam
			seam = slY;
			if (r + 1 < nrows) {
				const ptrdiff_t uo2 = (ptrdiff_t)(slY - 1) * wpr;
				up0 = pc[uo2]; up1 = pc[uo2 + GROUP];
				ct0 = pc[0]; ct1 = pc[GROUP];
			}
		} else {
			--seam;
			up0 = ct0; up1 = ct1;
			ct0 = dw0; ct1 = dw1;
		}
	}
}

// ---------------------------------------------------------------------------------------------- tiles: many sweeps per launch
// Lattices of a few million spins finish a colour half-sweep in about a microsecond of arithmetic, and a launch per colour costs
// several (kernel boundary, ramp, first loads): 2048^2 ran at 9.5 us per sweep, 4096^2 at 12.  No grid barrier is cheaper than that
// boundary, so this form removes the exchanges instead: a draw depends on (seed, site, iteration) only, hence a workgroup that holds
// its tile plus a halo can repeat its neighbours' updates of the halo bit for bit -- after h colour half-sweeps everything further
// than h rows / h sites from the edge of what was loaded is still exact.  2 ns halo rows and one halo word (32 sites >= 2 ns for
// ns <= 16) per side buy ns whole sweeps without a word from anybody else; the region shrinks by a row per half-sweep and ends on
// the tile.  The launch reads one buffer and writes the other (the neighbours' halo loads race with nothing).
// An item = one 32-site word of one row = 8 Philox blocks of one reference thread (tid from the row and the word's column,
// counter word 16 (2 it + colour) + 8 j + m: SURVEY 8a-R2), dealt to the threads of the workgroup in turn.
// Measured (tools/tile_probe.py, profiles/tile_probe_r04.txt, rocprof_r04_tiles_*.txt): 2048^2 427 -> 876 flips/ns, 4096^2 1362 -> 1607; the launches run back to
// back (19.8 us per 4 sweeps at 2048^2, ~7 of them fixed), the vector ALU is busy 59 % / 86 % of a wave's life at 2048^2 / 4096^2.  Tried and dropped: the blocks of
// an item 2 or 8 at a time instead of 4 (+-2 %), only the (ns + 1) / 2 blocks of a halo word that can reach the tile as items of their own (-2 .. -6 %: a second class
// of items costs more in uneven waves than it saves).
template <int NT>
__global__ void __launch_bounds__(NT) dense_tile_k(const TileParams p) {
	extern __shared__ uint32_t lds[];
	const int wpr = p.gx * 32;
	const int TW = p.TWI + 2;        // words of a tile row: halo word, TWI words, halo word
	const int HR = 2 * p.ns;         // halo rows on either side
	const int TRR = p.TR + 2 * HR;   // rows held
	const int plane = TRR * TW;
	uint32_t *tile[2] = {lds, lds + plane};
	PhiloxBlockConst *ktab = reinterpret_cast<PhiloxBlockConst *>(lds + 2 * plane); // [2 ns half-sweeps][16 blocks]
	const uint32_t inv_tw = 0xFFFFFFFFu / (uint32_t)TW + 1u; // i / TW = hi(i * inv_tw) for the i that occur (< 2^20)
	const int ntx = wpr / p.TWI;
	int t = (int)blockIdx.x;
	if (p.xcd_rows > 0) t = (t & 7) * p.xcd_rows + (t >> 3); // workgroup b runs on XCD b % 8 (observed, for speed only): every XCD a band of tiles, the same every launch
	const int tcy = t / ntx;
	const int tcx = t - tcy * ntx;
	const int row0 = tcy * p.TR - HR; // global row of tile row 0 (negative / beyond Y: periodic)
	const int col0 = tcx * p.TWI - 1; // global word column of tile column 0

	for (int i = threadIdx.x; i < 2 * p.ns * 16; i += NT) {
		const uint32_t hs = (uint32_t)i >> 4, b = (uint32_t)i & 15u;
		ktab[i] = philox_block_const(16u * (2u * p.it + hs) + b, p.seed_lo, p.seed_hi); // 2 (it + hs / 2) + (hs & 1) = 2 it + hs
	}
	for (int i0 = threadIdx.x; i0 < 2 * plane; i0 += 4 * NT) { // four loads in flight per thread (the words come from other XCDs' launches: memory latency)
		uint32_t v[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int i = i0 + k * NT;
			if (i < 2 * plane) {
				const int c = i >= plane, rem = i - (c ? plane : 0);
				const int r = (int)__umulhi((uint32_t)rem, inv_tw), w = rem - r * TW;
				int g = row0 + r, gc = col0 + w;
				g += g < 0 ? p.Y : 0; g -= g >= p.Y ? p.Y : 0;
				gc += gc < 0 ? wpr : 0; gc -= gc >= wpr ? wpr : 0;
				v[k] = p.src[c][(size_t)g * wpr + gc];
			}
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int i = i0 + k * NT;
			if (i < 2 * plane) lds[i] = v[k]; // (tile[1] follows tile[0])
		}
	}
	const uint32_t k2y = p.seed_hi + 2u * PHILOX_W1;
	for (int h = 0; h < 2 * p.ns; ++h) {
		__syncthreads();
		const int color = h & 1;
		const unsigned long long ctr = 2ull * p.it + (unsigned)h; // counter word / 16 (its high part: see dense_update_k)
		const uint32_t seed_lo_cy = p.seed_lo ^ (uint32_t)(ctr >> 28);
		const uint32_t *src = tile[1 - color];
		uint32_t *dst = tile[color];
		const PhiloxBlockConst *kt = ktab + 16 * h;
		const int nitems = (TRR - 2 - 2 * h) * TW; // rows [h + 1, TRR - 1 - h)
		for (int i = threadIdx.x; i < nitems; i += NT) {
			const int rr = (int)__umulhi((uint32_t)i, inv_tw), w = i - rr * TW, r = rr + h + 1;
			int g = row0 + r, gc = col0 + w;
			g += g < 0 ? p.Y : 0; g -= g >= p.Y ? p.Y : 0;
			gc += gc < 0 ? wpr : 0; gc -= gc >= wpr ? wpr : 0;
			const uint32_t grow = (uint32_t)g;
			const uint32_t tid = ((grow >> 4) * (uint32_t)p.gx + ((uint32_t)gc >> 5)) * 256u + (grow & 15u) * 16u + ((uint32_t)gc & 15u);
			const int j = (gc >> 4) & 1;
			const bool back = (color == 0) ? !(grow & 1u) : (grow & 1u); // readBack, optimized/main.cu:542
			const PhiloxRow pr = philox_row_setup(tid, seed_lo_cy, k2y);
			const int o = r * TW + w;
			const uint32_t up = src[o - TW], ct = src[o], dw = src[o + TW];
			const int ws = back ? w - 1 : w + 1; // beyond the halo word: whatever (it cannot reach the tile in 2 ns half-sweeps)
			const uint32_t side = (ws >= 0 && ws < TW) ? src[o + (ws - w)] : 0u;
			uint32_t me = dst[o];
			uint32_t c3 = 0u, c4 = 0u;
			// (four blocks at a time: their ten-round chains are independent, and a workgroup has few waves per SIMD to hide them behind)
			static_for<2>([&](auto Q) {
				uint32_t d[4][4];
				static_for<4>([&](auto M) { philox_block_pre(pr, kt[8 * j + 4 * Q.value + M.value], p.seed_lo, p.seed_hi, d[M.value][0], d[M.value][1], d[M.value][2], d[M.value][3]); });
				static_for<4>([&](auto M) { accept_bits<4 * Q.value + M.value>(c3, c4, d[M.value][0], d[M.value][1], d[M.value][2], d[M.value][3], p.n3, p.n4); });
			});
			me ^= word_flips(me, up, ct, dw, side, back, c3, c4);
			dst[o] = me;
		}
	}
	__syncthreads();
	unsigned ups = 0;
	for (int i = threadIdx.x; i < 2 * p.TR * p.TWI; i += NT) {
		const int c = i >= p.TR * p.TWI, rem = i - (c ? p.TR * p.TWI : 0);
		const int r = rem / p.TWI, w = rem - r * p.TWI;
		const int g = tcy * p.TR + r, gc = tcx * p.TWI + w;
		const uint32_t v = tile[c][(HR + r) * TW + 1 + w];
		uint32_t *q = p.dst[c] + (size_t)g * wpr + gc;
		*q = v;
		ups += (unsigned)__popc(v);
		if (g == 0) q[(size_t)p.Y * wpr] = v;                 // the mirror rows of a lone slab (launch_ranges: wrap)
		if (g == p.Y - 1) *(q - (ptrdiff_t)p.Y * wpr) = v;
	}
	if (p.cnt) { // a print point: getMagn_k's sum (optimized/main.cu:701-734) over the words this workgroup has just stored, one atomic per workgroup
		__shared__ unsigned wg_ups;
		if (threadIdx.x == 0) wg_ups = 0;
		__syncthreads();
		const unsigned s = (unsigned)wave_sum((unsigned long long)ups);
		if ((threadIdx.x & 63) == 0 && s) atomicAdd(&wg_ups, s);
		__syncthreads();
		if (threadIdx.x == 0) atomicAdd(p.cnt, (unsigned long long)wg_ups);
	}
}

// ---------------------------------------------------------------------------------------------- init
__global__ void __launch_bounds__(THREADS) dense_init_k(const InitParams p) {
	const int tx = threadIdx.x & (GROUP - 1);
	const long long unit_ll = flat_block() * GROUPS_PER_BLOCK + (threadIdx.x >> 4); // (row, bx)
	if (unit_ll >= (long long)p.gx * p.Y) return;
	const int unit = (int)unit_ll;
	const int lr = unit / p.gx;
	const int bx = unit - lr * p.gx;
	const int wpr = p.gx * 32;
	const uint32_t grow = p.row_base + (uint32_t)lr;
	const uint32_t tid = ((grow >> 4) * (uint32_t)p.gx + (uint32_t)bx) * 256u + (grow & 15u) * 16u + (uint32_t)tx;
	const PhiloxRow pr = philox_row_setup(tid, p.seed_lo, p.seed_hi + 2u * PHILOX_W1);
	const uint32_t cx_base = 16u * p.color; // it = 0, optimized/main.cu:116
	uint32_t *row = reinterpret_cast<uint32_t *>(p.dst) + (size_t)lr * wpr;
#pragma unroll
	for (int j = 0; j < 2; ++j) {
		uint32_t v = 0;
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			uint32_t o[4];
			philox_block(pr, cx_base + 8u * j + m, p.seed_lo, p.seed_hi, o[0], o[1], o[2], o[3]);
			// curand_uniform(x) < 0.5f  <=>  x < thr_half   (optimized/main.cu:133,:136)
			if (o[0] < p.thr_half) v |= 1u << (2 * m);
			if (o[1] < p.thr_half) v |= 1u << (16 + 2 * m);
			if (o[2] < p.thr_half) v |= 1u << (2 * m + 1);
			if (o[3] < p.thr_half) v |= 1u << (17 + 2 * m);
		}
		const int col = bx * 32 + tx + j * GROUP;
		row[col] = v;
		if (p.wrap) {
			if (lr == 0) row[(ptrdiff_t)p.Y * wpr + col] = v;
			if (lr == p.Y - 1) row[-(ptrdiff_t)p.Y * wpr + col] = v;
		}
	}
}

// ---------------------------------------------------------------------------------------------- bond sum
__global__ void __launch_bounds__(THREADS) dense_bond_equal_k(const BondParams p) {
	__shared__ unsigned long long part[THREADS / 64];
	const int wpr = p.gx * 32;
	const size_t total = (size_t)wpr * p.Y;
	const uint32_t *white = reinterpret_cast<const uint32_t *>(p.white);
	const uint32_t *black = reinterpret_cast<const uint32_t *>(p.black);
	unsigned long long acc = 0;
	for (size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * THREADS) {
		const int lr = (int)(i / wpr);
		const int col = (int)(i - (size_t)lr * wpr);
		const bool back = !((p.row_base + (uint32_t)lr) & 1u); // black sites
		const uint32_t *pc = white + (ptrdiff_t)lr * wpr;
		const int slV = p.slV, slY = p.slY;
		const int colS = back ? ((col % slV) == 0 ? col + slV - 1 : col - 1) : (((col + 1) % slV) == 0 ? col + 1 - slV : col + 1);
		const ptrdiff_t uo = (slY && (lr % slY) == 0) ? (ptrdiff_t)(slY - 1) * wpr : -(ptrdiff_t)wpr;
		const ptrdiff_t dwo = (slY && ((lr + 1) % slY) == 0) ? (ptrdiff_t)(1 - slY) * wpr : (ptrdiff_t)wpr;
		uint32_t n0, n1, n2;
		neighbour_planes(pc[col + uo], pc[col], pc[col + dwo], pc[colS], back, n0, n1, n2);
		const uint32_t me = black[i];
		// aligned neighbours a = n (up spin) or 4 - n (down spin), as planes a2 a1 a0
		const uint32_t a0 = n0;
		const uint32_t a1 = (me & n1) | (~me & (n0 ^ n1) & ~n2);
		const uint32_t a2 = (me & n2) | (~me & ~(n0 | n1 | n2));
		acc += (unsigned)__popc(a0) + 2u * (unsigned)__popc(a1) + 4u * (unsigned)__popc(a2);
	}
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long t = 0;
#pragma unroll
		for (int i = 0; i < THREADS / 64; ++i) t += part[i];
		atomicAdd(p.acc, t);
	}
}

// ---------------------------------------------------------------------------------------------- correlations input
// 32 spins -> the even bit positions of a 64-bit word
__device__ __forceinline__ unsigned long long spread32(unsigned long long x) {
	x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
	x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
	x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
	x = (x | (x << 2)) & 0x3333333333333333ull;
	x = (x | (x << 1)) & 0x5555555555555555ull;
	return x;
}

// bits64[r][c] holds lattice columns 64c..64c+63 of slab row r (see pack_bits_k in ising_kernels.hip)
__global__ void __launch_bounds__(THREADS) dense_pack_bits_k(const uint32_t *__restrict__ black, const uint32_t *__restrict__ white,
                                                             int wpr, int Y, uint32_t row_base, unsigned long long *__restrict__ bits64) {
	const size_t total = (size_t)wpr * Y;
	for (size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * THREADS) {
		const uint32_t r = (uint32_t)(i / wpr);
		const unsigned long long b = spread32(black[i]), w = spread32(white[i]);
		bits64[i] = ((row_base + r) & 1u) ? (w | (b << 1)) : (b | (w << 1));
	}
}

// -J with the dense layout: the coupling arrays are generated in the reference's nibble form (ham_init_*_k) and then
// transposed in place, 16 bytes at a time: 32 nibbles <right, left, down, up> = bits 0..3 -> four 32-bit planes
// {x: right, y: left, z: down, w: up}, bit s = site s of the vector (the dense spin word's bit order).
__global__ void __launch_bounds__(THREADS) ham_planes_k(uint4 *__restrict__ ham, size_t nvec) {
	for (size_t v = (size_t)blockIdx.x * THREADS + threadIdx.x; v < nvec; v += (size_t)gridDim.x * THREADS) {
		const uint4 n = ham[v];
		const uint32_t w[4] = {n.x, n.y, n.z, n.w};
		uint32_t pl[4] = {0u, 0u, 0u, 0u};
#pragma unroll
		for (int k = 0; k < 4; ++k) {
#pragma unroll
			for (int b = 0; b < 4; ++b) {
				uint32_t t = (w[k] >> b) & 0x11111111u; // bit b of 8 nibbles, at positions 0, 4, ..., 28
				t = (t | (t >> 3)) & 0x03030303u;
				t = (t | (t >> 6)) & 0x000F000Fu;
				t = (t | (t >> 12)) & 0xFFu;
				pl[b] |= t << (8 * k);
			}
		}
		ham[v] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
	}
}

// ---- boundary format.  The C-ABI speaks the reference's packed layout (16 spins per 64-bit word, bit 0 of each nibble);
// one dense 32-bit word is one reference 128-bit vector: bit k -> nibble k of word x (k < 16) / nibble k-16 of word y.
__device__ __forceinline__ unsigned long long nibbles_of(uint32_t bits16) { // 16 bits -> 16 nibbles
	unsigned long long x = bits16 & 0xFFFFu;
	x = (x | (x << 24)) & 0x000000FF000000FFull;
	x = (x | (x << 12)) & 0x000F000F000F000Full;
	x = (x | (x << 6)) & 0x0303030303030303ull;
	x = (x | (x << 3)) & 0x1111111111111111ull;
	return x;
}
__device__ __forceinline__ uint32_t bits_of(unsigned long long x) { // bit 0 of each of 16 nibbles -> 16 bits
	x &= 0x1111111111111111ull;
	x = (x | (x >> 3)) & 0x0303030303030303ull;
	x = (x | (x >> 6)) & 0x000F000F000F000Full;
	x = (x | (x >> 12)) & 0x000000FF000000FFull;
	x = (x | (x >> 24)) & 0xFFFFull;
	return (uint32_t)x;
}

__global__ void __launch_bounds__(THREADS) dense_to_packed_k(const uint32_t *__restrict__ dense, ulonglong2 *__restrict__ packed, size_t nvec) {
	for (size_t v = (size_t)blockIdx.x * THREADS + threadIdx.x; v < nvec; v += (size_t)gridDim.x * THREADS) {
		const uint32_t d = dense[v];
		packed[v] = make_ulonglong2(nibbles_of(d), nibbles_of(d >> 16));
	}
}

__global__ void __launch_bounds__(THREADS) packed_to_dense_k(const ulonglong2 *__restrict__ packed, uint32_t *__restrict__ dense, size_t nvec) {
	for (size_t v = (size_t)blockIdx.x * THREADS + threadIdx.x; v < nvec; v += (size_t)gridDim.x * THREADS) {
		const ulonglong2 p = packed[v];
		dense[v] = bits_of(p.x) | (bits_of(p.y) << 16);
	}
}

} // namespace

// ---------------------------------------------------------------------------------------------- launchers
hipError_t launch_dense_to_packed(const uint32_t *dense, uint64_t *packed, size_t nvec, hipStream_t stream) {
	if (!nvec) return hipSuccess;
	size_t blocks = (nvec + THREADS - 1) / THREADS;
	if (blocks > 16384) blocks = 16384;
	hipLaunchKernelGGL(dense_to_packed_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, dense, reinterpret_cast<ulonglong2 *>(packed), nvec);
	return hipGetLastError();
}

hipError_t launch_packed_to_dense(const uint64_t *packed, uint32_t *dense, size_t nvec, hipStream_t stream) {
	if (!nvec) return hipSuccess;
	size_t blocks = (nvec + THREADS - 1) / THREADS;
	if (blocks > 16384) blocks = 16384;
	hipLaunchKernelGGL(packed_to_dense_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, reinterpret_cast<const ulonglong2 *>(packed), dense, nvec);
	return hipGetLastError();
}

hipError_t launch_dense_update(const UpdateParams &p, int mode, hipStream_t stream) {
	if (p.nunits <= 0) return hipSuccess;
	const dim3 g0((p.nunits + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK), b0(THREADS);
	const bool generic = mode == 1;
	if (p.jdst) { // -J couplings
		if (p.slY) {
			if (generic) hipLaunchKernelGGL((dense_update_k<1, true, true>), g0, b0, 0, stream, p);
			else         hipLaunchKernelGGL((dense_update_k<0, true, true>), g0, b0, 0, stream, p);
		} else {
			if (generic) hipLaunchKernelGGL((dense_update_k<1, false, true>), g0, b0, 0, stream, p);
			else         hipLaunchKernelGGL((dense_update_k<0, false, true>), g0, b0, 0, stream, p);
		}
	} else if (p.slY) { // sub-lattices
		if (generic) hipLaunchKernelGGL((dense_update_k<1, true>), g0, b0, 0, stream, p);
		else         hipLaunchKernelGGL((dense_update_k<0, true>), g0, b0, 0, stream, p);
	} else if (generic)   hipLaunchKernelGGL((dense_update_k<1, false>), g0, b0, 0, stream, p);
	else                  hipLaunchKernelGGL((dense_update_k<0, false>), g0, b0, 0, stream, p);
	return hipGetLastError();
}

size_t dense_tiles_lds_bytes(const TileParams &p) {
	return ((size_t)2 * (p.TR + 4 * p.ns) * (p.TWI + 2) + (size_t)2 * p.ns * 16 * 3) * sizeof(uint32_t);
}

hipError_t launch_dense_tiles(const TileParams &p, int threads, hipStream_t stream) {
	const dim3 g((unsigned)((p.gx * 32 / p.TWI) * (p.Y / p.TR)));
	const size_t lds = dense_tiles_lds_bytes(p);
	switch (threads) {
	case 256:  hipLaunchKernelGGL(dense_tile_k<256>, g, dim3(256), lds, stream, p); break;
	case 512:  hipLaunchKernelGGL(dense_tile_k<512>, g, dim3(512), lds, stream, p); break;
	case 1024: hipLaunchKernelGGL(dense_tile_k<1024>, g, dim3(1024), lds, stream, p); break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

// ising_swap_couplings: the two coupling arrays change places, 16 bytes per lane and step (whatever form they are in)
__global__ void __launch_bounds__(THREADS) swap_vectors_k(uint4 *__restrict__ a, uint4 *__restrict__ b, size_t nvec) {
	for (size_t v = (size_t)blockIdx.x * THREADS + threadIdx.x; v < nvec; v += (size_t)gridDim.x * THREADS) {
		const uint4 x = a[v], y = b[v];
		a[v] = y;
		b[v] = x;
	}
}

hipError_t launch_swap_vectors(uint64_t *a, uint64_t *b, size_t nvec, hipStream_t stream) {
	size_t blocks = (nvec + THREADS - 1) / THREADS;
	if (blocks > 8192) blocks = 8192;
	hipLaunchKernelGGL(swap_vectors_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, reinterpret_cast<uint4 *>(a), reinterpret_cast<uint4 *>(b), nvec);
	return hipGetLastError();
}

hipError_t launch_ham_planes(uint64_t *ham, size_t nvec, hipStream_t stream) {
	size_t blocks = (nvec + THREADS - 1) / THREADS;
	if (blocks > 8192) blocks = 8192;
	hipLaunchKernelGGL(ham_planes_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, reinterpret_cast<uint4 *>(ham), nvec);
	return hipGetLastError();
}

hipError_t launch_dense_init(const InitParams &p, hipStream_t stream) {
	const long long units = (long long)p.gx * p.Y;
	hipLaunchKernelGGL(dense_init_k, flat_grid((units + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK), dim3(THREADS), 0, stream, p);
	return hipGetLastError();
}

hipError_t launch_dense_bond_equal(const BondParams &p, hipStream_t stream) {
	const size_t total = (size_t)p.gx * 32 * p.Y;
	size_t blocks = (total + THREADS - 1) / THREADS;
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(dense_bond_equal_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, p);
	return hipGetLastError();
}

hipError_t launch_dense_pack_bits(const uint64_t *black, const uint64_t *white, int wpr, int Y, uint32_t row_base, uint32_t *bits,
                                  hipStream_t stream) {
	size_t blocks = ((size_t)wpr * Y + THREADS - 1) / THREADS;
	if (blocks > 8192) blocks = 8192;
	hipLaunchKernelGGL(dense_pack_bits_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, reinterpret_cast<const uint32_t *>(black),
	                   reinterpret_cast<const uint32_t *>(white), wpr, Y, row_base, reinterpret_cast<unsigned long long *>(bits));
	return hipGetLastError();
}

} // namespace ising
