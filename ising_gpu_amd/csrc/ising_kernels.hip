// ising_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the checkerboard-Metropolis hot path.
//
// What each kernel replaces in the reference (file:line relative to /root/reference):
//   update_k      <- spinUpdateV_2D_k + loadTile ......... optimized/main.cu:463-670, :380-461
//   init_k        <- latticeInit_k ........................ optimized/main.cu:92-151
//   popcount_k    <- getMagn_k + __block_sum .............. optimized/main.cu:701-734, :672-699
//   bond_equal_k  <- (none: the reference has no energy observable)
//
// Design (see DESIGN.md for the measurements behind it):
//   * Thread geometry.  A lane owns what a reference thread owns: the two 128-bit vectors (2 x 32 spins) at
//     vector columns c and c+16 of one row, so the Philox subsequence id ("tid") and the 64-draw order are the
//     reference's.  Sixteen lanes = one reference block-row (256 B contiguous per vector
//     load); a wave64 holds four such groups on consecutive 32-vector column groups, so its two loads cover
//     2 KiB of contiguous HBM per row.
//   * A/B variant MODE 2: accept ranks through a 64 KiB LDS table: the top 16 bits of a draw index a byte that holds
//     [x<N3]+[x<N4]; the two table rows that straddle a threshold hold 4 and send that dword pair (3 % of
//     wave-pairs) through exact compares.  One v_lshrrev + ds_read_u8 + v_lshl_or per site instead of two VOPC
//     compares and two masked adds: +8 % in an ALU-only build, but equal in the real kernel, which runs against the
//     board power limit; MODE 0 (v_cmpx) is the default.
//   * Row marching instead of an LDS tile.  Each lane walks H consecutive rows with the (up, centre, down)
//     source rows in registers, so the opposite-colour array is streamed from HBM once per half-sweep plus
//     2/H halo rows; the 4-bit side-neighbour carry comes from one extra dword load that hits the line the
//     neighbouring lane just fetched.  (The reference stages an 18x34 tile in shared memory; on CDNA4 that
//     would spend LDS bandwidth and a barrier per tile for data the register window already holds.)
//   * Philox4x32-10 with the first two rounds hoisted.  The counter of draw block b of thread t is
//     (16(2it+c)+b, 0, tid, 0): the low word is wave-uniform (scalar ALU), the third is constant over the 16
//     blocks of a row, so rounds 1-2 cost no vector multiplies per block (16 v_mad_u64_u32 instead of 20).
//   * Integer accept test.  curand_uniform(x) <= exp_table[s][n] is monotone in the raw draw x, so it is
//     replaced by x < N(a) with a = number of aligned neighbours; a <= 2 always flips (table >= 1), a = 3 and
//     a = 4 use the host-computed N3 > N4.  Each site adds r = [x<N3] + [x<N4] into a nibble of a per-dword
//     accumulator; flips for 8 sites are then resolved with a handful of nibble-parallel word operations:
//     spin up   (a = n):   flip <=> n - r <= 2;   spin down (a = 4-n): flip <=> n + r >= 2.
//     Bit-exactness of this rewrite against the FP32 form is what tests/test_gpu_parity.py establishes.
#include "ising_device.hpp"

namespace ising {
namespace {

// Accept-rank accumulation for the four draws of one Philox block.  For each draw x the nibble of its site gets
// r = [x < n3] + [x < n4] added (n4 <= n3).  Draws 0/2 belong to nibbles NIB, NIB+1 of the dword rx (word x of the
// vector), draws 1/3 to the same nibbles of ry (word y).
//
// gfx950 costs (measured, profiles/ubench_r01.txt): a VOPC compare is a 4-cycle issue, v_cndmask another 4 plus
// a 2-state VCC hazard, while a plain VOP2 add is 2 cycles.  So the lane predicate is written straight into EXEC
// (v_cmpx) and the add runs under it: 12 cycles per site instead of ~20 for compare+select+add.  The second
// compare runs under the first one's EXEC, which is correct because x < n4 implies x < n3.
template <int NIB>
__device__ __forceinline__ void accept_rank(uint32_t &rx, uint32_t &ry, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3,
                                            uint32_t n3, uint32_t n4) {
#if defined(ISING_ACCEPT_PLAIN)
	constexpr uint32_t K0 = 1u << (4 * NIB), K1 = 1u << (4 * NIB + 4);
	rx += (o0 < n3 ? K0 : 0u) + (o0 < n4 ? K0 : 0u) + (o2 < n3 ? K1 : 0u) + (o2 < n4 ? K1 : 0u);
	ry += (o1 < n3 ? K0 : 0u) + (o1 < n4 ? K0 : 0u) + (o3 < n3 ? K1 : 0u) + (o3 < n4 ? K1 : 0u);
#else
	constexpr uint32_t K0 = 1u << (4 * NIB), K1 = 1u << (4 * NIB + 4);
	unsigned long long saved;
	asm volatile(
	    "s_mov_b64 %[sv], exec\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n3], %[o0]\n\t"
	    "v_add_u32_e32 %[rx], %[k0], %[rx]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n4], %[o0]\n\t"
	    "v_add_u32_e32 %[rx], %[k0], %[rx]\n\t"
	    "s_mov_b64 exec, %[sv]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n3], %[o1]\n\t"
	    "v_add_u32_e32 %[ry], %[k0], %[ry]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n4], %[o1]\n\t"
	    "v_add_u32_e32 %[ry], %[k0], %[ry]\n\t"
	    "s_mov_b64 exec, %[sv]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n3], %[o2]\n\t"
	    "v_add_u32_e32 %[rx], %[k1], %[rx]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n4], %[o2]\n\t"
	    "v_add_u32_e32 %[rx], %[k1], %[rx]\n\t"
	    "s_mov_b64 exec, %[sv]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n3], %[o3]\n\t"
	    "v_add_u32_e32 %[ry], %[k1], %[ry]\n\t"
	    "v_cmpx_gt_u32_e32 vcc, %[n4], %[o3]\n\t"
	    "v_add_u32_e32 %[ry], %[k1], %[ry]\n\t"
	    "s_mov_b64 exec, %[sv]"
	    : [rx] "+v"(rx), [ry] "+v"(ry), [sv] "=&s"(saved)
	    : [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2), [o3] "v"(o3), [n3] "s"(n3), [n4] "s"(n4), [k0] "n"(K0), [k1] "n"(K1)
	    : "vcc");
#endif
}

// cuRAND's curand_uniform: x*2^-32 + 2^-33 in FP32, one rounding (the product is exact).
__device__ __forceinline__ float u01(uint32_t x) {
	return __fmaf_rn(__uint2float_rn(x), 0x1p-32f, 0x1p-33f);
}

// Nibble-wise number of up neighbours for the four dwords of one vector.
// ct/up/dw: centre/up/down source vectors; side: the neighbouring vector's dword that supplies the carry nibble
// (dword 3 of the left vector when back, dword 0 of the right vector otherwise); optimized/main.cu:546-573,:623-635.
template <bool USEJ = false>
__device__ __forceinline__ void neighbour_sums(const uint4 &up, const uint4 &ct, const uint4 &dw, uint32_t side, bool back,
                                               uint32_t S[4], const uint4 &J = uint4()) {
	uint32_t sd[4];
	if (back) {
		sd[0] = __builtin_amdgcn_alignbit(ct.x, side, 28);
		sd[1] = __builtin_amdgcn_alignbit(ct.y, ct.x, 28);
		sd[2] = __builtin_amdgcn_alignbit(ct.z, ct.y, 28);
		sd[3] = __builtin_amdgcn_alignbit(ct.w, ct.z, 28);
	} else {
		sd[0] = __builtin_amdgcn_alignbit(ct.y, ct.x, 4);
		sd[1] = __builtin_amdgcn_alignbit(ct.z, ct.y, 4);
		sd[2] = __builtin_amdgcn_alignbit(ct.w, ct.z, 4);
		sd[3] = __builtin_amdgcn_alignbit(side, ct.w, 4);
	}
	const uint32_t u[4] = {up.x, up.y, up.z, up.w}, c[4] = {ct.x, ct.y, ct.z, ct.w}, d[4] = {dw.x, dw.y, dw.z, dw.w};
	const uint32_t j[4] = {J.x, J.y, J.z, J.w};
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		if (USEJ) {
			// coupling bits <up, down, left, right> = 0x8, 0x4, 0x2, 0x1 flip the neighbour's contribution
			// (optimized/main.cu:575-618); the side word holds the left neighbours when `back`, else the right ones
			const uint32_t ju = (j[k] & 0x88888888u) >> 3, jd = (j[k] & 0x44444444u) >> 2;
			const uint32_t jl = (j[k] & 0x22222222u) >> 1, jr = j[k] & 0x11111111u;
			S[k] = (u[k] ^ ju) + (d[k] ^ jd) + (c[k] ^ (back ? jr : jl)) + (sd[k] ^ (back ? jl : jr));
		} else {
			S[k] = u[k] + c[k] + d[k] + sd[k];
		}
	}
}

// Resolve the flips of 8 sites (one dword): S = neighbour-up counts, R6 = accept ranks + 6 per nibble.
//   spin down: flip <=> n + r >= 2  <=> bit 3 of (n + r + 6);   spin up: flip <=> n - r <= 2 <=> !bit 3 of (n + 5 - r)
__device__ __forceinline__ uint32_t apply_flips(uint32_t me, uint32_t S, uint32_t R6) {
	const uint32_t t0 = S + R6;
	const uint32_t t1 = (S + 0xBBBBBBBBu) - R6;               // == S + 0x55555555 - R as whole-word arithmetic
	const uint32_t m8 = me << 3;
	const uint32_t f = __builtin_amdgcn_bitop3_b32(t0, t1, m8, 0x72);  // m8 ? ~t1 : t0   (bit 3 of each nibble)
	return __builtin_amdgcn_bitop3_b32(me, f >> 3, 0x11111111u, 0x78); // me ^ ((f >> 3) & 0x1111...)
}

// ---------------------------------------------------------------------------------------------- update
// MODE 0: integer thresholds, v_cmpx accept (default).  MODE 1: generic FP32 table.  (A third form -- integer thresholds
// through a 64 KiB LDS rank table -- lost its A/B comparison in round 1 and was removed in round 4; LAB_NOTES.md keeps its numbers.)
// USEJ (-J couplings) and SUBL (sub-lattices) are template switches so the common case carries none of their code.
template <int MODE, bool USEJ = false, bool SUBL = false>
__global__ void __launch_bounds__(THREADS) update_k(const UpdateParams p) {
	__shared__ float sh_tab[10];
	if (MODE == 1) {
		if (threadIdx.x < 10) sh_tab[threadIdx.x] = p.tab[threadIdx.x];
		__syncthreads();
	}
	const int tx = threadIdx.x & (GROUP - 1);
	// XCD-aware block order: hardware block b runs on XCD b % 8 (observed, used for speed only), so give each XCD a
	// contiguous range of logical blocks; vertically adjacent strips then share halo rows through one L2.
	int lb = blockIdx.x;
#if defined(ISING_XCD_REMAP) // A/B tested at 65536^2: no measurable effect (within +-1 % noise), so off by default
	if ((gridDim.x & 7) == 0) lb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#endif
	const int unit = lb * (THREADS / GROUP) + (threadIdx.x >> 4);
	if (unit >= p.nunits) return;
	const int rng = unit >= p.nunits0;
	const int u = unit - (rng ? p.nunits0 : 0);
	const int sidx = u / p.gx;
	const int bx = u - sidx * p.gx;
	const int r0 = p.row_lo[rng] + sidx * p.H;
	const int nrows = min(p.H, p.row_hi[rng] - r0);
	const int vecs = p.gx * 32;
	const int col0 = bx * 32 + tx;
	// Side-neighbour carry dwords, as dword offsets from this lane's own vector in the same row, with the periodic
	// wrap of loadTile (optimized/main.cu:433,:441): dword 3 of the vector to the left / dword 0 of the one to the right.
	// With sub-lattices (--xsl) the wrap happens every slV vectors instead of once per row.
	const int slV = SUBL ? p.slV : vecs;
	const int offL0 = ((col0 % slV) == 0 ? slV - 1 : -1) * 4 + 3, offL1 = 15 * 4 + 3;
	const int offR0 = 4, offR1 = (((col0 + GROUP + 1) % slV) == 0 ? 1 - slV + GROUP : GROUP + 1) * 4;
	// Row wrap: without sub-lattices rows -1 and Y are the physical halo rows; with them (--ysl) the row above the
	// first row of a sub-lattice is its last row and vice versa (loadTile, optimized/main.cu:414,:422).
	// `seam` counts the rows left in the current sub-lattice (one integer division per strip, none per row).
	const int slY = SUBL ? p.slY : 0;
	const int r0_in_sl = slY ? r0 % slY : 1;
	int seam = slY ? slY - r0_in_sl : 0x7fffffff;

	// Rows -1 and Y of every colour array are physically present (halo rows), so row r lives at src + r*vecs.
	const uint4 *pc = reinterpret_cast<const uint4 *>(p.src) + ((ptrdiff_t)r0 * vecs + col0); // centre row, own vector
	uint4 *pm = reinterpret_cast<uint4 *>(p.dst) + ((ptrdiff_t)r0 * vecs + col0);
	const uint4 *pj = USEJ ? reinterpret_cast<const uint4 *>(p.jdst) + ((ptrdiff_t)r0 * vecs + col0) : nullptr;
	const ptrdiff_t mir0 = (ptrdiff_t)(p.mir0_bytes / 16), mirL = (ptrdiff_t)(p.mirL_bytes / 16); // see ballot_update_k

	const uint32_t k2y = p.seed_hi + 2u * PHILOX_W1;
	const uint32_t cx_base = 16u * (2u * p.it + p.color);
	// The draw-block counter 16(2 it + colour) + b is 64 bits wide in cuRAND; its high word (non-zero from iteration
	// 2^27 on) enters round 1 next to key word 0, so it folds into the seed operand of the per-row setup.
	const uint32_t seed_lo_cy = p.seed_lo ^ (uint32_t)((2ull * p.it + p.color) >> 28);

	const ptrdiff_t uo = (slY && r0_in_sl == 0) ? (ptrdiff_t)(slY - 1) * vecs : -(ptrdiff_t)vecs;
	uint4 up0 = pc[uo], up1 = pc[uo + GROUP];
	uint4 ct0 = pc[0], ct1 = pc[GROUP];

	for (int r = 0; r < nrows; ++r) {
		const int lr = r0 + r;
		const uint32_t grow = p.row_base + (uint32_t)lr;
		const bool back = (p.color == 0) ? !(grow & 1u) : (grow & 1u); // readBack, optimized/main.cu:542
		// issue this row's loads; they are consumed only after the 16 Philox blocks below
		const bool sl_last = SUBL && seam == 1; // last row of its sub-lattice: the row below is the sub-lattice's first row
		const ptrdiff_t dwo = sl_last ? (ptrdiff_t)(1 - slY) * vecs : (ptrdiff_t)vecs;
		const uint4 dw0 = pc[dwo], dw1 = pc[dwo + GROUP];
		const uint32_t *pcw = reinterpret_cast<const uint32_t *>(pc);
		const uint32_t side0 = pcw[back ? offL0 : offR0];
		const uint32_t side1 = pcw[back ? offL1 : offR1];
		uint4 me0 = pm[0], me1 = pm[GROUP];
		uint4 j0 = uint4(), j1 = uint4();
		if (USEJ) { j0 = pj[0]; j1 = pj[GROUP]; pj += vecs; }

		// stream id of the reference thread that owns these two vectors (optimized/main.cu:514-515)
		const uint32_t tid = ((grow >> 4) * (uint32_t)p.gx + (uint32_t)bx) * 256u + (grow & 15u) * 16u + (uint32_t)tx;
		const PhiloxRow pr = philox_row_setup(tid, seed_lo_cy, k2y);

		if (MODE == 0) {
			// accept ranks, pre-biased by 6 per nibble (see apply_flips)
			uint32_t R[2][4] = {{0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u}, {0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u}};
			static_for<16>([&](auto B) {
				constexpr int j = B.value >> 3, m = B.value & 7;
				uint32_t o[4];
				philox_block(pr, cx_base + (uint32_t)B.value, p.seed_lo, p.seed_hi, o[0], o[1], o[2], o[3]);
				// outputs: (nibble 2m, word x), (2m, y), (2m+1, x), (2m+1, y)   
				accept_rank<(2 * m) & 7>(R[j][m >> 2], R[j][2 + (m >> 2)], o[0], o[1], o[2], o[3], p.n3, p.n4);
			});
			uint32_t S[4];
			neighbour_sums<USEJ>(up0, ct0, dw0, side0, back, S, j0);
			me0 = make_uint4(apply_flips(me0.x, S[0], R[0][0]), apply_flips(me0.y, S[1], R[0][1]),
			                 apply_flips(me0.z, S[2], R[0][2]), apply_flips(me0.w, S[3], R[0][3]));
			neighbour_sums<USEJ>(up1, ct1, dw1, side1, back, S, j1);
			me1 = make_uint4(apply_flips(me1.x, S[0], R[1][0]), apply_flips(me1.y, S[1], R[1][1]),
			                 apply_flips(me1.z, S[2], R[1][2]), apply_flips(me1.w, S[3], R[1][3]));
		} else {
			// generic: the reference's own per-site FP32 test, optimized/main.cu:637-660
			uint32_t S[2][4];
			neighbour_sums<USEJ>(up0, ct0, dw0, side0, back, S[0], j0);
			neighbour_sums<USEJ>(up1, ct1, dw1, side1, back, S[1], j1);
			uint32_t mv[2][4] = {{me0.x, me0.y, me0.z, me0.w}, {me1.x, me1.y, me1.z, me1.w}};
#pragma unroll
			for (int j = 0; j < 2; ++j) {
#pragma unroll
				for (int m = 0; m < 8; ++m) {
					uint32_t o[4];
					philox_block(pr, cx_base + 8u * j + m, p.seed_lo, p.seed_hi, o[0], o[1], o[2], o[3]);
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						const int z = 2 * m + (q >> 1), w = q & 1;
						const int d = 2 * w + (z >> 3), sh = 4 * (z & 7);
						const uint32_t sp = (mv[j][d] >> sh) & 0xFu;
						const uint32_t n = (S[j][d] >> sh) & 0xFu;
						if (u01(o[q]) <= sh_tab[sp * 5 + n]) mv[j][d] ^= 1u << sh;
					}
				}
			}
			me0 = make_uint4(mv[0][0], mv[0][1], mv[0][2], mv[0][3]);
			me1 = make_uint4(mv[1][0], mv[1][1], mv[1][2], mv[1][3]);
		}
		pm[0] = me0;
		pm[GROUP] = me1;
		if (p.wrap) { // the halo rows that mirror this colour's edge rows
			if (lr == 0) { pm[mir0] = me0; pm[mir0 + GROUP] = me1; }
			if (lr == p.Y - 1) { pm[mirL] = me0; pm[mirL + GROUP] = me1; }
		}
		pc += vecs;
		pm += vecs;
		if (sl_last) {
			// the next row opens a new sub-lattice: the register window does not slide across the seam
			seam = slY;
			if (r + 1 < nrows) {
				const ptrdiff_t uo2 = (ptrdiff_t)(slY - 1) * vecs;
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

// ---------------------------------------------------------------------------------------------- init
__global__ void __launch_bounds__(THREADS) init_k(const InitParams p) {
	const int tx = threadIdx.x & (GROUP - 1);
	const long long unit_ll = flat_block() * GROUPS_PER_BLOCK + (threadIdx.x >> 4); // (row, bx)
	if (unit_ll >= (long long)p.gx * p.Y) return;
	const int unit = (int)unit_ll;
	const int lr = unit / p.gx;
	const int bx = unit - lr * p.gx;
	const int vecs = p.gx * 32;
	const uint32_t grow = p.row_base + (uint32_t)lr;
	const uint32_t tid = ((grow >> 4) * (uint32_t)p.gx + (uint32_t)bx) * 256u + (grow & 15u) * 16u + (uint32_t)tx;
	const PhiloxRow pr = philox_row_setup(tid, p.seed_lo, p.seed_hi + 2u * PHILOX_W1);
	const uint32_t cx_base = 16u * p.color; // it = 0, optimized/main.cu:116
	uint4 *row = reinterpret_cast<uint4 *>(p.dst) + (size_t)lr * vecs;
#pragma unroll
	for (int j = 0; j < 2; ++j) {
		uint32_t v[4] = {0, 0, 0, 0};
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			uint32_t o[4];
			philox_block(pr, cx_base + 8u * j + m, p.seed_lo, p.seed_hi, o[0], o[1], o[2], o[3]);
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int z = 2 * m + (q >> 1), w = q & 1;
				// curand_uniform(x) < 0.5f  <=>  x < thr_half   (optimized/main.cu:133,:136)
				if (o[q] < p.thr_half) v[2 * w + (z >> 3)] |= 1u << (4 * (z & 7));
			}
		}
		const uint4 val = make_uint4(v[0], v[1], v[2], v[3]);
		row[bx * 32 + tx + j * GROUP] = val;
		if (p.wrap) { // single slab: the halo rows mirror the opposite edge rows
			if (lr == 0) row[(ptrdiff_t)p.Y * vecs + bx * 32 + tx + j * GROUP] = val;
			if (lr == p.Y - 1) row[-(ptrdiff_t)p.Y * vecs + bx * 32 + tx + j * GROUP] = val;
		}
	}
}

// ---------------------------------------------------------------------------------------------- couplings (-J)
// hamiltInitB_k (optimized/main.cu:153-212): 256 draws per reference thread, generator offset 0: draw block
// b = 32 j + 2 z + h of vector j feeds bits 4z+2h and 4z+2h+1 of word x (outputs 0, 2) and word y (outputs 1, 3).
__global__ void __launch_bounds__(THREADS) ham_init_black_k(const HamInitParams p) {
	const int tx = threadIdx.x & (GROUP - 1);
	const long long unit_ll = flat_block() * GROUPS_PER_BLOCK + (threadIdx.x >> 4); // (row, bx)
	if (unit_ll >= (long long)p.gx * p.Y) return;
	const int unit = (int)unit_ll;
	const int lr = unit / p.gx;
	const int bx = unit - lr * p.gx;
	const int vecs = p.gx * 32;
	uint32_t grow = p.row_base + (uint32_t)lr;
	if (p.total_rows) grow %= p.total_rows; // (a ring of one wraps more than once: its ghost rows are its own rows)
	const uint32_t tid = ((grow >> 4) * (uint32_t)p.gx + (uint32_t)bx) * 256u + (grow & 15u) * 16u + (uint32_t)tx;
	const PhiloxRow pr = philox_row_setup(tid, p.seed_lo, p.seed_hi + 2u * PHILOX_W1);
	uint4 *row = reinterpret_cast<uint4 *>(p.hamB) + (size_t)lr * vecs;
	for (int j = 0; j < 2; ++j) {
		uint32_t v[4] = {0, 0, 0, 0};
#pragma unroll 4
		for (int b = 0; b < 32; ++b) {
			uint32_t o[4];
			philox_block(pr, (uint32_t)(32 * j + b), p.seed_lo, p.seed_hi, o[0], o[1], o[2], o[3]);
			const int z = b >> 1, bit = 4 * (z & 7) + 2 * (b & 1), d = z >> 3;
			if (o[0] < p.thr) v[d] |= 1u << bit;
			if (o[1] < p.thr) v[2 + d] |= 1u << bit;
			if (o[2] < p.thr) v[d] |= 2u << bit;
			if (o[3] < p.thr) v[2 + d] |= 2u << bit;
		}
		const uint4 val = make_uint4(v[0], v[1], v[2], v[3]);
		const int col = bx * 32 + tx + j * GROUP;
		row[col] = val;
		if (p.wrap) {
			if (lr == 0) row[(ptrdiff_t)p.Y * vecs + col] = val;
			if (lr == p.Y - 1) row[-(ptrdiff_t)p.Y * vecs + col] = val;
		}
	}
}

// hamiltInitW_k (optimized/main.cu:214-331) turned from a scatter with atomicOr into a gather: white word (i, q)
// collects, from the black words at the other ends of its bonds, the bit that describes the same bond:
//   down (0x4) <- up (0x8) of the black word below;  up (0x8) <- down (0x4) of the black word above;
//   even rows: left (0x2) <- right (0x1) of the same black nibble, right (0x1) <- left (0x2) of the next black nibble;
//   odd rows:  right (0x1) <- left (0x2) of the same black nibble, left (0x2) <- right (0x1) of the previous nibble.
__global__ void __launch_bounds__(THREADS) ham_init_white_k(const HamWhiteParams p) {
	const size_t total = (size_t)p.lld * p.Y;
	const unsigned long long M8 = 0x8888888888888888ull, M4 = 0x4444444444444444ull, M2 = 0x2222222222222222ull, M1 = 0x1111111111111111ull;
	for (size_t idx = blockIdx.x * (size_t)THREADS + threadIdx.x; idx < total; idx += (size_t)gridDim.x * THREADS) {
		const int i = (int)(idx / p.lld), q = (int)(idx - (size_t)i * p.lld);
		const ptrdiff_t uo = (p.slY && (i % p.slY) == 0) ? (ptrdiff_t)(p.slY - 1) * p.lld : -(ptrdiff_t)p.lld;
		const ptrdiff_t dwo = (p.slY && ((i + 1) % p.slY) == 0) ? (ptrdiff_t)(1 - p.slY) * p.lld : (ptrdiff_t)p.lld;
		const uint64_t *b = p.hamB + (ptrdiff_t)i * p.lld;
		const unsigned long long me = b[q];
		unsigned long long w = ((b[q + dwo] & M8) >> 1) | ((b[q + uo] & M4) << 1);
		if ((p.row_base + (uint32_t)i) & 1u) {
			const int qp = (q % p.slW) == 0 ? q + p.slW - 1 : q - 1;
			w |= ((me & M2) >> 1) | ((me & M1) << 5) | ((b[qp] & M1) >> 59);
		} else {
			const int qn = ((q + 1) % p.slW) == 0 ? q + 1 - p.slW : q + 1;
			w |= ((me & M1) << 1) | ((me & M2) >> 5) | ((b[qn] & M2) << 59);
		}
		p.hamW[idx] = w;
		if (p.wrap) {
			if (i == 0) p.hamW[idx + (size_t)p.Y * p.lld] = w;
			if (i == p.Y - 1) p.hamW[(ptrdiff_t)idx - (ptrdiff_t)p.Y * p.lld] = w;
		}
	}
}

// ---------------------------------------------------------------------------------------------- popcount

// ---- measurement aid (bench.py "alu_ceiling"): the draw of the update kernels and nothing else -- per lane `nrows` x 16
// Philox4x32-10 blocks exactly as the update kernels generate them (philox_row_setup + philox_block: rounds 1-2 of
// the wave-uniform counter word on the scalar unit), outputs XOR-folded into one word per lane.  sites/ns of this
// kernel is what a half-sweep could reach if accept test, word logic and memory cost nothing.
__global__ void __launch_bounds__(THREADS) philox_ceiling_k(uint32_t *__restrict__ out, uint32_t seed_lo, uint32_t seed_hi, int nrows, unsigned long long *__restrict__ clk) {
	const uint32_t tid0 = blockIdx.x * THREADS + threadIdx.x;
	const bool marks = clk != nullptr && blockIdx.x < 8u && threadIdx.x == 0; // one wave per XCD: the clock this launch runs at
	if (marks) { clk[4 * blockIdx.x] = __builtin_readcyclecounter(); clk[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
	uint32_t acc = 0;
	for (int r = 0; r < nrows; ++r) {
		const PhiloxRow pr = philox_row_setup(tid0 + (uint32_t)r * 0x10000u, seed_lo, seed_hi + 2u * PHILOX_W1);
		static_for<16>([&](auto B) {
			uint32_t o0, o1, o2, o3;
			uint32_t cx = 16u * (uint32_t)r + (uint32_t)B.value;
			asm volatile("" : "+s"(cx));
			philox_block(pr, cx, seed_lo, seed_hi, o0, o1, o2, o3);
			acc ^= xor3(o0, o1, o2) ^ o3;
		});
	}
	out[tid0] = acc;
	if (marks) { clk[4 * blockIdx.x + 2] = __builtin_readcyclecounter(); clk[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime(); }
}

__global__ void __launch_bounds__(THREADS) popcount_k(const uint4 *__restrict__ v, size_t nvec, unsigned long long *acc) {
	__shared__ unsigned long long part[THREADS / 64];
	unsigned long long c = 0;
	for (size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x; i < nvec; i += (size_t)gridDim.x * THREADS) {
		const uint4 w = v[i];
		c += __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w);
	}
	c = wave_sum(c);
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long t = 0;
#pragma unroll
		for (int i = 0; i < THREADS / 64; ++i) t += part[i];
		atomicAdd(acc, t);
	}
}

// ---------------------------------------------------------------------------------------------- bond sum
__global__ void __launch_bounds__(THREADS) bond_equal_k(const BondParams p) {
	__shared__ unsigned long long part[THREADS / 64];
	const int vecs = p.gx * 32;
	const size_t total = (size_t)vecs * p.Y;
	const uint4 *white = reinterpret_cast<const uint4 *>(p.white);
	const uint4 *black = reinterpret_cast<const uint4 *>(p.black);
	unsigned long long acc = 0;
	for (size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * THREADS) {
		const int lr = (int)(i / vecs);
		const int col = (int)(i - (size_t)lr * vecs);
		const uint32_t grow = p.row_base + (uint32_t)lr;
		const bool back = !(grow & 1u); // black sites
		const uint4 *pc = white + (ptrdiff_t)lr * vecs; // rows -1 and Y are the halo rows
		const int slV = p.slV, slY = p.slY;
		const int colL = (col % slV) == 0 ? col + slV - 1 : col - 1, colR = ((col + 1) % slV) == 0 ? col + 1 - slV : col + 1;
		const ptrdiff_t uo = (slY && (lr % slY) == 0) ? (ptrdiff_t)(slY - 1) * vecs : -(ptrdiff_t)vecs;
		const ptrdiff_t dwo = (slY && ((lr + 1) % slY) == 0) ? (ptrdiff_t)(1 - slY) * vecs : (ptrdiff_t)vecs;
		const uint32_t *pcw = reinterpret_cast<const uint32_t *>(pc);
		const uint32_t side = back ? pcw[4 * colL + 3] : pcw[4 * colR];
		uint32_t S[4];
		neighbour_sums(pc[col + uo], pc[col], pc[col + dwo], side, back, S);
		const uint4 me = black[i];
		const uint32_t mv[4] = {me.x, me.y, me.z, me.w};
#pragma unroll
		for (int d = 0; d < 4; ++d) {
			const uint32_t mask = mv[d] * 15u;                               // 0xF where spin up
			const uint32_t a = (S[d] & mask) | ((0x44444444u - S[d]) & ~mask); // aligned neighbours per nibble
			const uint32_t b = (a & 0x0F0F0F0Fu) + ((a >> 4) & 0x0F0F0F0Fu);
			acc += (b * 0x01010101u) >> 24;
		}
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

// ---------------------------------------------------------------------------------------------- correlations
// Replaces getCorr2D_k (optimized/main.cu:870-965).  The reference extracts two nibbles per (site, distance); here the
// lattice is first compressed to one bit per spin in lattice-column order (X bits per row), after which one
// 32-site comparison is an XOR + popcount: #equal - #unequal = 32 - 2*popc(a ^ b).

// 16 nibble-spins of one packed 64-bit word -> 16 bits (bit k = spin of nibble k)
__device__ __forceinline__ uint32_t nibbles_to_bits(uint64_t w) {
	uint64_t x = w & 0x1111111111111111ull;
	x = (x | (x >> 3)) & 0x0303030303030303ull;
	x = (x | (x >> 6)) & 0x000F000F000F000Full;
	x = (x | (x >> 12)) & 0x000000FF000000FFull;
	x = (x | (x >> 24)) & 0xFFFFull;
	return (uint32_t)x;
}
// 16 bits -> the even bit positions of a 32-bit word
__device__ __forceinline__ uint32_t spread16(uint32_t x) {
	x = (x | (x << 8)) & 0x00FF00FFu;
	x = (x | (x << 4)) & 0x0F0F0F0Fu;
	x = (x | (x << 2)) & 0x33333333u;
	x = (x | (x << 1)) & 0x55555555u;
	return x;
}

// bits[r][q] (q = 0..lld-1) holds lattice columns 32q..32q+31 of slab row r: column c is colour-site c/2 of the white
// array when (global row ^ c) is odd, of the black array otherwise (optimized/main.cu:928-929, dumpLattice :1165-1172).
__global__ void __launch_bounds__(THREADS) pack_bits_k(const uint64_t *__restrict__ black, const uint64_t *__restrict__ white,
                                                       int lld, int Y, uint32_t row_base, uint32_t *__restrict__ bits) {
	const size_t total = (size_t)lld * Y;
	for (size_t i = blockIdx.x * (size_t)THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * THREADS) {
		const uint32_t r = (uint32_t)(i / lld);
		const uint32_t b = nibbles_to_bits(black[i]), w = nibbles_to_bits(white[i]);
		const bool odd = (row_base + r) & 1u;
		// even row: columns 2k -> black k, 2k+1 -> white k; odd row: the other way round
		bits[i] = odd ? (spread16(w) | (spread16(b) << 1)) : (spread16(b) | (spread16(w) << 1));
	}
}

// grid (row chunks, ncorr): block (y = j-1) accumulates over its rows
//   sum_c [s(r,c) == s(r,c+j)] - [!=]  +  [s(r,c) == s(r+j,c)] - [!=]      (c periodic in X; rows r+j come from `bits`
// rows up to Y+ncorr-1, which the host fills with the rows that follow this slab).
__global__ void __launch_bounds__(THREADS) corr_k(const uint32_t *__restrict__ bits, int lld, int Y, int rows_per_block,
                                                  int slW, int slY, long long *__restrict__ sums) {
	// slW: words per periodic segment of a row (= lld without sub-lattices); slY: rows per sub-lattice, 0 = none (then the
	// vertical partner of row r is row r+j of `bits`, which holds ncorr extra rows).  getCorr2DRepl_k :967-1070.
	__shared__ long long part[THREADS / 64];
	const int j = blockIdx.y + 1;
	const int wsh = j >> 5, bsh = j & 31;
	const int r_lo = blockIdx.x * rows_per_block, r_hi = min(Y, r_lo + rows_per_block);
	long long acc = 0;
	for (int r = r_lo; r < r_hi; ++r) {
		int rv = r + j;
		if (slY && rv >= (r / slY + 1) * slY) rv -= slY;
		const uint32_t *row = bits + (size_t)r * lld, *rowv = bits + (size_t)rv * lld;
		for (int q = threadIdx.x; q < lld; q += THREADS) {
			const uint32_t a = row[q];
			const int seg = (q / slW) * slW;
			int q0 = q - seg + wsh; if (q0 >= slW) q0 -= slW;
			int q1 = q0 + 1;        if (q1 >= slW) q1 -= slW;
			const uint32_t h = bsh ? __builtin_amdgcn_alignbit(row[seg + q1], row[seg + q0], bsh) : row[seg + q0]; // columns +j
			acc += 64 - 2 * (int)(__popc(a ^ h) + __popc(a ^ rowv[q]));
		}
	}
	acc = (long long)wave_sum((unsigned long long)acc);
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) {
		long long t = 0;
#pragma unroll
		for (int i = 0; i < THREADS / 64; ++i) t += part[i];
		atomicAdd(reinterpret_cast<unsigned long long *>(sums + (j - 1)), (unsigned long long)t);
	}
}

} // namespace

// ---------------------------------------------------------------------------------------------- launchers
hipError_t launch_pack_bits(const uint64_t *black, const uint64_t *white, int lld, int Y, uint32_t row_base, uint32_t *bits,
                            hipStream_t stream) {
	size_t blocks = ((size_t)lld * Y + THREADS - 1) / THREADS;
	if (blocks > 8192) blocks = 8192;
	hipLaunchKernelGGL(pack_bits_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, black, white, lld, Y, row_base, bits);
	return hipGetLastError();
}

hipError_t launch_corr(const uint32_t *bits, int lld, int Y, int ncorr, int slW, int slY, long long *sums, hipStream_t stream) {
	const int rows_per_block = 16;
	const dim3 grid((Y + rows_per_block - 1) / rows_per_block, ncorr);
	hipLaunchKernelGGL(corr_k, grid, dim3(THREADS), 0, stream, bits, lld, Y, rows_per_block, slW, slY, sums);
	return hipGetLastError();
}

hipError_t launch_update(const UpdateParams &p, int mode, hipStream_t stream) {
	if (p.nunits <= 0) return hipSuccess;
	const dim3 g0((p.nunits + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK), b0(THREADS);
	const bool J = p.jdst != nullptr, S = p.slY != 0;
	if (mode == 1) {
		if (J && S)  hipLaunchKernelGGL((update_k<1, true, true>), g0, b0, 0, stream, p);
		else if (J)  hipLaunchKernelGGL((update_k<1, true, false>), g0, b0, 0, stream, p);
		else if (S)  hipLaunchKernelGGL((update_k<1, false, true>), g0, b0, 0, stream, p);
		else         hipLaunchKernelGGL((update_k<1, false, false>), g0, b0, 0, stream, p);
	} else {
		if (J && S)  hipLaunchKernelGGL((update_k<0, true, true>), g0, b0, 0, stream, p);
		else if (J)  hipLaunchKernelGGL((update_k<0, true, false>), g0, b0, 0, stream, p);
		else if (S)  hipLaunchKernelGGL((update_k<0, false, true>), g0, b0, 0, stream, p);
		else         hipLaunchKernelGGL((update_k<0, false, false>), g0, b0, 0, stream, p);
	}
	return hipGetLastError();
}

hipError_t launch_ham_init_black(const HamInitParams &p, hipStream_t stream) {
	const long long units = (long long)p.gx * p.Y;
	hipLaunchKernelGGL(ham_init_black_k, flat_grid((units + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK), dim3(THREADS), 0, stream, p);
	return hipGetLastError();
}

hipError_t launch_ham_init_white(const HamWhiteParams &p, hipStream_t stream) {
	size_t blocks = ((size_t)p.lld * p.Y + THREADS - 1) / THREADS;
	if (blocks > 8192) blocks = 8192;
	hipLaunchKernelGGL(ham_init_white_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, p);
	return hipGetLastError();
}

hipError_t launch_init(const InitParams &p, hipStream_t stream) {
	const long long units = (long long)p.gx * p.Y;
	const dim3 grid = flat_grid((units + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK), block(THREADS);
	hipLaunchKernelGGL(init_k, grid, block, 0, stream, p);
	return hipGetLastError();
}

hipError_t launch_philox_ceiling(uint32_t *out, int blocks, int nrows, hipStream_t stream, unsigned long long *clk) {
	hipLaunchKernelGGL(philox_ceiling_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, out, 0x1234567u, 0x89abcdeu, nrows, clk);
	return hipGetLastError();
}

hipError_t launch_popcount(const uint64_t *v, size_t nwords, unsigned long long *acc, hipStream_t stream) {
	const size_t nvec = nwords / 2;
	size_t blocks = (nvec + THREADS - 1) / THREADS;
	if (blocks > 2048) blocks = 2048;
	if (blocks == 0) return hipSuccess;
	hipLaunchKernelGGL(popcount_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, reinterpret_cast<const uint4 *>(v), nvec, acc);
	return hipGetLastError();
}

hipError_t launch_bond_equal(const BondParams &p, hipStream_t stream) {
	const size_t total = (size_t)p.gx * 32 * p.Y;
	size_t blocks = (total + THREADS - 1) / THREADS;
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(bond_equal_k, dim3((unsigned)blocks), dim3(THREADS), 0, stream, p);
	return hipGetLastError();
}

} // namespace ising
