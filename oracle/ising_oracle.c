/*
 * ising_oracle.c -- CPU restatement of the reference's multi-spin-coded checkerboard Metropolis path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (ising_gpu_amd/, the C-ABI library, the CLI) may link,
 * import or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and
 * only as the checker.
 *
 * What it restates (file:line are relative to /root/reference):
 *   - lattice layout and sizes ................ optimized/main.cu:1539-1550, :1673-1674
 *   - latticeInit_k ........................... optimized/main.cu:92-151
 *   - loadTile (periodic neighbour addressing)  optimized/main.cu:400-459
 *   - spinUpdateV_2D_k ........................ optimized/main.cu:511-573 (neighbour words, side shift),
 *                                               :620-668 (RNG offset, nibble sums, accept, store)
 *   - getMagn_k / countSpins .................. optimized/main.cu:710-732, :831-868
 *   - exp table ............................... optimized/main.cu:1684-1697
 *   - sweep order (black, then white, it=j+1) . optimized/main.cu:1763-1799
 *   - dumpLattice text format ................. optimized/main.cu:1140-1209
 * Third-party arithmetic the reference gets from cuRAND's device API (curand_kernel.h, NOT under
 * /root/reference; CUDA toolkit version unpinned by optimized/Makefile:1-2) is restated from the published
 * Philox4x32-10 algorithm (Salmon et al., SC'11; Random123) and cuRAND's documented semantics:
 *   curand_init(seed, subsequence, offset): key=(lo32(seed),hi32(seed)); 128-bit counter = offset/4 in
 *   words 0-1, subsequence in words 2-3; curand() hands out the 4 outputs of a block in order x,y,z,w,
 *   then increments the counter; curand_uniform(x) = x*2^-32 + 2^-33 in FP32 (result in (0,1]).
 * Parity is PINNED: tests/test_oracle_kat.py checks this file against the console transcripts in
 * optimized/README.md (:128-137, :187-196, :240-249, :307) -- the only golden vectors the reference has.
 *
 * The loops deliberately keep the reference's launch geometry (16x16-thread blocks, two 128-bit vectors
 * per thread, one sequential generator per thread) so that the site -> (stream, draw) mapping is the
 * reference's by construction rather than by a derived closed form.  The HIP kernels use the closed form;
 * agreement between the two is what the parity tests establish.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <omp.h>

#define BLK_X 16   /* optimized/main.cu:55 */
#define BLK_Y 16   /* optimized/main.cu:56 */
#define VEC_PER_THREAD 2 /* BMULT_X, optimized/main.cu:57 */
#define NIB 16     /* spins per 64-bit word at 4 bit/spin, optimized/main.cu:40,:1243 */

enum { ORC_BLACK = 0, ORC_WHITE = 1 };

/* ------------------------------------------------------------------ Philox4x32-10 (Random123 / cuRAND) */
typedef struct { uint32_t c[4]; uint32_t k[2]; uint32_t out[4]; int pos; } orc_gen;

static inline void philox_block(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
	uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
	uint32_t k0 = key[0], k1 = key[1];
	for (int r = 0; r < 10; r++) {
		const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
		const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
		const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
		const uint32_t n1 = (uint32_t)p1;
		const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
		const uint32_t n3 = (uint32_t)p0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
	}
	out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
	philox_block(ctr, key, out);
}

/* curand_init(seed, subsequence, offset, &st) for curandStatePhilox4_32_10_t */
static inline void gen_init(orc_gen *g, uint64_t seed, uint64_t subseq, uint64_t offset) {
	g->k[0] = (uint32_t)seed; g->k[1] = (uint32_t)(seed >> 32);
	const uint64_t blk = offset >> 2;
	g->c[0] = (uint32_t)blk; g->c[1] = (uint32_t)(blk >> 32);
	g->c[2] = (uint32_t)subseq; g->c[3] = (uint32_t)(subseq >> 32);
	g->pos = (int)(offset & 3);
	philox_block(g->c, g->k, g->out);
}

/* curand(&st): next 32-bit output */
static inline uint32_t gen_next(orc_gen *g) {
	const uint32_t r = g->out[g->pos++];
	if (g->pos == 4) {
		if (++g->c[0] == 0) if (++g->c[1] == 0) if (++g->c[2] == 0) ++g->c[3];
		philox_block(g->c, g->k, g->out);
		g->pos = 0;
	}
	return r;
}

/* curand_uniform: FP32, (0,1] */
static inline float u01(uint32_t x) {
	return (float)x * 0x1p-32f + 0x1p-33f;
}
float orc_uniform(uint32_t x) { return u01(x); }

/* ------------------------------------------------------------------ exp table, optimized/main.cu:1684-1697 */
void orc_exp_table(float temp, float tab[10]) {
	for (int i = 0; i < 2; i++) {
		for (int j = 0; j < 5; j++) {
			if (temp > 0) {
				tab[i*5 + j] = expf((i ? -2.0f : 2.0f) * (float)(j*2 - 4) * (1.0f / temp));
			} else {
				tab[i*5 + j] = (j == 2) ? 0.5f : (i ? -2.0f : 2.0f) * (float)(j*2 - 4);
			}
		}
	}
}

/* ------------------------------------------------------------------ geometry helpers */
typedef struct {
	int64_t lld;     /* 64-bit words per colour row = X/32 */
	int64_t vecs;    /* 128-bit vectors per colour row = lld/2 ("dimX") */
	int64_t gx, gy;  /* launch grid of the reference */
} orc_geom;

static int geom(int64_t X, int64_t Ytot, orc_geom *g) {
	if (X <= 0 || Ytot <= 0 || X % (2*NIB*2*BLK_X*VEC_PER_THREAD) || Ytot % BLK_Y) return -1; /* :1412-1421 */
	g->lld = (X/2)/NIB;
	g->vecs = g->lld/2;
	g->gx = g->vecs/(BLK_X*VEC_PER_THREAD);
	g->gy = Ytot/BLK_Y;
	return 0;
}

/* ------------------------------------------------------------------ latticeInit_k, optimized/main.cu:92-151 */
static void init_color(uint64_t *dst, const orc_geom *g, int64_t Y, int64_t row_base, uint64_t seed, int color) {
	#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < Y; i++) {
		const int64_t gi = row_base + i;
		const int64_t by = gi / BLK_Y; const int ty = (int)(gi % BLK_Y);
		for (int64_t bx = 0; bx < g->gx; bx++) {
			for (int tx = 0; tx < BLK_X; tx++) {
				const uint32_t tid = (uint32_t)((by*g->gx + bx)*BLK_X*BLK_Y + ty*BLK_X + tx); /* :112-113 */
				orc_gen st;
				gen_init(&st, seed, tid, (uint64_t)(2*NIB)*VEC_PER_THREAD*(2*0 + color)); /* :116, it = 0 */
				for (int j = 0; j < VEC_PER_THREAD; j++) {
					uint64_t x = 0, y = 0;
					for (int k = 0; k < 64; k += 4) {               /* :131-139 */
						if (u01(gen_next(&st)) < 0.5f) x |= 1ull << k;
						if (u01(gen_next(&st)) < 0.5f) y |= 1ull << k;
					}
					const int64_t col = bx*BLK_X*VEC_PER_THREAD + tx + j*BLK_X;
					dst[i*g->lld + 2*col]     = x;
					dst[i*g->lld + 2*col + 1] = y;
				}
			}
		}
	}
}

int orc_init_slab(uint64_t *black, uint64_t *white, int64_t X, int64_t Y, int64_t row_base, uint64_t seed) {
	orc_geom g;
	if (geom(X, Y, &g)) return -1;
	init_color(black, &g, Y, row_base, seed, ORC_BLACK);
	init_color(white, &g, Y, row_base, seed, ORC_WHITE);
	return 0;
}

int orc_init(uint64_t *black, uint64_t *white, int64_t X, int64_t Ytot, uint64_t seed) {
	return orc_init_slab(black, white, X, Ytot, 0, seed);
}

/* ------------------------------------------------------------------ neighbour words
 * A "view" is what one device sees: Y rows of the source colour starting at global row row_base, plus the two
 * rows just outside it (the reference reads them from the neighbouring GPU through managed memory,
 * optimized/main.cu:413-428, :1637-1642).  For the whole lattice the halo rows alias the array itself
 * (periodic wrap).  With sub-lattices (slY < total rows) the wrap happens inside the view instead.
 */
typedef struct {
	const uint64_t *src, *halo_top, *halo_bot;
	int64_t Y;        /* rows in the view */
	int64_t row_base; /* global row of view row 0 */
	int64_t slV, slY; /* periodic extents: vectors per row, rows; slY <= 0 means "use the halo rows" */
} orc_view;

static inline const uint64_t *view_row(const orc_view *v, const orc_geom *g, int64_t r) {
	if (r < 0) return v->halo_top;
	if (r >= v->Y) return v->halo_bot;
	return v->src + r*g->lld;
}

/* For destination colour `color`, view row i, vector `col`: nibble-wise neighbour-up counts of words (x,y).
 * jrow (may be NULL): the coupling words the reference passes as jDst for this row (:575-618): a set bit flips the
 * neighbour's contribution.  Bits per nibble: 0x8 up, 0x4 down, 0x2 left, 0x1 right (:588).
 */
static inline void neighbour_sums(const orc_view *v, const orc_geom *g, int color, int64_t i, int64_t col, const uint64_t *jrow, uint64_t sum[2]) {
	int64_t iu = i - 1, id = i + 1;
	if (v->slY > 0) {
		iu = (i % v->slY) == 0 ? i + v->slY - 1 : i - 1;          /* :414 */
		id = ((i + 1) % v->slY) == 0 ? i + 1 - v->slY : i + 1;    /* :422 */
	}
	const uint64_t *ru = view_row(v, g, iu), *rc = view_row(v, g, i), *rd = view_row(v, g, id);
	uint64_t ctx = rc[2*col], cty = rc[2*col + 1];
	uint64_t upx = ru[2*col], upy = ru[2*col + 1], dwx = rd[2*col], dwy = rd[2*col + 1];
	uint64_t sdx, sdy;
	const int64_t gi = v->row_base + i;
	const int readBack = (color == ORC_BLACK) ? !(gi & 1) : (int)(gi & 1); /* :542 */
	const int64_t slV = v->slV;
	if (readBack) {
		const int64_t cl = (col % slV) == 0 ? col + slV - 1 : col - 1; /* :433 */
		const uint64_t ly = rc[2*cl + 1];
		sdx = (ctx << 4) | (ly >> 60);   /* :560 */
		sdy = (cty << 4) | (ctx >> 60);  /* :561 */
	} else {
		const int64_t cr = ((col + 1) % slV) == 0 ? col + 1 - slV : col + 1; /* :441 */
		const uint64_t rx = rc[2*cr];
		sdy = (cty >> 4) | (rx << 60);   /* :569 */
		sdx = (ctx >> 4) | (cty << 60);  /* :570 */
	}
	if (jrow) {                          /* :575-618 */
		const uint64_t jx = jrow[2*col], jy = jrow[2*col + 1];
		upx ^= (jx & 0x8888888888888888ull) >> 3; upy ^= (jy & 0x8888888888888888ull) >> 3;
		dwx ^= (jx & 0x4444444444444444ull) >> 2; dwy ^= (jy & 0x4444444444444444ull) >> 2;
		if (readBack) {
			sdx ^= (jx & 0x2222222222222222ull) >> 1; sdy ^= (jy & 0x2222222222222222ull) >> 1;
			ctx ^= (jx & 0x1111111111111111ull);      cty ^= (jy & 0x1111111111111111ull);
		} else {
			ctx ^= (jx & 0x2222222222222222ull) >> 1; cty ^= (jy & 0x2222222222222222ull) >> 1;
			sdx ^= (jx & 0x1111111111111111ull);      sdy ^= (jy & 0x1111111111111111ull);
		}
	}
	sum[0] = ctx + upx + dwx + sdx;  /* :623-635, nibble sums never carry (max 4) */
	sum[1] = cty + upy + dwy + sdy;
}

/* ------------------------------------------------------------------ hamiltInitB_k, optimized/main.cu:153-212
 * Random coupling bits for the black array: for every nibble, bits l = 0..3 (<right, left, down, up> = 0x1,0x2,0x4,0x8,
 * :588) are set with probability prob; draws in the order k (nibble), l (bit), word x then word y (:189-200);
 * generator curand_init(seed, tid, 0) with the caller passing seed+1 (:1734).
 */
void orc_ham_init_black(uint64_t *hamB, int64_t X, int64_t Y, int64_t row_base, uint64_t seed, float prob) {
	orc_geom g;
	if (geom(X, Y, &g)) return;
	#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < Y; i++) {
		const int64_t gi = row_base + i;
		const int64_t by = gi / BLK_Y; const int ty = (int)(gi % BLK_Y);
		for (int64_t bx = 0; bx < g.gx; bx++) {
			for (int tx = 0; tx < BLK_X; tx++) {
				const uint32_t tid = (uint32_t)((by*g.gx + bx)*BLK_X*BLK_Y + ty*BLK_X + tx);
				orc_gen st;
				gen_init(&st, seed, tid, 0);
				for (int j = 0; j < VEC_PER_THREAD; j++) {
					uint64_t x = 0, y = 0;
					for (int k = 0; k < 64; k += 4) {
						for (int l = 0; l < 4; l++) {
							if (u01(gen_next(&st)) < prob) x |= 1ull << (k + l);
							if (u01(gen_next(&st)) < prob) y |= 1ull << (k + l);
						}
					}
					const int64_t col = bx*BLK_X*VEC_PER_THREAD + tx + j*BLK_X;
					hamB[i*g.lld + 2*col]     = x;
					hamB[i*g.lld + 2*col + 1] = y;
				}
			}
		}
	}
}

/* hamiltInitW_k, optimized/main.cu:214-331: every black vector scatters its four coupling bits to the white sites at
 * the other end of the bonds (atomicOr there; plain |= in this sequential restatement).  Whole lattice (all rows in
 * one array); xsl in 128-bit vectors, ysl in rows.  hamW must be zero on entry.
 */
void orc_ham_init_white(const uint64_t *hamB, uint64_t *hamW, int64_t X, int64_t Ytot, int64_t XSL, int64_t YSL) {
	orc_geom g;
	if (geom(X, Ytot, &g)) return;
	if (XSL <= 0) XSL = X;
	if (YSL <= 0) YSL = Ytot;
	const int64_t xsl = (XSL/2)/NIB/2, ysl = YSL;
	const uint64_t M8 = 0x8888888888888888ull, M4 = 0x4444444444444444ull, M2 = 0x2222222222222222ull, M1 = 0x1111111111111111ull;
	for (int64_t yoff = 0; yoff < Ytot; yoff++) {
		const int64_t upOff = (yoff % ysl) == 0 ? yoff + ysl - 1 : yoff - 1;       /* :306 */
		const int64_t dwOff = ((yoff + 1) % ysl) == 0 ? yoff - ysl + 1 : yoff + 1; /* :307 */
		const int readBack = !(yoff % 2);                                          /* :256 */
		for (int64_t xoff = 0; xoff < g.vecs; xoff++) {
			const uint64_t mx = hamB[yoff*g.lld + 2*xoff], my = hamB[yoff*g.lld + 2*xoff + 1];
			const uint64_t upx = (mx & M8) >> 1, upy = (my & M8) >> 1;   /* :245-246 */
			const uint64_t dwx = (mx & M4) << 1, dwy = (my & M4) << 1;   /* :248-249 */
			uint64_t ctx, cty, sdx, sdy;
			if (!readBack) {                                             /* :260-276 */
				ctx = (mx & M2) >> 1;
				cty = (my & M2) >> 1;
				ctx |= (mx & M1) << 5;
				cty |= (mx & M1) >> 59;
				cty |= (my & M1) << 5;
				sdx = (my & M1) >> 59;
				sdy = 0;
			} else {                                                     /* :278-294 */
				ctx = (mx & M1) << 1;
				cty = (my & M1) << 1;
				cty |= (my & M2) >> 5;
				ctx |= (my & M2) << 59;
				ctx |= (mx & M2) >> 5;
				sdy = (mx & M2) << 59;
				sdx = 0;
			}
			hamW[yoff*g.lld + 2*xoff]      |= ctx;  hamW[yoff*g.lld + 2*xoff + 1]  |= cty;
			hamW[upOff*g.lld + 2*xoff]     |= upx;  hamW[upOff*g.lld + 2*xoff + 1] |= upy;
			hamW[dwOff*g.lld + 2*xoff]     |= dwx;  hamW[dwOff*g.lld + 2*xoff + 1] |= dwy;
			const int64_t sideOff = readBack ? ((xoff % xsl) == 0 ? xoff + xsl - 1 : xoff - 1)
			                                 : (((xoff + 1) % xsl) == 0 ? xoff - xsl + 1 : xoff + 1); /* :322-323 */
			hamW[yoff*g.lld + 2*sideOff]     |= sdx;
			hamW[yoff*g.lld + 2*sideOff + 1] |= sdy;
		}
	}
}

/* ------------------------------------------------------------------ spinUpdateV_2D_k, optimized/main.cu:463-670
 * Updates view rows [r_lo, r_hi) (multiples of 16 are NOT required here: the reference launches whole 16-row
 * blocks, but every thread is independent, so any row range gives the same result for those rows).
 */
static void update_rows(uint64_t *dst, const orc_view *v, const orc_geom *g, const uint64_t *jdst, uint64_t seed, int it, int color,
                        const float tab[10], int64_t r_lo, int64_t r_hi) {
	#pragma omp parallel for schedule(static)
	for (int64_t i = r_lo; i < r_hi; i++) {
		const int64_t gi = v->row_base + i;
		const int64_t by = gi / BLK_Y; const int ty = (int)(gi % BLK_Y);
		for (int64_t bx = 0; bx < g->gx; bx++) {
			for (int tx = 0; tx < BLK_X; tx++) {
				const uint32_t tid = (uint32_t)((by*g->gx + bx)*BLK_X*BLK_Y + ty*BLK_X + tx); /* :514 */
				orc_gen st;
				gen_init(&st, seed, tid, (uint64_t)(2*NIB)*VEC_PER_THREAD*(2*(uint64_t)it + color)); /* :621 */
				for (int j = 0; j < VEC_PER_THREAD; j++) {
					const int64_t col = bx*BLK_X*VEC_PER_THREAD + tx + j*BLK_X;
					uint64_t sum[2];
					neighbour_sums(v, g, color, i, col, jdst ? jdst + i*g->lld : NULL, sum);
					uint64_t me[2] = { dst[i*g->lld + 2*col], dst[i*g->lld + 2*col + 1] };
					for (int z = 0; z < 64; z += 4) {   /* :637-660: x word first, then y word */
						for (int w = 0; w < 2; w++) {
							const int s = (int)((me[w] >> z) & 0xF);
							const int n = (int)((sum[w] >> z) & 0xF);
							if (u01(gen_next(&st)) <= tab[s*5 + n]) me[w] ^= 1ull << z;
						}
					}
					dst[i*g->lld + 2*col]     = me[0];
					dst[i*g->lld + 2*col + 1] = me[1];
				}
			}
		}
	}
}

/* whole lattice (all devices' rows in one array), optional sub-lattices */
int orc_update_color(uint64_t *black, uint64_t *white, int64_t X, int64_t Ytot, int64_t XSL, int64_t YSL,
                     uint64_t seed, int it, int color, const float tab[10]) {
	orc_geom g;
	if (geom(X, Ytot, &g)) return -1;
	if (XSL <= 0) XSL = X;
	if (YSL <= 0) YSL = Ytot;
	const int64_t slV = (XSL/2)/NIB/2;  /* (XSL/2)/SPIN_X_WORD/2, :1771 */
	if (slV <= 0 || g.vecs % slV || Ytot % YSL) return -2;
	const uint64_t *src = (color == ORC_BLACK) ? white : black;
	uint64_t *dst = (color == ORC_BLACK) ? black : white;
	const orc_view v = { src, src + (Ytot - 1)*g.lld, src, Ytot, 0, slV, YSL };
	update_rows(dst, &v, &g, NULL, seed, it, color, tab, 0, Ytot);
	return 0;
}

/* same with coupling arrays: the reference passes hamW as jDst when it updates BLACK and hamB when it updates WHITE
 * (optimized/main.cu:1774, :1795) -- restated as written. */
int orc_update_color_J(uint64_t *black, uint64_t *white, const uint64_t *hamB, const uint64_t *hamW, int64_t X, int64_t Ytot,
                       int64_t XSL, int64_t YSL, uint64_t seed, int it, int color, const float tab[10]) {
	orc_geom g;
	if (geom(X, Ytot, &g)) return -1;
	if (XSL <= 0) XSL = X;
	if (YSL <= 0) YSL = Ytot;
	const int64_t slV = (XSL/2)/NIB/2;
	if (slV <= 0 || g.vecs % slV || Ytot % YSL) return -2;
	const uint64_t *src = (color == ORC_BLACK) ? white : black;
	uint64_t *dst = (color == ORC_BLACK) ? black : white;
	const uint64_t *jdst = (color == ORC_BLACK) ? hamW : hamB;
	const orc_view v = { src, src + (Ytot - 1)*g.lld, src, Ytot, 0, slV, YSL };
	update_rows(dst, &v, &g, jdst, seed, it, color, tab, 0, Ytot);
	return 0;
}

/* one device's slab: Y rows starting at global row row_base, halo rows supplied by the caller */
int orc_update_color_slab(uint64_t *dst, const uint64_t *src, const uint64_t *halo_top, const uint64_t *halo_bot,
                          int64_t X, int64_t Y, int64_t row_base, uint64_t seed, int it, int color,
                          const float tab[10], int64_t r_lo, int64_t r_hi) {
	orc_geom g;
	if (geom(X, Y, &g) || r_lo < 0 || r_hi > Y || r_lo > r_hi) return -1;
	const orc_view v = { src, halo_top, halo_bot, Y, row_base, g.vecs, 0 };
	update_rows(dst, &v, &g, NULL, seed, it, color, tab, r_lo, r_hi);
	return 0;
}

/* sweeps first_it .. first_it+n-1 (it is 1-based: sweep j uses it=j+1, optimized/main.cu:1763-1799) */
int orc_sweep(uint64_t *black, uint64_t *white, int64_t X, int64_t Ytot, int64_t XSL, int64_t YSL,
              uint64_t seed, int first_it, int n, float temp) {
	float tab[10];
	orc_exp_table(temp, tab);
	for (int it = first_it; it < first_it + n; it++) {
		int rc = orc_update_color(black, white, X, Ytot, XSL, YSL, seed, it, ORC_BLACK, tab);
		if (rc) return rc;
		rc = orc_update_color(black, white, X, Ytot, XSL, YSL, seed, it, ORC_WHITE, tab);
		if (rc) return rc;
	}
	return 0;
}

/* ------------------------------------------------------------------ getMagn_k, optimized/main.cu:701-734 */
void orc_count(const uint64_t *black, const uint64_t *white, int64_t X, int64_t Ytot,
               uint64_t *up, uint64_t *down) {
	const int64_t n = (X/32)*Ytot;
	uint64_t p = 0;
	#pragma omp parallel for reduction(+:p) schedule(static)
	for (int64_t i = 0; i < n; i++) p += (uint64_t)__builtin_popcountll(black[i]) + (uint64_t)__builtin_popcountll(white[i]);
	*up = p;
	*down = (uint64_t)(2*n)*NIB - p;
}

/* ------------------------------------------------------------------ bond sum (build-side observable)
 * A = sum over black sites of the number of (white) neighbours equal to the site.  Every bond of the torus
 * has exactly one black end, so  sum_<ij> s_i s_j = 2A - 2N  and  E/N = -(2A - 2N)/N.
 */
static int64_t bond_rows(const uint64_t *black, const orc_view *v, const orc_geom *g) {
	int64_t A = 0;
	#pragma omp parallel for reduction(+:A) schedule(static)
	for (int64_t i = 0; i < v->Y; i++) {
		for (int64_t col = 0; col < g->vecs; col++) {
			uint64_t sum[2];
			neighbour_sums(v, g, ORC_BLACK, i, col, NULL, sum);
			for (int w = 0; w < 2; w++) {
				const uint64_t me = black[i*g->lld + 2*col + w];
				for (int z = 0; z < 64; z += 4) {
					const int s = (int)((me >> z) & 1), n = (int)((sum[w] >> z) & 0xF);
					A += s ? n : 4 - n;
				}
			}
		}
	}
	return A;
}

int64_t orc_bond_equal(const uint64_t *black, const uint64_t *white, int64_t X, int64_t Ytot,
                       int64_t XSL, int64_t YSL) {
	orc_geom g;
	if (geom(X, Ytot, &g)) return -1;
	if (XSL <= 0) XSL = X;
	if (YSL <= 0) YSL = Ytot;
	const orc_view v = { white, white + (Ytot - 1)*g.lld, white, Ytot, 0, (XSL/2)/NIB/2, YSL };
	return bond_rows(black, &v, &g);
}

int64_t orc_bond_equal_slab(const uint64_t *black, const uint64_t *white, const uint64_t *halo_top,
                            const uint64_t *halo_bot, int64_t X, int64_t Y, int64_t row_base) {
	orc_geom g;
	if (geom(X, Y, &g)) return -1;
	const orc_view v = { white, halo_top, halo_bot, Y, row_base, g.vecs, 0 };
	return bond_rows(black, &v, &g);
}

/* ------------------------------------------------------------------ getCorr2D_k + computeCorr, optimized/main.cu:870-965, :1072-1138
 * sums[j-1] = sum over all sites (row r, lattice column c) of  [s(r,c) == s(r,c+j) ? +1 : -1]  (columns periodic in X)
 *                                                           + [s(r,c) == s(r+j,c) ? +1 : -1]  (rows periodic in Ytot)
 * for j = 1..ncorr.  Lattice column c of row r is colour-site c/2 of the white array when (r ^ c) is odd, of the black
 * array otherwise (:928-929).  The reference accumulates these +-1 in doubles (exact) and prints sums/(2*X*Y*ndev).
 */
static inline int spin_at(const uint64_t *black, const uint64_t *white, int64_t lld, int64_t r, int64_t c) {
	const uint64_t *a = ((r ^ c) & 1) ? white : black;
	const int64_t k = c >> 1;
	return (int)((a[r*lld + k/NIB] >> (4*(k % NIB))) & 0xF);
}

void orc_corr(const uint64_t *black, const uint64_t *white, int64_t X, int64_t Ytot, int64_t XSL, int64_t YSL, int ncorr, int64_t *sums) {
	/* XSL/YSL > 0: getCorr2DRepl_k (optimized/main.cu:967-1070), both wraps stay inside the site's own sub-lattice */
	const int64_t lld = X/32;
	if (XSL <= 0) XSL = X;
	if (YSL <= 0) YSL = Ytot;
	for (int j = 1; j <= ncorr; j++) {
		int64_t acc = 0;
		#pragma omp parallel for reduction(+:acc) schedule(static)
		for (int64_t r = 0; r < Ytot; r++) {
			const int64_t rv = (r + j >= (r/YSL + 1)*YSL) ? r + j - YSL : r + j;   /* :1038, :934 */
			for (int64_t c = 0; c < X; c++) {
				const int64_t c0 = (c/XSL)*XSL;
				const int me = spin_at(black, white, lld, r, c);
				acc += (me == spin_at(black, white, lld, r, c0 + (c - c0 + j) % XSL)) ? 1 : -1;
				acc += (me == spin_at(black, white, lld, rv, c)) ? 1 : -1;
			}
		}
		sums[j - 1] = acc;
	}
}

/* ------------------------------------------------------------------ dumpLattice text, optimized/main.cu:1140-1209
 * One text row per lattice row, one hex digit per spin, colours interleaved by row parity.
 * Writes rows [row0,row0+nrows) into buf (X chars + '\n' per row); returns bytes written.
 */
int64_t orc_dump_rows(const uint64_t *black, const uint64_t *white, int64_t X, int64_t row0, int64_t nrows, char *buf) {
	static const char hex[] = "0123456789ABCDEF";
	const int64_t lld = X/32;
	char *p = buf;
	for (int64_t i = row0; i < row0 + nrows; i++) {
		for (int64_t j = 0; j < lld; j++) {
			const uint64_t b = black[i*lld + j], w = white[i*lld + j];
			for (int k = 0; k < 64; k += 4) {
				if (i & 1) { *p++ = hex[(w >> k) & 0xF]; *p++ = hex[(b >> k) & 0xF]; }
				else       { *p++ = hex[(b >> k) & 0xF]; *p++ = hex[(w >> k) & 0xF]; }
			}
		}
		*p++ = '\n';
	}
	return p - buf;
}

/* ------------------------------------------------------------------ closed-form draw lookup
 * Raw 32-bit Philox output the reference consumes for colour `color`, iteration `it`, global row i,
 * 64-bit word index q (0..lld-1) and nibble z.  Used by tests to cross-check the closed form the HIP
 * kernels rely on against the sequential generators above.
 */
uint32_t orc_site_draw(int64_t X, uint64_t seed, int it, int color, int64_t i, int64_t q, int z) {
	const int64_t gx = X/2048;
	const int64_t v = q >> 1; const int w = (int)(q & 1);
	const uint32_t tid = (uint32_t)(((i/16)*gx + v/32)*256 + (i%16)*16 + (v%16));
	const int j = (int)((v%32)/16);
	const uint64_t blk = 16ull*(2ull*(uint64_t)it + (uint64_t)color) + 8ull*(uint64_t)j + (uint64_t)(z/2);
	const uint32_t ctr[4] = { (uint32_t)blk, (uint32_t)(blk >> 32), tid, 0 };
	const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
	uint32_t out[4];
	philox_block(ctr, key, out);
	return out[2*(z & 1) + w];
}

/* thread control for the timed CPU baseline */
void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int orc_max_threads(void) { return omp_get_max_threads(); }
