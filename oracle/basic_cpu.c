/*
 * basic_cpu.c -- CPU restatement of the reference's *basic* (byte-per-spin) checkerboard Metropolis
 * algorithm, used ONLY as the reported CPU baseline (bench.py cpu_baseline) and in tests.
 *
 * TEST / BASELINE INFRASTRUCTURE ONLY -- never linked into the product.
 *
 * Follows /root/reference/basic_python/ising_basic.py:
 *   - state: two int8 arrays [n][m/2] (black, white), values +-1 ................ :203-209
 *   - init:  +1 if u > 0.5 else -1 ................................................ :69-71
 *   - per half-sweep: one float32 uniform per site, then update_lattice ........... :154-168, :175-195
 *   - update rule: joff parity rule, nn_sum of 4, flip if u < exp(-2*inv_temp*nn_sum*lij) (float64 exp)
 *     ........................................................................... :106-134
 *   - order black then white; report flips/ns = n*m*niters/t*1e-9 ................ :190-195, :254
 * PARITY UNPINNED: ising_basic.py needs numba-cuda, cupy-cuRAND and mpi4py plus a CUDA device, none of which
 * exist here, and the reference publishes no transcript for it.  The uniforms come from cuRAND's *host*
 * Philox generator there (stream ordering unspecified); here they come from Philox4x32-10 keyed by
 * (seed) with counter (site index, half-sweep index), i.e. the same distribution, not the same stream.
 * Only statistical agreement (|m|, energy per spin near -sqrt(2) at T_c) is claimed.
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>

#define TCRIT 2.26918531421f

static inline void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
	for (int r = 0; r < 10; r++) {
		const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
		const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
		c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
		k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
	}
	out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* fill `n` float32 uniforms in (0,1]; `stream` distinguishes half-sweeps */
static void fill_uniform(float *r, int64_t n, uint64_t seed, uint64_t stream) {
	#pragma omp parallel for schedule(static)
	for (int64_t b = 0; b < (n + 3)/4; b++) {
		uint32_t o[4];
		philox((uint32_t)b, (uint32_t)(b >> 32), (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
		for (int k = 0; k < 4 && 4*b + k < n; k++) r[4*b + k] = (float)o[k] * 0x1p-32f + 0x1p-33f;
	}
}

void basic_init(int8_t *black, int8_t *white, int64_t n, int64_t m, uint64_t seed, float *scratch) {
	const int64_t h = n*(m/2);
	fill_uniform(scratch, h, seed, 0);
	for (int64_t i = 0; i < h; i++) black[i] = scratch[i] > 0.5f ? 1 : -1;
	fill_uniform(scratch, h, seed, 1);
	for (int64_t i = 0; i < h; i++) white[i] = scratch[i] > 0.5f ? 1 : -1;
}

/* ising_basic.py:106-134 */
static void update_half(int8_t *lat, const int8_t *op, const float *rnd, int64_t n, int64_t mh, int is_black, double inv_temp) {
	#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < n; i++) {
		const int64_t ipp = (i + 1 < n) ? i + 1 : 0, inn = (i - 1 >= 0) ? i - 1 : n - 1;
		for (int64_t j = 0; j < mh; j++) {
			const int64_t jpp = (j + 1 < mh) ? j + 1 : 0, jnn = (j - 1 >= 0) ? j - 1 : mh - 1;
			int64_t joff;
			if (is_black) joff = (i % 2) ? jpp : jnn;
			else          joff = (i % 2) ? jnn : jpp;
			const int nn = op[inn*mh + j] + op[i*mh + j] + op[ipp*mh + j] + op[i*mh + joff];
			const int lij = lat[i*mh + j];
			const double acc = exp(-2.0 * inv_temp * nn * lij);
			if (rnd[i*mh + j] < acc) lat[i*mh + j] = (int8_t)-lij;
		}
	}
}

/* runs `niters` full sweeps starting at sweep index it0 (for distinct random streams) */
void basic_sweeps(int8_t *black, int8_t *white, int64_t n, int64_t m, float alpha, uint64_t seed,
                  int64_t it0, int64_t niters, float *scratch) {
	const double inv_temp = 1.0 / (double)(alpha * TCRIT);
	const int64_t mh = m/2;
	for (int64_t it = it0; it < it0 + niters; it++) {
		fill_uniform(scratch, n*mh, seed, 2 + 2*(uint64_t)it);
		update_half(black, white, scratch, n, mh, 1, inv_temp);
		fill_uniform(scratch, n*mh, seed, 3 + 2*(uint64_t)it);
		update_half(white, black, scratch, n, mh, 0, inv_temp);
	}
}

/* sum of spins and sum over bonds of s_i s_j (each bond once: every bond has exactly one black end) */
void basic_observables(const int8_t *black, const int8_t *white, int64_t n, int64_t m, int64_t *msum, int64_t *bonds) {
	const int64_t mh = m/2;
	int64_t s = 0, b = 0;
	#pragma omp parallel for reduction(+:s,b) schedule(static)
	for (int64_t i = 0; i < n; i++) {
		const int64_t ipp = (i + 1 < n) ? i + 1 : 0, inn = (i - 1 >= 0) ? i - 1 : n - 1;
		for (int64_t j = 0; j < mh; j++) {
			const int64_t jpp = (j + 1 < mh) ? j + 1 : 0, jnn = (j - 1 >= 0) ? j - 1 : mh - 1;
			const int64_t joff = (i % 2) ? jpp : jnn;
			s += black[i*mh + j] + white[i*mh + j];
			b += black[i*mh + j] * (white[inn*mh + j] + white[i*mh + j] + white[ipp*mh + j] + white[i*mh + joff]);
		}
	}
	*msum = s; *bonds = b;
}
