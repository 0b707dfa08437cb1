/*
 * ising_basic_main.c -- command-line front of the CPU baseline with the surface of the reference's basic_python/ising_basic.py
 * (BASELINE config 1: 1024 x 1024, alpha 1, seed 1234, 1000 sweeps).
 *
 * TEST / BASELINE INFRASTRUCTURE ONLY -- never linked into the product.
 *
 * Flags as /root/reference/basic_python/ising_basic.py:43-53: -x/--lattice-n rows, -y/--lattice-m columns, -w/--nwarmup,
 * -n/--niters, -a/--alpha, -s/--seed, -o/--write-lattice (-c/--use-common-seed is accepted: one process has one stream);
 * -t/--threads picks the OpenMP team (the reference runs on GPUs under MPI; here "nGPUs" is printed as 0).  The transcript
 * is the reference's (:208-259): "Starting warmup..." / "Starting trial iterations..." / "Completed i/n iterations..."
 * every 1000, then the REPORT block, and with -o "final_rank0.txt" in np.savetxt's '%d' form (:137-151).
 * The algorithm is basic_cpu.c's restatement (parity unpinned: same distribution, not the reference's cuRAND stream).
 */
#include <getopt.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void basic_init(int8_t *black, int8_t *white, int64_t n, int64_t m, uint64_t seed, float *scratch);
void basic_sweeps(int8_t *black, int8_t *white, int64_t n, int64_t m, float alpha, uint64_t seed, int64_t it0, int64_t niters, float *scratch);
void basic_observables(const int8_t *black, const int8_t *white, int64_t n, int64_t m, int64_t *msum, int64_t *bonds);

#define TCRIT 2.26918531421 /* ising_basic.py:33 */

static double now(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Python's str(float): shortest representation that round-trips */
static void py_float(double v, char *buf, size_t len) {
	for (int prec = 1; prec <= 17; prec++) {
		snprintf(buf, len, "%.*g", prec, v);
		if (strtod(buf, NULL) == v) break;
	}
	if (!strpbrk(buf, ".en")) strncat(buf, ".0", len - strlen(buf) - 1); /* 1 -> 1.0 (no exponent, no inf/nan) */
}

int main(int argc, char **argv) {
	long n = 40 * 128, m = 40 * 128, nwarmup = 100, niters = 1000, seed = 1234; /* defaults :43-49 */
	double alpha = 0.1;
	int write_lattice = 0, threads = 0;
	static struct option opts[] = {{"lattice-n", required_argument, 0, 'x'}, {"lattice-m", required_argument, 0, 'y'}, {"nwarmup", required_argument, 0, 'w'},
	                               {"niters", required_argument, 0, 'n'},    {"alpha", required_argument, 0, 'a'},     {"seed", required_argument, 0, 's'},
	                               {"write-lattice", no_argument, 0, 'o'},   {"use-common-seed", no_argument, 0, 'c'}, {"threads", required_argument, 0, 't'},
	                               {0, 0, 0, 0}};
	for (int ch; (ch = getopt_long(argc, argv, "x:y:w:n:a:s:oct:", opts, NULL)) != -1;) {
		switch (ch) {
		case 'x': n = atol(optarg); break;
		case 'y': m = atol(optarg); break;
		case 'w': nwarmup = atol(optarg); break;
		case 'n': niters = atol(optarg); break;
		case 'a': alpha = atof(optarg); break;
		case 's': seed = atol(optarg); break;
		case 'o': write_lattice = 1; break;
		case 'c': break;
		case 't': threads = atoi(optarg); break;
		default: return EXIT_FAILURE;
		}
	}
	if (m % 2 != 0) { fprintf(stderr, "lattice_m must be an even value. Aborting.\n"); return EXIT_FAILURE; }                 /* :56-57 */
	if (n % 2 != 0) { fprintf(stderr, "Slab width (lattice_n / nGPUs) must be an even value. Aborting.\n"); return EXIT_FAILURE; } /* :60-61 */
	if (n <= 0 || m <= 0 || nwarmup < 0 || niters < 0) { fprintf(stderr, "bad sizes\n"); return EXIT_FAILURE; }
#ifdef _OPENMP
	if (threads > 0) omp_set_num_threads(threads);
	threads = omp_get_max_threads();
#else
	threads = 1;
#endif
	const int64_t h = (int64_t)n * (m / 2);
	int8_t *black = malloc((size_t)h), *white = malloc((size_t)h);
	float *scratch = malloc((size_t)h * sizeof(float));
	if (!black || !white || !scratch) { fprintf(stderr, "out of memory\n"); return EXIT_FAILURE; }
	basic_init(black, white, n, m, (uint64_t)seed, scratch);

	printf("Starting warmup...\n"); /* :211-216 */
	fflush(stdout);
	basic_sweeps(black, white, n, m, (float)alpha, (uint64_t)seed, 0, nwarmup, scratch);
	printf("Starting trial iterations...\n"); /* :227-236 */
	fflush(stdout);
	const double t0 = now();
	for (long i = 0; i < niters; i++) {
		basic_sweeps(black, white, n, m, (float)alpha, (uint64_t)seed, nwarmup + i, 1, scratch);
		if (i % 1000 == 0) { printf("Completed %ld/%ld iterations...\n", i + 1, niters); fflush(stdout); }
	}
	const double t = now() - t0;

	int64_t msum = 0, bonds = 0;
	basic_observables(black, white, n, m, &msum, &bonds);
	const double mavg = (double)msum / ((double)n * (double)m);
	char fa[40], ft[40], fu[40], fm[40];
	py_float(alpha, fa, sizeof fa);
	py_float(t, ft, sizeof ft);
	py_float(((double)n * (double)m * (double)niters) / t * 1e-9, fu, sizeof fu);
	py_float(fabs(mavg), fm, sizeof fm);
	printf("REPORT:\n"); /* :246-256 */
	printf("\tnGPUs: 0\n");
	printf("\ttemperature: %s * %.11f\n", fa, TCRIT);
	printf("\tseed: %ld\n", seed);
	printf("\twarmup iterations: %ld\n", nwarmup);
	printf("\ttrial iterations: %ld\n", niters);
	printf("\tlattice dimensions: %ld x %ld\n", n, m);
	printf("\telapsed time: %s sec\n", ft);
	printf("\tupdates per ns: %s\n", fu);
	printf("\taverage magnetism (absolute): %s\n", fm);
	/* not in the reference: what the line above was measured on, and the energy per spin (-sqrt(2) at T_c for an infinite lattice) */
	printf("\thost threads: %d\n", threads);
	printf("\tenergy per spin: %.6f\n", -(double)bonds / ((double)n * (double)m));
	fflush(stdout);

	if (write_lattice) { /* write_lattice("final", ...), :137-151: colours interleaved by row parity, np.savetxt(fmt='%d') */
		printf("Writing lattice to final_rank0.txt...\n");
		FILE *f = fopen("final_rank0.txt", "w");
		if (!f) { perror("final_rank0.txt"); return EXIT_FAILURE; }
		const int64_t mh = m / 2;
		for (int64_t i = 0; i < n; i++) {
			for (int64_t j = 0; j < mh; j++) {
				const int b = black[i * mh + j], w = white[i * mh + j];
				if (i % 2) fprintf(f, "%d %d%s", w, b, j + 1 < mh ? " " : "\n"); /* odd rows: white first (:143-145) */
				else fprintf(f, "%d %d%s", b, w, j + 1 < mh ? " " : "\n");
			}
		}
		fclose(f);
	}
	free(black); free(white); free(scratch);
	return 0;
}
