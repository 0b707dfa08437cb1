// cuising_main.cpp -- command-line front with the reference's surface (optimized/main.cu main(), :1230-1926):
// same flags, same defaults and the same transcript lines, over the C-ABI of libising_hip.so.  No HIP calls here.
//
// Supported: -x -y -n/--nit -s/--seed -d/--devs -a/--alpha -t/--temp -p/--print -e/--exppr -m/--magn -u/--update
//            -o/--out -h, plus --energy (build-side addition: prints the energy per spin next to each
//            magnetisation line) and --devmap a,b,.. (place slab k on device devmap[k]; lets a 1-GPU box run -d N), --layout ballot|dense|nibble.
//            --xsl/--ysl (independent periodic sub-lattices, optimized/main.cu:1423-1462),
//            -c/--corr (two-point correlations file, optimized/main.cu:1072-1138).
//            -J <PROB> (random anti-ferromagnetic bonds, optimized/main.cu:153-331, :575-618).
// Build-side additions: --tsweep T0,T1,dT[,nequil[,nmeas[,stride]]] (temperature-sweep driver with <|m|>, <m^2>,
//            susceptibility, Binder cumulant, energy and specific heat per point; --tsweep-anneal, --tsweep-out PREFIX,
//            --tsweep-replicas K: K temperature points per batched launch on one GPU, 0 = by lattice size; --tsweep-no-batch: two
//            points side by side on streams of their own instead, round 2's form; --tsweep-cold: every point starts from the
//            ordered lattice -- the start that equilibrates below T_c, where a random one coarsens for ages),
//            --J-symmetric (with -J): the two coupling arrays change places after they are drawn, so that every colour's update reads
//            its own sites' bonds -- the +-J model with J_ij = J_ji; the reference's launches pair them the other way round,
//            --checkpoint FILE / --resume FILE (binary checkpoint, ising_ring_checkpoint_*), --transport copy|rccl.
#include "../../include/ising_hip.h"

#include <getopt.h>
#include <unistd.h>

#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int SPIN_X_WORD = 16;        // optimized/main.cu:1243
constexpr int X_MULT = 2048;           // 2*SPIN_X_WORD*2*BLOCK_X*BMULT_X, optimized/main.cu:1412
constexpr int Y_MULT = 16;             // BLOCK_Y*BMULT_Y, optimized/main.cu:1417
constexpr float ALPHA_DEF = 0.1f;      // optimized/main.cu:43
constexpr float MIN_TEMP = 0.05f * ISING_CRIT_TEMP; // optimized/main.cu:44
constexpr double TGT_MAGN_MAX_DIFF = 1.0E-3;         // optimized/main.cu:65
constexpr int MAX_EXP_TIME = 200, MIN_EXP_TIME = 152; // optimized/main.cu:67-68
constexpr int NUMIT_DEF = 1;
constexpr int MAX_CORR_LEN = 128;                     // optimized/main.cu:70

[[noreturn]] void die(const char *what) {
	fprintf(stderr, "%s: %s\n", what, ising_last_error());
	exit(EXIT_FAILURE);
}
#define CHECK(call) do { if ((call) != ISING_OK) die(#call); } while (0)

void usage(const char *pname) {
	const char *bname = strrchr(pname, '/');
	bname = bname ? bname + 1 : pname;
	fprintf(stdout,
	        "Usage: %s [options]\n"
	        "options:\n"
	        "\t-x|--x <HORIZ_DIM>     horizontal lattice dimension per GPU, multiple of %d\n"
	        "\t-y|--y <VERT_DIM>      vertical lattice dimension per GPU, multiple of %d\n"
	        "\t-n|--nit <NSTEPS>      number of iterations (default %d)\n"
	        "\t-d|--devs <NUM_DEVICES> number of GPUs, devices [0, NUM_DEVS-1] (default 1)\n"
	        "\t-s|--seed <SEED>       random seed (default %llu; 0 = random)\n"
	        "\t-a|--alpha <ALPHA>     temperature in T_CRIT units (default %f)\n"
	        "\t-t|--temp <TEMP>       absolute temperature; wins over -a (default %f)\n"
	        "\t-p|--print <STAT_FREQ> print magnetization every STAT_FREQ iterations\n"
	        "\t-e|--exppr             print magnetization at time steps 2^(x/4)\n"
	        "\t-m|--magn <TGT_MAGN>   stop when the magnetization reaches TGT_MAGN (needs -p or -e)\n"
	        "\t-u|--update <STEP,FREQ> add STEP to the temperature every FREQ iterations\n"
	        "\t-o|--out               dump the lattice whenever the magnetization is printed\n"
	        "\t   --energy            also print the energy per spin (not in the reference)\n"
	        "\t   --devmap <a,b,...>  device ordinal of each slab (default 0..NUM_DEVS-1)\n"
	        "\t   --layout <ballot|dense|nibble> device layout of the spin arrays: 1 bit/spin (two bit orders; default:\n"
	        "\t                       chosen by the library) or the reference's 4 bit/spin\n"
	        "\t   --xsl <HORIZ_SUB_DIM> horizontal sub-lattice dimension (divisor of -x, multiple of %d)\n"
	        "\t   --ysl <VERT_SUB_DIM>  vertical sub-lattice dimension (divisor of -y, multiple of %d)\n"
	        "\t-c|--corr              append the 128 two-point correlations to corr_{Y}x{X}_T_{TEMP}_{SEED} at every print\n"
	        "\t-J|--J <PROB>          probability [0.0-1.0] that a bond is anti-ferromagnetic (default 0.0)\n\n",
	        bname, X_MULT, Y_MULT, NUMIT_DEF, (unsigned long long)ISING_SEED_DEF, ALPHA_DEF, ALPHA_DEF * ISING_CRIT_TEMP, X_MULT, Y_MULT);
	exit(EXIT_SUCCESS);
}

// The -e series (optimized/main.cu:1211-1228, consumed at :1827): sweep indices (0-based; printed as index + 1) after which the magnetisation is
// reported.  The first one is MIN_EXP_TIME = 152 (iteration 153).  From there the ladder L_k = rint(2^(k/4)), k = 0, 1, 2, ... is climbed, and a rung
// becomes a print point when it is at least twice the last print point (305, 610, 1219, 2897, ... as iterations).  The climb stops behind the first
// rung at or past the run's length -- that rung may still have become a point -- or after `nsteps` rungs; MAX_EXP_TIME points at most.
std::vector<unsigned long long> exp_print_points(unsigned long long nsteps) {
	std::vector<unsigned long long> pts{(unsigned long long)MIN_EXP_TIME};
	unsigned long long rung = 0;
	for (unsigned long long k = 0; k < nsteps && rung < nsteps; ++k) {
		rung = (unsigned long long)rint(pow(2.0, (double)k / 4.0)); // (the reference's expression: the rounding must agree to the last rung)
		if (rung >= 2 * pts.back() && pts.size() < (size_t)MAX_EXP_TIME) pts.push_back(rung);
	}
	return pts;
}

struct Ring {
	std::vector<ising_ctx *> ctx;
	int n() const { return (int)ctx.size(); }

	void count(unsigned long long *up, unsigned long long *dw) {
		uint64_t u = 0, d = 0;
		CHECK(ising_ring_count(ctx.data(), n(), &u, &d));
		*up = u; *dw = d;
	}
	long long bond_equal() {
		int64_t A = 0;
		CHECK(ising_ring_bond_equal(ctx.data(), n(), &A));
		return A;
	}
	double energy(size_t nspins) { return -(2.0 * (double)bond_equal() - 2.0 * (double)nspins) / (double)nspins; }
	void dump(const char *prefix) {
		for (ising_ctx *c : ctx) CHECK(ising_dump_text(c, prefix));
	}
};

// ---- temperature sweep (BASELINE config 5).  Every measurement is a pair of exact integers: M = up - down
// and the bond sum E = 2N - 2A (A = ising_bond_equal: sum over black sites of aligned neighbours).  The moments are
// accumulated exactly (128-bit integers; M^4 as long double of the exact M^2) and only the final ratios are floating
// point:  chi = N (<m^2> - <|m|>^2) / T,  U4 = 1 - <m^4> / (3 <m^2>^2),  Cv = N (<e^2> - <e>^2) / T^2.
struct TsweepSpec {
	double t0 = 0, t1 = 0, dt = 0;
	int nequil = 16, nmeas = 16, stride = 1;
	bool anneal = false;
	const char *out = nullptr;
	int replicas = 0; // temperature points simulated side by side (fresh-start mode, one device); 0 = by lattice size
	int chains = 1;        // --tsweep-chains K: K independent lattices per temperature (seeds seed .. seed + K - 1): means with standard errors
	bool J_symmetric = false; // --J-symmetric (with -J)
	bool cold = false;     // --tsweep-cold: every point starts from the ordered lattice (all spins up) instead of the random one
	bool no_batch = false; // --tsweep-no-batch: the points side by side on streams of their own instead of batched launches (A/B)
};

std::string i128_str(__int128 v) {
	if (v == 0) return "0";
	const bool neg = v < 0;
	unsigned __int128 u = neg ? -(unsigned __int128)v : (unsigned __int128)v;
	std::string r;
	while (u) { r.insert(r.begin(), (char)('0' + (int)(u % 10))); u /= 10; }
	return neg ? "-" + r : r;
}

struct Moments {
	__int128 sM = 0, sAbsM = 0, sM2 = 0, sE = 0, sE2 = 0;
	long double sM4 = 0;
	int n = 0;
	void add(long long M, long long E) {
		const __int128 m2 = (__int128)M * M;
		sM += M; sAbsM += M < 0 ? -M : M; sM2 += m2; sM4 += (long double)m2 * (long double)m2;
		sE += E; sE2 += (__int128)E * E; n++;
	}
};

// Fresh-start mode on one device simulates several temperature points side by side: one context per replica, each on a
// stream of its own (ising_use_private_stream), launches interleaved 32 sweeps at a time.  A lattice of 8192^2 alone
// fills 70 % of an MI355X, two or three of them 90 %; and while the host reads one replica's counts the
// others keep the GPU busy.  Every point's series is what a run of its own gives (tests/test_gpu_tsweep_ckpt.py).
int run_tsweep(Ring &ring, const ising_config &base, const TsweepSpec &ts, size_t nspins, bool useJ) {
	const int ndev = ring.n();
	const int npts = (int)floor((ts.t1 - ts.t0) / ts.dt + 1e-9) + 1;
	// --tsweep-chains K: every temperature is simulated K times from seeds seed .. seed + K - 1 -- independent lattices and random
	// numbers --, so that the spread of the K averages is an honest standard error (the points of ONE chain share their random
	// numbers across T).  A job = (point, chain); the K chains of a point always sit in the same group of lattices.
	const int K = std::max(1, ts.chains);
	if (K > 1 && (ndev != 1 || ts.anneal)) { fprintf(stderr, "error: --tsweep-chains needs one device and fresh starts\n"); exit(EXIT_FAILURE); }
	const int njobs = npts * K;
	int nrep = 1;
	// Fresh-start mode on one device runs the temperature points as a BATCH: lattices of one shape share the tickets of every
	// fused launch (ising_batch_sweep) and one launch measures all of them -- 8192^2 alone fills 70 % of an MI355X, 31 of them
	// in one launch run at the large-lattice rate.  As many points at a time as fit 2^35 spins (4 GiB at 1 bit per spin), 64
	// at most; --tsweep-replicas K forces K.  Small lattices (the quad path: up to ~2^26 spins) batch as well from round 6 on -- one launch
	// per pass of all of them, the measurements inside the passes.  Lattices no batch can carry (-J) run two at a time on streams of their own, as in round 2.
	bool batched = false;
	if (ndev == 1 && !ts.anneal) {
		const int fit = (int)std::max<unsigned long long>(1, std::min<unsigned long long>(64, (1ull << 35) / nspins));
		nrep = ts.replicas > 0 ? ts.replicas * K : fit;
		nrep = std::max(K, std::min(nrep, njobs) / K * K); // (whole points: a multiple of K)
		batched = !useJ && !ts.no_batch;
		if (!batched && ts.replicas == 0 && K == 1) nrep = std::min(nrep, nspins < (1ull << 29) ? 2 : 1); // (streams of their own: two fill the chip)
	}
	printf("\nTemperature sweep: %d points, T = %f .. %f step %f, %d equilibration + %d x %d measurement sweeps per point, %s\n",
	       npts, ts.t0, ts.t0 + (npts - 1) * ts.dt, ts.dt, ts.nequil, ts.nmeas, ts.stride, ts.anneal ? (ts.cold ? "annealing from the ordered lattice" : "annealing") : (ts.cold ? "ordered start per point" : "fresh start per point"));
	// (every point starts from the same seed: the points share their initial lattice and their random numbers, so the curves'
	// statistical errors are correlated across T -- common random numbers; an annealing run chains the points instead)
	if (!ts.anneal && K == 1) printf("Temperature sweep: all points use seed %llu (common random numbers across T)\n", (unsigned long long)base.seed);
	if (K > 1) printf("Temperature sweep: %d independent chains per point (seeds %llu .. %llu): mean +- standard error of the chains' averages\n", K,
	                  (unsigned long long)base.seed, (unsigned long long)base.seed + K - 1);
	FILE *fcsv = nullptr, *fser = nullptr, *fchn = nullptr;
	if (ts.out) {
		fcsv = fopen((std::string(ts.out) + ".csv").c_str(), "w");
		fser = fopen((std::string(ts.out) + ".series.csv").c_str(), "w");
		if (!fcsv || !fser) { fprintf(stderr, "cannot open %s.csv / .series.csv for writing\n", ts.out); exit(EXIT_FAILURE); }
		fprintf(fcsv, "temp_bits,temp,nmeas,first_iter,last_iter,sum_M,sum_absM,sum_M2,sum_M4,sum_E,sum_E2,m_abs,m2,chi,U4,e,Cv%s\n", K > 1 ? ",chain,seed" : "");
		fprintf(fser, "temp_bits,iter,up,down,bond_equal%s\n", K > 1 ? ",chain" : "");
		if (K > 1) {
			fchn = fopen((std::string(ts.out) + ".chains.csv").c_str(), "w");
			if (!fchn) { fprintf(stderr, "cannot open %s.chains.csv for writing\n", ts.out); exit(EXIT_FAILURE); }
			fprintf(fchn, "temp_bits,temp,chains,m_abs,m_abs_err,m2,m2_err,chi,chi_err,U4,U4_err,e,e_err,Cv,Cv_err\n");
		}
	}
	// replica r > 0: a context of its own with the same configuration
	std::vector<Ring> reps(nrep);
	reps[0] = ring;
	for (int r = 1; r < nrep; r++) {
		ising_ctx *c = nullptr;
		ising_config cfg = base;
		cfg.seed = base.seed + (uint64_t)(r % K); // lattice r always carries chain r % K
		CHECK(ising_create(&cfg, &c));
		reps[r].ctx.push_back(c);
	}
	ising_batch *batch = nullptr;
	int batch_n = 0;
	auto make_batch = [&](int nb) -> bool { // a batch over the first nb lattices (the last group of points may be smaller)
		if (batch && batch_n == nb) return true;
		if (batch) { ising_batch_destroy(batch); batch = nullptr; }
		std::vector<ising_ctx *> mem;
		for (int j = 0; j < nb; j++) mem.push_back(reps[j].ctx[0]);
		if (ising_batch_create(mem.data(), nb, &batch) != ISING_OK) { batch = nullptr; return false; }
		batch_n = nb;
		return true;
	};
	if (batched && !make_batch(std::min(nrep, njobs))) {
		fprintf(stderr, "temperature sweep: no batched launches (%s)\n", ising_last_error());
		batched = false;
		// (two at a time on private streams: what round 2 did; more lattices than that only wait for each other)
		const int keep = (ts.replicas > 0 || K > 1) ? nrep : std::min(nrep, nspins < (1ull << 29) ? 2 : 1);
		for (int r = keep; r < nrep; r++) ising_destroy(reps[r].ctx[0]);
		nrep = keep;
		reps.resize(nrep);
	}
	bool quad_batch = false; // small lattices: one quad_pass_k launch per pass for all of them, the measurements inside the passes (ising_batch_sweep_counted)
	if (batched) {
		int h = 0, w = 0, qc = 0, qt = 0, qw = 0;
		CHECK(ising_batch_info(batch, &h, &w, nullptr));
		CHECK(ising_batch_quad_info(batch, &qc, &qt, &qw));
		quad_batch = qc > 0;
		if (quad_batch) fprintf(stderr, "temperature sweep: %d %s per batched launch (tiles of %d rows, passes of %d sweeps, %d waves per workgroup)\n", nrep, K > 1 ? "lattices" : "points", 4 * qc, qt, qw);
		else fprintf(stderr, "temperature sweep: %d %s per batched launch (strips of %d rows, %d workgroups per CU)\n", nrep, K > 1 ? "lattices" : "points", h, w);
	} else if (nrep > 1) {
		fprintf(stderr, "temperature sweep: %d points side by side, one stream each\n", nrep);
		for (Ring &rp : reps) CHECK(ising_use_private_stream(rp.ctx[0]));
	}
	std::vector<bool> have_J(nrep, false);
	std::vector<std::array<long double, 6>> chain_vals; // --tsweep-chains: the chains' averages of the point being completed
	std::vector<uint32_t> all_up; // --tsweep-cold: one slab's rows at 1 bit per spin, every spin up
	struct SeriesRow { int it; unsigned long long up, dw; long long A; };
	const long double N = (long double)nspins;
	long long total_sweeps = 0;
	int it = 0;
	const auto t0 = std::chrono::steady_clock::now();
	for (int k0 = 0; k0 < njobs; k0 += nrep) {
		const int nb = std::min(nrep, njobs - k0);
		std::vector<float> temps(nb);
		for (int j = 0; j < nb; j++) {
			temps[j] = (float)(ts.t0 + ((k0 + j) / K) * ts.dt);
			Ring &rp = reps[j];
			for (ising_ctx *c : rp.ctx) CHECK(ising_set_temperature(c, temps[j]));
			if (!ts.anneal || k0 == 0) {
				for (ising_ctx *c : rp.ctx) CHECK(ising_init_lattice(c));
				if (ts.cold) { // the ordered start: below T_c a random start coarsens for ages, this one equilibrates in a few correlation times
					if (all_up.empty()) all_up.assign((size_t)base.Y * (size_t)(base.X / 64), 0xFFFFFFFFu);
					for (ising_ctx *c : rp.ctx)
						for (int color = 0; color < 2; color++) CHECK(ising_write_bits(c, color, 0, base.Y, all_up.data()));
				}
				CHECK(ising_ring_exchange(rp.ctx.data(), rp.n(), ISING_BLACK));
				CHECK(ising_ring_exchange(rp.ctx.data(), rp.n(), ISING_WHITE));
				if (useJ && !have_J[j]) {
					CHECK(ising_ring_init_couplings(rp.ctx.data(), rp.n()));
					if (ts.J_symmetric) for (ising_ctx *c : rp.ctx) CHECK(ising_swap_couplings(c));
					have_J[j] = true;
				}
				it = 0;
			}
		}
		if (batched && !make_batch(nb)) { fprintf(stderr, "cannot batch %d lattices: %s\n", nb, ising_last_error()); exit(EXIT_FAILURE); }
		// equilibration, 32 sweeps (one fused launch) per replica at a time so that the replicas' launches alternate
		if (batched) CHECK(ising_batch_sweep(batch, it + 1, ts.nequil));
		for (int done = batched ? ts.nequil : 0; done < ts.nequil;) {
			const int n = nb > 1 ? std::min(32, ts.nequil - done) : ts.nequil - done;
			for (int j = 0; j < nb; j++) CHECK(ising_ring_sweep(reps[j].ctx.data(), reps[j].n(), it + done + 1, n));
			done += n;
		}
		it += ts.nequil;
		std::vector<Moments> mo(nb);
		std::vector<std::vector<SeriesRow>> rows(nb);
		const int first = it + ts.stride;
		// One device: measurements are enqueued behind their sweeps (ising_measure_enqueue) and read back in one go, so the
		// whole series of a point runs without a host round trip.  A ring of slabs reads its counters at every point.
		const bool async = ndev == 1;
		const int it_meas0 = it;
		auto take = [&](int j, int iter, unsigned long long up, long long A) {
			const unsigned long long dw = (unsigned long long)nspins - up;
			mo[j].add((long long)up - (long long)dw, 2 * (long long)nspins - 2 * A);
			if (fser) rows[j].push_back({iter, up, dw, A});
		};
		std::vector<int> fetched(nb, 0);
		auto fetch = [&](int j) {
			std::vector<uint64_t> ups(4096);
			std::vector<int64_t> As(4096);
			int n = 0;
			CHECK(ising_measure_fetch(reps[j].ctx[0], ups.data(), As.data(), 4096, &n));
			for (int i = 0; i < n; i++, fetched[j]++) take(j, it_meas0 + (fetched[j] + 1) * ts.stride, ups[i], As[i]);
		};
		// batched: one launch per `stride` sweeps of ALL points, one more for their measurements; read back every 1024
		auto fetch_batch = [&]() {
			std::vector<uint64_t> ups((size_t)1024 * nb);
			std::vector<int64_t> As((size_t)1024 * nb);
			int n = 0;
			CHECK(ising_batch_measure_fetch(batch, ups.data(), As.data(), 1024, &n));
			for (int i = 0; i < n; i++)
				for (int j = 0; j < nb; j++) take(j, it_meas0 + (fetched[j] + i + 1) * ts.stride, ups[(size_t)i * nb + j], As[(size_t)i * nb + j]);
			for (int j = 0; j < nb; j++) fetched[j] += n;
		};
		// a batch of small lattices whose measurements fall on multiples of the stride: the series rides inside the passes, 1024 measurements per read-back
		const bool inpass = batched && quad_batch && (it % ts.stride) == 0;
		for (int m = 0; inpass && m < ts.nmeas;) {
			const int nm = std::min(1024, ts.nmeas - m);
			std::vector<uint64_t> ups((size_t)nm * nb);
			std::vector<int64_t> As((size_t)nm * nb);
			int k = 0;
			CHECK(ising_batch_sweep_counted(batch, it + 1, nm * ts.stride, ts.stride, ups.data(), As.data(), nm, &k));
			if (k != nm) { fprintf(stderr, "temperature sweep: %d measurements came back, %d expected\n", k, nm); exit(EXIT_FAILURE); }
			for (int i = 0; i < k; i++)
				for (int j = 0; j < nb; j++) take(j, it + (i + 1) * ts.stride, ups[(size_t)i * nb + j], As[(size_t)i * nb + j]);
			for (int j = 0; j < nb; j++) fetched[j] += k;
			it += nm * ts.stride;
			m += nm;
		}
		for (int m = 0; batched && !inpass && m < ts.nmeas; m++) {
			CHECK(ising_batch_sweep(batch, it + 1, ts.stride));
			CHECK(ising_batch_measure_enqueue(batch));
			it += ts.stride;
			if ((m + 1) % 1024 == 0) fetch_batch();
		}
		if (batched && !inpass) fetch_batch();
		// One lattice at a time (annealing runs, --tsweep-replicas 1, rings of slabs) whose measurements fall on multiples of the stride: the whole
		// series of a point rides inside the launches (ising_ring_sweep_counted with the bond sums), 64 measurements per read-back
		const bool inlaunch = !batched && nb == 1 && (it % ts.stride) == 0;
		for (int m = 0; inlaunch && m < ts.nmeas;) {
			const int nm = std::min(64, ts.nmeas - m);
			uint64_t ups[64];
			int64_t eqs[64];
			int k = 0;
			CHECK(ising_ring_sweep_counted(reps[0].ctx.data(), reps[0].n(), it + 1, nm * ts.stride, ts.stride, ups, eqs, 64, &k));
			if (k != nm) { fprintf(stderr, "temperature sweep: %d measurements came back, %d expected\n", k, nm); exit(EXIT_FAILURE); }
			for (int i = 0; i < k; i++) take(0, it + (i + 1) * ts.stride, ups[i], eqs[i]);
			fetched[0] += k;
			it += nm * ts.stride;
			m += nm;
		}
		for (int m = 0; !batched && !inlaunch && m < ts.nmeas; m++) {
			for (int j = 0; j < nb; j++) {
				CHECK(ising_ring_sweep(reps[j].ctx.data(), reps[j].n(), it + 1, ts.stride));
				if (async) CHECK(ising_measure_enqueue(reps[j].ctx[0]));
			}
			it += ts.stride;
			for (int j = 0; j < nb; j++) {
				if (async) {
					if ((m + 1) % 4096 == 0) fetch(j);
				} else {
					unsigned long long up = 0, dw = 0;
					reps[j].count(&up, &dw);
					take(j, it, up, reps[j].bond_equal());
				}
			}
		}
		if (async && !batched && !inlaunch)
			for (int j = 0; j < nb; j++) fetch(j);
		for (int j = 0; j < nb; j++) {
			const float temp = temps[j];
			uint32_t tbits;
			memcpy(&tbits, &temp, 4);
			if (fser)
				for (const SeriesRow &r : rows[j]) {
					fprintf(fser, "%u,%d,%llu,%llu,%lld", tbits, r.it, r.up, r.dw, r.A);
					if (K > 1) fprintf(fser, ",%d", j % K);
					fprintf(fser, "\n");
				}
			total_sweeps += ts.nequil + (long long)ts.nmeas * ts.stride;
			const Moments &q = mo[j];
			const long double n = q.n;
			const long double mabs = (long double)q.sAbsM / (n * N), m2 = (long double)q.sM2 / (n * N * N), m4 = q.sM4 / (n * N * N * N * N);
			const long double e1 = (long double)q.sE / (n * N), e2 = (long double)q.sE2 / (n * N * N);
			const long double chi = N * (m2 - mabs * mabs) / temp, u4 = 1.0L - m4 / (3.0L * m2 * m2), cv = N * (e2 - e1 * e1) / ((long double)temp * temp);
			if (K == 1)
				printf("T = %f: <|m|> = %9.6f, <m^2> = %E, chi = %E, U4 = %9.6f, <e> = %9.6f, Cv = %E (iters %d-%d)\n", temp, (double)mabs, (double)m2,
				       (double)chi, (double)u4, (double)e1, (double)cv, first, it);
			if (fcsv) {
				fprintf(fcsv, "%u,%.9g,%d,%d,%d,%s,%s,%s,%.21Lg,%s,%s,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g", tbits, (double)temp, q.n, first, it,
				        i128_str(q.sM).c_str(), i128_str(q.sAbsM).c_str(), i128_str(q.sM2).c_str(), q.sM4, i128_str(q.sE).c_str(),
				        i128_str(q.sE2).c_str(), (double)mabs, (double)m2, (double)chi, (double)u4, (double)e1, (double)cv);
				if (K > 1) fprintf(fcsv, ",%d,%llu", j % K, (unsigned long long)base.seed + (unsigned long long)(j % K));
				fprintf(fcsv, "\n");
			}
			if (K > 1) { // the point is complete with its last chain: mean and standard error of the chains' averages
				if (j % K == 0) chain_vals.clear();
				chain_vals.push_back({mabs, m2, chi, u4, e1, cv});
				if (j % K == K - 1) {
					long double mean[6], err[6];
					for (int q6 = 0; q6 < 6; q6++) {
						long double sum = 0, dev2 = 0;
						for (const auto &cv6 : chain_vals) sum += cv6[q6];
						mean[q6] = sum / K;
						for (const auto &cv6 : chain_vals) dev2 += (cv6[q6] - mean[q6]) * (cv6[q6] - mean[q6]);
						err[q6] = sqrtl(dev2 / (K - 1) / K);
					}
					printf("T = %f: <|m|> = %9.6f +- %8.6f, chi = %E +- %.1E, U4 = %9.6f +- %8.6f, <e> = %9.6f +- %8.6f, Cv = %E +- %.1E (%d chains, iters %d-%d)\n",
					       temp, (double)mean[0], (double)err[0], (double)mean[2], (double)err[2], (double)mean[3], (double)err[3], (double)mean[4], (double)err[4],
					       (double)mean[5], (double)err[5], K, first, it);
					if (fchn) {
						fprintf(fchn, "%u,%.9g,%d", tbits, (double)temp, K);
						for (int q6 = 0; q6 < 6; q6++) fprintf(fchn, ",%.17g,%.17g", (double)mean[q6], (double)err[q6]);
						fprintf(fchn, "\n");
					}
				}
			}
		}
	}
	for (Ring &rp : reps) CHECK(ising_ring_synchronize(rp.ctx.data(), rp.n()));
	const double et = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	if (fcsv) fclose(fcsv);
	if (fser) fclose(fser);
	if (fchn) fclose(fchn);
	if (batch) ising_batch_destroy(batch);
	for (int r = 1; r < nrep; r++) ising_destroy(reps[r].ctx[0]);
	printf("\nTemperature sweep: %lld update steps in %E ms, %.2lf flips/ns (initialisation and measurements included)\n\n", total_sweeps, et,
	       (double)nspins * (double)total_sweeps / (et * 1.0E+6));
	return 0;
}

} // namespace

int main(int argc, char **argv) {
	int X = 0, Y = 0, dump_out = 0, nsteps = NUMIT_DEF, ndev = 1;
	unsigned long long seed = ISING_SEED_DEF;
	float alpha = -1.0f, temp = -1.0f, ramp_step = 0;
	int ramp_every = 0, print_every = 0, print_exp = 0, print_energy = 0;
	std::vector<unsigned long long> exp_points; // -e: sweep indices of the print points still to come are exp_points[exp_next ...]
	size_t exp_next = 0;
	double target_magn = -1.0;
	int with_sublattices = 0, XSL = 0, YSL = 0, NSLX = 1, NSLY = 1;
	int corr_out = 0;
	char cname[256];
	int with_couplings = 0;
	float antiferro_prob = 0.0f;
	std::vector<int> devmap;
	int layout = ISING_LAYOUT_AUTO, transport = ISING_TRANSPORT_AUTO;
	TsweepSpec ts;
	bool doTsweep = false, seedGiven = false;
	const char *ckptOut = nullptr, *ckptIn = nullptr;

	static struct option long_options[] = {
	    {"x", required_argument, 0, 'x'},      {"y", required_argument, 0, 'y'},     {"nit", required_argument, 0, 'n'},
	    {"seed", required_argument, 0, 's'},   {"out", no_argument, 0, 'o'},         {"devs", required_argument, 0, 'd'},
	    {"alpha", required_argument, 0, 'a'},  {"temp", required_argument, 0, 't'},  {"print", required_argument, 0, 'p'},
	    {"update", required_argument, 0, 'u'}, {"magn", required_argument, 0, 'm'},  {"exppr", no_argument, 0, 'e'},
	    {"corr", no_argument, 0, 'c'},         {"J", required_argument, 0, 'J'},     {"xsl", required_argument, 0, 1},
	    {"ysl", required_argument, 0, 2},      {"help", required_argument, 0, 'h'},  {"energy", no_argument, 0, 3},
	    {"devmap", required_argument, 0, 4},   {"layout", required_argument, 0, 5},  {"tsweep", required_argument, 0, 6},
	    {"tsweep-anneal", no_argument, 0, 7},  {"tsweep-out", required_argument, 0, 8}, {"checkpoint", required_argument, 0, 9},
	    {"resume", required_argument, 0, 10},  {"transport", required_argument, 0, 11}, {"tsweep-replicas", required_argument, 0, 12},
	    {"tsweep-no-batch", no_argument, 0, 13}, {"tsweep-cold", no_argument, 0, 14}, {"tsweep-chains", required_argument, 0, 15}, {"J-symmetric", no_argument, 0, 16},
	    {0, 0, 0, 0}};
	while (1) {
		int option_index = 0;
		const int och = getopt_long(argc, argv, "x:y:n:ohs:d:a:t:p:u:m:ecJ:r:", long_options, &option_index);
		if (och == -1) break;
		switch (och) {
		case 0: break;
		case 'x': X = atoi(optarg); break;
		case 'y': Y = atoi(optarg); break;
		case 'n': nsteps = atoi(optarg); break;
		case 'o': dump_out = 1; break;
		case 'h': usage(argv[0]); break;
		case 's':
			seed = atoll(optarg);
			seedGiven = true;
			if (seed == 0) seed = ((getpid() * rand()) & 0x7FFFFFFFF); // optimized/main.cu:1331-1333
			break;
		case 'd': ndev = atoi(optarg); break;
		case 'a': alpha = atof(optarg); break;
		case 't': temp = atof(optarg); break;
		case 'p': print_every = atoi(optarg); break;
		case 'e': print_exp = 1; break;
		case 'u': {
			char *t0 = strtok(optarg, ",");
			if (!t0) { fprintf(stderr, "cannot find temperature step in parameter...\n"); exit(EXIT_FAILURE); }
			char *t1 = strtok(NULL, ",");
			if (!t1) { fprintf(stderr, "cannot find iteration count in parameter...\n"); exit(EXIT_FAILURE); }
			ramp_step = atof(t0);
			ramp_every = atoi(t1);
			printf("tempUpdStep: %f, tempUpdFreq: %d\n", ramp_step, ramp_every);
		} break;
		case 'm': target_magn = atof(optarg); break;
		case 'c': corr_out = 1; break;
		case 'J':
			with_couplings = 1;
			antiferro_prob = atof(optarg);
			antiferro_prob = std::min(std::max(0.0f, antiferro_prob), 1.0f); // optimized/main.cu:1370
			break;
		case 1: with_sublattices = 1; XSL = atoi(optarg); break;
		case 2: with_sublattices = 1; YSL = atoi(optarg); break;
		case 3: print_energy = 1; break;
		case 4:
			for (char *tok = strtok(optarg, ","); tok; tok = strtok(NULL, ",")) devmap.push_back(atoi(tok));
			break;
		case 5:
			if (!strcmp(optarg, "ballot")) layout = ISING_LAYOUT_BALLOT;
			else if (!strcmp(optarg, "dense")) layout = ISING_LAYOUT_DENSE;
			else if (!strcmp(optarg, "nibble")) layout = ISING_LAYOUT_NIBBLE;
			else { fprintf(stderr, "error: --layout takes ballot, dense or nibble\n"); exit(EXIT_FAILURE); }
			break;
		case 6: {
			double v[6] = {0, 0, 0, 16, 16, 1};
			int nv = 0;
			for (char *tok = strtok(optarg, ","); tok && nv < 6; tok = strtok(NULL, ",")) v[nv++] = atof(tok);
			if (nv < 3 || v[2] <= 0 || v[1] < v[0] || v[3] < 0 || v[4] < 1 || v[5] < 1) {
				fprintf(stderr, "error: --tsweep takes T0,T1,DT[,NEQUIL[,NMEAS[,STRIDE]]] with T1 >= T0, DT > 0\n");
				exit(EXIT_FAILURE);
			}
			ts.t0 = v[0]; ts.t1 = v[1]; ts.dt = v[2]; ts.nequil = (int)v[3]; ts.nmeas = (int)v[4]; ts.stride = (int)v[5];
			doTsweep = true;
		} break;
		case 7: ts.anneal = true; break;
		case 8: ts.out = optarg; break;
		case 9: ckptOut = optarg; break;
		case 10: ckptIn = optarg; break;
		case 11:
			if (!strcmp(optarg, "copy")) transport = ISING_TRANSPORT_COPY;
			else if (!strcmp(optarg, "rccl")) transport = ISING_TRANSPORT_RCCL;
			else if (!strcmp(optarg, "auto")) transport = ISING_TRANSPORT_AUTO;
			else { fprintf(stderr, "error: --transport takes copy, rccl or auto\n"); exit(EXIT_FAILURE); }
			break;
		case 12:
			ts.replicas = atoi(optarg);
			if (ts.replicas < 0 || ts.replicas > 64) { fprintf(stderr, "error: --tsweep-replicas takes 0 (by lattice size) .. 64\n"); exit(EXIT_FAILURE); }
			break;
		case 13: ts.no_batch = true; break;
		case 14: ts.cold = true; break;
		case 16: ts.J_symmetric = true; break;
		case 15:
			ts.chains = atoi(optarg);
			if (ts.chains < 1 || ts.chains > 64) { fprintf(stderr, "error: --tsweep-chains takes 1 .. 64\n"); exit(EXIT_FAILURE); }
			break;
		case '?': exit(EXIT_FAILURE);
		default: fprintf(stderr, "unknown option: %c\n", och); exit(EXIT_FAILURE);
		}
	}

	ising_checkpoint_info ck;
	memset(&ck, 0, sizeof(ck));
	if (ckptIn) { // geometry, seed and temperature default to the checkpoint's
		if (doTsweep) { fprintf(stderr, "error: --resume and --tsweep exclude each other\n"); exit(EXIT_FAILURE); }
		CHECK(ising_checkpoint_info_read(ckptIn, &ck));
		if (ndev < 1 || (ck.Y_total % ndev)) { fprintf(stderr, "error: %d checkpoint rows do not split over %d devices\n", ck.Y_total, ndev); exit(EXIT_FAILURE); }
		if (!X) X = ck.X;
		if (!Y) Y = ck.Y_total / ndev;
		if (!seedGiven) seed = ck.seed;
		if (temp == -1.0f && alpha == -1.0f) temp = ck.temp;
		if (!with_sublattices && ck.XSL) { with_sublattices = 1; XSL = ck.XSL; YSL = ck.YSL; }
		if (!with_couplings && ck.use_J) { with_couplings = 1; antiferro_prob = ck.J_prob; }
	}
	// defaults and divisibility rules, optimized/main.cu:1395-1421
	if (!X || !Y) {
		if (!X) X = (Y && !(Y % X_MULT)) ? Y : X_MULT;
		if (!Y) Y = !(X % Y_MULT) ? X : Y_MULT;
	}
	if (!X || (X % 2) || ((X / 2) % (SPIN_X_WORD * 2 * 16 * 2))) {
		fprintf(stderr, "\nPlease specify an X dim multiple of %d\n\n", X_MULT);
		usage(argv[0]);
	}
	if (!Y || (Y % Y_MULT)) {
		fprintf(stderr, "\nPlease specify a Y dim multiple of %d\n\n", Y_MULT);
		usage(argv[0]);
	}
	if (with_sublattices) { // optimized/main.cu:1423-1457
		if (!XSL || !YSL) {
			if (!XSL) XSL = (YSL && !(YSL % X_MULT)) ? YSL : X_MULT;
			if (!YSL) YSL = !(XSL % Y_MULT) ? XSL : Y_MULT;
		}
		if ((X % XSL) || !XSL || (XSL % 2) || ((XSL / 2) % (SPIN_X_WORD * 2 * 16 * 2))) {
			fprintf(stderr, "\nPlease specify an X sub-lattice dim multiple of %d and divisor of %d\n\n", X_MULT, X);
			usage(argv[0]);
		}
		if ((Y % YSL) || !YSL || (YSL % Y_MULT)) {
			fprintf(stderr, "\nPlease specify a Y sub-lattice dim multiple of %d divisor of %d\n\n", Y_MULT, Y);
			usage(argv[0]);
		}
		NSLX = X / XSL;
		NSLY = Y / YSL;
	}
	if (temp == -1.0f) temp = (alpha == -1.0f) ? ALPHA_DEF * ISING_CRIT_TEMP : alpha * ISING_CRIT_TEMP; // :1465-1471
	if (print_exp && print_every) print_every = 0;
	const int j0 = ckptIn ? (int)ck.it : 0, jend = j0 + nsteps; // --resume: iterations continue where the checkpoint stopped
	if (print_exp) {
		exp_points = exp_print_points((unsigned long long)jend);
		while (exp_next < exp_points.size() && (long long)exp_points[exp_next] + 1 <= j0) exp_next++; // (--resume: points already behind us)
	}
	if (ndev < 1) { fprintf(stderr, "error: need at least one device\n"); exit(EXIT_FAILURE); }

	int visible = 0;
	CHECK(ising_device_count(&visible));
	if (devmap.empty()) for (int i = 0; i < ndev; i++) devmap.push_back(i);
	if ((int)devmap.size() != ndev) { fprintf(stderr, "error: --devmap needs %d entries\n", ndev); exit(EXIT_FAILURE); }
	for (int d : devmap) if (d < 0 || d >= visible) { fprintf(stderr, "error: device %d not available (%d visible)\n", d, visible); exit(EXIT_FAILURE); }

	printf("\nUsing GPUs:\n");
	for (int i = 0; i < ndev; i++) {
		char name[256];
		int cus = 0, thr = 0, major = 0, minor = 0;
		CHECK(ising_device_info(devmap[i], name, sizeof(name), &cus, &thr, &major, &minor));
		printf("\t%2d (%s, %d SMs, %d th/SM max, CC %d.%d, ECC %s)\n", i, name, cus, thr, major, minor, "on");
	}
	printf("\n");

	// optimized/main.cu:1508-1537: who reaches whose memory directly.  Slabs that --devmap puts on one device reach each other by definition; a missing link is the
	// reference's error and exit -- its remote loads need every pair -- unless the rows travel by RCCL, which routes around it.
	if (ndev > 1) {
		printf("GPUs direct access matrix:\n       ");
		for (int i = 0; i < ndev; i++) printf("%4d", i);
		int missing_links = 0;
		printf("\n");
		for (int i = 0; i < ndev; i++) {
			printf("GPU %2d:", i);
			for (int k = 0; k < ndev; k++) {
				int access = 1;
				if (devmap[i] != devmap[k]) {
					CHECK(ising_device_peer_access(devmap[i], devmap[k], &access));
					if (!access) missing_links++;
				}
				printf("%4c", access ? 'V' : 'X');
			}
			printf("\n");
		}
		printf("\n");
		if (missing_links && transport != ISING_TRANSPORT_RCCL) {
			fprintf(stderr, "error: %d direct memory links among devices missing\n", missing_links);
			exit(EXIT_FAILURE);
		}
	}

	const size_t row_words = (X / 2) / SPIN_X_WORD;
	const size_t slab_words = (size_t)Y * row_words;
	const size_t total_words = 2ull * ndev * slab_words;
	const int gridX = (int)((row_words / 2 + 31) / 32), gridY = (Y + 15) / 16; // the reference's launch grid = RNG stream geometry

	printf("Run configuration:\n");
	printf("\tspin/word: %d\n", SPIN_X_WORD);
	printf("\tspins: %zu\n", total_words * SPIN_X_WORD);
	printf("\tseed: %llu\n", seed);
	printf("\titerations: %d\n", nsteps);
	printf("\tblock (X, Y): %d, %d\n", 16, 16);
	printf("\ttile  (X, Y): %d, %d\n", 32, 16);
	printf("\tgrid  (X, Y): %d, %d\n", gridX, gridY);
	if (print_every) printf("\tprint magn. every %d steps\n", print_every);
	else if (print_exp) printf("\tprint magn. following exponential series\n");
	else printf("\tprint magn. at 1st and last step\n");
	if ((print_every || print_exp) && target_magn != -1.0) printf("\tearly exit if magn. == %lf+-%lf\n", target_magn, TGT_MAGN_MAX_DIFF);
	printf("\ttemp: %f (%f*T_crit)\n", temp, temp / ISING_CRIT_TEMP);
	if (!ramp_every) printf("\ttemp update not set\n");
	else printf("\ttemp update: %f / %d iterations\n", ramp_step, ramp_every);
	if (with_couplings) { // exactly one of the two lines, optimized/main.cu:1577-1581
		printf("\tusing Hamiltonian buffer, setting links to -1 with prob %G\n", antiferro_prob);
		if (ts.J_symmetric) printf("\tsymmetric bonds: each colour's update reads its own sites' links (not the reference's pairing)\n");
	} else {
		printf("\tnot using Hamiltonian buffer\n");
	}
	printf("\n");
	if (with_sublattices) { // optimized/main.cu:1583-1588
		printf("\tusing sub-lattices:\n");
		printf("\t\tno. of sub-lattices per GPU: %8d\n", NSLX * NSLY);
		printf("\t\tno. of sub-lattices (total): %8d\n", ndev * NSLX * NSLY);
		printf("\t\tsub-lattices size:           %7d x %7d\n\n", XSL, YSL);
	}
	printf("\tlocal lattice size:      %8d x %8d\n", Y, X);
	printf("\ttotal lattice size:      %8d x %8d\n", ndev * Y, X);
	printf("\tlocal lattice shape: 2 x %8d x %8zu (%12zu %s)\n", Y, row_words, slab_words * 2, "ulls");
	printf("\ttotal lattice shape: 2 x %8d x %8zu (%12zu %s)\n", ndev * Y, row_words, total_words, "ulls");
	printf("\tmemory: %.2lf MB (%.2lf MB per GPU)\n", (total_words * 8) / (1024.0 * 1024.0), slab_words * 2 * 8 / (1024.0 * 1024.0));

	Ring ring;
	ising_config cfg0; // (slab 0's configuration: --tsweep creates its replicas from it)
	memset(&cfg0, 0, sizeof(cfg0));
	if (ndev > 1) { printf("\nSetting up multi-gpu configuration:\n"); fflush(stdout); }
	for (int i = 0; i < ndev; i++) {
		ising_config cfg;
		memset(&cfg, 0, sizeof(cfg));
		cfg.X = X; cfg.Y = Y; cfg.nslabs = ndev; cfg.slab = i; cfg.seed = seed; cfg.temp = temp; cfg.device = devmap[i];
		cfg.strip_rows = 0; cfg.kernel = ISING_KERNEL_AUTO; cfg.layout = layout;
		// --tsweep with two lattices of ~2^26 spins side by side: two-row strips (3 workgroups per CU each: both fit the
		// chip) instead of the one-row strips a lone lattice of that size gets (4 per CU) -- 1.60 s against 1.65 s for config 5
		if (doTsweep && ndev == 1 && !ts.anneal && ts.replicas != 1 && (long long)X * Y >= (1LL << 26) && (long long)X * Y < 3 * (1LL << 25) && (Y % 2) == 0) cfg.strip_rows = 2;
		cfg.XSL = with_sublattices ? XSL : 0; cfg.YSL = with_sublattices ? YSL : 0;
		cfg.use_J = with_couplings; cfg.J_prob = antiferro_prob;
		ising_ctx *c = nullptr;
		CHECK(ising_create(&cfg, &c));
		ring.ctx.push_back(c);
		if (i == 0) cfg0 = cfg;
		if (ndev > 1) { printf("\tGPU %2d done\n", i); fflush(stdout); }
	}

	if (corr_out) { // optimized/main.cu:1660-1663
		if (with_sublattices && YSL < MAX_CORR_LEN) { fprintf(stderr, "-c needs sub-lattices of at least %d rows\n", MAX_CORR_LEN); exit(EXIT_FAILURE); }
		snprintf(cname, sizeof(cname), "corr_%dx%d_T_%f_%llu", Y, X, temp, seed);
		remove(cname);
	}
	if (transport != ISING_TRANSPORT_AUTO) CHECK(ising_ring_set_transport(ring.ctx.data(), ndev, transport));
	const size_t nspins = total_words * SPIN_X_WORD;
	if (doTsweep) {
		run_tsweep(ring, cfg0, ts, nspins, with_couplings);
		for (ising_ctx *c : ring.ctx) ising_destroy(c);
		return 0;
	}
	if (ckptIn) {
		int64_t it = 0;
		CHECK(ising_ring_checkpoint_load(ring.ctx.data(), ndev, ckptIn, &it));
		printf("\nResumed from %s: %lld iterations done\n", ckptIn, (long long)it);
	} else {
		for (ising_ctx *c : ring.ctx) CHECK(ising_init_lattice(c));
	}
	CHECK(ising_ring_exchange(ring.ctx.data(), ndev, ISING_BLACK));
	CHECK(ising_ring_exchange(ring.ctx.data(), ndev, ISING_WHITE));
	if (with_couplings) CHECK(ising_ring_init_couplings(ring.ctx.data(), ndev)); // optimized/main.cu:1729-1742
	// --J-symmetric: every colour's update reads its own sites' bonds (J_ij = J_ji) instead of the array the reference hands it
	if (with_couplings && ts.J_symmetric) for (ising_ctx *c : ring.ctx) CHECK(ising_swap_couplings(c));
	if (ndev > 1) {
		int tr = 0;
		CHECK(ising_ring_transport(ring.ctx.data(), ndev, &tr));
		fprintf(stderr, "halo rows travel by %s on a second stream per GPU\n", tr == ISING_TRANSPORT_RCCL ? "RCCL send/recv" : "peer-to-peer copies");
	}

	unsigned long long n_up = 0, n_down = 0;
	ring.count(&n_up, &n_down);
	printf("\nInitial magnetization: %9.6lf, up_s: %12llu, dw_s: %12llu\n",
	       fabs((double)n_up - (double)n_down) / (double)nspins, n_up, n_down);
	if (print_energy) printf("Initial energy/spin:   %9.6lf\n", ring.energy(nspins));
	CHECK(ising_ring_synchronize(ring.ctx.data(), ndev));

	auto report = [&](int iter, bool exp_style) -> bool {
		ring.count(&n_up, &n_down);
		const double magn = fabs((double)n_up - (double)n_down) / (double)nspins;
		if (exp_style) printf("        magnetization: %9.6lf (^2: %9.6lf), up_s: %12llu, dw_s: %12llu (iter: %8d)\n", magn, magn * magn, n_up, n_down, iter);
		else printf("        magnetization: %9.6lf, up_s: %12llu, dw_s: %12llu (iter: %8d)\n", magn, n_up, n_down, iter);
		if (print_energy) printf("        energy/spin:   %9.6lf (iter: %8d)\n", ring.energy(nspins), iter);
		if (corr_out) { // computeCorr, optimized/main.cu:1072-1138
			int64_t sums[MAX_CORR_LEN];
			CHECK(ising_ring_correlations(ring.ctx.data(), ndev, MAX_CORR_LEN, sums));
			FILE *fp = fopen(cname, "a");
			if (!fp) { fprintf(stderr, "cannot open %s\n", cname); exit(EXIT_FAILURE); }
			fprintf(fp, "%10d", iter);
			for (int i = 0; i < MAX_CORR_LEN; i++) fprintf(fp, " % -12G", (double)sums[i] / (2.0 * X * Y * ndev));
			fprintf(fp, "\n");
			fclose(fp);
		}
		if (dump_out) {
			char fname[256];
			snprintf(fname, sizeof(fname), "lattice_%dx%d_T_%f_IT_%08d_", Y, X, temp, iter);
			ring.dump(fname);
		}
		return target_magn != -1.0 && fabs(magn - target_magn) < TGT_MAGN_MAX_DIFF;
	};

	// hot loop, optimized/main.cu:1756-1871.  Sweeps between two host-side events (print, ramp) are enqueued as
	// one batch; the launches are asynchronous, so the GPU never waits for the host.
	const auto t0 = std::chrono::steady_clock::now();
	int j = j0;
	// Plain `-p N` runs on one GPU (the reference's usual command line: every number it publishes has the magnetisation every 16 sweeps inside
	// the timed loop, :1806-1810): the print points ride inside the library's launches (ising_sweep_counted) instead of cutting them into
	// pieces of N sweeps with a count and a read-back in between -- 16384^2 at -p 16: 3057 -> 3290 flips/ns.  The lines are the same; they
	// appear in bursts of up to 64.  --energy rides along (round 5: the launches' white levels count equal bonds, north_star's energy series).
	// Anything else a print point may do (-m early exit, -c, -o, the exponential series) keeps the reference's order of events below.
	// (several devices: ising_ring_sweep_counted -- every slab's deep launches count their own rows)
	const bool counted = print_every > 0 && !print_exp && target_magn == -1.0 && !corr_out && !dump_out;
	// (a burst is 64 print points on lattices of 2^26 spins and more, up to 4096 on smaller ones: a call costs a read-back and, on the small lattices' paths, a
	// conversion of the spins at either end -- 2048^2 at -p 16: 1473 flips/ns in bursts of 64, against 1788 without print points)
	const long long burst = 64 * std::max<long long>(1, std::min<long long>(64, (1LL << 26) / std::max<long long>(1, (long long)nspins)));
	std::vector<uint64_t> ups_v((size_t)burst + 16);
	std::vector<int64_t> eqs_v((size_t)burst + 16);
	while (counted && j < jend) {
		long long next = std::min<long long>(jend, (long long)(j / print_every + burst) * print_every);
		if (ramp_every) next = std::min<long long>(next, (long long)(j / ramp_every + 1) * ramp_every);
		uint64_t *ups = ups_v.data();
		int64_t *eqs = eqs_v.data();
		int k = 0;
		CHECK(ising_ring_sweep_counted(ring.ctx.data(), ndev, j + 1, (int)(next - j), print_every, ups, print_energy ? eqs : nullptr, (int)burst + 16, &k));
		for (int i = 0, it = (j / print_every + 1) * print_every; i < k; i++, it += print_every) {
			n_up = ups[i];
			n_down = nspins - ups[i];
			printf("        magnetization: %9.6lf, up_s: %12llu, dw_s: %12llu (iter: %8d)\n", fabs((double)n_up - (double)n_down) / (double)nspins, n_up, n_down, it);
			if (print_energy) printf("        energy/spin:   %9.6lf (iter: %8d)\n", -(2.0 * (double)eqs[i] - 2.0 * (double)nspins) / (double)nspins, it);
		}
		j = (int)next;
		if (ramp_every && (j % ramp_every) == 0) { // optimized/main.cu:1848-1860
			temp = std::max(MIN_TEMP, temp + ramp_step);
			printf("Changing temperature to %f\n", temp);
			for (int d = 0; d < ndev; d++) CHECK(ising_set_temperature(ring.ctx[d], temp));
			float tab[10];
			CHECK(ising_get_tables(ring.ctx[0], tab, nullptr));
			for (int i = 0; i < 2; i++)
				for (int k2 = 0; k2 < 5; k2++) printf("exp[%2d][%d]: %E\n", i ? 1 : -1, k2, tab[i * 5 + k2]);
		}
	}
	while (j < jend) {
		int next = jend; // first iteration index (1-based count) at which the host must look at the lattice
		if (print_every) next = std::min(next, (j / print_every + 1) * print_every);
		if (print_exp && exp_next < exp_points.size()) next = (int)std::min<long long>(next, (long long)exp_points[exp_next] + 1);
		if (ramp_every) next = std::min(next, (j / ramp_every + 1) * ramp_every);
		if (next <= j) next = j + 1;
		CHECK(ising_ring_sweep(ring.ctx.data(), ndev, j + 1, next - j));
		j = next;
		bool stop = false;
		if (print_every && (j % print_every) == 0) stop = report(j, false);
		if (!stop && print_exp && exp_next < exp_points.size() && exp_points[exp_next] == (unsigned long long)(j - 1)) {
			exp_next++;
			stop = report(j, true);
		}
		if (stop) break;
		if (ramp_every && (j % ramp_every) == 0) { // optimized/main.cu:1848-1860
			temp = std::max(MIN_TEMP, temp + ramp_step);
			printf("Changing temperature to %f\n", temp);
			for (ising_ctx *c : ring.ctx) CHECK(ising_set_temperature(c, temp));
			float tab[10];
			CHECK(ising_get_tables(ring.ctx[0], tab, nullptr));
			for (int i = 0; i < 2; i++)
				for (int k = 0; k < 5; k++) printf("exp[%2d][%d]: %E\n", i ? 1 : -1, k, tab[i * 5 + k]);
		}
	}
	CHECK(ising_ring_synchronize(ring.ctx.data(), ndev));
	const double et = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

	ring.count(&n_up, &n_down);
	printf("Final   magnetization: %9.6lf, up_s: %12llu, dw_s: %12llu (iter: %8d)\n\n",
	       fabs((double)n_up - (double)n_down) / (double)nspins, n_up, n_down, j);
	if (print_energy) printf("Final   energy/spin:   %9.6lf\n\n", ring.energy(nspins));

	// optimized/main.cu:1884-1890 (1.5 bytes per flip + the 20-byte table per reference block)
	const int jrun = j - j0;
	printf("Kernel execution time for %d update steps: %E ms, %.2lf flips/ns (BW: %.2lf GB/s)\n", jrun, et,
	       (double)nspins * jrun / (et * 1.0E+6),
	       (2ull * jrun * (8.0 * ((total_words / 2) + (total_words / 2) + (total_words / 2)) + 4.0 * 5 * gridX * gridY) / 1.0E+9) / (et / 1.0E+3));

	if (dump_out) {
		char fname[256];
		snprintf(fname, sizeof(fname), "lattice_%dx%d_T_%f_IT_%08d_", Y, X, temp, j);
		ring.dump(fname);
	}
	if (ckptOut) {
		CHECK(ising_ring_checkpoint_save(ring.ctx.data(), ndev, ckptOut, j));
		printf("Checkpoint written to %s (%d iterations done)\n", ckptOut, j);
	}
	for (ising_ctx *c : ring.ctx) ising_destroy(c);
	return 0;
}
