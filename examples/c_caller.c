/* The C ABI from plain C (what a cgo / JNI / FFI host binds): BASELINE config 2's lattice, 16 sweeps, counts and energy.
 *   gcc -std=c99 -Iinclude examples/c_caller.c -o c_caller -Lising_gpu_amd -lising_hip -Wl,-rpath,$PWD/ising_gpu_amd -Wl,-rpath,/opt/rocm/lib && ./c_caller */
#include <stdio.h>
#include <string.h>
#include "ising_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != ISING_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ising_last_error()); return 1; } } while (0)

int main(void) {
	ising_config cfg;
	memset(&cfg, 0, sizeof cfg);
	cfg.X = 16384; cfg.Y = 16384;      /* optimized/main.cu -x / -y */
	cfg.nslabs = 1; cfg.slab = 0;      /* one device holds the lattice */
	cfg.seed = 1234;                   /* -s */
	cfg.temp = ISING_CRIT_TEMP;        /* -a 1 */
	cfg.layout = ISING_LAYOUT_AUTO;    /* the library picks the device layout; the ABI speaks the reference's packed format */
	ising_ctx *ctx = NULL;
	CHECK(ising_create(&cfg, &ctx));
	CHECK(ising_init_lattice(ctx));
	uint64_t up = 0, down = 0;
	CHECK(ising_count(ctx, &up, &down));
	printf("sweeps 0: up %llu down %llu\n", (unsigned long long)up, (unsigned long long)down);
	float ms = 0;
	CHECK(ising_sweep_timed(ctx, /*first_it=*/1, /*nsweeps=*/16, &ms));
	int64_t bonds = 0;
	CHECK(ising_count(ctx, &up, &down));
	CHECK(ising_bond_equal(ctx, &bonds));
	printf("sweeps 16: up %llu down %llu bond_equal %lld (%.3f ms, %.0f flips/ns)\n", (unsigned long long)up, (unsigned long long)down, (long long)bonds, ms,
	       (double)cfg.X * cfg.Y * 16 / (ms * 1e6));
	CHECK(ising_destroy(ctx));
	return 0;
}
