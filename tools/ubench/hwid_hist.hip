// Workgroups of a persistent grid per CU: histogram over the chip, for a grid of `nwg` workgroups of `nt` threads whose
// dynamic LDS request caps the residency at `cap` per CU (mimics the update kernel's 6).
// build: hipcc --offload-arch=gfx950 -O2 -o hwid_hist hwid_hist.hip ; run: ./hwid_hist [threads] [workgroups] [cap]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void k(uint32_t *out, int spin) {
	extern __shared__ char lds[];
	uint32_t hw, xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; lds[0] = 1; }
	for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(127);
}
int main(int argc, char **argv) {
	const int nt = argc > 1 ? atoi(argv[1]) : 256, nwg = argc > 2 ? atoi(argv[2]) : 1280, cap = argc > 3 ? atoi(argv[3]) : 6;
	uint32_t *d;
	hipMalloc(&d, nwg * 8);
	const size_t lds = (160 * 1024 / cap) & ~1023u;
	hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipLaunchKernelGGL(k, dim3(nwg), dim3(nt), lds, 0, d, 3000);
	std::vector<uint32_t> h(2 * nwg);
	if (hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("failed\n"); return 1; }
	std::map<uint32_t, int> per_cu;
	for (int w = 0; w < nwg; ++w) {
		const uint32_t hw = h[2 * w], x = h[2 * w + 1] & 15;
		per_cu[(x << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)]++;
	}
	int hist[16] = {0};
	for (auto &kv : per_cu) hist[kv.second < 15 ? kv.second : 15]++;
	printf("%d workgroups of %d threads, cap %d per CU: %zu CUs used; CUs holding k workgroups:", nwg, nt, cap, per_cu.size());
	for (int i = 1; i < 16; ++i) if (hist[i]) printf(" k=%d: %d", i, hist[i]);
	printf("\n");
	return 0;
}
