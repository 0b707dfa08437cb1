// Where do the FIRST n workgroups of a grid land?  (round 6: the quad path's tiles are the first workgroups of their launch -- do two of them share a CU while
// other CUs hold none?)  build: hipcc --offload-arch=gfx950 -O2 -o wg_place wg_place.hip ; run: ./wg_place threads lds_bytes workgroups first_n
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void k(uint32_t *out, int spin) {
	extern __shared__ uint32_t sh[];
	uint32_t hw, xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; sh[0] = hw; }
	for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(127); // keep every workgroup resident while the others start
}
int main(int argc, char **argv) {
	const int nt = argc > 1 ? atoi(argv[1]) : 768, lds = argc > 2 ? atoi(argv[2]) : 24576, nwg = argc > 3 ? atoi(argv[3]) : 512, first = argc > 4 ? atoi(argv[4]) : 64;
	uint32_t *d;
	hipMalloc(&d, nwg * 8);
	if (lds > 65536) hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	hipLaunchKernelGGL(k, dim3(nwg), dim3(nt), lds, 0, d, 500);
	std::vector<uint32_t> h(2 * nwg);
	hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
	for (int n : {first, 2 * first, 4 * first, nwg}) {
		if (n > nwg) continue;
		std::map<uint32_t, int> cus;
		int perx[16] = {0};
		for (int w = 0; w < n; ++w) {
			const uint32_t hw = h[2 * w], x = h[2 * w + 1] & 15;
			cus[(x << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)]++;
			perx[x]++;
		}
		int mx = 0, two = 0;
		for (auto &kv : cus) { mx = kv.second > mx ? kv.second : mx; two += kv.second > 1; }
		printf("%d threads, %d B LDS: the first %4d workgroups sit on %3zu CUs (at most %d on one, %d CUs hold more than one); per XCD:", nt, lds, n, cus.size(), mx, two);
		for (int x = 0; x < 8; ++x) printf(" %d", perx[x]);
		printf("\n");
	}
	return 0;
}
