// Where do the waves of a persistent grid land?  Prints HW_ID / XCC_ID fields of every wave of the first workgroups.
// build: hipcc --offload-arch=gfx950 -O2 -o hwid hwid.hip ; run: ./hwid [threads per workgroup] [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(uint32_t *out, int spin) {
	uint32_t hw, xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
	if ((threadIdx.x & 63) == 0) { out[2 * w] = hw; out[2 * w + 1] = xcc; }
	for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(127); // keep every workgroup resident while the others start
}
int main(int argc, char **argv) {
	const int nt = argc > 1 ? atoi(argv[1]) : 512, nwg = argc > 2 ? atoi(argv[2]) : 768;
	const int nw = nwg * nt / 64;
	uint32_t *d;
	hipMalloc(&d, nw * 8);
	hipLaunchKernelGGL(k, dim3(nwg), dim3(nt), 0, 0, d, 2000);
	std::vector<uint32_t> h(2 * nw);
	hipMemcpy(h.data(), d, nw * 8, hipMemcpyDeviceToHost);
	for (int w = 0; w < nw; ++w) {
		const uint32_t hw = h[2 * w], x = h[2 * w + 1];
		if (w / (nt / 64) < 4 || (w / (nt / 64)) % 97 == 0)
			printf("wg %4d wave %2d: hw_id %08x wave_id %2u simd %u pipe %u cu %2u sh %u se %u  xcc %u\n", w / (nt / 64), w % (nt / 64), hw, hw & 15, (hw >> 4) & 3,
			       (hw >> 6) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, x & 15);
	}
	// histogram of wave_id over all waves
	int hist[16] = {0};
	for (int w = 0; w < nw; ++w) hist[h[2 * w] & 15]++;
	printf("wave_id histogram:");
	for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
	printf("\n");
	return 0;
}
