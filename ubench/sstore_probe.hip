// Probe: can wave-uniform ballots (SGPR pairs from v_cmp) be handed to per-lane registers through memory
// (s_store_dwordx4 -> s_dcache_wb -> vector load) fast enough to replace exec-masked per-lane accumulation?
// Each wave writes 128 ballots (1 KiB) per "row" to its own scratch slot, then every lane loads 16 B of them back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long* scratch, unsigned* out, unsigned* err, int rows, unsigned salt) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  unsigned long long* slot_v = scratch + (size_t)wave * 128;       // 1 KiB per wave (ring of 1 row)
  // wave-uniform pointer for the scalar stores
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)slot_v), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)slot_v >> 32));
  unsigned long long* slot = (unsigned long long*)(((uintptr_t)hi << 32) | lo);
  const unsigned thr = __builtin_amdgcn_readfirstlane(0x30000000u + (salt & 1));
  unsigned acc = 0, bad = 0;
  for (int r = 0; r < rows; ++r) {
    // 128 ballots: ballot b = lanes where hash(lane, r, b) is below a threshold
#pragma unroll
    for (int b = 0; b < 128; b += 2) {
      const unsigned h0 = (lane * 2654435761u) ^ ((r * 128 + b) * 40503u + salt), h1 = (lane * 2654435761u) ^ ((r * 128 + b + 1) * 40503u + salt);
      unsigned long long m0, m1;
      asm volatile("v_cmp_gt_u32_e64 %0, %4, %2\n\tv_cmp_gt_u32_e64 %1, %4, %3" : "=s"(m0), "=s"(m1) : "v"(h0 * 2246822519u), "v"(h1 * 2246822519u), "s"(thr));
      // one 16-byte scalar store for the pair
      const unsigned long long* p = slot + b;
      if (MODE >= 1) asm volatile("s_store_dwordx2 %0, %2, 0x0\n\ts_store_dwordx2 %1, %2, 0x8" :: "s"(m0), "s"(m1), "s"(p) : "memory");
      else asm volatile("" :: "s"(m0), "s"(m1));
    }
    if (MODE >= 2) asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    // every lane reads back ballots 2*lane and 2*lane+1 (bypassing the vector L1)
    unsigned long long v0 = 0, v1 = 0;
    if (MODE >= 3) { v0 = __builtin_nontemporal_load(slot + 2 * lane); v1 = __builtin_nontemporal_load(slot + 2 * lane + 1); }
    // verify: recompute bit `lane`... (cheap check: parity of popcounts against a recomputation by shuffles is too costly; check 1 bit)
    const unsigned hb = (lane * 2654435761u) ^ ((r * 128 + 2 * lane) * 40503u + salt);
    const bool mine = (hb * 2246822519u) < thr;   // bit `lane` of ballot 2*lane
    if (MODE >= 3 && (((v0 >> lane) & 1ull) != 0ull) != mine) bad++;
    acc += __popcll(v0) + __popcll(v1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // reads done before the slot is rewritten
  }
  out[wave * 64 + lane] = acc;
  if (bad) atomicAdd(err, bad);
}

int main() {
  const int blocks = 256 * 8, rows = 64;
  unsigned long long* scratch; unsigned *out, *err;
  CK(hipMalloc(&scratch, (size_t)blocks * 4 * 1024)); CK(hipMalloc(&out, blocks * 256 * 4)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 4; ++mode) {
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, scratch, out, err, rows, 1234u + rep);
    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, scratch, out, err, rows, 1234u + rep);
    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, scratch, out, err, rows, 1234u + rep);
    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, scratch, out, err, rows, 1234u + rep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("mode %d (0 cmp only, 1 +s_store, 2 +dcache_wb, 3 +readback): %.3f ms, cycles per ballot per SIMD %.2f\n", mode, best, best * 1e-3 * 2.4e9 / ((double)blocks * 4 * rows * 128 / 1024));
  }
  float best = 0;
  unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  const double ballots = (double)blocks * 4 * rows * 128;
  printf("scalar-store transpose: %.3f ms, %.1f Gballots/s (= %.1f Gsites/s if one ballot = 64 sites / 2 thresholds), errors %u\n",
         best, ballots / best * 1e-6, ballots * 32 / best * 1e-6, herr);
  // cost per ballot in SIMD cycles at 2.4 GHz: waves per SIMD = blocks*4/1024
  printf("cycles per ballot per SIMD: %.2f\n", best * 1e-3 * 2.4e9 / (ballots / 1024));
  return 0;
}
