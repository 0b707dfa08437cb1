// Prototype of a "ballot pipeline": per row, 16 hoisted Philox blocks; each of the 64 draws is compared against two
// thresholds with plain v_cmp (-> SGPR pairs = 64-lane ballots), the ballots go to a per-wave scratch slot with scalar
// stores, s_dcache_wb is issued once per row, and one row later every lane reads two ballots back and runs ~60 VALU
// operations of word logic.  Compared against the shipped form (v_cmpx + masked v_or per site).  No lattice traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
__device__ __forceinline__ uint32_t x3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ void mul(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) { unsigned long long p = (unsigned long long)a * b; hi = p >> 32; lo = (uint32_t)p; }
struct Row { uint32_t t_lo1, t_hi0, t_lo0, t_e; };
__device__ __forceinline__ Row setup(uint32_t tid, uint32_t k0x, uint32_t k2y) { Row r; uint32_t h; mul(M1, tid, h, r.t_lo1); mul(M0, h ^ k0x, r.t_hi0, r.t_lo0); r.t_e = r.t_lo0 ^ k2y; return r; }
__device__ __forceinline__ void block(const Row& pr, uint32_t cx, uint32_t sl, uint32_t sh, uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
  uint32_t a, b; mul(M0, cx, a, b); uint32_t c1z = a ^ sh; uint32_t k1x = sl + W0, k1y = sh + W1; uint32_t c, d; mul(M1, c1z, c, d);
  uint32_t c0 = (c ^ k1x) ^ pr.t_lo1, c1 = d, c2 = pr.t_hi0 ^ (b ^ k1y), c3 = pr.t_lo0; uint32_t kx = sl + 2 * W0, ky = sh + 2 * W1;
  { uint32_t h0, l0, h1, l1; mul(M0, c0, h0, l0); mul(M1, c2, h1, l1); c0 = h1 ^ (c1 ^ kx); c1 = l1; c2 = h0 ^ pr.t_e; c3 = l0; }
#pragma unroll
  for (int r = 3; r < 10; ++r) { kx += W0; ky += W1; uint32_t h0, l0, h1, l1; mul(M0, c0, h0, l0); mul(M1, c2, h1, l1); c0 = x3(h1, c1, kx); c1 = l1; c2 = x3(h0, c3, ky); c3 = l0; }
  o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}
template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long* scratch, uint32_t* out, uint32_t sl, uint32_t sh, uint32_t n3, uint32_t n4, int rows) {
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  unsigned long long* slot_v = scratch + (size_t)(tid >> 6) * 256;   // 2 x 1 KiB per wave
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)slot_v), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)slot_v >> 32));
  unsigned long long* slot = (unsigned long long*)(((uintptr_t)hi << 32) | lo);
  uint32_t acc = 0;
  for (int r = 0; r < rows; ++r) {
    Row pr = setup(tid + r * 7919u, sl, sh + 2 * W1);
    if (MODE == 0) {
      uint32_t c3[2] = {0, 0}, c4[2] = {0, 0};
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        uint32_t o0, o1, o2, o3; block(pr, 16u * r + b, sl, sh, o0, o1, o2, o3);
        unsigned long long sv;
        asm volatile("s_mov_b64 %[sv], exec\n\tv_cmpx_gt_u32_e32 vcc, %[n3], %[o0]\n\tv_or_b32_e32 %[c3], 0x10000, %[c3]\n\tv_cmpx_gt_u32_e32 vcc, %[n4], %[o0]\n\tv_or_b32_e32 %[c4], 0x10000, %[c4]\n\ts_mov_b64 exec, %[sv]\n\t"
                     "v_cmpx_gt_u32_e32 vcc, %[n3], %[o1]\n\tv_or_b32_e32 %[c3], 0x20000, %[c3]\n\tv_cmpx_gt_u32_e32 vcc, %[n4], %[o1]\n\tv_or_b32_e32 %[c4], 0x20000, %[c4]\n\ts_mov_b64 exec, %[sv]\n\t"
                     "v_cmpx_gt_u32_e32 vcc, %[n3], %[o2]\n\tv_or_b32_e32 %[c3], 0x40000, %[c3]\n\tv_cmpx_gt_u32_e32 vcc, %[n4], %[o2]\n\tv_or_b32_e32 %[c4], 0x40000, %[c4]\n\ts_mov_b64 exec, %[sv]\n\t"
                     "v_cmpx_gt_u32_e32 vcc, %[n3], %[o3]\n\tv_or_b32_e32 %[c3], 0x80000, %[c3]\n\tv_cmpx_gt_u32_e32 vcc, %[n4], %[o3]\n\tv_or_b32_e32 %[c4], 0x80000, %[c4]\n\ts_mov_b64 exec, %[sv]"
                     : [c3] "+v"(c3[b >> 3]), [c4] "+v"(c4[b >> 3]), [sv] "=&s"(sv) : [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2), [o3] "v"(o3), [n3] "s"(n3), [n4] "s"(n4) : "vcc");
      }
      // ~50 VALU ops of word logic
      uint32_t t = c3[0] ^ c4[1];
#pragma unroll
      for (int i = 0; i < 24; ++i) { t = (t ^ (c3[i & 1] >> (i & 7))) + (c4[i & 1] & (t << 1)); }
      acc += t;
    } else {
      unsigned long long* cur = slot + (r & 1) * 128;
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        uint32_t o0, o1, o2, o3; block(pr, 16u * r + b, sl, sh, o0, o1, o2, o3);
        unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
        asm volatile("v_cmp_gt_u32_e64 %0, %8, %10\n\tv_cmp_gt_u32_e64 %1, %9, %10\n\tv_cmp_gt_u32_e64 %2, %8, %11\n\tv_cmp_gt_u32_e64 %3, %9, %11\n\t"
                     "v_cmp_gt_u32_e64 %4, %8, %12\n\tv_cmp_gt_u32_e64 %5, %9, %12\n\tv_cmp_gt_u32_e64 %6, %8, %13\n\tv_cmp_gt_u32_e64 %7, %9, %13"
                     : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(m7)
                     : "s"(n3), "s"(n4), "v"(o0), "v"(o1), "v"(o2), "v"(o3));
        const unsigned long long* p = cur + 8 * b;
        asm volatile("s_store_dwordx2 %0, %8, 0x0\n\ts_store_dwordx2 %1, %8, 0x8\n\ts_store_dwordx2 %2, %8, 0x10\n\ts_store_dwordx2 %3, %8, 0x18\n\t"
                     "s_store_dwordx2 %4, %8, 0x20\n\ts_store_dwordx2 %5, %8, 0x28\n\ts_store_dwordx2 %6, %8, 0x30\n\ts_store_dwordx2 %7, %8, 0x38"
                     :: "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(m4), "s"(m5), "s"(m6), "s"(m7), "s"(p) : "memory");
      }
      // previous row: its write-back was issued a whole row ago
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (r > 0) {
        const unsigned long long* prev = slot + ((r - 1) & 1) * 128;
        const unsigned long long a3 = __builtin_nontemporal_load(prev + 2 * lane), a4 = __builtin_nontemporal_load(prev + 2 * lane + 1);
        uint32_t c3[2] = {(uint32_t)a3, (uint32_t)(a3 >> 32)}, c4[2] = {(uint32_t)a4, (uint32_t)(a4 >> 32)};
        uint32_t t = c3[0] ^ c4[1];
#pragma unroll
        for (int i = 0; i < 24; ++i) { t = (t ^ (c3[i & 1] >> (i & 7))) + (c4[i & 1] & (t << 1)); }
        acc += t;
      }
      asm volatile("s_dcache_wb" ::: "memory");
    }
  }
  out[tid] = acc;
}
int main() {
  uint32_t* out; unsigned long long* scratch;
  const uint32_t n3 = 736899936u, n4 = 126432037u; const int rows = 32;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int blocks : {2048, 4096, 8192}) {
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&scratch, (size_t)blocks * 4 * 2048));
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, scratch, out, 0x1234567u, 0x89abcdu, n3, n4, rows);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, scratch, out, 0x1234567u, 0x89abcdu, n3, n4, rows);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double sites = (double)blocks * 256 * rows * 64;
      printf("blocks %5d mode %s: %8.3f ms  %8.1f sites/ns\n", blocks, mode ? "BALLOT" : "CMPX  ", best, sites / best * 1e-6);
    }
    CK(hipFree(out)); CK(hipFree(scratch));
  }
  return 0;
}
