// Probe: accept-rank accumulation by LDS lookup (top 16 bits of the draw -> rank byte) vs v_cmpx + masked add.
// Both variants run the same hoisted Philox4x32-10 (16 blocks per "row"), no global memory traffic.
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
__global__ void __launch_bounds__(1024) k(uint32_t* out, uint32_t sl, uint32_t sh, uint32_t n3, uint32_t n4, int rows) {
  __shared__ uint8_t lut[MODE == 1 ? 65536 : 16];
  __shared__ uint32_t accw[MODE == 2 ? 8 * 1024 : 1];
  if (MODE == 1) {
    const uint32_t h3 = n3 >> 16, h4 = n4 >> 16;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) {
      uint32_t w = 0;
      for (int b = 0; b < 4; ++b) { uint32_t h = 4 * i + b; uint32_t r = (h < h3) + (h < h4); if (h == h3 || h == h4) r = 3; w |= r << (8 * b); }
      reinterpret_cast<uint32_t*>(lut)[i] = w;
    }
    __syncthreads();
  }
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int r = 0; r < rows; ++r) {
    Row pr = setup(tid + r * 7919u, sl, sh + 2 * W1);
    uint32_t R[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      uint32_t o0, o1, o2, o3; block(pr, 16u * r + b, sl, sh, o0, o1, o2, o3);
      uint32_t& rx = R[(b >> 2) * 2], & ry = R[(b >> 2) * 2 + 1];
      if (MODE == 2) {
        // accumulators live in LDS: word index (d*1024 + tid) -> conflict-free per instruction
        const unsigned ax = (unsigned)(((b >> 2) * 2) * 1024 + threadIdx.x) * 4u, ay = ax + 4096u;
        unsigned long long sv;
        unsigned k16 = 16u, k1 = 1u;
        asm volatile("s_mov_b64 %[sv], exec\n\tv_cmpx_gt_u32_e32 vcc, %[n3], %[o0]\n\tds_add_u32 %[ax], %[k16]\n\tv_cmpx_gt_u32_e32 vcc, %[n4], %[o0]\n\tds_add_u32 %[ax], %[k16]\n\ts_mov_b64 exec, %[sv]\n\t"
                     "v_cmpx_gt_u32_e32 vcc, %[n3], %[o1]\n\tds_add_u32 %[ay], %[k16]\n\tv_cmpx_gt_u32_e32 vcc, %[n4], %[o1]\n\tds_add_u32 %[ay], %[k16]\n\ts_mov_b64 exec, %[sv]\n\t"
                     "v_cmpx_gt_u32_e32 vcc, %[n3], %[o2]\n\tds_add_u32 %[ax], %[k1]\n\tv_cmpx_gt_u32_e32 vcc, %[n4], %[o2]\n\tds_add_u32 %[ax], %[k1]\n\ts_mov_b64 exec, %[sv]\n\t"
                     "v_cmpx_gt_u32_e32 vcc, %[n3], %[o3]\n\tds_add_u32 %[ay], %[k1]\n\tv_cmpx_gt_u32_e32 vcc, %[n4], %[o3]\n\tds_add_u32 %[ay], %[k1]\n\ts_mov_b64 exec, %[sv]"
                     : [sv] "=&s"(sv) : [ax] "v"(ax), [ay] "v"(ay), [k16] "v"(k16), [k1] "v"(k1), [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2), [o3] "v"(o3), [n3] "s"(n3), [n4] "s"(n4) : "vcc", "memory");
      } else if (MODE == 1) {
        uint32_t a0 = lut[o0 >> 16], a1 = lut[o1 >> 16], a2 = lut[o2 >> 16], a3 = lut[o3 >> 16];
        rx = (rx << 4) | a2; rx = (rx << 4) | a0; ry = (ry << 4) | a3; ry = (ry << 4) | a1;
      } else {
        unsigned long long sv;
        asm volatile("s_mov_b64 %[sv], exec\n\tv_cmpx_gt_u32_e32 vcc, %[n3], %[o0]\n\tv_add_u32_e32 %[rx], 16, %[rx]\n\tv_cmpx_gt_u32_e32 vcc, %[n4], %[o0]\n\tv_add_u32_e32 %[rx], 16, %[rx]\n\ts_mov_b64 exec, %[sv]\n\t"
                     "v_cmpx_gt_u32_e32 vcc, %[n3], %[o1]\n\tv_add_u32_e32 %[ry], 16, %[ry]\n\tv_cmpx_gt_u32_e32 vcc, %[n4], %[o1]\n\tv_add_u32_e32 %[ry], 16, %[ry]\n\ts_mov_b64 exec, %[sv]\n\t"
                     "v_cmpx_gt_u32_e32 vcc, %[n3], %[o2]\n\tv_add_u32_e32 %[rx], 1, %[rx]\n\tv_cmpx_gt_u32_e32 vcc, %[n4], %[o2]\n\tv_add_u32_e32 %[rx], 1, %[rx]\n\ts_mov_b64 exec, %[sv]\n\t"
                     "v_cmpx_gt_u32_e32 vcc, %[n3], %[o3]\n\tv_add_u32_e32 %[ry], 1, %[ry]\n\tv_cmpx_gt_u32_e32 vcc, %[n4], %[o3]\n\tv_add_u32_e32 %[ry], 1, %[ry]\n\ts_mov_b64 exec, %[sv]"
                     : [rx] "+v"(rx), [ry] "+v"(ry), [sv] "=&s"(sv) : [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2), [o3] "v"(o3), [n3] "s"(n3), [n4] "s"(n4) : "vcc");
      }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) acc += R[d];
    if (MODE == 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < 8; ++d) { acc += accw[d * 1024 + threadIdx.x]; accw[d * 1024 + threadIdx.x] = 0; }
    }
  }
  out[tid] = acc;
}
int main() {
  uint32_t* out; CK(hipMalloc(&out, 64 << 20));
  const uint32_t n3 = 736899936u, n4 = 126432037u; const int rows = 32;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int blocks : {256, 512, 1024, 2048}) {
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 0, 0, out, 0x1234567u, 0x89abcdu, n3, n4, rows);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 0, 0, out, 0x1234567u, 0x89abcdu, n3, n4, rows);
        else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 0, 0, out, 0x1234567u, 0x89abcdu, n3, n4, rows);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double sites = (double)blocks * 1024 * rows * 64;
      printf("blocks %5d mode %s: %8.3f ms  %8.1f sites/ns\n", blocks, mode == 0 ? "CMPX" : mode == 1 ? "LUT " : "DSOR", best, sites / best * 1e-6);
    }
  }
  return 0;
}
