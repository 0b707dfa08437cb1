// Instruction-rate and HBM microbenchmarks for gfx950 (MI355X).
// Test infrastructure only: informs the kernel design in DESIGN.md (Philox ALU ceiling, HBM ceiling).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;

// Eight independent dependency chains per lane; each asm statement is one VALU instruction.
#define OP8(INS) \
  asm volatile(INS : "+v"(a0) : "v"(b), "s"(s)); asm volatile(INS : "+v"(a1) : "v"(b), "s"(s)); \
  asm volatile(INS : "+v"(a2) : "v"(b), "s"(s)); asm volatile(INS : "+v"(a3) : "v"(b), "s"(s)); \
  asm volatile(INS : "+v"(a4) : "v"(b), "s"(s)); asm volatile(INS : "+v"(a5) : "v"(b), "s"(s)); \
  asm volatile(INS : "+v"(a6) : "v"(b), "s"(s)); asm volatile(INS : "+v"(a7) : "v"(b), "s"(s));

#define KERNEL32(NAME, INS) \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned s) { \
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
  unsigned b = blockIdx.x * 2654435761u + 12345u; \
  for (int i = 0; i < ITERS; ++i) { OP8(INS) OP8(INS) } \
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
}

KERNEL32(k_xor,     "v_xor_b32 %0, %0, %1")
KERNEL32(k_bitop3,  "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL32(k_add,     "v_add_u32 %0, %0, %1")
KERNEL32(k_mullo,   "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mulhi,   "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul24,   "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad24,   "v_mad_u32_u24 %0, %0, %1, %0")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_addc,    "v_addc_co_u32 %0, vcc, %0, %0, vcc")
KERNEL32(k_cmp,     "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_lshlor,  "v_lshl_or_b32 %0, %0, 1, %1")
KERNEL32(k_fma,     "v_fma_f32 %0, %0, %1, %0")
KERNEL32(k_cmp_s,   "v_cmp_lt_u32 s[20:21], %0, %2")
KERNEL32(k_dpp,     "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_add_lit,  "v_add_u32 %0, 0x10000000, %0")
KERNEL32(k_add_sgpr, "v_add_u32 %0, %2, %0")
KERNEL32(k_add_inl,  "v_add_u32 %0, 16, %0")
KERNEL32(k_and,      "v_and_b32 %0, %0, %1")
KERNEL32(k_or,       "v_or_b32 %0, %0, %1")
KERNEL32(k_sub,      "v_sub_u32 %0, %0, %1")
KERNEL32(k_lshl,     "v_lshlrev_b32 %0, 3, %0")
KERNEL32(k_lshr,     "v_lshrrev_b32 %0, 3, %0")
KERNEL32(k_min,      "v_min_u32 %0, %0, %1")
KERNEL32(k_mov,      "v_mov_b32 %0, %1")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 28")
KERNEL32(k_add3,     "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_bfe,      "v_bfe_u32 %0, %0, 4, 4")
KERNEL32(k_perm,     "v_perm_b32 %0, %0, %1, %2")
KERNEL32(k_xor_sg,   "v_xor_b32 %0, %2, %0")
KERNEL32(k_cmpx,     "v_cmpx_gt_u32 vcc, %2, %0\n\ts_mov_b64 exec, -1")
// the accept pattern: cmpx, add, cmpx, add, restore exec  (counts as 1 "op" = one site)
KERNEL32(k_accept,   "v_cmpx_gt_u32 vcc, %2, %1\n\tv_add_u32 %0, 16, %0\n\tv_cmpx_gt_u32 vcc, %2, %1\n\tv_add_u32 %0, 16, %0\n\ts_mov_b64 exec, -1")
KERNEL32(k_accept_lit, "v_cmpx_gt_u32 vcc, %2, %1\n\tv_add_u32 %0, 0x10000000, %0\n\tv_cmpx_gt_u32 vcc, %2, %1\n\tv_add_u32 %0, 0x10000000, %0\n\ts_mov_b64 exec, -1")
// compare+select+add as the compiler writes it
KERNEL32(k_cmpsel,   "v_cmp_gt_u32 vcc, %2, %1\n\ts_nop 1\n\tv_cndmask_b32 %1, 0, %1, vcc\n\tv_add_u32 %0, %1, %0")

// shader clock vs 100 MHz real-time counter under a VALU-heavy load
__global__ void __launch_bounds__(256) k_clock(unsigned long long* out, unsigned s) {
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned b = blockIdx.x * 2654435761u + 12345u;
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < ITERS * 4; ++i) { OP8("v_mul_hi_u32 %0, %0, %1") }
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345) out[0] = 0;
}

// v_mad_u64_u32: 64-bit destination -> use 64-bit lane values
#define OP8W(INS) \
  asm volatile(INS : "+v"(w0) : "v"(b), "s"(s) : "vcc"); asm volatile(INS : "+v"(w1) : "v"(b), "s"(s) : "vcc"); \
  asm volatile(INS : "+v"(w2) : "v"(b), "s"(s) : "vcc"); asm volatile(INS : "+v"(w3) : "v"(b), "s"(s) : "vcc"); \
  asm volatile(INS : "+v"(w4) : "v"(b), "s"(s) : "vcc"); asm volatile(INS : "+v"(w5) : "v"(b), "s"(s) : "vcc"); \
  asm volatile(INS : "+v"(w6) : "v"(b), "s"(s) : "vcc"); asm volatile(INS : "+v"(w7) : "v"(b), "s"(s) : "vcc");

#define KERNEL64(NAME, INS) \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned s) { \
  unsigned long long w0 = threadIdx.x, w1 = w0 + 1, w2 = w0 + 2, w3 = w0 + 3, w4 = w0 + 4, w5 = w0 + 5, w6 = w0 + 6, w7 = w0 + 7; \
  unsigned b = blockIdx.x * 2654435761u + 12345u; \
  for (int i = 0; i < ITERS; ++i) { OP8W(INS) OP8W(INS) } \
  unsigned long long r = w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7; \
  out[blockIdx.x * 256 + threadIdx.x] = (unsigned)r ^ (unsigned)(r >> 32); \
}
// %L0 / %H0 not available: operate on the pair; src0 = low half register of the pair via sub-register syntax is
// not expressible in inline asm, so use the whole-pair form d = lo(d)*b + d  via a temp operand trick:
KERNEL64(k_mad64,   "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_mad64z,  "v_mad_u64_u32 %0, vcc, %1, %2, 0")
KERNEL64(k_lshladd64, "v_lshl_add_u64 %0, %0, 1, %0")
// round 6: the quad word phase's candidates (64-bit shifts against funnel shifts, packed 16-bit shifts, bit-field insert, and-or)
KERNEL64(k_lshl64,    "v_lshlrev_b64 %0, 16, %0")
KERNEL64(k_lshr64,    "v_lshrrev_b64 %0, 15, %0")
KERNEL64(k_lshl64v,   "v_lshlrev_b64 %0, %1, %0")
KERNEL32(k_pklshl16,  "v_pk_lshlrev_b16 %0, 1, %0 op_sel_hi:[0,1]")
KERNEL32(k_pklshl16v, "v_pk_lshlrev_b16 %0, %1, %0")
KERNEL32(k_bfi,       "v_bfi_b32 %0, %1, %0, %1")
KERNEL32(k_andor,     "v_and_or_b32 %0, %0, %1, %1")
KERNEL32(k_or3,       "v_or3_b32 %0, %0, %1, %1")
KERNEL32(k_alignbitv, "v_alignbit_b32 %0, %0, %1, 16")
KERNEL32(k_accread,   "v_accvgpr_read_b32 %0, a0")
KERNEL32(k_readlane,  "v_readlane_b32 s20, %0, 3")
KERNEL32(k_bperm,     "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)")

// Full Philox4x32-10 block throughput, generic (all 10 rounds with vector multiplies)
__device__ __forceinline__ unsigned x3(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

template <int FIRST_ROUND>
__global__ void __launch_bounds__(256) k_philox(unsigned* out, unsigned seedlo, unsigned seedhi, int nblk) {
  unsigned tid = threadIdx.x + blockIdx.x * 256;
  unsigned acc = 0;
  for (int i = 0; i < nblk; ++i) {
    unsigned c0 = i, c1 = 0, c2 = tid, c3 = 0;
    unsigned k0 = seedlo, k1 = seedhi;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
      unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
      unsigned n0 = x3((unsigned)(p1 >> 32), c1, k0), n1 = (unsigned)p1;
      unsigned n2 = x3((unsigned)(p0 >> 32), c3, k1), n3 = (unsigned)p0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    // consume the four outputs with two compares each, like the accept test
    acc += (c0 < seedlo) + (c1 < seedlo) + (c2 < seedlo) + (c3 < seedlo);
    acc += (c0 < seedhi) + (c1 < seedhi) + (c2 < seedhi) + (c3 < seedhi);
  }
  out[tid] = acc;
}

// Philox only, XOR-reduced (no compares)
__global__ void __launch_bounds__(256) k_philox_only(unsigned* out, unsigned seedlo, unsigned seedhi, int nblk) {
  unsigned tid = threadIdx.x + blockIdx.x * 256;
  unsigned acc = 0;
  for (int i = 0; i < nblk; ++i) {
    unsigned c0 = i, c1 = 0, c2 = tid, c3 = 0;
    unsigned k0 = seedlo, k1 = seedhi;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
      unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
      unsigned n0 = x3((unsigned)(p1 >> 32), c1, k0), n1 = (unsigned)p1;
      unsigned n2 = x3((unsigned)(p0 >> 32), c3, k1), n3 = (unsigned)p0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    acc ^= x3(c0, c1, c2) ^ c3;
  }
  out[tid] = acc;
}

// HBM: copy / read / write with 16 B per lane, grid-stride
__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x, st = (size_t)gridDim.x * 256;
  for (; i < n; i += st) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ src, unsigned* out, size_t n) {
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x, st = (size_t)gridDim.x * 256;
  unsigned acc = 0;
  for (; i < n; i += st) { uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(uint4* __restrict__ dst, size_t n) {
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x, st = (size_t)gridDim.x * 256;
  for (; i < n; i += st) dst[i] = make_uint4((unsigned)i, 1, 2, 3);
}
// 2 reads + 1 write (the half-sweep's traffic shape: src, dst-in, dst-out)
__global__ void __launch_bounds__(256) k_rrw(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x, st = (size_t)gridDim.x * 256;
  for (; i < n; i += st) { uint4 x = a[i], y = b[i]; b[i] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w); }
}

template <typename F>
static float time_ms(F&& f, int reps = 5) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount; double ghz = p.clockRate * 1e-6;
  printf("device: %s, CUs %d, clock %.3f GHz, gcn %s\n", p.name, ncu, ghz, p.gcnArchName);
  unsigned* out; CK(hipMalloc(&out, 64u << 20));
  const int blocks = ncu * 8;
#define RUN(NAME) { float ms = time_ms([&]{ hipLaunchKernelGGL(NAME, dim3(blocks), dim3(256), 0, 0, out, 12345u); }); \
    double ops = (double)blocks * 256 * ITERS * 16; \
    printf("%-12s %8.3f ms  %9.1f Glane-ops/s  %6.2f lanes/clk/CU @%.2fGHz\n", #NAME, ms, ops / ms * 1e-6, ops / (ms * 1e-3) / (ncu * ghz * 1e9), ghz); }
  RUN(k_xor) RUN(k_bitop3) RUN(k_add) RUN(k_fma) RUN(k_lshlor) RUN(k_cndmask) RUN(k_addc) RUN(k_cmp) RUN(k_cmp_s) RUN(k_dpp)
  RUN(k_add_lit) RUN(k_add_sgpr) RUN(k_add_inl) RUN(k_and) RUN(k_or) RUN(k_sub) RUN(k_lshl) RUN(k_lshr) RUN(k_min) RUN(k_mov) RUN(k_xor_sg)
  RUN(k_alignbit) RUN(k_add3) RUN(k_bfe) RUN(k_perm) RUN(k_cmpx) RUN(k_accept) RUN(k_accept_lit) RUN(k_cmpsel)
  {
    unsigned long long* d; CK(hipMalloc(&d, blocks * 16));
    hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(256), 0, 0, d, 12345u); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 2); CK(hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost));
    double sc = 0, sr = 0; for (int i = 0; i < blocks; ++i) { sc += h[2 * i]; sr += h[2 * i + 1]; }
    int wc = 0; CK(hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0));
    printf("clock under v_mul_hi_u32 load: shader cycles / wall ticks = %.3f ; wall clock rate %d kHz -> %.3f GHz; cycles per wave-instr = %.2f\n",
           sc / sr, wc, sc / sr * wc * 1e-6, sc / blocks / (ITERS * 4.0 * 8) / 1.0);
  }
  RUN(k_mul24) RUN(k_mad24) RUN(k_mullo) RUN(k_mulhi) RUN(k_mad64) RUN(k_mad64z) RUN(k_lshladd64)
  RUN(k_lshl64) RUN(k_lshr64) RUN(k_lshl64v) RUN(k_pklshl16) RUN(k_pklshl16v) RUN(k_bfi) RUN(k_andor) RUN(k_or3) RUN(k_alignbitv) RUN(k_accread) RUN(k_readlane) RUN(k_bperm)
  {
    int nblk = 2048;
    float ms = time_ms([&]{ hipLaunchKernelGGL(k_philox<0>, dim3(blocks), dim3(256), 0, 0, out, 0x1234567u, 0x89abcdeu, nblk); });
    double nb = (double)blocks * 256 * nblk;
    printf("philox+8cmp  %8.3f ms  %9.2f Gblocks/s = %8.1f sites/ns\n", ms, nb / ms * 1e-6, 4 * nb / ms * 1e-6);
    ms = time_ms([&]{ hipLaunchKernelGGL(k_philox_only, dim3(blocks), dim3(256), 0, 0, out, 0x1234567u, 0x89abcdeu, nblk); });
    printf("philox only  %8.3f ms  %9.2f Gblocks/s = %8.1f sites/ns\n", ms, nb / ms * 1e-6, 4 * nb / ms * 1e-6);
  }
  {
    size_t bytes = (size_t)2 << 30, n = bytes / 16;
    uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    for (int g : {ncu * 4, ncu * 8, ncu * 16, ncu * 32}) {
      float ms = time_ms([&]{ hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n); });
      printf("copy  grid %6d: %7.3f ms  %7.1f GB/s (r+w)\n", g, ms, 2.0 * bytes / ms * 1e-6);
      ms = time_ms([&]{ hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, out, n); });
      printf("read  grid %6d: %7.3f ms  %7.1f GB/s\n", g, ms, 1.0 * bytes / ms * 1e-6);
      ms = time_ms([&]{ hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, b, n); });
      printf("write grid %6d: %7.3f ms  %7.1f GB/s\n", g, ms, 1.0 * bytes / ms * 1e-6);
      ms = time_ms([&]{ hipLaunchKernelGGL(k_rrw, dim3(g), dim3(256), 0, 0, a, b, n); });
      printf("rrw   grid %6d: %7.3f ms  %7.1f GB/s (2r+1w)\n", g, ms, 3.0 * bytes / ms * 1e-6);
    }
    // non-grid-stride copy: one 16 B element per thread
    float ms = time_ms([&]{ hipLaunchKernelGGL(k_copy, dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, b, n); });
    printf("copy  1 elem/thread: %7.3f ms  %7.1f GB/s (r+w)\n", ms, 2.0 * bytes / ms * 1e-6);
  }
  return 0;
}
