// ising_device.hpp -- device-side helpers shared by the kernel translation units: launch geometry constants, the
// compile-time loop, Philox4x32-10 with the hoisted first two rounds (see ising_kernels.hip header comment), wave sum.
#pragma once
#include <hip/hip_runtime.h>
#include "ising_kernels.h"
#include <utility>

namespace ising {
namespace {

constexpr int GROUP = 16;                 // lanes per reference block-row (BLOCK_X, optimized/main.cu:55)
constexpr int THREADS = 256;
constexpr int GROUPS_PER_BLOCK = THREADS / GROUP;
// One workgroup per unit of work, more workgroups than one grid dimension carries (HIP wants grid x block < 2^32 threads per
// dimension; the init kernels of a 2^39-spin lattice have 2^32): the grid folds into two dimensions.
constexpr unsigned FLAT_GRID_X = 1u << 20;
inline dim3 flat_grid(long long blocks) {
	if (blocks <= (long long)FLAT_GRID_X) return dim3((unsigned)(blocks > 0 ? blocks : 1));
	return dim3(FLAT_GRID_X, (unsigned)((blocks + FLAT_GRID_X - 1) / FLAT_GRID_X));
}
__device__ __forceinline__ long long flat_block() { return (long long)blockIdx.y * gridDim.x + blockIdx.x; }

template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
	(f(std::integral_constant<int, Is>{}), ...);
}
// compile-time unrolled loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
	static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
	return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

__device__ __forceinline__ void mul_hilo(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
	const unsigned long long p = (unsigned long long)a * b; // v_mad_u64_u32 / s_mul_hi_u32+s_mul_i32
	hi = (uint32_t)(p >> 32);
	lo = (uint32_t)p;
}

// Per-row Philox state shared by the 16 draw blocks of one lane-row.
struct PhiloxRow {
	uint32_t t_lo1;  // lo(M1*tid)                      -> c1.y
	uint32_t t_hi0;  // hi(M0*c1.x)
	uint32_t t_lo0;  // lo(M0*c1.x)                     -> c2.w
	uint32_t t_e;    // t_lo0 ^ key2.y
};

__device__ __forceinline__ PhiloxRow philox_row_setup(uint32_t tid, uint32_t k0x, uint32_t k2y) {
	PhiloxRow r;
	uint32_t hi1;
	mul_hilo(PHILOX_M1, tid, hi1, r.t_lo1);
	const uint32_t c1x = hi1 ^ k0x;
	mul_hilo(PHILOX_M0, c1x, r.t_hi0, r.t_lo0);
	r.t_e = r.t_lo0 ^ k2y;
	return r;
}

// One Philox4x32-10 block for counter (cx, 0, tid, 0): cx is wave-uniform, everything derived from it alone is
// computed on the scalar unit by the compiler.
// (A/B, rejected: forcing the block-constant round-1/2 values into VGPRs -- so that their XORs are 2-cycle VGPR-VGPR
// ops instead of 4-cycle SGPR-operand ops -- costs 48 VGPRs and ran 3.5 % slower.)
__device__ __forceinline__ void philox_block(const PhiloxRow &pr, uint32_t cx, uint32_t seed_lo, uint32_t seed_hi,
                                             uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3) {
	// round 1 (key 0), scalar half
	uint32_t s_hi0, s_lo0;
	mul_hilo(PHILOX_M0, cx, s_hi0, s_lo0);
	const uint32_t c1z = s_hi0 ^ seed_hi;
	// round 2 (key 1)
	const uint32_t k1x = seed_lo + PHILOX_W0, k1y = seed_hi + PHILOX_W1;
	uint32_t s_hi1, s_lo1;
	mul_hilo(PHILOX_M1, c1z, s_hi1, s_lo1);
	uint32_t c0 = (s_hi1 ^ k1x) ^ pr.t_lo1;
	uint32_t c1 = s_lo1;
	uint32_t c2 = pr.t_hi0 ^ (s_lo0 ^ k1y);
	uint32_t c3 = pr.t_lo0;
	// round 3 (key 2)
	uint32_t kx = seed_lo + 2u * PHILOX_W0, ky = seed_hi + 2u * PHILOX_W1;
	{
		uint32_t hi0, lo0, hi1, lo1;
		mul_hilo(PHILOX_M0, c0, hi0, lo0);
		mul_hilo(PHILOX_M1, c2, hi1, lo1);
		c0 = hi1 ^ (c1 ^ kx);
		c1 = lo1;
		c2 = hi0 ^ pr.t_e; // c3 ^ ky == t_lo0 ^ key2.y
		c3 = lo0;
	}
	// rounds 4..10 (keys 3..9)
#pragma unroll
	for (int r = 3; r < 10; ++r) {
		kx += PHILOX_W0;
		ky += PHILOX_W1;
		uint32_t hi0, lo0, hi1, lo1;
		mul_hilo(PHILOX_M0, c0, hi0, lo0);
		mul_hilo(PHILOX_M1, c2, hi1, lo1);
		c0 = xor3(hi1, c1, kx);
		c1 = lo1;
		c2 = xor3(hi0, c3, ky);
		c3 = lo0;
	}
	o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

// The three block-constant values of philox_block (rounds 1-2 of the wave-uniform counter word, folded with the keys),
// for callers that keep them somewhere else than in SGPRs (an LDS table: a VGPR-VGPR v_xor costs 2 cycles, one with
// an SGPR operand 4).
struct PhiloxBlockConst { uint32_t s0, s1, s2; };
__device__ __forceinline__ PhiloxBlockConst philox_block_const(uint32_t cx, uint32_t seed_lo, uint32_t seed_hi) {
	uint32_t s_hi0, s_lo0, s_hi1, s_lo1;
	mul_hilo(PHILOX_M0, cx, s_hi0, s_lo0);
	mul_hilo(PHILOX_M1, s_hi0 ^ seed_hi, s_hi1, s_lo1);
	PhiloxBlockConst k;
	k.s0 = s_hi1 ^ (seed_lo + PHILOX_W0);
	k.s1 = s_lo0 ^ (seed_hi + PHILOX_W1);
	k.s2 = s_lo1 ^ (seed_lo + 2u * PHILOX_W0);
	return k;
}
__device__ __forceinline__ void philox_block_pre(const PhiloxRow &pr, const PhiloxBlockConst &k, uint32_t seed_lo, uint32_t seed_hi,
                                                 uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3) {
	uint32_t c0 = k.s0 ^ pr.t_lo1, c2 = pr.t_hi0 ^ k.s1, c1, c3 = pr.t_lo0;
	uint32_t kx = seed_lo + 2u * PHILOX_W0, ky = seed_hi + 2u * PHILOX_W1;
	{
		uint32_t hi0, lo0, hi1, lo1;
		mul_hilo(PHILOX_M0, c0, hi0, lo0);
		mul_hilo(PHILOX_M1, c2, hi1, lo1);
		c0 = hi1 ^ k.s2;
		c1 = lo1;
		c2 = hi0 ^ pr.t_e;
		c3 = lo0;
	}
#pragma unroll
	for (int r = 3; r < 10; ++r) {
		kx += PHILOX_W0;
		ky += PHILOX_W1;
		uint32_t hi0, lo0, hi1, lo1;
		mul_hilo(PHILOX_M0, c0, hi0, lo0);
		mul_hilo(PHILOX_M1, c2, hi1, lo1);
		c0 = xor3(hi1, c1, kx);
		c1 = lo1;
		c2 = xor3(hi0, c3, ky);
		c3 = lo0;
	}
	o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
	for (int off = 32; off; off >>= 1) v += __shfl_down(v, off, 64);
	return v;
}

// truth table of a 3-input bit function for v_bitop3_b32: bit (a << 2 | b << 1 | c) = f(a, b, c)
template <typename F>
constexpr uint32_t tt3(F f) {
	uint32_t t = 0;
	for (int i = 0; i < 8; ++i) t |= (f((i >> 2) & 1, (i >> 1) & 1, i & 1) ? 1u : 0u) << i;
	return t;
}
#define BITOP3(a, b, c, ...) __builtin_amdgcn_bitop3_b32((a), (b), (c), tt3([](int x, int y, int z) { return (__VA_ARGS__); }))

// Metropolis flips of 32 sites: bit-sliced neighbour count n = up + dw + ct + sd (n0, k1 + k2 = the two carries into
// bit 1), a = aligned neighbours = n for an up spin, 4 - n for a down spin; a <= 2 always flips, a = 3 / 4 flips
// where the draw was below n3 / n4 (masks c3 / c4).  Same function as neighbour_planes + flip_mask of
// ising_dense.hip, arranged as six 3-input operations and six 2-input ones.
__device__ __forceinline__ uint32_t flips32(uint32_t me, uint32_t up, uint32_t ct, uint32_t dw, uint32_t sd, uint32_t c3, uint32_t c4) {
	const uint32_t s1 = BITOP3(up, dw, ct, x ^ y ^ z);
	const uint32_t k1 = BITOP3(up, dw, ct, (x & y) | (z & (x ^ y)));
	const uint32_t n0 = s1 ^ sd, k2 = s1 & sd;            // n1 = k1 ^ k2, n2 = k1 & k2
	// a == 3: n0 set and bit 1 of n equal to the spin (n = 3 up, n = 1 down)
	const uint32_t is3 = n0 & BITOP3(me, k1, k2, !(x ^ y ^ z));
	// a == 4: n == 4 for an up spin, n == 0 for a down spin
	const uint32_t p4 = BITOP3(me, k1, k2, x ? (y & z) : !(y | z));
	const uint32_t not4 = BITOP3(p4, me, n0, !(x & (y | !z)));
	return BITOP3(is3, c3, not4 | c4, x ? y : z);
}

} // namespace
} // namespace ising
