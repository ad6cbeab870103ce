// modarith.hpp -- 23-bit modular arithmetic for q = 8380417 on CDNA4 (gfx950) VALUs.
//
// Design rule: 32-bit integer multiplies (v_mul_lo_u32 / v_mul_hi_u32) are quarter rate on
// CDNA; the 24-bit forms v_mul_u32_u24 / v_mul_hi_u32_u24 / v_mad_u32_u24 are full rate.
// q < 2^23 and 2q < 2^24, so every multiplier operand is kept below 2^24 and every product
// is taken with the 24-bit instructions.  hipcc selects them from the masked C expressions
// below (the masks themselves fold away: the instructions ignore bits 31:24).
//
// Arithmetic spec being matched (bit-exact mod q): Barrett_8380417.v:146-283 (modmul),
// butterfly.v:27-250 (op set), ref_ntt.cpp:28-87 (C model).  Values are "lazy" residues:
// any uint32 congruent to the true value; canonical [0,q) only at kernel outputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dil {

constexpr uint32_t Q = 8380417u;        // 2^23 - 2^13 + 1   (params.h:33)
constexpr uint32_t Q2 = 2u * Q;
constexpr uint32_t MU46 = 8396807u;     // floor(2^46 / q)   (Barrett_8380417.v:189-219)
constexpr uint32_t F256 = 8347681u;     // 256^-1 mod q      (ref_ntt.cpp:64)

__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b)
{
    return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);                       // v_mul_u32_u24
}
__device__ __forceinline__ uint32_t mulhi24(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu)) >> 32);  // v_mul_hi_u32_u24
}
// keep a value opaque to the optimiser (stops it re-associating a 24-bit product into a
// quarter-rate 32-bit multiply by a negative constant)
__device__ __forceinline__ uint32_t opaque(uint32_t x)
{
    asm("" : "+v"(x));
    return x;
}

// any uint32 x  ->  x - floor(x / 2^23) * q  in [0, 2^23 + 2^22)  (2 instructions)
// (Barrett with the quotient estimate x>>23: q = 2^23 - 2^13 + 1, so the remainder is
//  x mod 2^23 + (x>>23) * 8191.)
__device__ __forceinline__ uint32_t red(uint32_t x)
{
    return x + (uint32_t)((int)(x >> 23) * (-(int)Q));
}

// [0, 2q) -> [0, q)
__device__ __forceinline__ uint32_t csub(uint32_t x)
{
    uint32_t y = x - Q;
    return y < x ? y : x;                                           // v_min_u32(x, x - q)
}

// any uint32 -> canonical [0, q)
__device__ __forceinline__ uint32_t canon(uint32_t x) { return csub(red(x)); }

// int32 in (-q, q) (or already canonical) -> canonical [0, q): min(x, x + q) as unsigned
__device__ __forceinline__ uint32_t canon_signed(int32_t v)
{
    uint32_t x = (uint32_t)v, y = x + Q;
    return y < x ? y : x;
}

// Shoup / Harvey multiplication by a constant w < q with companion wp = floor(w * 2^24 / q):
//   y < 2^24  ->  y * w mod q  in [0, 2q).   5 full-rate instructions.
__device__ __forceinline__ uint32_t shoup_mul(uint32_t y, uint32_t w, uint32_t wp)
{
    uint32_t qe = __builtin_amdgcn_alignbit(mulhi24(y, wp), mul24(y, wp), 24);
    return mul24(y, w) - opaque(mul24(qe, Q));
}

// Cooley-Tukey butterfly (ref_ntt.cpp:39-44 / butterfly.v FORWARD_NTT_MODE), lazy:
//   x' = x + w*y,  y' = x - w*y + 2q.   x, y any uint32 small enough not to overflow
//   (each layer adds at most 2q); y is pulled below 2^24 by red() for the multiplier.
__device__ __forceinline__ void ct_bfly(uint32_t& x, uint32_t& y, uint32_t w, uint32_t wp)
{
    uint32_t yr = red(y);
    uint32_t qe = __builtin_amdgcn_alignbit(mulhi24(yr, wp), mul24(yr, wp), 24);
    uint32_t xn = (mul24(yr, w) + x) - opaque(mul24(qe, Q));        // v_mad_u32_u24 + v_sub
    uint32_t b = opaque((x << 1) + Q2);                             // v_lshl_add_u32
    y = b - xn;
    x = xn;
}

// Gentleman-Sande butterfly (ref_ntt.cpp:76-81 / butterfly.v INVERSE_NTT_MODE), lazy:
//   x' = x + y,  y' = (x - y) * w.   BY = static bound on y in units of q.
template <uint32_t BY>
__device__ __forceinline__ void gs_bfly(uint32_t& x, uint32_t& y, uint32_t w, uint32_t wp)
{
    uint32_t d = x + (BY * Q) - y;
    x = x + y;
    y = shoup_mul(red(d), w, wp);
}

// Barrett product of two canonical residues, exactly the RTL datapath
// (Barrett_8380417.v: quo = ((x >> 22) * 8396807) >> 24): a, b in [0, q) -> a*b mod q in [0, 2q)
__device__ __forceinline__ uint32_t mulmod_lazy(uint32_t a, uint32_t b)
{
    uint32_t lo = mul24(a, b), hi = mulhi24(a, b);
    uint32_t xs = __builtin_amdgcn_alignbit(hi, lo, 22);
    uint32_t qe = __builtin_amdgcn_alignbit(mulhi24(xs, MU46), mul24(xs, MU46), 24);
    return lo - opaque(mul24(qe, Q));
}
__device__ __forceinline__ uint32_t mulmod(uint32_t a, uint32_t b) { return csub(mulmod_lazy(a, b)); }

}  // namespace dil
