// modarith.hpp -- modular arithmetic for q = 8380417 on CDNA4 (gfx950) VALUs.
//
// Measured on MI355X (profiles/r01_ubench_valu_rates.txt): v_add/v_sub/v_and/v_xor/shifts-right
// issue at ~2.5 cycles per wave64 instruction; EVERY other integer op -- v_mul_lo_u32,
// v_mul_hi_{u,i}32, the 24-bit multiplies, v_mad_*, v_min/max, v_add3, v_bfi, v_cndmask(e64),
// DPP moves -- issues at ~4.4 cycles, v_mad_{u,i}64_{u,i}32 at ~5.2, and VCC-form v_cndmask at
// ~22.  So 32-bit multiplies cost the same as 24-bit ones, a 64-bit multiply-accumulate costs
// barely more than one multiply, and the cheapest exact reduction is signed Montgomery with
// R = 2^32 (3 multiplies + 1 subtract for a constant operand, no range fix-ups):
//
//     mont_tw(y, w)      = y * w            (w stored as w~ = w * 2^32 mod q, and w~ * q^-1)
//     mont_red64(p)      = p * 2^-32 mod q  (p a 64-bit sum of products)
//
// Arithmetic spec being matched bit-exactly mod q: Barrett_8380417.v:146-283 (modmul),
// butterfly.v:27-250 (op set), ref_ntt.cpp:28-87 (C model).  Values in flight are lazy signed
// residues (any int32 congruent to the true value); canonical [0, q) only at kernel outputs
// (the RTL convention, butterfly.v:194-195).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dil {

constexpr int32_t Q = 8380417;             // 2^23 - 2^13 + 1   (params.h:33)
constexpr uint32_t QINV = 58728449u;       // q^-1 mod 2^32
constexpr int32_t F256 = 8347681;          // 256^-1 mod q      (ref_ntt.cpp:64)

// signed high product as ONE v_mul_hi_i32.  (Left to itself hipcc sometimes hoists the sign
// extension of a loop-invariant operand and then expands the product into 3 multiplies + fix-ups.)
__device__ __forceinline__ int32_t mulhi_i32(int32_t a, int32_t b)
{
    int32_t d;
    asm("v_mul_hi_i32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// sign mask (all-ones iff x < 0) as ONE v_ashrrev_i32, opaque to the optimiser: left visible,
// LLVM canonicalises (x >> 31) & c into v_cmp + VCC-form v_cndmask, which issues ~5x slower here.
__device__ __forceinline__ int32_t sgn(int32_t x)
{
    int32_t m;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m) : "v"(x));
    return m;
}

// Two forms of the constant product, chosen per translation unit (constexpr MAD64 below):
//   0  v_mul_lo_u32 + 2 x v_mul_hi_i32 + v_sub          (4 instructions, ~15.7 issue cycles; needs wq = wt q^-1)
//   1  p = y wt as v_mad_i64_i32, m = lo32(p) q^-1, hi32(p - m q) as a second v_mad_i64_i32   (3 instructions, ~14.8 cycles)
// Same integer either way (p - m q has a zero low word, so its high word is hi(p) - hi(m q)).  Measured (profiles/r04c_*):
// the VALU-bound fused pipelines gain 2-2.6 % with form 1, the HBM-bound standalone transforms LOSE 5 % (the 64-bit pairs
// cost registers there), so pipelines.hip / wire_kernels.hip select 1 and kernels.hip keeps 0.
// One form per translation unit: pipelines.hip and wire_kernels.hip define DIL_PRODUCT_MAD64 before they include this header.
#ifdef DIL_PRODUCT_MAD64
constexpr bool MAD64 = true;
#else
constexpr bool MAD64 = false;
#endif
// (the instruction also writes a carry mask: it goes to a scratch SGPR pair of the compiler's choosing; VCC measured the same)
__device__ __forceinline__ int64_t mad64(int32_t a, int32_t b, int64_t c)     // a * b + c, one v_mad_i64_i32
{
    int64_t d;
    uint64_t carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int64_t mul64(int32_t a, int32_t b)                 // a * b, one v_mad_i64_i32 with a zero addend
{
    int64_t d;
    uint64_t carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ int64_t mul64_s(int32_t a, uint32_t b_scalar)       // the same with a wave-uniform (SGPR) factor
{
    int64_t d;
    uint64_t carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "s"(b_scalar));
    return d;
}

// y * w for a table constant w = (wt, wq):  wt = centred(w * 2^32 mod q), wq = wt * q^-1 mod 2^32.
// Any int32 y; |result| < q (|y * wt| < 2^31 * q/2).
__device__ __forceinline__ int32_t mont_tw(int32_t y, int32_t wt, uint32_t wq)
{
    if constexpr (MAD64) {
        const int64_t p = mul64(y, wt);
        const int32_t m = (int32_t)((uint32_t)p * QINV);
        return (int32_t)(mad64(m, -Q, p) >> 32);
    } else {
        const int32_t m = (int32_t)((uint32_t)y * wq);
        return mulhi_i32(y, wt) - mulhi_i32(m, Q);
    }
}
// the same with a wave-uniform constant as the SCALAR operand of the multiplies (ntt_core.hpp TwLdsC: the uniform pass)
__device__ __forceinline__ int32_t mont_tw_s(int32_t y, uint32_t wt, uint32_t wq)
{
    if constexpr (MAD64) {
        const int64_t p = mul64_s(y, wt);
        const int32_t m = (int32_t)((uint32_t)p * QINV);
        return (int32_t)(mad64(m, -Q, p) >> 32);
    } else {
        int32_t m, h;
        asm("v_mul_lo_u32 %0, %1, %2" : "=v"(m) : "v"(y), "s"(wq));
        asm("v_mul_hi_i32 %0, %1, %2" : "=v"(h) : "v"(y), "s"(wt));
        return h - mulhi_i32(m, Q);
    }
}

// p * 2^-32 mod q for |p| < 2^31 * q;  |result| < q.
__device__ __forceinline__ int32_t mont_red64(int64_t p)
{
    const int32_t m = (int32_t)((uint32_t)p * QINV);
    if constexpr (MAD64) {
        return (int32_t)(mad64(m, -Q, p) >> 32);
    } else {
        return (int32_t)(p >> 32) - mulhi_i32(m, Q);
    }
}

// a * b * 2^-32 mod q (generic Montgomery product; v_mad_i64_i32 gives the 64-bit product)
__device__ __forceinline__ int32_t mont_mul(int32_t a, int32_t b) { return mont_red64((int64_t)a * (int64_t)b); }

// (-q, q) -> [0, q)
__device__ __forceinline__ uint32_t canon_small(int32_t t) { return (uint32_t)(t + (sgn(t) & Q)); }

// any int32 with |x| < 2^31 - 2^22 -> [0, q):  x - round(x / 2^23) * q lies in (-q, q)
__device__ __forceinline__ uint32_t canon_any(int32_t x)
{
    const int32_t k = (x + (1 << 22)) >> 23;
    return canon_small(x - k * Q);
}

// (-2q, 2q) -> [0, q) by unsigned minima: x + 2q in (0, 4q), then min(t, t - 2q), min(t, t - q) (a subtraction that wraps is never the
// minimum).  add + 2 x (sub + v_min_u32): 16 issue cycles against canon_any's 20, for sums / differences of two residues in (-q, q).
__device__ __forceinline__ uint32_t canon_pm2q(int32_t x)
{
    uint32_t t = (uint32_t)x + 2u * (uint32_t)Q;
    t = min(t, t - 2u * (uint32_t)Q);
    return min(t, t - (uint32_t)Q);
}
// r in (-q, q), any representative of a value whose true size is known: V = v + off with 0 <= V < q  ->  V exactly.
// x = r + off is V - q, V or V + q; as unsigned numbers exactly one of x, x + q, x - q is V and it is the smallest (a wrapped one is
// huge): add + 2 add + v_min3_u32.  This is how the pipelines read SMALL results (c s, c t0, w0 - c s2) off a lazy residue without
// canonicalising it first.
__device__ __forceinline__ uint32_t min3_u32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t exact_plus(int32_t r, uint32_t off)
{
    const uint32_t x = (uint32_t)r + off;
    return min3_u32(x, x + (uint32_t)Q, x - (uint32_t)Q);
}
// [0, 2q) -> [0, q)
__device__ __forceinline__ uint32_t canon_2q(uint32_t x) { return min(x, x - (uint32_t)Q); }

// the constant 2^32 mod q in table form (wt = 2^64 mod q centred): mont_tw(mont_mul(a, b), R2_WT, R2_WQ) == a * b mod q
constexpr int32_t R2_WT = 2365951;
constexpr uint32_t R2_WQ = 2145647103u;        // R2_WT * q^-1 mod 2^32   (checked in tests/test_model_and_cabi.py)

// Cooley-Tukey butterfly (ref_ntt.cpp:39-44 / butterfly.v FORWARD_NTT_MODE), lazy signed:
//   x' = x + w*y,  y' = x - w*y.   |w*y| < 0.75 q, so each layer widens x by < q.
__device__ __forceinline__ void ct_bfly(int32_t& x, int32_t& y, int32_t wt, uint32_t wq)
{
    const int32_t t = mont_tw(y, wt, wq);
    y = x - t;
    x = x + t;
}

// Gentleman-Sande butterfly (ref_ntt.cpp:76-81 / butterfly.v INVERSE_NTT_MODE), lazy signed:
//   x' = x + y,  y' = (x - y) * w.   The sums double per layer: 8 layers from |x| < q stay
//   below 256 q < 2^31.
__device__ __forceinline__ void gs_bfly(int32_t& x, int32_t& y, int32_t wt, uint32_t wq)
{
    const int32_t d = x - y;
    x = x + y;
    y = mont_tw(d, wt, wq);
}

}  // namespace dil
