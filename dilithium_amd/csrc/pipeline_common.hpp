// pipeline_common.hpp -- device helpers of the fused Dilithium pipelines (pipelines.hip):
// parameter sets, Decompose / UseHint / MakeHint / norm checks, LDS table staging, matrix-row
// streaming and the 64-bit multiply-accumulate.
#pragma once
#include "device_common.hpp"
#include "kernels.hpp"
#include "ntt_core.hpp"

namespace dil {

// The wave's index inside its workgroup as a SCALAR: threadIdx.x >> 6 is the same in all 64 lanes, but the compiler does
// not know that; read through v_readfirstlane it lands in an SGPR, and so does everything derived from it -- the item
// number and every per-item base pointer of the wave-per-item kernels -- which turns 64-bit VGPR address arithmetic into
// scalar arithmetic + `global_load ... v_off, s[base]` and frees the VGPR pairs the pointers were held in.
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// (1:0) exchange of the transforms inside the wave-per-item pipelines (ntt_core.hpp): through LDS where the kernel's
// LDS budget has the 1 KiB per wave to spare (USE = true), else in registers.
template <bool USE>
struct X10Pick {
    using type = X10Dpp;
    static constexpr int DW = 0;
};
template <>
struct X10Pick<true> {
    using type = X10Lds;
    static constexpr int DW = 256;          // LDS dwords per wave
};

// ---------------------------------------------------------------------------------------
// Dilithium element-wise tail: Decompose / UseHint / MakeHint / norm checks
// ---------------------------------------------------------------------------------------
template <int LEVEL>
struct Par;
template <>
struct Par<2> {
    static constexpr int K = 4, L = 4, OMEGA = 80, BETA = 78, TAU = 39;
    static constexpr int32_t GAMMA1 = 1 << 17, GAMMA2 = (Q - 1) / 88;
};
template <>
struct Par<3> {
    static constexpr int K = 6, L = 5, OMEGA = 55, BETA = 196, TAU = 49;
    static constexpr int32_t GAMMA1 = 1 << 19, GAMMA2 = (Q - 1) / 32;
};
template <>
struct Par<5> {
    static constexpr int K = 8, L = 7, OMEGA = 75, BETA = 120, TAU = 60;
    static constexpr int32_t GAMMA1 = 1 << 19, GAMMA2 = (Q - 1) / 32;
};

// a canonical -> (a1 = HighBits, a0 = LowBits centred in (-gamma2, gamma2]); equals the RTL's
// threshold map decomp_map1.v:37-171 + coeff_decomposer.v:70-89 (checked over all of [0,q))
template <int LEVEL>
__device__ __forceinline__ void decompose(uint32_t a, uint32_t& a1, int32_t& a0)
{
    uint32_t t = (a + 127) >> 7;
    if (LEVEL == 2) {
        t = (t * 11275u + (1u << 23)) >> 24;
        t ^= (uint32_t)sgn((int32_t)(43 - t)) & t;
    } else {
        t = (t * 1025u + (1u << 21)) >> 22;
        t &= 15;
    }
    int32_t r = (int32_t)a - (int32_t)t * (2 * Par<LEVEL>::GAMMA2);
    r -= sgn((Q - 1) / 2 - r) & Q;
    a1 = t;
    a0 = r;
}

// UseHint (usehint.v:140-159) without Decompose.  With v = ceil(a / gamma2) - 1 (half-buckets of width gamma2):
//   a0 > 0  <=>  v even,  a1 = (v + 1) >> 1  (mod 16 / 44),  hinted: a1 + 1 if a0 > 0 else a1 - 1 = a1 + 1 - 2 (v & 1)
// gamma2 = 2^8 * 1023 (levels 3, 5) / 2^9 * 186 (level 2), so v = ((a - 1) >> 8 | 9) * magic >> s exactly over all of
// [0, q) (checked exhaustively against the Decompose form, tests/test_model_and_cabi.py).  One multiply, the rest
// full-rate shifts / adds; compare-free (VCC-form v_cndmask costs ~22 cycles on gfx950, profiles/r01_ubench_valu_rates.txt).
template <int LEVEL>
__device__ __forceinline__ uint32_t use_hint(uint32_t a, uint32_t hint)
{
    const int32_t x = ((int32_t)a - 1) >> (LEVEL == 2 ? 9 : 8);
    const int32_t v = (x * (LEVEL == 2 ? 22551 : 32801)) >> (LEVEL == 2 ? 22 : 25);
    int32_t hm = 0 - (int32_t)(hint & 1u);                 // all-ones iff hinted
    asm("" : "+v"(hm));                                    // keep it a mask (no compare + select)
    const int32_t odd = v & 1;
    int32_t n = ((v + 1) >> 1) + (((odd ^ 1) - odd) & hm);
    if (LEVEL == 2) {
        n += sgn(n) & 44;                                  // -1 -> 43
        n -= sgn(43 - n) & 44;                             // 44 -> 0
    } else {
        n &= 15;
    }
    return (uint32_t)n;
}

// MakeHint (makehint.v:98-99), compare-free: hint unless s <= g2, or s > q - g2, or (s == q - g2 and a1 == 0)
template <int LEVEL>
__device__ __forceinline__ uint32_t make_hint(uint32_t s, uint32_t a1)
{
    constexpr int32_t G2 = Par<LEVEL>::GAMMA2;
    const int32_t v = (int32_t)s;                          // s in [0, q)
    const int32_t le = ~sgn(G2 - v);                       // s <= g2
    const int32_t gt = sgn((Q - G2) - v);                  // s >  q - g2
    const int32_t d = v - (Q - G2);
    const int32_t eq = ~sgn(d | -d);                       // s == q - g2
    const int32_t z1 = ~sgn((int32_t)a1 | -(int32_t)a1);   // a1 == 0
    const int32_t none = le | gt | (eq & z1);
    return (uint32_t)(~none) & 1u;
}

// MakeHint as a PREDICATE (one compare chain, the mask lives in SGPRs): with t = (s + gamma2 - 1) mod q the no-hint interval
// s <= gamma2 or s > q - gamma2 is t < 2 gamma2, and s == q - gamma2 is t == q - 1.  3 VALU + 3 compares instead of ~20 bit-field
// operations; the caller turns the wave's mask into bytes / counts (sign phase 2).  Same truth table as make_hint (checked against
// it over all of [0, q) x {a1 == 0, a1 != 0} in tests/test_gpu_dispatch_parity.py through the oracle's outputs).
template <int LEVEL>
__device__ __forceinline__ bool make_hint_p(uint32_t s, uint32_t a1)
{
    constexpr uint32_t G2 = (uint32_t)Par<LEVEL>::GAMMA2;
    const uint32_t t = canon_2q(s + (G2 - 1));
    return t >= 2u * G2 && !(t == (uint32_t)Q - 1u && a1 == 0u);
}
// a wave mask (SGPR pair) -> 0 / 1 per lane as ONE v_cndmask_b32 in its SGPR-mask form (the VCC form issues 5 x slower on gfx950)
__device__ __forceinline__ uint32_t mask_to_01(uint64_t mask)
{
    uint32_t v;
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(v) : "s"(mask));
    return v;
}

__device__ __forceinline__ bool norm_reject(uint32_t x, uint32_t bound)   // norm_check.v:84-105
{
    return x >= bound && x <= (uint32_t)Q - bound;
}

// Two SMALL products from one inverse transform.  c has tau coefficients +-1 and the secret vectors are tiny (|s| <= eta), so
// |c s1|, |c s2| <= beta = tau eta <= 196 while a residue mod q has 23 bits: by linearity INTT(c^ o (s1^ + 2^11 s2^)) =
// c s1 + 2^11 c s2, and both parts are read off the one result exactly -- sign phase 2 needs 1 + 2 K transforms instead of
// 1 + L + 2 K (level 5: 17 for 24).  Exact while |c s1|, |c s2| <= 1023 (|x + 2^11 y| <= 1023 + 2^11 * 1023 < q / 2): every
// polynomial a secret-key BYTE STRING can decode to qualifies at every level (worst case: level 3's 4-bit fields give
// |s| <= 11, tau = 49: 539), not only well-formed keys; the low-level entry points state the bound (include/dil256.h).
// (c t0 reaches 2^18 and has no partner.)  The reference computes the products one by one (combined_top.v:1994-2229); the
// results are the same integers.
struct SmallPair {
    static constexpr int BITS = 11, HALF = 1 << (BITS - 1);
    static constexpr int32_t SHIFT_R = (int32_t)((1ull << (32 + BITS)) % (uint64_t)Q);     // 2^11 in Montgomery form: mont_mul(x, SHIFT_R) = 2^11 x
    static constexpr int32_t OFF = HALF + (HALF << BITS);                                  // makes both digits non-negative
    // c^ s1^ + (2^11 c^) s2^, Montgomery-reduced: |ch s1| < 9 q^2, |cp s2| < q^2 -- far below mont_red64's 2^31 q
    __device__ __forceinline__ static int32_t mul(int32_t ch, int32_t s1, int32_t cp, int32_t s2)
    {
        return mont_red64((int64_t)ch * s1 + (int64_t)cp * s2);
    }
    // one key for the whole launch: its L rows s1^[l] + 2^11 s2^[l] (canonical), formed once per workgroup into LDS
    template <int L>
    __device__ __forceinline__ static void stage_key(int32_t* dst, const int32_t* __restrict__ s1hat, const int32_t* __restrict__ s2hat)
    {
        for (int i = threadIdx.x; i < L * 256; i += blockDim.x) dst[i] = (int32_t)canon_pm2q(s1hat[i] + mont_mul(s2hat[i], SHIFT_R));
    }
    // r in (-q, q), r == x + 2^11 y (mod q) with |x|, |y| < 2^10  ->  x, y
    __device__ __forceinline__ static void split(int32_t r, int32_t& x, int32_t& y)
    {
        const uint32_t t = exact_plus(r, OFF);             // (x + 2^10) + 2^11 (y + 2^10) in [0, 2^22): its own canonical residue
        x = (int32_t)(t & ((1u << BITS) - 1)) - HALF;
        y = (int32_t)(t >> BITS) - HALF;
    }
};

// Sign phase 2, one coefficient of row k after the transforms (FSM2 combined_top.v:1981-2229: norm_check.v:84-105, makehint.v:98-99),
// in EXACT centred integers instead of canonical residues (round 4: 16 instead of 23 instructions per coefficient).
//   w0   LowBits(w) as phase 1 leaves it: the residue in [0, q) of w0c in (-gamma2, gamma2]
//   cs2  c s2, exact (|cs2| <= 1023: SmallPair::split, or small_exact() of a lone row)
//   b    INTT(c^ o t0^): any representative in (-q, q) of ct0, |ct0| <= tau 2^12 < 2^18
// r0c = w0c - cs2 and ct0c are read off with exact_plus(), so r0c + ct0c is the centred representative of (r0 + ct0) mod q and the
// reference's tests on residues become interval tests on small integers:
//   reject 2:  |r0c| >= gamma2 - beta        reject 4:  |ct0c| >= gamma2
//   hint (makehint.v:98-99: none if s <= gamma2, s > q - gamma2, or s == q - gamma2 with w1 == 0)
//          <=>  s_c > gamma2  or  s_c < -gamma2  or  (s_c == -gamma2 and w1 != 0)
template <int LEVEL>
struct Phase2Coef {
    static constexpr int32_t G2 = Par<LEVEL>::GAMMA2, GB = Par<LEVEL>::GAMMA2 - Par<LEVEL>::BETA;
    // C1 = (q - 1) / 2 makes r0c the CENTRED representative of w0 - cs2 for every w0 in [0, q) -- also one that is no LowBits value
    // (the low-level entry point accepts any residue; the interval tests then agree with the reference's on residues everywhere)
    static constexpr uint32_t C1 = (uint32_t)(Q - 1) / 2, C2 = 1u << 18;
    uint32_t u1, u2;          // r0c + C1 in [0, q), ct0c + C2 in [0, 2^19)
    __device__ __forceinline__ Phase2Coef(uint32_t w0, int32_t cs2, int32_t b)
    {
        const uint32_t x = w0 - (uint32_t)cs2 + C1;        // in [C1 - 1023, q + C1 + 1023): r0c + C1, plus q in the upper part
        u1 = min(x, x - (uint32_t)Q);
        u2 = exact_plus(b, C2);
    }
    __device__ __forceinline__ bool rej_r0() const { return u1 - (C1 - (uint32_t)GB + 1u) >= 2u * (uint32_t)GB - 1u; }     // |r0c| >= gamma2 - beta
    __device__ __forceinline__ bool rej_ct0() const { return u2 - (C2 - (uint32_t)G2 + 1u) >= 2u * (uint32_t)G2 - 1u; }   // |ct0c| >= gamma2
    __device__ __forceinline__ bool hint(uint32_t w1) const
    {
        const uint32_t v = u1 + u2;                                              // s_c + C1 + C2
        const bool outside = v - (C1 + C2 - (uint32_t)G2 + 1u) >= 2u * (uint32_t)G2;      // s_c >= gamma2 + 1 or s_c <= -gamma2
        return outside && !(v == C1 + C2 - (uint32_t)G2 && w1 == 0u);
    }
    __device__ __forceinline__ uint32_t r0_canonical() const        // (the signing loop parks r0 in its scratch)
    {
        const int32_t r = (int32_t)(u1 - C1);
        return (uint32_t)(r + (sgn(r) & Q));
    }
};
// a lone row's product (no partner to pair with): lazy residue of a value with |v| <= 1023 -> v
__device__ __forceinline__ int32_t small_exact(int32_t r) { return (int32_t)exact_plus(r, 1024u) - 1024; }

// ---------------------------------------------------------------------------------------
// Fused pipelines.  One workgroup per item (signature / verification), one wave per
// polynomial row; NTT-domain vectors shared through LDS as LAZY signed residues (no
// canonicalisation between stages); twiddles LDS-resident; pointwise products accumulate as
// 64-bit integers (v_mad_i64_i32) and are Montgomery-reduced once per output coefficient --
// the 2^-32 this leaves is cancelled by the pipeline-flavour inverse table (f = 2^32 / 256).
// LDS map (dwords): [0,2048) fwd twiddles | [2048,4096) inv twiddles | 8*256 vec | chat[256] | flags[4]
// ---------------------------------------------------------------------------------------
constexpr int LDS_VEC = 2 * TW_TABLE_DWORDS;
constexpr int LDS_CHAT = LDS_VEC + 8 * 256;
constexpr int LDS_FLAGS = LDS_CHAT + 256;
constexpr int LDS_SCR = LDS_FLAGS + 4;                 // 64 dwords of byte-plane scratch per wave (<= 8 waves)
constexpr int LDS_DWORDS = LDS_SCR + 8 * 64;

__device__ __forceinline__ void stage_tables(uint32_t* lds, const uint32_t* __restrict__ fwd_tab,
                                             const uint32_t* __restrict__ inv_tab)
{
    for (int i = threadIdx.x; i < TW_TABLE_DWORDS / 4; i += blockDim.x) {
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(fwd_tab)[i];
        reinterpret_cast<uint4*>(lds + TW_TABLE_DWORDS)[i] = reinterpret_cast<const uint4*>(inv_tab)[i];
    }
}

// Twiddle tables of a pipeline kernel: the full [pass][half][lane][4] images (16 KiB) or the compact ones (ntt_core.hpp TwLdsC,
// 5.25 KiB, the wave-uniform pass in SGPRs).  PipeTables<C>::DWORDS of LDS at `lds`, stage() before the kernel's first barrier.
template <bool COMPACT>
struct PipeTables;
template <>
struct PipeTables<false> {
    static constexpr int DWORDS = 2 * TW_TABLE_DWORDS;
    using Fwd = TwLds;
    using Inv = TwLds;
    __device__ __forceinline__ static void stage(uint32_t* lds, const uint32_t* __restrict__ f, const uint32_t* __restrict__ i) { stage_tables(lds, f, i); }
    __device__ __forceinline__ static Fwd fwd(const uint32_t* lds, const uint32_t* __restrict__, int lane) { return TwLds{lds, lane}; }
    __device__ __forceinline__ static Inv inv(const uint32_t* lds, const uint32_t* __restrict__, int lane) { return TwLds{lds + TW_TABLE_DWORDS, lane}; }
};
template <>
struct PipeTables<true> {
    static constexpr int DWORDS = 2 * TWC_DWORDS;
    using Fwd = TwLdsC<true>;
    using Inv = TwLdsC<false>;
    __device__ __forceinline__ static void stage(uint32_t* lds, const uint32_t* __restrict__ f, const uint32_t* __restrict__ i)
    {
        for (int g = threadIdx.x; g < TWC_DWORDS / 4; g += blockDim.x) {
            reinterpret_cast<uint4*>(lds)[g] = reinterpret_cast<const uint4*>(f)[twc_source_granule<true>(g)];
            reinterpret_cast<uint4*>(lds + TWC_DWORDS)[g] = reinterpret_cast<const uint4*>(i)[twc_source_granule<false>(g)];
        }
    }
    __device__ __forceinline__ static Fwd fwd(const uint32_t* lds, const uint32_t* __restrict__ f, int lane) { return Fwd(lds, f, lane); }
    __device__ __forceinline__ static Inv inv(const uint32_t* lds, const uint32_t* __restrict__ i, int lane) { return Inv(lds + TWC_DWORDS, i, lane); }
};

// Byte planes (h, w1) move as whole dwords: 64 lanes x 4 bytes = one coalesced 256-B row per
// instruction.  global_store_byte / global_load_ubyte of 64-byte runs measured 2x the whole
// kernel's time (profiles/r01_fused_ablation.txt), so the re-layout between the INTT's strided
// order (lane + 64 m) and packed order (4 lane + j) goes through a 256-B per-wave LDS scratch.
__device__ __forceinline__ void store_row_u8(uint8_t* __restrict__ out_row, const uint32_t (&v)[4], uint32_t* scratch, int lane)
{
    uint8_t* sc = reinterpret_cast<uint8_t*>(scratch);
#pragma unroll
    for (int m = 0; m < 4; m++) sc[lane + 64 * m] = (uint8_t)v[m];
    reinterpret_cast<uint32_t*>(out_row)[lane] = scratch[lane];
}
__device__ __forceinline__ uint32_t load_row_u8(const uint8_t* __restrict__ row, int lane)   // issue early, unpack late
{
    return reinterpret_cast<const uint32_t*>(row)[lane];
}
__device__ __forceinline__ void unpack_row_u8(uint32_t (&v)[4], uint32_t packed, uint32_t* scratch, int lane)
{
    const uint8_t* sc = reinterpret_cast<const uint8_t*>(scratch);
    scratch[lane] = packed;
#pragma unroll
    for (int m = 0; m < 4; m++) v[m] = sc[lane + 64 * m];
}

// w1 on the wire: 4 bits (levels 3, 5) or 6 bits (level 2) per coefficient (encoder.v:96-133)
template <int LEVEL>
struct W1Pack {
    static constexpr int BITS = LEVEL == 2 ? 6 : 4;
    static constexpr int ROW_BYTES = 32 * BITS;       // 192 / 128 per polynomial
};

// w1 row (4 values per lane, strided order) -> packed bit stream, stored coalesced (encoder.v:96-133)
template <int LEVEL>
__device__ __forceinline__ void store_row_w1_packed(uint8_t* __restrict__ out_row, const uint32_t (&v)[4], uint32_t* scratch, int lane)
{
    uint8_t* sc = reinterpret_cast<uint8_t*>(scratch);
#pragma unroll
    for (int m = 0; m < 4; m++) sc[lane + 64 * m] = (uint8_t)v[m];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (W1Pack<LEVEL>::BITS == 4) {
        if (lane < 32) {                  // 8 coefficients -> one dword
            const uint32_t a = scratch[2 * lane], b = scratch[2 * lane + 1];
            uint32_t x = (a | (a >> 4)) & 0x00FF00FFu;
            x = (x | (x >> 8)) & 0xFFFFu;
            uint32_t y = (b | (b >> 4)) & 0x00FF00FFu;
            y = (y | (y >> 8)) & 0xFFFFu;
            reinterpret_cast<uint32_t*>(out_row)[lane] = x | (y << 16);
        }
    } else {
        if (lane < 16) {                  // 16 coefficients -> 96 bits
            uint32_t w[4];
#pragma unroll
            for (int i = 0; i < 4; i++) w[i] = scratch[4 * lane + i];
            uint64_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint64_t c = (w[k >> 2] >> (8 * (k & 3))) & 0x3Fu;
                const int bit = 6 * k;
                if (bit < 64) lo |= c << bit;
                if (bit + 6 > 64) hi |= (bit >= 64) ? c << (bit - 64) : c >> (64 - bit);
            }
            uint32_t* o = reinterpret_cast<uint32_t*>(out_row) + 3 * lane;
            o[0] = (uint32_t)lo;
            o[1] = (uint32_t)(lo >> 32);
            o[2] = (uint32_t)hi;
        }
    }
}

// Decompose for sign phase 1's outputs: w1 = HighBits(a) and w0 = LowBits(a) AS A RESIDUE in [0, q) (DECOMP, combined_top.v:1946).
// LowBits is a - a1 * 2 gamma2 centred into (-gamma2, gamma2]; its residue is that difference plus q where negative -- the
// centring step of decompose() and the way back cancel (the a1 = 16 | 44 -> 0 wrap leaves a itself, which is its own residue):
// four instructions fewer per coefficient.  Identical to decompose() + canonicalisation over all of [0, q) (checked in
// tests/test_model_and_cabi.py through the oracle).
template <int LEVEL>
__device__ __forceinline__ void decompose_w0res(uint32_t a, uint32_t& a1, uint32_t& w0)
{
    uint32_t t = (a + 127) >> 7;
    if (LEVEL == 2) {
        t = (t * 11275u + (1u << 23)) >> 24;
        t ^= (uint32_t)sgn((int32_t)(43 - t)) & t;
    } else {
        t = (t * 1025u + (1u << 21)) >> 22;
        t &= 15;
    }
    const int32_t r = (int32_t)a - (int32_t)t * (2 * Par<LEVEL>::GAMMA2);
    a1 = t;
    w0 = (uint32_t)(r + (sgn(r) & Q));
}

// Sign phase 1's output stage with ONE transposition (round 4).  The INTT leaves coefficient lane + 64 m in register m; w0
// (int32 row), w1 (byte row) and w1 packed (4 | 6 bits) all want natural order.  Round 3 moved each plane through a byte scratch
// of its own (8 ds_write_b8 -- four lanes per dword, serialised by the LDS -- three reads, and w0 as four 256-byte strided stores):
// 22 % of the shared-key phase-1 kernel (profiles/r04a_ab_mvsabl.txt).  w0 < 2^23 and w1 < 2^6 share a dword: four
// conflict-free ds_write_b32 + one ds_read_b128 give every lane coefficients 4 lane .. 4 lane + 3 of both, w0 leaves as ONE 1-KiB
// dwordx4 store, the w1 bytes as two v_perm_b32 + one 256-byte store, and the packed row is put together across neighbouring lanes
// with one DPP move (encoder.v:96-133: 8 coefficients per dword at 4 bits, 16 per three dwords at 6).
template <int LEVEL>
__device__ __forceinline__ void emit_w1w0_row_t(uint8_t* __restrict__ w1p_row, uint8_t* __restrict__ w1_row, int32_t* __restrict__ w0_row,
                                                const int32_t (&r)[4], uint32_t* xb, int lane)
{
#pragma unroll
    for (int m = 0; m < 4; m++) {
        uint32_t a1, w0;
        decompose_w0res<LEVEL>(canon_small(r[m]), a1, w0);
        xb[lane + 64 * m] = w0 | (a1 << 24);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const uint4 q = *reinterpret_cast<const uint4*>(xb + 4 * lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // bytes 3 of (q.x, q.y, q.z, q.w) -> one dword (v_perm_b32: selector bytes 0-3 pick from the second operand, 4-7 from the first, 0x0c = 0)
    const uint32_t b = __builtin_amdgcn_perm(q.y, q.x, 0x0c0c0703u) | __builtin_amdgcn_perm(q.w, q.z, 0x07030c0cu);
    if (w1_row) {          // the public planes: w0 int32, w1 bytes
        st_nt4(w0_row + 4 * lane, q.x & 0xFFFFFFu, q.y & 0xFFFFFFu, q.z & 0xFFFFFFu, q.w & 0xFFFFFFu);
        reinterpret_cast<uint32_t*>(w1_row)[lane] = b;
    } else {               // inside the signing loop (W0W1 plane, see emit_matvec_row): the transposed dwords leave as they are
        st_nt4(w0_row + 4 * lane, q.x, q.y, q.z, q.w);
    }
    if (w1p_row) {
        if (W1Pack<LEVEL>::BITS == 4) {
            uint32_t n = (b | (b >> 4)) & 0x00FF00FFu;
            n = (n | (n >> 8)) & 0xFFFFu;                                       // this lane's 4 nibbles
            const uint32_t nb = (uint32_t)__builtin_amdgcn_mov_dpp((int)n, 0xB1, 0xF, 0xF, true);      // lane ^ 1's
            if (!(lane & 1)) reinterpret_cast<uint32_t*>(w1p_row)[lane >> 1] = n | (nb << 16);
        } else {
            const uint32_t c = (b & 0x3Fu) | ((b >> 2) & 0xFC0u) | ((b >> 4) & 0x3F000u) | ((b >> 6) & 0xFC0000u);   // 4 x 6 bits
            const uint32_t cn = (uint32_t)__builtin_amdgcn_mov_dpp((int)c, 0xF9, 0xF, 0xF, true);   // quad_perm [1,2,3,3]: the next lane of the quad
            const uint32_t t = (uint32_t)lane & 3u;                              // 4 lanes = 96 bits = 3 dwords, written by lanes t = 0 .. 2
            const uint32_t d = (c >> (8 * t)) | (cn << (24 - 8 * t));
            if (t < 3) reinterpret_cast<uint32_t*>(w1p_row)[3 * (lane >> 2) + t] = d;
        }
    }
}

// Output stage of one mat-vec row: r[] = INTT output (|r| < q, strided order) ->
//   OUT_W   : w row, canonical int32             (matvec)
//   OUT_W1W0: w1 = HighBits as bytes, w0 = LowBits as residue in [0,q)  (sign phase 1, DECOMP :1946); if w_out is not
//             null in this mode it is the PACKED w1 plane ([rows][W1Pack::ROW_BYTES] bytes).
//             w1_out == nullptr: the signing loop's private W0W1 plane -- one dword per coefficient, w0 | w1 << 24 (w0 < 2^23,
//             w1 < 2^6), written where w0 goes: FSM2 of the reference reads w0 / w1 straight from BRAM, never re-encoded
//             (combined_top.v:1946-2229); phase 2 gets both from ONE strided load and no byte plane crosses HBM
template <int LEVEL, int OUT>
__device__ __forceinline__ void emit_matvec_row(int32_t* __restrict__ w_out, uint8_t* __restrict__ w1_out,
                                                int32_t* __restrict__ w0_out, size_t o, const int32_t (&r)[4],
                                                uint32_t* scratch, int lane, uint32_t* xbuf = nullptr)
{
    // xbuf (optional, compile-time null or not after inlining): a 1 KiB per-wave LDS buffer through which the row is turned from
    // the INTT's strided order (lane + 64 m) into row order (4 lane + j): OUT_W leaves as ONE 1-KiB dwordx4 store per wave
    // instead of four 256-byte dword stores, OUT_W1W0 takes the one-transposition stage above
    if (OUT != OUT_W && xbuf) {
        emit_w1w0_row_t<LEVEL>(w_out ? reinterpret_cast<uint8_t*>(w_out) + (o >> 8) * W1Pack<LEVEL>::ROW_BYTES : nullptr,
                               w1_out ? w1_out + o : nullptr, w0_out + o, r, xbuf, lane);
        return;
    }
    uint32_t wb[4], ov[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const uint32_t v = canon_small(r[m]);
        if (OUT == OUT_W) {
            ov[m] = v;
        } else {
            decompose_w0res<LEVEL>(v, wb[m], ov[m]);
        }
    }
    if (OUT != OUT_W && w_out)       // sign phase 1: w1 ALSO leaves packed (the challenge hash's input) -- no pack_w1 launch
        store_row_w1_packed<LEVEL>(reinterpret_cast<uint8_t*>(w_out) + (o >> 8) * W1Pack<LEVEL>::ROW_BYTES, wb, scratch, lane);
    if (OUT != OUT_W && !w1_out) {   // W0W1 plane
#pragma unroll
        for (int m = 0; m < 4; m++) ov[m] |= wb[m] << 24;
    }
    int32_t* dst = (OUT == OUT_W ? w_out : w0_out) + o;
    if (xbuf) {
#pragma unroll
        for (int m = 0; m < 4; m++) xbuf[lane + 64 * m] = ov[m];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const uint4 q = *reinterpret_cast<const uint4*>(xbuf + 4 * lane);
        st_nt4(dst + 4 * lane, q.x, q.y, q.z, q.w);
    } else {
#pragma unroll
        for (int m = 0; m < 4; m++) st_nt(dst + lane + 64 * m, (int32_t)ov[m]);
    }
    if (OUT != OUT_W && w1_out) store_row_u8(w1_out + o, wb, scratch, lane);
}

// strided load of one polynomial (natural order) into NTT-input registers.
// Cache policy, measured with interleaved A/B runs at batch 8192 (profiles/r02_fused_ab.txt): non-temporal is right for the
// mat-vec and sign phase-2 kernels (sign2 level 5: 81 vs 88 us), the DEFAULT policy is right for the verify kernels' z / c /
// t1 loads (level 3 distinct pk: 61.4 vs 66.6 us) -- so the policy is the caller's choice.
template <bool NT = true>
__device__ __forceinline__ void load_strided(int32_t (&r)[4], const int32_t* __restrict__ a, int lane)
{
#pragma unroll
    for (int m = 0; m < 4; m++) r[m] = NT ? ld_nt(a + lane + 64 * m) : a[lane + 64 * m];
}

template <int L, int FMT = A_I32>
struct ARow;
template <int L>
struct ARow<L, A_I32> {
    static constexpr int PD = 256;           // dwords per polynomial in HBM
    int4 v[L];
    // stream = true: this row is read once (per-item A): non-temporal; false: shared A, keep it cached
    __device__ __forceinline__ void load(const int32_t* __restrict__ Arow, int lane, bool stream)
    {
        if (stream) {
#pragma unroll
            for (int l = 0; l < L; l++) v[l] = ld_nt4(Arow + l * 256 + 4 * lane);
        } else {
#pragma unroll
            for (int l = 0; l < L; l++) v[l] = *reinterpret_cast<const int4*>(Arow + l * 256 + 4 * lane);
        }
    }
    __device__ __forceinline__ int4 get(int l) const { return v[l]; }
};
// 24-bit packed coefficients (kernels.hpp A_P24): the lane's 4 coefficients are 12 contiguous bytes = one dwordx3 load,
// unpacked when used (3 VGPRs per polynomial in flight instead of 4)
template <int L>
struct ARow<L, A_P24> {
    static constexpr int PD = 192;
    uint32_t v[L][3];
    __device__ __forceinline__ void load(const int32_t* __restrict__ Arow, int lane, bool stream)
    {
#pragma unroll
        for (int l = 0; l < L; l++) {
            const int32_t* p = Arow + l * 192 + 3 * lane;
            if (stream) {
                v[l][0] = (uint32_t)__builtin_nontemporal_load(p);
                v[l][1] = (uint32_t)__builtin_nontemporal_load(p + 1);
                v[l][2] = (uint32_t)__builtin_nontemporal_load(p + 2);
            } else {
                v[l][0] = (uint32_t)p[0];
                v[l][1] = (uint32_t)p[1];
                v[l][2] = (uint32_t)p[2];
            }
        }
    }
    __device__ __forceinline__ int4 get(int l) const
    {
        const uint32_t a = v[l][0], b = v[l][1], c = v[l][2];
        return make_int4((int32_t)(a & 0xFFFFFFu), (int32_t)(__builtin_amdgcn_alignbit(b, a, 24) & 0xFFFFFFu),
                         (int32_t)(__builtin_amdgcn_alignbit(c, b, 16) & 0xFFFFFFu), (int32_t)(c >> 8));
    }
};

// acc += sum_l A[k][l] o vhat[l] for the lane's 4 coefficients, as 64-bit integers
template <int L, int FMT>
__device__ __forceinline__ void mac_row(int64_t (&acc)[4], const ARow<L, FMT>& A, const uint32_t* vec_lds, int lane)
{
#pragma unroll
    for (int l = 0; l < L; l++) {
        const int4 z = *reinterpret_cast<const int4*>(vec_lds + l * 256 + 4 * lane);
        const int4 a = A.get(l);
        acc[0] += (int64_t)a.x * z.x;
        acc[1] += (int64_t)a.y * z.y;
        acc[2] += (int64_t)a.z * z.z;
        acc[3] += (int64_t)a.w * z.w;
    }
}


}  // namespace dil
