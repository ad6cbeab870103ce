// coop_bodies.hpp -- the SHAKE-bound operations of the scheme with ONE sponge per wavefront (keccak_coop.hpp), as device functions
// for a workgroup of one wave: kernels of their own (coop_kernels.hip) and roles inside the composite launches (wire_kernels.hip,
// codec_kernels.hip).  The launchers pick these forms while a call has fewer sponges than about three per SIMD (option coop_max):
// a permutation then takes 2.4 us instead of the 5.9 / 9.7 us of the two-lane / lane-per-sponge forms (profiles/r05d_keccak_coop_asm.txt),
// and the byte stream of a sponge is spread over the lanes, so absorbing is one coalesced load per block and the samplers test a
// whole block of candidates at once (ballot + prefix count) instead of walking it byte by byte.
//   H(mu || w1), SampleInBall   gen_c.v:163-196,318-339          ExpandMask   expandmask_ext.v:98,131-185, rejection_y.v:97-99
//   ExpandA                     gen_a_ext.v, rejection_a.v:67-73   ExpandS      gen_s.v, rejection_s.v
//   mu = H(tr || M)             expandmask_ext.v:131-185
#pragma once
#include "keccak_coop.hpp"

namespace dil {
namespace coop {

constexpr uint32_t Q = 8380417u;

__device__ __forceinline__ int lanes_below(uint64_t mask)       // set bits of `mask` in the lanes below this one
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t ld_u32u(const uint8_t* p)    // any alignment
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// A SHAKE sponge on the wave: v = this lane's dword of the state.  RATE_WORDS = 17 (SHAKE256) or 21 (SHAKE128).
template <int RATE_WORDS>
struct Sponge {
    Lane k;
    uint32_t v;
    bool rate_lane;                                   // this lane holds a dword of the rate part
    __device__ __forceinline__ void init(int lane)
    {
        k.init(lane);
        v = 0;
        rate_lane = k.word >= 0 && k.word < RATE_WORDS;
    }
    __device__ __forceinline__ int dword() const { return k.dword; }            // 0 .. 2 RATE_WORDS - 1 in the rate lanes
    // SHAKE suffix after a message that ends `fill` whole words into the current block (wave-uniform)
    __device__ __forceinline__ void pad(int fill)
    {
        if (k.dword == 2 * fill) v ^= 0x1Fu;
        if (k.dword == 2 * RATE_WORDS - 1) v ^= 0x80000000u;
    }
    __device__ __forceinline__ void permute() { v = coop::permute(v, k); }
    // Absorb a message of n_words 64-bit words (wave-uniform; ld(d) = its dword d) into an EMPTY sponge, pad, permute: the state
    // then holds the first output block.  Block b + 1 is loaded before the permutation that follows block b.
    template <class LD>
    __device__ __forceinline__ void absorb_all(LD ld, int n_words)
    {
        const int nblk = n_words / RATE_WORDS + 1, d0 = k.dword;
        auto fetch = [&](int b) -> uint32_t {
            const int d = 2 * RATE_WORDS * b + d0;
            return (rate_lane && d < 2 * n_words) ? ld(d) : 0u;
        };
        uint32_t nxt = fetch(0);
#pragma unroll 1
        for (int b = 0; b < nblk; b++) {
            const uint32_t cur = nxt;
            if (b + 1 < nblk) nxt = fetch(b + 1);
            v ^= cur;
            if (b == nblk - 1) pad(n_words - b * RATE_WORDS);
            permute();
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// SampleInBall on the wave.  The candidate bytes of a rate block are tested 64 at a time: byte k is taken for i = i_next +
// (accepted bytes before it) iff it is <= that i -- a fixed point reached from "all taken" in as many ballots as there are
// rejected bytes whose verdict the shrinking estimate flips (2-3 in practice; each ballot settles at least the next byte).  The
// accepted bytes are the tokens b_0 .. b_{tau-1}; the serial  c[i] = c[b]; c[b] = +-1  then only decides WHERE each sign ends up:
// token t sits at b_t until a later token t' draws the same position, which moves it to i_t' = 256 - tau + t' (where it may be
// hit again).  Every lane follows its own token through the tau steps (one v_readlane per step), no LDS round trip per step.
// ---------------------------------------------------------------------------------------------------------------------
struct SibShared {
    uint32_t blk[36];          // the rate block as dwords (+ slack: the candidate reads look one dword ahead)
    uint32_t tokb[64];         // accepted bytes in order
    int32_t c[256];
};

// the points where the wave's LDS writes must be visible to its own reads: a workgroup of ONE wave uses the barrier (the stand-alone
// kernels); a wave inside a larger workgroup (verify_wire_wpi_kernel's fused SampleInBall) only orders its own accesses -- a wave's
// LDS instructions execute in order, the fences keep the compiler from moving them
template <bool WAVE_LOCAL>
__device__ __forceinline__ void sib_sync()
{
    if constexpr (WAVE_LOCAL) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// sp: the SHAKE256(c~) sponge after its first squeeze permutation.  On return sh.c holds c (+1 / -1 / 0), visible to the wave.
template <bool WAVE_LOCAL = false>
__device__ __forceinline__ void sib_sample(Sponge<17>& sp, int tau, SibShared& sh, int lane)
{
    const int i0 = 256 - tau;
    int i_next = i0, start = 8;
    uint32_t s_lo = 0, s_hi = 0;
    bool first = true;
    reinterpret_cast<int4*>(sh.c)[lane] = make_int4(0, 0, 0, 0);
    for (;;) {
        if (sp.rate_lane) sh.blk[sp.k.dword] = sp.v;
        sib_sync<WAVE_LOCAL>();
        if (first) {
            s_lo = sh.blk[0];
            s_hi = sh.blk[1];
            first = false;
        }
        for (int base = start; base < 136 && i_next < 256; base += 64) {
            const int pos = base + lane;
            const bool valid = pos < 136;
            const uint32_t b = valid ? (sh.blk[pos >> 2] >> (8 * (pos & 3))) & 255u : 256u;
            uint64_t mask = __ballot(valid);
            for (;;) {
                const int i = i_next + lanes_below(mask);
                const uint64_t m2 = __ballot((int)b <= i && i < 256);
                if (m2 == mask) break;
                mask = m2;
            }
            if ((mask >> lane) & 1) sh.tokb[i_next - i0 + lanes_below(mask)] = b;
            i_next += __popcll(mask);
        }
        if (i_next >= 256) break;
        sib_sync<WAVE_LOCAL>();                       // the block has been read
        sp.permute();
        start = 0;
    }
    sib_sync<WAVE_LOCAL>();
    const uint32_t tb = sh.tokb[lane < tau ? lane : 0];
    uint32_t pos = tb;
#pragma unroll 1
    for (int tp = 1; tp < tau; tp++) {
        const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)tb, tp);
        if (lane < tp) {
            const uint32_t hit = (uint32_t)((int32_t)((pos ^ sb) - 1u) >> 31);      // all-ones iff pos == sb
            pos = bfi(hit, (uint32_t)(i0 + tp), pos);
        }
    }
    if (lane < tau) {
        const uint32_t sign = ((lane < 32 ? s_lo >> lane : s_hi >> (lane - 32)) & 1u);
        sh.c[pos] = sign ? -1 : 1;
    }
    sib_sync<WAVE_LOCAL>();
}

// c~ (32 bytes, any alignment) -> the SampleInBall sponge after its first permutation
__device__ __forceinline__ void sib_seed(Sponge<17>& sp, const uint8_t* ctilde)
{
    const int d = sp.k.dword;
    sp.v = (d >= 0 && d < 8) ? ld_u32u(ctilde + 4 * d) : 0u;
    sp.pad(4);
    sp.permute();
}
// output forms of c: polynomial (canonical int32, 1 KiB) / compact bits (wire_kernels.hip decode_c: bit m = c[lane + 64 m] != 0, bit 4 + m = sign)
__device__ __forceinline__ void sib_store_poly(int32_t* __restrict__ c_out, const SibShared& sh, int lane)
{
    int4 v = reinterpret_cast<const int4*>(sh.c)[lane];
    v.x += (v.x >> 31) & (int32_t)Q;
    v.y += (v.y >> 31) & (int32_t)Q;
    v.z += (v.z >> 31) & (int32_t)Q;
    v.w += (v.w >> 31) & (int32_t)Q;
    reinterpret_cast<int4*>(c_out)[lane] = v;
}
__device__ __forceinline__ void sib_store_bits(uint32_t* __restrict__ cbits, const SibShared& sh, int lane)
{
    uint32_t w = 0;
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int32_t v = sh.c[lane + 64 * m];
        w |= (uint32_t)(v & 1) << m;
        w |= ((uint32_t)v >> 31) << (4 + m);
    }
    cbits[lane] = w;
}

// ---------------------------------------------------------------------------------------------------------------------
// bodies: one sponge = one workgroup of 64 threads
// ---------------------------------------------------------------------------------------------------------------------
// c~ = SHAKE256(mu (64 B) || w1_packed (w1_words words)); the sponge is left holding the digest in dwords 0..7
__device__ __forceinline__ void challenge_absorb(Sponge<17>& sp, const uint32_t* __restrict__ mu, const uint32_t* __restrict__ w1p, int w1_words)
{
    sp.absorb_all([&](int d) { return d < 16 ? mu[d] : w1p[d - 16]; }, 8 + w1_words);
}

// the signing loop's challenge: c~ written out, c = SampleInBall(c~) as a polynomial
__device__ __forceinline__ void challenge_sample_body(uint32_t* __restrict__ ctilde_out, int32_t* __restrict__ c_out, const uint32_t* __restrict__ mu,
                                                      const uint32_t* __restrict__ w1p, int w1_words, int tau, size_t item, SibShared& sh)
{
    const int lane = threadIdx.x;
    Sponge<17> sp;
    sp.init(lane);
    challenge_absorb(sp, mu + item * 16, w1p + item * (size_t)w1_words * 2, w1_words);
    const int d = sp.k.dword;
    const bool dig = d >= 0 && d < 8;
    if (dig) ctilde_out[item * 8 + d] = sp.v;
    sp.v = dig ? sp.v : 0u;                            // SampleInBall's sponge: the digest is its whole message
    sp.pad(4);
    sp.permute();
    sib_sample(sp, tau, sh, lane);
    sib_store_poly(c_out + item * 256, sh, lane);
}

// expect == nullptr: digest -> out32[item];  else verdict[item] |= (digest != expect bytes)
__device__ __forceinline__ void challenge_hash_body(uint32_t* __restrict__ out32, int32_t* __restrict__ verdict, const uint32_t* __restrict__ mu,
                                                    const uint32_t* __restrict__ w1p, int w1_words, const uint8_t* __restrict__ expect,
                                                    size_t expect_stride, size_t item)
{
    Sponge<17> sp;
    sp.init(threadIdx.x);
    challenge_absorb(sp, mu + item * 16, w1p + item * (size_t)w1_words * 2, w1_words);
    const int d = sp.k.dword;
    const bool dig = d >= 0 && d < 8;
    if (expect) {
        const bool diff = dig && sp.v != ld_u32u(expect + item * expect_stride + 4 * d);
        if (__ballot(diff) != 0 && threadIdx.x == 0) verdict[item] |= 1;
    } else if (dig) {
        out32[item * 8 + d] = sp.v;
    }
}

// out = SHAKE256(in), whole 64-bit words
__device__ __forceinline__ void shake256_body(uint32_t* __restrict__ out, int out_words, const uint32_t* __restrict__ in, int in_words)
{
    Sponge<17> sp;
    sp.init(threadIdx.x);
    sp.absorb_all([&](int d) { return in[d]; }, in_words);
#pragma unroll 1
    for (int k0 = 0; k0 < out_words; k0 += 17) {
        if (k0) sp.permute();
        const int g = 2 * k0 + sp.k.dword;
        if (sp.rate_lane && g < 2 * out_words) out[g] = sp.v;
    }
}

// rho' = SHAKE256(key (32 B) || mu (64 B), 64): deterministic signing's seed of the mask (combined_top.v sign set-up, :1694-1790)
__device__ __forceinline__ void rhoprime_body(uint32_t* __restrict__ rp, const uint32_t* __restrict__ key, const uint32_t* __restrict__ mu)
{
    Sponge<17> sp;
    sp.init(threadIdx.x);
    sp.absorb_all([&](int d) { return d < 8 ? key[d] : mu[d - 8]; }, 12);
    const int d = sp.k.dword;
    if (d >= 0 && d < 16) rp[d] = sp.v;
}

// ExpandMask's sponge: SHAKE256(rho' (64 B) || LE16(nonce)), before its first permutation
__device__ __forceinline__ void mask_seed(Sponge<17>& sp, const uint32_t* __restrict__ rhoprime, uint32_t nonce)
{
    const int d = sp.k.dword;
    sp.v = (d >= 0 && d < 16) ? rhoprime[d] : d == 16 ? (nonce | (0x1Fu << 16)) : d == 33 ? 0x80000000u : 0u;
}
// y as the raw B-bit stream (Y_PACKED): 32 B bytes per polynomial, every block one coalesced store
template <int B>
__device__ __forceinline__ void expand_mask_raw_body(uint32_t* __restrict__ yp, const uint32_t* __restrict__ rhoprime, uint32_t nonce)
{
    constexpr int DW = 8 * B;                      // 160 (144) dwords
    Sponge<17> sp;
    sp.init(threadIdx.x);
    mask_seed(sp, rhoprime, nonce);
#pragma unroll 1
    for (int blk = 0; blk < 5; blk++) {
        sp.permute();
        const int g = 34 * blk + sp.k.dword;
        if (sp.rate_lane && g < DW) yp[g] = sp.v;
    }
}
// y as canonical int32: the stream is collected in LDS (stream: 8 B + 2 dwords), every lane then cuts its four coefficients out of it
template <int B>
__device__ __forceinline__ void expand_mask_body(int32_t* __restrict__ y, const uint32_t* __restrict__ rhoprime, uint32_t nonce, uint32_t* stream)
{
    constexpr int DW = 8 * B;
    constexpr int32_t GAMMA1 = 1 << (B - 1);
    constexpr uint32_t MASK = (1u << B) - 1;
    const int lane = threadIdx.x;
    Sponge<17> sp;
    sp.init(lane);
    mask_seed(sp, rhoprime, nonce);
#pragma unroll 1
    for (int blk = 0; blk < 5; blk++) {
        sp.permute();
        const int g = 34 * blk + sp.k.dword;
        if (sp.rate_lane && g < DW) stream[g] = sp.v;
    }
    if (lane == 0) stream[DW] = 0;
    __syncthreads();
    int32_t c[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int o = B * (4 * lane + m), dw = o >> 5;
        const uint32_t f = __builtin_amdgcn_alignbit(stream[dw + 1], stream[dw], (uint32_t)(o & 31)) & MASK;
        const int32_t v = GAMMA1 - (int32_t)f;
        c[m] = v + ((v >> 31) & (int32_t)Q);
    }
    reinterpret_cast<int4*>(y)[lane] = make_int4(c[0], c[1], c[2], c[3]);
}

// ExpandA: one polynomial = RejUniform(SHAKE128(rho || LE16(nonce))), nonce = (i << 8) | j.  The 56 three-byte candidates of a rate
// block are tested by 56 lanes at once.  blk: 44 dwords of LDS.
__device__ __forceinline__ void expand_a_body(int32_t* __restrict__ poly, const uint32_t* __restrict__ rho, uint32_t nonce, uint32_t* blk)
{
    const int lane = threadIdx.x;
    Sponge<21> sp;
    sp.init(lane);
    const int d = sp.k.dword;
    sp.v = (d >= 0 && d < 8) ? rho[d] : d == 8 ? (nonce | (0x1Fu << 16)) : d == 41 ? 0x80000000u : 0u;
    if (lane == 0) blk[42] = 0;
    int cnt = 0;
    while (cnt < 256) {
        sp.permute();
        if (sp.rate_lane) blk[d] = sp.v;
        __syncthreads();
        const int o = 3 * lane, dw = (o >> 2) < 42 ? (o >> 2) : 41;
        const uint32_t cand = __builtin_amdgcn_alignbit(blk[dw + 1], blk[dw], (uint32_t)(8 * (o & 3))) & 0x7FFFFFu;
        const bool acc = lane < 56 && cand < Q;
        const uint64_t mask = __ballot(acc);
        const int at = cnt + lanes_below(mask);
        if (acc && at < 256) poly[at] = (int32_t)cand;
        cnt += __popcll(mask);
        __syncthreads();
    }
}

// ExpandS: one polynomial of s1 / s2 = RejEta(SHAKE256(rho' (64 B, any alignment) || LE16(nonce))): nibbles below 15 (eta 2) / 9 (eta 4),
// 272 per rate block, tested 64 at a time; canonical out.  blk: 36 dwords of LDS.
template <int ETA>
__device__ __forceinline__ void expand_s_body(int32_t* __restrict__ poly, const uint8_t* __restrict__ rhoprime, uint32_t nonce, uint32_t* blk)
{
    constexpr uint32_t LIM = ETA == 2 ? 15 : 9;
    const int lane = threadIdx.x;
    Sponge<17> sp;
    sp.init(lane);
    const int d = sp.k.dword;
    sp.v = (d >= 0 && d < 16) ? ld_u32u(rhoprime + 4 * d) : d == 16 ? (nonce | (0x1Fu << 16)) : d == 33 ? 0x80000000u : 0u;
    int cnt = 0;
    while (cnt < 256) {
        sp.permute();
        if (sp.rate_lane) blk[d] = sp.v;
        __syncthreads();
        for (int base = 0; base < 272 && cnt < 256; base += 64) {
            const int n = base + lane, by = n >> 1;
            const bool valid = n < 272;
            const uint32_t nib = valid ? (blk[by >> 2] >> (8 * (by & 3) + 4 * (n & 1))) & 15u : 15u;
            const bool acc = valid && nib < LIM;
            const uint64_t mask = __ballot(acc);
            const int at = cnt + lanes_below(mask);
            if (acc && at < 256) {
                const int v = ETA == 2 ? 2 - (int)(nib - ((205 * nib) >> 10) * 5) : 4 - (int)nib;
                poly[at] = v + ((v >> 31) & (int32_t)Q);
            }
            cnt += __popcll(mask);
        }
        __syncthreads();
    }
}

// mu = SHAKE256(tr (32 B) || M, 64) for a message of `len` bytes at any alignment: lane's dword of block b is message bytes
// [136 b + 4 dword - 32, +4) (the first 8 dwords of block 0 are tr); the dword in which the message ends carries the SHAKE suffix.
__device__ __forceinline__ void mu_body(uint32_t* __restrict__ mu_out, const uint32_t* __restrict__ tr, const uint8_t* __restrict__ msg, uint32_t len)
{
    Sponge<17> sp;
    sp.init(threadIdx.x);
    const int d0 = sp.k.dword;
    const uint32_t total = 32u + len;                                  // bytes absorbed before the padding
    const int nblk = (int)(total / 136u) + 1;
    auto fetch = [&](int b) -> uint32_t {
        if (!sp.rate_lane) return 0u;
        const uint32_t o = 136u * (uint32_t)b + 4u * (uint32_t)d0;    // byte offset of this lane's dword in tr || M
        uint32_t v = 0;
        if (o + 4 <= 32) v = tr[o >> 2];
        else if (o + 4 <= total) v = ld_u32u(msg + (o - 32));
        else if (o <= total) {                                         // the message ends inside (or right before) this dword
            for (uint32_t j = o; j < total; j++) v |= (uint32_t)msg[j - 32] << (8 * (j - o));
            v |= 0x1Fu << (8 * (total - o));
        }
        return v;
    };
    uint32_t nxt = fetch(0);
#pragma unroll 1
    for (int b = 0; b < nblk; b++) {
        const uint32_t cur = nxt;
        if (b + 1 < nblk) nxt = fetch(b + 1);
        sp.v ^= cur;
        if (b == nblk - 1 && d0 == 33) sp.v ^= 0x80000000u;
        sp.permute();
    }
    if (d0 >= 0 && d0 < 16) mu_out[d0] = sp.v;
}

}  // namespace coop
}  // namespace dil
