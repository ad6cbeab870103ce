// keccak.hpp -- Keccak-f[1600] / SHAKE128 / SHAKE256, one sponge per LANE (64 independent
// sponges per wavefront, the whole 1600-bit state in 50 VGPRs).  Row N1 of SURVEY 8(f): what the
// reference does with three VHDL Keccak cores (keccak_*.vhd, sha3_*.vhd; control word
// {final, mode, outbits, inbits}, keccak_datapath.vhd:97,116-117) feeding its samplers.
// Round = 182 instructions (lane per sponge) / ~120 (two lanes per sponge) thanks to v_bitop3_b32 and v_alignbit_b32.
// Written from FIPS 202; checked against hashlib (tests/test_gpu_hash.py) and, through the
// samplers, against the reference's KAT vectors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dil {

// threads per workgroup of the lane-per-sponge kernels (one wave; 256 measured no better, profiles/r01_keccak_rates.txt)
constexpr int HASH_BS = 64;

__device__ __constant__ uint64_t KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
    0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

// 64-bit rotate by a compile-time amount as two v_alignbit_b32 (the compiler's own lowering is a
// pair of 64-bit shifts + or, which are slow multi-pass ops on CDNA)
// gfx950 has a 3-input bitwise op with an 8-bit truth table (index = a<<2 | b<<1 | c): one instruction for the
// 3-way xors of theta and for chi's  a ^ (~b & c)
__device__ __forceinline__ uint32_t xor3_32(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t chi_32(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xD2); }
__device__ __forceinline__ uint64_t xor3_64(uint64_t a, uint64_t b, uint64_t c)
{
    return ((uint64_t)xor3_32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32)) << 32) |
           xor3_32((uint32_t)a, (uint32_t)b, (uint32_t)c);
}
__device__ __forceinline__ uint64_t chi_64(uint64_t a, uint64_t b, uint64_t c)
{
    return ((uint64_t)chi_32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32)) << 32) |
           chi_32((uint32_t)a, (uint32_t)b, (uint32_t)c);
}

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int n)
{
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (n == 32) return ((uint64_t)lo << 32) | hi;
    uint32_t rl, rh;
    if (n < 32) {
        rh = __builtin_amdgcn_alignbit(hi, lo, 32 - n);
        rl = __builtin_amdgcn_alignbit(lo, hi, 32 - n);
    } else {
        rh = __builtin_amdgcn_alignbit(lo, hi, 64 - n);
        rl = __builtin_amdgcn_alignbit(hi, lo, 64 - n);
    }
    return ((uint64_t)rh << 32) | rl;
}

// 24 rounds; the round body is fully unrolled (static lane indices keep the state in registers),
// the round loop is not (2 KB of code instead of 50 KB)
__device__ __forceinline__ void keccak_f1600(uint64_t (&a)[25])
{
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        const uint64_t c0 = xor3_64(xor3_64(a[0], a[5], a[10]), a[15], a[20]);
        const uint64_t c1 = xor3_64(xor3_64(a[1], a[6], a[11]), a[16], a[21]);
        const uint64_t c2 = xor3_64(xor3_64(a[2], a[7], a[12]), a[17], a[22]);
        const uint64_t c3 = xor3_64(xor3_64(a[3], a[8], a[13]), a[18], a[23]);
        const uint64_t c4 = xor3_64(xor3_64(a[4], a[9], a[14]), a[19], a[24]);
        // theta: a[x][y] ^= c[x-1] ^ rot(c[x+1], 1) as ONE 3-input xor per word
        const uint64_t r0 = rotl64(c1, 1), r1 = rotl64(c2, 1), r2 = rotl64(c3, 1), r3 = rotl64(c4, 1), r4 = rotl64(c0, 1);
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
            a[y] = xor3_64(a[y], c4, r0);
            a[y + 1] = xor3_64(a[y + 1], c0, r1);
            a[y + 2] = xor3_64(a[y + 2], c1, r2);
            a[y + 3] = xor3_64(a[y + 3], c2, r3);
            a[y + 4] = xor3_64(a[y + 4], c3, r4);
        }
        // rho + pi
        uint64_t b[25];
        b[0] = a[0];
        b[10] = rotl64(a[1], 1);   b[20] = rotl64(a[2], 62);  b[5] = rotl64(a[3], 28);   b[15] = rotl64(a[4], 27);
        b[16] = rotl64(a[5], 36);  b[1] = rotl64(a[6], 44);   b[11] = rotl64(a[7], 6);   b[21] = rotl64(a[8], 55);
        b[6] = rotl64(a[9], 20);   b[7] = rotl64(a[10], 3);   b[17] = rotl64(a[11], 10); b[2] = rotl64(a[12], 43);
        b[12] = rotl64(a[13], 25); b[22] = rotl64(a[14], 39); b[23] = rotl64(a[15], 41); b[8] = rotl64(a[16], 45);
        b[18] = rotl64(a[17], 15); b[3] = rotl64(a[18], 21);  b[13] = rotl64(a[19], 8);  b[14] = rotl64(a[20], 18);
        b[24] = rotl64(a[21], 2);  b[9] = rotl64(a[22], 61);  b[19] = rotl64(a[23], 56); b[4] = rotl64(a[24], 14);
        // chi
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
            a[y] = chi_64(b[y], b[y + 1], b[y + 2]);
            a[y + 1] = chi_64(b[y + 1], b[y + 2], b[y + 3]);
            a[y + 2] = chi_64(b[y + 2], b[y + 3], b[y + 4]);
            a[y + 3] = chi_64(b[y + 3], b[y + 4], b[y]);
            a[y + 4] = chi_64(b[y + 4], b[y], b[y + 1]);
        }
        a[0] ^= KECCAK_RC[round];
    }
}

// A sponge whose input is short and known up front: absorb `nbytes` (<= rate - 1) bytes given as
// little-endian 64-bit words (zero padded), pad with the SHAKE suffix 0x1F ... 0x80, permute.
// RATE_WORDS = 21 (SHAKE128, 168 B) or 17 (SHAKE256, 136 B).
template <int RATE_WORDS>
struct Shake {
    uint64_t s[25];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int i = 0; i < 25; i++) s[i] = 0;
    }
    // xor one input word into the state (caller guarantees word < RATE_WORDS)
    __device__ __forceinline__ void absorb_word(int word, uint64_t v) { s[word] ^= v; }
    // finish a message that ends after `total_bytes` bytes in the CURRENT block
    __device__ __forceinline__ void finish(int bytes_in_block)
    {
        xor_byte(bytes_in_block, 0x1F);
        s[RATE_WORDS - 1] ^= 0x8000000000000000ull;
        keccak_f1600(s);
    }
    __device__ __forceinline__ void xor_byte(int pos, uint64_t v)
    {
        // pos is uniform/compile-time at every call site; the switch folds away
#pragma unroll
        for (int w = 0; w < RATE_WORDS; w++)
            if (w == (pos >> 3)) s[w] ^= v << (8 * (pos & 7));
    }
    __device__ __forceinline__ void next_block() { keccak_f1600(s); }

    // Absorb `n` 64-bit words from src into a sponge whose current block already holds W0 words
    // (compile time), permuting after each full block; returns the fill of the last, partial block.
    // Whole blocks are absorbed with static state indices and all RATE_WORDS loads in flight at once
    // (n is wave-uniform, so the partial tail is a run of scalar-predicated static accesses too).
    // Software-pipelined: the words of block b + 1 (or of the partial tail) are LOADED before the permutation that follows
    // block b and xored in after it -- a lane's message is a strided stream of its own (64 cache lines per load
    // instruction), and with the loads at their first use every block of a long sponge (H(mu || w1): 7 blocks, tr = H(pk):
    // 15) paid a memory round trip in front of its permutation.
    template <int W0>
    __device__ __forceinline__ int absorb(const uint64_t* __restrict__ src, int n)
    {
        int k = 0;
        bool pend = false;                           // a full block is absorbed and waits for its permutation
        if (W0 != 0) {
            if (n < RATE_WORDS - W0) {
#pragma unroll
                for (int t = W0; t < RATE_WORDS; t++)
                    if (t - W0 < n) s[t] ^= src[t - W0];
                return W0 + n;
            }
#pragma unroll
            for (int t = W0; t < RATE_WORDS; t++) s[t] ^= src[t - W0];
            k = RATE_WORDS - W0;
            pend = true;
        }
#pragma unroll 1
        for (;;) {
            const int left = n - k;                  // wave-uniform
            const bool full = left >= RATE_WORDS;
            uint64_t v[RATE_WORDS];
#pragma unroll
            for (int t = 0; t < RATE_WORDS; t++) v[t] = (full || t < left) ? src[k + t] : 0;
            __builtin_amdgcn_sched_barrier(0);
            if (pend) keccak_f1600(s);
#pragma unroll
            for (int t = 0; t < RATE_WORDS; t++) s[t] ^= v[t];
            if (!full) return left;
            k += RATE_WORDS;
            pend = true;
        }
    }
    // pad (SHAKE suffix) after a message that left `fill` whole words in the current block, permute
    __device__ __forceinline__ void finish_words(int fill)
    {
#pragma unroll
        for (int t = 0; t < RATE_WORDS; t++)
            if (t == fill) s[t] ^= 0x1Full;
        s[RATE_WORDS - 1] ^= 0x8000000000000000ull;
        keccak_f1600(s);
    }
    // squeeze n words to dst (n wave-uniform)
    __device__ __forceinline__ void squeeze(uint64_t* __restrict__ dst, int n)
    {
#pragma unroll 1
        for (int k = 0; k < n; k += RATE_WORDS) {
            if (k) keccak_f1600(s);
#pragma unroll
            for (int t = 0; t < RATE_WORDS; t++)
                if (k + t < n) dst[k + t] = s[t];
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Two lanes per sponge, for latency-bound hashing (few, long sponges: H(mu || w1), tr = H(pk), small batches).
// Lane 2i holds the LOW 32 bits of the 25 state words of sponge i, lane 2i+1 the HIGH 32 bits.  theta's parities,
// chi and iota are lane-local 32-bit work; a 64-bit rotation by n is, in both lanes,
//     own' = alignbit(own, partner, 32 - n)   (n < 32)      own' = alignbit(partner, own, 64 - n)   (n > 32)
// with the partner's half fetched by one DPP quad_perm[1,0,3,2] move.  160 instructions per round per lane instead of
// 270, i.e. 1.7x shorter latency per permutation at 1.2x the issue slots per sponge -- so it is used only where one
// sponge per lane leaves the SIMDs under-occupied.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t k2_partner(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
}
template <int N>
__device__ __forceinline__ uint32_t k2_rot(uint32_t own)
{
    static_assert(N > 0 && N < 64 && N != 32, "rotation amount");
    const uint32_t par = k2_partner(own);
    return N < 32 ? __builtin_amdgcn_alignbit(own, par, 32 - N) : __builtin_amdgcn_alignbit(par, own, 64 - N);
}

__device__ __forceinline__ void keccak2_f1600(uint32_t (&a)[25], bool hi)
{
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        const uint32_t c0 = xor3_32(xor3_32(a[0], a[5], a[10]), a[15], a[20]);
        const uint32_t c1 = xor3_32(xor3_32(a[1], a[6], a[11]), a[16], a[21]);
        const uint32_t c2 = xor3_32(xor3_32(a[2], a[7], a[12]), a[17], a[22]);
        const uint32_t c3 = xor3_32(xor3_32(a[3], a[8], a[13]), a[18], a[23]);
        const uint32_t c4 = xor3_32(xor3_32(a[4], a[9], a[14]), a[19], a[24]);
        const uint32_t r0 = k2_rot<1>(c1), r1 = k2_rot<1>(c2), r2 = k2_rot<1>(c3), r3 = k2_rot<1>(c4), r4 = k2_rot<1>(c0);
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
            a[y] = xor3_32(a[y], c4, r0);
            a[y + 1] = xor3_32(a[y + 1], c0, r1);
            a[y + 2] = xor3_32(a[y + 2], c1, r2);
            a[y + 3] = xor3_32(a[y + 3], c2, r3);
            a[y + 4] = xor3_32(a[y + 4], c3, r4);
        }
        uint32_t b[25];
        b[0] = a[0];
        b[10] = k2_rot<1>(a[1]);   b[20] = k2_rot<62>(a[2]);  b[5] = k2_rot<28>(a[3]);   b[15] = k2_rot<27>(a[4]);
        b[16] = k2_rot<36>(a[5]);  b[1] = k2_rot<44>(a[6]);   b[11] = k2_rot<6>(a[7]);   b[21] = k2_rot<55>(a[8]);
        b[6] = k2_rot<20>(a[9]);   b[7] = k2_rot<3>(a[10]);   b[17] = k2_rot<10>(a[11]); b[2] = k2_rot<43>(a[12]);
        b[12] = k2_rot<25>(a[13]); b[22] = k2_rot<39>(a[14]); b[23] = k2_rot<41>(a[15]); b[8] = k2_rot<45>(a[16]);
        b[18] = k2_rot<15>(a[17]); b[3] = k2_rot<21>(a[18]);  b[13] = k2_rot<8>(a[19]);  b[14] = k2_rot<18>(a[20]);
        b[24] = k2_rot<2>(a[21]);  b[9] = k2_rot<61>(a[22]);  b[19] = k2_rot<56>(a[23]); b[4] = k2_rot<14>(a[24]);
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
            a[y] = chi_32(b[y], b[y + 1], b[y + 2]);
            a[y + 1] = chi_32(b[y + 1], b[y + 2], b[y + 3]);
            a[y + 2] = chi_32(b[y + 2], b[y + 3], b[y + 4]);
            a[y + 3] = chi_32(b[y + 3], b[y + 4], b[y]);
            a[y + 4] = chi_32(b[y + 4], b[y], b[y + 1]);
        }
        const uint64_t rc = KECCAK_RC[round];
        a[0] ^= hi ? (uint32_t)(rc >> 32) : (uint32_t)rc;
    }
}

// The two-lane sponge: same interface as Shake<>, message/digest addressed as 32-bit halves (src32 = the 64-bit-word
// stream viewed as dwords; this lane takes dword 2w + hi of word w).
template <int RATE_WORDS>
struct Shake2 {
    uint32_t s[25];
    bool hi;
    __device__ __forceinline__ void init(bool high_half)
    {
        hi = high_half;
#pragma unroll
        for (int i = 0; i < 25; i++) s[i] = 0;
    }
    template <int W0>          // software-pipelined as Shake::absorb
    __device__ __forceinline__ int absorb(const uint32_t* __restrict__ src32, int n)
    {
        const uint32_t* src = src32 + (hi ? 1 : 0);
        int k = 0;
        bool pend = false;
        if (W0 != 0) {
            if (n < RATE_WORDS - W0) {
#pragma unroll
                for (int t = W0; t < RATE_WORDS; t++)
                    if (t - W0 < n) s[t] ^= src[2 * (t - W0)];
                return W0 + n;
            }
#pragma unroll
            for (int t = W0; t < RATE_WORDS; t++) s[t] ^= src[2 * (t - W0)];
            k = RATE_WORDS - W0;
            pend = true;
        }
#pragma unroll 1
        for (;;) {
            const int left = n - k;
            const bool full = left >= RATE_WORDS;
            uint32_t v[RATE_WORDS];
#pragma unroll
            for (int t = 0; t < RATE_WORDS; t++) v[t] = (full || t < left) ? src[2 * (k + t)] : 0u;
            __builtin_amdgcn_sched_barrier(0);
            if (pend) keccak2_f1600(s, hi);
#pragma unroll
            for (int t = 0; t < RATE_WORDS; t++) s[t] ^= v[t];
            if (!full) return left;
            k += RATE_WORDS;
            pend = true;
        }
    }
    __device__ __forceinline__ void finish_words(int fill)
    {
#pragma unroll
        for (int t = 0; t < RATE_WORDS; t++)
            if (t == fill) s[t] ^= hi ? 0u : 0x1Fu;
        s[RATE_WORDS - 1] ^= hi ? 0x80000000u : 0u;
        keccak2_f1600(s, hi);
    }
    __device__ __forceinline__ void squeeze(uint32_t* __restrict__ dst32, int n)
    {
        uint32_t* dst = dst32 + (hi ? 1 : 0);
#pragma unroll 1
        for (int k = 0; k < n; k += RATE_WORDS) {
            if (k) keccak2_f1600(s, hi);
#pragma unroll
            for (int t = 0; t < RATE_WORDS; t++)
                if (k + t < n) dst[2 * (k + t)] = s[t];
        }
    }
};

// One interface over both sponge forms for the samplers: state words are set and read as 64-bit values; in the two-lane
// form each lane keeps its half and word() rebuilds the 64-bit value with one DPP exchange (both lanes of a pair then
// run the same sampling logic and agree on every counter; only writer() lanes store).
template <int RATE_WORDS, bool TWO>
struct LaneSponge;
template <int RATE_WORDS>
struct LaneSponge<RATE_WORDS, false> {
    Shake<RATE_WORDS> sp;
    __device__ __forceinline__ void init(bool) { sp.init(); }
    __device__ __forceinline__ bool writer() const { return true; }
    __device__ __forceinline__ void set(int w, uint64_t v) { sp.s[w] = v; }
    __device__ __forceinline__ void pad_end() { sp.s[RATE_WORDS - 1] ^= 0x8000000000000000ull; }
    __device__ __forceinline__ void permute() { keccak_f1600(sp.s); }
    __device__ __forceinline__ uint64_t word(int w) const { return sp.s[w]; }
};
template <int RATE_WORDS>
struct LaneSponge<RATE_WORDS, true> {
    Shake2<RATE_WORDS> sp;
    __device__ __forceinline__ void init(bool hi) { sp.init(hi); }
    __device__ __forceinline__ bool writer() const { return !sp.hi; }
    __device__ __forceinline__ void set(int w, uint64_t v) { sp.s[w] = sp.hi ? (uint32_t)(v >> 32) : (uint32_t)v; }
    __device__ __forceinline__ void pad_end() { sp.s[RATE_WORDS - 1] ^= sp.hi ? 0x80000000u : 0u; }
    __device__ __forceinline__ void permute() { keccak2_f1600(sp.s, sp.hi); }
    __device__ __forceinline__ uint64_t word(int w) const
    {
        const uint32_t own = sp.s[w], par = k2_partner(own);
        return sp.hi ? (((uint64_t)own << 32) | par) : (((uint64_t)par << 32) | own);
    }
};

// Output staging of the lane-per-sponge samplers.  Each lane emits the coefficients of its own polynomial one at a
// time; written straight to global memory that is a 4-byte store per lane per coefficient, 64 different cache lines
// per store instruction (measured: a third of ExpandA's time).  Instead each lane keeps a ring of 32 coefficients in
// LDS ([slot][lane]: conflict-free) and flushes 16 at a time as four 16-byte stores to its 64-byte-aligned segment.
struct CoeffSink {
    static constexpr int RING = 32, CHUNK = 16;
    static constexpr int LDS_DWORDS_PER_WAVE = RING * 64;
    uint32_t* ring;      // this wave's staging area, already offset by the lane
    int32_t* dst;        // this lane's polynomial (1 KiB aligned)
    int flushed;
    // a lane without a polynomial (past the end of the batch) starts "all flushed" and never stores
    __device__ __forceinline__ CoeffSink(uint32_t* wave_ring, int lane, int32_t* poly, bool live = true)
        : ring(wave_ring + lane), dst(poly), flushed(live ? 0 : 256) {}
    __device__ __forceinline__ void put(int cnt, int32_t v) { ring[(cnt & (RING - 1)) * 64] = (uint32_t)v; }
    // call at least every RING - CHUNK emitted coefficients
    __device__ __forceinline__ void flush_if_ready(int cnt)
    {
        if (cnt - flushed >= CHUNK) {
            const int base = flushed & (RING - 1);
#pragma unroll
            for (int q = 0; q < CHUNK / 4; q++) {
                int4 v;
                v.x = (int32_t)ring[(base + 4 * q) * 64];
                v.y = (int32_t)ring[(base + 4 * q + 1) * 64];
                v.z = (int32_t)ring[(base + 4 * q + 2) * 64];
                v.w = (int32_t)ring[(base + 4 * q + 3) * 64];
                *reinterpret_cast<int4*>(dst + flushed + 4 * q) = v;
            }
            flushed += CHUNK;
        }
    }
};

// Wave-synchronous variant of the staging above for the one-wave-per-workgroup samplers whose 64 lanes advance almost in
// step (ExpandA: a lane falls behind only through 0.1 % rejections).  When EVERY lane of the wave has 16 unflushed
// coefficients the wave writes that chunk for all 64 polynomials TRANSPOSED: lane l stores the 16-byte piece l & 3 of
// polynomial 16 j + (l >> 2), j = 0..3, so each store instruction writes 16 complete, contiguous 64-byte segments instead
// of 64 separate 16-byte pieces of 64 different cache lines (measured: the per-lane stores cost ExpandA 70 of its 190 us,
// profiles/r02_expand_a.txt).  Lanes may run ahead of the slowest by up to RING - CHUNK - 8 coefficients; if one ever gets
// further (probability ~ 0 for hashed seeds, but it must stay correct) the wave drops to the per-lane flush for good.
template <bool P24>      // P24: polynomials leave as 24-bit packed coefficients (768 bytes each) instead of int32 (1 KiB)
struct CoeffSinkWaveT {
    static constexpr int POLY_DW = P24 ? 192 : 256;
    // store coefficients [at, at + 4) of the polynomial starting at dword pointer `poly`
    __device__ __forceinline__ static void put4(int32_t* poly, int at, int4 v)
    {
        if (P24) {
            uint32_t* d = reinterpret_cast<uint32_t*>(poly) + (at >> 2) * 3;
            const uint32_t a = (uint32_t)v.x, b = (uint32_t)v.y, c = (uint32_t)v.z, e = (uint32_t)v.w;
            d[0] = a | (b << 24);
            d[1] = (b >> 8) | (c << 16);
            d[2] = (c >> 16) | (e << 8);
        } else {
            *reinterpret_cast<int4*>(poly + at) = v;
        }
    }
    // 16 coefficients per polynomial and flush: one 64-byte segment (int32) or a 48-byte piece (24-bit).  The 48-byte pieces
    // straddle the memory system's 32-byte sectors (PMC: 263 MB written for 188 MB of payload); flushing 32 coefficients =
    // 96 bytes = three whole sectors from a ring twice as deep was tried and is SLOWER (keygen level 3, 8192 keys: 473 vs
    // 426 us -- the deeper ring halves the resident waves), so 16 it stays.
    static constexpr int CHUNK = 16, RING = 2 * CHUNK;
    static constexpr int LPP = CHUNK / 4, PPI = 64 / LPP;        // lanes per polynomial, polynomials per store instruction
    static constexpr int LDS_DWORDS_PER_WAVE = RING * 64;
    uint32_t* wave_ring;      // [slot][lane]
    uint32_t* ring;           // wave_ring + lane
    int32_t* wave_dst;        // polynomial of lane 0 (1 KiB per lane, consecutive)
    int lane;
    int live_polys;           // polynomials of this wave that exist (lanes >= live_polys have none)
    int flushed;              // wave-uniform while `uniform`
    bool uniform;
    __device__ __forceinline__ CoeffSinkWaveT(uint32_t* wr, int ln, int32_t* dst0, int nlive)
        : wave_ring(wr), ring(wr + ln), wave_dst(dst0), lane(ln), live_polys(nlive), flushed(0), uniform(true) {}
    __device__ __forceinline__ void put(int cnt, int32_t v) { ring[(cnt & (RING - 1)) * 64] = (uint32_t)v; }
    __device__ __forceinline__ void flush_if_ready(int cnt)
    {
        if (uniform) {
            if (__all(cnt - flushed >= CHUNK)) {
                const int base = flushed & (RING - 1), q = lane % LPP;
#pragma unroll
                for (int j = 0; j < 64 / PPI; j++) {
                    const int p = PPI * j + lane / LPP;
                    int4 v;
                    v.x = (int32_t)wave_ring[(base + 4 * q) * 64 + p];
                    v.y = (int32_t)wave_ring[(base + 4 * q + 1) * 64 + p];
                    v.z = (int32_t)wave_ring[(base + 4 * q + 2) * 64 + p];
                    v.w = (int32_t)wave_ring[(base + 4 * q + 3) * 64 + p];
                    if (p < live_polys) put4(wave_dst + p * POLY_DW, flushed + 4 * q, v);
                }
                flushed += CHUNK;
            } else if (__any(cnt - flushed > RING - 8)) {
                uniform = false;                  // a lane is about to lap the ring: per-lane flushing from here on
            }
        }
        if (!uniform && lane < live_polys && cnt - flushed >= CHUNK) {
            const int base = flushed & (RING - 1);
#pragma unroll
            for (int q = 0; q < CHUNK / 4; q++) {
                int4 v;
                v.x = (int32_t)ring[(base + 4 * q) * 64];
                v.y = (int32_t)ring[(base + 4 * q + 1) * 64];
                v.z = (int32_t)ring[(base + 4 * q + 2) * 64];
                v.w = (int32_t)ring[(base + 4 * q + 3) * 64];
                put4(wave_dst + lane * POLY_DW, flushed + 4 * q, v);
            }
            flushed += CHUNK;
        }
    }
};
using CoeffSinkWave = CoeffSinkWaveT<false>;

}  // namespace dil
