// hash_kernels.hip -- row N1 of SURVEY 8(f): the SHAKE-bound samplers of Dilithium on the GPU, so
// that sign / verify batches no longer bounce through the host between kernels.
//   ExpandA        gen_a_ext.v, sampler_a_ext.v:107,129 (nonce (i<<8)|j), rejection_a.v:67-73
//   ExpandMask     expandmask_ext.v:98,131-185, sampler_y_ext.v:101,119, rejection_y.v:97-99
//   SampleInBall   gen_c.v:163-196,318-339
//   H(mu || w1)    keccak_top (3 instances, combined_top.v:225-231) + encoder.v:96-133 (w1 packing)
// One sponge per lane (keccak.hpp).  Conventions are those the reference's KATs obey (round-3
// v3.1); parity is checked against hashlib and the KAT vectors.
#include "keccak.hpp"
#include "kernels.hpp"
#include "sampler_bodies.hpp"

#include <atomic>
#include <type_traits>

namespace dil {

// Up to this many sponges ExpandMask runs the two-lanes-per-sponge form (1.45 x shorter dependency chain, 1.4 x the
// instructions).  Measured again in round 2 (scripts/bench_two_lane.py, profiles/r02_sign_round.txt): equal at 10-14 k
// sponges, the lane-per-sponge form ahead from 33 k on (level 3, 81920 sponges: 93 vs 101 us) -- 16384 stays.
std::atomic<int> two_lane_max_sponges{16384};

constexpr uint32_t QU = 8380417u;

// ---------------------------------------------------------------------------------------
// generic batched SHAKE256 with one input length for the whole batch:
//   out[i][0..out_bytes) = SHAKE256(in[i][0..in_bytes)),  in_bytes, out_bytes multiples of 8
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(HASH_BS) void shake256_batch_kernel(uint64_t* __restrict__ out, int out_words,
                                                            const uint64_t* __restrict__ in, int in_words, size_t batch)
{
    const size_t i = (size_t)blockIdx.x * HASH_BS + threadIdx.x;
    if (i >= batch) return;
    Shake<17> sp;
    sp.init();
    const int fill = sp.absorb<0>(in + i * (size_t)in_words, in_words);
    sp.finish_words(fill);
    sp.squeeze(out + i * (size_t)out_words, out_words);
}

// ---------------------------------------------------------------------------------------
// ExpandA: A[item][i][j] = RejUniform(SHAKE128(rho || byte j || byte i)), 3-byte little-endian
// candidates masked to 23 bits, accepted when < q.  One lane per polynomial.
// ---------------------------------------------------------------------------------------
template <bool TWO>            // TWO: two lanes per sponge (one or a few keys: the five permutations per polynomial are pure latency)
__global__ __launch_bounds__(HASH_BS) void expand_a_kernel(int32_t* __restrict__ A, const uint64_t* __restrict__ rho,
                                                      size_t rho_stride_words, int K, int L, size_t nitems)
{
    __shared__ uint32_t ring[(HASH_BS / 64) * CoeffSink::LDS_DWORDS_PER_WAVE];
    expand_a_body<TWO>(A, rho, rho_stride_words, K, L, nitems, blockIdx.x, ring);
}

// Key generation with few keys: ExpandA (two lanes per sponge) and ExpandS (a lane per sponge) are both latency-bound and
// independent -- one launch (was: ExpandS on a helper stream, fork / join events).  First `a_blocks` workgroups: A; the others: s1, s2.
template <int ETA>
__global__ __launch_bounds__(HASH_BS) void expand_a_s_kernel(int32_t* __restrict__ A, const uint64_t* __restrict__ rho, size_t rho_stride_words,
                                                             int K, int L, unsigned a_blocks, int32_t* __restrict__ s1, int32_t* __restrict__ s2,
                                                             const uint8_t* __restrict__ rhoprime, size_t rp_stride, size_t nkeys)
{
    __shared__ uint32_t lds[EXPAND_S_LDS_DWORDS];
    static_assert(EXPAND_S_LDS_DWORDS >= (HASH_BS / 64) * CoeffSink::LDS_DWORDS_PER_WAVE, "ExpandA's ring fits");
    if (blockIdx.x < a_blocks) expand_a_body<true>(A, rho, rho_stride_words, K, L, nkeys, blockIdx.x, lds);
    else expand_s_fast_body<ETA>(s1, s2, L, rhoprime, rp_stride, 0, L + K, nkeys, blockIdx.x - a_blocks, lds);
}

template <bool P24>        // P24: A leaves as 24-bit packed coefficients, 768 bytes per polynomial (the internal format of the composite calls)
__global__ __launch_bounds__(HASH_BS) void expand_a_fast_kernel(int32_t* __restrict__ A, const uint64_t* __restrict__ rho,
                                                                size_t rho_stride_words, int K, int L, size_t nitems)
{
    __shared__ uint32_t ring[CoeffSinkWaveT<P24>::LDS_DWORDS_PER_WAVE];
    expand_a_fast_body<P24>(A, rho, rho_stride_words, K, L, nitems, blockIdx.x, ring);
}

template <bool TWO>
__device__ __forceinline__ typename std::conditional<TWO, CoeffSink, CoeffSinkWave>::type make_sink(uint32_t* ring, int32_t* y, size_t p,
                                                                                                      size_t first, size_t total, bool wr);
template <>
__device__ __forceinline__ CoeffSink make_sink<true>(uint32_t* ring, int32_t* y, size_t p, size_t, size_t, bool wr)
{
    return CoeffSink(ring + (threadIdx.x >> 6) * CoeffSink::LDS_DWORDS_PER_WAVE, threadIdx.x & 63, y + p * 256, wr);
}
template <>
__device__ __forceinline__ CoeffSinkWave make_sink<false>(uint32_t* ring, int32_t* y, size_t, size_t first, size_t total, bool)
{
    return CoeffSinkWave(ring, threadIdx.x & 63, y + first * 256, (int)(total - first < 64 ? total - first : 64));
}

// ---------------------------------------------------------------------------------------
// ExpandMask: y[item][l] = gamma1 - Unpack_B(SHAKE256(rho' || LE16(kappa[item] + l))),
// B = 18 (gamma1 = 2^17) or 20 (2^19) bits.  Output canonical in [0, q).  One lane per polynomial.
// ---------------------------------------------------------------------------------------
template <int B, bool TWO>     // TWO: two lanes per sponge (few items: latency-bound)
__global__ __launch_bounds__(HASH_BS) void expand_mask_kernel(int32_t* __restrict__ y, const uint64_t* __restrict__ rhoprime,
                                                         const uint32_t* __restrict__ kappa, int L, size_t nitems)
{
    constexpr int32_t GAMMA1 = 1 << (B - 1);
    constexpr uint64_t MASK = (1ull << B) - 1;
    const size_t t = (size_t)blockIdx.x * HASH_BS + threadIdx.x, total = nitems * (size_t)L;
    size_t p = TWO ? t >> 1 : t;
    if (TWO && p >= total) return;                     // (two-lane: whole pairs leave together)
    if (p >= total) p = total - 1;                     // lane per sponge: lanes past the end run along -- they store for others
    const size_t item = p / (size_t)L;
    const uint32_t nonce = (kappa[item] + (uint32_t)(p % (size_t)L)) & 0xFFFFu;
    LaneSponge<17, TWO> sp;
    sp.init(TWO && (t & 1));
#pragma unroll
    for (int w = 0; w < 8; w++) sp.set(w, rhoprime[item * 8 + w]);
    sp.set(8, (uint64_t)nonce | (0x1Full << 16));
    sp.pad_end();
    const bool wr = sp.writer();
    __shared__ uint32_t ring[(HASH_BS / 64) * CoeffSink::LDS_DWORDS_PER_WAVE];
    // Nothing in this sampler depends on the data (no rejection: every lane is at the same coefficient), so the lane-per-
    // sponge form always takes the wave-synchronous transposed flush: whole 64-byte segments per store instruction
    // instead of a 16-byte piece per lane.
    const size_t first = (size_t)blockIdx.x * HASH_BS;
    constexpr bool LS = TWO;
    typename std::conditional<LS, CoeffSink, CoeffSinkWave>::type sink = make_sink<LS>(ring, y, p, first, total, wr && (TWO || t < total));
    uint64_t buf = 0;
    int nbits = 0, cnt = 0;          // wave-uniform
    while (cnt < 256) {
        sp.permute();
#pragma unroll
        for (int w = 0; w < 17; w++) {
            const uint64_t word = sp.word(w);
            // consume `word` (64 fresh bits) behind the nbits (< B) left in buf
            int avail = 64;
            if (nbits > 0 && cnt < 256) {
                const int need = B - nbits;
                const uint32_t f = (uint32_t)((buf | (word << nbits)) & MASK);
                const int32_t v = GAMMA1 - (int32_t)f;
                if (wr) sink.put(cnt, v + ((v >> 31) & (int32_t)QU));
                cnt++;
                avail -= need;
            }
            uint64_t rest = (avail == 64) ? word : (word >> (64 - avail));
            while (avail >= B && cnt < 256) {
                const int32_t v = GAMMA1 - (int32_t)(rest & MASK);
                if (wr) sink.put(cnt, v + ((v >> 31) & (int32_t)QU));
                cnt++;
                rest >>= B;
                avail -= B;
            }
            buf = rest;
            nbits = avail;
            if (wr) sink.flush_if_ready(cnt);  // <= 4 coefficients per 64-bit word
        }
    }
}

// ExpandMask WITHOUT the unpacking: ExpandMask has no rejection, so the B-bit packed form of y (gamma1 - y, the layout of z on
// the wire: decoder.v:89-143) IS the first 32 B bytes of the SHAKE256 stream (expandmask_ext.v:98, sampler_y_ext.v).  The signing
// loop's large rounds keep y in that form -- 640 (576) bytes per polynomial instead of 1 KiB of int32, no per-coefficient
// extraction here -- and the consumers (sign phase 1 / phase 2, pipelines.hip) unpack in their load stage the way the verify
// kernels read z.  One lane per sponge; each rate block leaves through an LDS transpose so that 17 consecutive lanes write one
// polynomial's contiguous 136 bytes (a lane's own stream would be 8 bytes per instruction at a 640-byte stride).
template <int B>
__global__ __launch_bounds__(HASH_BS) void expand_mask_raw_kernel(uint8_t* __restrict__ yp, const uint64_t* __restrict__ rhoprime,
                                                                  const uint32_t* __restrict__ kappa, int L, size_t nitems)
{
    constexpr int POLYB = 32 * B, QW = POLYB / 8, NBLK = (QW + 16) / 17, LASTW = QW - 17 * (NBLK - 1);   // 80 (72) qwords, 5 blocks, 12 (4) in the last
    static_assert(HASH_BS == 64, "one wave per workgroup");
    const int lane = threadIdx.x;
    const size_t first = (size_t)blockIdx.x * 64, total = nitems * (size_t)L;
    size_t p = first + lane;
    if (p >= total) p = total - 1;                   // lanes past the end run along (they help store) but own nothing
    const int live = (int)(total - first < 64 ? total - first : 64);
    const size_t item = p / (size_t)L;
    const uint32_t nonce = (kappa[item] + (uint32_t)(p % (size_t)L)) & 0xFFFFu;
    Shake<17> sp;
    sp.init();
#pragma unroll
    for (int w = 0; w < 8; w++) sp.s[w] = rhoprime[item * 8 + w];
    sp.s[8] = (uint64_t)nonce | (0x1Full << 16);
    sp.s[16] ^= 0x8000000000000000ull;
    __shared__ uint64_t ring[17 * 65];               // [word][lane], rows padded to 65 qwords: conflict-free both ways
    uint8_t* wave_dst = yp + first * POLYB;
    auto flush = [&](int blk, auto nw_c) {
        constexpr int NW = decltype(nw_c)::value;
#pragma unroll
        for (int w = 0; w < NW; w++) ring[w * 65 + lane] = sp.s[w];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const int i = lane + 64 * j, pl = i / NW, w = i - pl * NW;
            if (pl < live) *reinterpret_cast<uint64_t*>(wave_dst + (size_t)pl * POLYB + 136 * blk + 8 * w) = ring[w * 65 + pl];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
#pragma unroll 1
    for (int blk = 0; blk < NBLK - 1; blk++) {
        keccak_f1600(sp.s);
        flush(blk, std::integral_constant<int, 17>());
    }
    keccak_f1600(sp.s);
    flush(NBLK - 1, std::integral_constant<int, LASTW>());
}

// The raw stream from TWO lanes per sponge, for rounds too narrow to fill the chip with one sponge per lane (a lone wave pays ~5
// cycles per dependent instruction: 8 us per permutation here against 9.1): 32 polynomials per one-wave workgroup, lane 2i the even
// dwords of sponge i's stream, lane 2i + 1 the odd ones; each rate block leaves through an LDS transpose as 34 consecutive dwords
// per polynomial.  The narrow late rounds of the signing loop were 76 us of int32 extraction (expand_mask2_kernel); this is ~45.
template <int B>
__global__ __launch_bounds__(HASH_BS) void expand_mask_raw2_kernel(uint8_t* __restrict__ yp, const uint64_t* __restrict__ rhoprime,
                                                                   const uint32_t* __restrict__ kappa, int L, size_t nitems)
{
    constexpr int POLYB = 32 * B, DW = POLYB / 4, NBLK = (DW + 33) / 34, LASTD = DW - 34 * (NBLK - 1);   // 160 (144) dwords, 5 blocks, 24 (8) in the last
    static_assert(HASH_BS == 64, "one wave per workgroup");
    const int lane = threadIdx.x, col = lane >> 1;
    const bool hi = (lane & 1) != 0;
    const size_t first = (size_t)blockIdx.x * 32, total = nitems * (size_t)L;
    size_t p = first + col;
    if (p >= total) p = total - 1;                   // pairs past the end run along (they help store) but own nothing
    const int live = (int)(total - first < 32 ? total - first : 32);
    const size_t item = p / (size_t)L;
    const uint32_t nonce = (kappa[item] + (uint32_t)(p % (size_t)L)) & 0xFFFFu;
    Shake2<17> sp;
    sp.init(hi);
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const uint64_t v = rhoprime[item * 8 + w];
        sp.s[w] = hi ? (uint32_t)(v >> 32) : (uint32_t)v;
    }
    sp.s[8] = hi ? 0u : (nonce | (0x1Fu << 16));
    sp.s[16] ^= hi ? 0x80000000u : 0u;
    __shared__ uint32_t ring[34 * 33];               // [stream dword][sponge], rows padded to 33 dwords: conflict-free both ways
    uint8_t* wave_dst = yp + first * POLYB;
    auto flush = [&](int blk, auto nd_c) {
        constexpr int ND = decltype(nd_c)::value;    // stream dwords of this block that belong to the polynomial (even)
#pragma unroll
        for (int w = 0; w < ND / 2; w++) ring[(2 * w + (hi ? 1 : 0)) * 33 + col] = sp.s[w];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int j = 0; j < (32 * ND + 63) / 64; j++) {
            const int i = lane + 64 * j, pl = i / ND, d = i - pl * ND;
            if (pl < live) *reinterpret_cast<uint32_t*>(wave_dst + (size_t)pl * POLYB + 136 * blk + 4 * d) = ring[d * 33 + pl];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
#pragma unroll 1
    for (int blk = 0; blk < NBLK - 1; blk++) {
        keccak2_f1600(sp.s, hi);
        flush(blk, std::integral_constant<int, 34>());
    }
    keccak2_f1600(sp.s, hi);
    flush(NBLK - 1, std::integral_constant<int, LASTD>());
}

// Two-lane form of ExpandMask for few entries (the narrow late rounds of the signing loop, single signatures): a lone wave
// pays ~5 cycles per dependent instruction, so the extraction is unrolled at compile time -- nothing in it depends on the data.
// Lane 2i holds the even dwords of sponge i's output stream, lane 2i + 1 the odd ones; after each permutation both lanes fetch
// the partner's 17 dwords (one DPP move each) and see the block as 34 stream dwords; coefficient c is bits [B c, B c + B) of the
// 5-block stream: one v_bfe, or one v_alignbit across two dwords (`carry` = the previous block's last dword).  ~360 instructions
// per block instead of ~1800 in the generic bit-buffer loop of expand_mask_kernel<B, true>.
template <int B>
__global__ __launch_bounds__(HASH_BS) void expand_mask2_kernel(int32_t* __restrict__ y, const uint64_t* __restrict__ rhoprime,
                                                               const uint32_t* __restrict__ kappa, int L, size_t nitems)
{
    constexpr int32_t GAMMA1 = 1 << (B - 1);
    constexpr uint32_t MASK = (1u << B) - 1;
    constexpr int BLOCK_BITS = 17 * 64;
    const size_t t = (size_t)blockIdx.x * HASH_BS + threadIdx.x, total = nitems * (size_t)L;
    const size_t p = t >> 1;
    if (p >= total) return;                            // whole pairs leave together
    const bool hi = (t & 1) != 0;
    const size_t item = p / (size_t)L;
    const uint32_t nonce = (kappa[item] + (uint32_t)(p % (size_t)L)) & 0xFFFFu;
    Shake2<17> sp;
    sp.init(hi);
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const uint64_t v = rhoprime[item * 8 + w];
        sp.s[w] = hi ? (uint32_t)(v >> 32) : (uint32_t)v;
    }
    sp.s[8] = hi ? 0u : (nonce | (0x1Fu << 16));
    sp.s[16] ^= hi ? 0x80000000u : 0u;
    __shared__ uint32_t ring[(HASH_BS / 64) * CoeffSink::LDS_DWORDS_PER_WAVE];
    CoeffSink sink(ring + (threadIdx.x >> 6) * CoeffSink::LDS_DWORDS_PER_WAVE, threadIdx.x & 63, y + p * 256, !hi);
    uint32_t carry = 0;
#pragma unroll
    for (int blk = 0; blk < 5; blk++) {
        keccak2_f1600(sp.s, hi);
        uint32_t D[34];                                // the block as stream dwords (same in both lanes of the pair)
#pragma unroll
        for (int w = 0; w < 17; w++) {
            const uint32_t own = sp.s[w], par = k2_partner(own);
            D[2 * w] = hi ? par : own;
            D[2 * w + 1] = hi ? own : par;
        }
#pragma unroll
        for (int c = 0; c < 256; c++) {
            const int o = B * c, e = o + B;
            if (e > BLOCK_BITS * blk && e <= BLOCK_BITS * (blk + 1)) {          // coefficient c ends in this block (static)
                const int lj = (o >> 5) - 34 * blk, sh = o & 31;                 // local dword of its first bit; -1: previous block
                const uint32_t lo = lj < 0 ? carry : D[lj < 0 ? 0 : lj];
                uint32_t f;
                if (sh + B <= 32) f = (lo >> sh) & MASK;
                else f = __builtin_amdgcn_alignbit(D[lj + 1], lo, sh) & MASK;
                const int32_t v = GAMMA1 - (int32_t)f;
                sink.put(c, v + ((v >> 31) & (int32_t)QU));                      // (the partner lane writes its own, unused column)
                if ((c & (CoeffSink::CHUNK - 1)) == CoeffSink::CHUNK - 1 && !hi) sink.flush_if_ready(c + 1);
            }
        }
        carry = D[33];
    }
}

// ---------------------------------------------------------------------------------------
// SampleInBall: c = tau coefficients +-1 from SHAKE256(c~): 8 sign bytes, then for
// i = 256-tau..255 draw bytes until b <= i; c[i] = c[b]; c[b] = 1 - 2*sign.
// One lane per item; the lane's c[] and its current rate block live in LDS, lane-interleaved.
// Output canonical (+1 -> 1, -1 -> q-1).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sample_in_ball_kernel(int32_t* __restrict__ c_out, const uint64_t* __restrict__ ctilde,
                                                            int tau, size_t nitems)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[SibLds<64>::BYTES];
    sample_in_ball_poly_body<64>(c_out, reinterpret_cast<const uint8_t*>(ctilde), 32, tau, nitems, blockIdx.x, reinterpret_cast<int8_t*>(lds),
                                 reinterpret_cast<uint32_t*>(lds + SibLds<64>::CL_BYTES));
}

// ---------------------------------------------------------------------------------------
// w1 packing (encoder.v:96-133): [rows][256] bytes -> 4 bits (levels 3/5: 128 B per row) or
// 6 bits (level 2: 192 B per row) little-endian bit stream.  16 coefficients per thread.
// ---------------------------------------------------------------------------------------
template <int BITS>
__global__ __launch_bounds__(256) void pack_w1_kernel(uint32_t* __restrict__ out, const uint4* __restrict__ in, size_t nvec16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec16; i += stride) {
        const uint4 v = in[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if (BITS == 4) {
            uint32_t o[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                uint32_t acc = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) acc |= ((w[2 * h + (k >> 2)] >> (8 * (k & 3))) & 0xFu) << (4 * k);
                o[h] = acc;
            }
            reinterpret_cast<uint2*>(out)[i] = make_uint2(o[0], o[1]);
        } else {
            uint64_t lo = 0, hi = 0;     // 96 bits
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint64_t c = (w[k >> 2] >> (8 * (k & 3))) & 0x3Fu;
                const int bit = 6 * k;
                if (bit < 64) lo |= c << bit;
                if (bit + 6 > 64) hi |= (bit >= 64) ? c << (bit - 64) : c >> (64 - bit);
            }
            out[3 * i] = (uint32_t)lo;
            out[3 * i + 1] = (uint32_t)(lo >> 32);
            out[3 * i + 2] = (uint32_t)hi;
        }
    }
}

// ---------------------------------------------------------------------------------------
// c~ = SHAKE256(mu (64 B) || w1_packed (wbytes))  -- the challenge hash of sign and verify.
// expect == nullptr : write the 32-byte digest to out32[i]
// expect != nullptr : verdict[i] |= (digest != expect[i])        (VY_COMPARE, combined_top.v:1501)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(HASH_BS) void challenge_hash_kernel(uint64_t* __restrict__ out32, int32_t* __restrict__ verdict,
                                                            const uint64_t* __restrict__ mu, const uint64_t* __restrict__ w1p,
                                                            int w1_words, const uint8_t* __restrict__ expect, size_t expect_stride,
                                                            size_t batch)
{
    const size_t i = (size_t)blockIdx.x * HASH_BS + threadIdx.x;
    if (i >= batch) return;
    Shake<17> sp;
    sp.init();
#pragma unroll
    for (int t = 0; t < 8; t++) sp.s[t] = mu[i * 8 + t];
    const int fill = sp.absorb<8>(w1p + i * (size_t)w1_words, w1_words);
    sp.finish_words(fill);
    if (expect) {
        uint64_t d = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            uint64_t e;
            __builtin_memcpy(&e, expect + i * expect_stride + 8 * t, 8);      // any alignment
            d |= sp.s[t] ^ e;
        }
        if (d) verdict[i] |= 1;
    } else {
#pragma unroll
        for (int t = 0; t < 4; t++) out32[i * 4 + t] = sp.s[t];
    }
}

// ---------------------------------------------------------------------------------------
// mu[i] = SHAKE256(tr[i] (32 B) || M_i, 64) for RAGGED messages: M_i = msgs[offsets[i] .. + lengths[i]), any alignment.
// What the reference's top level absorbs itself: rtl_src/expandmask_ext.v:131-185, bus order (mlen, tr, m)
// rtl_tb/tb_sign_top.v:57-69, tb_verify_top.v:58-68.  One sponge per lane; every lane walks its own message 8 bytes
// at a time with static state indices (a lane's predicate selects "full word", "last partial word + pad" or "done");
// the wave runs as many permutations as its longest message needs, a lane latches its digest after its own last one.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ld_u64u(const uint8_t* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__global__ __launch_bounds__(HASH_BS) void mu_kernel(uint64_t* __restrict__ mu, const uint8_t* __restrict__ tr, size_t tr_stride,
                                                     const uint8_t* __restrict__ msgs, size_t msgs_bytes, const uint64_t* __restrict__ offsets,
                                                     const uint32_t* __restrict__ lengths, int32_t* __restrict__ bad, size_t batch)
{
    const size_t i = (size_t)blockIdx.x * HASH_BS + threadIdx.x;
    const bool live = i < batch;
    Shake<17> sp;
    sp.init();
    const uint8_t* mp = msgs;
    uint32_t rem = 0;
    if (live) {
        const uint64_t* t = reinterpret_cast<const uint64_t*>(tr + i * tr_stride);
#pragma unroll
        for (int w = 0; w < 4; w++) sp.s[w] = t[w];
        // an item whose (offset, length) leaves the blob is never read: it is hashed as an EMPTY message and flagged
        const uint64_t off = offsets[i];
        const uint32_t len = lengths[i];
        const bool inside = off <= msgs_bytes && len <= msgs_bytes - off;
        if (inside) {
            mp = msgs + off;
            rem = len;
        }
        if (bad) bad[i] = inside ? 0 : 1;
    }
    bool padded = !live, finished = !live;
    uint64_t out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int first = 4;                                   // the first block already holds tr in words 0..3
    while (true) {
#pragma unroll
        for (int w = 0; w < 17; w++) {
            if (w < first || padded) continue;
            if (rem >= 8) {
                sp.s[w] ^= ld_u64u(mp);
                mp += 8;
                rem -= 8;
            } else {                                  // the message ends inside this word: its bytes, then the SHAKE suffix
                uint64_t v = 0;
                for (uint32_t b = 0; b < rem; b++) v |= (uint64_t)mp[b] << (8 * b);
                v |= 0x1Full << (8 * rem);
                sp.s[w] ^= v;
                sp.s[16] ^= 0x8000000000000000ull;
                rem = 0;
                padded = true;
            }
        }
        first = 0;
        keccak_f1600(sp.s);
        if (padded && !finished) {
#pragma unroll
            for (int w = 0; w < 8; w++) out[w] = sp.s[w];
            finished = true;
        }
        if (__all(finished)) break;
    }
    if (live) {
#pragma unroll
        for (int w = 0; w < 8; w++) mu[i * 8 + w] = out[w];
    }
}

hipError_t launch_mu(uint8_t* mu, const uint8_t* tr, size_t tr_stride, const uint8_t* msgs, size_t msgs_bytes, const uint64_t* offsets,
                     const uint32_t* lengths, int32_t* bad, size_t batch, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    if (coop_wanted(batch)) return launch_coop_mu(mu, tr, tr_stride, msgs, msgs_bytes, offsets, lengths, bad, batch, s);
    hipLaunchKernelGGL(mu_kernel, (int)((batch + HASH_BS - 1) / HASH_BS), HASH_BS, 0, s, reinterpret_cast<uint64_t*>(mu), tr, tr_stride, msgs,
                       msgs_bytes, offsets, lengths, bad, batch);
    return hipGetLastError();
}

// verdict[i] = 2 if ||z_i||_inf >= bound else 0 (norm_check.v:84-105 on canonical residues); one wave per item
__global__ __launch_bounds__(256) void z_norm_kernel(int32_t* __restrict__ verdict, const int32_t* __restrict__ z, int npolys,
                                                     uint32_t bound, size_t batch)
{
    const int lane = threadIdx.x & 63;
    const size_t it = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (it >= batch) return;
    bool rej = false;
    const int4* src = reinterpret_cast<const int4*>(z + it * (size_t)npolys * 256);
    for (int k = lane; k < npolys * 64; k += 64) {
        const int4 v = src[k];
        const int32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int32_t x = e[j] % (int32_t)QU;
            x += (x >> 31) & (int32_t)QU;
            rej |= ((uint32_t)x >= bound) && ((uint32_t)x <= QU - bound);
        }
    }
    const bool any = __ballot(rej) != 0;
    if (lane == 0) verdict[it] = any ? 2 : 0;
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
// Two-lane-per-sponge forms of the two long-sponge kernels (keccak.hpp, Shake2): chosen by the launchers when one
// sponge per lane would leave the SIMDs under-occupied.  Thread t = sponge t/2, half t&1; a dead sponge in a live wave
// still runs (both lanes of a pair must execute the DPP exchanges) but neither loads nor stores.
__global__ __launch_bounds__(HASH_BS) void shake256_batch2_kernel(uint32_t* __restrict__ out, int out_words,
                                                             const uint32_t* __restrict__ in, int in_words, size_t batch)
{
    const size_t t = (size_t)blockIdx.x * HASH_BS + threadIdx.x;
    const size_t i = t >> 1;
    if (i >= batch) return;                                   // whole pairs leave together
    Shake2<17> sp;
    sp.init(t & 1);
    const int fill = sp.absorb<0>(in + i * (size_t)in_words * 2, in_words);
    sp.finish_words(fill);
    sp.squeeze(out + i * (size_t)out_words * 2, out_words);
}

__global__ __launch_bounds__(HASH_BS) void challenge_hash2_kernel(uint32_t* __restrict__ out32, int32_t* __restrict__ verdict,
                                                             const uint32_t* __restrict__ mu, const uint32_t* __restrict__ w1p,
                                                             int w1_words, const uint8_t* __restrict__ expect, size_t expect_stride,
                                                             size_t batch)
{
    const size_t t = (size_t)blockIdx.x * HASH_BS + threadIdx.x;
    const size_t i = t >> 1;
    if (i >= batch) return;
    const int hi = (int)(t & 1);
    Shake2<17> sp;
    sp.init(hi);
#pragma unroll
    for (int w = 0; w < 8; w++) sp.s[w] = mu[i * 16 + 2 * w + hi];
    const int fill = sp.absorb<8>(w1p + i * (size_t)w1_words * 2, w1_words);
    sp.finish_words(fill);
    if (expect) {
        uint32_t d = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t e;
            __builtin_memcpy(&e, expect + i * expect_stride + 4 * (2 * w + hi), 4);   // any alignment
            d |= sp.s[w] ^ e;
        }
        if (d) atomicOr(&verdict[i], 1);                      // both halves may flag the same item
    } else {
#pragma unroll
        for (int w = 0; w < 4; w++) out32[i * 8 + 2 * w + hi] = sp.s[w];
    }
}

// ---------------------------------------------------------------------------------------
// The signing loop's challenge as ONE unit, as gen_c.v is one module (absorb mu || w1: gen_c.v:163-196; sample c: :318-339):
// c~ = SHAKE256(mu || w1_packed) is written out (it is a signature field) AND stays in the sponge's registers as the input
// block of SampleInBall's own SHAKE256(c~) -- no second launch, no c~ round trip.  TWO: two lanes per sponge (32 signatures
// per one-wave workgroup) for rounds that would leave the SIMDs under-occupied, else one sponge per lane (64 signatures).
// ---------------------------------------------------------------------------------------
template <bool TWO>
__global__ __launch_bounds__(64) void challenge_sample_kernel(uint32_t* __restrict__ ctilde_out, int32_t* __restrict__ c_out,
                                                              const uint32_t* __restrict__ mu, const uint32_t* __restrict__ w1p, int w1_words,
                                                              int tau, size_t batch)
{
    constexpr int ITEMS = TWO ? 32 : 64;
    __shared__ __attribute__((aligned(16))) uint8_t lds[SibLds<ITEMS>::BYTES];
    int8_t* cl = reinterpret_cast<int8_t*>(lds);
    uint32_t* rb = reinterpret_cast<uint32_t*>(lds + SibLds<ITEMS>::CL_BYTES);
    const int lane = threadIdx.x, col = TWO ? lane >> 1 : lane;
    const size_t base = (size_t)blockIdx.x * ITEMS, item = base + col;
    const bool live = item < batch;
    const size_t ii = live ? item : batch - 1;               // a dead column hashes a valid entry again and stores nothing
    sib_clear(cl, SibLds<ITEMS>::CL_BYTES);
    if (TWO) {
        const int hi = lane & 1;
        Shake2<17> sp;
        sp.init(hi);
#pragma unroll
        for (int w = 0; w < 8; w++) sp.s[w] = mu[ii * 16 + 2 * w + hi];
        const int fill = sp.template absorb<8>(w1p + ii * (size_t)w1_words * 2, w1_words);
        sp.finish_words(fill);
        if (live) {
#pragma unroll
            for (int w = 0; w < 4; w++) ctilde_out[item * 8 + 2 * w + hi] = sp.s[w];
        }
        // SampleInBall's sponge: the 32 digest bytes are its whole message and already sit in words 0..3
#pragma unroll
        for (int w = 4; w < 25; w++) sp.s[w] = 0;
        sp.s[4] = hi ? 0u : 0x1Fu;
        sp.s[16] = hi ? 0x80000000u : 0u;
        keccak2_f1600(sp.s, sp.hi);
        __syncthreads();                                     // cl cleared by all lanes before any lane writes its column
        sib_sample<ITEMS>(SibTwoLane{sp}, col, live ? tau : 0, cl, rb);
    } else {
        const uint64_t* mu64 = reinterpret_cast<const uint64_t*>(mu);
        Shake<17> sp;
        sp.init();
#pragma unroll
        for (int t = 0; t < 8; t++) sp.s[t] = mu64[ii * 8 + t];
        const int fill = sp.template absorb<8>(reinterpret_cast<const uint64_t*>(w1p) + ii * (size_t)w1_words, w1_words);
        sp.finish_words(fill);
        if (live) {
#pragma unroll
            for (int t = 0; t < 4; t++) reinterpret_cast<uint64_t*>(ctilde_out)[item * 4 + t] = sp.s[t];
        }
#pragma unroll
        for (int w = 4; w < 25; w++) sp.s[w] = 0;
        sp.s[4] = 0x1Full;
        sp.s[16] = 0x8000000000000000ull;
        keccak_f1600(sp.s);
        __syncthreads();
        sib_sample<ITEMS>(SibOneLane{sp}, col, live ? tau : 0, cl, rb);
    }
    sib_store_poly<ITEMS>(c_out, batch, base, cl);
}

// one sponge per lane fills the chip from about one wave per SIMD; below that the two-lane form is faster
static inline bool few_sponges(size_t batch) { return batch < 65536; }

hipError_t launch_challenge_hash(uint8_t* out32, int32_t* verdict, const uint8_t* mu, const uint8_t* w1p, int level,
                                 const uint8_t* expect, size_t batch, hipStream_t s, size_t expect_stride)
{
    if (batch == 0) return hipSuccess;
    if (level != 2 && level != 3 && level != 5) return hipErrorInvalidValue;
    const int K = level == 2 ? 4 : level == 3 ? 6 : 8;
    const int words = K * (level == 2 ? 192 : 128) / 8;
    if (coop_wanted(batch)) return launch_coop_challenge_hash(out32, verdict, mu, w1p, words, expect, expect_stride, batch, s);
    if (few_sponges(batch)) {
        hipLaunchKernelGGL(challenge_hash2_kernel, (int)((2 * batch + HASH_BS - 1) / HASH_BS), HASH_BS, 0, s, reinterpret_cast<uint32_t*>(out32),
                           verdict, reinterpret_cast<const uint32_t*>(mu), reinterpret_cast<const uint32_t*>(w1p), words,
                           expect, expect_stride, batch);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(challenge_hash_kernel, (int)((batch + HASH_BS - 1) / HASH_BS), HASH_BS, 0, s, reinterpret_cast<uint64_t*>(out32), verdict,
                       reinterpret_cast<const uint64_t*>(mu), reinterpret_cast<const uint64_t*>(w1p), words,
                       expect, expect_stride, batch);
    return hipGetLastError();
}

hipError_t launch_challenge_sample(uint8_t* ctilde, int32_t* c, const uint8_t* mu, const uint8_t* w1p, int level, size_t batch, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    if (level != 2 && level != 3 && level != 5) return hipErrorInvalidValue;
    const int K = level == 2 ? 4 : level == 3 ? 6 : 8;
    const int words = K * (level == 2 ? 192 : 128) / 8, tau = level == 2 ? 39 : level == 3 ? 49 : 60;
    if (coop_wanted(batch)) return launch_coop_challenge_sample(ctilde, c, mu, w1p, words, tau, batch, s);
    if (few_sponges(batch))
        hipLaunchKernelGGL(challenge_sample_kernel<true>, (int)((batch + 31) / 32), 64, 0, s, reinterpret_cast<uint32_t*>(ctilde), c,
                           reinterpret_cast<const uint32_t*>(mu), reinterpret_cast<const uint32_t*>(w1p), words, tau, batch);
    else
        hipLaunchKernelGGL(challenge_sample_kernel<false>, (int)((batch + 63) / 64), 64, 0, s, reinterpret_cast<uint32_t*>(ctilde), c,
                           reinterpret_cast<const uint32_t*>(mu), reinterpret_cast<const uint32_t*>(w1p), words, tau, batch);
    return hipGetLastError();
}

hipError_t launch_z_norm(int32_t* verdict, const int32_t* z, int level, size_t batch, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    if (level != 2 && level != 3 && level != 5) return hipErrorInvalidValue;
    const int L = level == 2 ? 4 : level == 3 ? 5 : 7;
    const uint32_t bound = (level == 2 ? (1u << 17) : (1u << 19)) - (level == 2 ? 78u : level == 3 ? 196u : 120u);
    hipLaunchKernelGGL(z_norm_kernel, (int)((batch + 3) / 4), 256, 0, s, verdict, z, L, bound, batch);
    return hipGetLastError();
}

hipError_t launch_shake256(uint64_t* out, int out_bytes, const uint64_t* in, int in_bytes, size_t batch, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    if ((out_bytes & 7) || (in_bytes & 7) || out_bytes <= 0 || in_bytes < 0) return hipErrorInvalidValue;
    if (coop_wanted(batch)) return launch_coop_shake256(out, out_bytes, in, in_bytes, batch, s);
    if (few_sponges(batch)) {
        hipLaunchKernelGGL(shake256_batch2_kernel, (int)((2 * batch + HASH_BS - 1) / HASH_BS), HASH_BS, 0, s, reinterpret_cast<uint32_t*>(out),
                           out_bytes / 8, reinterpret_cast<const uint32_t*>(in), in_bytes / 8, batch);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(shake256_batch_kernel, (int)((batch + HASH_BS - 1) / HASH_BS), HASH_BS, 0, s, out, out_bytes / 8, in, in_bytes / 8, batch);
    return hipGetLastError();
}

hipError_t launch_expand_a(int32_t* A, const uint8_t* rho, size_t rho_stride_bytes, int level, size_t nitems, hipStream_t s, int a_fmt)
{
    if (rho_stride_bytes & 7) return hipErrorInvalidValue;
    if (nitems == 0) return hipSuccess;
    const int K = level == 2 ? 4 : level == 3 ? 6 : 8, L = level == 2 ? 4 : level == 3 ? 5 : 7;
    if (level != 2 && level != 3 && level != 5) return hipErrorInvalidValue;
    const size_t total = nitems * (size_t)(K * L);
    if (a_fmt == A_P24) {        // packed output: always the throughput kernel (callers ask for it only on large batches)
        hipLaunchKernelGGL(expand_a_fast_kernel<true>, (int)((total + HASH_BS - 1) / HASH_BS), HASH_BS, 0, s, A,
                           reinterpret_cast<const uint64_t*>(rho), rho_stride_bytes / 8, K, L, nitems);
        return hipGetLastError();
    }
    if (coop_wanted(total)) return launch_coop_expand_a(A, rho, rho_stride_bytes, K, L, nitems, s);      // a sponge per wavefront
    if (total <= EA_TWO_LANE_MAX) {        // latency-bound: two lanes per sponge
        hipLaunchKernelGGL(expand_a_kernel<true>, (int)((2 * total + HASH_BS - 1) / HASH_BS), HASH_BS, 0, s, A,
                           reinterpret_cast<const uint64_t*>(rho), rho_stride_bytes / 8, K, L, nitems);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(expand_a_fast_kernel<false>, (int)((total + HASH_BS - 1) / HASH_BS), HASH_BS, 0, s, A, reinterpret_cast<const uint64_t*>(rho),
                       rho_stride_bytes / 8, K, L, nitems);
    return hipGetLastError();
}

hipError_t launch_expand_a_s(int32_t* A, const uint8_t* rho, size_t rho_stride_bytes, int32_t* s1, int32_t* s2, const uint8_t* rhoprime,
                             size_t rp_stride, int level, int eta, size_t nkeys, hipStream_t s)
{
    if (nkeys == 0) return hipSuccess;
    if (level != 2 && level != 3 && level != 5) return hipErrorInvalidValue;
    if ((rho_stride_bytes & 7) || (reinterpret_cast<uintptr_t>(rho) & 7)) return hipErrorInvalidValue;
    const int K = level == 2 ? 4 : level == 3 ? 6 : 8, L = level == 2 ? 4 : level == 3 ? 5 : 7;
    if (coop_wanted(nkeys * (size_t)(K * L))) return launch_coop_expand_a_s(A, rho, rho_stride_bytes, s1, s2, rhoprime, rp_stride, K, L, eta, nkeys, s);
    const unsigned a_blocks = (unsigned)((2 * nkeys * (size_t)(K * L) + HASH_BS - 1) / HASH_BS);
    const unsigned s_blocks = (unsigned)((nkeys * (size_t)(K + L) + HASH_BS - 1) / HASH_BS);
    if (eta == 2)
        hipLaunchKernelGGL(expand_a_s_kernel<2>, a_blocks + s_blocks, HASH_BS, 0, s, A, reinterpret_cast<const uint64_t*>(rho), rho_stride_bytes / 8,
                           K, L, a_blocks, s1, s2, rhoprime, rp_stride, nkeys);
    else
        hipLaunchKernelGGL(expand_a_s_kernel<4>, a_blocks + s_blocks, HASH_BS, 0, s, A, reinterpret_cast<const uint64_t*>(rho), rho_stride_bytes / 8,
                           K, L, a_blocks, s1, s2, rhoprime, rp_stride, nkeys);
    return hipGetLastError();
}

hipError_t launch_expand_mask(int32_t* y, const uint8_t* rhoprime, const uint32_t* kappa, int level, size_t nitems, hipStream_t s)
{
    if (nitems == 0) return hipSuccess;
    if (level != 2 && level != 3 && level != 5) return hipErrorInvalidValue;
    const int L = level == 2 ? 4 : level == 3 ? 5 : 7;
    const size_t total = nitems * (size_t)L;
    const uint64_t* rp = reinterpret_cast<const uint64_t*>(rhoprime);
    if (coop_wanted(total)) return launch_coop_expand_mask(y, false, rhoprime, kappa, level, nitems, s);
    if (total <= (size_t)two_lane_max_sponges.load(std::memory_order_relaxed)) {        // latency-bound: two lanes per sponge
        const int grid = (int)((2 * total + HASH_BS - 1) / HASH_BS);
        if (level == 2) hipLaunchKernelGGL(expand_mask2_kernel<18>, grid, HASH_BS, 0, s, y, rp, kappa, L, nitems);
        else hipLaunchKernelGGL(expand_mask2_kernel<20>, grid, HASH_BS, 0, s, y, rp, kappa, L, nitems);
        return hipGetLastError();
    }
    const int grid = (int)((total + HASH_BS - 1) / HASH_BS);
    if (level == 2) hipLaunchKernelGGL((expand_mask_kernel<18, false>), grid, HASH_BS, 0, s, y, rp, kappa, L, nitems);
    else hipLaunchKernelGGL((expand_mask_kernel<20, false>), grid, HASH_BS, 0, s, y, rp, kappa, L, nitems);
    return hipGetLastError();
}

hipError_t launch_expand_mask_packed(uint8_t* yp, const uint8_t* rhoprime, const uint32_t* kappa, int level, size_t nitems, hipStream_t s)
{
    if (nitems == 0) return hipSuccess;
    if (level != 2 && level != 3 && level != 5) return hipErrorInvalidValue;
    if (reinterpret_cast<uintptr_t>(yp) & 7) return hipErrorInvalidValue;
    const int L = level == 2 ? 4 : level == 3 ? 5 : 7;
    const size_t total = nitems * (size_t)L;
    const uint64_t* rp = reinterpret_cast<const uint64_t*>(rhoprime);
    if (coop_wanted(total)) return launch_coop_expand_mask(yp, true, rhoprime, kappa, level, nitems, s);
    if (total <= (size_t)two_lane_max_sponges.load(std::memory_order_relaxed)) {        // latency-bound: two lanes per sponge
        const int grid = (int)((total + 31) / 32);
        if (level == 2) hipLaunchKernelGGL(expand_mask_raw2_kernel<18>, grid, HASH_BS, 0, s, yp, rp, kappa, L, nitems);
        else hipLaunchKernelGGL(expand_mask_raw2_kernel<20>, grid, HASH_BS, 0, s, yp, rp, kappa, L, nitems);
        return hipGetLastError();
    }
    const int grid = (int)((total + HASH_BS - 1) / HASH_BS);
    if (level == 2) hipLaunchKernelGGL(expand_mask_raw_kernel<18>, grid, HASH_BS, 0, s, yp, rp, kappa, L, nitems);
    else hipLaunchKernelGGL(expand_mask_raw_kernel<20>, grid, HASH_BS, 0, s, yp, rp, kappa, L, nitems);
    return hipGetLastError();
}

hipError_t launch_sample_in_ball(int32_t* c, const uint8_t* ctilde, int level, size_t nitems, hipStream_t s)
{
    if (nitems == 0) return hipSuccess;
    if (level != 2 && level != 3 && level != 5) return hipErrorInvalidValue;
    const int tau = level == 2 ? 39 : level == 3 ? 49 : 60;
    if (coop_wanted_sib(nitems)) return launch_coop_sample_in_ball(c, nullptr, ctilde, 32, tau, nitems, s);
    hipLaunchKernelGGL(sample_in_ball_kernel, (int)((nitems + 63) / 64), 64, 0, s, c, reinterpret_cast<const uint64_t*>(ctilde), tau, nitems);
    return hipGetLastError();
}

hipError_t launch_pack_w1(uint8_t* out, const uint8_t* w1, int level, size_t nitems, const Tables& t, hipStream_t s)
{
    if (nitems == 0) return hipSuccess;
    if (level != 2 && level != 3 && level != 5) return hipErrorInvalidValue;
    const int K = level == 2 ? 4 : level == 3 ? 6 : 8;
    const size_t nvec = nitems * (size_t)K * 16;
    const size_t blocks = (nvec + 255) / 256;
    const int grid = (int)(blocks < (size_t)t.num_cus * 8 ? blocks : (size_t)t.num_cus * 8);
    if (level == 2) hipLaunchKernelGGL(pack_w1_kernel<6>, grid, 256, 0, s, reinterpret_cast<uint32_t*>(out), reinterpret_cast<const uint4*>(w1), nvec);
    else hipLaunchKernelGGL(pack_w1_kernel<4>, grid, 256, 0, s, reinterpret_cast<uint32_t*>(out), reinterpret_cast<const uint4*>(w1), nvec);
    return hipGetLastError();
}

}  // namespace dil
