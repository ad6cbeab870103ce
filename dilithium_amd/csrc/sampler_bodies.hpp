// sampler_bodies.hpp -- the two latency-bound samplers of a verification as device functions, so that they can run as
// kernels of their own (hash_kernels.hip: expand_a_kernel<TWO>; wire_kernels.hip: sample_in_ball_bits_kernel) and side by
// side in one launch (wire_kernels.hip: expand_a_sib_kernel).  gen_a_ext.v + rejection_a.v:67-73; gen_c.v:163-196,318-339.
#pragma once
#include "keccak.hpp"
#include "modarith.hpp"

namespace dil {

constexpr uint32_t QU_BODY = 8380417u;

// every lane counts, only `writer` lanes store (two-lane sponges: both lanes of a pair run this with the same words)
__device__ __forceinline__ void emit23(uint32_t v, CoeffSink& sink, int& cnt, bool writer)
{
    // branch-free (a lone wave pays ~5 cycles per instruction, and a divergent branch per candidate costs a dozen): the candidate is
    // written to the lane's ring column unconditionally -- a rejected one, or one after the 256th, is overwritten or never flushed --
    // and the count advances by the accept mask.  (`writer` lanes are the ones that flush; the partner lane's column is its own.)
    (void)writer;
    v &= 0x7FFFFFu;
    sink.put(cnt, (int32_t)v);
    const int acc = (int)((v - QU_BODY) >> 31) & (int)((uint32_t)(cnt - 256) >> 31);      // v < q and cnt < 256
    cnt += acc;
}

// body of expand_a_kernel<TWO> (hash_kernels.hip) for workgroup `block`; `ring`: CoeffSink::LDS_DWORDS_PER_WAVE dwords of LDS
template <bool TWO>            // TWO: two lanes per sponge (one or a few keys: the five permutations per polynomial are pure latency)
__device__ __forceinline__ void expand_a_body(int32_t* __restrict__ A, const uint64_t* __restrict__ rho, size_t rho_stride_words, int K,
                                              int L, size_t nitems, unsigned block, uint32_t* ring)
{
    const size_t t = (size_t)block * HASH_BS + threadIdx.x;
    const size_t p = TWO ? t >> 1 : t;
    const size_t total = nitems * (size_t)(K * L);
    const bool live = p < total;                       // (two-lane: whole pairs are live or dead together)
    const size_t item = live ? p / (size_t)(K * L) : 0;
    const int ij = (int)(p % (size_t)(K * L)), i = ij / L, j = ij % L;
    LaneSponge<21, TWO> sp;
    sp.init(TWO && (t & 1));
#pragma unroll
    for (int w = 0; w < 4; w++) sp.set(w, rho[item * rho_stride_words + w]);
    sp.set(4, (uint64_t)j | ((uint64_t)i << 8) | (0x1Full << 16));
    sp.pad_end();
    const bool wr = live && sp.writer();
    CoeffSink sink(ring + (threadIdx.x >> 6) * CoeffSink::LDS_DWORDS_PER_WAVE, threadIdx.x & 63, A + p * 256, wr);
    int cnt = live ? 0 : 256;
    while (__any(cnt < 256)) {
        sp.permute();
#pragma unroll
        for (int g = 0; g < 7; g++) {
            const uint64_t w0 = sp.word(3 * g), w1 = sp.word(3 * g + 1), w2 = sp.word(3 * g + 2);
            emit23((uint32_t)w0, sink, cnt, wr);
            emit23((uint32_t)(w0 >> 24), sink, cnt, wr);
            emit23((uint32_t)((w0 >> 48) | (w1 << 16)), sink, cnt, wr);
            emit23((uint32_t)(w1 >> 8), sink, cnt, wr);
            emit23((uint32_t)(w1 >> 32), sink, cnt, wr);
            emit23((uint32_t)((w1 >> 56) | (w2 << 8)), sink, cnt, wr);
            emit23((uint32_t)(w2 >> 16), sink, cnt, wr);
            emit23((uint32_t)(w2 >> 40), sink, cnt, wr);
            if (wr) sink.flush_if_ready(cnt);
        }
    }
}


// Lane-per-sponge ExpandA, branch-free candidate handling (the throughput form; the kernel above stays for the latency-
// bound two-lane case).  A candidate is written to the lane's LDS ring slot `cnt` UNCONDITIONALLY and cnt advances by
// (v < q) as sign-bit arithmetic: a rejected candidate is simply overwritten by the next one -- no compare, no divergent
// branch, no VCC (56 candidates per rate block; VCC-form selects cost ~22 cycles each on gfx950).  The first four blocks
// cannot reach 256 coefficients (4 x 56 = 224), so only later blocks pay for the `cnt < 256` clamp.
template <bool CLAMP, int RING>
__device__ __forceinline__ void emit23b(uint32_t v, uint32_t& val, uint32_t& slot, int& cnt)
{
    val = v & 0x7FFFFFu;
    slot = (uint32_t)(cnt & (RING - 1)) * 64u;
    int acc = (int)((val - QU_BODY) >> 31);                      // 1 iff v < q
    if (CLAMP) acc &= (int)((uint32_t)(cnt - 256) >> 31);        // ... and cnt < 256
    cnt += acc;
}
template <bool CLAMP, class EaSink>
__device__ __forceinline__ void expand_a_block(const Shake<21>& sp, EaSink& sink, int& cnt)
{
#pragma unroll
    for (int g = 0; g < 7; g++) {
        const uint64_t w0 = sp.s[3 * g], w1 = sp.s[3 * g + 1], w2 = sp.s[3 * g + 2];
        // the eight candidates of three state words: values and ring slots first (a chain of adds), then the eight LDS
        // writes back to back from eight different registers
        uint32_t val[8], slot[8];
        emit23b<CLAMP, EaSink::RING>((uint32_t)w0, val[0], slot[0], cnt);
        emit23b<CLAMP, EaSink::RING>((uint32_t)(w0 >> 24), val[1], slot[1], cnt);
        emit23b<CLAMP, EaSink::RING>((uint32_t)((w0 >> 48) | (w1 << 16)), val[2], slot[2], cnt);
        emit23b<CLAMP, EaSink::RING>((uint32_t)(w1 >> 8), val[3], slot[3], cnt);
        emit23b<CLAMP, EaSink::RING>((uint32_t)(w1 >> 32), val[4], slot[4], cnt);
        emit23b<CLAMP, EaSink::RING>((uint32_t)((w1 >> 56) | (w2 << 8)), val[5], slot[5], cnt);
        emit23b<CLAMP, EaSink::RING>((uint32_t)(w2 >> 16), val[6], slot[6], cnt);
        emit23b<CLAMP, EaSink::RING>((uint32_t)(w2 >> 40), val[7], slot[7], cnt);
        uint32_t* at[8];
#pragma unroll
        for (int e = 0; e < 8; e++) at[e] = sink.ring + slot[e];
        __builtin_amdgcn_sched_barrier(0);          // addresses and values complete: eight stores in a row, no register reused between them
#pragma unroll
        for (int e = 0; e < 8; e++) *at[e] = val[e];
        __builtin_amdgcn_sched_barrier(0);
        sink.flush_if_ready(cnt);
    }
}
// body of expand_a_fast_kernel<P24> (hash_kernels.hip) for workgroup `block`; `ring`: CoeffSinkWaveT<P24>::LDS_DWORDS_PER_WAVE dwords
template <bool P24>        // P24: A leaves as 24-bit packed coefficients, 768 bytes per polynomial (the internal format of the composite calls)
__device__ __forceinline__ void expand_a_fast_body(int32_t* __restrict__ A, const uint64_t* __restrict__ rho, size_t rho_stride_words, int K,
                                                   int L, size_t nitems, unsigned block, uint32_t* ring)
{
    const size_t p = (size_t)block * HASH_BS + threadIdx.x;
    const size_t total = nitems * (size_t)(K * L);
    const bool live = p < total;
    const size_t item = live ? p / (size_t)(K * L) : 0;
    const int ij = (int)(p % (size_t)(K * L)), i = ij / L, j = ij % L;
    Shake<21> sp;
    sp.init();
#pragma unroll
    for (int w = 0; w < 4; w++) sp.s[w] = rho[item * rho_stride_words + w];
    sp.s[4] = (uint64_t)j | ((uint64_t)i << 8) | (0x1Full << 16);
    sp.s[20] ^= 0x8000000000000000ull;
    // a lane without a polynomial runs along (its ring column is its own) but never stores
    const size_t first = (size_t)block * HASH_BS;             // one wave per workgroup: polynomial of lane 0
    CoeffSinkWaveT<P24> sink(ring, threadIdx.x & 63, A + first * CoeffSinkWaveT<P24>::POLY_DW, (int)(total - first < 64 ? total - first : 64));
    int cnt = 0;
    // Where the time goes (profiles/r02_expand_a.txt, ablations with the sponge kept observable; level 3, 8192 keys): permutations +
    // candidate arithmetic 160 of 183 us (7.7 G perm/s: ONE generation of 3.75 waves per SIMD, i.e. four waves deep on most SIMDs;
    // 9.6 G perm/s at 32768 keys), ring writes 8, flush reads 5, store issue 3, the HBM write stream 8.  Tried and not kept: chunk
    // rotation / chunk-major store layouts, 1.25 - 5 resident waves per SIMD, phase-shifted workgroups, s_sleep staggering, several
    // concurrent launches inside a composite call.
#pragma unroll 1
    for (int blk = 0; blk < 4; blk++) {
        keccak_f1600(sp.s);
        expand_a_block<false>(sp, sink, cnt);
    }
    do {
        keccak_f1600(sp.s);
        expand_a_block<true>(sp, sink, cnt);
    } while (__any(cnt < 256));
}


constexpr int EXPAND_S_LDS_DWORDS = 64 * 69;
// body of expand_s_fast_kernel<ETA> (codec_kernels.hip: gen_s.v / rejection_s.v) for workgroup `block`; buf: EXPAND_S_LDS_DWORDS of LDS
template <int ETA>
__device__ __forceinline__ void expand_s_fast_body(int32_t* __restrict__ s, int32_t* __restrict__ s_tail, int split,
                                                   const uint8_t* __restrict__ rhoprime, size_t rp_stride, int nonce0, int polys,
                                                   size_t nitems, unsigned block, uint32_t* buf)
{
    constexpr int ROW_DW = 69, LIM = ETA == 2 ? 15 : 9;
    const int lane = threadIdx.x;
    const size_t first = (size_t)block * HASH_BS, total = nitems * (size_t)polys;
    const size_t p = first + lane;
    const bool live = p < total;
    const size_t item = live ? p / (size_t)polys : 0;
    const int j = (int)(p % (size_t)polys);
    Shake<17> sp;
    sp.init();
    const uint8_t* rp = rhoprime + item * rp_stride;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        uint32_t lo, hi;
        __builtin_memcpy(&lo, rp + 8 * w, 4);          // any alignment (single dword loads on this target)
        __builtin_memcpy(&hi, rp + 8 * w + 4, 4);
        sp.s[w] = ((uint64_t)hi << 32) | lo;
    }
    sp.s[8] = (uint64_t)(uint32_t)(nonce0 + j) | (0x1Full << 16);
    sp.s[16] ^= 0x8000000000000000ull;
    uint8_t* mine = reinterpret_cast<uint8_t*>(buf) + lane * (ROW_DW * 4);
    int cnt = live ? 0 : 256;
    do {
        keccak_f1600(sp.s);
#pragma unroll
        for (int w = 0; w < 17; w++) {
            const int32_t act = sgn(cnt - 256);                       // this word still counts (<= 15 bytes of overshoot)
            const uint32_t half[2] = {(uint32_t)sp.s[w], (uint32_t)(sp.s[w] >> 32)};
#pragma unroll
            for (int n = 0; n < 16; n++) {
                const uint32_t nib = (half[n >> 3] >> (4 * (n & 7))) & 15u;
                mine[cnt] = (uint8_t)nib;
                cnt -= sgn((int32_t)nib - LIM) & act;
            }
        }
    } while (__any(cnt < 256));
    __syncthreads();
    const int nlive = (int)(total - first < 64 ? total - first : 64);
    size_t it = first / (size_t)polys;                                // wave-uniform walk over this wave's polynomials
    int jj = (int)(first % (size_t)polys);
    for (int q = 0; q < nlive; q++) {
        const uint32_t d = buf[q * ROW_DW + lane];
        int32_t c[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int nib = (int)((d >> (8 * m)) & 15u);
            const int v = ETA == 2 ? 2 - (nib - ((205 * nib) >> 10) * 5) : 4 - nib;
            c[m] = v + ((v >> 31) & (int32_t)QU_BODY);
        }
        int32_t* out = jj < split ? s + (it * split + jj) * 256 : s_tail + (it * (size_t)(polys - split) + (jj - split)) * 256;
        *reinterpret_cast<int4*>(out + 4 * lane) = make_int4(c[0], c[1], c[2], c[3]);
        if (++jj == polys) {
            jj = 0;
            it++;
        }
    }
}


// SampleInBall (gen_c.v:163-196,318-339), one lane per signature: c~ -> SHAKE256 -> 8 sign bytes, then for i = 256 - tau .. 255
// bytes b until b <= i;  c[i] = c[b];  c[b] = +-1.  The lane's c[] and its current rate block live in LDS:
//   cl   int8 [256][PITCH]   PITCH = ITEMS + 4 bytes: a multiple of 4 (dword reads in the output stage) whose dword count per
//                            row is odd, so the transposed reads of the output stage are bank-conflict-free
//   rb   uint32 [34][ITEMS]  the 136-byte rate block as dwords (spilled with 34 stores instead of 136 byte stores)
// ITEMS (64 or 32) signatures per workgroup of one wave: with 32, lanes 32..63 mirror lanes 0..31 (same item, same LDS cells,
// same values) and the footprint halves -- for launches that share the CU with LDS-hungry neighbours.
template <int ITEMS>
struct SibLds {
    static constexpr int PITCH = ITEMS + 4;
    static constexpr int CL_BYTES = 256 * PITCH, RB_DWORDS = 34 * ITEMS, BYTES = CL_BYTES + 4 * RB_DWORDS;
};

// The sponge behind the sampling loop, in either form of keccak.hpp: how its rate block reaches rb, and the 64 sign bits.
struct SibOneLane {                               // one sponge per lane
    Shake<17>& sp;
    template <int ITEMS>
    __device__ __forceinline__ void spill(uint32_t* rb, int col) const
    {
#pragma unroll
        for (int w = 0; w < 17; w++) {
            rb[(2 * w) * ITEMS + col] = (uint32_t)sp.s[w];
            rb[(2 * w + 1) * ITEMS + col] = (uint32_t)(sp.s[w] >> 32);
        }
    }
    __device__ __forceinline__ uint64_t signs() const { return sp.s[0]; }
    __device__ __forceinline__ void permute() { keccak_f1600(sp.s); }
};
struct SibTwoLane {                               // two lanes per sponge: each lane spills its own half of every word, both lanes of
    Shake2<17>& sp;                               // a pair then run the same loop on the same LDS cells with the same values
    template <int ITEMS>
    __device__ __forceinline__ void spill(uint32_t* rb, int col) const
    {
#pragma unroll
        for (int w = 0; w < 17; w++) rb[(2 * w + (sp.hi ? 1 : 0)) * ITEMS + col] = sp.s[w];
    }
    __device__ __forceinline__ uint64_t signs() const
    {
        const uint32_t own = sp.s[0], par = k2_partner(own);
        return sp.hi ? (((uint64_t)own << 32) | par) : (((uint64_t)par << 32) | own);
    }
    __device__ __forceinline__ void permute() { keccak2_f1600(sp.s, sp.hi); }
};

__device__ __forceinline__ void sib_clear(int8_t* cl, int cl_bytes)
{
    for (int k = threadIdx.x; k < cl_bytes / 16; k += 64) reinterpret_cast<uint4*>(cl)[k] = make_uint4(0, 0, 0, 0);
}

// The sampling loop: `sq` is the sponge of column `col` after its first squeeze permutation; cl must be clear (and the clear
// visible: __syncthreads between sib_clear and this).  tau = 0 for a column without a signature.
template <int ITEMS, class SQ>
__device__ __forceinline__ void sib_sample(SQ sq, int col, int tau, int8_t* cl, uint32_t* rb)
{
    constexpr int PITCH = SibLds<ITEMS>::PITCH;
    uint64_t signs = sq.signs();
    sq.template spill<ITEMS>(rb, col);
    // Every lane consumes ONE byte per step (so the read position is wave-uniform and the loop is flat): the byte is either
    // taken for the lane's current i or skipped.  The wave runs max-over-lanes(bytes consumed) ~ tau + 12 steps, where a loop
    // over i with an inner rejection loop runs sum-over-i(max-over-lanes(tries)) ~ 2.7 tau.
    int pos = 8, i = 256 - tau;
    while (__any(i < 256)) {
        if (pos == 136) {
            sq.permute();
            sq.template spill<ITEMS>(rb, col);
            pos = 0;
        }
        const int b = (int)((rb[(pos >> 2) * ITEMS + col] >> (8 * (pos & 3))) & 255u);
        pos++;
        if (i < 256 && b <= i) {
            cl[i * PITCH + col] = cl[b * PITCH + col];
            cl[b * PITCH + col] = (int8_t)(1 - 2 * (int)(signs & 1));
            signs >>= 1;
            i++;
        }
    }
    __syncthreads();
}

// fills cl for the workgroup's signatures (item = base + (lane & (ITEMS - 1)); c~ at ctilde + item * ct_stride, any alignment)
template <int ITEMS>
__device__ __forceinline__ void sample_in_ball_core(const uint8_t* __restrict__ ctilde, size_t ct_stride, int tau, size_t nitems, size_t base,
                                                    int8_t* cl, uint32_t* rb)
{
    const int lane = threadIdx.x, col = lane & (ITEMS - 1);
    const size_t item = base + col;
    const bool live = item < nitems;
    sib_clear(cl, SibLds<ITEMS>::CL_BYTES);
    Shake<17> sp;
    sp.init();
    if (live) {
        const uint8_t* ct = ctilde + item * ct_stride;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t lo, hi;
            __builtin_memcpy(&lo, ct + 8 * w, 4);
            __builtin_memcpy(&hi, ct + 8 * w + 4, 4);
            sp.s[w] = ((uint64_t)hi << 32) | lo;
        }
    }
    sp.s[4] = 0x1Full;
    sp.s[16] ^= 0x8000000000000000ull;
    keccak_f1600(sp.s);
    __syncthreads();                              // cl cleared by all lanes before any lane writes its column
    sib_sample<ITEMS>(SibOneLane{sp}, col, tau, cl, rb);
}

// output stage, compact form (wire_kernels.hip decode_c): cbits[item][lane] bit m = c[lane + 64 m] != 0, bit 4 + m = its sign
template <int ITEMS>
__device__ __forceinline__ void sample_in_ball_bits_body(uint32_t* __restrict__ cbits, const uint8_t* __restrict__ ctilde, size_t ct_stride,
                                                         int tau, size_t nitems, unsigned block, int8_t* cl, uint32_t* rb)
{
    constexpr int PITCH = SibLds<ITEMS>::PITCH;
    const int lane = threadIdx.x;
    const size_t base = (size_t)block * ITEMS;
    sample_in_ball_core<ITEMS>(ctilde, ct_stride, tau, nitems, base, cl, rb);
    for (int t0 = 0; t0 < ITEMS; t0 += 4) {                           // four signatures per step: one dword of each row
        if (base + t0 >= nitems) break;
        uint32_t d[4];
#pragma unroll
        for (int m = 0; m < 4; m++) d[m] = *reinterpret_cast<const uint32_t*>(cl + (lane + 64 * m) * PITCH + t0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (base + t0 + j < nitems) {
                uint32_t w = 0;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const uint32_t b = d[m] >> (8 * j);            // the byte is 0x00, 0x01 or 0xFF: bit 0 = non-zero, bit 7 = negative
                    w |= (b & 1u) << m;                            // (compare-free: a VCC-form select costs ~22 cycles on this chip)
                    w |= ((b >> 7) & 1u) << (4 + m);
                }
                cbits[(base + t0 + j) * 64 + lane] = w;
            }
        }
    }
}

// output stage, polynomial form (the signing loop): c[item][256] int32, canonical (+1 -> 1, -1 -> q - 1)
template <int ITEMS>
__device__ __forceinline__ void sib_store_poly(int32_t* __restrict__ c_out, size_t nitems, size_t base, const int8_t* cl)
{
    constexpr int PITCH = SibLds<ITEMS>::PITCH;
    const int lane = threadIdx.x;
    for (int t0 = 0; t0 < ITEMS; t0 += 4) {
        if (base + t0 >= nitems) break;
        uint32_t d[4];
#pragma unroll
        for (int m = 0; m < 4; m++) d[m] = *reinterpret_cast<const uint32_t*>(cl + (lane + 64 * m) * PITCH + t0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (base + t0 + j < nitems) {
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int v = (int)(int8_t)(d[m] >> (8 * j));
                    c_out[(base + t0 + j) * 256 + lane + 64 * m] = v + ((v >> 31) & (int32_t)QU_BODY);
                }
            }
        }
    }
}
template <int ITEMS>
__device__ __forceinline__ void sample_in_ball_poly_body(int32_t* __restrict__ c_out, const uint8_t* __restrict__ ctilde, size_t ct_stride,
                                                         int tau, size_t nitems, unsigned block, int8_t* cl, uint32_t* rb)
{
    const size_t base = (size_t)block * ITEMS;
    sample_in_ball_core<ITEMS>(ctilde, ct_stride, tau, nitems, base, cl, rb);
    sib_store_poly<ITEMS>(c_out, nitems, base, cl);
}

}  // namespace dil
