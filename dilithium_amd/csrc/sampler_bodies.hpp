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
    v &= 0x7FFFFFu;
    if (v < QU_BODY && cnt < 256) {
        if (writer) sink.put(cnt, (int32_t)v);
        cnt++;
    }
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


// body of sample_in_ball_bits_kernel (wire_kernels.hip) for workgroup `block`; LDS: cl[256 * 64] = c[idx][lane],
// rb[136 * 64] = rate block bytes [pos][lane]
__device__ __forceinline__ void sample_in_ball_bits_body(uint32_t* __restrict__ cbits, const uint8_t* __restrict__ ctilde, size_t ct_stride,
                                                         int tau, size_t nitems, unsigned block, int8_t* cl, uint8_t* rb)
{
    const int lane = threadIdx.x;
    const size_t base = (size_t)block * 64;
    const size_t item = base + lane;
    const bool live = item < nitems;
    for (int k = 0; k < 256; k++) cl[k * 64 + lane] = 0;
    Shake<17> sp;
    sp.init();
    if (live) {
        const uint8_t* ct = ctilde + item * ct_stride;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint64_t v = 0;
#pragma unroll
            for (int b = 0; b < 8; b++) v |= (uint64_t)ct[8 * w + b] << (8 * b);
            sp.s[w] = v;
        }
    }
    sp.s[4] = 0x1Full;
    sp.s[16] ^= 0x8000000000000000ull;
    keccak_f1600(sp.s);
    uint64_t signs = sp.s[0];
    auto spill = [&]() {
#pragma unroll
        for (int w = 0; w < 17; w++)
#pragma unroll
            for (int b = 0; b < 8; b++) rb[(8 * w + b) * 64 + lane] = (uint8_t)(sp.s[w] >> (8 * b));
    };
    spill();
    int pos = 8;
    for (int i = 256 - tau; i < 256; i++) {
        int b;
        do {
            if (pos == 136) {
                keccak_f1600(sp.s);
                spill();
                pos = 0;
            }
            b = rb[pos * 64 + lane];
            pos++;
        } while (b > i);
        cl[i * 64 + lane] = cl[b * 64 + lane];
        cl[b * 64 + lane] = (int8_t)(1 - 2 * (int)(signs & 1));
        signs >>= 1;
    }
    __syncthreads();
    // item t of this block, consumer lane `lane`: coefficients lane + 64 m
    for (int t = 0; t < 64; t++) {
        if (base + t >= nitems) break;
        uint32_t w = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int v = cl[(lane + 64 * m) * 64 + t];
            w |= (uint32_t)(v != 0) << m;
            w |= (uint32_t)(v < 0) << (4 + m);
        }
        cbits[(base + t) * 64 + lane] = w;
    }
}


}  // namespace dil
