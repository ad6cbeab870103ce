// wire_common.hpp -- device helpers shared by the wire-format verify kernels (wire_kernels.hip, gen_kernels.hip): lane
// addressing of the packed fields (decoder.v:89-143), hint decoding with the reference's validity checks
// (usehint.v:92-114), the compact challenge form and z's range tracking (norm_check.v:84-105).
#pragma once
#include "pipeline_common.hpp"

namespace dil {

// one 4-byte load at any byte address (hipcc emits a single global_load_dword: the target runs in unaligned-access mode)
__device__ __forceinline__ uint32_t ld_u32u(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// Lane addressing of one BITS-wide packed polynomial (32 * BITS bytes) in NTT-input order: lane j wants coefficients
// j + 64 m.  Coefficient i sits at bit BITS * i; 64 coefficients are exactly 8 * BITS bytes, so the dword of (j, m) is
// at byte off0 + 8 * BITS * m with the same shift for every m -- except that the last dwords of the polynomial would
// reach past its end (and, for the last item of a batch, past the buffer): m = 3 clamps to the polynomial's last 4 bytes.
template <int BITS>
struct PackedLane {
    uint32_t off0, sh0, off3, sh3;
    __device__ __forceinline__ explicit PackedLane(int lane)
    {
        const uint32_t bit = BITS * (uint32_t)lane;
        off0 = bit >> 3;
        sh0 = bit & 7;
        const uint32_t want = off0 + 24 * BITS, last = 32 * BITS - 4;
        off3 = want < last ? want : last;
        sh3 = sh0 + 8 * (want - off3);
    }
    __device__ __forceinline__ void load(uint32_t (&raw)[4], const uint8_t* __restrict__ poly) const
    {
        raw[0] = ld_u32u(poly + off0);
        raw[1] = ld_u32u(poly + off0 + 8 * BITS);
        raw[2] = ld_u32u(poly + off0 + 16 * BITS);
        raw[3] = ld_u32u(poly + off3);
    }
    __device__ __forceinline__ void fields(uint32_t (&f)[4], const uint32_t (&raw)[4]) const
    {
        constexpr uint32_t MASK = (1u << BITS) - 1;
        f[0] = (raw[0] >> sh0) & MASK;
        f[1] = (raw[1] >> sh0) & MASK;
        f[2] = (raw[2] >> sh0) & MASK;
        f[3] = (raw[3] >> sh3) & MASK;
    }
};

template <int LEVEL>
struct Wire {
    static constexpr int ZBITS = LEVEL == 2 ? 18 : 20;
    static constexpr int W1_ROW_BYTES = W1Pack<LEVEL>::ROW_BYTES;     // 192 / 128
    static constexpr int Z_BYTES = Par<LEVEL>::L * 32 * ZBITS;
    static constexpr int HINT_BYTES = Par<LEVEL>::OMEGA + Par<LEVEL>::K;
};

// raw (packed) z of one item, prefetched a whole row phase ahead: 4 dwords per polynomial, as RawPolys
template <int LEVEL>
struct RawZ {
    uint32_t v[Par<LEVEL>::L][4];
    __device__ __forceinline__ void load(const uint8_t* __restrict__ zbase, const PackedLane<Wire<LEVEL>::ZBITS>& pl)
    {
#pragma unroll
        for (int l = 0; l < Par<LEVEL>::L; l++) pl.load(v[l], zbase + l * (32 * Wire<LEVEL>::ZBITS));
    }
};

// Hint bytes -> per-row bitmap in LDS ([K][8] dwords), with the reference decoder's validity checks (usehint.v:92-114,
// the same as hint_unpack_kernel): counts monotone and <= omega, positions strictly increasing inside a row, zero padding.
// hb0 / hb1 = hint bytes `lane` and `64 + lane` of the item (prefetched).  Returns true if the encoding is malformed.
template <int LEVEL>
__device__ __forceinline__ bool hints_to_bitmap(uint32_t* bm, uint32_t* scratch, uint32_t hb0, uint32_t hb1, int lane)
{
    constexpr int K = Par<LEVEL>::K, OMEGA = Par<LEVEL>::OMEGA;
    uint8_t* sc = reinterpret_cast<uint8_t*>(scratch);
    sc[lane] = (uint8_t)hb0;
    if (64 + lane < OMEGA + K) sc[64 + lane] = (uint8_t)hb1;
    if (lane < K * 8) bm[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    bool err = false;
    int cnt[K], prev = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        cnt[k] = sc[OMEGA + k];
        if (cnt[k] < prev || cnt[k] > OMEGA) err = true;
        prev = cnt[k];
    }
    const int total = err ? 0 : prev;
#pragma unroll
    for (int r = 0; r < (OMEGA + 63) / 64; r++) {
        const int t = 64 * r + lane;
        if (t < OMEGA) {
            const int pos = sc[t];
            if (t < total) {
                int row = 0, row_start = 0;
#pragma unroll
                for (int k = 0; k < K; k++)
                    if (cnt[k] <= t) {
                        row = k + 1;
                        row_start = cnt[k];
                    }
                if (t > row_start && pos <= (int)sc[t - 1]) err = true;
                if (row < K) atomicOr(&bm[row * 8 + (pos >> 5)], 1u << (pos & 31));
            } else if (pos != 0) {
                err = true;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    return __ballot(err) != 0;
}

// hint bits of row k for the lane's coefficients lane + 64 m
__device__ __forceinline__ void row_hint_bits(uint32_t (&hb)[4], const uint32_t* bm, int k, int lane)
{
#pragma unroll
    for (int m = 0; m < 4; m++) hb[m] = (bm[k * 8 + 2 * m + (lane >> 5)] >> (lane & 31)) & 1u;
}

// c of SampleInBall in the compact per-lane form sample_in_ball_bits_kernel writes: bit m = c[lane + 64 m] != 0,
// bit 4 + m = its sign (1 = -1)
__device__ __forceinline__ void decode_c(int32_t (&c)[4], uint32_t cb)
{
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int32_t nz = (int32_t)((cb >> m) & 1u), neg = (int32_t)((cb >> (4 + m)) & 1u);
        c[m] = nz - 2 * (nz & neg);
    }
}

// z = gamma1 - field (centred); tracks max |z| for the norm check
template <int LEVEL>
__device__ __forceinline__ void decode_z(int32_t (&z)[4], const uint32_t (&raw)[4], const PackedLane<Wire<LEVEL>::ZBITS>& pl, int32_t& zmax)
{
    uint32_t f[4];
    pl.fields(f, raw);
#pragma unroll
    for (int m = 0; m < 4; m++) {
        z[m] = Par<LEVEL>::GAMMA1 - (int32_t)f[m];
        zmax = max(zmax, max(z[m], -z[m]));
    }
}

}  // namespace dil
