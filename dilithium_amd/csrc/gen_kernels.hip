// gen_kernels.hip -- wire-format verification with the matrix GENERATED INSIDE the consuming kernel (SURVEY 8(f) row N1
// cashed in): A = ExpandA(rho) never crosses HBM.  What the reference does with its SHAKE128 core feeding the
// rejection sampler straight into the multiply-accumulate (gen_a_ext.v, sampler_a_ext.v:107,129, rejection_a.v:67-73,
// combined_top.v:1149-1207) is done here inside one wavefront that changes shape twice:
//
//   phase 1  wave-per-polynomial (64 lanes x 4 coefficients, ntt_core.hpp): decode z (decoder.v:89-143), ||z|| check
//            (norm_check.v:84-105), z^ = NTT(z) -> LDS;  c (compact SampleInBall bits) and t1 2^13 -> NTT;
//            w[i] := -c^ o t1^[i] (Montgomery-reduced) -> LDS accumulators
//   phase 2  LANE-per-sponge (keccak.hpp): lane (item, i, j) runs SHAKE128(rho || j || i) and, for every accepted 23-bit
//            candidate a = A^[i][j][k] (k = its running count), does  w[i][k] += a * z^[j][k] * 2^-32 mod q  with one LDS read
//            and one LDS atomic add -- the L lanes of a row add into the same word; rejections (0.1 %) only make the
//            lanes' k drift apart, which costs nothing in this form (no transposition, no ring, no flush)
//   phase 3  wave-per-polynomial again: w[i] -> INTT -> UseHint with the signature's hint bits (usehint.v:92-159) ->
//            w1 packed (encoder.v:96-133)
//
// LDS layout (one wave per workgroup, IPW items per wave so that IPW * K * L <= 64 sponge lanes):
//     zl[k][slot]   slot = item * L + j      256 x ZS dwords      wl[k][row]   row = item * K + i      256 x WR dwords
// Coefficient-major with the slot / row innermost: lanes at the same k touch ZS (WR) consecutive dwords, i.e. distinct
// banks, and the lanes that share a slot read the same word (a broadcast).  The wave-layout phases pay bank conflicts on
// their strided accesses to these arrays; they are ~70 LDS instructions per item against ~600 in phase 2.
// Arithmetic: the same as verify_wire_wpi_kernel -- sum of Montgomery-reduced products == Montgomery reduction of the
// sum (mod q); the pipeline-flavour inverse table cancels the 2^-32.
#include "launch_util.hpp"
#include "wire_common.hpp"
#include "keccak.hpp"

namespace dil {


template <int LEVEL>
struct Gen {
    static constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L, KL = K * L;
#ifdef DIL_GEN_IPW2
    static constexpr int IPW = LEVEL == 2 ? DIL_GEN_IPW2 : 64 / KL;
#else
    static constexpr int IPW = LEVEL == 2 ? 3 : 64 / KL;        // level 2: 3 items = 24.4 KiB -> 6 waves per CU (4 items: 32.5 KiB -> 4)
#endif
    static constexpr int ZS = IPW * L, WR = IPW * K;
    static constexpr int BM_DW = IPW * K * 8;                  // hint bitmaps
    static constexpr int LDS_DW = 256 * (ZS + WR) + BM_DW;
};

// The candidates of one rate block (56 three-byte groups), eight at a time, software-pipelined by hand: the positions
// k of a group depend only on the accept masks before it (a chain of subtractions, no memory), so the eight z^ reads of
// group g + 1 are issued BEFORE the products and atomic adds of group g -- the wave never sits on a single LDS round trip.
// `dead` = 0x7FFFFF in lanes without a polynomial: every candidate becomes 2^23 - 1 >= q, i.e. rejected.  Branch-free:
// a rejected candidate adds 0 at the current position.
template <bool CLAMP, int ZS>
struct GenGroup {
    uint32_t v[8];
    int32_t mask[8];
    int idx[8];
    int32_t z[8];
    __device__ __forceinline__ void one(int e, uint32_t raw, uint32_t dead, const uint32_t* zb, int& cnt)
    {
        v[e] = (raw & 0x7FFFFFu) | dead;
        mask[e] = sgn((int32_t)(v[e] - (uint32_t)Q));         // all-ones iff v < q
        idx[e] = cnt;
        if (CLAMP) {                                          // only from the fifth block on can a lane have all 256
            mask[e] &= sgn(cnt - 256);
            idx[e] = min(cnt, 255);
        }
        cnt -= mask[e];
        z[e] = (int32_t)zb[idx[e] * ZS];
    }
    __device__ __forceinline__ void pick(const uint64_t (&s)[25], int g, uint32_t dead, const uint32_t* zb, int& cnt)
    {
        const uint64_t w0 = s[3 * g], w1 = s[3 * g + 1], w2 = s[3 * g + 2];
        one(0, (uint32_t)w0, dead, zb, cnt);
        one(1, (uint32_t)(w0 >> 24), dead, zb, cnt);
        one(2, (uint32_t)((w0 >> 48) | (w1 << 16)), dead, zb, cnt);
        one(3, (uint32_t)(w1 >> 8), dead, zb, cnt);
        one(4, (uint32_t)(w1 >> 32), dead, zb, cnt);
        one(5, (uint32_t)((w1 >> 56) | (w2 << 8)), dead, zb, cnt);
        one(6, (uint32_t)(w2 >> 16), dead, zb, cnt);
        one(7, (uint32_t)(w2 >> 40), dead, zb, cnt);
    }
    template <int WR>
    __device__ __forceinline__ void mac(uint32_t* wb) const
    {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int32_t p = mont_red64((int64_t)(int32_t)v[e] * z[e]) & mask[e];
            __hip_atomic_fetch_add(reinterpret_cast<int32_t*>(wb + idx[e] * WR), p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
};

template <bool CLAMP, int ZS, int WR>
__device__ __forceinline__ void gen_block(const uint64_t (&s)[25], uint32_t dead, const uint32_t* zb, uint32_t* wb, int& cnt)
{
    GenGroup<CLAMP, ZS> grp[2];
    grp[0].pick(s, 0, dead, zb, cnt);
#pragma unroll
    for (int g = 0; g < 7; g++) {
        if (g + 1 < 7) grp[(g + 1) & 1].pick(s, g + 1, dead, zb, cnt);
        grp[g & 1].template mac<WR>(wb);
    }
}

template <int LEVEL>
__global__ __launch_bounds__(64) void verify_wire_gen_kernel(
    uint8_t* __restrict__ w1p_out, int32_t* __restrict__ verdict, const uint8_t* __restrict__ pk, size_t pk_stride,
    const uint8_t* __restrict__ sig, size_t sig_stride, const uint32_t* __restrict__ cbits, size_t batch,
    const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab)
{
    using G = Gen<LEVEL>;
    using W = Wire<LEVEL>;
    constexpr int K = G::K, L = G::L, KL = G::KL, IPW = G::IPW, ZS = G::ZS, WR = G::WR;
    __shared__ __attribute__((aligned(16))) uint32_t lds[G::LDS_DW];
    uint32_t* zl = lds;
    uint32_t* wl = lds + 256 * ZS;
    uint32_t* bm = wl + 256 * WR;
    const int lane = threadIdx.x;
    const size_t it0 = (size_t)blockIdx.x * IPW;
    const X10Dpp lm(lane);

    // ---- phase 1: wave layout ------------------------------------------------------------------------------
    {
        TwRegs twf;
        twf.load(fwd_tab, lane);
        const PackedLane<W::ZBITS> plz(lane);
        const PackedLane<10> plt(lane);
        // hint bytes -> per-row bitmaps for every item first: the decoder's byte scratch is the (still unwritten) z^ array
        uint32_t badmask = 0;
#pragma unroll
        for (int t = 0; t < IPW; t++) {
            if (it0 + t < batch) {
                const uint8_t* hp = sig + (it0 + t) * sig_stride + 32 + W::Z_BYTES;
                const uint32_t hb0 = (lane < W::HINT_BYTES) ? hp[lane] : 0;          // (never past the signature's end)
                const uint32_t hb1 = (64 + lane < W::HINT_BYTES) ? hp[64 + lane] : 0;
                badmask |= (hints_to_bitmap<LEVEL>(bm + t * K * 8, zl, hb0, hb1, lane) ? 1u : 0u) << t;
            }
        }
#pragma unroll 1
        for (int t = 0; t < IPW; t++) {
            const size_t it = it0 + t;
            if (it >= batch) break;          // ragged last wave: the missing items' lanes are dead in phase 2
            const uint8_t* sg = sig + it * sig_stride;
            const uint8_t* t1it = pk + it * pk_stride + 32;
            RawZ<LEVEL> zr;
            zr.load(sg + 32, plz);
            const uint32_t cb = cbits[it * 64 + lane];
            const bool bad = (badmask >> t) & 1u;
            int32_t zmax = 0;
#pragma unroll
            for (int l = 0; l < L; l++) {
                int32_t r[4];
                decode_z<LEVEL>(r, zr.v[l], plz, zmax);
                ntt_fwd_core(r, twf, lm);
#pragma unroll
                for (int m = 0; m < 4; m++) zl[(4 * lane + m) * ZS + t * L + l] = (uint32_t)r[m];
            }
            const bool zrej = __ballot(zmax >= Par<LEVEL>::GAMMA1 - Par<LEVEL>::BETA) != 0;
            if (lane == 0) verdict[it] = (zrej ? 2 : 0) | (bad ? 4 : 0);
            int32_t ch[4];
            decode_c(ch, cb);
            ntt_fwd_core(ch, twf, lm);
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                uint32_t tn[4], f[4];
                plt.load(tn, t1it + k * 320);
                plt.fields(f, tn);
                int32_t th[4];
#pragma unroll
                for (int m = 0; m < 4; m++) th[m] = (int32_t)(f[m] << 13);   // decoder.v:96-100
                ntt_fwd_core(th, twf, lm);
#pragma unroll
                for (int m = 0; m < 4; m++)
                    wl[(4 * lane + m) * WR + t * K + k] = (uint32_t)mont_red64(-(int64_t)ch[m] * th[m]);
            }
        }
    }
    __syncthreads();

    // ---- phase 2: lane per sponge ----------------------------------------------------------------------------
    {
        const int t = lane / KL, ij = lane - t * KL, i = ij / L, j = ij - i * L;
        const bool live = lane < IPW * KL && it0 + t < batch;
        const uint32_t dead = live ? 0u : 0x7FFFFFu;
        const int tt = live ? t : 0, ii = live ? i : 0, jj = live ? j : 0;
        const uint64_t* rho = reinterpret_cast<const uint64_t*>(pk + (it0 + tt) * pk_stride);
        uint64_t s[25];
#pragma unroll
        for (int w = 0; w < 4; w++) s[w] = rho[w];
        s[4] = (uint64_t)jj | ((uint64_t)ii << 8) | (0x1Full << 16);
#pragma unroll
        for (int w = 5; w < 25; w++) s[w] = 0;
        s[20] = 0x8000000000000000ull;
        const uint32_t* zb = zl + tt * L + jj;
        uint32_t* wb = wl + tt * K + ii;
        int cnt = live ? 0 : 256;
#pragma unroll 1
        for (int blk = 0; blk < 4; blk++) {
            keccak_f1600(s);
            gen_block<false, ZS, WR>(s, dead, zb, wb, cnt);
        }
        do {
            keccak_f1600(s);
            gen_block<true, ZS, WR>(s, dead, zb, wb, cnt);
        } while (__any(cnt < 256));
    }
    __syncthreads();

    // ---- phase 3: wave layout ------------------------------------------------------------------------------
    {
        TwRegs twi;
        twi.load(inv_tab, lane);
        uint32_t* sc = zl;                       // z^ is dead: byte scratch of the packed w1 store
#pragma unroll 1
        for (int t = 0; t < IPW; t++) {
            const size_t it = it0 + t;
            if (it >= batch) break;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                int32_t r[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    // |sum| < (L + 1) q < 2^26: fold to (-q, q) for the inverse transform (2^23 = 2^13 - 1 mod q)
                    const int32_t x = (int32_t)wl[(4 * lane + m) * WR + t * K + k];
                    const int32_t hi = x >> 23;
                    int32_t y = (x & 0x7FFFFF) + (hi << 13) - hi;            // in (-2^16, 2^23 + 2^16)
                    y -= ~sgn(y - Q) & Q;                                      // y >= q: y - q
                    r[m] = y;
                }
                uint32_t hb[4];
                row_hint_bits(hb, bm + t * K * 8, k, lane);
                ntt_inv_core(r, twi, lm);
                uint32_t wb4[4];
#pragma unroll
                for (int m = 0; m < 4; m++) wb4[m] = use_hint<LEVEL>(canon_small(r[m]), hb[m]);
                store_row_w1_packed<LEVEL>(w1p_out + (it * K + k) * W::W1_ROW_BYTES, wb4, sc, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        }
    }
}

template <int LEVEL>
static hipError_t launch_gen_level(uint8_t* w1p, int32_t* verdict, const uint8_t* pk, size_t pk_stride, const uint8_t* sig, size_t sig_stride,
                                   const uint32_t* cbits, size_t batch, const Tables& t, hipStream_t s)
{
    constexpr int IPW = Gen<LEVEL>::IPW;
    const size_t g = (batch + IPW - 1) / IPW;
    hipLaunchKernelGGL(verify_wire_gen_kernel<LEVEL>, (unsigned)g, 64, 0, s, w1p, verdict, pk, pk_stride, sig, sig_stride, cbits, batch,
                       t.fwd, t.inv_pipe);
    return hipGetLastError();
}

hipError_t launch_verify_wire_gen(int level, uint8_t* w1p, int32_t* verdict, const uint8_t* pk, size_t pk_stride, const uint8_t* sig,
                                  size_t sig_stride, const uint32_t* cbits, size_t batch, const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    switch (level) {
    case 2: return launch_gen_level<2>(w1p, verdict, pk, pk_stride, sig, sig_stride, cbits, batch, t, s);
    case 3: return launch_gen_level<3>(w1p, verdict, pk, pk_stride, sig, sig_stride, cbits, batch, t, s);
    case 5: return launch_gen_level<5>(w1p, verdict, pk, pk_stride, sig, sig_stride, cbits, batch, t, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace dil
