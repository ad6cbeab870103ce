// pipelines.hip -- fused Dilithium pipelines on top of the wavefront NTT (ntt_core.hpp):
// mat-vec (H9), verify core (H8), sign inner loop phases 1 and 2 (H10).
//   rtl_src/combined_top.v:1207-1469 (verify), :1850-1933 (mat-vec / FSM1), :1946-2229 (FSM2)
// Two shapes: workgroup-per-item (small batches, low latency) and wave-per-item (large batches).
#define DIL_PRODUCT_MAD64   // the constant products as two v_mad_i64_i32 (modarith.hpp MAD64): these kernels are VALU-bound
#include "launch_util.hpp"
#include "pipeline_common.hpp"
#include "wire_common.hpp"

namespace dil {

// H9 mat-vec  w = INTT(A o NTT(y))   (OUT_W)   and sign phase 1 = mat-vec + Decompose (OUT_W1W0)
template <int K, int L, int LEVEL, int OUT, int AF>
__global__ __launch_bounds__(64 * (K > L ? K : L)) void matvec_kernel(
    int32_t* __restrict__ w_out, uint8_t* __restrict__ w1_out, int32_t* __restrict__ w0_out,
    const int32_t* __restrict__ A, const int32_t* __restrict__ y, size_t batch, int shared_A,
    KeyMap km, const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[LDS_DWORDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    stage_tables(lds, fwd_tab, inv_tab);
    __syncthreads();
    const TwLds twf{lds, lane}, twi{lds + TW_TABLE_DWORDS, lane};
    const LaneMasks lm(lane);
    uint32_t* sc = lds + LDS_SCR + wv * 64;      // this wave's byte-plane scratch
    uint32_t* vec = lds + LDS_VEC;
    for (size_t it = blockIdx.x; it < batch; it += gridDim.x) {
        ARow<L, AF> Ar;
        if (wv < K) Ar.load(A + ((shared_A ? 0 : km.key(it) * K) + wv) * (size_t)L * ARow<L, AF>::PD, lane, !shared_A && km.S == 1);
        if (wv < L) {
            int32_t r[4];
            load_strided(r, y + (it * L + wv) * 256, lane);
            ntt_fwd_core(r, twf, lm);
            *reinterpret_cast<int4*>(vec + wv * 256 + 4 * lane) = make_int4(r[0], r[1], r[2], r[3]);
        }
        __syncthreads();
        if (wv < K) {
            int64_t acc[4] = {0, 0, 0, 0};
            mac_row<L>(acc, Ar, vec, lane);
            int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
            ntt_inv_core(r, twi, lm);
            emit_matvec_row<LEVEL, OUT>(w_out, w1_out, w0_out, (it * K + wv) * 256, r, sc, lane);
        }
        __syncthreads();
    }
}

// H8 verify core:  w1 = UseHint(h, INTT(A o NTT(z) - NTT(c) o NTT(t1 * 2^13)))
// (combined_top.v VY_NTT_Z :1207, VY_NTT_T1 :1259, VY_NTT_C :1314, VY_MULT_AZ :1347-1386,
//  VY_MULT_CT1 :1387, VY_SUB_AZ_CT1 :1415, VY_INTT :1435, VY_GENW1 :1470)
template <int LEVEL>
__global__ __launch_bounds__(64 * (Par<LEVEL>::K > Par<LEVEL>::L + 1 ? Par<LEVEL>::K : Par<LEVEL>::L + 1))
void verify_kernel(uint8_t* __restrict__ w1_out, const int32_t* __restrict__ A,
                   const int32_t* __restrict__ z, const int32_t* __restrict__ c,
                   const int32_t* __restrict__ t1, const uint8_t* __restrict__ h, size_t batch,
                   int shared_pk, const uint32_t* __restrict__ fwd_tab,
                   const uint32_t* __restrict__ inv_tab)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    __shared__ __attribute__((aligned(16))) uint32_t lds[LDS_DWORDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    stage_tables(lds, fwd_tab, inv_tab);
    __syncthreads();
    const TwLds twf{lds, lane}, twi{lds + TW_TABLE_DWORDS, lane};
    const LaneMasks lm(lane);
    uint32_t* sc = lds + LDS_SCR + wv * 64;      // this wave's byte-plane scratch
    uint32_t* vec = lds + LDS_VEC;
    uint32_t* chat = lds + LDS_CHAT;
    for (size_t it = blockIdx.x; it < batch; it += gridDim.x) {
        ARow<L> Ar;
        int32_t th[4] = {0, 0, 0, 0};
        uint32_t hb[4] = {0, 0, 0, 0};
        const size_t o = (it * K + wv) * 256;
        if (wv < K) {    // issue this row's loads first: A (L x 1 KiB), t1, h
            Ar.load(A + ((shared_pk ? 0 : it * K) + wv) * (size_t)L * 256, lane, !shared_pk);
            const int32_t* src = t1 + ((shared_pk ? 0 : it * K) + wv) * 256;
#pragma unroll
            for (int m = 0; m < 4; m++) th[m] = src[lane + 64 * m];
            unpack_row_u8(hb, load_row_u8(h + o, lane), sc, lane);
        }
        if (wv <= L) {   // waves 0..L-1: z_l ; wave L: c
            int32_t r[4];
            const int32_t* src = (wv < L) ? z + (it * L + wv) * 256 : c + it * 256;
            load_strided(r, src, lane);
            ntt_fwd_core(r, twf, lm);
            uint32_t* dst = (wv < L) ? vec + wv * 256 : chat;
            *reinterpret_cast<int4*>(dst + 4 * lane) = make_int4(r[0], r[1], r[2], r[3]);
        }
        if (wv < K) {    // t1_k * 2^13 (decoder.v:96-100), t1 is 10 bits
#pragma unroll
            for (int m = 0; m < 4; m++) th[m] = (th[m] & 0x3FF) << 13;
            ntt_fwd_core(th, twf, lm);
        }
        __syncthreads();
        if (wv < K) {
            int64_t acc[4] = {0, 0, 0, 0};
            mac_row<L>(acc, Ar, vec, lane);
            const int4 ch = *reinterpret_cast<const int4*>(chat + 4 * lane);
            acc[0] -= (int64_t)ch.x * th[0];
            acc[1] -= (int64_t)ch.y * th[1];
            acc[2] -= (int64_t)ch.z * th[2];
            acc[3] -= (int64_t)ch.w * th[3];
            int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
            ntt_inv_core(r, twi, lm);
            uint32_t wb[4];
#pragma unroll
            for (int m = 0; m < 4; m++) wb[m] = use_hint<LEVEL>(canon_small(r[m]), hb[m]);
            store_row_u8(w1_out + o, wb, sc, lane);
        }
        __syncthreads();
    }
}

// H10 sign phase 2 (operator 1 of the RTL, FSM2 combined_top.v:1981-2229):
//   c^ = NTT(c);  z_l = y_l + INTT(c^ o s1^_l)          reject ||z||  >= gamma1 - beta  (bit 0)
//   r0 = w0_k - INTT(c^ o s2^_k)                         reject ||r0|| >= gamma2 - beta  (bit 1)
//   ct0 = INTT(c^ o t0^_k)                               reject ||ct0||>= gamma2         (bit 2)
//   h_k = MakeHint(r0 + ct0, w1_k)                       reject #h > omega               (bit 3)
template <int LEVEL>
__global__ __launch_bounds__(64 * (Par<LEVEL>::K > Par<LEVEL>::L + 1 ? Par<LEVEL>::K : Par<LEVEL>::L + 1))
void sign2_kernel(int32_t* __restrict__ z_out, uint8_t* __restrict__ h_out, int32_t* __restrict__ flags_out,
                  const int32_t* __restrict__ c, const int32_t* __restrict__ y,
                  const int32_t* __restrict__ w0, const uint8_t* __restrict__ w1,
                  const int32_t* __restrict__ s1hat, const int32_t* __restrict__ s2hat,
                  const int32_t* __restrict__ t0hat, size_t batch, int shared_key,
                  KeyMap km, const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    __shared__ __attribute__((aligned(16))) uint32_t lds[LDS_DWORDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    stage_tables(lds, fwd_tab, inv_tab);
    if (threadIdx.x < 4) lds[LDS_FLAGS + threadIdx.x] = 0;
    __syncthreads();
    const TwLds twf{lds, lane}, twi{lds + TW_TABLE_DWORDS, lane};
    const LaneMasks lm(lane);
    uint32_t* sc = lds + LDS_SCR + wv * 64;      // this wave's byte-plane scratch
    uint32_t* chat = lds + LDS_CHAT;
    uint32_t* fl = lds + LDS_FLAGS;   // [0] reject bits, [1] hint count
    for (size_t it = blockIdx.x; it < batch; it += gridDim.x) {
        if (wv == L) {
            int32_t r[4];
            load_strided(r, c + it * 256, lane);
            ntt_fwd_core(r, twf, lm);
            *reinterpret_cast<int4*>(chat + 4 * lane) = make_int4(r[0], r[1], r[2], r[3]);
        }
        __syncthreads();
        const int4 ch = *reinterpret_cast<const int4*>(chat + 4 * lane);
        uint32_t bits = 0, nh = 0;
        if (wv < L) {
            const int4 s = *reinterpret_cast<const int4*>(s1hat + ((shared_key ? 0 : km.key(it) * L) + wv) * 256 + 4 * lane);
            int32_t r[4] = {mont_mul(ch.x, s.x), mont_mul(ch.y, s.y), mont_mul(ch.z, s.z), mont_mul(ch.w, s.w)};
            ntt_inv_core(r, twi, lm);
            const size_t o = (it * L + wv) * 256;
            bool rej = false;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t v = canon_any(r[m] + y[o + lane + 64 * m]);
                rej |= norm_reject(v, Par<LEVEL>::GAMMA1 - Par<LEVEL>::BETA);
                z_out[o + lane + 64 * m] = (int32_t)v;
            }
            if (__ballot(rej)) bits |= 1;
        }
        if (wv < K) {
            const size_t ko = ((shared_key ? 0 : km.key(it) * K) + wv) * 256 + 4 * lane;
            const int4 s2 = *reinterpret_cast<const int4*>(s2hat + ko);
            const int4 t0 = *reinterpret_cast<const int4*>(t0hat + ko);
            int32_t a[4] = {mont_mul(ch.x, s2.x), mont_mul(ch.y, s2.y), mont_mul(ch.z, s2.z), mont_mul(ch.w, s2.w)};
            int32_t b[4] = {mont_mul(ch.x, t0.x), mont_mul(ch.y, t0.y), mont_mul(ch.z, t0.z), mont_mul(ch.w, t0.w)};
            ntt_inv_core(a, twi, lm);
            ntt_inv_core(b, twi, lm);
            const size_t o = (it * K + wv) * 256;
            bool rej1 = false, rej2 = false;
            uint32_t w1v[4], hv[4];
            unpack_row_u8(w1v, load_row_u8(w1 + o, lane), sc, lane);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t ct0 = canon_small(b[m]);
                const uint32_t r0 = canon_any(w0[o + lane + 64 * m] - a[m]);
                rej1 |= norm_reject(r0, Par<LEVEL>::GAMMA2 - Par<LEVEL>::BETA);
                rej2 |= norm_reject(ct0, Par<LEVEL>::GAMMA2);
                const uint32_t s = canon_small((int32_t)(r0 + ct0) - Q);
                hv[m] = make_hint<LEVEL>(s, w1v[m]);
                nh += __popcll(__ballot(hv[m]));
            }
            store_row_u8(h_out + o, hv, sc, lane);
            if (__ballot(rej1)) bits |= 2;
            if (__ballot(rej2)) bits |= 4;
        }
        if (lane == 0) {
            if (bits) atomicOr(&fl[0], bits);
            if (nh) atomicAdd(&fl[1], nh);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t f = fl[0] | (fl[1] > (uint32_t)Par<LEVEL>::OMEGA ? 8u : 0u);
            flags_out[it] = (int32_t)f;
            fl[0] = 0;
            fl[1] = 0;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// Wave-per-item variants of the fused pipelines (large batches).
// One wavefront carries one whole item through every stage: the L NTT-domain vectors sit in a
// private LDS slice of the wave (keeping them in 4L VGPRs was tried first and cost occupancy), the
// matrix rows stream through registers, and there is no barrier after the one-time twiddle staging.  Rows are software-prefetched: the loads of row
// k+1 are issued as soon as the MACs of row k have consumed the row registers, and fly under
// NTT(t1_k) + INTT(row k).  With batch >= 8 items per SIMD this keeps the VALUs busier than
// the workgroup-per-item kernels above (which remain the low-latency path for small batches).
// ---------------------------------------------------------------------------------------
// raw (time-domain) inputs of one item, prefetched a whole row phase ahead
template <int NP, bool NT = true>
struct RawPolys {
    int32_t v[NP][4];
    __device__ __forceinline__ void load(const int32_t* __restrict__ base, int lane)
    {
#pragma unroll
        for (int p = 0; p < NP; p++) load_strided<NT>(v[p], base + p * 256, lane);
    }
};

// Time-domain y of the signing loop in either of its two HBM forms (kernels.hpp Y_I32 / Y_PACKED): int32 [L][256] canonical, or
// the B-bit packed SHAKE256 stream ExpandMask squeezes (expand_mask_raw_kernel: gamma1 - y, 32 B bytes per polynomial), read the
// way the verify kernels read z.  raw() issues the loads of one polynomial (4 dwords per lane either way), value() turns them
// into the lane's coefficients lane + 64 m -- canonical (Y_I32) or centred in [-gamma1, gamma1] (Y_PACKED); both are valid
// transform inputs and valid addends of z = y + c s1.
template <int LEVEL, int YF>
struct YSrc;
template <int LEVEL>
struct YSrc<LEVEL, Y_I32> {
    static constexpr size_t POLY = 1024;             // bytes per polynomial
    __device__ __forceinline__ explicit YSrc(int) {}
    __device__ __forceinline__ void raw(int32_t (&r)[4], const int32_t* __restrict__ base, size_t poly, int lane) const
    {
        load_strided(r, base + poly * 256, lane);
    }
    __device__ __forceinline__ void value(int32_t (&)[4]) const {}
};
template <int LEVEL>
struct YSrc<LEVEL, Y_PACKED> {
    static constexpr int B = Wire<LEVEL>::ZBITS;
    static constexpr size_t POLY = 32 * B;
    PackedLane<B> pl;
    __device__ __forceinline__ explicit YSrc(int lane) : pl(lane) {}
    __device__ __forceinline__ void raw(int32_t (&r)[4], const int32_t* __restrict__ base, size_t poly, int) const
    {
        uint32_t u[4];
        pl.load(u, reinterpret_cast<const uint8_t*>(base) + poly * POLY);
#pragma unroll
        for (int m = 0; m < 4; m++) r[m] = (int32_t)u[m];
    }
    __device__ __forceinline__ void value(int32_t (&r)[4]) const
    {
        uint32_t u[4] = {(uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]}, f[4];
        pl.fields(f, u);
#pragma unroll
        for (int m = 0; m < 4; m++) r[m] = Par<LEVEL>::GAMMA1 - (int32_t)f[m];
    }
};

#define DIL_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Matrix rows in flight per wave in the per-item-matrix kernels: with ONE row buffer the loads of row k + 1 are issued only after
// the multiply-accumulate of row k has drained the registers, and fly under one INTT; with a ring of two the stream has a whole
// row phase more to land -- mat-vec / sign phase 1 with a matrix per item: level 3 75.5 -> 63.0 us, level 5 134 -> 111 us,
// BASELINE configs[2] (level 2, 4096 items) 20.2 -> 18.0 us; four rows are no better (profiles/r03q_ab_mr.txt).  (The fused
// VERIFY kernels did not gain from the same ring -- profiles/r03k_ab_vw.txt; the wire-format one spills with it at level 5,
// 128 -> 191 us, profiles/r03r_rows_keygen_wire.txt -- and keep one buffer.)
#define DIL_MV_ROWS(K) 2
#define DIL_KG_ROWS(K) 1          // keygen's fused kernel (24-bit packed matrix): two rows in flight change nothing (profiles/r03r_rows_keygen_wire.txt)
// mat-vec / sign phase 1, wave-per-item.  Per item: issue row-0 loads | L forward NTTs on registers
// loaded during the PREVIOUS item's row phase, y^ -> this wave's LDS slice | issue the NEXT item's y
// loads | K rows: MAC from LDS, prefetch row k+1, INTT, (Decompose), store.
template <int K, int L, int LEVEL, int OUT, int AF, int YF>
__global__ __launch_bounds__(256) void matvec_wpi_kernel(
    int32_t* __restrict__ w_out, uint8_t* __restrict__ w1_out, int32_t* __restrict__ w0_out,
    const int32_t* __restrict__ A, const int32_t* __restrict__ y, size_t batch, int shared_A,
    KeyMap km, const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab)
{
    using XP = X10Pick<true>;
    using PT = PipeTables<true>;
    __shared__ __attribute__((aligned(16))) uint32_t lds[PT::DWORDS + 4 * L * 256 + 4 * 64 + 4 * XP::DW];
    const int lane = threadIdx.x & 63, wv = wave_in_block();
    PT::stage(lds, fwd_tab, inv_tab);
    const typename PT::Fwd twf = PT::fwd(lds, fwd_tab, lane);
    const typename PT::Inv twi = PT::inv(lds, inv_tab, lane);
    uint32_t* xb = lds + PT::DWORDS + 4 * L * 256 + 4 * 64 + wv * XP::DW;   // exchange buffer, also the output transpose
    const typename XP::type lm(xb, lane);
    uint32_t* sc = lds + PT::DWORDS + 4 * L * 256 + wv * 64;   // byte-plane scratch
    uint32_t* yl = lds + PT::DWORDS + wv * (L * 256);
    const size_t nwaves = (size_t)gridDim.x * 4;
    size_t it = (size_t)blockIdx.x * 4 + wv;
    const YSrc<LEVEL, YF> ys(lane);
    RawPolys<L> yr;
    auto load_y = [&](size_t i) {
#pragma unroll
        for (int l = 0; l < L; l++) ys.raw(yr.v[l], y, i * L + l, lane);
    };
    if (it < batch) load_y(it);
    __syncthreads();                               // tables staged (the only barrier)
    for (; it < batch; it += nwaves) {
        constexpr int PD = ARow<L, AF>::PD;
        constexpr int NR = DIL_MV_ROWS(K);           // matrix rows in flight per wave
        const int32_t* Ait = A + (shared_A ? 0 : km.key(it) * K) * (size_t)L * PD;
        ARow<L, AF> Ar[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) Ar[j].load(Ait + (size_t)j * L * PD, lane, !shared_A && km.S == 1);
#pragma unroll
        for (int l = 0; l < L; l++) ys.value(yr.v[l]);
        ntt_fwd_coreN<L>(yr.v, twf, lm);
#pragma unroll
        for (int l = 0; l < L; l++) *reinterpret_cast<int4*>(yl + l * 256 + 4 * lane) = make_int4(yr.v[l][0], yr.v[l][1], yr.v[l][2], yr.v[l][3]);
        DIL_SCHED_FENCE();
        const size_t itn = it + nwaves;
        if (itn < batch) load_y(itn);
        if constexpr (NR == 2 && K % 2 == 0) {
            // rows k and k + 1 sit in the ring's two buffers: both multiply-accumulates, then both inverse transforms side by side
#pragma unroll
            for (int k = 0; k < K; k += 2) {
                int64_t acc[4] = {0, 0, 0, 0}, acd[4] = {0, 0, 0, 0};
                mac_row<L>(acc, Ar[0], yl, lane);
                mac_row<L>(acd, Ar[1], yl, lane);
                if (k + 2 < K) {
                    Ar[0].load(Ait + (size_t)(k + 2) * L * PD, lane, !shared_A && km.S == 1);
                    Ar[1].load(Ait + (size_t)(k + 3) * L * PD, lane, !shared_A && km.S == 1);
                }
                int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
                int32_t rd[4] = {mont_red64(acd[0]), mont_red64(acd[1]), mont_red64(acd[2]), mont_red64(acd[3])};
                DIL_SCHED_FENCE();
                ntt_inv_core2(r, rd, twi, lm);
                DIL_SCHED_FENCE();
                emit_matvec_row<LEVEL, OUT>(w_out, w1_out, w0_out, (it * K + k) * 256, r, sc, lane, nullptr);
                emit_matvec_row<LEVEL, OUT>(w_out, w1_out, w0_out, (it * K + k + 1) * 256, rd, sc, lane, nullptr);
            }
            continue;
        }
        constexpr int ROW_UNROLL = NR > 1 ? K : 1;           // the ring's slot index must be static
#pragma unroll ROW_UNROLL
        for (int k = 0; k < K; k++) {
            const int slot = NR > 1 ? k % NR : 0;
            int64_t acc[4] = {0, 0, 0, 0};
            mac_row<L>(acc, Ar[slot], yl, lane);
            if (k + NR < K) Ar[slot].load(Ait + (size_t)(k + NR) * L * PD, lane, !shared_A && km.S == 1);
            int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
            DIL_SCHED_FENCE();
            ntt_inv_core(r, twi, lm);
            DIL_SCHED_FENCE();
            emit_matvec_row<LEVEL, OUT>(w_out, w1_out, w0_out, (it * K + k) * 256, r, sc, lane, nullptr);
        }
    }
}

// A row of 256 BITS-wide fields, given in an LDS buffer in natural order, as the reference's packed byte stream
// (encoder.v:96-133): 8 BITS dwords, one per lane and pass; dword d takes bits [32 d, 32 d + 32) of the stream from the
// (at most 5) fields that overlap it.  dst must be 4-byte aligned.
template <int BITS>
__device__ __forceinline__ void pack_fields_from_lds(uint8_t* __restrict__ dst, const uint32_t* vals, int lane)
{
    constexpr int NDW = 8 * BITS, TERMS = (BITS - 1 + 32 + BITS - 1) / BITS;
#pragma unroll
    for (int pass = 0; pass < (NDW + 63) / 64; pass++) {
        const int d = 64 * pass + lane;
        if (d < NDW) {
            const uint32_t bit0 = 32u * (uint32_t)d, i0 = bit0 / BITS, o = bit0 - i0 * BITS;
            uint64_t acc = 0;
#pragma unroll
            for (int t = 0; t < TERMS; t++) acc |= (uint64_t)vals[min(i0 + t, 255u)] << (BITS * t);
            reinterpret_cast<uint32_t*>(dst)[d] = (uint32_t)(acc >> o);
        }
    }
}

// Key generation's t = A s1 + s2 with its output stage fused in (combined_top.v keygen :921-1079): wave per key, as
// matvec_wpi_kernel, then per row  t = w + s2,  (t1, t0) = Power2Round(t, 13),  t1 packed (10 bits) straight into the
// public key and 2^12 - t0 packed (13 bits) straight into the secret key -- no int32 w / t1 / t0 in HBM, no power2round /
// pack launches on the critical path to tr = H(pk).  s1, s2: ExpandS output, canonical.
template <int LEVEL, int AF>
__global__ __launch_bounds__(256) void keygen_wpi_kernel(
    uint8_t* __restrict__ pk, size_t pk_stride, uint8_t* __restrict__ sk, size_t sk_stride, size_t sk_t0_offset,
    const int32_t* __restrict__ A, const int32_t* __restrict__ s1, const int32_t* __restrict__ s2, size_t batch,
    const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L, PD = ARow<L, AF>::PD;
    using XP = X10Pick<true>;
    static_assert(XP::DW >= 256, "the exchange buffer doubles as the packing buffer");
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * TW_TABLE_DWORDS + 4 * L * 256 + 4 * XP::DW];
    const int lane = threadIdx.x & 63, wv = wave_in_block();
    stage_tables(lds, fwd_tab, inv_tab);
    const TwLds twf{lds, lane}, twi{lds + TW_TABLE_DWORDS, lane};
    uint32_t* xb = lds + 2 * TW_TABLE_DWORDS + 4 * L * 256 + wv * XP::DW;
    const typename XP::type lm(xb, lane);
    uint32_t* yl = lds + 2 * TW_TABLE_DWORDS + wv * (L * 256);
    const size_t nwaves = (size_t)gridDim.x * 4;
    size_t it = (size_t)blockIdx.x * 4 + wv;
    RawPolys<L> yr;
    if (it < batch) yr.load(s1 + it * L * 256, lane);
    __syncthreads();                               // tables staged (the only barrier)
    for (; it < batch; it += nwaves) {
        const int32_t* Ait = A + it * (size_t)(K * L) * PD;
        const int32_t* s2it = s2 + it * (size_t)K * 256;
        constexpr int NR = DIL_KG_ROWS(K);
        ARow<L, AF> Ar[NR];
        int32_t e[NR][4];
#pragma unroll
        for (int j = 0; j < NR; j++) {
            Ar[j].load(Ait + (size_t)j * L * PD, lane, true);
            load_strided(e[j], s2it + j * 256, lane);
        }
#pragma unroll
        for (int l = 0; l < L; l++) {
            ntt_fwd_core(yr.v[l], twf, lm);
            *reinterpret_cast<int4*>(yl + l * 256 + 4 * lane) = make_int4(yr.v[l][0], yr.v[l][1], yr.v[l][2], yr.v[l][3]);
        }
        DIL_SCHED_FENCE();
        const size_t itn = it + nwaves;
        if (itn < batch) yr.load(s1 + itn * L * 256, lane);
        constexpr int ROW_UNROLL = NR > 1 ? K : 1;           // the ring's slot index must be static
#pragma unroll ROW_UNROLL
        for (int k = 0; k < K; k++) {
            const int slot = NR > 1 ? k % NR : 0;
            int64_t acc[4] = {0, 0, 0, 0};
            mac_row<L>(acc, Ar[slot], yl, lane);
            int32_t e2[4] = {e[slot][0], e[slot][1], e[slot][2], e[slot][3]};
            if (k + NR < K) {
                Ar[slot].load(Ait + (size_t)(k + NR) * L * PD, lane, true);
                load_strided(e[slot], s2it + (k + NR) * 256, lane);
            }
            int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
            DIL_SCHED_FENCE();
            ntt_inv_core(r, twi, lm);
            DIL_SCHED_FENCE();
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t t = canon_small((int32_t)canon_small(r[m]) + e2[m] - Q);     // w + s2 mod q, both canonical
                hi[m] = (t + (1u << 12) - 1) >> 13;
                lo[m] = (1u << 12) - (t - (hi[m] << 13));                                    // 2^12 - t0, 13 bits
            }
#pragma unroll
            for (int m = 0; m < 4; m++) xb[lane + 64 * m] = hi[m];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            pack_fields_from_lds<10>(pk + it * pk_stride + 32 + (size_t)k * 320, xb, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int m = 0; m < 4; m++) xb[lane + 64 * m] = lo[m];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            pack_fields_from_lds<13>(sk + it * sk_stride + sk_t0_offset + (size_t)k * 416, xb, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
    }
}

// waves per SIMD the register allocator aims for: 4 at level 2 (118 VGPRs), 3 at levels 3 / 5 (138 / 168 VGPRs; forcing 4
// there was measured slower in both rounds -- the HBM stream is throughput-limited, more waves only add pressure)
#define DIL_VW_WAVES(LEVEL) ((LEVEL) == 2 ? 4 : 3)
// exchange policy of the VALU-bound kernels: all three exchanges of a transform through a 1-KiB per-wave LDS buffer (ntt_core.hpp XAllLds)
struct S2X {
    using type = XAllLds;
    static constexpr int DW = 256;
};
template <int LEVEL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DIL_VW_WAVES(LEVEL), DIL_VW_WAVES(LEVEL)))) void verify_wpi_kernel(
    uint8_t* __restrict__ w1_out, const int32_t* __restrict__ A, const int32_t* __restrict__ z,
    const int32_t* __restrict__ c, const int32_t* __restrict__ t1, const uint8_t* __restrict__ h, size_t batch,
    const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    using XP = X10Pick<true>;
    using PT = PipeTables<true>;
    __shared__ __attribute__((aligned(16))) uint32_t lds[PT::DWORDS + 4 * L * 256 + 4 * 64 + 4 * XP::DW];
    const int lane = threadIdx.x & 63, wv = wave_in_block();
    PT::stage(lds, fwd_tab, inv_tab);
    const typename PT::Fwd twf = PT::fwd(lds, fwd_tab, lane);
    const typename PT::Inv twi = PT::inv(lds, inv_tab, lane);
    const typename XP::type lm(lds + PT::DWORDS + 4 * L * 256 + 4 * 64 + wv * XP::DW, lane);
    uint32_t* sc = lds + PT::DWORDS + 4 * L * 256 + wv * 64;   // byte-plane scratch
    uint32_t* zl = lds + PT::DWORDS + wv * (L * 256);   // this wave's private slice
    const size_t nwaves = (size_t)gridDim.x * 4;
    size_t it = (size_t)blockIdx.x * 4 + wv;
    int32_t zc[L + 1][4];               // z[0 .. L-1] and c: verify keeps the default cache policy for its time-domain inputs
    auto load_zc = [&](size_t i) {
#pragma unroll
        for (int l = 0; l < L; l++) load_strided<false>(zc[l], z + (i * L + l) * 256, lane);
        load_strided<false>(zc[L], c + i * 256, lane);
    };
    if (it < batch) load_zc(it);
    __syncthreads();                               // tables staged (the only barrier)
    for (; it < batch; it += nwaves) {
        const int32_t* Ait = A + it * (size_t)(K * L) * 256;       // a key per item (one key for the batch: verify_shared_kernel)
        const int32_t* t1it = t1 + it * (size_t)K * 256;
        const uint8_t* hit = h + it * K * 256;
        // row 0 operands fly under the z-phase
        ARow<L> Ar;
        Ar.load(Ait, lane, true);
        int32_t tn[4];
        uint32_t hn;
        load_strided<false>(tn, t1it, lane);
        hn = load_row_u8(hit, lane);
        // z-phase
        ntt_fwd_coreN<L + 1>(zc, twf, lm);
#pragma unroll
        for (int l = 0; l < L; l++) *reinterpret_cast<int4*>(zl + l * 256 + 4 * lane) = make_int4(zc[l][0], zc[l][1], zc[l][2], zc[l][3]);
        int32_t ch[4] = {zc[L][0], zc[L][1], zc[L][2], zc[L][3]};
        DIL_SCHED_FENCE();
        // next item's time-domain inputs: a whole row phase to land
        const size_t itn = it + nwaves;
        if (itn < batch) load_zc(itn);
        // row k's INTT runs beside row k + 1's NTT(t1 2^13): th is always one row ahead
        int32_t th[4];
#pragma unroll
        for (int m = 0; m < 4; m++) th[m] = (tn[m] & 0x3FF) << 13;   // decoder.v:96-100
        load_strided<false>(tn, t1it + 256, lane);
        ntt_fwd_core(th, twf, lm);
        for (int k = 0; k < K; k++) {
            int64_t acc[4] = {0, 0, 0, 0};
            mac_row<L>(acc, Ar, zl, lane);
            uint32_t hb[4];
            unpack_row_u8(hb, hn, sc, lane);
#pragma unroll
            for (int m = 0; m < 4; m++) acc[m] -= (int64_t)ch[m] * th[m];
            int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
            if (k + 1 < K) {
                Ar.load(Ait + (size_t)(k + 1) * L * 256, lane, true);
                hn = load_row_u8(hit + (k + 1) * 256, lane);
#pragma unroll
                for (int m = 0; m < 4; m++) th[m] = (tn[m] & 0x3FF) << 13;
                if (k + 2 < K) load_strided<false>(tn, t1it + (k + 2) * 256, lane);
                DIL_SCHED_FENCE();
                ntt_fwd_inv_pair(th, r, twf, twi, lm);
            } else {
                DIL_SCHED_FENCE();
                ntt_inv_core(r, twi, lm);
            }
            DIL_SCHED_FENCE();
            const size_t o = (it * K + k) * 256;
            uint32_t wb[4];
#pragma unroll
            for (int m = 0; m < 4; m++) wb[m] = use_hint<LEVEL>(canon_small(r[m]), hb[m]);
            store_row_u8(w1_out + o, wb, sc, lane);
        }
    }
}

// Phase 2 is VALU-bound (84 % VALU-busy at 7 waves per SIMD, profiles/r03a_sign_pmc.txt) and uses almost no LDS: all three
// exchanges of its transforms go through a 1-KiB per-wave LDS buffer (ntt_core.hpp XAllLds) instead of permlane / DPP / v_bfi --
// measured -10 % at every level (level 5, one key, 8192 attempts: 78.2 -> 69.5 us; profiles/r03l_ab_s2.txt).  The gain needs
// occupancy: at 2-3 waves per SIMD the LDS round trips are exposed and the register form wins (profiles/r03c_tune_xchg.txt).
// Waves per SIMD.  Round 3 held these kernels at five (<= 96 VGPRs: the per-item-key form of the paired rows wanted 105-113 registers
// and lost more to exposed latency at four waves than the saved transforms gave back).  With two transforms side by side per row
// (ntt_inv_core2) a wave hides its own latency and wants the registers instead: four waves (<= 128 VGPRs, no spills) beat five
// and six at every level and key form -- level 5, 8192 attempts, one key: 53.2 us at four, 55.3 at five, 53.8 at six; a key per
// attempt: 64.1 / 64.9 / 76.4 (spills) -- profiles/r04e_ab_s2_waves_prefetch.txt.  A row-ahead prefetch of w0 / y / w1 does not pay
// (vmcnt retires in order: waiting for the row's key operand then waits for the prefetch too).
#define DIL_S2_ATTR __attribute__((amdgpu_waves_per_eu(4)))
template <int LEVEL, int YF, bool SH, bool SMALL>
__global__ __launch_bounds__(256) DIL_S2_ATTR void sign2_wpi_kernel(
    int32_t* __restrict__ z_out, uint8_t* __restrict__ h_out, int32_t* __restrict__ flags_out,
    const int32_t* __restrict__ c, const int32_t* __restrict__ y, const int32_t* __restrict__ w0,
    const uint8_t* __restrict__ w1, const int32_t* __restrict__ s1hat, const int32_t* __restrict__ s2hat,
    const int32_t* __restrict__ t0hat, size_t batch, int shared_key, KeyMap km, const uint32_t* __restrict__ fwd_tab,
    const uint32_t* __restrict__ inv_tab)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    constexpr bool STAGED = SH && SMALL;        // the paired key rows live in LDS
    using XP = S2X;
    using PT = PipeTables<true>;
    constexpr int PAIR_AT = PT::DWORDS + 4 * 64 + 4 * XP::DW;
    __shared__ __attribute__((aligned(16))) uint32_t lds[PAIR_AT + (STAGED ? L * 256 : 0)];
    const int lane = threadIdx.x & 63, wv = wave_in_block();
    PT::stage(lds, fwd_tab, inv_tab);
    if (STAGED) SmallPair::stage_key<L>(reinterpret_cast<int32_t*>(lds + PAIR_AT), s1hat, s2hat);
    __syncthreads();
    const int32_t* s12 = reinterpret_cast<const int32_t*>(lds + PAIR_AT);
    const typename PT::Fwd twf = PT::fwd(lds, fwd_tab, lane);
    const typename PT::Inv twi = PT::inv(lds, inv_tab, lane);
    const typename XP::type lm(lds + PT::DWORDS + 4 * 64 + wv * XP::DW, lane);
    uint32_t* sc = lds + PT::DWORDS + wv * 64;   // byte-plane scratch
    const YSrc<LEVEL, YF> ys(lane);
    const size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t it = (size_t)blockIdx.x * 4 + wv; it < batch; it += nwaves) {
        const int32_t* s1 = s1hat + (shared_key ? 0 : km.key(it) * L) * 256;
        const int32_t* s2 = s2hat + (shared_key ? 0 : km.key(it) * K) * 256;
        const int32_t* t0 = t0hat + (shared_key ? 0 : km.key(it) * K) * 256;
        int32_t ch[4], cp[4];
        load_strided(ch, c + it * 256, lane);
        int4 n1 = make_int4(0, 0, 0, 0), n2 = n1;
        if (!STAGED || L < K) n2 = *reinterpret_cast<const int4*>(s2 + (STAGED ? L : 0) * 256 + 4 * lane);   // STAGED: s2's own rows matter from row L on
        if (!STAGED) n1 = *reinterpret_cast<const int4*>(s1 + 4 * lane);
        ntt_fwd_core(ch, twf, lm);
        if (SMALL && !SH) {
#pragma unroll
            for (int m = 0; m < 4; m++) cp[m] = mont_mul(ch[m], SmallPair::SHIFT_R);      // c^ * 2^11
        }
        uint32_t bits = 0, nh = 0;
        // Row k (SMALL): c s1[k] and c s2[k] from ONE inverse transform (SmallPair: both products are tiny, so c^ o (s1^ + 2^11 s2^)
        // carries them side by side in one residue), c t0[k] from a second -- 1 + 2 K transforms per attempt instead of 1 + L + 2 K.
        for (int k = 0; k < K; k++) {
            const bool zrow = k < L, pair = SMALL && zrow;
            const int4 a1 = n1, a2 = n2, b0 = *reinterpret_cast<const int4*>(t0 + k * 256 + 4 * lane);
            const size_t o = (it * K + k) * 256, oz = (it * L + k) * 256;
            int32_t wv0[4], yv[4];
            uint32_t wv1[4], hv[4];
            load_strided(wv0, w0 + o, lane);
            const uint32_t w1p = w1 ? load_row_u8(w1 + o, lane) : 0u;         // (w1 == nullptr: the loop's W0W1 plane, w0 | w1 << 24)
            if (zrow) ys.raw(yv, y, it * L + k, lane);
            if (k + 1 < K) {
                if (!STAGED && k + 1 < L) n1 = *reinterpret_cast<const int4*>(s1 + (k + 1) * 256 + 4 * lane);
                if (!STAGED || k + 1 > L) n2 = *reinterpret_cast<const int4*>(s2 + (k + 1) * 256 + 4 * lane);
            }
            int32_t a[4];
            if (pair && SH) {
                const int4 p = *reinterpret_cast<const int4*>(s12 + k * 256 + 4 * lane);
                a[0] = mont_mul(ch[0], p.x); a[1] = mont_mul(ch[1], p.y); a[2] = mont_mul(ch[2], p.z); a[3] = mont_mul(ch[3], p.w);
            } else if (pair) {
                a[0] = SmallPair::mul(ch[0], a1.x, cp[0], a2.x); a[1] = SmallPair::mul(ch[1], a1.y, cp[1], a2.y);
                a[2] = SmallPair::mul(ch[2], a1.z, cp[2], a2.z); a[3] = SmallPair::mul(ch[3], a1.w, cp[3], a2.w);
            } else {
                a[0] = mont_mul(ch[0], a2.x); a[1] = mont_mul(ch[1], a2.y); a[2] = mont_mul(ch[2], a2.z); a[3] = mont_mul(ch[3], a2.w);
            }
            int32_t b[4] = {mont_mul(ch[0], b0.x), mont_mul(ch[1], b0.y), mont_mul(ch[2], b0.z), mont_mul(ch[3], b0.w)};
            ntt_inv_core2(a, b, twi, lm);           // the row's two transforms side by side (ntt_core.hpp)
            int32_t cs1[4] = {0, 0, 0, 0};
            if (!SMALL && zrow) {                   // generic: c s1[k] has a transform of its own
                cs1[0] = mont_mul(ch[0], a1.x); cs1[1] = mont_mul(ch[1], a1.y); cs1[2] = mont_mul(ch[2], a1.z); cs1[3] = mont_mul(ch[3], a1.w);
                ntt_inv_core(cs1, twi, lm);
            }
            bool rej0 = false, rej1 = false, rej2 = false;
            if (zrow) {
                ys.value(yv);
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    if (pair) SmallPair::split(a[m], cs1[m], a[m]);              // a[m] := c s2, exact
                    const uint32_t v = canon_pm2q(yv[m] + cs1[m]);
                    rej0 |= norm_reject(v, Par<LEVEL>::GAMMA1 - Par<LEVEL>::BETA);
                    st_nt(z_out + oz + lane + 64 * m, (int32_t)v);
                }
            }
            if (SMALL && !zrow) {                   // a lone s2 row: its product is as small as the paired ones
#pragma unroll
                for (int m = 0; m < 4; m++) a[m] = small_exact(a[m]);
            }
            if (w1) {
                unpack_row_u8(wv1, w1p, sc, lane);
            } else {
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    wv1[m] = (uint32_t)wv0[m] >> 24;
                    wv0[m] &= 0xFFFFFF;
                }
            }
#pragma unroll
            for (int m = 0; m < 4; m++) {
                uint64_t hm;
                if (SMALL) {
                    const Phase2Coef<LEVEL> pc((uint32_t)wv0[m], a[m], b[m]);
                    rej1 |= pc.rej_r0();
                    rej2 |= pc.rej_ct0();
                    hm = __ballot(pc.hint(wv1[m]));
                } else {
                    const uint32_t ct0 = canon_small(b[m]);
                    const uint32_t r0 = canon_pm2q(wv0[m] - a[m]);
                    rej1 |= norm_reject(r0, Par<LEVEL>::GAMMA2 - Par<LEVEL>::BETA);
                    rej2 |= norm_reject(ct0, Par<LEVEL>::GAMMA2);
                    hm = __ballot(make_hint_p<LEVEL>(canon_2q(r0 + ct0), wv1[m]));
                }
                hv[m] = mask_to_01(hm);
                nh += __popcll(hm);
            }
            store_row_u8(h_out + o, hv, sc, lane);
            if (__ballot(rej0)) bits |= 1;
            if (__ballot(rej1)) bits |= 2;
            if (__ballot(rej2)) bits |= 4;
        }
        if (lane == 0) flags_out[it] = (int32_t)(bits | (nh > (uint32_t)Par<LEVEL>::OMEGA ? 8u : 0u));
    }
}

// Sign phase 2 for the signing LOOP (dil_sign_dev): same arithmetic, but an attempt is abandoned at its first failed
// check; flags reports only that first failure (2: an r0 row, 1: a z row, 4: a c t0 row, | 8 for too many hints); z and h are
// complete only when flags == 0, which is all the loop reads.  r0 is parked in the attempt's own w0 scratch until (C).
//   one key (SH):   rows k = 0 .. K-1 in turn: r0[k] = w0[k] - c s2[k], then z[k] = y[k] + c s1[k] for k < L -- one inverse
//                   transform per row gives both (SmallPair, the key's paired rows staged in LDS) -- then (C) c t0 and the hints.
//                   (w0 / y of row k + 1 loaded under row k: 1259 -> 1274 us per 8192 level-3 signatures, 5528 -> 5569 per 65536 -- not kept.)
//   a key per item: (A) all r0 rows (61 % of level-5 attempts fail one), (B) all z rows (34 %), (C): every key row is HBM traffic
//                   of its own there, so the rows that reject most go first and nothing is paired.
// Expected inverse transforms per level-5 attempt: ~6 (one key) / ~10 (a key per item) instead of 16 / 23 for the full kernels.
template <int LEVEL, int YF, bool SH>
__global__ __launch_bounds__(256) DIL_S2_ATTR void sign2_early_wpi_kernel(
    int32_t* __restrict__ z_out, uint8_t* __restrict__ h_out, int32_t* __restrict__ flags_out,
    const int32_t* __restrict__ c, const int32_t* __restrict__ y, int32_t* __restrict__ w0,
    const uint8_t* __restrict__ w1, const int32_t* __restrict__ s1hat, const int32_t* __restrict__ s2hat,
    const int32_t* __restrict__ t0hat, size_t batch, int shared_key, KeyMap km, const uint32_t* __restrict__ fwd_tab,
    const uint32_t* __restrict__ inv_tab)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    using XP = S2X;
    using PT = PipeTables<true>;
    constexpr int PAIR_AT = PT::DWORDS + 4 * 64 + 4 * XP::DW;
    __shared__ __attribute__((aligned(16))) uint32_t lds[PAIR_AT + (SH ? L * 256 : 0)];
    const int lane = threadIdx.x & 63, wv = wave_in_block();
    PT::stage(lds, fwd_tab, inv_tab);
    if (SH) SmallPair::stage_key<L>(reinterpret_cast<int32_t*>(lds + PAIR_AT), s1hat, s2hat);
    __syncthreads();
    const int32_t* s12 = reinterpret_cast<const int32_t*>(lds + PAIR_AT);
    const uint32_t W0MASK = w1 ? 0xFFFFFFFFu : 0xFFFFFFu;      // w1 == nullptr: w0 is the loop's W0W1 plane (w0 | w1 << 24, emit_matvec_row)
    const typename PT::Fwd twf = PT::fwd(lds, fwd_tab, lane);
    const typename PT::Inv twi = PT::inv(lds, inv_tab, lane);
    const typename XP::type lm(lds + PT::DWORDS + 4 * 64 + wv * XP::DW, lane);
    uint32_t* sc = lds + PT::DWORDS + wv * 64;   // byte-plane scratch
    const YSrc<LEVEL, YF> ys(lane);
    const size_t nwaves = (size_t)gridDim.x * 4;
    // entries in turn: by stride, or (signing loop) from this workgroup's queue -- the next ticket is drawn at the top of an
    // entry and needed at its end
    const uint32_t parts = gridDim.x < (uint32_t)TICKET_PARTS ? gridDim.x : (uint32_t)TICKET_PARTS, part = blockIdx.x % parts;
    uint32_t* const queue = km.ticket ? km.ticket + part * TICKET_STRIDE : nullptr;
    auto draw = [&]() -> size_t {
        uint32_t tk = 0;
        if (lane == 0) tk = atomicAdd(queue, 1u);
        return part + (size_t)parts * __builtin_amdgcn_readfirstlane(tk);
    };
    size_t un = 0;
    for (size_t u = queue ? draw() : (size_t)blockIdx.x * 4 + wv; u < batch; u = un) {
        un = queue ? draw() : u + nwaves;
        // A speculative round (km.spec_n pending items x km.S attempts each, entry = item * S + attempt) is walked attempt-major:
        // by the time the waves reach attempt a of an item, its earlier attempts have mostly reported, and an entry behind an
        // accepted one is dropped unread -- the loop only ever takes an item's FIRST accepted attempt, so the signatures do not
        // depend on what is dropped (a stale or late flag merely costs the work).  Flags are preset to -1 by the round's set-up.
        size_t it = u;
        int32_t earlier = -1;
        if (km.spec_n) {
            const uint32_t at = (uint32_t)u / km.spec_n, j = (uint32_t)u - at * km.spec_n;
            it = (size_t)j * km.S + at;
            if (lane < (int)at) earlier = __hip_atomic_load(flags_out + (size_t)j * km.S + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int32_t* s1 = s1hat + (shared_key ? 0 : km.key(it) * L) * 256;
        const int32_t* s2 = s2hat + (shared_key ? 0 : km.key(it) * K) * 256;
        const int32_t* t0 = t0hat + (shared_key ? 0 : km.key(it) * K) * 256;
        int32_t ch[4];
        load_strided(ch, c + it * 256, lane);
        int4 kn = make_int4(0, 0, 0, 0);
        if (__ballot(earlier == 0)) {
            if (lane == 0) __hip_atomic_store(flags_out + it, (int32_t)FLAG_SUPERSEDED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        if (SH && L < K) kn = *reinterpret_cast<const int4*>(s2 + L * 256 + 4 * lane);         // one key: s2's own rows matter from row L on
        if (!SH) kn = *reinterpret_cast<const int4*>(s2 + 4 * lane);
        ntt_fwd_core(ch, twf, lm);
        uint32_t bits = 0, nh = 0;
        if constexpr (SH) {
        // (A + B) row k: r0[k] = w0[k] - c s2[k] and, for k < L, z[k] = y[k] + c s1[k] -- both products from ONE inverse transform
        // (SmallPair), r0 checked before z
#pragma unroll 1
        for (int k = 0; k < K; k++) {
            const bool pair = k < L;
            const int4 a2 = kn;
            const size_t o = (it * K + k) * 256, oz = (it * L + k) * 256;
            int32_t wv0[4], yv[4];
            load_strided(wv0, w0 + o, lane);
            if (pair) ys.raw(yv, y, it * L + k, lane);
            if (k + 1 < K && k + 1 > L) kn = *reinterpret_cast<const int4*>(s2 + (k + 1) * 256 + 4 * lane);
            int32_t a[4];
            if (pair) {
                const int4 p = *reinterpret_cast<const int4*>(s12 + k * 256 + 4 * lane);
                a[0] = mont_mul(ch[0], p.x); a[1] = mont_mul(ch[1], p.y); a[2] = mont_mul(ch[2], p.z); a[3] = mont_mul(ch[3], p.w);
            } else {
                a[0] = mont_mul(ch[0], a2.x); a[1] = mont_mul(ch[1], a2.y); a[2] = mont_mul(ch[2], a2.z); a[3] = mont_mul(ch[3], a2.w);
            }
            ntt_inv_core(a, twi, lm);
            bool rej = false, rejz = false;
            if (pair) {
                ys.value(yv);
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    int32_t cs1;
                    SmallPair::split(a[m], cs1, a[m]);
                    const uint32_t v = canon_pm2q(yv[m] + cs1);
                    rejz |= norm_reject(v, Par<LEVEL>::GAMMA1 - Par<LEVEL>::BETA);
                    z_out[oz + lane + 64 * m] = (int32_t)v;
                }
            }
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t r0 = canon_pm2q((int32_t)((uint32_t)wv0[m] & W0MASK) - a[m]);
                rej |= norm_reject(r0, Par<LEVEL>::GAMMA2 - Par<LEVEL>::BETA);
                w0[o + lane + 64 * m] = (int32_t)(r0 | ((uint32_t)wv0[m] & ~W0MASK));      // (the W0W1 plane keeps w1 in its top byte)
            }
            if (__ballot(rej)) {
                bits = 2;
                break;
            }
            if (__ballot(rejz)) {
                bits = 1;
                break;
            }
        }
        } else {
        // a key per item: every row of s1^ / s2^ is HBM traffic of its own, so the rows that reject most go first and nothing is paired
        // (A) r0 rows
#pragma unroll 1
        for (int k = 0; k < K; k++) {
            const int4 a2 = kn;
            const size_t o = (it * K + k) * 256;
            int32_t wv0[4];
            load_strided(wv0, w0 + o, lane);
            if (k + 1 < K) kn = *reinterpret_cast<const int4*>(s2 + (k + 1) * 256 + 4 * lane);
            int32_t a[4] = {mont_mul(ch[0], a2.x), mont_mul(ch[1], a2.y), mont_mul(ch[2], a2.z), mont_mul(ch[3], a2.w)};
            ntt_inv_core(a, twi, lm);
            bool rej = false;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t r0 = canon_pm2q((int32_t)((uint32_t)wv0[m] & W0MASK) - a[m]);
                rej |= norm_reject(r0, Par<LEVEL>::GAMMA2 - Par<LEVEL>::BETA);
                w0[o + lane + 64 * m] = (int32_t)(r0 | ((uint32_t)wv0[m] & ~W0MASK));      // (the W0W1 plane keeps w1 in its top byte)
            }
            if (__ballot(rej)) {
                bits = 2;
                break;
            }
        }
        // (B) z rows
        if (!bits) {
            kn = *reinterpret_cast<const int4*>(s1 + 4 * lane);
#pragma unroll 1
            for (int l = 0; l < L; l++) {
                const int4 sv = kn;
                const size_t o = (it * L + l) * 256;
                int32_t yv[4];
                ys.raw(yv, y, it * L + l, lane);
                if (l + 1 < L) kn = *reinterpret_cast<const int4*>(s1 + (l + 1) * 256 + 4 * lane);
                int32_t r[4] = {mont_mul(ch[0], sv.x), mont_mul(ch[1], sv.y), mont_mul(ch[2], sv.z), mont_mul(ch[3], sv.w)};
                ntt_inv_core(r, twi, lm);
                ys.value(yv);
                bool rej = false;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const uint32_t v = canon_pm2q(r[m] + yv[m]);
                    rej |= norm_reject(v, Par<LEVEL>::GAMMA1 - Par<LEVEL>::BETA);
                    z_out[o + lane + 64 * m] = (int32_t)v;
                }
                if (__ballot(rej)) {
                    bits = 1;
                    break;
                }
            }
        }
        }
        // (C) c t0 rows and the hints
        if (!bits) {
            kn = *reinterpret_cast<const int4*>(t0 + 4 * lane);
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int4 b0 = kn;
                const size_t o = (it * K + k) * 256;
                int32_t r0v[4];
                uint32_t wv1[4], hv[4];
                load_strided(r0v, w0 + o, lane);          // this lane's own stores of stage (A)
                const uint32_t w1p = w1 ? load_row_u8(w1 + o, lane) : 0u;
                if (k + 1 < K) kn = *reinterpret_cast<const int4*>(t0 + (k + 1) * 256 + 4 * lane);
                int32_t b[4] = {mont_mul(ch[0], b0.x), mont_mul(ch[1], b0.y), mont_mul(ch[2], b0.z), mont_mul(ch[3], b0.w)};
                ntt_inv_core(b, twi, lm);
                bool rej = false;
                if (w1) {
                    unpack_row_u8(wv1, w1p, sc, lane);
                } else {
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        wv1[m] = (uint32_t)r0v[m] >> 24;
                        r0v[m] &= 0xFFFFFF;
                    }
                }
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const uint32_t ct0 = canon_small(b[m]);
                    rej |= norm_reject(ct0, Par<LEVEL>::GAMMA2);
                    const uint64_t hm = __ballot(make_hint_p<LEVEL>(canon_2q((uint32_t)r0v[m] + ct0), wv1[m]));
                    hv[m] = mask_to_01(hm);
                    nh += __popcll(hm);
                }
                store_row_u8(h_out + o, hv, sc, lane);
                if (__ballot(rej)) {
                    bits = 4;
                    break;
                }
            }
        }
        // (agent scope: the waves that look at this flag before starting a later attempt of the same item may sit on another XCD)
        if (lane == 0)
            __hip_atomic_store(flags_out + it, (int32_t)(bits | (nh > (uint32_t)Par<LEVEL>::OMEGA ? 8u : 0u)), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
}

// (twiddles in registers instead of LDS were tried in the shared-key kernels: no gain, the verify variant spilled)
// ---------------------------------------------------------------------------------------
// Shared-key wave-per-item kernels.  When one key serves the whole batch (one signer, or many
// signatures under one public key) its NTT-domain material -- A [K][L] (+ t1^ for verify; s1^,
// s2^, t0^ for sign phase 2) -- is staged ONCE per persistent workgroup into LDS (30-64 KiB)
// and every multiply-accumulate reads both operands from LDS.  No per-item matrix traffic at
// all: measured, re-reading an L2-resident A per item cost as much as streaming it from HBM.
// Workgroups are as large as LDS allows (NW waves, 1 workgroup per CU).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_polys(uint32_t* lds_dst, const int32_t* __restrict__ src, int npolys)
{
    for (int i = threadIdx.x; i < npolys * 64; i += blockDim.x)
        reinterpret_cast<uint4*>(lds_dst)[i] = reinterpret_cast<const uint4*>(src)[i];
}

template <int L>
__device__ __forceinline__ void mac_row_lds(int64_t (&acc)[4], const uint32_t* a_row, const uint32_t* vec, int lane)
{
#pragma unroll
    for (int l = 0; l < L; l++) {
        const int4 a = *reinterpret_cast<const int4*>(a_row + l * 256 + 4 * lane);
        const int4 z = *reinterpret_cast<const int4*>(vec + l * 256 + 4 * lane);
        acc[0] += (int64_t)a.x * z.x;
        acc[1] += (int64_t)a.y * z.y;
        acc[2] += (int64_t)a.z * z.z;
        acc[3] += (int64_t)a.w * z.w;
    }
}

// y^ stays in REGISTERS here (4 L VGPRs): a lane multiplies the four coefficients it produced itself, so the vector needs no
// LDS slice -- that slice (L KiB per wave) was what capped the workgroup at 12 waves = 3 per SIMD at level 5, and three waves per
// SIMD issue no more than two (scripts/tune_xchg.hip, profiles/r03c_tune_xchg.txt: 785 / 785 / 689 cycles per transform at 2 / 3 /
// 4 waves).  Now 16 waves = 4 per SIMD at every level, no ds_write of y^ and half the ds_read_b128 of the multiply-accumulate.
// Level 5: nothing is prefetched.  Three of the seven polynomials would fit beside seven transforms in flight under 128 VGPRs and
// make the kernel 2 % faster on its own (45.5 vs 46.1 us) -- but with them bench.py's attempt (phase 1, then phase 2, over ONE set of
// buffers: 223 MB, Infinity-Cache-resident) takes 134.7 instead of 103.9 us, while the same pair over two rotating sets (HBM-streaming)
// is unchanged (105.9 vs 106.6): profiles/r04q_ab_pf.txt, r04r_ab_pair.txt.  Unexplained; the cache-resident regime is the one the
// signing loop's narrow rounds run in, so the prefetch stays off where it was off.
#define DIL_MVS_PFN(L) ((L) <= 5 ? (L) : 0)
#define DIL_MVS_WGS 1          // 16-wave workgroups per CU the shared-key kernels are built for (8 waves per SIMD need <= 64 VGPRs)
template <int K, int L, int LEVEL, int OUT, int NW, int YF>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(DIL_MVS_WGS * NW / 4))) void matvec_shared_kernel(
    int32_t* __restrict__ w_out, uint8_t* __restrict__ w1_out, int32_t* __restrict__ w0_out,
    const int32_t* __restrict__ A, const int32_t* __restrict__ y, size_t batch,
    const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab)
{
    // All three exchanges through LDS (S2X = XAllLds), as in phase 2: with y^ in registers the LDS pipe serves only A and the
    // twiddles, and at 4 waves per SIMD the exchange-free transforms win 3-6 % (level 5 sign phase 1: 51.1 -> 48.0 us).  The (1:0)
    // exchange ALONE through LDS (one ds_write_b128 + four ds_read_b32) costs 20 % here: profiles/r03m_ab_x.txt.
    // Round 4: compact twiddle tables (5.25 KiB) and the byte-plane scratch aliased onto the wave's exchange buffer (both are
    // wave-private and used in turn): level 5 is 5.25 + 56 + 16 = 77.25 KiB per workgroup, so TWO 16-wave workgroups share a CU.
    using XP = S2X;
    using PT = PipeTables<true>;
    __shared__ __attribute__((aligned(16))) uint32_t lds[PT::DWORDS + K * L * 256 + NW * XP::DW];
    const int lane = threadIdx.x & 63, wv = wave_in_block();
    PT::stage(lds, fwd_tab, inv_tab);
    uint32_t* Al = lds + PT::DWORDS;
    stage_polys(Al, A, K * L);
    const typename PT::Fwd twf = PT::fwd(lds, fwd_tab, lane);
    const typename PT::Inv twi = PT::inv(lds, inv_tab, lane);
    uint32_t* xb = Al + K * L * 256 + wv * XP::DW;
    const typename XP::type lm(xb, lane);
    uint32_t* sc = xb;                            // byte-plane scratch (64 dwords) = the head of the exchange buffer
    const size_t nwaves = (size_t)gridDim.x * NW;
    size_t it = (size_t)blockIdx.x * NW + wv;
    const YSrc<LEVEL, YF> ys(lane);
    // the next item's y is prefetched a whole row phase ahead where the registers allow it; at level 5 that second copy (28 VGPRs)
    // does not fit beside seven transforms in flight under the 128-register cap: y is loaded where it is used
    constexpr int PF = DIL_MVS_PFN(L);             // polynomials of the NEXT item's y that are prefetched a whole row phase ahead
    RawPolys<PF ? PF : 1> yr;
    auto load_y = [&](size_t i) {
#pragma unroll
        for (int l = 0; l < PF; l++) ys.raw(yr.v[l], y, i * L + l, lane);
    };
    if (it < batch) load_y(it);
    __syncthreads();                               // tables + key staged (the only barrier)
    for (; it < batch; it += nwaves) {
        int32_t yh[L][4];
#pragma unroll
        for (int l = PF; l < L; l++) ys.raw(yh[l], y, it * L + l, lane);        // the rest is fetched now and lands under the first group's transforms
#pragma unroll
        for (int l = 0; l < PF; l++) {
#pragma unroll
            for (int m = 0; m < 4; m++) yh[l][m] = yr.v[l][m];
        }
        if constexpr (PF > 0 && PF < L) {
            // two groups: the prefetched polynomials are transformed while the others' loads are in flight
            int32_t (&ga)[PF][4] = reinterpret_cast<int32_t (&)[PF][4]>(yh[0]);
            int32_t (&gb)[L - PF][4] = reinterpret_cast<int32_t (&)[L - PF][4]>(yh[PF]);
#pragma unroll
            for (int l = 0; l < PF; l++) ys.value(yh[l]);
            ntt_fwd_coreN<PF>(ga, twf, lm);
#pragma unroll
            for (int l = PF; l < L; l++) ys.value(yh[l]);
            ntt_fwd_coreN<L - PF>(gb, twf, lm);
        } else {
#pragma unroll
            for (int l = 0; l < L; l++) ys.value(yh[l]);
            ntt_fwd_coreN<L>(yh, twf, lm);
        }
        DIL_SCHED_FENCE();
        const size_t itn = it + nwaves;
        if (itn < batch) load_y(itn);
        static_assert(K % 2 == 0, "rows are processed in pairs");
        for (int k = 0; k < K; k += 2) {
            int64_t acc[4] = {0, 0, 0, 0}, acd[4] = {0, 0, 0, 0};
#pragma unroll
            for (int l = 0; l < L; l++) {
                const int4 a = (*reinterpret_cast<const int4*>(Al + (k * L + l) * 256 + 4 * lane));
                const int4 d = (*reinterpret_cast<const int4*>(Al + ((k + 1) * L + l) * 256 + 4 * lane));
                acc[0] += (int64_t)a.x * yh[l][0];
                acc[1] += (int64_t)a.y * yh[l][1];
                acc[2] += (int64_t)a.z * yh[l][2];
                acc[3] += (int64_t)a.w * yh[l][3];
                acd[0] += (int64_t)d.x * yh[l][0];
                acd[1] += (int64_t)d.y * yh[l][1];
                acd[2] += (int64_t)d.z * yh[l][2];
                acd[3] += (int64_t)d.w * yh[l][3];
            }
            int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
            int32_t rd[4] = {mont_red64(acd[0]), mont_red64(acd[1]), mont_red64(acd[2]), mont_red64(acd[3])};
            DIL_SCHED_FENCE();
            ntt_inv_core2(r, rd, twi, lm);
            DIL_SCHED_FENCE();
            (emit_matvec_row<LEVEL, OUT>(w_out, w1_out, w0_out, (it * K + k) * 256, r, sc, lane, xb));
            (emit_matvec_row<LEVEL, OUT>(w_out, w1_out, w0_out, (it * K + k + 1) * 256, rd, sc, lane, xb));
        }
        if (false)
        for (int k = 0; k < K; k++) {
            int64_t acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int l = 0; l < L; l++) {
                const int4 a = (*reinterpret_cast<const int4*>(Al + (k * L + l) * 256 + 4 * lane));
                acc[0] += (int64_t)a.x * yh[l][0];
                acc[1] += (int64_t)a.y * yh[l][1];
                acc[2] += (int64_t)a.z * yh[l][2];
                acc[3] += (int64_t)a.w * yh[l][3];
            }
            int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
            DIL_SCHED_FENCE();
            ntt_inv_core(r, twi, lm);
            DIL_SCHED_FENCE();
            (emit_matvec_row<LEVEL, OUT>(w_out, w1_out, w0_out, (it * K + k) * 256, r, sc, lane, xb));
        }
    }
}

// verify, shared public key: A and t1^ = NTT(t1 2^13) live in LDS; t1^ is computed once per
// workgroup by its first K waves (VY_NTT_T1, combined_top.v:1259) -- cheaper than a second launch.
// z^ stays in registers (a lane multiplies the coefficients it transformed itself), as y^ in matvec_shared_kernel: no per-wave
// LDS slice, 16 waves per workgroup at every level (level 5 was 11), and all three exchanges through LDS.
template <int LEVEL, int NW>
__global__ __launch_bounds__(64 * NW) void verify_shared_kernel(
    uint8_t* __restrict__ w1_out, const int32_t* __restrict__ A, const int32_t* __restrict__ z,
    const int32_t* __restrict__ c, const int32_t* __restrict__ t1, const uint8_t* __restrict__ h, size_t batch,
    const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    using XP = S2X;
    using PT = PipeTables<true>;
    __shared__ __attribute__((aligned(16))) uint32_t lds[PT::DWORDS + (K * L + K) * 256 + NW * 64 + NW * XP::DW];
    const int lane = threadIdx.x & 63, wv = wave_in_block();
    PT::stage(lds, fwd_tab, inv_tab);
    uint32_t* Al = lds + PT::DWORDS;
    uint32_t* Tl = Al + K * L * 256;
    stage_polys(Al, A, K * L);
    const typename PT::Fwd twf = PT::fwd(lds, fwd_tab, lane);
    const typename PT::Inv twi = PT::inv(lds, inv_tab, lane);
    const typename XP::type lm(Tl + K * 256 + NW * 64 + wv * XP::DW, lane);
    uint32_t* sc = Tl + K * 256 + wv * 64;        // byte-plane scratch
    const size_t nwaves = (size_t)gridDim.x * NW;
    size_t it = (size_t)blockIdx.x * NW + wv;
    // the next item's z is prefetched a whole row phase ahead where the registers allow it (levels 2, 3); at level 5 that second
    // copy (28 VGPRs) would spill under the 128-register cap of a 16-wave workgroup, so z is loaded where it is used -- four
    // waves per SIMD cover the latency
    constexpr bool PFZ = L <= 5;
    RawPolys<PFZ ? L : 1, false> zr;
    int32_t cr[4] = {0, 0, 0, 0};
    if (it < batch) {
        if (PFZ) zr.load(z + it * L * 256, lane);
        load_strided<false>(cr, c + it * 256, lane);
    }
    __syncthreads();                               // tables + A staged
    for (int k = wv; k < K; k += NW) {             // t1_k * 2^13 (decoder.v:96-100) -> NTT -> LDS, lazy residues
        int32_t th[4];
#pragma unroll
        for (int m = 0; m < 4; m++) th[m] = (t1[k * 256 + lane + 64 * m] & 0x3FF) << 13;
        ntt_fwd_core(th, twf, lm);
        *reinterpret_cast<int4*>(Tl + k * 256 + 4 * lane) = make_int4(th[0], th[1], th[2], th[3]);
    }
    __syncthreads();
    for (; it < batch; it += nwaves) {
        const uint8_t* hit = h + it * K * 256;
        uint32_t hn = load_row_u8(hit, lane);
        int32_t zh[L][4];
        if (!PFZ) {
#pragma unroll
            for (int l = 0; l < L; l++) load_strided<false>(zh[l], z + (it * L + l) * 256, lane);
        }
        if (PFZ) {
#pragma unroll
            for (int l = 0; l < L; l++) {
#pragma unroll
                for (int m = 0; m < 4; m++) zh[l][m] = zr.v[l][m];
            }
        }
        int32_t ch[4] = {cr[0], cr[1], cr[2], cr[3]};
        ntt_fwd_coreN<L>(zh, twf, lm);                  // the L forward transforms side by side; c joins the first inverse pair's slot
        ntt_fwd_core(ch, twf, lm);
        DIL_SCHED_FENCE();
        const size_t itn = it + nwaves;
        if (itn < batch) {
            if (PFZ) zr.load(z + itn * L * 256, lane);
            load_strided<false>(cr, c + itn * 256, lane);
        }
        static_assert(K % 2 == 0, "rows are processed in pairs");
        for (int k = 0; k < K; k += 2) {                // rows k, k + 1: both multiply-accumulates, then both inverse transforms side by side
            int64_t acc[4] = {0, 0, 0, 0}, acd[4] = {0, 0, 0, 0};
#pragma unroll
            for (int l = 0; l < L; l++) {
                const int4 a = *reinterpret_cast<const int4*>(Al + (k * L + l) * 256 + 4 * lane);
                const int4 d = *reinterpret_cast<const int4*>(Al + ((k + 1) * L + l) * 256 + 4 * lane);
                acc[0] += (int64_t)a.x * zh[l][0];
                acc[1] += (int64_t)a.y * zh[l][1];
                acc[2] += (int64_t)a.z * zh[l][2];
                acc[3] += (int64_t)a.w * zh[l][3];
                acd[0] += (int64_t)d.x * zh[l][0];
                acd[1] += (int64_t)d.y * zh[l][1];
                acd[2] += (int64_t)d.z * zh[l][2];
                acd[3] += (int64_t)d.w * zh[l][3];
            }
            const int4 th = *reinterpret_cast<const int4*>(Tl + k * 256 + 4 * lane);
            const int4 td = *reinterpret_cast<const int4*>(Tl + (k + 1) * 256 + 4 * lane);
            acc[0] -= (int64_t)ch[0] * th.x; acc[1] -= (int64_t)ch[1] * th.y; acc[2] -= (int64_t)ch[2] * th.z; acc[3] -= (int64_t)ch[3] * th.w;
            acd[0] -= (int64_t)ch[0] * td.x; acd[1] -= (int64_t)ch[1] * td.y; acd[2] -= (int64_t)ch[2] * td.z; acd[3] -= (int64_t)ch[3] * td.w;
            const uint32_t hn2 = load_row_u8(hit + (k + 1) * 256, lane);
            int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
            int32_t rd[4] = {mont_red64(acd[0]), mont_red64(acd[1]), mont_red64(acd[2]), mont_red64(acd[3])};
            DIL_SCHED_FENCE();
            ntt_inv_core2(r, rd, twi, lm);
            DIL_SCHED_FENCE();
            uint32_t hb[4], wb[4];
            unpack_row_u8(hb, hn, sc, lane);
#pragma unroll
            for (int m = 0; m < 4; m++) wb[m] = use_hint<LEVEL>(canon_small(r[m]), hb[m]);
            store_row_u8(w1_out + (it * K + k) * 256, wb, sc, lane);
            unpack_row_u8(hb, hn2, sc, lane);
#pragma unroll
            for (int m = 0; m < 4; m++) wb[m] = use_hint<LEVEL>(canon_small(rd[m]), hb[m]);
            store_row_u8(w1_out + (it * K + k + 1) * 256, wb, sc, lane);
            if (k + 2 < K) hn = load_row_u8(hit + (k + 2) * 256, lane);
        }
        if (false)
        for (int k = 0; k < K; k++) {
            int64_t acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int l = 0; l < L; l++) {
                const int4 a = *reinterpret_cast<const int4*>(Al + (k * L + l) * 256 + 4 * lane);
                acc[0] += (int64_t)a.x * zh[l][0];
                acc[1] += (int64_t)a.y * zh[l][1];
                acc[2] += (int64_t)a.z * zh[l][2];
                acc[3] += (int64_t)a.w * zh[l][3];
            }
            const int4 th = *reinterpret_cast<const int4*>(Tl + k * 256 + 4 * lane);
            acc[0] -= (int64_t)ch[0] * th.x;
            acc[1] -= (int64_t)ch[1] * th.y;
            acc[2] -= (int64_t)ch[2] * th.z;
            acc[3] -= (int64_t)ch[3] * th.w;
            uint32_t hb[4];
            unpack_row_u8(hb, hn, sc, lane);
            if (k + 1 < K) hn = load_row_u8(hit + (k + 1) * 256, lane);
            int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
            DIL_SCHED_FENCE();
            ntt_inv_core(r, twi, lm);
            DIL_SCHED_FENCE();
            const size_t o = (it * K + k) * 256;
            uint32_t wb[4];
#pragma unroll
            for (int m = 0; m < 4; m++) wb[m] = use_hint<LEVEL>(canon_small(r[m]), hb[m]);
            store_row_u8(w1_out + o, wb, sc, lane);
        }
    }
}

// LDS budget (160 KiB): tables 16 KiB + key + NW * L KiB of per-wave vector slices
template <int LEVEL> struct SharedNW;
// (y^ / z^ live in registers, the workgroup is 16 waves at every level: tables 16 + key K L (+ K) + 16 x 1.25 KiB)
#define DIL_MVS_NW 16
template <> struct SharedNW<2> { static constexpr int MATVEC = DIL_MVS_NW, VERIFY = DIL_MVS_NW; };
template <> struct SharedNW<3> { static constexpr int MATVEC = DIL_MVS_NW, VERIFY = DIL_MVS_NW; };
template <> struct SharedNW<5> { static constexpr int MATVEC = DIL_MVS_NW, VERIFY = DIL_MVS_NW; };   // 16 + 56 + 8 + 16 x 1.25 = 100 KiB

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
// wave-per-item pays once every SIMD has several items to interleave; below that the
// workgroup-per-item kernels expose more parallelism per item (lower latency)
static inline bool use_wpi(size_t batch, const Tables& t)
{
    if (t.fused_mode == 1) return false;
    if (t.fused_mode == 2) return true;
    return batch >= (size_t)t.num_cus * 8;
}

template <int LEVEL, int OUT, int AF, int YF>
static hipError_t launch_matvec_level(int32_t* w, uint8_t* w1, int32_t* w0, const int32_t* A, const int32_t* y,
                                      size_t batch, int shared_A, const Tables& t, hipStream_t s, KeyMap km)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    if (YF == Y_PACKED && !use_wpi(batch, t)) return hipErrorInvalidValue;       // packed y: wave-per-item / shared-key shapes only
    if (AF == A_P24 && shared_A) return hipErrorInvalidValue;      // the shared-key kernels keep A in LDS: nothing to save
    if (AF == A_I32 && use_wpi(batch, t) && shared_A) {
        constexpr int NW = SharedNW<LEVEL>::MATVEC;
        const int g = grid_for((batch + NW - 1) / NW,
                               t.num_cus * resident_blocks_per_cu(matvec_shared_kernel<K, L, LEVEL, OUT, NW, YF>, 64 * NW, DIL_MVS_WGS, t.device));
        note_launch(OUT == OUT_W ? "matvec_shared" : "sign1_shared", g, NW, batch);
        hipLaunchKernelGGL((matvec_shared_kernel<K, L, LEVEL, OUT, NW, YF>), g, 64 * NW, 0, s, w, w1, w0, A, y, batch, t.fwd,
                           t.inv_pipe);
        return hipGetLastError();
    }
    if (use_wpi(batch, t)) {
        const int g = grid_for((batch + 3) / 4,
                               t.num_cus * resident_blocks_per_cu(matvec_wpi_kernel<K, L, LEVEL, OUT, AF, YF>, 256, t.wpi_blocks_per_cu, t.device));
        note_launch(OUT == OUT_W ? "matvec_wpi" : "sign1_wpi", g, 4, batch);
        hipLaunchKernelGGL((matvec_wpi_kernel<K, L, LEVEL, OUT, AF, YF>), g, 256, 0, s, w, w1, w0, A, y, batch, shared_A, km, t.fwd,
                           t.inv_pipe);
        return hipGetLastError();
    }
    const int grid = grid_for(batch, t.num_cus * t.fused_wgs_per_cu);
    hipLaunchKernelGGL((matvec_kernel<K, L, LEVEL, OUT, AF>), grid, 64 * (K > L ? K : L), 0, s, w, w1, w0, A, y, batch,
                       shared_A, km, t.fwd, t.inv_pipe);
    return hipGetLastError();
}

hipError_t launch_matvec(int level, int out_mode, int32_t* w, uint8_t* w1, int32_t* w0, const int32_t* A,
                         const int32_t* y, size_t batch, int shared_A, const Tables& t, hipStream_t s, KeyMap km, uint8_t* w1_packed,
                         int a_fmt, int y_fmt)
{
    if (batch == 0) return hipSuccess;
    if (out_mode != OUT_W) w = reinterpret_cast<int32_t*>(w1_packed);      // the kernels' w slot carries packed w1 in this mode
    if (y_fmt == Y_PACKED && out_mode == OUT_W) return hipErrorInvalidValue;   // packed y exists only inside the signing loop (phase 1)
#define DIL_MV2(LV, AF)                                                                                                      \
    return out_mode == OUT_W    ? launch_matvec_level<LV, OUT_W, AF, Y_I32>(w, w1, w0, A, y, batch, shared_A, t, s, km)          \
           : y_fmt == Y_PACKED ? launch_matvec_level<LV, OUT_W1W0, AF, Y_PACKED>(w, w1, w0, A, y, batch, shared_A, t, s, km)   \
                               : launch_matvec_level<LV, OUT_W1W0, AF, Y_I32>(w, w1, w0, A, y, batch, shared_A, t, s, km)
#define DIL_MV(LV)                      \
    if (a_fmt == A_P24) DIL_MV2(LV, A_P24); \
    DIL_MV2(LV, A_I32)
    switch (level) {
    case 2: DIL_MV(2);
    case 3: DIL_MV(3);
    case 5: DIL_MV(5);
    default: return hipErrorInvalidValue;
    }
#undef DIL_MV
#undef DIL_MV2
}

// the wave-per-item / shared-key shapes serve this batch (keygen's fused output stage, the signing loop's packed y)
bool fused_wpi_shape(size_t batch, const Tables& t) { return use_wpi(batch, t); }
// keygen's fused mat-vec + Power2Round + t1 / t0 packing; false: this batch is served by the unfused kernels instead
bool keygen_fused_available(size_t batch, const Tables& t) { return use_wpi(batch, t); }

template <int LEVEL>
static hipError_t launch_keygen_level(uint8_t* pk, size_t pk_stride, uint8_t* sk, size_t sk_stride, size_t sk_t0_offset, const int32_t* A,
                                      const int32_t* s1, const int32_t* s2, size_t batch, const Tables& t, hipStream_t s, int a_fmt)
{
    if (a_fmt == A_P24) {
        const int g = grid_for((batch + 3) / 4, t.num_cus * resident_blocks_per_cu(keygen_wpi_kernel<LEVEL, A_P24>, 256, t.wpi_blocks_per_cu, t.device));
        note_launch("keygen_wpi", g, 4, batch);
        hipLaunchKernelGGL((keygen_wpi_kernel<LEVEL, A_P24>), g, 256, 0, s, pk, pk_stride, sk, sk_stride, sk_t0_offset, A, s1, s2, batch, t.fwd, t.inv_pipe);
    } else {
        const int g = grid_for((batch + 3) / 4, t.num_cus * resident_blocks_per_cu(keygen_wpi_kernel<LEVEL, A_I32>, 256, t.wpi_blocks_per_cu, t.device));
        note_launch("keygen_wpi", g, 4, batch);
        hipLaunchKernelGGL((keygen_wpi_kernel<LEVEL, A_I32>), g, 256, 0, s, pk, pk_stride, sk, sk_stride, sk_t0_offset, A, s1, s2, batch, t.fwd, t.inv_pipe);
    }
    return hipGetLastError();
}

hipError_t launch_keygen_matvec(int level, uint8_t* pk, size_t pk_stride, uint8_t* sk, size_t sk_stride, size_t sk_t0_offset,
                                const int32_t* A, const int32_t* s1, const int32_t* s2, size_t batch, const Tables& t, hipStream_t s, int a_fmt)
{
    if (batch == 0) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(pk) | reinterpret_cast<uintptr_t>(sk) | pk_stride | sk_stride | sk_t0_offset) & 3) return hipErrorInvalidValue;
    switch (level) {
    case 2: return launch_keygen_level<2>(pk, pk_stride, sk, sk_stride, sk_t0_offset, A, s1, s2, batch, t, s, a_fmt);
    case 3: return launch_keygen_level<3>(pk, pk_stride, sk, sk_stride, sk_t0_offset, A, s1, s2, batch, t, s, a_fmt);
    case 5: return launch_keygen_level<5>(pk, pk_stride, sk, sk_stride, sk_t0_offset, A, s1, s2, batch, t, s, a_fmt);
    default: return hipErrorInvalidValue;
    }
}

template <int LEVEL>
static void launch_verify_wpi(uint8_t* w1, const int32_t* A, const int32_t* z, const int32_t* c, const int32_t* t1,
                              const uint8_t* h, size_t batch, int shared_pk, const Tables& t, hipStream_t s)
{
    if (shared_pk) {
        constexpr int NW = SharedNW<LEVEL>::VERIFY;
        const int g = grid_for((batch + NW - 1) / NW, t.num_cus);
        note_launch("verify_shared", g, NW, batch);
        hipLaunchKernelGGL((verify_shared_kernel<LEVEL, NW>), g, 64 * NW, 0, s, w1, A, z, c, t1, h, batch, t.fwd, t.inv_pipe);
    } else {
        const int g = grid_for((batch + 3) / 4, t.num_cus * resident_blocks_per_cu(verify_wpi_kernel<LEVEL>, 256, t.wpi_blocks_per_cu, t.device));
        note_launch("verify_wpi", g, 4, batch);
        hipLaunchKernelGGL((verify_wpi_kernel<LEVEL>), g, 256, 0, s, w1, A, z, c, t1, h, batch, t.fwd, t.inv_pipe);
    }
}

hipError_t launch_verify(int level, uint8_t* w1, const int32_t* A, const int32_t* z, const int32_t* c,
                         const int32_t* t1, const uint8_t* h, size_t batch, int shared_pk,
                         const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    if (use_wpi(batch, t)) {
        switch (level) {
        case 2: launch_verify_wpi<2>(w1, A, z, c, t1, h, batch, shared_pk, t, s); break;
        case 3: launch_verify_wpi<3>(w1, A, z, c, t1, h, batch, shared_pk, t, s); break;
        case 5: launch_verify_wpi<5>(w1, A, z, c, t1, h, batch, shared_pk, t, s); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    const int grid = grid_for(batch, t.num_cus * t.fused_wgs_per_cu);
#define DIL_VY(LV)                                                                                             \
    hipLaunchKernelGGL(verify_kernel<LV>, grid,                                                                \
                       64 * (Par<LV>::K > Par<LV>::L + 1 ? Par<LV>::K : Par<LV>::L + 1), 0, s, w1, A, z, c, t1, \
                       h, batch, shared_pk, t.fwd, t.inv_pipe);                                                     \
    break
    switch (level) {
    case 2: DIL_VY(2);
    case 3: DIL_VY(3);
    case 5: DIL_VY(5);
    default: return hipErrorInvalidValue;
    }
#undef DIL_VY
    return hipGetLastError();
}

hipError_t launch_sign2(int level, int32_t* z, uint8_t* h, int32_t* flags, const int32_t* c, const int32_t* y,
                        const int32_t* w0, const uint8_t* w1, const int32_t* s1hat, const int32_t* s2hat,
                        const int32_t* t0hat, size_t batch, int shared_key, const Tables& t, hipStream_t s, KeyMap km,
                        int32_t* w0_scratch, int y_fmt, bool small_key)
{
    if (batch == 0) return hipSuccess;
    // a speculative round is spec_n items x S attempts, entry = item * S + attempt; the kernel reads an item's earlier attempts one per
    // lane, so S cannot exceed the wave width (scheme.hip's s_max = 64 is that bound)
    if (km.spec_n && (km.S > 64 || (size_t)km.spec_n * km.S != batch)) return hipErrorInvalidValue;
    if (y_fmt == Y_PACKED && !(use_wpi(batch, t) && small_key)) return hipErrorInvalidValue;      // packed y: the signing loop's wave-per-item shapes only
    if (!w1 && !use_wpi(batch, t)) return hipErrorInvalidValue;        // w1 == nullptr = the loop's W0W1 plane in w0: the wave-per-item kernels only
    if (use_wpi(batch, t) && w0_scratch) {      // the signing loop's early-exit form (w0 is its own scratch, reused for r0)
        if (w0_scratch != w0) return hipErrorInvalidValue;
#define DIL_S2E2(LV, YF, SH)                                                                                                     \
    {                                                                                                                            \
        const int g = grid_for((batch + 3) / 4, t.num_cus * resident_blocks_per_cu(sign2_early_wpi_kernel<LV, YF, SH>, 256, t.wpi_blocks_per_cu, t.device)); \
        note_launch("sign2_early_wpi", g, 4, batch);                                                                             \
        hipLaunchKernelGGL((sign2_early_wpi_kernel<LV, YF, SH>), g, 256, 0, s, z, h, flags, c, y, w0_scratch, w1, s1hat, s2hat, t0hat, batch, shared_key, \
                           km, t.fwd, t.inv_pipe);                                                                               \
    }
    /* the staged, paired rows of the one-key form need the small-key promise; without it the per-item form serves one key too */ \
#define DIL_S2E(LV)                                                \
    if (y_fmt == Y_PACKED && shared_key) DIL_S2E2(LV, Y_PACKED, true) \
    else if (y_fmt == Y_PACKED) DIL_S2E2(LV, Y_PACKED, false)      \
    else if (shared_key && small_key) DIL_S2E2(LV, Y_I32, true)    \
    else DIL_S2E2(LV, Y_I32, false)                                \
    break
        switch (level) {
        case 2: DIL_S2E(2);
        case 3: DIL_S2E(3);
        case 5: DIL_S2E(5);
        default: return hipErrorInvalidValue;
        }
#undef DIL_S2E
#undef DIL_S2E2
        return hipGetLastError();
    }
    if (use_wpi(batch, t)) {
#define DIL_S2W2(LV, YF, SH, SM)                                                                                                 \
    {                                                                                                                            \
        const int g = grid_for((batch + 3) / 4, t.num_cus * resident_blocks_per_cu(sign2_wpi_kernel<LV, YF, SH, SM>, 256, t.wpi_blocks_per_cu, t.device)); \
        note_launch("sign2_wpi", g, 4, batch);                                                                                   \
        hipLaunchKernelGGL((sign2_wpi_kernel<LV, YF, SH, SM>), g, 256, 0, s, z, h, flags, c, y, w0, w1, s1hat, s2hat, t0hat, batch, shared_key, km, t.fwd, \
                           t.inv_pipe);                                                                                          \
    }
#define DIL_S2W(LV)                                                \
    if (y_fmt == Y_PACKED && shared_key) DIL_S2W2(LV, Y_PACKED, true, true) \
    else if (y_fmt == Y_PACKED) DIL_S2W2(LV, Y_PACKED, false, true) \
    else if (shared_key && small_key) DIL_S2W2(LV, Y_I32, true, true) \
    else if (small_key) DIL_S2W2(LV, Y_I32, false, true)           \
    else if (shared_key) DIL_S2W2(LV, Y_I32, true, false)          \
    else DIL_S2W2(LV, Y_I32, false, false)                         \
    break
        switch (level) {
        case 2: DIL_S2W(2);
        case 3: DIL_S2W(3);
        case 5: DIL_S2W(5);
        default: return hipErrorInvalidValue;
        }
#undef DIL_S2W
#undef DIL_S2W2
        return hipGetLastError();
    }
    const int grid = grid_for(batch, t.num_cus * t.fused_wgs_per_cu);
#define DIL_S2(LV)                                                                                             \
    hipLaunchKernelGGL(sign2_kernel<LV>, grid,                                                                 \
                       64 * (Par<LV>::K > Par<LV>::L + 1 ? Par<LV>::K : Par<LV>::L + 1), 0, s, z, h, flags, c, \
                       y, w0, w1, s1hat, s2hat, t0hat, batch, shared_key, km, t.fwd, t.inv_pipe);                   \
    break
    switch (level) {
    case 2: DIL_S2(2);
    case 3: DIL_S2(3);
    case 5: DIL_S2(5);
    default: return hipErrorInvalidValue;
    }
#undef DIL_S2
    return hipGetLastError();
}

}  // namespace dil
