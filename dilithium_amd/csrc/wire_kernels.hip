// wire_kernels.hip -- the verify pipeline with the reference's WIRE FORMATS in its load / store stages
// (SURVEY 8(f) row N2 cashed in): the fused kernels read the packed signature and public key directly --
//   z        18 / 20 bits per coefficient, gamma1 - z          decoder.v:89-143, uncenter_coeff.v:49-65
//   t1       10 bits per coefficient (x 2^13 on the fly)        decoder.v:96-100
//   hints    omega position bytes + K cumulative counts         usehint.v:92-114
//   c        SampleInBall(c~) as 2 bits per coefficient         gen_c.v:163-196,318-339
// -- and write w1 already packed (4 / 6 bits, encoder.v:96-133), so that no int32 z / t1 / h / w1 temporary ever
// crosses HBM: a level-3 verification moves 30 KiB (A) + 3.2 (z) + 1.9 (t1) + 0.06 (hints) + 0.25 (c) + 0.75 KiB (w1)
// instead of 45 KiB plus the codec kernels' own read + write of the same fields.  The ||z|| < gamma1 - beta check
// (norm_check.v:84-105) and the hint-encoding validation ride along (z is in registers anyway).
// Same arithmetic as verify_wpi_kernel / verify_shared_kernel (pipelines.hip): combined_top.v:1207-1469.
#define DIL_PRODUCT_MAD64   // the constant products as two v_mad_i64_i32 (modarith.hpp MAD64): these kernels are VALU-bound
#include <algorithm>
#include "launch_util.hpp"
#include "wire_common.hpp"
#include "keccak.hpp"
#include "sampler_bodies.hpp"
#include "coop_bodies.hpp"

namespace dil {

#define DIL_SCHED_FENCE_W() __builtin_amdgcn_sched_barrier(0)
// shape of verify_wire_shared_kernel per level: waves per workgroup, exchange policy (all through LDS, or in registers where
// the 15 LDS addresses would push a 16-wave workgroup past its 128 registers), prefetch of the next item's packed z
template <int LEVEL> struct WireSh;
#define DIL_VWS_DUAL 1      // transforms side by side (forward: all L; inverse: row pairs)
template <> struct WireSh<2> { static constexpr int NW = 16; static constexpr bool PFZ = false, DUAL = DIL_VWS_DUAL; using X = XAllLds; };
template <> struct WireSh<3> { static constexpr int NW = 16; static constexpr bool PFZ = false, DUAL = DIL_VWS_DUAL; using X = X10Dpp; };
template <> struct WireSh<5> { static constexpr int NW = 12; static constexpr bool PFZ = false, DUAL = DIL_VWS_DUAL; using X = XAllLds; };   // 12 waves: up to 168 VGPRs

// ---------------------------------------------------------------------------------------------------------
// distinct public keys: wave per item, A streamed from HBM (expanded by expand_a_kernel), everything else packed
// ---------------------------------------------------------------------------------------------------------
// forward transforms in pairs, NTT(t1[k+1] 2^13) beside INTT(row k): level 3 83.8 -> 81.2 us, level 2 65.5 -> 62.1 us per 8192; not at
// level 5, where the second chain's registers spill under the 168-VGPR cap (126.5 -> 130.2 us): profiles/r04l_ab_wire_matvec.txt
#define DIL_WW_DUAL(LEVEL) ((LEVEL) != 5)
#define DIL_WW_WAVES(LEVEL) 3      // waves per SIMD the register allocator aims for (168 VGPRs)
// T1H: the caller keeps t1^ = NTT(t1 2^13) of every key beside its matrix (dil_expand_t1_dev: VY_NTT_T1 of combined_top.v:1259-1313 done
// once per key instead of once per verification): the K transforms of t1 leave the kernel -- L + 1 forward and K inverse remain --
// for 6 KiB of int32 per key in place of 1.9 KiB of packed t1.
// SIB (round 6): c = SampleInBall(c~) is computed INSIDE the item loop by the wave that owns the item -- one SHAKE256 state over the wavefront
// (keccak_coop.hpp: 29 VALU instructions per round) and the ballot sampler of coop_bodies.hpp, working in the wave's z^ area of LDS before
// the forward transforms fill it -- instead of a sample_in_ball_bits_kernel launch in front (24-32 us per 8192, serial) and 256 B of
// compact c per item through HBM.  The reference feeds c straight from the sampler into the transform too (gen_c.v:163-196,318-339 ->
// VY_NTT_C, combined_top.v:1314).  cbits == nullptr selects it at run time (same template otherwise).
template <int LEVEL, int AF, bool T1H = false, bool SIB = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DIL_WW_WAVES(LEVEL), DIL_WW_WAVES(LEVEL)))) void verify_wire_wpi_kernel(
    uint8_t* __restrict__ w1p_out, int32_t* __restrict__ verdict, const int32_t* __restrict__ A,
    const uint8_t* __restrict__ pk, size_t pk_stride, const uint8_t* __restrict__ sig, size_t sig_stride,
    const uint32_t* __restrict__ cbits, size_t batch, const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab,
    const int32_t* __restrict__ t1hat = nullptr)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    constexpr bool DUAL = DIL_WW_DUAL(LEVEL) && !T1H;
    using W = Wire<LEVEL>;
    // per wave: L KiB of z^ | 64 dwords byte scratch | 64 dwords hint bitmap
    using XP = X10Pick<true>;
    using PT = PipeTables<true>;
    constexpr int WAVE_DW = L * 256 + 64 + 64 + XP::DW;
    __shared__ __attribute__((aligned(16))) uint32_t lds[PT::DWORDS + 4 * WAVE_DW];
    const int lane = threadIdx.x & 63, wv = wave_in_block();
    PT::stage(lds, fwd_tab, inv_tab);
    const typename PT::Fwd twf = PT::fwd(lds, fwd_tab, lane);
    const typename PT::Inv twi = PT::inv(lds, inv_tab, lane);
    uint32_t* zl = lds + PT::DWORDS + wv * WAVE_DW;
    uint32_t* sc = zl + L * 256;
    uint32_t* bm = sc + 64;
    const typename XP::type lm(bm + 64, lane);
    const PackedLane<W::ZBITS> plz(lane);
    const PackedLane<10> plt(lane);
    const size_t nwaves = (size_t)gridDim.x * 4;
    size_t it = (size_t)blockIdx.x * 4 + wv;
    RawZ<LEVEL> zr;
    uint32_t cb = 0, hb0 = 0, hb1 = 0;
    static_assert(sizeof(coop::SibShared) <= (size_t)L * 1024, "SampleInBall works in the wave's z^ area");
    auto load_item = [&](size_t i) {
        const uint8_t* sg = sig + i * sig_stride;
        zr.load(sg + 32, plz);
        if constexpr (SIB) {                                   // this lane's dword of c~ in the sponge's own lane order: state words 0..3 sit in lanes
            const int l5 = lane & 31;                          // 0..3 (low halves) and 32..35 (high halves) -- keccak_coop.hpp Lane::init
            cb = l5 < 4 ? coop::ld_u32u(sg + 4 * (2 * l5 + (lane >> 5))) : 0u;
        } else {
            cb = cbits[i * 64 + lane];
        }
        hb0 = (lane < W::HINT_BYTES) ? sg[32 + W::Z_BYTES + lane] : 0;          // (61 hint bytes at level 3: never read past the signature)
        hb1 = (64 + lane < W::HINT_BYTES) ? sg[32 + W::Z_BYTES + 64 + lane] : 0;
    };
    auto t1_row = [&](int32_t (&th)[4], const uint32_t (&tn)[4]) {               // t1[k] 2^13 from its packed 10-bit fields (decoder.v:96-100)
        uint32_t f[4];
        plt.fields(f, tn);
#pragma unroll
        for (int m = 0; m < 4; m++) th[m] = (int32_t)(f[m] << 13);
    };
    if (it < batch) load_item(it);
    __syncthreads();                               // tables staged (the only barrier)
    for (; it < batch; it += nwaves) {
        constexpr int PD = ARow<L, AF>::PD;
        const int32_t* Ait = A + it * (size_t)(K * L) * PD;
        const uint8_t* t1it = pk + it * pk_stride + 32;
        ARow<L, AF> Ar;
        Ar.load(Ait, lane, true);
        uint32_t tn[4] = {0, 0, 0, 0};
        int4 thn = make_int4(0, 0, 0, 0);                       // T1H: row k + 1 of t1^, one row ahead like A
        const int32_t* thit = T1H ? t1hat + it * (size_t)K * 256 : nullptr;
        if constexpr (T1H) thn = ld_nt4(thit + 4 * lane);
        else plt.load(tn, t1it);
        const bool bad = hints_to_bitmap<LEVEL>(bm, sc, hb0, hb1, lane);
        int32_t zmax = 0;
        int32_t ch[4];
        if constexpr (SIB) {
            coop::SibShared& sh = *reinterpret_cast<coop::SibShared*>(zl);     // (free until the forward transforms below store z^)
            // the sponge's dozen per-lane constants are rebuilt per item from an opaque copy of the lane number: kept across the item loop
            // they cost 12 VGPRs of a kernel that sits at its 168-register cap (level 5: 52 spilled dwords against 16)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            coop::Sponge<17> sp;
            sp.init(lane_o);
            sp.v = cb;
            sp.pad(4);
            sp.permute();
            coop::sib_sample<true>(sp, Par<LEVEL>::TAU, sh, lane);
#pragma unroll
            for (int m = 0; m < 4; m++) ch[m] = sh.c[lane + 64 * m];
            coop::sib_sync<true>();                                            // ... read before z^ overwrites it
        } else {
            decode_c(ch, cb);
        }
        if constexpr (DUAL) {
        // the L + 1 forward transforms two at a time (ntt_core.hpp ntt_fwd_core2: one set of twiddle reads, two dependency chains)
#pragma unroll
        for (int l = 0; l + 1 < L; l += 2) {
            int32_t r[4], q[4];
            decode_z<LEVEL>(r, zr.v[l], plz, zmax);
            decode_z<LEVEL>(q, zr.v[l + 1], plz, zmax);
            ntt_fwd_core2(r, q, twf, lm);
            *reinterpret_cast<int4*>(zl + l * 256 + 4 * lane) = make_int4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<int4*>(zl + (l + 1) * 256 + 4 * lane) = make_int4(q[0], q[1], q[2], q[3]);
        }
        if (L & 1) {
            int32_t r[4];
            decode_z<LEVEL>(r, zr.v[L - 1], plz, zmax);
            ntt_fwd_core2(r, ch, twf, lm);
            *reinterpret_cast<int4*>(zl + (L - 1) * 256 + 4 * lane) = make_int4(r[0], r[1], r[2], r[3]);
        } else {
            ntt_fwd_core(ch, twf, lm);
        }
        } else {
#pragma unroll
        for (int l = 0; l < L; l++) {
            int32_t r[4];
            decode_z<LEVEL>(r, zr.v[l], plz, zmax);
            ntt_fwd_core(r, twf, lm);
            *reinterpret_cast<int4*>(zl + l * 256 + 4 * lane) = make_int4(r[0], r[1], r[2], r[3]);
        }
        ntt_fwd_core(ch, twf, lm);
        }
        DIL_SCHED_FENCE_W();
        const size_t itn = it + nwaves;
        if (itn < batch) load_item(itn);
        const bool zrej = __ballot(zmax >= Par<LEVEL>::GAMMA1 - Par<LEVEL>::BETA) != 0;
        if (lane == 0) verdict[it] = (zrej ? 2 : 0) | (bad ? 4 : 0);
        // DUAL: row k's INTT runs beside row k + 1's NTT(t1 2^13) (ntt_fwd_inv_pair): th is always one row ahead
        int32_t th[4] = {0, 0, 0, 0};
        if constexpr (DUAL) {
            t1_row(th, tn);
            plt.load(tn, t1it + 320);
            ntt_fwd_core(th, twf, lm);
        }
        for (int k = 0; k < K; k++) {
            int64_t acc[4] = {0, 0, 0, 0};
            mac_row<L>(acc, Ar, zl, lane);
            uint32_t hb[4];
            row_hint_bits(hb, bm, k, lane);
            int32_t r[4];
            if constexpr (DUAL) {
#pragma unroll
            for (int m = 0; m < 4; m++) acc[m] -= (int64_t)ch[m] * th[m];
#pragma unroll
            for (int m = 0; m < 4; m++) r[m] = mont_red64(acc[m]);
            if (k + 1 < K) {
                Ar.load(Ait + (size_t)(k + 1) * L * PD, lane, true);
                t1_row(th, tn);
                if (k + 2 < K) plt.load(tn, t1it + (k + 2) * 320);
                DIL_SCHED_FENCE_W();
                ntt_fwd_inv_pair(th, r, twf, twi, lm);
            } else {
                DIL_SCHED_FENCE_W();
                ntt_inv_core(r, twi, lm);
            }
            DIL_SCHED_FENCE_W();
            } else if constexpr (T1H) {
            th[0] = thn.x, th[1] = thn.y, th[2] = thn.z, th[3] = thn.w;
#pragma unroll
            for (int m = 0; m < 4; m++) acc[m] -= (int64_t)ch[m] * th[m];
#pragma unroll
            for (int m = 0; m < 4; m++) r[m] = mont_red64(acc[m]);
            if (k + 1 < K) {
                Ar.load(Ait + (size_t)(k + 1) * L * PD, lane, true);
                thn = ld_nt4(thit + (k + 1) * 256 + 4 * lane);
            }
            DIL_SCHED_FENCE_W();
            ntt_inv_core(r, twi, lm);
            DIL_SCHED_FENCE_W();
            } else {
            t1_row(th, tn);
            if (k + 1 < K) {
                Ar.load(Ait + (size_t)(k + 1) * L * PD, lane, true);
                plt.load(tn, t1it + (k + 1) * 320);
            }
            DIL_SCHED_FENCE_W();
            ntt_fwd_core(th, twf, lm);
            DIL_SCHED_FENCE_W();
#pragma unroll
            for (int m = 0; m < 4; m++) acc[m] -= (int64_t)ch[m] * th[m];
#pragma unroll
            for (int m = 0; m < 4; m++) r[m] = mont_red64(acc[m]);
            DIL_SCHED_FENCE_W();
            ntt_inv_core(r, twi, lm);
            DIL_SCHED_FENCE_W();
            }
            uint32_t wb[4];
#pragma unroll
            for (int m = 0; m < 4; m++) wb[m] = use_hint<LEVEL>(canon_small(r[m]), hb[m]);
            store_row_w1_packed<LEVEL>(w1p_out + (it * K + k) * W::W1_ROW_BYTES, wb, sc, lane);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// one public key for the batch: A (expanded once) and t1^ = NTT(t1 2^13) LDS-resident, per item only the signature
// ---------------------------------------------------------------------------------------------------------
template <int LEVEL, int NW>
__global__ __launch_bounds__(64 * NW) void verify_wire_shared_kernel(
    uint8_t* __restrict__ w1p_out, int32_t* __restrict__ verdict, const int32_t* __restrict__ A,
    const uint8_t* __restrict__ pk, const uint8_t* __restrict__ sig, size_t sig_stride,
    const uint32_t* __restrict__ cbits, size_t batch, const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    using W = Wire<LEVEL>;
    // z^ in registers (a lane multiplies the coefficients it transformed itself), all three exchanges through LDS, 16 waves per
    // workgroup at every level -- as verify_shared_kernel / matvec_shared_kernel (pipelines.hip)
    constexpr int XDW = 256;
    constexpr int WAVE_DW = 64 + 64 + XDW;           // byte scratch | hint bitmap | exchange buffer
    using PT = PipeTables<true>;
    __shared__ __attribute__((aligned(16))) uint32_t lds[PT::DWORDS + (K * L + K) * 256 + NW * WAVE_DW];
    const int lane = threadIdx.x & 63, wv = wave_in_block();
    PT::stage(lds, fwd_tab, inv_tab);
    uint32_t* Al = lds + PT::DWORDS;
    uint32_t* Tl = Al + K * L * 256;
    for (int i = threadIdx.x; i < K * L * 64; i += blockDim.x)
        reinterpret_cast<uint4*>(Al)[i] = reinterpret_cast<const uint4*>(A)[i];
    const typename PT::Fwd twf = PT::fwd(lds, fwd_tab, lane);
    const typename PT::Inv twi = PT::inv(lds, inv_tab, lane);
    uint32_t* sc = Tl + K * 256 + wv * WAVE_DW;
    uint32_t* bm = sc + 64;
    const typename WireSh<LEVEL>::X lm(bm + 64, lane);
    const PackedLane<W::ZBITS> plz(lane);
    const size_t nwaves = (size_t)gridDim.x * NW;
    size_t it = (size_t)blockIdx.x * NW + wv;
    // packed z (4 dwords per polynomial and lane) is prefetched one item ahead where the registers allow it; at level 5 the
    // second copy would spill under the 128-register cap of a 16-wave workgroup and z is loaded where it is decoded
    constexpr bool PFZ = WireSh<LEVEL>::PFZ;
    RawZ<LEVEL> zr;
    uint32_t cb = 0, hb0 = 0, hb1 = 0;
    auto load_item = [&](size_t i) {
        const uint8_t* sg = sig + i * sig_stride;
        if (PFZ) zr.load(sg + 32, plz);
        cb = cbits[i * 64 + lane];
        hb0 = (lane < W::HINT_BYTES) ? sg[32 + W::Z_BYTES + lane] : 0;          // (61 hint bytes at level 3: never read past the signature)
        hb1 = (64 + lane < W::HINT_BYTES) ? sg[32 + W::Z_BYTES + 64 + lane] : 0;
    };
    if (it < batch) load_item(it);
    __syncthreads();                               // tables + A staged
    for (int k = wv; k < K; k += NW) {             // t1_k 2^13 -> NTT -> LDS, lazy residues (VY_NTT_T1, combined_top.v:1259)
        const PackedLane<10> plt(lane);
        uint32_t tn[4], f[4];
        plt.load(tn, pk + 32 + k * 320);
        plt.fields(f, tn);
        int32_t th[4];
#pragma unroll
        for (int m = 0; m < 4; m++) th[m] = (int32_t)(f[m] << 13);
        ntt_fwd_core(th, twf, lm);
        *reinterpret_cast<int4*>(Tl + k * 256 + 4 * lane) = make_int4(th[0], th[1], th[2], th[3]);
    }
    __syncthreads();
    for (; it < batch; it += nwaves) {
        const bool bad = hints_to_bitmap<LEVEL>(bm, sc, hb0, hb1, lane);
        int32_t zmax = 0;
        int32_t zh[L][4];
        if (!PFZ) zr.load(sig + it * sig_stride + 32, plz);
#pragma unroll
        for (int l = 0; l < L; l++) decode_z<LEVEL>(zh[l], zr.v[l], plz, zmax);
        int32_t ch[4];
        decode_c(ch, cb);
        if constexpr (WireSh<LEVEL>::DUAL) {
            ntt_fwd_coreN<L>(zh, twf, lm);          // side by side: one set of twiddle reads, L dependency chains
        } else {
#pragma unroll
            for (int l = 0; l < L; l++) ntt_fwd_core(zh[l], twf, lm);
        }
        ntt_fwd_core(ch, twf, lm);
        DIL_SCHED_FENCE_W();
        const size_t itn = it + nwaves;
        if (itn < batch) load_item(itn);
        const bool zrej = __ballot(zmax >= Par<LEVEL>::GAMMA1 - Par<LEVEL>::BETA) != 0;
        if (lane == 0) verdict[it] = (zrej ? 2 : 0) | (bad ? 4 : 0);
        if constexpr (WireSh<LEVEL>::DUAL) {
        for (int k = 0; k < K; k += 2) {                // rows k, k + 1: both multiply-accumulates, then both inverse transforms side by side
            int64_t acc[4] = {0, 0, 0, 0}, acd[4] = {0, 0, 0, 0};
#pragma unroll
            for (int l = 0; l < L; l++) {
                const int4 a = *reinterpret_cast<const int4*>(Al + (k * L + l) * 256 + 4 * lane);
                const int4 d = *reinterpret_cast<const int4*>(Al + ((k + 1) * L + l) * 256 + 4 * lane);
                acc[0] += (int64_t)a.x * zh[l][0]; acc[1] += (int64_t)a.y * zh[l][1]; acc[2] += (int64_t)a.z * zh[l][2]; acc[3] += (int64_t)a.w * zh[l][3];
                acd[0] += (int64_t)d.x * zh[l][0]; acd[1] += (int64_t)d.y * zh[l][1]; acd[2] += (int64_t)d.z * zh[l][2]; acd[3] += (int64_t)d.w * zh[l][3];
            }
            const int4 th = *reinterpret_cast<const int4*>(Tl + k * 256 + 4 * lane);
            const int4 td = *reinterpret_cast<const int4*>(Tl + (k + 1) * 256 + 4 * lane);
            acc[0] -= (int64_t)ch[0] * th.x; acc[1] -= (int64_t)ch[1] * th.y; acc[2] -= (int64_t)ch[2] * th.z; acc[3] -= (int64_t)ch[3] * th.w;
            acd[0] -= (int64_t)ch[0] * td.x; acd[1] -= (int64_t)ch[1] * td.y; acd[2] -= (int64_t)ch[2] * td.z; acd[3] -= (int64_t)ch[3] * td.w;
            int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
            int32_t rd[4] = {mont_red64(acd[0]), mont_red64(acd[1]), mont_red64(acd[2]), mont_red64(acd[3])};
            DIL_SCHED_FENCE_W();
            ntt_inv_core2(r, rd, twi, lm);
            DIL_SCHED_FENCE_W();
            uint32_t hb[4], wb[4];
            row_hint_bits(hb, bm, k, lane);
#pragma unroll
            for (int m = 0; m < 4; m++) wb[m] = use_hint<LEVEL>(canon_small(r[m]), hb[m]);
            store_row_w1_packed<LEVEL>(w1p_out + (it * K + k) * W::W1_ROW_BYTES, wb, sc, lane);
            row_hint_bits(hb, bm, k + 1, lane);
#pragma unroll
            for (int m = 0; m < 4; m++) wb[m] = use_hint<LEVEL>(canon_small(rd[m]), hb[m]);
            store_row_w1_packed<LEVEL>(w1p_out + (it * K + k + 1) * W::W1_ROW_BYTES, wb, sc, lane);
        }
        } else
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
            row_hint_bits(hb, bm, k, lane);
            int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
            DIL_SCHED_FENCE_W();
            ntt_inv_core(r, twi, lm);
            DIL_SCHED_FENCE_W();
            uint32_t wb[4];
#pragma unroll
            for (int m = 0; m < 4; m++) wb[m] = use_hint<LEVEL>(canon_small(r[m]), hb[m]);
            store_row_w1_packed<LEVEL>(w1p_out + (it * K + k) * W::W1_ROW_BYTES, wb, sc, lane);
        }
    }
}

// LDS budget (160 KiB): tables 16 + A K*L + t1^ K + NW * 1.5 KiB
template <int LEVEL> struct WireNW { static constexpr int N = WireSh<LEVEL>::NW; };   // level 5: 16 + 56 + 8 + 18 = 98 KiB

// ---------------------------------------------------------------------------------------------------------
// SampleInBall (gen_c.v:163-196,318-339) into the compact per-lane form the kernels above consume:
// cbits[item][lane] bit m = c[lane + 64 m] != 0, bit 4 + m = sign.  c~ is read in place from the signature (any alignment).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sample_in_ball_bits_kernel(uint32_t* __restrict__ cbits, const uint8_t* __restrict__ ctilde,
                                                                 size_t ct_stride, int tau, size_t nitems)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[SibLds<64>::BYTES];
    sample_in_ball_bits_body<64>(cbits, ctilde, ct_stride, tau, nitems, blockIdx.x, reinterpret_cast<int8_t*>(lds),
                                 reinterpret_cast<uint32_t*>(lds + SibLds<64>::CL_BYTES));
}

// Verification under few public keys: ExpandA of the key(s) -- two lanes per sponge, a 5-permutation dependency chain -- and
// SampleInBall of the signatures -- one lane per item, a serial loop -- are both latency-bound and independent.  They used to
// meet through a helper stream (fork event, join event: ~15 us of a 140-us call); here they are ONE launch whose first
// `a_blocks` workgroups expand the matrix and whose other workgroups sample the challenges, side by side on different CUs.
__global__ __launch_bounds__(64) void expand_a_sib_kernel(int32_t* __restrict__ A, const uint64_t* __restrict__ rho, size_t rho_stride_words,
                                                          int K, int L, size_t nkeys, unsigned a_blocks, uint32_t* __restrict__ cbits,
                                                          const uint8_t* __restrict__ ctilde, size_t ct_stride, int tau, size_t nitems, int coop_a)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[SibLds<64>::BYTES];
    if (blockIdx.x < a_blocks && coop_a) {     // few keys under many signatures: a polynomial per workgroup, its sponge spread over the wave
        const size_t p = blockIdx.x, key = p / (size_t)(K * L);
        const int ij = (int)(p % (size_t)(K * L)), i = ij / L, j = ij % L;
        coop::expand_a_body(A + p * 256, reinterpret_cast<const uint32_t*>(rho + key * rho_stride_words), (uint32_t)j | ((uint32_t)i << 8),
                            reinterpret_cast<uint32_t*>(lds));
    } else if (blockIdx.x < a_blocks)
        expand_a_body<true>(A, rho, rho_stride_words, K, L, nkeys, blockIdx.x, reinterpret_cast<uint32_t*>(lds));
    else
        sample_in_ball_bits_body<64>(cbits, ctilde, ct_stride, tau, nitems, blockIdx.x - a_blocks, reinterpret_cast<int8_t*>(lds),
                                     reinterpret_cast<uint32_t*>(lds + SibLds<64>::CL_BYTES));
}

// ---------------------------------------------------------------------------------------------------------
// Set-up of a signing call in ONE launch (was: ExpandA on a helper stream beside three unpack launches, three NTT launches,
// two field copies, a SHAKE256 launch and a memset on the caller's stream, joined by events).  Workgroups of one wave, three roles:
//   [0, a_blocks)                A = ExpandA(rho) of the key(s), two lanes per sponge -- only when the keys are few (latency-bound);
//                                many keys run the throughput kernel beside this launch instead (a_blocks = 0)
//   [a_blocks, +u_blocks)        grid-stride over the nk (L + 2K) polynomials of the secret key(s), one per wave and step: s1 / s2 (eta - x, 3 | 4 bits) or t0 (2^12 - x, 13 bits) read from
//                                the packed key (decoder.v:89-143), NTT, canonical out -- s1^ s2^ t0^ never exist in time domain
//   the rest                     rho' = SHAKE256(key || mu, 64) for 64 messages per workgroup, key read in place from sk; the
//                                message's attempt counter is cleared on the way (combined_top.v sign set-up, :1694-1790)
// ---------------------------------------------------------------------------------------------------------
template <int LEVEL>
__global__ __launch_bounds__(64) void sign_setup_kernel(int32_t* __restrict__ A, unsigned a_blocks, unsigned u_blocks, int32_t* __restrict__ s1h,
                                                        int32_t* __restrict__ s2h, int32_t* __restrict__ t0h, const uint8_t* __restrict__ sk,
                                                        size_t sk_bytes, size_t nk, uint64_t* __restrict__ rp, int32_t* __restrict__ attempts,
                                                        const uint64_t* __restrict__ mu, size_t key_stride, size_t batch,
                                                        const uint32_t* __restrict__ fwd_tab, int coop_a, int coop_rp)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L, ETA = LEVEL == 3 ? 4 : 2, EB = LEVEL == 3 ? 4 : 3, NP = L + 2 * K;
    __shared__ uint32_t ring[CoeffSink::LDS_DWORDS_PER_WAVE];
    const int lane = threadIdx.x;
    if (blockIdx.x < a_blocks) {
        if (coop_a) {                             // a polynomial per workgroup, its sponge spread over the wave (coop_bodies.hpp)
            const size_t p = blockIdx.x, key = p / (size_t)(K * L);
            const int ij = (int)(p % (size_t)(K * L)), i = ij / L, j = ij % L;
            coop::expand_a_body(A + p * 256, reinterpret_cast<const uint32_t*>(sk + key * sk_bytes), (uint32_t)j | ((uint32_t)i << 8), ring);
        } else {
            expand_a_body<true>(A, reinterpret_cast<const uint64_t*>(sk), sk_bytes / 8, K, L, nk, blockIdx.x, ring);
        }
        return;
    }
    const size_t u0 = blockIdx.x - a_blocks;
    if (u0 < u_blocks) {                       // persistent over the polynomials: the twiddles stay in registers
      TwRegs tw;
      tw.load(fwd_tab, lane);
      const X10Dpp lm(lane);
      for (size_t u = u0; u < nk * NP; u += u_blocks) {
        const size_t key = u / NP;
        const int q = (int)(u % NP);
        const uint8_t* base = sk + key * sk_bytes + 96;
        int32_t r[4];
        int32_t* out;
        if (q < L + K) {
            const PackedLane<EB> pl(lane);
            uint32_t raw[4], f[4];
            pl.load(raw, base + (size_t)q * (32 * EB));
            pl.fields(f, raw);
#pragma unroll
            for (int m = 0; m < 4; m++) r[m] = ETA - (int32_t)f[m];
            out = q < L ? s1h + (key * L + q) * 256 : s2h + (key * K + (q - L)) * 256;
        } else {
            const PackedLane<13> pl(lane);
            uint32_t raw[4], f[4];
            pl.load(raw, base + (size_t)(L + K) * (32 * EB) + (size_t)(q - L - K) * 416);
            pl.fields(f, raw);
#pragma unroll
            for (int m = 0; m < 4; m++) r[m] = (1 << 12) - (int32_t)f[m];
            out = t0h + (key * K + (q - L - K)) * 256;
        }
        ntt_fwd_core(r, tw, lm);
        *reinterpret_cast<int4*>(out + 4 * lane) = make_int4((int32_t)canon_any(r[0]), (int32_t)canon_any(r[1]), (int32_t)canon_any(r[2]),
                                                             (int32_t)canon_any(r[3]));
      }
      return;
    }
    if (coop_rp) {                                // a message per workgroup
        const size_t it = u0 - u_blocks;
        coop::rhoprime_body(reinterpret_cast<uint32_t*>(rp + it * 8), reinterpret_cast<const uint32_t*>(sk + it * key_stride + 32),
                            reinterpret_cast<const uint32_t*>(mu + it * 8));
        if (lane == 0) attempts[it] = 0;
        return;
    }
    const size_t item = (u0 - u_blocks) * 64 + lane;
    if (item >= batch) return;
    const uint64_t* key = reinterpret_cast<const uint64_t*>(sk + item * key_stride + 32);
    Shake<17> sp;
    sp.init();
#pragma unroll
    for (int w = 0; w < 4; w++) sp.s[w] = key[w];
#pragma unroll
    for (int w = 0; w < 8; w++) sp.s[4 + w] = mu[item * 8 + w];
    sp.s[12] = 0x1Full;
    sp.s[16] ^= 0x8000000000000000ull;
    keccak_f1600(sp.s);
#pragma unroll
    for (int w = 0; w < 8; w++) rp[item * 8 + w] = sp.s[w];
    attempts[item] = 0;
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
// t1^[key][k] = NTT(t1[k] 2^13), canonical: one wave per polynomial, read from the packed public key (decoder.v:96-100)
__global__ __launch_bounds__(256) void expand_t1_kernel(int32_t* __restrict__ t1hat, const uint8_t* __restrict__ pk, size_t pk_stride, int K, size_t npolys,
                                                        const uint32_t* __restrict__ fwd_tab)
{
    const int lane = threadIdx.x & 63;
    TwRegs tw;
    tw.load(fwd_tab, lane);
    const X10Dpp lm(lane);
    const PackedLane<10> plt(lane);
    for (size_t u = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); u < npolys; u += (size_t)gridDim.x * 4) {
        const size_t key = u / (size_t)K;
        const int k = (int)(u % (size_t)K);
        uint32_t raw[4], f[4];
        plt.load(raw, pk + key * pk_stride + 32 + (size_t)k * 320);
        plt.fields(f, raw);
        int32_t r[4];
#pragma unroll
        for (int m = 0; m < 4; m++) r[m] = (int32_t)(f[m] << 13);
        ntt_fwd_core(r, tw, lm);
        *reinterpret_cast<int4*>(t1hat + u * 256 + 4 * lane) = make_int4((int32_t)canon_any(r[0]), (int32_t)canon_any(r[1]), (int32_t)canon_any(r[2]),
                                                                         (int32_t)canon_any(r[3]));
    }
}
hipError_t launch_expand_t1(int32_t* t1hat, const uint8_t* pk, size_t pk_stride, int level, size_t nkeys, const Tables& t, hipStream_t s)
{
    if (nkeys == 0) return hipSuccess;
    if (level != 2 && level != 3 && level != 5) return hipErrorInvalidValue;
    const int K = level == 2 ? 4 : level == 3 ? 6 : 8;
    const size_t npolys = nkeys * (size_t)K;
    hipLaunchKernelGGL(expand_t1_kernel, grid_for((npolys + 3) / 4, t.num_cus * 8), 256, 0, s, t1hat, pk, pk_stride, K, npolys, t.fwd);
    return hipGetLastError();
}

template <int LEVEL>
static hipError_t launch_verify_wire_level(uint8_t* w1p, int32_t* verdict, const int32_t* A, const uint8_t* pk, size_t pk_stride,
                                           const uint8_t* sig, size_t sig_stride, const uint32_t* cbits, size_t batch, int shared_pk,
                                           const Tables& t, hipStream_t s, int a_fmt, const int32_t* t1hat)
{
    if (shared_pk && a_fmt != A_I32) return hipErrorInvalidValue;
    if (!cbits && (shared_pk || t1hat || a_fmt != A_I32)) return hipErrorInvalidValue;      // the fused SampleInBall exists in the plain a-key-per-item form only
    if (t1hat && !shared_pk) {                    // keys whose t1^ the caller keeps beside A (a key per item; int32 A only)
        if (a_fmt != A_I32) return hipErrorInvalidValue;
        const int g = grid_for((batch + 3) / 4,
                               t.num_cus * resident_blocks_per_cu(verify_wire_wpi_kernel<LEVEL, A_I32, true>, 256, t.wpi_blocks_per_cu, t.device));
        note_launch("verify_wire_wpi", g, 4, batch);
        hipLaunchKernelGGL((verify_wire_wpi_kernel<LEVEL, A_I32, true>), g, 256, 0, s, w1p, verdict, A, pk, pk_stride, sig, sig_stride, cbits, batch,
                           t.fwd, t.inv_pipe, t1hat);
        return hipGetLastError();
    }
    if (shared_pk) {
        constexpr int NW = WireNW<LEVEL>::N;
        const int g = grid_for((batch + NW - 1) / NW, t.num_cus);
        note_launch("verify_wire_shared", g, NW, batch);
        hipLaunchKernelGGL((verify_wire_shared_kernel<LEVEL, NW>), g, 64 * NW, 0, s, w1p, verdict, A, pk, sig, sig_stride, cbits, batch,
                           t.fwd, t.inv_pipe);
    } else if (a_fmt == A_P24) {
        const int g = grid_for((batch + 3) / 4,
                               t.num_cus * resident_blocks_per_cu(verify_wire_wpi_kernel<LEVEL, A_P24>, 256, t.wpi_blocks_per_cu, t.device));
        note_launch("verify_wire_wpi", g, 4, batch);
        hipLaunchKernelGGL((verify_wire_wpi_kernel<LEVEL, A_P24>), g, 256, 0, s, w1p, verdict, A, pk, pk_stride, sig, sig_stride, cbits, batch,
                           t.fwd, t.inv_pipe, nullptr);
    } else if (!cbits) {                          // SampleInBall inside the kernel
        const int g = grid_for((batch + 3) / 4,
                               t.num_cus * resident_blocks_per_cu(verify_wire_wpi_kernel<LEVEL, A_I32, false, true>, 256, t.wpi_blocks_per_cu, t.device));
        note_launch("verify_wire_wpi", g, 4, batch);
        hipLaunchKernelGGL((verify_wire_wpi_kernel<LEVEL, A_I32, false, true>), g, 256, 0, s, w1p, verdict, A, pk, pk_stride, sig, sig_stride, cbits, batch,
                           t.fwd, t.inv_pipe, nullptr);
    } else {
        const int g = grid_for((batch + 3) / 4,
                               t.num_cus * resident_blocks_per_cu(verify_wire_wpi_kernel<LEVEL, A_I32>, 256, t.wpi_blocks_per_cu, t.device));
        note_launch("verify_wire_wpi", g, 4, batch);
        hipLaunchKernelGGL((verify_wire_wpi_kernel<LEVEL, A_I32>), g, 256, 0, s, w1p, verdict, A, pk, pk_stride, sig, sig_stride, cbits, batch,
                           t.fwd, t.inv_pipe, nullptr);
    }
    return hipGetLastError();
}

hipError_t launch_verify_wire(int level, uint8_t* w1p, int32_t* verdict, const int32_t* A, const uint8_t* pk, size_t pk_stride,
                              const uint8_t* sig, size_t sig_stride, const uint32_t* cbits, size_t batch, int shared_pk,
                              const Tables& t, hipStream_t s, int a_fmt, const int32_t* t1hat)
{
    if (batch == 0) return hipSuccess;
    switch (level) {
    case 2: return launch_verify_wire_level<2>(w1p, verdict, A, pk, pk_stride, sig, sig_stride, cbits, batch, shared_pk, t, s, a_fmt, t1hat);
    case 3: return launch_verify_wire_level<3>(w1p, verdict, A, pk, pk_stride, sig, sig_stride, cbits, batch, shared_pk, t, s, a_fmt, t1hat);
    case 5: return launch_verify_wire_level<5>(w1p, verdict, A, pk, pk_stride, sig, sig_stride, cbits, batch, shared_pk, t, s, a_fmt, t1hat);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_sign_setup(int level, int32_t* A, bool expand_a_here, int32_t* s1h, int32_t* s2h, int32_t* t0h, const uint8_t* sk,
                             size_t nk, uint8_t* rp, int32_t* attempts, const uint8_t* mu, size_t key_stride, size_t batch, const Tables& t,
                             hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    if (level != 2 && level != 3 && level != 5) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(sk) | reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(rp) | key_stride) & 7) return hipErrorInvalidValue;
    const int K = level == 2 ? 4 : level == 3 ? 6 : 8, L = level == 2 ? 4 : level == 3 ? 5 : 7;
    const size_t skb = 96 + (size_t)(L + K) * 32 * (level == 3 ? 4 : 3) + (size_t)K * 416;
    const int coop_a = expand_a_here && coop_wanted(nk * (size_t)(K * L)), coop_rp = coop_wanted(batch);
    const unsigned a_blocks = !expand_a_here ? 0u : coop_a ? (unsigned)(nk * (size_t)(K * L)) : (unsigned)((2 * nk * (size_t)(K * L) + 63) / 64);
    const size_t npoly = nk * (size_t)(L + 2 * K);
    const unsigned u_blocks = (unsigned)std::min<size_t>(npoly, (size_t)t.num_cus * 32);       // 8 waves per SIMD, grid-stride over the polynomials
    const size_t blocks = a_blocks + u_blocks + (coop_rp ? batch : (batch + 63) / 64);
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
#define DIL_SS(LV)                                                                                                                   \
    hipLaunchKernelGGL(sign_setup_kernel<LV>, (unsigned)blocks, 64, 0, s, A, a_blocks, u_blocks, s1h, s2h, t0h, sk, skb, nk,                   \
                       reinterpret_cast<uint64_t*>(rp), attempts, reinterpret_cast<const uint64_t*>(mu), key_stride, batch, t.fwd, coop_a, coop_rp)
    if (level == 2) DIL_SS(2);
    else if (level == 3) DIL_SS(3);
    else DIL_SS(5);
#undef DIL_SS
    return hipGetLastError();
}

hipError_t launch_expand_a_sib(int32_t* A, const uint8_t* rho, size_t rho_stride_bytes, size_t nkeys, uint32_t* cbits, const uint8_t* ctilde,
                               size_t ct_stride, int level, size_t nitems, hipStream_t s)
{
    if (nitems == 0 || nkeys == 0) return hipSuccess;
    if (level != 2 && level != 3 && level != 5) return hipErrorInvalidValue;
    if ((rho_stride_bytes & 7) || (reinterpret_cast<uintptr_t>(rho) & 7)) return hipErrorInvalidValue;
    const int K = level == 2 ? 4 : level == 3 ? 6 : 8, L = level == 2 ? 4 : level == 3 ? 5 : 7;
    const int tau = level == 2 ? 39 : level == 3 ? 49 : 60;
    if (coop_wanted(nkeys * (size_t)(K * L)) && coop_wanted_sib(nitems))       // both jobs one sponge per wavefront
        return launch_coop_expand_a_sib(A, rho, rho_stride_bytes, nkeys, K, L, cbits, ctilde, ct_stride, tau, nitems, s);
    // (many signatures under few keys: the matrix still takes the one-sponge-per-wavefront form -- 17 instead of 47 us for one key --
    //  beside SampleInBall's lane-per-item workgroups)
    const int coop_a = coop_wanted(nkeys * (size_t)(K * L));
    const unsigned a_blocks = coop_a ? (unsigned)(nkeys * (size_t)(K * L)) : (unsigned)((2 * nkeys * (size_t)(K * L) + 63) / 64);
    const unsigned c_blocks = (unsigned)((nitems + 63) / 64);
    hipLaunchKernelGGL(expand_a_sib_kernel, a_blocks + c_blocks, 64, 0, s, A, reinterpret_cast<const uint64_t*>(rho), rho_stride_bytes / 8, K, L,
                       nkeys, a_blocks, cbits, ctilde, ct_stride, tau, nitems, coop_a);
    return hipGetLastError();
}

hipError_t launch_sample_in_ball_bits(uint32_t* cbits, const uint8_t* ctilde, size_t ct_stride, int level, size_t nitems, hipStream_t s)
{
    if (nitems == 0) return hipSuccess;
    if (level != 2 && level != 3 && level != 5) return hipErrorInvalidValue;
    const int tau = level == 2 ? 39 : level == 3 ? 49 : 60;
    if (coop_wanted_sib(nitems)) return launch_coop_sample_in_ball(nullptr, cbits, ctilde, ct_stride, tau, nitems, s);
    hipLaunchKernelGGL(sample_in_ball_bits_kernel, (int)((nitems + 63) / 64), 64, 0, s, cbits, ctilde, ct_stride, tau, nitems);
    return hipGetLastError();
}

}  // namespace dil
