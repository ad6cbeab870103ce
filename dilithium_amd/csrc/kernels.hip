// kernels.hip -- hand-written CDNA4 (gfx950) kernels for the Dilithium NTT hot path.
//
// One 64-lane wavefront owns one polynomial (256 x int32 = 1 KiB): 4 coefficients per lane,
// butterflies in VGPRs, exchanges by cross-lane ops (ntt_core.hpp), signed Montgomery
// arithmetic on 32-bit multiplies (modarith.hpp).  No MFMA: this is 32-bit integer work bounded by HBM bandwidth
// and VALU issue.  Every polynomial crosses HBM exactly once per kernel.
//
// Reference behaviour implemented (file:line relative to GMUCERG/Dilithium):
//   ntt / invntt / pointwise_barrett          dilithium-256/reference_code/ref_ntt.cpp:28-87
//   ntt2x2_ref / invntt2x2_ref (same maps)    reference_code/ref_ntt2x2.cpp:37-145
//   ntt2x2_fwdntt / _invntt / _mul on `bram`  hardware_code/ntt2x2_{fwdntt,invntt,mul}.cpp,
//                                             address_encoder_decoder.cpp:34-55
//   butterfly op set FWD/INV/MAC/ADD/SUB      rtl_src/butterfly.v:27-250, butterfly2x2.v
//   verify core, mat-vec, sign inner loop     rtl_src/combined_top.v:1207-1469, :1850-1933,
//                                             :1946-2229; decompose/usehint/makehint/norm
//                                             coeff_decomposer.v, usehint.v:140-159,
//                                             makehint.v:98-99, norm_check.v:84-105
// Outputs are canonical residues in [0, q) (the RTL's convention, butterfly.v:194-195); the
// reference C++ returns (-q, q) and compares canonically (util.cpp:98-112).
#include "kernels.hpp"
#include "ntt_core.hpp"

namespace dil {

// ---------------------------------------------------------------------------------------
// address translation of the hardware model's `bram` (address_encoder_decoder.cpp:34-55)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int resolve_row(int mapping, int addr)
{
    if (mapping == MAP_AFTER_INVNTT) return (addr & 15) * 4 + (addr >> 4);
    if (mapping == MAP_AFTER_NTT) return (addr & 3) * 16 + (addr >> 2);
    return addr;
}

// LAYOUT = LAYOUT_POLY : plain data_t[256] in reference order (ref_ntt.h API)
// LAYOUT = LAYOUT_BRAM : `bram` rows behind `mapping`; the transform leaves its output rows at
//                        the model's post-transform permutation (ntt2x2_test.cpp:55,76,129-132)
template <int LAYOUT>
__device__ __forceinline__ int fwd_in_off(int i, int mapping)
{
    if (LAYOUT == LAYOUT_POLY) return i;
    return 4 * resolve_row(mapping, i >> 2) + (i & 3);
}
template <int LAYOUT>
__device__ __forceinline__ int fwd_out_row_off(int row, int mapping)
{
    if (LAYOUT == LAYOUT_POLY) return 4 * row;
    return 4 * resolve_row(mapping, resolve_row(MAP_AFTER_NTT, row));
}
template <int LAYOUT>
__device__ __forceinline__ int inv_in_row_off(int row, int mapping)
{
    if (LAYOUT == LAYOUT_POLY) return 4 * row;
    return 4 * resolve_row(mapping, row);
}
template <int LAYOUT>
__device__ __forceinline__ int inv_out_off(int i, int mapping)
{
    if (LAYOUT == LAYOUT_POLY) return i;
    return 4 * resolve_row(mapping, resolve_row(MAP_AFTER_INVNTT, i >> 2)) + (i & 3);
}

// Streaming accesses use the non-temporal cache policy: every polynomial is touched exactly
// once per kernel, and measured on MI355X nt loads + stores lift the in-place 1 KiB-in /
// 1 KiB-out stream from 4.7 to 5.2 TB/s (profiles/r01_tune_ntt.txt).
__device__ __forceinline__ int32_t ld_nt(const int32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_nt(int32_t* p, int32_t v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ int4 ld_nt4(const int32_t* p)
{
    int4 v;
    v.x = __builtin_nontemporal_load(p);
    v.y = __builtin_nontemporal_load(p + 1);
    v.z = __builtin_nontemporal_load(p + 2);
    v.w = __builtin_nontemporal_load(p + 3);
    return v;      // hipcc merges the four into one global_load_dwordx4 ... nt
}
__device__ __forceinline__ void st_nt4(int32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    __builtin_nontemporal_store((int32_t)a, p);
    __builtin_nontemporal_store((int32_t)b, p + 1);
    __builtin_nontemporal_store((int32_t)c, p + 2);
    __builtin_nontemporal_store((int32_t)d, p + 3);
}

// ---------------------------------------------------------------------------------------
// H2/H5/H6 forward NTT, batched, in place.  Persistent waves, grid-stride over polynomials,
// the next polynomial's loads are issued before the current one is transformed.
// HBM traffic: 1 KiB in (4 coalesced 256-B dword loads per wave) + 1 KiB out (one 1-KiB
// dwordx4 store per wave) per polynomial.
// ---------------------------------------------------------------------------------------
template <int LAYOUT>
__global__ __launch_bounds__(256) void ntt_fwd_kernel(int32_t* __restrict__ polys, size_t batch,
                                                       const uint32_t* __restrict__ tw_tab, int mapping)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    if (wave >= batch) return;
    int off[4];
#pragma unroll
    for (int m = 0; m < 4; m++) off[m] = fwd_in_off<LAYOUT>(lane + 64 * m, mapping);
    const int out_off = fwd_out_row_off<LAYOUT>(lane, mapping);
    int32_t nxt[4];
#pragma unroll
    for (int m = 0; m < 4; m++) nxt[m] = ld_nt(polys + wave * 256 + off[m]);
    TwRegs tw;
    tw.load(tw_tab, lane);
    const LaneMasks lm(lane);
    for (size_t p = wave; p < batch; p += nwaves) {
        int32_t r[4] = {nxt[0], nxt[1], nxt[2], nxt[3]};
        const size_t pn = p + nwaves;
        if (pn < batch) {
#pragma unroll
            for (int m = 0; m < 4; m++) nxt[m] = ld_nt(polys + pn * 256 + off[m]);
        }
        ntt_fwd_core(r, tw, lm);
        st_nt4(polys + p * 256 + out_off, canon_any(r[0]), canon_any(r[1]), canon_any(r[2]), canon_any(r[3]));
    }
}

// H3/H5/H6 inverse NTT (x 256^-1), batched, in place.  Inputs in (-q, q) (or canonical).
template <int LAYOUT>
__global__ __launch_bounds__(256) void ntt_inv_kernel(int32_t* __restrict__ polys, size_t batch,
                                                       const uint32_t* __restrict__ tw_tab, int mapping)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    if (wave >= batch) return;
    const int in_off = inv_in_row_off<LAYOUT>(lane, mapping);
    int off[4];
#pragma unroll
    for (int m = 0; m < 4; m++) off[m] = inv_out_off<LAYOUT>(lane + 64 * m, mapping);
    int4 nxt = ld_nt4(polys + wave * 256 + in_off);
    TwRegs tw;
    tw.load(tw_tab, lane);
    const LaneMasks lm(lane);
    for (size_t p = wave; p < batch; p += nwaves) {
        int32_t r[4] = {nxt.x, nxt.y, nxt.z, nxt.w};
        const size_t pn = p + nwaves;
        if (pn < batch) nxt = ld_nt4(polys + pn * 256 + in_off);
        ntt_inv_core(r, tw, lm);
#pragma unroll
        for (int m = 0; m < 4; m++) st_nt(polys + p * 256 + off[m], (int32_t)canon_small(r[m]));
    }
}

// ---------------------------------------------------------------------------------------
// H4 / butterfly.v MULT / ADD / SUB modes: element-wise ops on whole polynomials.
// 4 coefficients (16 B) per thread, grid-stride.  c may alias a (ntt2x2_test.cpp:102).
//   OP_MUL: c = a*b        OP_MAC: c = acc + a*b        OP_ADD: c = a+b       OP_SUB: c = a-b
// ---------------------------------------------------------------------------------------
// true product of two residues given as any int32 with |a|,|b| < 2^31 / ... (here: < 2^24):
// Montgomery-reduce the 64-bit product, then multiply by 2^64 mod q to cancel the 2^-32.
__device__ __forceinline__ int32_t mulmod_true(int32_t a, int32_t b) { return mont_tw(mont_mul(a, b), R2_WT, R2_WQ); }

template <int OP>
__global__ __launch_bounds__(256) void pointwise_kernel(int32_t* c, const int32_t* a, const int32_t* b,
                                                         const int32_t* acc, size_t nvec4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec4; i += stride) {
        const int4 va = reinterpret_cast<const int4*>(a)[i];
        const int4 vb = reinterpret_cast<const int4*>(b)[i];
        const int32_t x[4] = {va.x, va.y, va.z, va.w};
        const int32_t y[4] = {vb.x, vb.y, vb.z, vb.w};
        uint32_t o[4];
        if (OP == OP_MUL) {
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = canon_small(mulmod_true(x[k], y[k]));
        } else if (OP == OP_MAC) {
            const int4 vc = reinterpret_cast<const int4*>(acc)[i];
            const int32_t z[4] = {vc.x, vc.y, vc.z, vc.w};
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = canon_any(z[k] + mulmod_true(x[k], y[k]));
        } else if (OP == OP_ADD) {
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = canon_any(x[k] + y[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = canon_any(x[k] - y[k]);
        }
        reinterpret_cast<uint4*>(c)[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ntt2x2_mul on `bram` (ntt2x2_mul.cpp:33-59): ram[map(l)][k] *= mul_ram[l][k]; one thread per row
__global__ __launch_bounds__(256) void bram_mul_kernel(int32_t* ram, const int32_t* mul_ram, size_t batch, int mapping)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < batch * 64; i += stride) {
        const size_t p = i >> 6;
        const int l = (int)(i & 63);
        int4* dst = reinterpret_cast<int4*>(ram + p * 256 + 4 * resolve_row(mapping, l));
        const int4 va = *dst;
        const int4 vb = *reinterpret_cast<const int4*>(mul_ram + p * 256 + 4 * l);
        int4 o;
        o.x = (int32_t)canon_small(mulmod_true(va.x, vb.x));
        o.y = (int32_t)canon_small(mulmod_true(va.y, vb.y));
        o.z = (int32_t)canon_small(mulmod_true(va.z, vb.z));
        o.w = (int32_t)canon_small(mulmod_true(va.w, vb.w));
        *dst = o;
    }
}

// ---------------------------------------------------------------------------------------
// Dilithium element-wise tail: Decompose / UseHint / MakeHint / norm checks
// ---------------------------------------------------------------------------------------
template <int LEVEL>
struct Par;
template <>
struct Par<2> {
    static constexpr int K = 4, L = 4, OMEGA = 80, BETA = 78;
    static constexpr int32_t GAMMA1 = 1 << 17, GAMMA2 = (Q - 1) / 88;
};
template <>
struct Par<3> {
    static constexpr int K = 6, L = 5, OMEGA = 55, BETA = 196;
    static constexpr int32_t GAMMA1 = 1 << 19, GAMMA2 = (Q - 1) / 32;
};
template <>
struct Par<5> {
    static constexpr int K = 8, L = 7, OMEGA = 75, BETA = 120;
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
        t ^= (uint32_t)(((int32_t)(43 - t)) >> 31) & t;
    } else {
        t = (t * 1025u + (1u << 21)) >> 22;
        t &= 15;
    }
    int32_t r = (int32_t)a - (int32_t)t * (2 * Par<LEVEL>::GAMMA2);
    r -= (((Q - 1) / 2 - r) >> 31) & Q;
    a1 = t;
    a0 = r;
}

template <int LEVEL>
__device__ __forceinline__ uint32_t use_hint(uint32_t a, uint32_t hint)   // usehint.v:140-159
{
    uint32_t a1;
    int32_t a0;
    decompose<LEVEL>(a, a1, a0);
    if (!hint) return a1;
    if (LEVEL == 2) return (a0 > 0) ? ((a1 == 43) ? 0 : a1 + 1) : ((a1 == 0) ? 43 : a1 - 1);
    return (a0 > 0) ? ((a1 + 1) & 15) : ((a1 - 1) & 15);
}

template <int LEVEL>
__device__ __forceinline__ uint32_t make_hint(uint32_t s, uint32_t a1)   // makehint.v:98-99
{
    constexpr uint32_t G2 = Par<LEVEL>::GAMMA2;
    bool none = (s <= G2) || (s > Q - G2) || (s == Q - G2 && a1 == 0);
    return none ? 0u : 1u;
}

__device__ __forceinline__ bool norm_reject(uint32_t x, uint32_t bound)   // norm_check.v:84-105
{
    return x >= bound && x <= (uint32_t)Q - bound;
}

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
constexpr int LDS_DWORDS = LDS_FLAGS + 4;

__device__ __forceinline__ void stage_tables(uint32_t* lds, const uint32_t* __restrict__ fwd_tab,
                                             const uint32_t* __restrict__ inv_tab)
{
    for (int i = threadIdx.x; i < TW_TABLE_DWORDS / 4; i += blockDim.x) {
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(fwd_tab)[i];
        reinterpret_cast<uint4*>(lds + TW_TABLE_DWORDS)[i] = reinterpret_cast<const uint4*>(inv_tab)[i];
    }
}

// strided load of one polynomial (natural order) into NTT-input registers
__device__ __forceinline__ void load_strided(int32_t (&r)[4], const int32_t* __restrict__ a, int lane)
{
#pragma unroll
    for (int m = 0; m < 4; m++) r[m] = ld_nt(a + lane + 64 * m);
}

template <int L>
struct ARow {
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
};

// acc += sum_l A[k][l] o vhat[l] for the lane's 4 coefficients, as 64-bit integers
template <int L>
__device__ __forceinline__ void mac_row(int64_t (&acc)[4], const ARow<L>& A, const uint32_t* vec_lds, int lane)
{
#pragma unroll
    for (int l = 0; l < L; l++) {
        const int4 z = *reinterpret_cast<const int4*>(vec_lds + l * 256 + 4 * lane);
        acc[0] += (int64_t)A.v[l].x * z.x;
        acc[1] += (int64_t)A.v[l].y * z.y;
        acc[2] += (int64_t)A.v[l].z * z.z;
        acc[3] += (int64_t)A.v[l].w * z.w;
    }
}

// H9 mat-vec  w = INTT(A o NTT(y))   (OUT_W)   and sign phase 1 = mat-vec + Decompose (OUT_W1W0)
template <int K, int L, int LEVEL, int OUT>
__global__ __launch_bounds__(64 * (K > L ? K : L)) void matvec_kernel(
    int32_t* __restrict__ w_out, uint8_t* __restrict__ w1_out, int32_t* __restrict__ w0_out,
    const int32_t* __restrict__ A, const int32_t* __restrict__ y, size_t batch, int shared_A,
    const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[LDS_DWORDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    stage_tables(lds, fwd_tab, inv_tab);
    __syncthreads();
    const TwLds twf{lds, lane}, twi{lds + TW_TABLE_DWORDS, lane};
    const LaneMasks lm(lane);
    uint32_t* vec = lds + LDS_VEC;
    for (size_t it = blockIdx.x; it < batch; it += gridDim.x) {
        ARow<L> Ar;
        if (wv < K) Ar.load(A + ((shared_A ? 0 : it * K) + wv) * (size_t)L * 256, lane, !shared_A);
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
            const size_t o = (it * K + wv) * 256;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t v = canon_small(r[m]);
                if (OUT == OUT_W) {
                    w_out[o + lane + 64 * m] = (int32_t)v;
                } else {
                    uint32_t a1;
                    int32_t a0;
                    decompose<LEVEL>(v, a1, a0);
                    w1_out[o + lane + 64 * m] = (uint8_t)a1;
                    w0_out[o + lane + 64 * m] = a0 + ((a0 >> 31) & Q);
                }
            }
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
#pragma unroll
            for (int m = 0; m < 4; m++) hb[m] = h[o + lane + 64 * m];
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
#pragma unroll
            for (int m = 0; m < 4; m++)
                w1_out[o + lane + 64 * m] = (uint8_t)use_hint<LEVEL>(canon_small(r[m]), hb[m]);
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
                  const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    __shared__ __attribute__((aligned(16))) uint32_t lds[LDS_DWORDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    stage_tables(lds, fwd_tab, inv_tab);
    if (threadIdx.x < 4) lds[LDS_FLAGS + threadIdx.x] = 0;
    __syncthreads();
    const TwLds twf{lds, lane}, twi{lds + TW_TABLE_DWORDS, lane};
    const LaneMasks lm(lane);
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
            const int4 s = *reinterpret_cast<const int4*>(s1hat + ((shared_key ? 0 : it * L) + wv) * 256 + 4 * lane);
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
            const size_t ko = ((shared_key ? 0 : it * K) + wv) * 256 + 4 * lane;
            const int4 s2 = *reinterpret_cast<const int4*>(s2hat + ko);
            const int4 t0 = *reinterpret_cast<const int4*>(t0hat + ko);
            int32_t a[4] = {mont_mul(ch.x, s2.x), mont_mul(ch.y, s2.y), mont_mul(ch.z, s2.z), mont_mul(ch.w, s2.w)};
            int32_t b[4] = {mont_mul(ch.x, t0.x), mont_mul(ch.y, t0.y), mont_mul(ch.z, t0.z), mont_mul(ch.w, t0.w)};
            ntt_inv_core(a, twi, lm);
            ntt_inv_core(b, twi, lm);
            const size_t o = (it * K + wv) * 256;
            bool rej1 = false, rej2 = false;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t ct0 = canon_small(b[m]);
                const uint32_t r0 = canon_any(w0[o + lane + 64 * m] - a[m]);
                rej1 |= norm_reject(r0, Par<LEVEL>::GAMMA2 - Par<LEVEL>::BETA);
                rej2 |= norm_reject(ct0, Par<LEVEL>::GAMMA2);
                uint32_t s = r0 + ct0;
                s -= (s >= (uint32_t)Q) ? (uint32_t)Q : 0u;
                const uint32_t hb = make_hint<LEVEL>(s, w1[o + lane + 64 * m]);
                h_out[o + lane + 64 * m] = (uint8_t)hb;
                nh += __popcll(__ballot(hb));
            }
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
// One wavefront carries one whole item through every stage: the L NTT-domain vectors stay in
// its registers (4L VGPRs), the matrix rows stream through, no LDS data exchange and no
// barrier after the one-time twiddle staging.  Rows are software-prefetched: the loads of row
// k+1 are issued as soon as the MACs of row k have consumed the row registers, and fly under
// NTT(t1_k) + INTT(row k).  With batch >= 8 items per SIMD this keeps the VALUs busier than
// the workgroup-per-item kernels above (which remain the low-latency path for small batches).
// ---------------------------------------------------------------------------------------
template <int L>
__device__ __forceinline__ void mac_row_regs(int64_t (&acc)[4], const ARow<L>& A, const int32_t (&vh)[L][4])
{
#pragma unroll
    for (int l = 0; l < L; l++) {
        acc[0] += (int64_t)A.v[l].x * vh[l][0];
        acc[1] += (int64_t)A.v[l].y * vh[l][1];
        acc[2] += (int64_t)A.v[l].z * vh[l][2];
        acc[3] += (int64_t)A.v[l].w * vh[l][3];
    }
}

// forward-transform L consecutive polynomials (+ optionally one extra from `tail`) into
// registers, loading polynomial l+1 while l is being transformed
template <int L, bool TAIL, class TW>
__device__ __forceinline__ void fwd_vector(int32_t (&vh)[L][4], int32_t (&th)[4], const int32_t* __restrict__ v,
                                           const int32_t* __restrict__ tail, const TW& twf, const LaneMasks& lm, int lane)
{
    int32_t cur[4], nxt[4] = {0, 0, 0, 0};
    load_strided(cur, v, lane);
#pragma unroll
    for (int l = 0; l < L; l++) {
        if (l + 1 < L) load_strided(nxt, v + (l + 1) * 256, lane);
        else if (TAIL) load_strided(nxt, tail, lane);
        ntt_fwd_core(cur, twf, lm);
#pragma unroll
        for (int m = 0; m < 4; m++) { vh[l][m] = cur[m]; cur[m] = nxt[m]; }
    }
    if (TAIL) {
#pragma unroll
        for (int m = 0; m < 4; m++) th[m] = cur[m];     // loaded, NOT yet transformed
    }
}

template <int K, int L, int LEVEL, int OUT>
__global__ __launch_bounds__(256) void matvec_wpi_kernel(
    int32_t* __restrict__ w_out, uint8_t* __restrict__ w1_out, int32_t* __restrict__ w0_out,
    const int32_t* __restrict__ A, const int32_t* __restrict__ y, size_t batch, int shared_A,
    const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * TW_TABLE_DWORDS];
    const int lane = threadIdx.x & 63;
    stage_tables(lds, fwd_tab, inv_tab);
    __syncthreads();
    const TwLds twf{lds, lane}, twi{lds + TW_TABLE_DWORDS, lane};
    const LaneMasks lm(lane);
    const size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t it = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < batch; it += nwaves) {
        const int32_t* Ait = A + (shared_A ? 0 : it * K) * (size_t)L * 256;
        ARow<L> Ar;
        Ar.load(Ait, lane, !shared_A);
        int32_t vh[L][4], dummy[4];
        fwd_vector<L, false>(vh, dummy, y + it * L * 256, nullptr, twf, lm, lane);
        for (int k = 0; k < K; k++) {
            int64_t acc[4] = {0, 0, 0, 0};
            mac_row_regs<L>(acc, Ar, vh);
            if (k + 1 < K) Ar.load(Ait + (size_t)(k + 1) * L * 256, lane, !shared_A);
            int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
            ntt_inv_core(r, twi, lm);
            const size_t o = (it * K + k) * 256;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t v = canon_small(r[m]);
                if (OUT == OUT_W) {
                    st_nt(w_out + o + lane + 64 * m, (int32_t)v);
                } else {
                    uint32_t a1;
                    int32_t a0;
                    decompose<LEVEL>(v, a1, a0);
                    w1_out[o + lane + 64 * m] = (uint8_t)a1;
                    st_nt(w0_out + o + lane + 64 * m, a0 + ((a0 >> 31) & Q));
                }
            }
        }
    }
}

#define DIL_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// raw (time-domain) inputs of one item, prefetched a whole row phase ahead
template <int NP>
struct RawPolys {
    int32_t v[NP][4];
    __device__ __forceinline__ void load(const int32_t* __restrict__ base, int lane)
    {
#pragma unroll
        for (int p = 0; p < NP; p++) load_strided(v[p], base + p * 256, lane);
    }
};

// verify, wave-per-item.  Per item:  issue row-0 operand loads | z-phase: L+1 forward NTTs on
// registers that were loaded during the PREVIOUS item's row phase, z^ -> this wave's LDS slice |
// issue the NEXT item's z/c loads | K rows: MAC from LDS, prefetch row k+1, NTT(t1_k), INTT, UseHint.
template <int LEVEL>
__global__ __launch_bounds__(256) void verify_wpi_kernel(
    uint8_t* __restrict__ w1_out, const int32_t* __restrict__ A, const int32_t* __restrict__ z,
    const int32_t* __restrict__ c, const int32_t* __restrict__ t1, const uint8_t* __restrict__ h, size_t batch,
    int shared_pk, const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * TW_TABLE_DWORDS + 4 * L * 256];
    const int lane = threadIdx.x & 63;
    stage_tables(lds, fwd_tab, inv_tab);
    const TwLds twf{lds, lane}, twi{lds + TW_TABLE_DWORDS, lane};
    const LaneMasks lm(lane);
    uint32_t* zl = lds + 2 * TW_TABLE_DWORDS + (threadIdx.x >> 6) * (L * 256);   // this wave's private slice
    const size_t nwaves = (size_t)gridDim.x * 4;
    size_t it = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    RawPolys<L> zr;
    int32_t cr[4] = {0, 0, 0, 0};
    if (it < batch) {
        zr.load(z + it * L * 256, lane);
        load_strided(cr, c + it * 256, lane);
    }
    __syncthreads();                               // tables staged (the only barrier)
    for (; it < batch; it += nwaves) {
        const int32_t* Ait = A + (shared_pk ? 0 : it * K) * (size_t)L * 256;
        const int32_t* t1it = t1 + (shared_pk ? 0 : it * K) * 256;
        const uint8_t* hit = h + it * K * 256;
        // row 0 operands fly under the z-phase
        ARow<L> Ar;
        Ar.load(Ait, lane, !shared_pk);
        int32_t tn[4];
        uint32_t hn[4];
        load_strided(tn, t1it, lane);
#pragma unroll
        for (int m = 0; m < 4; m++) hn[m] = hit[lane + 64 * m];
        // z-phase
#pragma unroll
        for (int l = 0; l < L; l++) {
            ntt_fwd_core(zr.v[l], twf, lm);
            *reinterpret_cast<int4*>(zl + l * 256 + 4 * lane) = make_int4(zr.v[l][0], zr.v[l][1], zr.v[l][2], zr.v[l][3]);
        }
        int32_t ch[4] = {cr[0], cr[1], cr[2], cr[3]};
        ntt_fwd_core(ch, twf, lm);
        DIL_SCHED_FENCE();
        // next item's time-domain inputs: a whole row phase to land
        const size_t itn = it + nwaves;
        if (itn < batch) {
            zr.load(z + itn * L * 256, lane);
            load_strided(cr, c + itn * 256, lane);
        }
        for (int k = 0; k < K; k++) {
            int64_t acc[4] = {0, 0, 0, 0};
            mac_row<L>(acc, Ar, zl, lane);
            int32_t th[4];
            uint32_t hb[4];
#pragma unroll
            for (int m = 0; m < 4; m++) { th[m] = (tn[m] & 0x3FF) << 13; hb[m] = hn[m]; }   // decoder.v:96-100
            if (k + 1 < K) {
                Ar.load(Ait + (size_t)(k + 1) * L * 256, lane, !shared_pk);
                load_strided(tn, t1it + (k + 1) * 256, lane);
#pragma unroll
                for (int m = 0; m < 4; m++) hn[m] = hit[(k + 1) * 256 + lane + 64 * m];
            }
            DIL_SCHED_FENCE();     // keep the stages from being interleaved (register pressure, not ILP, is the limit)
            ntt_fwd_core(th, twf, lm);
            DIL_SCHED_FENCE();
#pragma unroll
            for (int m = 0; m < 4; m++) acc[m] -= (int64_t)ch[m] * th[m];
            int32_t r[4] = {mont_red64(acc[0]), mont_red64(acc[1]), mont_red64(acc[2]), mont_red64(acc[3])};
            DIL_SCHED_FENCE();
            ntt_inv_core(r, twi, lm);
            DIL_SCHED_FENCE();
            const size_t o = (it * K + k) * 256;
#pragma unroll
            for (int m = 0; m < 4; m++)
                w1_out[o + lane + 64 * m] = (uint8_t)use_hint<LEVEL>(canon_small(r[m]), hb[m]);
        }
    }
}

template <int LEVEL>
__global__ __launch_bounds__(256) void sign2_wpi_kernel(
    int32_t* __restrict__ z_out, uint8_t* __restrict__ h_out, int32_t* __restrict__ flags_out,
    const int32_t* __restrict__ c, const int32_t* __restrict__ y, const int32_t* __restrict__ w0,
    const uint8_t* __restrict__ w1, const int32_t* __restrict__ s1hat, const int32_t* __restrict__ s2hat,
    const int32_t* __restrict__ t0hat, size_t batch, int shared_key, const uint32_t* __restrict__ fwd_tab,
    const uint32_t* __restrict__ inv_tab)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * TW_TABLE_DWORDS];
    const int lane = threadIdx.x & 63;
    stage_tables(lds, fwd_tab, inv_tab);
    __syncthreads();
    const TwLds twf{lds, lane}, twi{lds + TW_TABLE_DWORDS, lane};
    const LaneMasks lm(lane);
    const size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t it = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < batch; it += nwaves) {
        const int32_t* s1 = s1hat + (shared_key ? 0 : it * L) * 256;
        const int32_t* s2 = s2hat + (shared_key ? 0 : it * K) * 256;
        const int32_t* t0 = t0hat + (shared_key ? 0 : it * K) * 256;
        int32_t ch[4];
        load_strided(ch, c + it * 256, lane);
        int4 sn = *reinterpret_cast<const int4*>(s1 + 4 * lane);
        ntt_fwd_core(ch, twf, lm);
        uint32_t bits = 0, nh = 0;
        for (int l = 0; l < L; l++) {
            const int4 s = sn;
            const size_t o = (it * L + l) * 256;
            int32_t yv[4];
            load_strided(yv, y + o, lane);
            if (l + 1 < L) sn = *reinterpret_cast<const int4*>(s1 + (l + 1) * 256 + 4 * lane);
            int32_t r[4] = {mont_mul(ch[0], s.x), mont_mul(ch[1], s.y), mont_mul(ch[2], s.z), mont_mul(ch[3], s.w)};
            ntt_inv_core(r, twi, lm);
            bool rej = false;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t v = canon_any(r[m] + yv[m]);
                rej |= norm_reject(v, Par<LEVEL>::GAMMA1 - Par<LEVEL>::BETA);
                st_nt(z_out + o + lane + 64 * m, (int32_t)v);
            }
            if (__ballot(rej)) bits |= 1;
        }
        for (int k = 0; k < K; k++) {
            const int4 a2 = *reinterpret_cast<const int4*>(s2 + k * 256 + 4 * lane);
            const int4 b0 = *reinterpret_cast<const int4*>(t0 + k * 256 + 4 * lane);
            const size_t o = (it * K + k) * 256;
            int32_t wv0[4];
            uint32_t wv1[4];
            load_strided(wv0, w0 + o, lane);
#pragma unroll
            for (int m = 0; m < 4; m++) wv1[m] = w1[o + lane + 64 * m];
            int32_t a[4] = {mont_mul(ch[0], a2.x), mont_mul(ch[1], a2.y), mont_mul(ch[2], a2.z), mont_mul(ch[3], a2.w)};
            int32_t b[4] = {mont_mul(ch[0], b0.x), mont_mul(ch[1], b0.y), mont_mul(ch[2], b0.z), mont_mul(ch[3], b0.w)};
            ntt_inv_core(a, twi, lm);
            ntt_inv_core(b, twi, lm);
            bool rej1 = false, rej2 = false;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t ct0 = canon_small(b[m]);
                const uint32_t r0 = canon_any(wv0[m] - a[m]);
                rej1 |= norm_reject(r0, Par<LEVEL>::GAMMA2 - Par<LEVEL>::BETA);
                rej2 |= norm_reject(ct0, Par<LEVEL>::GAMMA2);
                uint32_t s = r0 + ct0;
                s -= (s >= (uint32_t)Q) ? (uint32_t)Q : 0u;
                const uint32_t hb = make_hint<LEVEL>(s, wv1[m]);
                h_out[o + lane + 64 * m] = (uint8_t)hb;
                nh += __popcll(__ballot(hb));
            }
            if (__ballot(rej1)) bits |= 2;
            if (__ballot(rej2)) bits |= 4;
        }
        if (lane == 0) flags_out[it] = (int32_t)(bits | (nh > (uint32_t)Par<LEVEL>::OMEGA ? 8u : 0u));
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
// resident blocks per CU of a kernel (occupancy API, cached per kernel): persistent grids are
// sized to what is actually co-resident so that no block waits for another to retire
template <class KernelT>
static int resident_blocks_per_cu(KernelT kernel, int block_threads, int cap)
{
    static int cached = 0;          // one instance per KernelT instantiation... but KernelT is a type:
    static const void* cached_for = nullptr;
    const void* key = reinterpret_cast<const void*>(kernel);
    if (cached_for != key) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block_threads, 0) != hipSuccess || n < 1) n = 1;
        cached = n;
        cached_for = key;
    }
    return cached < cap ? cached : cap;
}

static inline int grid_for(size_t work_blocks, int max_blocks)
{
    if (work_blocks < 1) work_blocks = 1;
    return (int)(work_blocks < (size_t)max_blocks ? work_blocks : (size_t)max_blocks);
}

hipError_t launch_ntt(bool inverse, int layout, int mapping, int32_t* polys, size_t batch,
                      const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    const int grid = grid_for((batch + 3) / 4, t.num_cus * t.ntt_blocks_per_cu);
    const uint32_t* tab = inverse ? t.inv : t.fwd;   // standalone flavour of the inverse table
    if (!inverse) {
        if (layout == LAYOUT_POLY) hipLaunchKernelGGL(ntt_fwd_kernel<LAYOUT_POLY>, grid, 256, 0, s, polys, batch, tab, mapping);
        else hipLaunchKernelGGL(ntt_fwd_kernel<LAYOUT_BRAM>, grid, 256, 0, s, polys, batch, tab, mapping);
    } else {
        if (layout == LAYOUT_POLY) hipLaunchKernelGGL(ntt_inv_kernel<LAYOUT_POLY>, grid, 256, 0, s, polys, batch, tab, mapping);
        else hipLaunchKernelGGL(ntt_inv_kernel<LAYOUT_BRAM>, grid, 256, 0, s, polys, batch, tab, mapping);
    }
    return hipGetLastError();
}

hipError_t launch_pointwise(int op, int32_t* c, const int32_t* a, const int32_t* b, const int32_t* acc,
                            size_t batch, const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    const size_t nvec4 = batch * 64;
    const int grid = grid_for((nvec4 + 255) / 256, t.num_cus * 8);
    switch (op) {
    case OP_MUL: hipLaunchKernelGGL(pointwise_kernel<OP_MUL>, grid, 256, 0, s, c, a, b, acc, nvec4); break;
    case OP_MAC: hipLaunchKernelGGL(pointwise_kernel<OP_MAC>, grid, 256, 0, s, c, a, b, acc, nvec4); break;
    case OP_ADD: hipLaunchKernelGGL(pointwise_kernel<OP_ADD>, grid, 256, 0, s, c, a, b, acc, nvec4); break;
    case OP_SUB: hipLaunchKernelGGL(pointwise_kernel<OP_SUB>, grid, 256, 0, s, c, a, b, acc, nvec4); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_bram_mul(int32_t* ram, const int32_t* mul_ram, size_t batch, int mapping,
                           const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    const int grid = grid_for((batch * 64 + 255) / 256, t.num_cus * 8);
    hipLaunchKernelGGL(bram_mul_kernel, grid, 256, 0, s, ram, mul_ram, batch, mapping);
    return hipGetLastError();
}

// wave-per-item pays once every SIMD has several items to interleave; below that the
// workgroup-per-item kernels expose more parallelism per item (lower latency)
static inline bool use_wpi(size_t batch, const Tables& t)
{
    if (t.fused_mode == 1) return false;
    if (t.fused_mode == 2) return true;
    return batch >= (size_t)t.num_cus * 8;
}

template <int LEVEL, int OUT>
static hipError_t launch_matvec_level(int32_t* w, uint8_t* w1, int32_t* w0, const int32_t* A, const int32_t* y,
                                      size_t batch, int shared_A, const Tables& t, hipStream_t s)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    if (use_wpi(batch, t)) {
        const int g = grid_for((batch + 3) / 4,
                               t.num_cus * resident_blocks_per_cu(matvec_wpi_kernel<K, L, LEVEL, OUT>, 256, t.wpi_blocks_per_cu));
        hipLaunchKernelGGL((matvec_wpi_kernel<K, L, LEVEL, OUT>), g, 256, 0, s, w, w1, w0, A, y, batch, shared_A, t.fwd,
                           t.inv_pipe);
        return hipGetLastError();
    }
    const int grid = grid_for(batch, t.num_cus * t.fused_wgs_per_cu);
    hipLaunchKernelGGL((matvec_kernel<K, L, LEVEL, OUT>), grid, 64 * (K > L ? K : L), 0, s, w, w1, w0, A, y, batch,
                       shared_A, t.fwd, t.inv_pipe);
    return hipGetLastError();
}

hipError_t launch_matvec(int level, int out_mode, int32_t* w, uint8_t* w1, int32_t* w0, const int32_t* A,
                         const int32_t* y, size_t batch, int shared_A, const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
#define DIL_MV(LV)                                                                                   \
    return out_mode == OUT_W ? launch_matvec_level<LV, OUT_W>(w, w1, w0, A, y, batch, shared_A, t, s) \
                             : launch_matvec_level<LV, OUT_W1W0>(w, w1, w0, A, y, batch, shared_A, t, s)
    switch (level) {
    case 2: DIL_MV(2);
    case 3: DIL_MV(3);
    case 5: DIL_MV(5);
    default: return hipErrorInvalidValue;
    }
#undef DIL_MV
}

hipError_t launch_verify(int level, uint8_t* w1, const int32_t* A, const int32_t* z, const int32_t* c,
                         const int32_t* t1, const uint8_t* h, size_t batch, int shared_pk,
                         const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    if (use_wpi(batch, t)) {
        switch (level) {
        case 2: hipLaunchKernelGGL(verify_wpi_kernel<2>, grid_for((batch + 3) / 4, t.num_cus * resident_blocks_per_cu(verify_wpi_kernel<2>, 256, t.wpi_blocks_per_cu)), 256, 0, s, w1, A, z, c, t1, h, batch, shared_pk, t.fwd, t.inv_pipe); break;
        case 3: hipLaunchKernelGGL(verify_wpi_kernel<3>, grid_for((batch + 3) / 4, t.num_cus * resident_blocks_per_cu(verify_wpi_kernel<3>, 256, t.wpi_blocks_per_cu)), 256, 0, s, w1, A, z, c, t1, h, batch, shared_pk, t.fwd, t.inv_pipe); break;
        case 5: hipLaunchKernelGGL(verify_wpi_kernel<5>, grid_for((batch + 3) / 4, t.num_cus * resident_blocks_per_cu(verify_wpi_kernel<5>, 256, t.wpi_blocks_per_cu)), 256, 0, s, w1, A, z, c, t1, h, batch, shared_pk, t.fwd, t.inv_pipe); break;
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
                        const int32_t* t0hat, size_t batch, int shared_key, const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    if (use_wpi(batch, t)) {
        switch (level) {
        case 2: hipLaunchKernelGGL(sign2_wpi_kernel<2>, grid_for((batch + 3) / 4, t.num_cus * resident_blocks_per_cu(sign2_wpi_kernel<2>, 256, t.wpi_blocks_per_cu)), 256, 0, s, z, h, flags, c, y, w0, w1, s1hat, s2hat, t0hat, batch, shared_key, t.fwd, t.inv_pipe); break;
        case 3: hipLaunchKernelGGL(sign2_wpi_kernel<3>, grid_for((batch + 3) / 4, t.num_cus * resident_blocks_per_cu(sign2_wpi_kernel<3>, 256, t.wpi_blocks_per_cu)), 256, 0, s, z, h, flags, c, y, w0, w1, s1hat, s2hat, t0hat, batch, shared_key, t.fwd, t.inv_pipe); break;
        case 5: hipLaunchKernelGGL(sign2_wpi_kernel<5>, grid_for((batch + 3) / 4, t.num_cus * resident_blocks_per_cu(sign2_wpi_kernel<5>, 256, t.wpi_blocks_per_cu)), 256, 0, s, z, h, flags, c, y, w0, w1, s1hat, s2hat, t0hat, batch, shared_key, t.fwd, t.inv_pipe); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    const int grid = grid_for(batch, t.num_cus * t.fused_wgs_per_cu);
#define DIL_S2(LV)                                                                                             \
    hipLaunchKernelGGL(sign2_kernel<LV>, grid,                                                                 \
                       64 * (Par<LV>::K > Par<LV>::L + 1 ? Par<LV>::K : Par<LV>::L + 1), 0, s, z, h, flags, c, \
                       y, w0, w1, s1hat, s2hat, t0hat, batch, shared_key, t.fwd, t.inv_pipe);                       \
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
