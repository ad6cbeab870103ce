// kernels.hip -- hand-written CDNA4 (gfx950) kernels for the Dilithium NTT hot path.
//
// One 64-lane wavefront owns one polynomial (256 x int32 = 1 KiB): 4 coefficients per lane,
// butterflies in VGPRs, exchanges by cross-lane ops (ntt_core.hpp), 24-bit full-rate
// multiplies (modarith.hpp).  No MFMA: this is 32-bit integer work bounded by HBM bandwidth
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

// ---------------------------------------------------------------------------------------
// H2/H5/H6 forward NTT, batched, in place.  Persistent waves, grid-stride over polynomials.
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
    TwRegs tw;
    tw.load(tw_tab, lane);
    for (size_t p = wave; p < batch; p += nwaves) {
        int32_t* a = polys + p * 256;
        uint32_t r[4];
#pragma unroll
        for (int m = 0; m < 4; m++) r[m] = (uint32_t)a[fwd_in_off<LAYOUT>(lane + 64 * m, mapping)] + Q;
        ntt_fwd_core(r, tw, lane);
        uint4 o = make_uint4(canon(r[0]), canon(r[1]), canon(r[2]), canon(r[3]));
        *reinterpret_cast<uint4*>(a + fwd_out_row_off<LAYOUT>(lane, mapping)) = o;
    }
}

// H3/H5/H6 inverse NTT (x 256^-1), batched, in place.
template <int LAYOUT>
__global__ __launch_bounds__(256) void ntt_inv_kernel(int32_t* __restrict__ polys, size_t batch,
                                                       const uint32_t* __restrict__ tw_tab, int mapping)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    TwRegs tw;
    tw.load(tw_tab, lane);
    for (size_t p = wave; p < batch; p += nwaves) {
        int32_t* a = polys + p * 256;
        uint4 v = *reinterpret_cast<const uint4*>(a + inv_in_row_off<LAYOUT>(lane, mapping));
        uint32_t r[4] = {red(v.x + Q), red(v.y + Q), red(v.z + Q), red(v.w + Q)};
        ntt_inv_core(r, tw, lane);
#pragma unroll
        for (int m = 0; m < 4; m++) a[inv_out_off<LAYOUT>(lane + 64 * m, mapping)] = (int32_t)csub(r[m]);
    }
}

// ---------------------------------------------------------------------------------------
// H4 / butterfly.v MULT / ADD / SUB modes: element-wise ops on whole polynomials.
// 4 coefficients (16 B) per thread, grid-stride.  c may alias a (ntt2x2_test.cpp:102).
//   OP_MUL: c = a*b        OP_MAC: c = acc + a*b        OP_ADD: c = a+b       OP_SUB: c = a-b
// ---------------------------------------------------------------------------------------
template <int OP>
__global__ __launch_bounds__(256) void pointwise_kernel(int32_t* c, const int32_t* a, const int32_t* b,
                                                         const int32_t* acc, size_t nvec4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec4; i += stride) {
        int4 va = reinterpret_cast<const int4*>(a)[i];
        int4 vb = reinterpret_cast<const int4*>(b)[i];
        uint32_t x[4] = {canon_signed(va.x), canon_signed(va.y), canon_signed(va.z), canon_signed(va.w)};
        uint32_t y[4] = {canon_signed(vb.x), canon_signed(vb.y), canon_signed(vb.z), canon_signed(vb.w)};
        uint32_t o[4];
        if (OP == OP_MUL) {
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = mulmod(x[k], y[k]);
        } else if (OP == OP_MAC) {
            int4 vc = reinterpret_cast<const int4*>(acc)[i];
            uint32_t z[4] = {canon_signed(vc.x), canon_signed(vc.y), canon_signed(vc.z), canon_signed(vc.w)};
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = canon(z[k] + mulmod_lazy(x[k], y[k]));
        } else if (OP == OP_ADD) {
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = csub(x[k] + y[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = csub(x[k] + Q - y[k]);
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
        int4 va = *dst;
        int4 vb = *reinterpret_cast<const int4*>(mul_ram + p * 256 + 4 * l);
        int4 o;
        o.x = (int32_t)mulmod(canon_signed(va.x), canon_signed(vb.x));
        o.y = (int32_t)mulmod(canon_signed(va.y), canon_signed(vb.y));
        o.z = (int32_t)mulmod(canon_signed(va.z), canon_signed(vb.z));
        o.w = (int32_t)mulmod(canon_signed(va.w), canon_signed(vb.w));
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
        t = (mul24(t, 11275) + (1u << 23)) >> 24;
        t ^= (uint32_t)(((int32_t)(43 - t)) >> 31) & t;
    } else {
        t = (mul24(t, 1025) + (1u << 21)) >> 22;
        t &= 15;
    }
    int32_t r = (int32_t)a - (int32_t)mul24(t, 2 * Par<LEVEL>::GAMMA2);
    r -= (((int32_t)(Q - 1) / 2 - r) >> 31) & (int32_t)Q;
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
    return x >= bound && x <= Q - bound;
}

// ---------------------------------------------------------------------------------------
// Fused pipelines.  One workgroup per item (signature / verification), one wave per
// polynomial row; NTT-domain vectors shared through LDS; twiddles LDS-resident.
// LDS map (dwords): [0,2048) fwd twiddles | [2048,4096) inv twiddles | [4096, 4096+8*256) vec
// | chat[256] | flags[4]
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

// strided load of one polynomial (natural order, any int32 in [-q, 2^31)) into NTT-input regs
__device__ __forceinline__ void load_strided(uint32_t (&r)[4], const int32_t* __restrict__ a, int lane)
{
#pragma unroll
    for (int m = 0; m < 4; m++) r[m] = (uint32_t)a[lane + 64 * m] + Q;
}

// accumulate sum_l A[k][l] o vhat[l] for the lane's 4 coefficients, lazily (each term < 2q)
template <int L>
__device__ __forceinline__ void mac_row(uint32_t (&acc)[4], const int32_t* __restrict__ Arow,
                                        const uint32_t* vec_lds, int lane)
{
    uint4 av[L];
#pragma unroll
    for (int l = 0; l < L; l++) av[l] = *reinterpret_cast<const uint4*>(Arow + l * 256 + 4 * lane);
#pragma unroll
    for (int l = 0; l < L; l++) {
        uint4 z = *reinterpret_cast<const uint4*>(vec_lds + l * 256 + 4 * lane);
        acc[0] += mulmod_lazy(av[l].x, z.x);
        acc[1] += mulmod_lazy(av[l].y, z.y);
        acc[2] += mulmod_lazy(av[l].z, z.z);
        acc[3] += mulmod_lazy(av[l].w, z.w);
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
    uint32_t* vec = lds + LDS_VEC;
    for (size_t it = blockIdx.x; it < batch; it += gridDim.x) {
        if (wv < L) {
            uint32_t r[4];
            load_strided(r, y + (it * L + wv) * 256, lane);
            ntt_fwd_core(r, twf, lane);
            *reinterpret_cast<uint4*>(vec + wv * 256 + 4 * lane) =
                make_uint4(canon(r[0]), canon(r[1]), canon(r[2]), canon(r[3]));
        }
        __syncthreads();
        if (wv < K) {
            const int32_t* Arow = A + ((shared_A ? 0 : it * K) + wv) * (size_t)L * 256;
            uint32_t acc[4] = {0, 0, 0, 0};
            mac_row<L>(acc, Arow, vec, lane);
#pragma unroll
            for (int m = 0; m < 4; m++) acc[m] = red(acc[m]);
            ntt_inv_core(acc, twi, lane);
            const size_t o = (it * K + wv) * 256;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                uint32_t v = csub(acc[m]);
                if (OUT == OUT_W) {
                    w_out[o + lane + 64 * m] = (int32_t)v;
                } else {
                    uint32_t a1;
                    int32_t a0;
                    decompose<LEVEL>(v, a1, a0);
                    w1_out[o + lane + 64 * m] = (uint8_t)a1;
                    w0_out[o + lane + 64 * m] = a0 < 0 ? a0 + (int32_t)Q : a0;
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
    uint32_t* vec = lds + LDS_VEC;
    uint32_t* chat = lds + LDS_CHAT;
    for (size_t it = blockIdx.x; it < batch; it += gridDim.x) {
        if (wv <= L) {   // waves 0..L-1: z_l ; wave L: c
            uint32_t r[4];
            const int32_t* src = (wv < L) ? z + (it * L + wv) * 256 : c + it * 256;
            load_strided(r, src, lane);
            ntt_fwd_core(r, twf, lane);
            uint32_t* dst = (wv < L) ? vec + wv * 256 : chat;
            *reinterpret_cast<uint4*>(dst + 4 * lane) =
                make_uint4(canon(r[0]), canon(r[1]), canon(r[2]), canon(r[3]));
        }
        uint32_t th[4] = {0, 0, 0, 0};
        if (wv < K) {    // t1_k * 2^13 (decoder.v:96-100), t1 is 10 bits
            const int32_t* src = t1 + ((shared_pk ? 0 : it * K) + wv) * 256;
#pragma unroll
            for (int m = 0; m < 4; m++) th[m] = ((uint32_t)src[lane + 64 * m] & 0x3FFu) << 13;
            ntt_fwd_core(th, twf, lane);
#pragma unroll
            for (int m = 0; m < 4; m++) th[m] = canon(th[m]);
        }
        __syncthreads();
        if (wv < K) {
            const int32_t* Arow = A + ((shared_pk ? 0 : it * K) + wv) * (size_t)L * 256;
            uint32_t acc[4] = {0, 0, 0, 0};
            mac_row<L>(acc, Arow, vec, lane);
            uint4 ch = *reinterpret_cast<const uint4*>(chat + 4 * lane);
            acc[0] += Q2 - mulmod_lazy(ch.x, th[0]);
            acc[1] += Q2 - mulmod_lazy(ch.y, th[1]);
            acc[2] += Q2 - mulmod_lazy(ch.z, th[2]);
            acc[3] += Q2 - mulmod_lazy(ch.w, th[3]);
#pragma unroll
            for (int m = 0; m < 4; m++) acc[m] = red(acc[m]);
            ntt_inv_core(acc, twi, lane);
            const size_t o = (it * K + wv) * 256;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                uint32_t hb = h[o + lane + 64 * m];
                w1_out[o + lane + 64 * m] = (uint8_t)use_hint<LEVEL>(csub(acc[m]), hb);
            }
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
    uint32_t* chat = lds + LDS_CHAT;
    uint32_t* fl = lds + LDS_FLAGS;   // [0] reject bits, [1] hint count
    for (size_t it = blockIdx.x; it < batch; it += gridDim.x) {
        if (wv == L) {
            uint32_t r[4];
            load_strided(r, c + it * 256, lane);
            ntt_fwd_core(r, twf, lane);
            *reinterpret_cast<uint4*>(chat + 4 * lane) =
                make_uint4(canon(r[0]), canon(r[1]), canon(r[2]), canon(r[3]));
        }
        __syncthreads();
        const uint4 ch = *reinterpret_cast<const uint4*>(chat + 4 * lane);
        uint32_t bits = 0, nh = 0;
        if (wv < L) {
            const uint4 s = *reinterpret_cast<const uint4*>(s1hat + ((shared_key ? 0 : it * L) + wv) * 256 + 4 * lane);
            uint32_t r[4] = {mulmod_lazy(ch.x, s.x), mulmod_lazy(ch.y, s.y), mulmod_lazy(ch.z, s.z), mulmod_lazy(ch.w, s.w)};
            ntt_inv_core(r, twi, lane);
            const size_t o = (it * L + wv) * 256;
            bool rej = false;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                uint32_t v = csub(csub(r[m]) + canon_signed(y[o + lane + 64 * m]));
                rej |= norm_reject(v, Par<LEVEL>::GAMMA1 - Par<LEVEL>::BETA);
                z_out[o + lane + 64 * m] = (int32_t)v;
            }
            if (__ballot(rej)) bits |= 1;
        }
        if (wv < K) {
            const size_t ko = ((shared_key ? 0 : it * K) + wv) * 256 + 4 * lane;
            const uint4 s2 = *reinterpret_cast<const uint4*>(s2hat + ko);
            const uint4 t0 = *reinterpret_cast<const uint4*>(t0hat + ko);
            uint32_t a[4] = {mulmod_lazy(ch.x, s2.x), mulmod_lazy(ch.y, s2.y), mulmod_lazy(ch.z, s2.z), mulmod_lazy(ch.w, s2.w)};
            uint32_t b[4] = {mulmod_lazy(ch.x, t0.x), mulmod_lazy(ch.y, t0.y), mulmod_lazy(ch.z, t0.z), mulmod_lazy(ch.w, t0.w)};
            ntt_inv_core(a, twi, lane);
            ntt_inv_core(b, twi, lane);
            const size_t o = (it * K + wv) * 256;
            bool rej1 = false, rej2 = false;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                uint32_t cs2 = csub(a[m]), ct0 = csub(b[m]);
                uint32_t r0 = csub(canon_signed(w0[o + lane + 64 * m]) + Q - cs2);
                rej1 |= norm_reject(r0, Par<LEVEL>::GAMMA2 - Par<LEVEL>::BETA);
                rej2 |= norm_reject(ct0, Par<LEVEL>::GAMMA2);
                uint32_t hb = make_hint<LEVEL>(csub(r0 + ct0), w1[o + lane + 64 * m]);
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
// launchers
// ---------------------------------------------------------------------------------------
static inline int grid_for(size_t work_blocks, int max_blocks)
{
    if (work_blocks < 1) work_blocks = 1;
    return (int)(work_blocks < (size_t)max_blocks ? work_blocks : (size_t)max_blocks);
}

hipError_t launch_ntt(bool inverse, int layout, int mapping, int32_t* polys, size_t batch,
                      const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    const int grid = grid_for((batch + 3) / 4, t.num_cus * 8);
    const uint32_t* tab = inverse ? t.inv : t.fwd;
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

template <int LEVEL, int OUT>
static hipError_t launch_matvec_level(int32_t* w, uint8_t* w1, int32_t* w0, const int32_t* A, const int32_t* y,
                                      size_t batch, int shared_A, const Tables& t, hipStream_t s)
{
    constexpr int K = Par<LEVEL>::K, L = Par<LEVEL>::L;
    const int grid = grid_for(batch, t.num_cus * 4);
    hipLaunchKernelGGL((matvec_kernel<K, L, LEVEL, OUT>), grid, 64 * (K > L ? K : L), 0, s, w, w1, w0, A, y, batch,
                       shared_A, t.fwd, t.inv);
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
    const int grid = grid_for(batch, t.num_cus * 4);
#define DIL_VY(LV)                                                                                             \
    hipLaunchKernelGGL(verify_kernel<LV>, grid,                                                                \
                       64 * (Par<LV>::K > Par<LV>::L + 1 ? Par<LV>::K : Par<LV>::L + 1), 0, s, w1, A, z, c, t1, \
                       h, batch, shared_pk, t.fwd, t.inv);                                                     \
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
    const int grid = grid_for(batch, t.num_cus * 4);
#define DIL_S2(LV)                                                                                             \
    hipLaunchKernelGGL(sign2_kernel<LV>, grid,                                                                 \
                       64 * (Par<LV>::K > Par<LV>::L + 1 ? Par<LV>::K : Par<LV>::L + 1), 0, s, z, h, flags, c, \
                       y, w0, w1, s1hat, s2hat, t0hat, batch, shared_key, t.fwd, t.inv);                       \
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
