// ntt_core.hpp -- one n=256 NTT / INTT per 64-lane wavefront, 4 coefficients per lane.
//
// This is the GPU counterpart of the reference's radix-2x2 unit: ref_ntt2x2.cpp:37-82 /
// :100-145 (two layers per pass, 4 passes) == butterfly2x2.v, with the twiddle schedule
// of twiddle_resolver.v:87-130 and the 64-row x 4-coefficient `bram` (config.h:29-36)
// mapped onto 64 lanes x 4 VGPRs.  What the FPGA does with PISO/FIFO reordering between
// rounds (fifo.h:173-229, address_unit.v) is done here with three in-register 4x4
// transposes between the register index and one lane bit-pair:
//     lane bits 5:4  ->  v_permlane32_swap + v_permlane16_swap   (gfx950)
//     lane bits 3:2  ->  DPP row_shr/row_shl with bank masks
//     lane bits 1:0  ->  DPP quad_perm + v_bfi
// No LDS traffic, no barriers inside a transform.
//
// Index bookkeeping (pos = coefficient index, P3..P0 = its four bit-pairs, P3 = bits 7:6):
//   forward  load   lane = (P2,P1,P0) reg = P3      a[lane + 64 m]        (4 coalesced dword loads)
//            pass 0 (layers len 128, 64)            twiddle k1 = 1
//            xchg 5:4 -> lane = (P3,P1,P0) reg = P2 ; pass 1, k1 = 4  + (lane >> 4)
//            xchg 3:2 -> lane = (P3,P2,P0) reg = P1 ; pass 2, k1 = 16 + (lane >> 2)
//            xchg 1:0 -> lane = (P3,P2,P1) reg = P0 ; pass 3, k1 = 64 + lane
//            store  lane j holds out[4j..4j+3]  == one `bram` row        (1 dwordx4 store)
//   inverse runs the same ladder backwards (row load, ..., strided store).
// Twiddles per pass (forward): z[k1], z[2k1], z[2k1+1]  (ref_ntt2x2.cpp:50-55).
#pragma once
#include "modarith.hpp"

namespace dil {

// per-lane twiddles of one pass, Montgomery form: 8 dwords = two 16-byte loads
//   forward: {wa~, waq, wb0~, wb0q, wb1~, wb1q, -, -}
//   inverse: {wa0~, wa0q, wa1~, wa1q, wb~, wbq, f~, fq}   (f = 256^-1 [* 2^32], last pass only)
struct Tw8 {
    uint32_t v[8];
};

// table layout [pass][half][lane][4]: each of the two 16-byte reads is lane-linear (stride 16 B),
// which is conflict-free for ds_read_b128 and one 1-KiB coalesced global_load_dwordx4
constexpr int TW_PASS_STRIDE = 64 * 8;      // dwords per pass
constexpr int TW_TABLE_DWORDS = 4 * TW_PASS_STRIDE;

__device__ __forceinline__ Tw8 load_tw8(const uint32_t* pass_base, int lane)
{
    const uint4 a = *reinterpret_cast<const uint4*>(pass_base + 4 * lane);
    const uint4 b = *reinterpret_cast<const uint4*>(pass_base + 256 + 4 * lane);
    Tw8 t;
    t.v[0] = a.x; t.v[1] = a.y; t.v[2] = a.z; t.v[3] = a.w;
    t.v[4] = b.x; t.v[5] = b.y; t.v[6] = b.z; t.v[7] = b.w;
    return t;
}

// twiddle providers ------------------------------------------------------------------
// registers: the 4 passes' twiddles live in VGPRs for the lifetime of a persistent wave
struct TwRegs {
    Tw8 p[4];
    __device__ __forceinline__ void load(const uint32_t* __restrict__ tab, int lane)
    {
#pragma unroll
        for (int i = 0; i < 4; i++) p[i] = load_tw8(tab + i * TW_PASS_STRIDE, lane);
    }
    template <int PASS>
    __device__ __forceinline__ Tw8 get() const { return p[PASS]; }
};

// LDS: the table is staged once per workgroup; each pass reads its 8 dwords (2 x ds_read_b128,
// lane-linear, conflict-free).  Used by the fused pipelines, where VGPRs are better spent on
// polynomial state.
struct TwLds {
    const uint32_t* tab;   // LDS pointer, [4][2][64][4]
    int lane;
    template <int PASS>
    __device__ __forceinline__ Tw8 get() const { return load_tw8(tab + PASS * TW_PASS_STRIDE, lane); }
};

// LDS, compact (round 4): a pass's twiddles depend on lane >> s only -- forward pass p on lane >> (6 - 2p), inverse pass p on
// lane >> 2p -- so the [pass][half][lane][4] image holds 1 + 4 + 16 + 64 distinct 32-byte entries per direction, not 4 x 64.
// The compact table keeps exactly those: the wave-uniform pass (forward 0, inverse 3) rides in SGPRs as the scalar operand of its
// multiplies (no LDS read at all), the 4- and 16-entry passes are broadcast reads (lanes of one entry read the same 16 bytes:
// no bank conflict), the 64-entry pass is lane-linear as before.  2 x 2688 B instead of 2 x 8 KiB of LDS per workgroup -- what
// lets a second 16-wave workgroup of the shared-key kernels fit a CU (pipelines.hip) -- and 6 instead of 8 ds_read_b128 per transform.
struct TwU {
    uint32_t v[8];        // same slots as Tw8, wave-uniform
};
constexpr int TWC_P4 = 0, TWC_P16 = 32, TWC_P64 = 160;     // dword offsets of the 4-, 16- and 64-entry passes: [half][entry][4]
constexpr int TWC_DWORDS = 672;                             // per direction
template <bool FWD>
struct TwLdsC {
    const uint32_t* tab;   // LDS, this direction's compact table
    int lane;
    TwU u;                 // the uniform pass
    // global_tab: the full [pass][half][lane][4] image of this direction (kernel argument -> scalar loads)
    __device__ __forceinline__ TwLdsC(const uint32_t* lds_tab, const uint32_t* __restrict__ global_tab, int lane_) : tab(lds_tab), lane(lane_)
    {
        const uint32_t* g = global_tab + (FWD ? 0 : 3 * TW_PASS_STRIDE);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            u.v[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)g[i]);
            u.v[4 + i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)g[256 + i]);
        }
    }
    template <int PASS>
    __device__ __forceinline__ auto get() const
    {
        if constexpr (PASS == (FWD ? 0 : 3)) {
            return u;
        } else {
            constexpr int E = FWD ? (PASS == 1 ? 4 : PASS == 2 ? 16 : 64) : (PASS == 0 ? 64 : PASS == 1 ? 16 : 4);
            constexpr int OFF = E == 4 ? TWC_P4 : E == 16 ? TWC_P16 : TWC_P64;
            const int e = E == 64 ? lane : E == 16 ? lane >> 2 : lane >> 4;
            const uint4 a = *reinterpret_cast<const uint4*>(tab + OFF + 4 * e);
            const uint4 b = *reinterpret_cast<const uint4*>(tab + OFF + 4 * E + 4 * e);
            Tw8 t;
            t.v[0] = a.x; t.v[1] = a.y; t.v[2] = a.z; t.v[3] = a.w;
            t.v[4] = b.x; t.v[5] = b.y; t.v[6] = b.z; t.v[7] = b.w;
            return t;
        }
    }
};
// one 16-byte granule of the compact table (g in [0, 168)) <- the full image: which (pass, half, lane) it is
template <bool FWD>
__device__ __forceinline__ int twc_source_granule(int g)
{
    int e_log, h, e;           // entries = 1 << e_log
    if (g < 8) { e_log = 2; h = g >> 2; e = g & 3; }
    else if (g < 40) { e_log = 4; h = (g - 8) >> 4; e = (g - 8) & 15; }
    else { e_log = 6; h = (g - 40) >> 6; e = (g - 40) & 63; }
    const int pass = FWD ? e_log / 2 : 3 - e_log / 2;
    const int lane = e << (6 - e_log);
    return pass * (TW_PASS_STRIDE / 4) + h * 64 + lane;
}

// cross-lane 4x4 transposes -----------------------------------------------------------
// 2x2 step on a register pair (X = index bit 0, Y = index bit 1) against one lane bit:
//   X[lanes with bit = 1]  <->  Y[partner lanes with bit = 0]
__device__ __forceinline__ void swap_b5(int32_t& x, int32_t& y)
{
    auto s = __builtin_amdgcn_permlane32_swap((uint32_t)x, (uint32_t)y, false, false);
    x = (int32_t)s[0];
    y = (int32_t)s[1];
}
__device__ __forceinline__ void swap_b4(int32_t& x, int32_t& y)
{
    auto s = __builtin_amdgcn_permlane16_swap((uint32_t)x, (uint32_t)y, false, false);
    x = (int32_t)s[0];
    y = (int32_t)s[1];
}
template <int M>   // M = 8 (lane bit 3) or 4 (lane bit 2): DPP within a row of 16 lanes
__device__ __forceinline__ void swap_row(int32_t& x, int32_t& y)
{
    constexpr int SHR = 0x110 | M, SHL = 0x100 | M;
    constexpr int BANK1 = (M == 8) ? 0xC : 0xA, BANK0 = (M == 8) ? 0x3 : 0x5;
    const int32_t nx = __builtin_amdgcn_update_dpp(x, y, SHR, 0xF, BANK1, false);
    const int32_t ny = __builtin_amdgcn_update_dpp(y, x, SHL, 0xF, BANK0, false);
    x = nx;
    y = ny;
}
// M = 2 (lane bit 1) or 1 (lane bit 0): DPP quad_perm + bit-field insert under a per-lane mask
// (v_bfi_b32: VCC-form v_cndmask measured 5x slower on gfx950, see modarith.hpp)
__device__ __forceinline__ int32_t bfi(uint32_t mask, int32_t a, int32_t b)   // mask ? a : b, bitwise
{
    int32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(a), "v"(b));
    return d;
}
template <int M>
__device__ __forceinline__ void swap_quad(int32_t& x, int32_t& y, uint32_t mask)
{
    constexpr int PERM = (M == 1) ? 0xB1 : 0x4E;   // quad_perm [1,0,3,2] / [2,3,0,1]
    const int32_t ys = __builtin_amdgcn_mov_dpp(y, PERM, 0xF, 0xF, true);
    const int32_t xs = __builtin_amdgcn_mov_dpp(x, PERM, 0xF, 0xF, true);
    const int32_t nx = bfi(mask, ys, x);
    const int32_t ny = bfi(mask, y, xs);
    x = nx;
    y = ny;
}

// per-lane constants of the exchanges (computed once per wave)
struct LaneMasks {
    uint32_t b1, b0;   // all-ones where lane bit 1 / bit 0 is set
    __device__ __forceinline__ explicit LaneMasks(int lane) : b1(0u - ((lane >> 1) & 1)), b0(0u - (lane & 1)) {}
    __device__ __forceinline__ void operator()(int32_t (&r)[4]) const;     // the in-register (1:0) exchange, below
};

__device__ __forceinline__ void xchg_54(int32_t (&r)[4])
{
    swap_b5(r[0], r[2]);
    swap_b5(r[1], r[3]);
    swap_b4(r[0], r[1]);
    swap_b4(r[2], r[3]);
}
__device__ __forceinline__ void xchg_32(int32_t (&r)[4])
{
    swap_row<8>(r[0], r[2]);
    swap_row<8>(r[1], r[3]);
    swap_row<4>(r[0], r[1]);
    swap_row<4>(r[2], r[3]);
}
__device__ __forceinline__ void xchg_10(int32_t (&r)[4], const LaneMasks& lm)
{
    swap_quad<2>(r[0], r[2], lm.b1);
    swap_quad<2>(r[1], r[3], lm.b1);
    swap_quad<1>(r[0], r[1], lm.b0);
    swap_quad<1>(r[2], r[3], lm.b0);
}

// The transforms take the (1:0) exchange as a policy object `x(r)`:
//   X10Dpp  the in-register form above: 8 DPP moves + 8 v_bfi = 16 half-rate VALU instructions, ~70 issue cycles, a
//           quarter of a transform's exchange + butterfly time (quad-granular DPP has no lane mask, hence the selects);
//   X10Lds  the same 4x4 transpose through a 1 KiB per-wave LDS buffer: one ds_write_b128 + four ds_read_b32, no VALU
//           work at all.  Used by the fused pipelines, whose VALUs are the co-limiter and whose LDS pipe is mostly idle.
//           Lane (q, a) -- q = lane >> 2, a = lane & 3 -- writes its four registers to 16-byte slot lane ^ s,
//           s = (q >> 1) & 3, and reads register a of lanes (q, 0..3): dword 16 q + 4 (b ^ s) + a.  The xor keeps the
//           b128 store conflict-free (a group of 8 consecutive lanes still covers 8 consecutive slots) and spreads the
//           32 lanes of each read over 32 banks (natural slots would be 4-way conflicts).
struct X10Dpp {
    LaneMasks lm;
    __device__ __forceinline__ explicit X10Dpp(int lane) : lm(lane) {}
    __device__ __forceinline__ X10Dpp(uint32_t*, int lane) : lm(lane) {}      // same signature as X10Lds
    __device__ __forceinline__ void operator()(int32_t (&r)[4]) const { xchg_10(r, lm); }
};
struct X10Lds {
    uint32_t* wslot;        // this lane's 16-byte slot in the wave's exchange buffer
    const uint32_t* rd[4];  // the four dwords it reads back
    __device__ __forceinline__ X10Lds(uint32_t* wave_buf /* 256 dwords */, int lane)
    {
        const uint32_t q = (uint32_t)lane >> 2, a = (uint32_t)lane & 3, s = (q >> 1) & 3;
        wslot = wave_buf + 4 * ((uint32_t)lane ^ s);
#pragma unroll
        for (uint32_t b = 0; b < 4; b++) rd[b] = wave_buf + 16 * q + 4 * (b ^ s) + a;
    }
    __device__ __forceinline__ void operator()(int32_t (&r)[4]) const
    {
        *reinterpret_cast<int4*>(wslot) = make_int4(r[0], r[1], r[2], r[3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // same-wave LDS accesses execute in order
#pragma unroll
        for (int b = 0; b < 4; b++) r[b] = (int32_t)rd[b][0];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

__device__ __forceinline__ void LaneMasks::operator()(int32_t (&r)[4]) const { xchg_10(r, *this); }

// All three exchanges through LDS (XAllLds): the VALU-bound pipelines (sign phase 2: 84 % VALU-busy, profiles/r03a_sign_pmc.txt)
// spend 28 % of a transform's issue cycles on the exchanges (4 permlane swaps, 16 DPP moves, 8 v_bfi = ~138 of ~500 cycles);
// through a 1 KiB per-wave buffer each one is 4 ds_write_b32 + 1 ds_read_b128 (about 20 LDS-pipe cycles) and no VALU work at
// all.  Exchange at lane bit-pair position S (0, 2 or 4): lane (hi, b, lo) register m afterwards holds what lane (hi, m, lo)
// register b held.  The writer lane (field value bw) stores its register j to the dword the reader lane (field value j) finds at
// offset bw of its 16-byte slot:  dword = 4 * slot(lane with field <- j) + bw, and every lane reads its own slot.
// Bank arithmetic (MI355X_MICROARCH.md, LDS table): a ds_write_b32 is served in two groups of 32 lanes against 32 banks of 4 bytes,
// a ds_read_b128 in four irregular groups of 16 lanes and a ds_read_b64 in two groups of 32, both against 64 banks.  Round 3's
// slot maps were derived for 64 write banks and measured 2-way on every store (SQ_LDS_BANK_CONFLICT = 77-84 % of the LDS
// instruction cycles of the sign kernels, profiles/r03z_sign_pmc.txt; with these maps 18 % / 27 %, profiles/r04e_sign_pmc.txt -- what
// is left are the byte-plane scratch stores.  In TIME the two sets of maps are within noise of each other, as the guide says of
// 2-way ds_write_b32 conflicts: profiles/r04c_ab_dualN_slots_bf64.txt).  These are conflict-free on both sides:
//   S = 2  lane = (l5, l4, b, lo):  slot = 32 l5 + 16 b1 + 8 b0 + 4 l4 + lo        store bank = 4 (4 l4 + lo) + bw
//   S = 0  lane = (q, b), q = 4 bit: slot = 32 q3 + 16 b0 + 8 b1 + ((q & 7) ^ b0)  store bank = 4 ((q & 7) ^ j0) + bw
//   S = 4  lane = (b, lo4): the 32 lanes of a store group hold only two values of bw, so one 16-byte slot per reader can use at
//          most half the banks.  The reader's four dwords are therefore split over two 8-byte halves in two planes -- writers
//          with bw < 2 fill plane 0, the others plane 1, dword = 128 plane + 2 reader_lane + (bw & 1), store bank = 2 lo4 + bw,
//          -- and read back as two ds_read_b64 (the same 4 LDS cycles as one ds_read_b128).
template <int S>
__device__ __forceinline__ uint32_t xslot(uint32_t lane, uint32_t field)
{
    static_assert(S == 2 || S == 0, "S = 4 uses the two-plane form");
    if (S == 2) return 32 * (lane >> 5) + 16 * (field >> 1) + 8 * (field & 1) + 4 * ((lane >> 4) & 1) + (lane & 3);
    const uint32_t q = lane >> 2;
    return 32 * (q >> 3) + 16 * (field & 1) + 8 * (field >> 1) + ((q & 7) ^ (field & 1));
}
template <int S>
struct XLdsAt {
    static constexpr bool TWO_PLANES = S == 4;
    uint32_t* wr[4];          // where this lane's register j goes
    const uint32_t* rd;       // this lane's 16-byte slot (two planes: its 8 bytes of plane 0; plane 1 is 128 dwords on)
    __device__ __forceinline__ void init(uint32_t* wave_buf /* 256 dwords */, int lane)
    {
        const uint32_t l = (uint32_t)lane, bw = (l >> S) & 3u;
        if constexpr (TWO_PLANES) {
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) wr[j] = wave_buf + 128 * (bw >> 1) + 2 * (16 * j + (l & 15u)) + (bw & 1u);
            rd = wave_buf + 2 * l;
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) wr[j] = wave_buf + 4 * xslot<S>(l, j) + bw;
            rd = wave_buf + 4 * xslot<S>(l, bw);
        }
    }
    __device__ __forceinline__ void operator()(int32_t (&r)[4]) const
    {
#pragma unroll
        for (int j = 0; j < 4; j++) *wr[j] = (uint32_t)r[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // same-wave LDS accesses execute in order
        if constexpr (TWO_PLANES) {
            const int2 a = *reinterpret_cast<const int2*>(rd), b = *reinterpret_cast<const int2*>(rd + 128);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
        } else {
            const int4 v = *reinterpret_cast<const int4*>(rd);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
        }
    }
};
struct XAllLds {
    XLdsAt<4> e54;
    XLdsAt<2> e32;
    XLdsAt<0> e10;
    __device__ __forceinline__ XAllLds(uint32_t* wave_buf /* 256 dwords */, int lane)
    {
        e54.init(wave_buf, lane);
        e32.init(wave_buf, lane);
        e10.init(wave_buf, lane);
    }
    __device__ __forceinline__ void x54(int32_t (&r)[4]) const { e54(r); }
    __device__ __forceinline__ void x32(int32_t (&r)[4]) const { e32(r); }
    __device__ __forceinline__ void operator()(int32_t (&r)[4]) const { e10(r); }
};
// (5:4) and (3:2) of a policy that only provides the (1:0) exchange stay in registers
template <class X>
__device__ __forceinline__ auto do_x54(const X& x, int32_t (&r)[4], int) -> decltype(x.x54(r), void()) { x.x54(r); }
template <class X>
__device__ __forceinline__ void do_x54(const X&, int32_t (&r)[4], long) { xchg_54(r); }
template <class X>
__device__ __forceinline__ auto do_x32(const X& x, int32_t (&r)[4], int) -> decltype(x.x32(r), void()) { x.x32(r); }
template <class X>
__device__ __forceinline__ void do_x32(const X&, int32_t (&r)[4], long) { xchg_32(r); }

// one forward radix-2x2 pass on the lane's 4-tuple (ref_ntt2x2.cpp:57-79 / butterfly2x2.v)
__device__ __forceinline__ void fwd_pass(int32_t (&r)[4], const Tw8& t)
{
    ct_bfly(r[0], r[2], (int32_t)t.v[0], t.v[1]);
    ct_bfly(r[1], r[3], (int32_t)t.v[0], t.v[1]);
    ct_bfly(r[0], r[1], (int32_t)t.v[2], t.v[3]);
    ct_bfly(r[2], r[3], (int32_t)t.v[4], t.v[5]);
}

// the same pass with wave-uniform twiddles as the SCALAR operand of the multiplies (TwLdsC: forward pass 0)
__device__ __forceinline__ void ct_bfly_s(int32_t& x, int32_t& y, uint32_t wt, uint32_t wq)
{
    const int32_t t = mont_tw_s(y, wt, wq);
    y = x - t;
    x = x + t;
}
__device__ __forceinline__ void fwd_pass(int32_t (&r)[4], const TwU& t)
{
    ct_bfly_s(r[0], r[2], t.v[0], t.v[1]);
    ct_bfly_s(r[1], r[3], t.v[0], t.v[1]);
    ct_bfly_s(r[0], r[1], t.v[2], t.v[3]);
    ct_bfly_s(r[2], r[3], t.v[4], t.v[5]);
}

// Forward NTT.  In: r[m] = a[lane + 64 m], any int32 with |a| < 2^31 - 6q.
// Out: r[m] = lazy residue (|.| < |in| + 6q) of ntt(a)[4 lane + m]; canon_any() to leave the chip.
// Twiddles are fetched ONE PASS AHEAD and pinned there with a scheduling fence: with
// LDS-resident tables (TwLds) each fetch is two ds_read_b128 whose latency would otherwise sit
// exposed at the head of every pass (hipcc sinks the loads to their first use).
#define DIL_TW_FENCE() __builtin_amdgcn_sched_barrier(0)
template <class TW, class X10>
__device__ __forceinline__ void ntt_fwd_core(int32_t (&r)[4], const TW& tw, const X10& x10)
{
    const auto t0 = tw.template get<0>();
    const auto t1 = tw.template get<1>();
    DIL_TW_FENCE();
    fwd_pass(r, t0);
    do_x54(x10, r, 0);
    const auto t2 = tw.template get<2>();
    DIL_TW_FENCE();
    fwd_pass(r, t1);
    do_x32(x10, r, 0);
    const auto t3 = tw.template get<3>();
    DIL_TW_FENCE();
    fwd_pass(r, t2);
    x10(r);
    fwd_pass(r, t3);
}

// one inverse pass (ref_ntt2x2.cpp:122-140, without the per-butterfly halving -- the 2^-8 is
// applied once, folded into the last pass's constants)
template <bool LAST>
__device__ __forceinline__ void inv_pass(int32_t (&r)[4], const Tw8& t)
{
    gs_bfly(r[0], r[1], (int32_t)t.v[0], t.v[1]);
    gs_bfly(r[2], r[3], (int32_t)t.v[2], t.v[3]);
    gs_bfly(r[0], r[2], (int32_t)t.v[4], t.v[5]);
    gs_bfly(r[1], r[3], (int32_t)t.v[4], t.v[5]);
    if (LAST) {
        r[0] = mont_tw(r[0], (int32_t)t.v[6], t.v[7]);
        r[1] = mont_tw(r[1], (int32_t)t.v[6], t.v[7]);
    }
}

__device__ __forceinline__ void gs_bfly_s(int32_t& x, int32_t& y, uint32_t wt, uint32_t wq)
{
    const int32_t d = x - y;
    x = x + y;
    y = mont_tw_s(d, wt, wq);
}
// The last inverse pass (wave-uniform twiddles).  Its two extra products are not only the 256^-1: the x outputs of Gentleman-Sande
// butterflies are plain sums, r[0] has grown to 256 q by now, and the multiplication by f is also its reduction into (-q, q) --
// which is why f cannot simply be folded into an operand of the preceding pointwise product (tried in round 4: r[0] would need a
// reduction of the same cost).
template <bool LAST>
__device__ __forceinline__ void inv_pass(int32_t (&r)[4], const TwU& t)
{
    static_assert(LAST, "the wave-uniform inverse pass is the last");
    gs_bfly_s(r[0], r[1], t.v[0], t.v[1]);
    gs_bfly_s(r[2], r[3], t.v[2], t.v[3]);
    gs_bfly_s(r[0], r[2], t.v[4], t.v[5]);
    gs_bfly_s(r[1], r[3], t.v[4], t.v[5]);
    r[0] = mont_tw_s(r[0], t.v[6], t.v[7]);
    r[1] = mont_tw_s(r[1], t.v[6], t.v[7]);
}

// Inverse NTT.  In: r[m] = a[4 lane + m] with |a| < q.  Out: r[m] in (-q, q) congruent to
// invntt(a)[lane + 64 m] (the 256^-1 of ref_ntt.cpp:83-86 included); canon_small() for [0, q).
template <class TW, class X10>
__device__ __forceinline__ void ntt_inv_core(int32_t (&r)[4], const TW& tw, const X10& x10)
{
    const auto t0 = tw.template get<0>();
    const auto t1 = tw.template get<1>();
    DIL_TW_FENCE();
    inv_pass<false>(r, t0);
    x10(r);
    const auto t2 = tw.template get<2>();
    DIL_TW_FENCE();
    inv_pass<false>(r, t1);
    do_x32(x10, r, 0);
    const auto t3 = tw.template get<3>();
    DIL_TW_FENCE();
    inv_pass<false>(r, t2);
    do_x54(x10, r, 0);
    inv_pass<true>(r, t3);
}

// Two transforms side by side on one wave (round 4).  A wave alone exposes every LDS round trip of a transform -- the twiddle
// reads and, with the exchanges through LDS, three write -> read turnarounds -- and the VALU-bound pipelines run at 4-6 waves per
// SIMD, too few to cover them (SQ_WAIT_INST_ANY 36 % of the sign kernels' wave cycles, profiles/r03z_sign_pmc.txt).  Two
// independent polynomials per pass give the scheduler a second dependency chain: b's butterflies run under a's exchange and vice
// versa, and ONE set of twiddle reads serves both (half the ds_read_b128 per transform).  The two exchanges of a pass go through the
// SAME 1-KiB buffer back to back: a wave's LDS instructions execute in order, so b's stores cannot overtake a's read.
template <class TW, class X10>
__device__ __forceinline__ void ntt_fwd_core2(int32_t (&a)[4], int32_t (&b)[4], const TW& tw, const X10& x10)
{
    const auto t0 = tw.template get<0>();
    const auto t1 = tw.template get<1>();
    DIL_TW_FENCE();
    fwd_pass(a, t0);
    do_x54(x10, a, 0);
    fwd_pass(b, t0);
    do_x54(x10, b, 0);
    const auto t2 = tw.template get<2>();
    DIL_TW_FENCE();
    fwd_pass(a, t1);
    do_x32(x10, a, 0);
    fwd_pass(b, t1);
    do_x32(x10, b, 0);
    const auto t3 = tw.template get<3>();
    DIL_TW_FENCE();
    fwd_pass(a, t2);
    x10(a);
    fwd_pass(b, t2);
    x10(b);
    fwd_pass(a, t3);
    fwd_pass(b, t3);
}
template <class TW, class X10>
__device__ __forceinline__ void ntt_inv_core2(int32_t (&a)[4], int32_t (&b)[4], const TW& tw, const X10& x10)
{
    const auto t0 = tw.template get<0>();
    const auto t1 = tw.template get<1>();
    DIL_TW_FENCE();
    inv_pass<false>(a, t0);
    x10(a);
    inv_pass<false>(b, t0);
    x10(b);
    const auto t2 = tw.template get<2>();
    DIL_TW_FENCE();
    inv_pass<false>(a, t1);
    do_x32(x10, a, 0);
    inv_pass<false>(b, t1);
    do_x32(x10, b, 0);
    const auto t3 = tw.template get<3>();
    DIL_TW_FENCE();
    inv_pass<false>(a, t2);
    do_x54(x10, a, 0);
    inv_pass<false>(b, t2);
    do_x54(x10, b, 0);
    inv_pass<true>(a, t3);
    inv_pass<true>(b, t3);
}

// A forward and an inverse transform side by side (the fused verify kernel: NTT(t1[k+1] 2^13) beside INTT(row k)): two independent
// chains again, each with its own table; the exchanges share the wave's buffer in program order.
template <class TWF, class TWI, class X10>
__device__ __forceinline__ void ntt_fwd_inv_pair(int32_t (&f)[4], int32_t (&g)[4], const TWF& twf, const TWI& twi, const X10& x10)
{
    const auto f0 = twf.template get<0>();
    const auto g0 = twi.template get<0>();
    const auto f1 = twf.template get<1>();
    const auto g1 = twi.template get<1>();
    DIL_TW_FENCE();
    fwd_pass(f, f0);
    do_x54(x10, f, 0);
    inv_pass<false>(g, g0);
    x10(g);
    const auto f2 = twf.template get<2>();
    const auto g2 = twi.template get<2>();
    DIL_TW_FENCE();
    fwd_pass(f, f1);
    do_x32(x10, f, 0);
    inv_pass<false>(g, g1);
    do_x32(x10, g, 0);
    const auto f3 = twf.template get<3>();
    const auto g3 = twi.template get<3>();
    DIL_TW_FENCE();
    fwd_pass(f, f2);
    x10(f);
    inv_pass<false>(g, g2);
    do_x54(x10, g, 0);
    fwd_pass(f, f3);
    inv_pass<true>(g, g3);
}

// N forward transforms side by side (the L polynomials of a vector that lives in registers anyway): one set of twiddle reads for
// all of them, N independent dependency chains per pass
template <int N, class TW, class X10>
__device__ __forceinline__ void ntt_fwd_coreN(int32_t (&v)[N][4], const TW& tw, const X10& x10)
{
    const auto t0 = tw.template get<0>();
    const auto t1 = tw.template get<1>();
    DIL_TW_FENCE();
#pragma unroll
    for (int i = 0; i < N; i++) {
        fwd_pass(v[i], t0);
        do_x54(x10, v[i], 0);
    }
    const auto t2 = tw.template get<2>();
    DIL_TW_FENCE();
#pragma unroll
    for (int i = 0; i < N; i++) {
        fwd_pass(v[i], t1);
        do_x32(x10, v[i], 0);
    }
    const auto t3 = tw.template get<3>();
    DIL_TW_FENCE();
#pragma unroll
    for (int i = 0; i < N; i++) {
        fwd_pass(v[i], t2);
        x10(v[i]);
    }
#pragma unroll
    for (int i = 0; i < N; i++) fwd_pass(v[i], t3);
}

}  // namespace dil
