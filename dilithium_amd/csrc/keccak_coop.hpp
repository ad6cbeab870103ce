// keccak_coop.hpp -- Keccak-f[1600] with ONE state spread over a wavefront, for the latency-bound sponge chains
// (narrow signing rounds, single signatures / verifications / key generations).  What the reference does with one
// state per Keccak core and a round per cycle (rtl_src/keccak_datapath.vhd:97,116-117; three cores, combined_top.v:225-231;
// the challenge path gen_c.v:163-196,318-339).  keccak.hpp's lane-per-sponge forms run a 120..182-instruction round as one
// dependent chain per lane; here a round is 29 VALU instructions + one LDS gather (2.36 us per permutation, profiles/r05d_keccak_coop_asm.txt):
//
//   lane L = 32 h + l : dword h (0 = low, 1 = high half) of state word w,  l = w (w < 15) or w + 1 (w >= 15), w = x + 5 y
//   -> row 0 of a half-wave holds the planes y = 0, 1, 2 as three groups of five lanes, row 1 the planes y = 3, 4
//      (lanes 15, 26..31 of each half idle and hold zero)
//   theta  column parity: two DPP row shifts + one v_permlane16_swap; the x -+ 1 neighbours of the parity by DPP (cyclic over five
//          lanes: a shifted copy supplies the wrapped lane); rot(C, 1) needs the other half: one v_permlane32_swap; D goes back
//          to the other planes by two more row shifts
//   rho+pi ONE gather: every lane fetches "its" source dword and that word's other half with two ds_bpermute_b32 (the half swap
//          of the rotations by >= 32 is folded into the gather addresses) and one v_alignbit_b32 with a per-lane amount
//   chi    the x + 1, x + 2 neighbours by DPP, one v_bitop3_b32
//   iota   the round constants sit one per lane in a VGPR that shifts down one lane per round (DPP wave_shl:1)
//
// Written from FIPS 202; checked against a host model (scripts/tune_keccak_coop.hip), the lane-per-sponge forms and hashlib (tests/test_gpu_coop.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dil {

__device__ __constant__ uint8_t KECCAK_RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
__device__ __constant__ uint32_t KECCAK_RC32[48] = {      // [0..24): low halves, [24..48): high halves
    0x00000001u, 0x00008082u, 0x0000808au, 0x80008000u, 0x0000808bu, 0x80000001u, 0x80008081u, 0x00008009u,
    0x0000008au, 0x00000088u, 0x80008009u, 0x8000000au, 0x8000808bu, 0x0000008bu, 0x00008089u, 0x00008003u,
    0x00008002u, 0x00000080u, 0x0000800au, 0x8000000au, 0x80008081u, 0x00008080u, 0x80000001u, 0x80008008u,
    0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u,
    0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x80000000u,
    0x80000000u, 0x80000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x80000000u, 0x00000000u, 0x80000000u};

namespace coop {

constexpr int SHL(int n) { return 0x100 | n; }       // DPP row_shl:n -- lane i reads lane i + n of its row of 16
constexpr int SHR(int n) { return 0x110 | n; }       // DPP row_shr:n -- lane i reads lane i - n
constexpr int WAVE_SHL1 = 0x130;                     // lane i reads lane i + 1 across the whole wave

template <int CTRL, int ROWS = 0xF, int BANKS = 0xF>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)           // lanes without a source (or masked off) read 0
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, BANKS, true);
}
template <int CTRL, int ROWS = 0xF, int BANKS = 0xF>
__device__ __forceinline__ uint32_t dpp_keep(uint32_t old, uint32_t v)   // lanes without a source (or masked off) keep `old`
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROWS, BANKS, false);
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b)   // mask ? a : b, bitwise (v_bfi_b32)
{
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(a), "v"(b));
    return d;
}

__device__ __forceinline__ int lane_of_word(int w) { return w < 15 ? w : w + 1; }

// The per-lane constants of the permutation and the sponge around it (a dozen VGPRs, computed once per kernel).
struct Lane {
    uint32_t idx_own, idx_par;   // ds_bpermute byte addresses of rho + pi
    uint32_t sh;                 // v_alignbit amount
    uint32_t m_g0;               // all-ones in the first five lanes of every row (where the parities live)
    uint32_t m_x4, m_x3;         // all-ones where x == 4 / x >= 3 (the lanes whose x + 1 / x + 2 neighbour wraps)
    uint32_t m_hi;               // all-ones in lanes 32..63
    uint32_t m_iota;             // all-ones in lanes 0 and 32 (state word 0)
    uint32_t rc0;                // round constants: lane r = low half of RC[r], lane 32 + r = high half
    int word;                    // state word of this lane, -1 for an idle lane
    int half;                    // 0 / 1
    int dword;                   // 2 * word + half: this lane's dword of the state seen as 50 little-endian dwords (-1: idle)

    __device__ __forceinline__ void init(int lane)
    {
        const int l = lane & 31;
        half = lane >> 5;
        const bool live = l < 26 && l != 15;
        word = live ? (l < 15 ? l : l - 1) : -1;
        dword = live ? 2 * word + half : -1;
        const int X = live ? word % 5 : 0, Y = live ? word / 5 : 0;
        const int ws = (X + 3 * Y) % 5 + 5 * X;                  // pi: b[X][Y] = rot(a[(X + 3 Y) % 5][X])
        const int n = KECCAK_RHO[ws], m = n & 31;
        const int hs = half ^ (n >= 32 ? 1 : 0);                 // rotation by >= 32 = half swap + rotation by n - 32
        sh = (uint32_t)((32 - m) & 31);
        idx_own = live ? 4u * (uint32_t)(32 * hs + lane_of_word(ws)) : 4u * (uint32_t)lane;
        idx_par = live ? (m == 0 ? idx_own : 4u * (uint32_t)(32 * (hs ^ 1) + lane_of_word(ws))) : idx_own;
        m_g0 = (lane & 15) < 5 ? ~0u : 0u;
        m_x4 = live && X == 4 ? ~0u : 0u;
        m_x3 = live && X >= 3 ? ~0u : 0u;
        m_hi = half ? ~0u : 0u;
        m_iota = l == 0 ? ~0u : 0u;
        rc0 = l < 24 ? KECCAK_RC32[24 * half + l] : 0u;
    }
};

// 24 rounds on the wave's state: v = this lane's dword (idle lanes: 0 in, 0 out).  Reference form, as the compiler schedules it
// (34 VALU instructions per round); permute() below is the same round written out by hand.
__device__ __forceinline__ uint32_t permute_c(uint32_t v, const Lane& k)
{
    uint32_t rc = k.rc0;
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        // theta
        uint32_t p = v ^ dpp0<SHL(5)>(v);
        p ^= dpp0<SHL(10), 0x5>(v);                                    // rows 0 / 2: the third plane of the row
        const auto pq = __builtin_amdgcn_permlane16_swap(p, p, false, false);
        const uint32_t c = pq[0] ^ pq[1];                                // column parity C[x] in the first five lanes of rows 0..3
        const uint32_t cm = dpp_keep<SHR(1)>(dpp0<SHL(4)>(c), c);        // C[x - 1]: lane 0 has no source under row_shr:1 and keeps lane 4's
        const uint32_t cp = dpp_keep<SHL(1), 0xF, 0xD>(dpp0<SHR(4)>(c), c);   // C[x + 1]: bank 1 (lane 4) keeps lane 0's
        const auto hl = __builtin_amdgcn_permlane32_swap(cp, cp, false, false);    // [0] = low halves everywhere, [1] = high halves
        const uint32_t other = bfi(k.m_hi, hl[0], hl[1]);
        const uint32_t rot = __builtin_amdgcn_alignbit(cp, other, 31);   // this half of rot64(C[x + 1], 1)
        const uint32_t d = __builtin_amdgcn_bitop3_b32(cm, rot, k.m_g0, 0x28);   // (cm ^ rot) & m_g0
        v ^= d;
        v ^= dpp0<SHR(5)>(d);
        v ^= dpp0<SHR(10), 0x5>(d);
        // rho + pi
        const uint32_t own = (uint32_t)__builtin_amdgcn_ds_bpermute((int)k.idx_own, (int)v);
        const uint32_t par = (uint32_t)__builtin_amdgcn_ds_bpermute((int)k.idx_par, (int)v);
        const uint32_t b = __builtin_amdgcn_alignbit(own, par, k.sh);
        // chi
        const uint32_t b1 = bfi(k.m_x4, dpp0<SHR(4)>(b), dpp0<SHL(1)>(b));
        const uint32_t b2 = bfi(k.m_x3, dpp0<SHR(3)>(b), dpp0<SHL(2)>(b));
        v = __builtin_amdgcn_bitop3_b32(b, b1, b2, 0xD2);                // b ^ (~b1 & b2)
        // iota
        v = __builtin_amdgcn_bitop3_b32(v, rc, k.m_iota, 0x78);          // v ^ (rc & m_iota)
        rc = dpp0<WAVE_SHL1>(rc);
    }
    return v;
}

// The same 24 rounds by hand: 29 VALU instructions per round.  What the compiler cannot do: fold the row shifts into the xors
// (v_xor_b32_dpp), select the wrapped x + 1 neighbour with v_cndmask_b32_dpp under a loop-invariant VCC (= lanes with x != 4), place the
// independent instructions into the wait states the DPP / v_permlane reads need after a VALU write, and run the iota bookkeeping under
// the gather's LDS round trip.  (The one v_cndmask_b32_dpp per round is NOT the 23-cycle instruction profiles/r01_ubench_valu_rates.txt shows
// for VCC-form selects in a dependent chain: with VCC loop-invariant, mov_dpp + v_bfi_b32 in its place measured 1-3 % slower,
// profiles/r05o_keccak_coop_bfi.txt.)
__device__ __forceinline__ uint32_t permute(uint32_t v, const Lane& k)
{
    uint32_t rc = k.rc0, p, q, cm, cp, t, u, d, own, par, b, b1, b2, ri;
    const uint64_t not_x4 = __builtin_amdgcn_ballot_w64(k.m_x4 == 0);
    asm volatile(
        "s_mov_b64 vcc, %[nx4]\n\t"
        "s_movk_i32 s30, 24\n"
        "1:\n\t"
        // theta: column parity
        "v_xor_b32_dpp %[p], %[v], %[v] row_shl:5 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_and_b32 %[ri], %[rc], %[mi]\n\t"
        "v_xor_b32_dpp %[p], %[v], %[p] row_shl:10 row_mask:0x5 bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32 %[q], %[p]\n\t"
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %[p], %[q]\n\t"
        "v_xor_b32 %[p], %[p], %[q]\n\t"
        "v_mov_b32_dpp %[rc], %[rc] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 0\n\t"
        // C[x - 1], C[x + 1] (cyclic over the five lanes of the first group of each row)
        "v_mov_b32_dpp %[cm], %[p] row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %[cp], %[p] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %[cm], %[p] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[cp], %[p] row_shl:1 row_mask:0xf bank_mask:0xd\n\t"
        "v_mov_b32 %[t], %[cp]\n\t"
        "v_mov_b32 %[u], %[cp]\n\t"
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %[t], %[u]\n\t"
        "v_bfi_b32 %[t], %[mh], %[t], %[u]\n\t"
        "v_alignbit_b32 %[t], %[cp], %[t], 31\n\t"
        "v_bitop3_b32 %[d], %[cm], %[t], %[mg] bitop3:0x28\n\t"
        "v_xor_b32 %[v], %[v], %[d]\n\t"
        "s_nop 0\n\t"
        "v_xor_b32_dpp %[v], %[d], %[v] row_shr:5 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_xor_b32_dpp %[v], %[d], %[v] row_shr:10 row_mask:0x5 bank_mask:0xf bound_ctrl:0\n\t"
        // rho + pi
        "ds_bpermute_b32 %[own], %[io], %[v]\n\t"
        "ds_bpermute_b32 %[par], %[ip], %[v]\n\t"
        "s_add_i32 s30, s30, -1\n\t"
        "s_cmp_lg_u32 s30, 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_alignbit_b32 %[b], %[own], %[par], %[sh]\n\t"
        "s_nop 1\n\t"
        // chi
        "v_mov_b32_dpp %[b1], %[b] row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %[b2], %[b] row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_mov_b32_dpp %[t], %[b] row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_cndmask_b32_dpp %[b1], %[b], %[b1], vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_bfi_b32 %[b2], %[m3], %[t], %[b2]\n\t"
        "v_bitop3_b32 %[v], %[b], %[b1], %[b2] bitop3:0xd2\n\t"
        // iota
        "v_xor_b32 %[v], %[v], %[ri]\n\t"
        "s_nop 0\n\t"                      // (the next round opens with a DPP read of v)
        "s_cbranch_scc1 1b\n\t"
        : [v] "+v"(v), [rc] "+v"(rc), [p] "=&v"(p), [q] "=&v"(q), [cm] "=&v"(cm), [cp] "=&v"(cp), [t] "=&v"(t), [u] "=&v"(u), [d] "=&v"(d),
          [own] "=&v"(own), [par] "=&v"(par), [b] "=&v"(b), [b1] "=&v"(b1), [b2] "=&v"(b2), [ri] "=&v"(ri)
        : [nx4] "s"(not_x4), [mi] "v"(k.m_iota), [mh] "v"(k.m_hi), [mg] "v"(k.m_g0), [m3] "v"(k.m_x3), [io] "v"(k.idx_own), [ip] "v"(k.idx_par),
          [sh] "v"(k.sh)
        : "vcc", "scc", "s30", "memory");
    return v;
}

}  // namespace coop
}  // namespace dil
