// tune_xchg.hip -- compute-only microbenchmark of the wavefront transform's exchange / twiddle variants (round 3):
// how many issue cycles does a transform cost with the exchanges in registers (permlane / DPP / v_bfi), with the (1:0)
// exchange through LDS, or with all three through LDS -- and with the twiddles read from LDS per pass or held in VGPRs --
// at the occupancies the fused pipelines run at.  Body = what sign phase 2 does per row: 4 Montgomery products, one inverse
// transform, a cheap reduction that keeps the result live.   Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
// scripts/tune_xchg.hip -Ldilithium_amd -ldil256 -Wl,-rpath,$PWD/dilithium_amd -o scripts/bin/tune_xchg
#include "../dilithium_amd/csrc/ntt_core.hpp"
#include "../include/dil256.h"
#include <stdio.h>
#include <stdlib.h>
using namespace dil;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { X_REG = 0, X_10LDS = 1, X_ALL = 2 };
enum { T_LDS = 0, T_REG = 1 };

template <int XV> struct XSel;
template <> struct XSel<X_REG> { using type = X10Dpp; };
template <> struct XSel<X_10LDS> { using type = X10Lds; };
template <> struct XSel<X_ALL> { using type = XAllLds; };

// (occupancy is set by the launch: dynamic LDS padding so that exactly BPC workgroups fit a CU.  NOT by amdgpu_waves_per_eu(n, n):
// its upper bound makes hipcc pad the VGPR allocation -- (1, 1) gave next_free_vgpr 257 and one wave per SIMD whatever was launched.)
template <int XV, int TV, bool INV, int MINW>
__global__ __launch_bounds__(256) void xform_variant(
    int32_t* __restrict__ out, const int32_t* __restrict__ key, int iters, const uint32_t* __restrict__ tab)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];       // TW_TABLE_DWORDS + 4 * 256 used, the rest is occupancy padding
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < TW_TABLE_DWORDS / 4; i += blockDim.x) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(tab)[i];
    __syncthreads();
    const typename XSel<XV>::type x(lds + TW_TABLE_DWORDS + wv * 256, lane);
    const TwLds tl{lds, lane};
    TwRegs tr;
    if (TV == T_REG) tr.load(tab, lane);
    int32_t ch[4] = {lane * 7 + 1, lane * 11 + 3, lane * 13 + 5, lane * 17 + 7};
    int32_t acc[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        const int4 s = *reinterpret_cast<const int4*>(key + (it & 7) * 256 + 4 * lane);      // L2-resident key rows, as the shared-key phase 2
        int32_t r[4] = {mont_mul(ch[0], s.x), mont_mul(ch[1], s.y), mont_mul(ch[2], s.z), mont_mul(ch[3], s.w)};
        if (INV) {
            if (TV == T_REG) ntt_inv_core(r, tr, x); else ntt_inv_core(r, tl, x);
        } else {
            if (TV == T_REG) ntt_fwd_core(r, tr, x); else ntt_fwd_core(r, tl, x);
        }
#pragma unroll
        for (int m = 0; m < 4; m++) acc[m] += (int32_t)canon_small(INV ? r[m] : (int32_t)canon_any(r[m]));
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) out[blockIdx.x * 256 + threadIdx.x] = acc[0];
}

// decomposition of the register-exchange transform's cost: PART 0 = products + reduction only (no transform), 1 = the four
// butterfly passes without any exchange, 2 = the three exchanges without butterflies, 3 = everything (as xform_variant)
struct XNone {
    __device__ __forceinline__ XNone(uint32_t*, int) {}
    __device__ __forceinline__ void x54(int32_t (&)[4]) const {}
    __device__ __forceinline__ void x32(int32_t (&)[4]) const {}
    __device__ __forceinline__ void operator()(int32_t (&)[4]) const {}
};
template <int PART>
__global__ __launch_bounds__(256) void part_variant(int32_t* __restrict__ out, const int32_t* __restrict__ key, int iters, const uint32_t* __restrict__ tab)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];       // occupancy padding only
    const int lane = threadIdx.x & 63;
    TwRegs tr;
    tr.load(tab, lane);
    const X10Dpp xr(lane);
    const XNone xn(nullptr, lane);
    int32_t ch[4] = {lane * 7 + 1, lane * 11 + 3, lane * 13 + 5, lane * 17 + 7};
    int32_t acc[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        const int4 s = *reinterpret_cast<const int4*>(key + (it & 7) * 256 + 4 * lane);
        int32_t r[4] = {mont_mul(ch[0], s.x), mont_mul(ch[1], s.y), mont_mul(ch[2], s.z), mont_mul(ch[3], s.w)};
        if (PART == 1) ntt_inv_core(r, tr, xn);
        if (PART == 2) {
            xr(r);
            xchg_32(r);
            xchg_54(r);
        }
        if (PART == 3) ntt_inv_core(r, tr, xr);
#pragma unroll
        for (int m = 0; m < 4; m++) acc[m] += (int32_t)canon_small(r[m]);
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) out[blockIdx.x * 256 + threadIdx.x] = acc[0];
}

// correctness of the policies against each other: one transform of a given polynomial per wave, written out
template <int XV, bool INV>
__global__ __launch_bounds__(256) void xform_check(int32_t* __restrict__ polys, size_t n, const uint32_t* __restrict__ tab)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const typename XSel<XV>::type x(lds + wv * 256, lane);
    TwRegs tr;
    tr.load(tab, lane);
    const size_t p = (size_t)blockIdx.x * 4 + wv;
    if (p >= n) return;
    int32_t r[4];
    if (INV) {
        const int4 v = *reinterpret_cast<const int4*>(polys + p * 256 + 4 * lane);
        r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
        ntt_inv_core(r, tr, x);
#pragma unroll
        for (int m = 0; m < 4; m++) polys[p * 256 + lane + 64 * m] = (int32_t)canon_small(r[m]);
    } else {
#pragma unroll
        for (int m = 0; m < 4; m++) r[m] = polys[p * 256 + lane + 64 * m];
        ntt_fwd_core(r, tr, x);
        *reinterpret_cast<int4*>(polys + p * 256 + 4 * lane) = make_int4((int32_t)canon_any(r[0]), (int32_t)canon_any(r[1]), (int32_t)canon_any(r[2]), (int32_t)canon_any(r[3]));
    }
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    static uint32_t h_tab[3 * 2048];
    dil_host_twiddle_tables(h_tab, h_tab + 2048, h_tab + 4096);
    uint32_t* d_tab;
    CK(hipMalloc(&d_tab, sizeof(h_tab)));
    CK(hipMemcpy(d_tab, h_tab, sizeof(h_tab), hipMemcpyHostToDevice));
    int32_t *d_key, *d_out;
    CK(hipMalloc(&d_key, 8 * 1024));
    CK(hipMalloc(&d_out, (size_t)cus * 8 * 1024));
    static int32_t hk[8 * 256];
    for (int i = 0; i < 8 * 256; i++) hk[i] = (int32_t)((i * 2654435761u) % 8380417u);
    CK(hipMemcpy(d_key, hk, sizeof(hk), hipMemcpyHostToDevice));

    // ---- correctness: every policy == the register policy, forward and inverse, 1024 polynomials
    {
        const size_t n = 1024;
        static int32_t h[1024 * 256], ref[1024 * 256], got[1024 * 256];
        for (size_t i = 0; i < n * 256; i++) h[i] = (int32_t)((i * 2246822519u + 12345u) % 8380417u);
        int32_t* d;
        CK(hipMalloc(&d, sizeof(h)));
        int bad = 0;
#define CHECK(XV, INV)                                                            \
    CK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));                        \
    xform_check<X_REG, INV><<<n / 4, 256>>>(d, n, d_tab + (INV ? 2048 : 0));      \
    CK(hipMemcpy(ref, d, sizeof(h), hipMemcpyDeviceToHost));                      \
    CK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));                        \
    xform_check<XV, INV><<<n / 4, 256>>>(d, n, d_tab + (INV ? 2048 : 0));         \
    CK(hipMemcpy(got, d, sizeof(h), hipMemcpyDeviceToHost));                      \
    {                                                                             \
        size_t nb = 0;                                                            \
        for (size_t i = 0; i < n * 256; i++) nb += ref[i] != got[i];              \
        printf("check policy %d %s: %zu mismatches\n", XV, INV ? "inv" : "fwd", nb); \
        bad += nb != 0;                                                           \
    }
        CHECK(X_10LDS, false) CHECK(X_10LDS, true) CHECK(X_ALL, false) CHECK(X_ALL, true)
        if (bad) return 1;
    }

    const int iters = 256;
#define LDSB(BPC) ((size_t)((160 * 1024 / (BPC)) & ~255))          /* exactly BPC workgroups of 4 waves per CU */

    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
#define RUN(XV, TV, INV, MINW, BPC, label)                                                                         \
    {                                                                                                              \
        CK(hipFuncSetAttribute((const void*)xform_variant<XV, TV, INV, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        for (int w = 0; w < 2; w++) xform_variant<XV, TV, INV, MINW><<<cus * BPC, 256, LDSB(BPC)>>>(d_out, d_key, iters, d_tab + (INV ? 2048 : 0)); \
        CK(hipDeviceSynchronize());                                                                                \
        CK(hipEventRecord(a));                                                                                     \
        for (int w = 0; w < 5; w++) xform_variant<XV, TV, INV, MINW><<<cus * BPC, 256, LDSB(BPC)>>>(d_out, d_key, iters, d_tab + (INV ? 2048 : 0)); \
        CK(hipEventRecord(b));                                                                                     \
        CK(hipEventSynchronize(b));                                                                                \
        float ms;                                                                                                  \
        CK(hipEventElapsedTime(&ms, a, b));                                                                        \
        ms /= 5;                                                                                                   \
        const double xf = (double)cus * BPC * 4 * iters;                                                           \
        printf("%-46s waves/SIMD %d  %8.1f us  %6.3f G xform/s  %6.0f cycles/xform/SIMD @2.4GHz\n", label, BPC, ms * 1e3, xf / (ms * 1e-3) / 1e9, \
               ms * 1e-3 * 2.4e9 / (xf / (cus * 4.0)));                                                            \
    }
#define RUNP(PART, BPC, label)                                                                                     \
    {                                                                                                              \
        CK(hipFuncSetAttribute((const void*)part_variant<PART>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        for (int w = 0; w < 2; w++) part_variant<PART><<<cus * BPC, 256, LDSB(BPC)>>>(d_out, d_key, iters, d_tab + 2048);     \
        CK(hipDeviceSynchronize());                                                                                \
        CK(hipEventRecord(a));                                                                                     \
        for (int w = 0; w < 5; w++) part_variant<PART><<<cus * BPC, 256, LDSB(BPC)>>>(d_out, d_key, iters, d_tab + 2048);     \
        CK(hipEventRecord(b));                                                                                     \
        CK(hipEventSynchronize(b));                                                                                \
        float ms;                                                                                                  \
        CK(hipEventElapsedTime(&ms, a, b));                                                                        \
        ms /= 5;                                                                                                   \
        const double xf = (double)cus * BPC * 4 * iters;                                                           \
        printf("%-46s waves/SIMD %d  %8.1f us  %6.0f cycles/iter/SIMD @2.4GHz\n", label, BPC, ms * 1e3, ms * 1e-3 * 2.4e9 / (xf / (cus * 4.0))); \
    }
    RUNP(0, 1, "part: products + canon only") RUNP(0, 2, "part: products + canon only") RUNP(0, 4, "part: products + canon only") RUNP(0, 8, "part: products + canon only")
    RUNP(1, 4, "part: + 4 butterfly passes, no exchange") RUNP(1, 8, "part: + 4 butterfly passes, no exchange")
    RUNP(2, 4, "part: + 3 register exchanges, no butterflies") RUNP(2, 8, "part: + 3 register exchanges, no butterflies")
    RUNP(3, 1, "part: full inverse transform") RUNP(3, 2, "part: full inverse transform") RUNP(3, 4, "part: full inverse transform") RUNP(3, 8, "part: full inverse transform")
#define SWEEP(XV, TV, INV, label) \
    RUN(XV, TV, INV, 1, 1, label) RUN(XV, TV, INV, 1, 2, label) RUN(XV, TV, INV, 1, 3, label) RUN(XV, TV, INV, 1, 4, label) RUN(XV, TV, INV, 1, 6, label) RUN(XV, TV, INV, 1, 8, label)
    SWEEP(X_REG, T_LDS, true, "inv  exchanges REG      twiddles LDS")
    SWEEP(X_REG, T_REG, true, "inv  exchanges REG      twiddles VGPR")
    SWEEP(X_10LDS, T_LDS, true, "inv  exchange (1:0) LDS twiddles LDS")
    SWEEP(X_10LDS, T_REG, true, "inv  exchange (1:0) LDS twiddles VGPR")
    SWEEP(X_ALL, T_LDS, true, "inv  exchanges ALL LDS  twiddles LDS")
    SWEEP(X_ALL, T_REG, true, "inv  exchanges ALL LDS  twiddles VGPR")
    SWEEP(X_REG, T_LDS, false, "fwd  exchanges REG      twiddles LDS")
    SWEEP(X_REG, T_REG, false, "fwd  exchanges REG      twiddles VGPR")
    SWEEP(X_ALL, T_LDS, false, "fwd  exchanges ALL LDS  twiddles LDS")
    SWEEP(X_ALL, T_REG, false, "fwd  exchanges ALL LDS  twiddles VGPR")
    return 0;
}
