// tune_ntt2.hip -- round 3: how close can the standalone forward NTT get to its own traffic-only skeleton?
// (bench.py: skeleton 22.9 us per 65536 polynomials = 5.86 TB/s, transform 26.1 us = 5.14 TB/s, compute alone 18.3 us.)
// Variants: prefetch depth D (polynomials in flight per wave beyond the one being transformed), polynomials transformed
// back to back per loop trip (G), persistent blocks per CU; each with and without the arithmetic.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/tune_ntt2.hip -Ldilithium_amd -ldil256 -Wl,-rpath,... -o scripts/bin/tune_ntt2
#include "../dilithium_amd/csrc/device_common.hpp"
#include "../dilithium_amd/csrc/ntt_core.hpp"
#include "../include/dil256.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using namespace dil;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int D, bool COMPUTE, bool INV>
__global__ __launch_bounds__(256) void ntt_pf(int32_t* __restrict__ polys, size_t batch, const uint32_t* __restrict__ tw_tab)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    if (wave >= batch) return;
    TwRegs tw;
    if (COMPUTE) tw.load(tw_tab, lane);
    const LaneMasks lm(lane);
    int32_t q[D][4];                 // the next D polynomials of this wave
    auto load = [&](int32_t (&r)[4], size_t p) {
        if (!INV) {
#pragma unroll
            for (int m = 0; m < 4; m++) r[m] = ld_nt(polys + p * 256 + lane + 64 * m);
        } else {
            const int4 v = ld_nt4(polys + p * 256 + 4 * lane);
            r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
        }
    };
#pragma unroll
    for (int d = 0; d < D; d++)
        if (wave + d * nwaves < batch) load(q[d], wave + d * nwaves);
    for (size_t p = wave; p < batch; p += nwaves) {
        int32_t r[4] = {q[0][0], q[0][1], q[0][2], q[0][3]};
#pragma unroll
        for (int d = 0; d + 1 < D; d++)
#pragma unroll
            for (int m = 0; m < 4; m++) q[d][m] = q[d + 1][m];
        const size_t pn = p + (size_t)D * nwaves;
        if (pn < batch) load(q[D - 1], pn);
        if (COMPUTE) {
            if (!INV) ntt_fwd_core(r, tw, lm); else ntt_inv_core(r, tw, lm);
        }
        if (!INV) {
            if (COMPUTE) st_nt4(polys + p * 256 + 4 * lane, canon_any(r[0]), canon_any(r[1]), canon_any(r[2]), canon_any(r[3]));
            else st_nt4(polys + p * 256 + 4 * lane, (uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]);
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++) st_nt(polys + p * 256 + lane + 64 * m, COMPUTE ? (int32_t)canon_small(r[m]) : r[m]);
        }
    }
}

// a workgroup owns a CONTIGUOUS chunk of polynomials (its waves interleave inside it) instead of the grid-strided assignment:
// bigger sequential runs per CU
template <int D, bool COMPUTE>
__global__ __launch_bounds__(256) void ntt_chunk(int32_t* __restrict__ polys, size_t batch, const uint32_t* __restrict__ tw_tab)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t per = (batch + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < batch ? lo + per : batch;
    TwRegs tw;
    if (COMPUTE) tw.load(tw_tab, lane);
    const LaneMasks lm(lane);
    int32_t q[D][4];
#pragma unroll
    for (int d = 0; d < D; d++)
        if (lo + wv + 4 * d < hi) {
#pragma unroll
            for (int m = 0; m < 4; m++) q[d][m] = ld_nt(polys + (lo + wv + 4 * d) * 256 + lane + 64 * m);
        }
    for (size_t p = lo + wv; p < hi; p += 4) {
        int32_t r[4] = {q[0][0], q[0][1], q[0][2], q[0][3]};
#pragma unroll
        for (int d = 0; d + 1 < D; d++)
#pragma unroll
            for (int m = 0; m < 4; m++) q[d][m] = q[d + 1][m];
        const size_t pn = p + 4 * D;
        if (pn < hi) {
#pragma unroll
            for (int m = 0; m < 4; m++) q[D - 1][m] = ld_nt(polys + pn * 256 + lane + 64 * m);
        }
        if (COMPUTE) {
            ntt_fwd_core(r, tw, lm);
            st_nt4(polys + p * 256 + 4 * lane, canon_any(r[0]), canon_any(r[1]), canon_any(r[2]), canon_any(r[3]));
        } else {
            st_nt4(polys + p * 256 + 4 * lane, (uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]);
        }
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
    const size_t NB = 8, batch = 65536;
    int32_t* d;
    CK(hipMalloc(&d, NB * batch * 1024));
    std::vector<int32_t> h(batch * 256);
    for (size_t i = 0; i < h.size(); i++) h[i] = (int32_t)((i * 2654435761u) % 8380417u);
    for (size_t b = 0; b < NB; b++) CK(hipMemcpy(d + b * batch * 256, h.data(), batch * 1024, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    // warm the clocks
    for (int i = 0; i < 2000; i++) ntt_pf<1, true, false><<<cus * 8, 256>>>(d + (i % NB) * batch * 256, batch, d_tab);
    CK(hipDeviceSynchronize());
#define TIME(label, launch)                                                                            \
    {                                                                                                  \
        float best = 1e9, sum = 0;                                                                     \
        for (int rep = 0; rep < 5; rep++) {                                                            \
            for (int i = 0; i < 16; i++) { launch; }                                                   \
            CK(hipEventRecord(a));                                                                     \
            for (int i = 0; i < 256; i++) { launch; }                                                  \
            CK(hipEventRecord(b));                                                                     \
            CK(hipEventSynchronize(b));                                                                \
            float ms;                                                                                  \
            CK(hipEventElapsedTime(&ms, a, b));                                                        \
            ms /= 256;                                                                                 \
            best = ms < best ? ms : best;                                                              \
            sum += ms;                                                                                 \
        }                                                                                              \
        printf("%-58s best %7.2f us  mean %7.2f us  %7.1f GB/s\n", label, best * 1e3, sum / 5 * 1e3, 2048.0 * batch / (sum / 5 * 1e-3) / 1e9); \
    }
#define PF(D, C, INV, BPC, label) TIME(label, (ntt_pf<D, C, INV><<<cus * BPC, 256>>>(d + (i % NB) * batch * 256, batch, d_tab + (INV ? 2048 : 0))))
    PF(1, false, false, 8, "fwd traffic-only  D=1 bpc8")
    PF(1, true, false, 8, "fwd transform     D=1 bpc8   (the shipped kernel's shape)")
    PF(2, false, false, 8, "fwd traffic-only  D=2 bpc8")
    PF(2, true, false, 8, "fwd transform     D=2 bpc8")
    PF(3, true, false, 8, "fwd transform     D=3 bpc8")
    PF(1, true, false, 6, "fwd transform     D=1 bpc6")
    PF(2, true, false, 6, "fwd transform     D=2 bpc6")
    PF(2, true, false, 5, "fwd transform     D=2 bpc5")
    PF(2, true, false, 4, "fwd transform     D=2 bpc4")
    PF(3, true, false, 4, "fwd transform     D=3 bpc4")
    PF(2, false, false, 4, "fwd traffic-only  D=2 bpc4")
    PF(4, true, false, 4, "fwd transform     D=4 bpc4")
    PF(1, false, true, 8, "inv traffic-only  D=1 bpc8")
    PF(1, true, true, 8, "inv transform     D=1 bpc8   (the shipped kernel's shape)")
    PF(2, true, true, 8, "inv transform     D=2 bpc8")
    PF(2, true, true, 4, "inv transform     D=2 bpc4")
    PF(3, true, true, 4, "inv transform     D=3 bpc4")
#define CH(D, C, BPC, label) TIME(label, (ntt_chunk<D, C><<<cus * BPC, 256>>>(d + (i % NB) * batch * 256, batch, d_tab)))
    CH(1, false, 8, "fwd traffic-only  contiguous chunk per workgroup D=1 bpc8")
    CH(1, true, 8, "fwd transform     contiguous chunk per workgroup D=1 bpc8")
    CH(2, true, 8, "fwd transform     contiguous chunk per workgroup D=2 bpc8")
    CH(2, true, 4, "fwd transform     contiguous chunk per workgroup D=2 bpc4")
    return 0;
}
