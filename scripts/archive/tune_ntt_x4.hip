// tune_ntt_x4.hip -- the two memory-side levers the round-4 review left untried on the standalone forward NTT (kernels.hip ntt_fwd_kernel:
// 4 x global_load_dword per lane in, one dwordx4 out; 25.8 us per 65536 polynomials = 0.65 of 8 TB/s, its traffic-only skeleton 23.1 us):
//   x4    one global_load_dwordx4 per lane (1 KiB per wave instruction) and the transposition to the transform's lane + 64 m layout
//         through a 1-KiB per-wave LDS slot (ds_write_b128 + 4 x ds_read_b32, conflict-free both ways)
//   dma   the same slot filled by global_load_lds_dwordx4 (no VGPR staging, no ds_write), double-buffered: the next polynomial's DMA
//         is in flight under the current transform
// each as traffic-only skeleton and with the arithmetic.   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/tune_ntt_x4.hip -Ldilithium_amd
//   -ldil256 -Wl,-rpath,$PWD/dilithium_amd -o scripts/bin/tune_ntt_x4
#include "../dilithium_amd/csrc/device_common.hpp"
#include "../dilithium_amd/csrc/ntt_core.hpp"
#include "../include/dil256.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
using namespace dil;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { LD_DWORD = 0, LD_X4_LDS = 1, LD_DMA = 2 };

template <int MODE, bool COMPUTE, bool NT>
__global__ __launch_bounds__(256) void ntt_var(int32_t* __restrict__ polys, size_t batch, const uint32_t* __restrict__ tw_tab)
{
    __shared__ __attribute__((aligned(16))) uint32_t slots[4 * 2 * 256];        // per wave: two 1-KiB slots
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 4 + wv, nwaves = (size_t)gridDim.x * 4;
    if (wave >= batch) return;
    TwRegs tw;
    if (COMPUTE) tw.load(tw_tab, lane);
    const LaneMasks lm(lane);
    uint32_t* slot = slots + wv * 512;
    if (MODE == LD_DWORD) {
        int32_t nx[4];
#pragma unroll
        for (int m = 0; m < 4; m++) nx[m] = NT ? ld_nt(polys + wave * 256 + lane + 64 * m) : polys[wave * 256 + lane + 64 * m];
        for (size_t p = wave; p < batch; p += nwaves) {
            int32_t r[4] = {nx[0], nx[1], nx[2], nx[3]};
            const size_t pn = p + nwaves;
            if (pn < batch) {
#pragma unroll
                for (int m = 0; m < 4; m++) nx[m] = NT ? ld_nt(polys + pn * 256 + lane + 64 * m) : polys[pn * 256 + lane + 64 * m];
            }
            if (COMPUTE) ntt_fwd_core(r, tw, lm);
            st_nt4(polys + p * 256 + 4 * lane, COMPUTE ? canon_any(r[0]) : (uint32_t)r[0], COMPUTE ? canon_any(r[1]) : (uint32_t)r[1],
                   COMPUTE ? canon_any(r[2]) : (uint32_t)r[2], COMPUTE ? canon_any(r[3]) : (uint32_t)r[3]);
        }
    } else if (MODE == LD_X4_LDS) {
        int4 nx = NT ? ld_nt4(polys + wave * 256 + 4 * lane) : *reinterpret_cast<const int4*>(polys + wave * 256 + 4 * lane);
        for (size_t p = wave; p < batch; p += nwaves) {
            *reinterpret_cast<int4*>(slot + 4 * lane) = nx;
            const size_t pn = p + nwaves;
            if (pn < batch) nx = NT ? ld_nt4(polys + pn * 256 + 4 * lane) : *reinterpret_cast<const int4*>(polys + pn * 256 + 4 * lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            int32_t r[4];
#pragma unroll
            for (int m = 0; m < 4; m++) r[m] = (int32_t)slot[lane + 64 * m];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (COMPUTE) ntt_fwd_core(r, tw, lm);
            st_nt4(polys + p * 256 + 4 * lane, COMPUTE ? canon_any(r[0]) : (uint32_t)r[0], COMPUTE ? canon_any(r[1]) : (uint32_t)r[1],
                   COMPUTE ? canon_any(r[2]) : (uint32_t)r[2], COMPUTE ? canon_any(r[3]) : (uint32_t)r[3]);
        }
    } else {
        using lds_ptr = __attribute__((address_space(3))) uint32_t*;
        auto dma = [&](size_t p, int s) {
            __builtin_amdgcn_global_load_lds(polys + p * 256 + 4 * lane, (lds_ptr)(slot + s * 256), 16, 0, NT ? 2 : 0);
        };
        int s = 0;
        dma(wave, 0);
        for (size_t p = wave; p < batch; p += nwaves, s ^= 1) {
            const size_t pn = p + nwaves;
            if (pn < batch) {
                dma(pn, s ^ 1);
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");          // the older DMA (this polynomial's) has landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            int32_t r[4];
#pragma unroll
            for (int m = 0; m < 4; m++) r[m] = (int32_t)slot[s * 256 + lane + 64 * m];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // slot s read before the DMA after next may overwrite it
            if (COMPUTE) ntt_fwd_core(r, tw, lm);
            // (stores also count in vmcnt: the wait above then covers the previous trip's store too -- it is older than both DMAs)
            st_nt4(polys + p * 256 + 4 * lane, COMPUTE ? canon_any(r[0]) : (uint32_t)r[0], COMPUTE ? canon_any(r[1]) : (uint32_t)r[1],
                   COMPUTE ? canon_any(r[2]) : (uint32_t)r[2], COMPUTE ? canon_any(r[3]) : (uint32_t)r[3]);
        }
    }
}

// The shipped shape with the co-resident workgroups of a CU started out of phase: all 32 waves of a CU otherwise run load burst ->
// transform -> store burst in lockstep for the eight polynomials each of them owns.  slot = blockIdx / 256 (2048 blocks: the g-th
// workgroup of its CU), delay = slot * STEP * 64 cycles (s_sleep).
template <int STEP>
__global__ __launch_bounds__(256) void ntt_stagger(int32_t* __restrict__ polys, size_t batch, const uint32_t* __restrict__ tw_tab)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 4 + wv, nwaves = (size_t)gridDim.x * 4;
    if (wave >= batch) return;
    TwRegs tw;
    tw.load(tw_tab, lane);
    const LaneMasks lm(lane);
    const int slot = (int)(blockIdx.x >> 8) * 4 + wv * (STEP < 0 ? 1 : 0);
    for (int i = 0; i < slot * (STEP < 0 ? -STEP : STEP); i++) __builtin_amdgcn_s_sleep(1);
    int32_t nx[4];
#pragma unroll
    for (int m = 0; m < 4; m++) nx[m] = ld_nt(polys + wave * 256 + lane + 64 * m);
    for (size_t p = wave; p < batch; p += nwaves) {
        int32_t r[4] = {nx[0], nx[1], nx[2], nx[3]};
        const size_t pn = p + nwaves;
        if (pn < batch) {
#pragma unroll
            for (int m = 0; m < 4; m++) nx[m] = ld_nt(polys + pn * 256 + lane + 64 * m);
        }
        ntt_fwd_core(r, tw, lm);
        st_nt4(polys + p * 256 + 4 * lane, canon_any(r[0]), canon_any(r[1]), canon_any(r[2]), canon_any(r[3]));
    }
}

// LDS-DMA with D polynomials in flight beyond the current one (D + 1 slots of 1 KiB per wave): prefetch depth without VGPRs.
// gfx9 retires VMEM operations in issue order on one counter: after DMA(p) were issued D stores and D DMAs, so vmcnt(2 D) = "DMA(p) landed".
template <int D, bool COMPUTE, int WPB>
__global__ __launch_bounds__(64 * WPB) void ntt_dma_deep(int32_t* __restrict__ polys, size_t batch, const uint32_t* __restrict__ tw_tab)
{
    __shared__ __attribute__((aligned(16))) uint32_t slots[WPB * (D + 1) * 256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * WPB + wv, nwaves = (size_t)gridDim.x * WPB;
    if (wave >= batch) return;
    TwRegs tw;
    if (COMPUTE) tw.load(tw_tab, lane);
    const LaneMasks lm(lane);
    uint32_t* slot = slots + wv * (D + 1) * 256;
    using lds_ptr = __attribute__((address_space(3))) uint32_t*;
    auto dma = [&](size_t p, int s) { __builtin_amdgcn_global_load_lds(polys + p * 256 + 4 * lane, (lds_ptr)(slot + s * 256), 16, 0, 2); };
#pragma unroll
    for (int d = 0; d < D; d++)
        if (wave + d * nwaves < batch) dma(wave + d * nwaves, d);
    int s = 0;
    for (size_t p = wave; p < batch; p += nwaves) {
        const size_t pn = p + (size_t)D * nwaves;
        const int sn = s + D > D ? s + D - (D + 1) : s + D;
        if (pn < batch) dma(pn, sn);
        // wave-uniform tail: fewer operations behind DMA(p) when no more DMAs are issued -- wait for everything there
        if (pn < batch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * D) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int32_t r[4];
#pragma unroll
        for (int m = 0; m < 4; m++) r[m] = (int32_t)slot[s * 256 + lane + 64 * m];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (COMPUTE) ntt_fwd_core(r, tw, lm);
        st_nt4(polys + p * 256 + 4 * lane, COMPUTE ? canon_any(r[0]) : (uint32_t)r[0], COMPUTE ? canon_any(r[1]) : (uint32_t)r[1],
               COMPUTE ? canon_any(r[2]) : (uint32_t)r[2], COMPUTE ? canon_any(r[3]) : (uint32_t)r[3]);
        s = s == D ? 0 : s + 1;
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
    int32_t *d, *ref;
    CK(hipMalloc(&d, NB * batch * 1024));
    CK(hipMalloc(&ref, batch * 1024));
    std::vector<int32_t> h(batch * 256), want(batch * 256), got(batch * 256);
    for (size_t i = 0; i < h.size(); i++) h[i] = (int32_t)((i * 2654435761u) % 8380417u);
    // correctness of the two new load paths against the shipped library's transform
    CK(hipMemcpy(ref, h.data(), batch * 1024, hipMemcpyHostToDevice));
    if (dil_ntt_dev(ref, batch, nullptr)) { printf("dil_ntt_dev failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(want.data(), ref, batch * 1024, hipMemcpyDeviceToHost));
    auto check = [&](const char* name, auto launch) {
        CK(hipMemcpy(d, h.data(), batch * 1024, hipMemcpyHostToDevice));
        launch();
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), d, batch * 1024, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < got.size(); i++) bad += got[i] != want[i];
        printf("check %-28s %s (%zu of %zu coefficients differ)\n", name, bad ? "MISMATCH" : "ok", bad, got.size());
    };
    check("dword loads", [&] { ntt_var<LD_DWORD, true, true><<<cus * 8, 256>>>(d, batch, d_tab); });
    check("dwordx4 + LDS transposition", [&] { ntt_var<LD_X4_LDS, true, true><<<cus * 8, 256>>>(d, batch, d_tab); });
    check("LDS-DMA", [&] { ntt_var<LD_DMA, true, true><<<cus * 8, 256>>>(d, batch, d_tab); });
    check("LDS-DMA 2 ahead", [&] { ntt_dma_deep<2, true, 4><<<cus * 8, 256>>>(d, batch, d_tab); });
    check("LDS-DMA 3 ahead", [&] { ntt_dma_deep<3, true, 4><<<cus * 6, 256>>>(d, batch, d_tab); });
    check("LDS-DMA 4 ahead, 8 waves", [&] { ntt_dma_deep<4, true, 8><<<cus * 3, 512>>>(d, batch, d_tab); });
    for (size_t b = 0; b < NB; b++) CK(hipMemcpy(d + b * batch * 256, h.data(), batch * 1024, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2000; i++) ntt_var<LD_DWORD, true, true><<<cus * 8, 256>>>(d + (i % NB) * batch * 256, batch, d_tab);
    CK(hipDeviceSynchronize());
#define TIME(label, KERN, BPC)                                                                         \
    {                                                                                                  \
        std::vector<float> t;                                                                          \
        for (int rep = 0; rep < 5; rep++) {                                                            \
            for (int i = 0; i < 16; i++) KERN<<<cus * BPC, 256>>>(d + (i % NB) * batch * 256, batch, d_tab); \
            CK(hipEventRecord(a));                                                                     \
            for (int i = 0; i < 256; i++) KERN<<<cus * BPC, 256>>>(d + (i % NB) * batch * 256, batch, d_tab); \
            CK(hipEventRecord(b));                                                                     \
            CK(hipEventSynchronize(b));                                                                \
            float ms;                                                                                  \
            CK(hipEventElapsedTime(&ms, a, b));                                                        \
            t.push_back(ms / 256 * 1e3f);                                                              \
        }                                                                                              \
        std::sort(t.begin(), t.end());                                                                 \
        printf("%-70s %7.2f us  %7.1f GB/s  %.3f of 8 TB/s\n", label, t[2], 2048.0 * batch / t[2] / 1e3, 2048.0 * batch / t[2] / 1e3 / 8000); \
    }
    TIME("traffic only  dword loads (the shipped shape)            8 blocks/CU", (ntt_var<LD_DWORD, false, true>), 8)
    TIME("traffic only  dwordx4 + LDS transposition                8 blocks/CU", (ntt_var<LD_X4_LDS, false, true>), 8)
    TIME("traffic only  LDS-DMA nt                                 8 blocks/CU", (ntt_var<LD_DMA, false, true>), 8)
    TIME("traffic only  LDS-DMA default policy                     8 blocks/CU", (ntt_var<LD_DMA, false, false>), 8)
    TIME("traffic only  LDS-DMA nt                                 4 blocks/CU", (ntt_var<LD_DMA, false, true>), 4)
    TIME("traffic only  LDS-DMA nt                                 6 blocks/CU", (ntt_var<LD_DMA, false, true>), 6)
    TIME("transform     dword loads (the shipped shape)            8 blocks/CU", (ntt_var<LD_DWORD, true, true>), 8)
    TIME("transform     dwordx4 + LDS transposition                8 blocks/CU", (ntt_var<LD_X4_LDS, true, true>), 8)
    TIME("transform     dwordx4 + LDS transposition                6 blocks/CU", (ntt_var<LD_X4_LDS, true, true>), 6)
    TIME("transform     LDS-DMA nt                                 8 blocks/CU", (ntt_var<LD_DMA, true, true>), 8)
    TIME("transform     LDS-DMA default policy                     8 blocks/CU", (ntt_var<LD_DMA, true, false>), 8)
    TIME("transform     LDS-DMA nt                                 6 blocks/CU", (ntt_var<LD_DMA, true, true>), 6)
    TIME("transform     LDS-DMA nt                                 4 blocks/CU", (ntt_var<LD_DMA, true, true>), 4)
#define TIMED(label, KERN, BPC, TPB)                                                                  \
    {                                                                                                  \
        std::vector<float> t;                                                                          \
        for (int rep = 0; rep < 5; rep++) {                                                            \
            for (int i = 0; i < 16; i++) KERN<<<cus * BPC, TPB>>>(d + (i % NB) * batch * 256, batch, d_tab); \
            CK(hipEventRecord(a));                                                                     \
            for (int i = 0; i < 256; i++) KERN<<<cus * BPC, TPB>>>(d + (i % NB) * batch * 256, batch, d_tab); \
            CK(hipEventRecord(b));                                                                     \
            CK(hipEventSynchronize(b));                                                                \
            float ms;                                                                                  \
            CK(hipEventElapsedTime(&ms, a, b));                                                        \
            t.push_back(ms / 256 * 1e3f);                                                              \
        }                                                                                              \
        std::sort(t.begin(), t.end());                                                                 \
        printf("%-70s %7.2f us  %7.1f GB/s  %.3f of 8 TB/s\n", label, t[2], 2048.0 * batch / t[2] / 1e3, 2048.0 * batch / t[2] / 1e3 / 8000); \
    }
    TIMED("traffic only  LDS-DMA 2 ahead                            8 blocks/CU", (ntt_dma_deep<2, false, 4>), 8, 256)
    TIMED("traffic only  LDS-DMA 3 ahead                            6 blocks/CU", (ntt_dma_deep<3, false, 4>), 6, 256)
    TIMED("transform     LDS-DMA 2 ahead                            8 blocks/CU", (ntt_dma_deep<2, true, 4>), 8, 256)
    TIMED("transform     LDS-DMA 2 ahead                            6 blocks/CU", (ntt_dma_deep<2, true, 4>), 6, 256)
    TIMED("transform     LDS-DMA 3 ahead                            8 blocks/CU", (ntt_dma_deep<3, true, 4>), 8, 256)
    TIMED("transform     LDS-DMA 3 ahead                            6 blocks/CU", (ntt_dma_deep<3, true, 4>), 6, 256)
    TIMED("transform     LDS-DMA 3 ahead                            4 blocks/CU", (ntt_dma_deep<3, true, 4>), 4, 256)
    TIMED("transform     LDS-DMA 4 ahead, 8 waves per block         4 blocks/CU", (ntt_dma_deep<4, true, 8>), 4, 512)
    TIMED("transform     LDS-DMA 4 ahead, 8 waves per block         3 blocks/CU", (ntt_dma_deep<4, true, 8>), 3, 512)
    TIMED("transform     LDS-DMA 6 ahead                            4 blocks/CU", (ntt_dma_deep<6, true, 4>), 4, 256)
    TIMED("transform     shipped shape, co-resident workgroups 0.1 us apart   8 blocks/CU", (ntt_stagger<1>), 8, 256)
    TIMED("transform     shipped shape, 0.2 us apart                          8 blocks/CU", (ntt_stagger<2>), 8, 256)
    TIMED("transform     shipped shape, 0.4 us apart                          8 blocks/CU", (ntt_stagger<4>), 8, 256)
    TIMED("transform     shipped shape, 0.8 us apart                          8 blocks/CU", (ntt_stagger<8>), 8, 256)
    TIMED("transform     shipped shape, every wave 0.1 us apart               8 blocks/CU", (ntt_stagger<-1>), 8, 256)
    TIMED("transform     shipped shape, no delay (reference)                  8 blocks/CU", (ntt_stagger<0>), 8, 256)
    return 0;
}
