// tune.hip -- kernel-variant exploration harness for the batched NTT (built against the real
// ntt_core.hpp).  Prints achieved algorithmic GB/s (2048 B per transform) per variant.
#include "../dilithium_amd/csrc/ntt_core.hpp"
#include "../include/dil256.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using namespace dil;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { V_FULL = 0, V_COPY = 1, V_COMPUTE = 2, V_NOPREFETCH = 3, V_NT = 4, V_PF2 = 5 };

template <int V, int WPB>
__global__ __launch_bounds__(64 * WPB) void fwd_variant(int32_t* __restrict__ polys, size_t batch,
                                                         const uint32_t* __restrict__ tw_tab)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * WPB;
    if (wave >= batch) return;
    TwRegs tw;
    tw.load(tw_tab, lane);
    const LaneMasks lm(lane);
    if (V == V_NOPREFETCH) {
        for (size_t p = wave; p < batch; p += nwaves) {
            int32_t r[4];
#pragma unroll
            for (int m = 0; m < 4; m++) r[m] = polys[p * 256 + lane + 64 * m];
            ntt_fwd_core(r, tw, lm);
            *reinterpret_cast<uint4*>(polys + p * 256 + 4 * lane) =
                make_uint4(canon_any(r[0]), canon_any(r[1]), canon_any(r[2]), canon_any(r[3]));
        }
        return;
    }
    if (V == V_COMPUTE) {
        int32_t r[4] = {lane, lane * 3, lane * 5, lane * 7};
        for (size_t p = wave; p < batch; p += nwaves) {
            ntt_fwd_core(r, tw, lm);
            r[0] = canon_any(r[0]); r[1] = canon_any(r[1]); r[2] = canon_any(r[2]); r[3] = canon_any(r[3]);
        }
        if (r[0] == 0x12345) polys[wave] = r[1] + r[2] + r[3];
        return;
    }
    if (V == V_PF2) {   // two polynomials in flight
        int32_t n1[4], n2[4];
#pragma unroll
        for (int m = 0; m < 4; m++) n1[m] = polys[wave * 256 + lane + 64 * m];
        const size_t p2 = wave + nwaves;
        if (p2 < batch) {
#pragma unroll
            for (int m = 0; m < 4; m++) n2[m] = polys[p2 * 256 + lane + 64 * m];
        }
        for (size_t p = wave; p < batch; p += nwaves) {
            int32_t r[4] = {n1[0], n1[1], n1[2], n1[3]};
#pragma unroll
            for (int m = 0; m < 4; m++) n1[m] = n2[m];
            const size_t pn = p + 2 * nwaves;
            if (pn < batch) {
#pragma unroll
                for (int m = 0; m < 4; m++) n2[m] = polys[pn * 256 + lane + 64 * m];
            }
            ntt_fwd_core(r, tw, lm);
            *reinterpret_cast<uint4*>(polys + p * 256 + 4 * lane) =
                make_uint4(canon_any(r[0]), canon_any(r[1]), canon_any(r[2]), canon_any(r[3]));
        }
        return;
    }
    int32_t nxt[4];
#pragma unroll
    for (int m = 0; m < 4; m++)
        nxt[m] = (V == V_NT) ? __builtin_nontemporal_load(polys + wave * 256 + lane + 64 * m) : polys[wave * 256 + lane + 64 * m];
    for (size_t p = wave; p < batch; p += nwaves) {
        int32_t r[4] = {nxt[0], nxt[1], nxt[2], nxt[3]};
        const size_t pn = p + nwaves;
        if (pn < batch) {
#pragma unroll
            for (int m = 0; m < 4; m++)
                nxt[m] = (V == V_NT) ? __builtin_nontemporal_load(polys + pn * 256 + lane + 64 * m) : polys[pn * 256 + lane + 64 * m];
        }
        if (V != V_COPY) ntt_fwd_core(r, tw, lm);
        uint4 o = (V == V_COPY) ? make_uint4(r[0], r[1], r[2], r[3])
                                : make_uint4(canon_any(r[0]), canon_any(r[1]), canon_any(r[2]), canon_any(r[3]));
        if (V == V_NT) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(polys + p * 256 + 4 * lane);
            __builtin_nontemporal_store(o.x, dst); __builtin_nontemporal_store(o.y, dst + 1);
            __builtin_nontemporal_store(o.z, dst + 2); __builtin_nontemporal_store(o.w, dst + 3);
        } else {
            *reinterpret_cast<uint4*>(polys + p * 256 + 4 * lane) = o;
        }
    }
}

// inverse variants: IV=0 strided dword stores; 1 = nt loads + nt strided stores; 2 = nt + LDS transpose -> dwordx4 nt store
template <int IV, int WPB>
__global__ __launch_bounds__(64 * WPB) void inv_variant(int32_t* __restrict__ polys, size_t batch,
                                                         const uint32_t* __restrict__ tw_tab)
{
    __shared__ __attribute__((aligned(16))) int32_t tr[WPB][260];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * WPB + wv;
    const size_t nwaves = (size_t)gridDim.x * WPB;
    if (wave >= batch) return;
    TwRegs tw;
    tw.load(tw_tab, lane);
    const LaneMasks lm(lane);
    auto ld = [&](size_t p) {
        const int32_t* s = polys + p * 256 + 4 * lane;
        if (IV == 0) return *reinterpret_cast<const int4*>(s);
        int4 v;
        v.x = __builtin_nontemporal_load(s); v.y = __builtin_nontemporal_load(s + 1);
        v.z = __builtin_nontemporal_load(s + 2); v.w = __builtin_nontemporal_load(s + 3);
        return v;
    };
    int4 nxt = ld(wave);
    for (size_t p = wave; p < batch; p += nwaves) {
        int32_t r[4] = {nxt.x, nxt.y, nxt.z, nxt.w};
        const size_t pn = p + nwaves;
        if (pn < batch) nxt = ld(pn);
        ntt_inv_core(r, tw, lm);
        if (IV == 2) {
#pragma unroll
            for (int m = 0; m < 4; m++) tr[wv][lane + 64 * m + (m)] = (int32_t)canon_small(r[m]);   // +m pad: rows of 65
            int32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { const int i = 4 * lane + j; o[j] = tr[wv][i + (i >> 6)]; }
            int32_t* d = polys + p * 256 + 4 * lane;
            __builtin_nontemporal_store(o[0], d); __builtin_nontemporal_store(o[1], d + 1);
            __builtin_nontemporal_store(o[2], d + 2); __builtin_nontemporal_store(o[3], d + 3);
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if (IV == 0) polys[p * 256 + lane + 64 * m] = (int32_t)canon_small(r[m]);
                else __builtin_nontemporal_store((int32_t)canon_small(r[m]), polys + p * 256 + lane + 64 * m);
            }
        }
    }
}

// plain dwordx4 copy (in place) as the bandwidth yardstick
__global__ __launch_bounds__(256) void copy16(uint4* p, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint4 v = p[i];
        v.x += 1;
        p[i] = v;
    }
}

template <class F>
float time_it(F&& launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) launch(i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    static uint32_t h_tab[3 * 2048];
    dil_host_twiddle_tables(h_tab, h_tab + 2048, h_tab + 4096);
    uint32_t* d_tab;
    CK(hipMalloc(&d_tab, sizeof(h_tab)));
    CK(hipMemcpy(d_tab, h_tab, sizeof(h_tab), hipMemcpyHostToDevice));
    const size_t NB = 16;                       // rotating 64 MiB batches = 1 GiB
    const size_t batch = 65536;
    int32_t* d;
    CK(hipMalloc(&d, NB * batch * 1024));
    std::vector<int32_t> h(batch * 256);
    for (size_t i = 0; i < h.size(); i++) h[i] = (int32_t)((i * 2654435761u) % 8380417u);
    for (size_t b = 0; b < NB; b++) CK(hipMemcpy(d + b * batch * 256, h.data(), batch * 1024, hipMemcpyHostToDevice));

    auto report = [&](const char* name, float ms, size_t nb) {
        printf("%-44s %8.2f us  %8.1f GB/s  %6.3f G NTT/s\n", name, ms * 1e3, 2048.0 * nb / (ms * 1e-3) / 1e9, nb / (ms * 1e-3) / 1e9);
    };
    {
        float ms = time_it([&](int i) { copy16<<<cus * 8, 256>>>((uint4*)(d + (i % NB) * batch * 256), batch * 64); }, 64);
        report("copy16 in-place 64MiB rotating", ms, batch);
        ms = time_it([&](int i) { copy16<<<cus * 8, 256>>>((uint4*)d, NB * batch * 64); }, 8);
        report("copy16 in-place 1GiB", ms, NB * batch);
    }
#define RUN(V, WPB, BPC, label)                                                                                   \
    {                                                                                                             \
        float ms = time_it([&](int i) {                                                                           \
            fwd_variant<V, WPB><<<cus * BPC, 64 * WPB>>>(d + (i % NB) * batch * 256, batch, d_tab); }, 64);       \
        report(label " batch 64Ki rotating", ms, batch);                                                          \
    }
    RUN(V_FULL, 4, 8, "full       wpb4 bpc8 ");
    RUN(V_FULL, 4, 4, "full       wpb4 bpc4 ");
    RUN(V_FULL, 4, 6, "full       wpb4 bpc6 ");
    RUN(V_FULL, 4, 2, "full       wpb4 bpc2 ");
    RUN(V_FULL, 1, 32, "full       wpb1 bpc32");
    RUN(V_FULL, 2, 16, "full       wpb2 bpc16");
    RUN(V_FULL, 8, 4, "full       wpb8 bpc4 ");
    RUN(V_NOPREFETCH, 4, 8, "noprefetch wpb4 bpc8 ");
    RUN(V_PF2, 4, 8, "prefetch2  wpb4 bpc8 ");
    RUN(V_PF2, 4, 4, "prefetch2  wpb4 bpc4 ");
    RUN(V_NT, 4, 8, "nontemporal wpb4 bpc8");
    RUN(V_NT, 4, 6, "nontemporal wpb4 bpc6");
    RUN(V_NT, 4, 5, "nontemporal wpb4 bpc5");
    RUN(V_NT, 4, 4, "nontemporal wpb4 bpc4");
    RUN(V_NT, 4, 3, "nontemporal wpb4 bpc3");
    RUN(V_NT, 8, 2, "nontemporal wpb8 bpc2");
    RUN(V_NT, 8, 3, "nontemporal wpb8 bpc3");
    RUN(V_NT, 8, 4, "nontemporal wpb8 bpc4");
    RUN(V_NT, 1, 16, "nontemporal wpb1 bpc16");
    RUN(V_NT, 1, 24, "nontemporal wpb1 bpc24");
#define RUNI(IV, WPB, BPC, label)                                                                                 \
    {                                                                                                             \
        float ms = time_it([&](int i) {                                                                           \
            inv_variant<IV, WPB><<<cus * BPC, 64 * WPB>>>(d + (i % NB) * batch * 256, batch, d_tab + 2048); }, 64); \
        report(label " batch 64Ki rotating", ms, batch);                                                          \
    }
    RUNI(0, 4, 8, "INV plain strided    wpb4 bpc8");
    RUNI(1, 4, 8, "INV nt strided       wpb4 bpc8");
    RUNI(1, 4, 6, "INV nt strided       wpb4 bpc6");
    RUNI(1, 4, 4, "INV nt strided       wpb4 bpc4");
    RUNI(2, 4, 8, "INV nt lds-transpose wpb4 bpc8");
    RUNI(2, 4, 6, "INV nt lds-transpose wpb4 bpc6");
    RUNI(2, 4, 4, "INV nt lds-transpose wpb4 bpc4");
    RUN(V_COPY, 4, 8, "copy-only  wpb4 bpc8 ");
    RUN(V_COPY, 4, 4, "copy-only  wpb4 bpc4 ");
    RUN(V_COMPUTE, 4, 1, "compute-only wpb4 bpc1");
    RUN(V_COMPUTE, 4, 2, "compute-only wpb4 bpc2");
    RUN(V_COMPUTE, 4, 3, "compute-only wpb4 bpc3");
    RUN(V_COMPUTE, 4, 5, "compute-only wpb4 bpc5");
    RUN(V_COMPUTE, 4, 6, "compute-only wpb4 bpc6");
    RUN(V_COMPUTE, 4, 8, "compute-only wpb4 bpc8");
    RUN(V_COMPUTE, 4, 4, "compute-only wpb4 bpc4");
    {
        float ms = time_it([&](int i) { fwd_variant<V_FULL, 4><<<cus * 8, 256>>>(d, NB * batch, d_tab); }, 8);
        report("full wpb4 bpc8 batch 1Mi (1 GiB)", ms, NB * batch);
        ms = time_it([&](int i) { fwd_variant<V_PF2, 4><<<cus * 8, 256>>>(d, NB * batch, d_tab); }, 8);
        report("prefetch2 wpb4 bpc8 batch 1Mi", ms, NB * batch);
        ms = time_it([&](int i) { fwd_variant<V_COPY, 4><<<cus * 8, 256>>>(d, NB * batch, d_tab); }, 8);
        report("copy-only wpb4 bpc8 batch 1Mi", ms, NB * batch);
        ms = time_it([&](int i) { fwd_variant<V_COMPUTE, 4><<<cus * 8, 256>>>(d, NB * batch, d_tab); }, 8);
        report("compute-only wpb4 bpc8 batch 1Mi", ms, NB * batch);
    }
    return 0;
}
