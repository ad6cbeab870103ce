// tune_verify_mem.hip -- the memory side of the fused level-3 verify core (pipelines.hip verify_wpi_kernel) as a skeleton: the same
// operands in the same order from the same five arrays, a key per item, persistent waves, one row of A ahead -- and a checksum instead
// of the arithmetic.  Under four rotating input sets the real kernel runs at its memory-only time (68.2 vs 67.5 us, profiles/
// r05j_ab_verify_sets.txt) = 5.6 TB/s, where a plain read-only stream reaches 6.4-7.2 TB/s on this chip (profiles/r01_membench.txt):
// which property of the access pattern costs the difference?
// Answer (profiles/r05k_verify_mem_skeleton.txt): none of the read side's.  Without the w1 stores the same reads take 58.8 us = 6.4 TB/s,
// the plain read-only rate of this chip; the 12.6 MB of w1 (3.5 % of the bytes) cost the other 6.7 us (10 %).  The cost follows the bytes
// stored (0.8 MB: +1.2 us, 2.1 MB: +2.6 us, 12.6 MB: +6.7 us -- a written byte costs about three read ones in this stream), most of it
// even when the rows land in an L2-resident window, and neither the store form (per item, nt / sc0 / sc1 policies, 128-B rows, x2 / x4 by
// fewer lanes, two or three rows per instruction: 63.3 - 66.6 us) nor its place in the wave's instruction stream (every row held back until
// the wave's last load has returned: 65.3 us) moves it.  WR variants that leave some lane's result unused let the compiler drop loads
// (an early run "found" 51 us that way): every form below keeps every lane of every row live.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/tune_verify_mem.hip -o scripts/bin/tune_verify_mem
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int K = 6, L = 5;

__device__ __forceinline__ int4 ld4(const int32_t* p, bool nt)
{
    int4 v;
    if (nt) {
        v.x = __builtin_nontemporal_load(p);
        v.y = __builtin_nontemporal_load(p + 1);
        v.z = __builtin_nontemporal_load(p + 2);
        v.w = __builtin_nontemporal_load(p + 3);
    } else {
        v = *reinterpret_cast<const int4*>(p);
    }
    return v;
}
__device__ __forceinline__ int32_t ld1(const int32_t* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ int32_t fold(int4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// SMALL4: z / c / t1 as one dwordx4 per lane and polynomial (else four strided dwords, as the transforms want them)
// NT_SMALL: non-temporal policy for z / c / t1 / h too        AHEAD: rows of A in flight beyond the current one (1 = the kernel)
// WPS: waves per SIMD asked of the allocator                  CONTIG: a workgroup walks a contiguous run of items
template <bool SMALL4, bool NT_SMALL, int AHEAD, int WPS, bool CONTIG, int WR = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void skel(uint8_t* __restrict__ w1, const int32_t* __restrict__ A,
                                                                                           const int32_t* __restrict__ z, const int32_t* __restrict__ c,
                                                                                           const int32_t* __restrict__ t1, const uint8_t* __restrict__ h,
                                                                                           size_t batch)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t nwaves = (size_t)gridDim.x * 4;
    size_t it, step, end;
    if (CONTIG) {
        const size_t per = (batch + gridDim.x - 1) / gridDim.x;
        it = (size_t)blockIdx.x * per + wv;
        end = std::min(batch, (size_t)(blockIdx.x + 1) * per);
        step = 4;
    } else {
        it = (size_t)blockIdx.x * 4 + wv;
        end = batch;
        step = nwaves;
    }
    auto small = [&](const int32_t* p) -> int32_t {
        if (SMALL4) return fold(ld4(p + 4 * lane, NT_SMALL));
        int32_t s = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) s ^= ld1(p + lane + 64 * m, NT_SMALL);
        return s;
    };
    int32_t zc[L + 1];
    auto load_zc = [&](size_t i) {
#pragma unroll
        for (int l = 0; l < L; l++) zc[l] = small(z + (i * L + l) * 256);
        zc[L] = small(c + i * 256);
    };
    if (it < end) load_zc(it);
    uint32_t keep[3][K] = {};          // WR == 15: every row of the wave's (at most three) items held back until its last load has returned
    size_t kept_it[3] = {};
    int trip = 0;
    for (; it < end; it += step, trip++) {
        const int32_t* Ait = A + it * (size_t)(K * L) * 256;
        const int32_t* t1it = t1 + it * (size_t)K * 256;
        const uint8_t* hit = h + it * K * 256;
        int4 Ar[AHEAD][L];
#pragma unroll
        for (int a = 0; a < AHEAD; a++)
#pragma unroll
            for (int l = 0; l < L; l++) Ar[a][l] = ld4(Ait + (size_t)(a * L + l) * 256 + 4 * lane, true);
        int32_t tn = small(t1it);
        uint32_t hn = reinterpret_cast<const uint32_t*>(hit)[lane];
        int32_t acc = 0;
        uint32_t rows[K] = {};
#pragma unroll
        for (int l = 0; l <= L; l++) acc ^= zc[l];
        const size_t itn = it + step;
        if (itn < end) load_zc(itn);
#pragma unroll
        for (int k = 0; k < K; k++) {
            int32_t r = acc ^ tn ^ (int32_t)hn;
#pragma unroll
            for (int l = 0; l < L; l++) r ^= fold(Ar[0][l]);
#pragma unroll
            for (int a = 0; a + 1 < AHEAD; a++)
#pragma unroll
                for (int l = 0; l < L; l++) Ar[a][l] = Ar[a + 1][l];
            if (k + AHEAD < K) {
#pragma unroll
                for (int l = 0; l < L; l++) Ar[AHEAD - 1][l] = ld4(Ait + (size_t)((k + AHEAD) * L + l) * 256 + 4 * lane, true);
            }
            if (k + 1 < K) {
                tn = small(t1it + (k + 1) * 256);
                hn = reinterpret_cast<const uint32_t*>(hit + (k + 1) * 256)[lane];
            }
            uint32_t* wp = reinterpret_cast<uint32_t*>(w1 + (it * K + k) * 256) + lane;
            if (WR == 1) *wp = (uint32_t)r;
            else if (WR == 3) __builtin_nontemporal_store((uint32_t)r, wp);
            else if (WR == 16) { if (k == 0 && trip == 0) *wp = (uint32_t)r; else rows[k] = (uint32_t)r; }    // one row per wave and launch
            else if (WR == 17) { if (k == 0) *wp = (uint32_t)r; else rows[k] = (uint32_t)r; }                 // one row per item
            else if (WR == 4) reinterpret_cast<uint32_t*>(w1 + ((it & 1023) * K + k) * 256)[lane] = (uint32_t)r;   // 1.5 MiB, L2-resident
            else if (WR == 5) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(wp), "v"(r) : "memory");
            else if (WR == 6) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(wp), "v"(r) : "memory");
            else if (WR == 7) asm volatile("global_store_dword %0, %1, off nt" ::"v"(wp), "v"(r) : "memory");
            else if (WR == 8) asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(wp), "v"(r) : "memory");
            // (the narrower forms gather their dwords from the other lanes, as the real re-layout would: every lane's r stays live)
            else if (WR == 10) {
                const uint32_t a = __shfl((uint32_t)r, (2 * lane) & 63), b = __shfl((uint32_t)r, (2 * lane + 1) & 63);
                if (lane < 32) *reinterpret_cast<uint2*>(w1 + (it * K + k) * 256 + 8 * lane) = make_uint2(a, b);
            } else if (WR == 11) {
                uint32_t q[4];
#pragma unroll
                for (int j = 0; j < 4; j++) q[j] = __shfl((uint32_t)r, (4 * lane + j) & 63);
                if (lane < 16) *reinterpret_cast<uint4*>(w1 + (it * K + k) * 256 + 16 * lane) = make_uint4(q[0], q[1], q[2], q[3]);
            } else if (WR == 12) *reinterpret_cast<uint16_t*>(w1 + (it * K + k) * 128 + 2 * lane) = (uint16_t)r;
            else if (WR == 13) {          // two rows per store instruction: 512 B as dwordx2 from 64 lanes, every other row
                if (k & 1) *reinterpret_cast<uint2*>(w1 + (it * K + k - 1) * 256 + 8 * lane) = make_uint2(rows[k - 1], (uint32_t)r);
                else rows[k] = (uint32_t)r;
            } else if (WR == 14) {        // three rows per store instruction (768 B as dwordx3... as x4 of which one is padding: dwordx3)
                if (k % 3 == 2) {
                    uint32_t* o = reinterpret_cast<uint32_t*>(w1 + (it * K + k - 2) * 256) + 3 * lane;
                    *reinterpret_cast<uint3*>(o) = make_uint3(rows[k - 2], rows[k - 1], (uint32_t)r);
                } else rows[k] = (uint32_t)r;
            } else if (WR == 9) {
                const uint32_t a = (uint32_t)r ^ __shfl((uint32_t)r, lane ^ 32);
                if (lane < 32) *reinterpret_cast<uint32_t*>(w1 + (it * K + k) * 128 + 4 * lane) = a;   // 128 B per row (packed)
            }
            else rows[k] = (uint32_t)r;
        }
        if (WR == 15) {
#pragma unroll
            for (int t = 0; t < 3; t++)
                if (t == trip) {
                    kept_it[t] = it;
#pragma unroll
                    for (int k = 0; k < K; k++) keep[t][k] = rows[k];
                }
        }
        if (WR == 2) {                 // the item's K rows of w1 in one go (1.5 KiB contiguous)
#pragma unroll
            for (int k = 0; k < K; k++) reinterpret_cast<uint32_t*>(w1 + (it * K + k) * 256)[lane] = rows[k];
        } else if (WR == 0 || WR == 16 || WR == 17) {
            uint32_t x = 0;
#pragma unroll
            for (int k = 0; k < K; k++) x = x * 31u + rows[k];     // (a plain xor would cancel the K copies of acc and with them the z / c loads)
            if (x == 0x12345678u) w1[it] = 1;
        }
    }
    if (WR == 15) {
#pragma unroll
        for (int t = 0; t < 3; t++)
            if (t < trip) {
#pragma unroll
                for (int k = 0; k < K; k++) reinterpret_cast<uint32_t*>(w1 + (kept_it[t] * K + k) * 256)[lane] = keep[t][k];
            }
    }
}

// K waves share an item: wave k owns row k of A (5 KiB), one of the L + 1 time-domain polynomials (z_0 .. z_{L-1}, c), t1[k], h[k], w1[k]
// -- per wave and trip the "one-row item" of a_only<1>; the item's transformed z^ would meet in LDS, so a workgroup barrier per item
// (BARRIER) stands in for that hand-over.  PF: the next item's row is loaded before the barrier.
template <bool BARRIER, int WGS_ITEMS>
__global__ __launch_bounds__(64 * K) void skel_rows(uint8_t* __restrict__ w1, const int32_t* __restrict__ A, const int32_t* __restrict__ z,
                                                    const int32_t* __restrict__ c, const int32_t* __restrict__ t1, const uint8_t* __restrict__ h, size_t batch)
{
    __shared__ int32_t share[K * 64];
    const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    auto row = [&](size_t it, int4 (&a)[L], int32_t& zc, int32_t& tn, uint32_t& hn) {
#pragma unroll
        for (int l = 0; l < L; l++) a[l] = ld4(A + (it * K + k) * (size_t)L * 256 + l * 256 + 4 * lane, true);
        const int32_t* zp = k < L ? z + (it * L + k) * 256 : c + it * 256;        // (K = L + 1 at level 3)
        zc = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) zc ^= zp[lane + 64 * m];
        tn = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) tn ^= t1[(it * K + k) * 256 + lane + 64 * m];
        hn = reinterpret_cast<const uint32_t*>(h + (it * K + k) * 256)[lane];
    };
    int4 a[L];
    int32_t zc, tn;
    uint32_t hn;
    size_t it = blockIdx.x;
    if (it < batch) row(it, a, zc, tn, hn);
    for (; it < batch; it += gridDim.x) {
        int32_t r = zc ^ tn ^ (int32_t)hn;
#pragma unroll
        for (int l = 0; l < L; l++) r ^= fold(a[l]);
        const size_t itn = it + gridDim.x;
        if (itn < batch) row(itn, a, zc, tn, hn);
        if (BARRIER) {
            share[k * 64 + lane] = r;
            __syncthreads();
            r ^= share[((k + 1) % K) * 64 + lane];
            __syncthreads();
        }
        reinterpret_cast<uint32_t*>(w1 + (it * K + k) * 256)[lane] = (uint32_t)r;
    }
}

// One RECORD per item (an API change: SURVEY 8(d) fixes separate operand arrays): [A 30 KiB | z 5 | c 1 | t1 6 | h 1.5 KiB] contiguous,
// 44544 B, read front to back by the item's wave in 5-KiB pieces; w1 to its own array.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void skel_aos(uint8_t* __restrict__ w1, const int32_t* __restrict__ rec, size_t batch)
{
    constexpr int REC_DW = (K * L + L + 1 + K) * 256 + K * 64;      // dwords per record
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t it = (size_t)blockIdx.x * 4 + wv; it < batch; it += nwaves) {
        const int32_t* p = rec + it * (size_t)REC_DW;
        int32_t acc = 0;
        int4 cur[L];
#pragma unroll
        for (int l = 0; l < L; l++) cur[l] = ld4(p + l * 256 + 4 * lane, true);
        constexpr int PIECES = (K * L + L + 1 + K) / L;             // 8 pieces of 5 KiB (+ 2 polynomials + the hints)
        for (int q = 0; q < PIECES; q++) {
            int4 nx[L];
            if (q + 1 < PIECES) {
#pragma unroll
                for (int l = 0; l < L; l++) nx[l] = ld4(p + (size_t)((q + 1) * L + l) * 256 + 4 * lane, true);
            }
#pragma unroll
            for (int l = 0; l < L; l++) acc ^= fold(cur[l]);
#pragma unroll
            for (int l = 0; l < L; l++) cur[l] = nx[l];
            if (q < K) reinterpret_cast<uint32_t*>(w1 + (it * K + q) * 256)[lane] = (uint32_t)acc;
        }
        const int4 t0 = ld4(p + (size_t)(PIECES * L) * 256 + 4 * lane, true), t1v = ld4(p + (size_t)(PIECES * L + 1) * 256 + 4 * lane, true);
        const int32_t hh = p[(K * L + L + 1 + K) * 256 + lane];
        if ((acc ^ fold(t0) ^ fold(t1v) ^ hh) == 0x12345678) w1[it] = 1;
    }
}

// references in the same run: a plain grid-stride read of the same bytes (what the chip gives a read-only stream today), and the
// matrix stream alone in the kernel's order (an item's 30 KiB per wave, 5 KiB at a time)
__global__ __launch_bounds__(256) void plain_read(uint32_t* __restrict__ out, const int4* __restrict__ p, size_t nvec)
{
    int32_t s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) s ^= fold(ld4(reinterpret_cast<const int32_t*>(p + i), true));
    if (s == 0x12345678) out[threadIdx.x] = (uint32_t)s;
}
template <int ROWS_PER_ITEM>
__global__ __launch_bounds__(256) void a_only(uint32_t* __restrict__ out, const int32_t* __restrict__ A, size_t batch)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t nwaves = (size_t)gridDim.x * 4;
    int32_t s = 0;
    for (size_t it = (size_t)blockIdx.x * 4 + wv; it < batch; it += nwaves) {
        const int32_t* Ait = A + it * (size_t)(ROWS_PER_ITEM * L) * 256;
        int4 cur[L];
#pragma unroll
        for (int l = 0; l < L; l++) cur[l] = ld4(Ait + l * 256 + 4 * lane, true);
        for (int k = 0; k < ROWS_PER_ITEM; k++) {
            int4 nx[L];
            if (k + 1 < ROWS_PER_ITEM) {
#pragma unroll
                for (int l = 0; l < L; l++) nx[l] = ld4(Ait + (size_t)((k + 1) * L + l) * 256 + 4 * lane, true);
            }
#pragma unroll
            for (int l = 0; l < L; l++) s ^= fold(cur[l]);
#pragma unroll
            for (int l = 0; l < L; l++) cur[l] = nx[l];
        }
    }
    if (s == 0x12345678) out[threadIdx.x] = (uint32_t)s;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t n = 8192, NS = 4;
    const size_t bA = n * K * L * 1024, bz = n * L * 1024, bc = n * 1024, bt = n * K * 1024, bh = n * K * 256;
    int32_t *A, *z, *c, *t1;
    uint8_t *h, *w1;
    CK(hipMalloc(&A, NS * bA)); CK(hipMalloc(&z, NS * bz)); CK(hipMalloc(&c, NS * bc)); CK(hipMalloc(&t1, NS * bt)); CK(hipMalloc(&h, NS * bh));
    CK(hipMalloc(&w1, NS * bh));
    CK(hipMemset(A, 1, NS * bA)); CK(hipMemset(z, 2, NS * bz)); CK(hipMemset(c, 3, NS * bc)); CK(hipMemset(t1, 4, NS * bt)); CK(hipMemset(h, 0, NS * bh));
    const double bytes = (double)(bA + bz + bc + bt + 2 * bh);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN(label, KERN, BPC)                                                                                                      \
    {                                                                                                                              \
        auto go = [&](int i) {                                                                                                     \
            const size_t s = i % NS;                                                                                               \
            KERN<<<cus * BPC, 256>>>(w1 + s * bh, A + s * (bA / 4), z + s * (bz / 4), c + s * (bc / 4), t1 + s * (bt / 4), h + s * bh, n); \
        };                                                                                                                         \
        for (int i = 0; i < 40; i++) go(i);                                                                                        \
        CK(hipDeviceSynchronize());                                                                                                \
        std::vector<float> t;                                                                                                      \
        for (int rep = 0; rep < 5; rep++) {                                                                                        \
            CK(hipEventRecord(e0));                                                                                                \
            for (int i = 0; i < 200; i++) go(i);                                                                                   \
            CK(hipEventRecord(e1));                                                                                                \
            CK(hipEventSynchronize(e1));                                                                                           \
            float ms;                                                                                                              \
            CK(hipEventElapsedTime(&ms, e0, e1));                                                                                  \
            t.push_back(ms / 200 * 1e3f);                                                                                          \
        }                                                                                                                          \
        std::sort(t.begin(), t.end());                                                                                             \
        printf("%-86s %7.2f us  %6.0f GB/s  %.3f of 8 TB/s\n", label, t[2], bytes / t[2] / 1e3, bytes / t[2] / 1e3 / 8000);        \
    }
    {
        auto tm = [&](const char* label, auto go, double by) {
            for (int i = 0; i < 20; i++) go(i);
            CK(hipDeviceSynchronize());
            std::vector<float> t;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0));
                for (int i = 0; i < 100; i++) go(i);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                t.push_back(ms / 100 * 1e3f);
            }
            std::sort(t.begin(), t.end());
            printf("%-86s %7.2f us  %6.0f GB/s  %.3f of 8 TB/s\n", label, t[2], by / t[2] / 1e3, by / t[2] / 1e3 / 8000);
        };
        for (int bpc : {4, 8, 16})
            tm(bpc == 4 ? "plain grid-stride nt read of one set's A (240 MiB), 4 blocks/CU" : bpc == 8 ? "  8 blocks/CU" : "  16 blocks/CU",
               [&](int i) { plain_read<<<cus * bpc, 256>>>(reinterpret_cast<uint32_t*>(w1), reinterpret_cast<const int4*>(A + (i % NS) * (bA / 4)), bA / 16); }, (double)bA);
        tm("the matrix stream alone, kernel order (item = 30 KiB per wave, a 5-KiB row ahead), 3 blocks/CU",
           [&](int i) { a_only<K><<<cus * 3, 256>>>(reinterpret_cast<uint32_t*>(w1), A + (i % NS) * (bA / 4), n); }, (double)bA);
        tm("  the same bytes as 49152 items of one row (5 KiB per wave and trip)",
           [&](int i) { a_only<1><<<cus * 3, 256>>>(reinterpret_cast<uint32_t*>(w1), A + (i % NS) * (bA / 4), n * K); }, (double)bA);
        tm("  6 blocks/CU", [&](int i) { a_only<K><<<cus * 6, 256>>>(reinterpret_cast<uint32_t*>(w1), A + (i % NS) * (bA / 4), n); }, (double)bA);
    }
    // warm
    RUN("(warm-up)", (skel<false, false, 1, 3, false>), 3)
    RUN("as the kernel: strided dwords (default policy) for z c t1, A nt x4 one row ahead, 3 waves/SIMD", (skel<false, false, 1, 3, false>), 3)
    RUN("  no w1 stores at all (reads only)", (skel<false, false, 1, 3, false, 0>), 3)
    RUN("  w1 stored once per item (1.5 KiB) instead of once per row", (skel<false, false, 1, 3, false, 2>), 3)
    RUN("  every w1 row held in registers until the wave's last load is back, stored then", (skel<false, false, 1, 3, false, 15>), 3)
    RUN("  ONE row stored per wave and launch (0.8 MB), the others only summed", (skel<false, false, 1, 3, false, 16>), 3)
    RUN("  one row of the six stored per item (2.1 MB)", (skel<false, false, 1, 3, false, 17>), 3)
    RUN("  w1 rows to a 1.5-MiB window that stays in L2 (no DRAM writes)", (skel<false, false, 1, 3, false, 4>), 3)
    RUN("  w1 rows with __builtin_nontemporal_store", (skel<false, false, 1, 3, false, 3>), 3)
    RUN("  w1 rows with global_store_dword nt", (skel<false, false, 1, 3, false, 7>), 3)
    RUN("  w1 rows with global_store_dword sc1", (skel<false, false, 1, 3, false, 6>), 3)
    RUN("  w1 rows with global_store_dword sc0 sc1", (skel<false, false, 1, 3, false, 5>), 3)
    RUN("  w1 rows with global_store_dword sc0 sc1 nt", (skel<false, false, 1, 3, false, 8>), 3)
    RUN("  w1 rows packed to 128 B (half the lanes store a dword)", (skel<false, false, 1, 3, false, 9>), 3)
    RUN("  w1 rows packed to 128 B (64 lanes store a ushort)", (skel<false, false, 1, 3, false, 12>), 3)
    RUN("  256-B rows stored by 32 lanes as dwordx2", (skel<false, false, 1, 3, false, 10>), 3)
    RUN("  256-B rows stored by 16 lanes as dwordx4", (skel<false, false, 1, 3, false, 11>), 3)
    RUN("  two rows per store (512 B as dwordx2 of 64 lanes)", (skel<false, false, 1, 3, false, 13>), 3)
    RUN("  three rows per store (768 B as dwordx3 of 64 lanes)", (skel<false, false, 1, 3, false, 14>), 3)
    RUN("  + nt on z c t1", (skel<false, true, 1, 3, false>), 3)
    RUN("  z c t1 as dwordx4", (skel<true, false, 1, 3, false>), 3)
    RUN("  z c t1 as dwordx4 + nt", (skel<true, true, 1, 3, false>), 3)
    RUN("  A two rows ahead", (skel<false, false, 2, 3, false>), 3)
    RUN("  A two rows ahead, small x4 nt", (skel<true, true, 2, 3, false>), 3)
    RUN("  A three rows ahead, small x4 nt", (skel<true, true, 3, 3, false>), 3)
    RUN("  contiguous run of items per workgroup", (skel<false, false, 1, 3, true>), 3)
    RUN("  contiguous + small x4 nt + A two ahead", (skel<true, true, 2, 3, true>), 3)
    RUN("  2 waves/SIMD", (skel<false, false, 1, 2, false>), 2)
    RUN("  2 waves/SIMD, small x4 nt, A two ahead", (skel<true, true, 2, 2, false>), 2)
    RUN("  4 waves/SIMD", (skel<false, false, 1, 4, false>), 4)
    RUN("  4 waves/SIMD, small x4 nt", (skel<true, true, 1, 4, false>), 4)
    RUN("  6 waves/SIMD, small x4 nt", (skel<true, true, 1, 6, false>), 6)
    RUN("  8 waves/SIMD, small x4 nt", (skel<true, true, 1, 8, false>), 8)
#define RUNR(label, KERN, BLOCKS)                                                                                                  \
    {                                                                                                                              \
        auto go = [&](int i) {                                                                                                     \
            const size_t s = i % NS;                                                                                               \
            KERN<<<BLOCKS, 64 * K>>>(w1 + s * bh, A + s * (bA / 4), z + s * (bz / 4), c + s * (bc / 4), t1 + s * (bt / 4), h + s * bh, n); \
        };                                                                                                                         \
        for (int i = 0; i < 40; i++) go(i);                                                                                        \
        CK(hipDeviceSynchronize());                                                                                                \
        std::vector<float> t;                                                                                                      \
        for (int rep = 0; rep < 5; rep++) {                                                                                        \
            CK(hipEventRecord(e0));                                                                                                \
            for (int i = 0; i < 200; i++) go(i);                                                                                   \
            CK(hipEventRecord(e1));                                                                                                \
            CK(hipEventSynchronize(e1));                                                                                           \
            float ms;                                                                                                              \
            CK(hipEventElapsedTime(&ms, e0, e1));                                                                                  \
            t.push_back(ms / 200 * 1e3f);                                                                                          \
        }                                                                                                                          \
        std::sort(t.begin(), t.end());                                                                                             \
        printf("%-86s %7.2f us  %6.0f GB/s  %.3f of 8 TB/s\n", label, t[2], bytes / t[2] / 1e3, bytes / t[2] / 1e3 / 8000);        \
    }
    {
        int32_t* rec;
        const size_t rec_b = ((size_t)(K * L + L + 1 + K) * 1024 + K * 256);
        CK(hipMalloc(&rec, NS * n * rec_b));
        CK(hipMemset(rec, 5, NS * n * rec_b));
        auto tm2 = [&](const char* label, int bpc) {
            auto go = [&](int i) { skel_aos<<<cus * bpc, 256>>>(w1 + (i % NS) * bh, rec + (i % NS) * (n * rec_b / 4), n); };
            for (int i = 0; i < 40; i++) go(i);
            CK(hipDeviceSynchronize());
            std::vector<float> t;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0));
                for (int i = 0; i < 200; i++) go(i);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                t.push_back(ms / 200 * 1e3f);
            }
            std::sort(t.begin(), t.end());
            printf("%-86s %7.2f us  %6.0f GB/s  %.3f of 8 TB/s\n", label, t[2], bytes / t[2] / 1e3, bytes / t[2] / 1e3 / 8000);
        };
        tm2("one contiguous 44.5-KiB record per item (AoS), wave per item, 3 blocks/CU", 3);
        tm2("  4 blocks/CU", 4);
        CK(hipFree(rec));
    }
    RUNR("K waves share an item (a row each), no barrier, 2 workgroups per CU", (skel_rows<false, 0>), cus * 2)
    RUNR("  3 workgroups per CU", (skel_rows<false, 0>), cus * 3)
    RUNR("  4 workgroups per CU", (skel_rows<false, 0>), cus * 4)
    RUNR("  with a workgroup barrier pair per item, 2 per CU", (skel_rows<true, 0>), cus * 2)
    RUNR("  with a workgroup barrier pair per item, 3 per CU", (skel_rows<true, 0>), cus * 3)
    RUNR("  with a workgroup barrier pair per item, 4 per CU", (skel_rows<true, 0>), cus * 4)
    RUNR("  with a workgroup barrier pair per item, 5 per CU", (skel_rows<true, 0>), cus * 5)
    return 0;
}
