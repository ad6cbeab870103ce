// ubench3.hip -- Keccak-f[1600] issue-rate probe: the permutation alone (state in registers, no memory), for different
// resident waves per SIMD and block sizes, plus a synthetic xor/alignbit/bfi mix with the same instruction ratio.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../dilithium_amd/csrc/keccak.hpp"

template <int BS>
__global__ __launch_bounds__(BS) void kperm(uint64_t* out, int perms)
{
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = (uint64_t)(threadIdx.x + blockIdx.x * BS) * 0x9E3779B97F4A7C15ull + i;
    for (int p = 0; p < perms; p++) dil::keccak_f1600(a);
    uint64_t x = 0;
#pragma unroll
    for (int i = 0; i < 25; i++) x ^= a[i];
    out[(size_t)blockIdx.x * BS + threadIdx.x] = x;
}

#define R8(X) X X X X X X X X
__global__ __launch_bounds__(256) void kmix(uint32_t* out, int iters, uint32_t c)
{
    uint32_t r0 = threadIdx.x * 0x9E3779B9u, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3, r4 = r0 * 9 + 4, r5 = r0 * 11 + 5, r6 = r0 * 13 + 6, r7 = r0 * 15 + 7;
    for (int i = 0; i < iters; i++) {
        // 40 instrs: 24 xor, 8 alignbit, 8 bfi  (Keccak round ratio ~ 152:58:50)
        asm volatile(R8("v_xor_b32 %0, %0, %8\nv_xor_b32 %1, %1, %8\nv_alignbit_b32 %2, %2, %3, 7\nv_xor_b32 %4, %4, %8\nv_bfi_b32 %5, %5, %8, %6\n")
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

template <int BS>
void run(uint64_t* d, int cus, int waves_per_simd)
{
    const int perms = 200;
    const int blocks = cus * 4 * waves_per_simd * 64 / BS;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    kperm<BS><<<blocks, BS>>>(d, 5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    kperm<BS><<<blocks, BS>>>(d, perms);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    double total = (double)blocks * BS * perms;
    double cyc = ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * perms * 24);     // SIMD cycles per wave-round
    printf("keccak perm BS=%3d waves/SIMD=%d: %8.3f ms  %7.3f G perm/s   %6.0f SIMD-cycles per wave-round (@2.4GHz)\n", BS, waves_per_simd, ms,
           total / (ms * 1e6), cyc);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    uint64_t* d;
    (void)hipMalloc(&d, (size_t)cus * 4 * 16 * 64 * 8);
    for (int w : {1, 2, 4, 6, 8}) run<64>(d, cus, w);
    for (int w : {1, 2, 4, 6, 8}) run<256>(d, cus, w);
    {
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        for (int bpc : {1, 2, 4, 8}) {
            const int iters = 2000;
            kmix<<<cus * bpc, 256>>>((uint32_t*)d, 10, 123);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            kmix<<<cus * bpc, 256>>>((uint32_t*)d, iters, 123);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms;
            (void)hipEventElapsedTime(&ms, a, b);
            double wi = (double)bpc * iters * 40;
            printf("mix 24xor:8alignbit:8bfi waves/SIMD=%d: %7.3f ms  %.2f cycles/instr @2.4GHz (expected %.2f from isolated rates)\n", bpc, ms,
                   ms * 1e-3 * 2.4e9 / wi, (24 * 2.6 + 16 * 4.4) / 40);
        }
    }
    return 0;
}
