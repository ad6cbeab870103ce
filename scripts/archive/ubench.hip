// ubench.hip -- VALU issue-rate microbenchmark for the instructions the NTT butterflies use.
// Reports wave-instructions / ns / SIMD relative to v_add_u32 (full rate = 1 per 2 cycles).
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench.hip -o /tmp/ubench ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(X) X X X X X X X X
#define BODY(INS)                                                                        \
    asm volatile(REP8(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" \
                      INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n") \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
#define BODY3(INS)                                                                       \
    asm volatile(REP8(INS " %0, %0, %8, %0\n" INS " %1, %1, %8, %1\n" INS " %2, %2, %8, %2\n" INS " %3, %3, %8, %3\n" \
                      INS " %4, %4, %8, %4\n" INS " %5, %5, %8, %5\n" INS " %6, %6, %8, %6\n" INS " %7, %7, %8, %7\n") \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
#define BODYI(INS, IMM)                                                                  \
    asm volatile(REP8(INS " %0, %0, %8, " IMM "\n" INS " %1, %1, %8, " IMM "\n" INS " %2, %2, %8, " IMM "\n" INS " %3, %3, %8, " IMM "\n" \
                      INS " %4, %4, %8, " IMM "\n" INS " %5, %5, %8, " IMM "\n" INS " %6, %6, %8, " IMM "\n" INS " %7, %7, %8, " IMM "\n") \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
#define BODYDPP(CTRL)                                                                    \
    asm volatile(REP8("v_mov_b32_dpp %0, %1 " CTRL "\nv_mov_b32_dpp %1, %2 " CTRL "\nv_mov_b32_dpp %2, %3 " CTRL "\nv_mov_b32_dpp %3, %4 " CTRL "\n" \
                      "v_mov_b32_dpp %4, %5 " CTRL "\nv_mov_b32_dpp %5, %6 " CTRL "\nv_mov_b32_dpp %6, %7 " CTRL "\nv_mov_b32_dpp %7, %0 " CTRL "\n") \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
#define BODYSWAP(INS)                                                                    \
    asm volatile(REP8(INS " %0, %1\ns_nop 0\n" INS " %2, %3\ns_nop 0\n" INS " %4, %5\ns_nop 0\n" INS " %6, %7\ns_nop 0\n" \
                      INS " %1, %2\ns_nop 0\n" INS " %3, %4\ns_nop 0\n" INS " %5, %6\ns_nop 0\n" INS " %7, %0\ns_nop 0\n") \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));

template <int WHICH>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t c)
{
    uint32_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    for (int i = 0; i < iters; i++) {
        if (WHICH == 0) { BODY("v_add_u32") }
        if (WHICH == 1) { BODY("v_mul_u32_u24") }
        if (WHICH == 2) { BODY("v_mul_hi_u32_u24") }
        if (WHICH == 3) { BODY3("v_mad_u32_u24") }
        if (WHICH == 4) { BODYI("v_alignbit_b32", "24") }
        if (WHICH == 5) { BODY("v_mul_lo_u32") }
        if (WHICH == 6) { BODY("v_mul_hi_u32") }
        if (WHICH == 7) { BODY3("v_lshl_add_u32") }
        if (WHICH == 8) { BODYDPP("row_shr:8 row_mask:0xf bank_mask:0xc") }
        if (WHICH == 9) { BODYDPP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
        if (WHICH == 10) { BODYSWAP("v_permlane32_swap_b32") }
        if (WHICH == 11) { BODYSWAP("v_permlane16_swap_b32") }
        if (WHICH == 12) { BODY("v_min_u32") }
        if (WHICH == 13) { BODY3("v_mad_i32_i24") }
        if (WHICH == 14) { BODY("v_lshrrev_b32") }
        if (WHICH == 15) { asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c) : "vcc"); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

template <int WHICH>
double run(const char* name, uint32_t* d, int cus, double base)
{
    const int iters = 2000, per_iter = 64;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<WHICH><<<cus * 8, 256>>>(d, 10, 3);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<WHICH><<<cus * 8, 256>>>(d, iters, 3);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD: 8 waves/SIMD * iters * per_iter
    double wi = 8.0 * iters * per_iter;
    double ns = ms * 1e6;
    double rate = wi / ns;   // wave-instr per ns per SIMD
    printf("%-28s %8.3f ms  %7.4f wave-instr/ns/SIMD  rel %.3f  (cycles/instr @2.4GHz %.2f)\n", name, ms, rate,
           base > 0 ? rate / base : 1.0, 2.4 / rate);
    return rate;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("%s CUs=%d clock=%d kHz\n", p.name, cus, p.clockRate);
    uint32_t* d;
    hipMalloc(&d, cus * 8 * 256 * 4);
    double base = run<0>("v_add_u32", d, cus, 0);
    run<1>("v_mul_u32_u24", d, cus, base);
    run<2>("v_mul_hi_u32_u24", d, cus, base);
    run<3>("v_mad_u32_u24", d, cus, base);
    run<4>("v_alignbit_b32", d, cus, base);
    run<5>("v_mul_lo_u32", d, cus, base);
    run<6>("v_mul_hi_u32", d, cus, base);
    run<7>("v_lshl_add_u32", d, cus, base);
    run<8>("v_mov_dpp row_shr", d, cus, base);
    run<9>("v_mov_dpp quad_perm", d, cus, base);
    run<10>("v_permlane32_swap(+s_nop)", d, cus, base);
    run<11>("v_permlane16_swap(+s_nop)", d, cus, base);
    run<12>("v_min_u32", d, cus, base);
    run<13>("v_mad_i32_i24", d, cus, base);
    run<14>("v_lshrrev_b32", d, cus, base);
    run<15>("v_cndmask_b32_e64(vcc-less)", d, cus, base);
    return 0;
}
