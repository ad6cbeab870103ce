// ubench2.hip -- second VALU issue-rate probe: candidates for a 32-bit (Montgomery/Shoup) butterfly
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define R8(X) X X X X X X X X
#define OPS(a) "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
#define B2(INS) asm volatile(R8(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n") : OPS(0) : "v"(c));
#define B2R(INS) asm volatile(R8(INS " %0, %8, %0\n" INS " %1, %8, %1\n" INS " %2, %8, %2\n" INS " %3, %8, %3\n" INS " %4, %8, %4\n" INS " %5, %8, %5\n" INS " %6, %8, %6\n" INS " %7, %8, %7\n") : OPS(0) : "v"(c));
#define B3(INS) asm volatile(R8(INS " %0, %0, %8, %0\n" INS " %1, %1, %8, %1\n" INS " %2, %2, %8, %2\n" INS " %3, %3, %8, %3\n" INS " %4, %4, %8, %4\n" INS " %5, %5, %8, %5\n" INS " %6, %6, %8, %6\n" INS " %7, %7, %8, %7\n") : OPS(0) : "v"(c));
#define B3S(INS) asm volatile(R8(INS " %0, %0, %8, s[10:11]\n" INS " %1, %1, %8, s[10:11]\n" INS " %2, %2, %8, s[10:11]\n" INS " %3, %3, %8, s[10:11]\n" INS " %4, %4, %8, s[10:11]\n" INS " %5, %5, %8, s[10:11]\n" INS " %6, %6, %8, s[10:11]\n" INS " %7, %7, %8, s[10:11]\n") : OPS(0) : "v"(c) : "s10", "s11");
#define BVCC(INS) asm volatile(R8(INS " %0, %0, %8, vcc\n" INS " %1, %1, %8, vcc\n" INS " %2, %2, %8, vcc\n" INS " %3, %3, %8, vcc\n" INS " %4, %4, %8, vcc\n" INS " %5, %5, %8, vcc\n" INS " %6, %6, %8, vcc\n" INS " %7, %7, %8, vcc\n") : OPS(0) : "v"(c));
#define B64(INS) asm volatile(R8(INS " %0, s[10:11], %5, %4, %0\n" INS " %1, s[10:11], %6, %4, %1\n" INS " %2, s[10:11], %7, %4, %2\n" INS " %3, s[10:11], %8, %4, %3\n" INS " %0, s[10:11], %6, %4, %0\n" INS " %1, s[10:11], %7, %4, %1\n" INS " %2, s[10:11], %8, %4, %2\n" INS " %3, s[10:11], %5, %4, %3\n") : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(c), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "s10", "s11");
#define BDS(INS, EXTRA) asm volatile(R8(INS " %0, %0" EXTRA "\n" INS " %1, %1" EXTRA "\n" INS " %2, %2" EXTRA "\n" INS " %3, %3" EXTRA "\n" INS " %4, %4" EXTRA "\n" INS " %5, %5" EXTRA "\n" INS " %6, %6" EXTRA "\n" INS " %7, %7" EXTRA "\ns_waitcnt lgkmcnt(0)\n") : OPS(0) : "v"(c));
#define BDS2(INS) asm volatile(R8(INS " %0, %8, %0\n" INS " %1, %8, %1\n" INS " %2, %8, %2\n" INS " %3, %8, %3\n" INS " %4, %8, %4\n" INS " %5, %8, %5\n" INS " %6, %8, %6\n" INS " %7, %8, %7\ns_waitcnt lgkmcnt(0)\n") : OPS(0) : "v"(c));
#define BDPP2(INS, CTRL) asm volatile(R8(INS " %0, %1, %0 " CTRL "\n" INS " %1, %2, %1 " CTRL "\n" INS " %2, %3, %2 " CTRL "\n" INS " %3, %4, %3 " CTRL "\n" INS " %4, %5, %4 " CTRL "\n" INS " %5, %6, %5 " CTRL "\n" INS " %6, %7, %6 " CTRL "\n" INS " %7, %0, %7 " CTRL "\n") : OPS(0) : "v"(c));

template <int W>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t c)
{
    uint32_t r0 = threadIdx.x * 0x9E3779B9u, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3, r4 = r0 * 9 + 4, r5 = r0 * 11 + 5, r6 = r0 * 13 + 6, r7 = r0 * 15 + 7;
    uint64_t q0 = r0, q1 = r1, q2 = r2, q3 = r3;
    asm volatile("s_mov_b32 s10, 0x55555555\ns_mov_b32 s11, 0x55555555" ::: "s10", "s11");
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\ns_mov_b32 vcc_hi, 0x55555555" ::: "vcc");
    for (int i = 0; i < iters; i++) {
        if (W == 0) { B2("v_add_u32") }
        if (W == 1) { B2("v_sub_u32") }
        if (W == 2) { B2("v_mul_lo_u32") }
        if (W == 3) { B2("v_mul_hi_u32") }
        if (W == 4) { B2("v_mul_hi_i32") }
        if (W == 5) { B3("v_add3_u32") }
        if (W == 6) { B2R("v_ashrrev_i32") }
        if (W == 7) { B2("v_and_b32") }
        if (W == 8) { B2("v_xor_b32") }
        if (W == 9) { B2("v_max_i32") }
        if (W == 10) { BVCC("v_cndmask_b32_e32") }
        if (W == 11) { B3S("v_cndmask_b32_e64") }
        if (W == 12) { B64("v_mad_u64_u32") }
        if (W == 13) { B3("v_bfi_b32") }
        if (W == 14) { BDS("ds_swizzle_b32", " offset:swizzle(SWAP,1)") }
        if (W == 15) { BDS2("ds_bpermute_b32") }
        if (W == 16) { BDPP2("v_add_u32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
        if (W == 17) { BDPP2("v_cndmask_b32_dpp", ", vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
        if (W == 18) { B3("v_mad_u32_u24") }
        if (W == 19) { B2("v_mul_u32_u24") }
        if (W == 20) { B3("v_perm_b32") }
        if (W == 21) { B2("v_min_u32") }
        if (W == 22) { B2("v_lshlrev_b32") }
        if (W == 23) { B3("v_lshl_or_b32") }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3);
}

template <int W>
double run(const char* name, uint32_t* d, int cus, double base, int blocks_per_cu = 8)
{
    const int iters = 1000, per_iter = 64;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    k<W><<<cus * blocks_per_cu, 256>>>(d, 10, 0x7FE001u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<W><<<cus * blocks_per_cu, 256>>>(d, iters, 0x7FE001u);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    double wi = (double)blocks_per_cu * iters * per_iter;
    double rate = wi / (ms * 1e6);
    printf("%-34s wps=%d %8.3f ms  %7.4f wi/ns/SIMD  rel %.3f  cyc@2.4GHz %.2f\n", name, blocks_per_cu, ms, rate,
           base > 0 ? rate / base : 1.0, 2.4 / rate);
    return rate;
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    uint32_t* d;
    (void)hipMalloc(&d, cus * 8 * 256 * 4);
    double base = run<0>("v_add_u32", d, cus, 0);
    run<0>("v_add_u32", d, cus, base, 1);
    run<0>("v_add_u32", d, cus, base, 2);
    run<1>("v_sub_u32", d, cus, base);
    run<2>("v_mul_lo_u32", d, cus, base);
    run<2>("v_mul_lo_u32", d, cus, base, 1);
    run<3>("v_mul_hi_u32", d, cus, base);
    run<4>("v_mul_hi_i32", d, cus, base);
    run<5>("v_add3_u32", d, cus, base);
    run<6>("v_ashrrev_i32", d, cus, base);
    run<7>("v_and_b32", d, cus, base);
    run<8>("v_xor_b32", d, cus, base);
    run<9>("v_max_i32", d, cus, base);
    run<10>("v_cndmask_b32_e32 vcc", d, cus, base);
    run<11>("v_cndmask_b32_e64 sgpr", d, cus, base);
    run<12>("v_mad_u64_u32", d, cus, base);
    run<13>("v_bfi_b32", d, cus, base);
    run<14>("ds_swizzle_b32 (8 + wait)", d, cus, base);
    run<15>("ds_bpermute_b32 (8 + wait)", d, cus, base);
    run<16>("v_add_u32_dpp quad_perm", d, cus, base);
    run<17>("v_cndmask_b32_dpp quad_perm", d, cus, base);
    run<18>("v_mad_u32_u24", d, cus, base);
    run<19>("v_mul_u32_u24", d, cus, base);
    run<19>("v_mul_u32_u24", d, cus, base, 1);
    run<20>("v_perm_b32", d, cus, base);
    run<21>("v_min_u32", d, cus, base);
    run<22>("v_lshlrev_b32", d, cus, base);
    run<23>("v_lshl_or_b32", d, cus, base);
    return 0;
}
