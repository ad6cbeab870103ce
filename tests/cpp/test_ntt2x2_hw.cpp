// Our main for the reference's hardware-model test hardware_code/ntt2x2_test.cpp:141-197,
// written against the reference's API names and linked against the GPU drop-in: the four checks
// ntt2x2_MUL (:87-107), ntt2x2_NTT (:41-58), ntt2x2_INVNTT (:64-81) and polymul (:109-137,
// b = 31 a as in :171-172), each under the same row mapping the reference expects, with the gold
// side computed by the CPU oracle instead of the reference's CPU functions.
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "../../include/dil256_ref.hpp"

extern "C" {
void orc_ntt(int32_t*);
void orc_invntt(int32_t*);
void orc_ntt2x2(int32_t*);
void orc_invntt2x2(int32_t*);
void orc_pointwise(int32_t*, const int32_t*, const int32_t*);
}

#ifndef TESTS
#define TESTS 5000
#endif

// compare_bram_array of util.cpp:85-140: canonicalise both sides, walk rows under `mapping`
static int compare_bram_array(const bram* ram, const data_t gold[DILITHIUM_N], const char* what, enum MAPPING mapping)
{
    for (int i = 0; i < DILITHIUM_N; i += 4) {
        const unsigned addr = resolve_address(mapping, i / 4);
        for (int j = 0; j < 4; j++) {
            const data_t g = ((gold[i + j] % DILITHIUM_Q) + DILITHIUM_Q) % DILITHIUM_Q;
            const data_t t = ((ram->coeffs[addr][j] % DILITHIUM_Q) + DILITHIUM_Q) % DILITHIUM_Q;
            if (g != t) {
                printf("%s Error at index: %d => %u (gold %d test %d)\n", what, i + j, addr, g, t);
                return 1;
            }
        }
    }
    return 0;
}

static int t_ntt(data_t r[DILITHIUM_N])
{
    bram ram;
    reshape(&ram, r);
    ntt2x2_fwdntt(&ram, FORWARD_NTT_MODE, NATURAL);
    orc_ntt2x2(r);
    return compare_bram_array(&ram, r, "ntt2x2_NTT", AFTER_NTT);
}
static int t_invntt(data_t r[DILITHIUM_N])
{
    bram ram;
    reshape(&ram, r);
    ntt2x2_invntt(&ram, INVERSE_NTT_MODE, NATURAL);
    orc_invntt2x2(r);
    return compare_bram_array(&ram, r, "ntt2x2_INVNTT", AFTER_INVNTT);
}
static int t_mul(data_t r[DILITHIUM_N], data_t m[DILITHIUM_N])
{
    bram ram, mul_ram;
    reshape(&ram, r);
    reshape(&mul_ram, m);
    ntt2x2_mul(&ram, &mul_ram, NATURAL);
    orc_pointwise(r, r, m);
    return compare_bram_array(&ram, r, "ntt2x2_MUL", NATURAL);
}
static int t_polymul(data_t a[DILITHIUM_N], data_t b[DILITHIUM_N])
{
    bram ra, rb;
    int ret = 0;
    reshape(&ra, a);
    reshape(&rb, b);
    ntt2x2_fwdntt(&ra, FORWARD_NTT_MODE, NATURAL);
    ntt2x2_fwdntt(&rb, FORWARD_NTT_MODE, NATURAL);
    orc_ntt(a);
    orc_ntt(b);
    ret |= compare_bram_array(&ra, a, "FORWARD_NTT_MODE A", AFTER_NTT);
    ret |= compare_bram_array(&rb, b, "FORWARD_NTT_MODE B", AFTER_NTT);
    ntt2x2_mul(&ra, &rb, NATURAL);
    orc_pointwise(a, a, b);
    ret |= compare_bram_array(&ra, a, "MUL A*B", AFTER_NTT);
    ntt2x2_invntt(&ra, INVERSE_NTT_MODE, AFTER_NTT);
    orc_invntt(a);
    ret |= compare_bram_array(&ra, a, "INVERSE_NTT_MODE(A*B)", NATURAL);
    return ret;
}

int main()
{
    printf("Test for DILITHIUM_N = %u\n", DILITHIUM_N);
    srand(12345);
    data_t r_invntt[DILITHIUM_N], r_mul[DILITHIUM_N], test_ram[DILITHIUM_N], r_ntt[DILITHIUM_N], a[DILITHIUM_N], b[DILITHIUM_N];
    int ret = 0;
    for (int k = 0; k < TESTS && !ret; k++) {
        for (int i = 0; i < DILITHIUM_N; i++) {
            r_invntt[i] = rand() % DILITHIUM_Q;
            r_mul[i] = rand() % DILITHIUM_Q;
            test_ram[i] = rand() % DILITHIUM_Q;
            r_ntt[i] = rand() % DILITHIUM_Q;
            const data_t t5 = rand() % DILITHIUM_Q;
            a[i] = t5;
            b[i] = (data_t)(((data2_t)t5 * 31) % DILITHIUM_Q);
        }
        ret |= t_mul(r_mul, test_ram);
        ret |= t_ntt(r_ntt);
        ret |= t_invntt(r_invntt);
        ret |= t_polymul(a, b);
    }
    // the table the drop-in exports must be the reference's zetas_barrett (consts.cpp:64-97)
    if (zetas_barrett[0] != 0 || zetas_barrett[1] != -3572223 || zetas_barrett[2] != 3765607 || zetas_barrett[255] != -731434) ret |= 2;
    printf(ret ? "ERROR\n" : "OK\n");
    return ret;
}
