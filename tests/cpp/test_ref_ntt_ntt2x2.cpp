// Our main for the reference's differential test reference_code/ref_test_ntt_ntt2x2.cpp:44-93,
// written against the reference's API names (include/dil256_ref.hpp) and linked against the GPU
// drop-in.  Same seed (srand(0)), same input law (rand() % Q), same congruence criterion
// ((gold - x) % Q == 0, :31-42).  Where the reference compares its two CPU implementations with
// each other, this compares the drop-in's ntt2x2_ref / invntt2x2_ref with its ntt / invntt AND
// with the CPU oracle (oracle/dil_oracle.c), so a shared GPU bug cannot hide.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/dil256_ref.hpp"

extern "C" {
void orc_ntt(int32_t*);
void orc_invntt(int32_t*);
}

#ifndef TESTS
#define TESTS 20000
#endif

static int compare_array(const data_t* a_gold, const data_t* a)
{
    for (int i = 0; i < DILITHIUM_N; i++)
        if ((a_gold[i] - a[i]) % DILITHIUM_Q != 0) {
            printf("%d: %d != %d\n", i, a_gold[i], a[i]);
            return 1;
        }
    return 0;
}

int main()
{
    data_t a[DILITHIUM_N], g[DILITHIUM_N], o[DILITHIUM_N];
    srand(0);
    printf("Test Forward NTT = %u :", TESTS);
    for (int j = 0; j < TESTS; j++) {
        for (int i = 0; i < DILITHIUM_N; i++) a[i] = g[i] = o[i] = rand() % DILITHIUM_Q;
        ntt2x2_ref(a);
        ntt(g);
        orc_ntt(o);
        if (compare_array(g, a) || compare_array(o, a)) return 1;
    }
    printf("OK\n");
    printf("Test Inverse NTT = %u :", TESTS);
    for (int j = 0; j < TESTS; j++) {
        for (int i = 0; i < DILITHIUM_N; i++) a[i] = g[i] = o[i] = rand() % DILITHIUM_Q;
        invntt2x2_ref(a);
        invntt(g);
        orc_invntt(o);
        if (compare_array(g, a) || compare_array(o, a)) return 1;
    }
    printf("OK\n");
    return 0;
}
