// The reference's two differential tests at THEIR iteration counts -- reference_code/ref_test_ntt_ntt2x2.cpp:29
// (100 000 forward + 100 000 inverse) and hardware_code/ntt2x2_test.cpp:139 (1 000 000 x {ntt2x2_MUL, ntt2x2_NTT,
// ntt2x2_INVNTT, polymul}) -- with the device side called through the BATCHED host-pointer C-ABI (include/dil256.h)
// in chunks, because one GPU round trip per polynomial would take hours.  Same input law (rand() % Q, b = 31 a),
// same row mappings, same canonical comparison (util.cpp:98-112); the gold side is the CPU oracle
// (oracle/dil_oracle.c, pinned to the compiled reference by tests/test_oracle.py), one polynomial at a time.
//   usage: test_batched_differential [ref_iters [hw_iters]]     (defaults 100000 1000000)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/dil256.h"

extern "C" {
void orc_ntt(int32_t*);
void orc_invntt(int32_t*);
void orc_ntt2x2(int32_t*);
void orc_invntt2x2(int32_t*);
void orc_pointwise(int32_t*, const int32_t*, const int32_t*);
unsigned orc_resolve_address(int mapping, unsigned addr);
}

namespace {
constexpr int N = DIL_N, Q = DIL_Q;
constexpr size_t CHUNK = 16384;

int fail(const char* what, size_t item, int idx, int32_t gold, int32_t got)
{
    printf("%s: item %zu index %d: gold %d test %d\nERROR\n", what, item, idx, gold, got);
    return 1;
}
int32_t canon(int32_t v) { return (int32_t)((((int64_t)v % Q) + Q) % Q); }
// `ram` (device result, bram rows behind `mapping`) against `gold` (reference order), canonically
int compare_bram(const int32_t* ram, const int32_t* gold, int mapping, const char* what, size_t item)
{
    for (int r = 0; r < N / 4; r++) {
        const unsigned addr = orc_resolve_address(mapping, (unsigned)r);
        for (int j = 0; j < 4; j++)
            if (canon(ram[4 * addr + j]) != canon(gold[4 * r + j])) return fail(what, item, 4 * r + j, canon(gold[4 * r + j]), canon(ram[4 * addr + j]));
    }
    return 0;
}
#define CK(call)                                                                  \
    do {                                                                          \
        const int rc__ = (call);                                                  \
        if (rc__) {                                                               \
            printf("%s failed: %d (%s)\nERROR\n", #call, rc__, dil_error_string(rc__)); \
            return 1;                                                             \
        }                                                                         \
    } while (0)
}  // namespace

int main(int argc, char** argv)
{
    const size_t ref_iters = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000;
    const size_t hw_iters = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1000000;
    CK(dil_init(0));
    std::vector<int32_t> a(CHUNK * N), g(CHUNK * N);

    // ---- ref_test_ntt_ntt2x2.cpp:51-90 -----------------------------------------------------------
    srand(0);
    for (int inverse = 0; inverse < 2; inverse++) {
        printf("Test %s NTT = %zu :", inverse ? "Inverse" : "Forward", ref_iters);
        for (size_t done = 0; done < ref_iters; done += CHUNK) {
            const size_t n = ref_iters - done < CHUNK ? ref_iters - done : CHUNK;
            for (size_t i = 0; i < n * N; i++) a[i] = g[i] = rand() % Q;
            CK(inverse ? dil_invntt_host(a.data(), n) : dil_ntt_host(a.data(), n));
            for (size_t p = 0; p < n; p++) {
                inverse ? orc_invntt(&g[p * N]) : orc_ntt(&g[p * N]);
                for (int i = 0; i < N; i++)
                    if ((g[p * N + i] - a[p * N + i]) % Q != 0) return fail("ntt vs gold", done + p, i, g[p * N + i], a[p * N + i]);
            }
        }
        printf("OK\n");
    }

    // ---- ntt2x2_test.cpp:141-197 -----------------------------------------------------------------
    printf("Test for DILITHIUM_N = %u, %zu iterations :", N, hw_iters);
    srand(12345);
    std::vector<int32_t> r_inv(CHUNK * N), r_mul(CHUNK * N), t_ram(CHUNK * N), r_ntt(CHUNK * N), pa(CHUNK * N), pb(CHUNK * N);
    std::vector<int32_t> d0(CHUNK * N), d1(CHUNK * N);
    for (size_t done = 0; done < hw_iters; done += CHUNK) {
        const size_t n = hw_iters - done < CHUNK ? hw_iters - done : CHUNK;
        for (size_t k = 0; k < n; k++)
            for (int i = 0; i < N; i++) {
                const size_t o = k * N + i;
                r_inv[o] = rand() % Q;
                r_mul[o] = rand() % Q;
                t_ram[o] = rand() % Q;
                r_ntt[o] = rand() % Q;
                const int32_t t5 = rand() % Q;
                pa[o] = t5;
                pb[o] = (int32_t)(((int64_t)t5 * 31) % Q);
            }
        const size_t bytes = n * N * sizeof(int32_t);
        // ntt2x2_MUL (:87-107): ram = r_mul (reshape keeps the memory image), mul_ram = test_ram, NATURAL
        memcpy(d0.data(), r_mul.data(), bytes);
        CK(dil_bram_mul_host(d0.data(), t_ram.data(), n, DIL_MAP_NATURAL));
        for (size_t k = 0; k < n; k++) {
            orc_pointwise(&r_mul[k * N], &r_mul[k * N], &t_ram[k * N]);
            if (compare_bram(&d0[k * N], &r_mul[k * N], DIL_MAP_NATURAL, "ntt2x2_MUL", done + k)) return 1;
        }
        // ntt2x2_NTT (:41-58)
        memcpy(d0.data(), r_ntt.data(), bytes);
        CK(dil_bram_fwdntt_host(d0.data(), n, DIL_MAP_NATURAL));
        for (size_t k = 0; k < n; k++) {
            orc_ntt2x2(&r_ntt[k * N]);
            if (compare_bram(&d0[k * N], &r_ntt[k * N], DIL_MAP_AFTER_NTT, "ntt2x2_NTT", done + k)) return 1;
        }
        // ntt2x2_INVNTT (:64-81)
        memcpy(d0.data(), r_inv.data(), bytes);
        CK(dil_bram_invntt_host(d0.data(), n, DIL_MAP_NATURAL));
        for (size_t k = 0; k < n; k++) {
            orc_invntt2x2(&r_inv[k * N]);
            if (compare_bram(&d0[k * N], &r_inv[k * N], DIL_MAP_AFTER_INVNTT, "ntt2x2_INVNTT", done + k)) return 1;
        }
        // polymul (:109-137)
        memcpy(d0.data(), pa.data(), bytes);
        memcpy(d1.data(), pb.data(), bytes);
        CK(dil_bram_fwdntt_host(d0.data(), n, DIL_MAP_NATURAL));
        CK(dil_bram_fwdntt_host(d1.data(), n, DIL_MAP_NATURAL));
        for (size_t k = 0; k < n; k++) {
            orc_ntt(&pa[k * N]);
            orc_ntt(&pb[k * N]);
            if (compare_bram(&d0[k * N], &pa[k * N], DIL_MAP_AFTER_NTT, "FORWARD_NTT_MODE A", done + k)) return 1;
            if (compare_bram(&d1[k * N], &pb[k * N], DIL_MAP_AFTER_NTT, "FORWARD_NTT_MODE B", done + k)) return 1;
        }
        CK(dil_bram_mul_host(d0.data(), d1.data(), n, DIL_MAP_NATURAL));
        for (size_t k = 0; k < n; k++) {
            orc_pointwise(&pa[k * N], &pa[k * N], &pb[k * N]);
            if (compare_bram(&d0[k * N], &pa[k * N], DIL_MAP_AFTER_NTT, "MUL A*B", done + k)) return 1;
        }
        CK(dil_bram_invntt_host(d0.data(), n, DIL_MAP_AFTER_NTT));
        for (size_t k = 0; k < n; k++) {
            orc_invntt(&pa[k * N]);
            if (compare_bram(&d0[k * N], &pa[k * N], DIL_MAP_NATURAL, "INVERSE_NTT_MODE(A*B)", done + k)) return 1;
        }
    }
    printf("OK\n");
    return 0;
}
