// A C++ host program driving EVERY visible GPU from one process through the device-resident multi-GPU layer of the C-ABI
// (include/dil256.h: dil_*_multi_dev): each device computes its slab of the batch in place inside a full-size result array of its
// own, one RCCL collective over xGMI completes the arrays (north_star "RCCL over xGMI only for the final gather").
// Checked here without any oracle: the gathered forward NTT of a ragged batch equals the single-device transform, the gathered
// inverse brings the input back, a signing batch under one key (sharded, slabs gathered to every device and, separately, to a root)
// equals the single-device signing call byte for byte and verifies on the multi-device path.
//   usage: test_multi_dev [ndev (0 = all)] [batch]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/dil256.h"

#define CK(call)                                                                                                         \
    do {                                                                                                                 \
        const int rc__ = (int)(call);                                                                                    \
        if (rc__) {                                                                                                      \
            printf("%s failed: %d (%s) %s\nERROR\n", #call, rc__, dil_error_string(rc__), rc__ == DIL_ERR_RCCL ? dil_multi_last_error() : ""); \
            return 1;                                                                                                    \
        }                                                                                                                \
    } while (0)

int main(int argc, char** argv)
{
    int have = 0;
    CK(dil_device_count(&have));
    int G = argc > 1 ? atoi(argv[1]) : 0;
    if (G <= 0 || G > have) G = have;
    const size_t n = argc > 2 ? strtoull(argv[2], nullptr, 10) : 4099;       // ragged for every G > 1
    const int level = 3;
    const size_t pkb = dil_pk_bytes(level), skb = dil_sk_bytes(level), sgb = dil_sig_bytes(level);
    CK(dil_multi_init(G));
    srand(11);
    // ---- NTT: slab in place inside a full-size array per device -----------------------------------------------------------
    std::vector<int32_t> a(n * 256), ref(n * 256), got(n * 256);
    for (auto& v : a) v = (int32_t)((unsigned)rand() % DIL_Q);
    ref = a;
    CK(hipSetDevice(0));
    CK(dil_ntt_host(ref.data(), n));                                         // single-device reference
    std::vector<int32_t*> d_polys(G);
    for (int g = 0; g < G; g++) {
        size_t lo, hi;
        dil_shard_range(n, g, G, &lo, &hi);
        CK(hipSetDevice(g));
        CK(hipMalloc(&d_polys[g], n * 1024));
        CK(hipMemset(d_polys[g], 0xEE, n * 1024));                            // poison: only the gather may fill the other slabs
        CK(hipMemcpy(d_polys[g] + lo * 256, a.data() + lo * 256, (hi - lo) * 1024, hipMemcpyHostToDevice));
    }
    CK(dil_ntt_multi_dev(d_polys.data(), n, /*inverse*/ 0, /*gather_root: all*/ -1, G));
    for (int g = 0; g < G; g++) {
        CK(hipSetDevice(g));
        CK(hipMemcpy(got.data(), d_polys[g], n * 1024, hipMemcpyDeviceToHost));
        if (got != ref) return printf("device %d: gathered forward NTT differs from the single-device transform\nERROR\n", g), 1;
    }
    CK(dil_ntt_multi_dev(d_polys.data(), n, /*inverse*/ 1, /*gather_root*/ 0, G));   // every device inverts its slab, root 0 collects
    CK(hipSetDevice(0));
    CK(hipMemcpy(got.data(), d_polys[0], n * 1024, hipMemcpyDeviceToHost));
    if (got != a) return printf("gathered inverse NTT on the root is not the input\nERROR\n"), 1;
    // ---- signing under one key, sharded; signatures gathered to every device ------------------------------------------------
    const size_t m = n < 2000 ? n : 1501;
    std::vector<uint8_t> seed(32), pk(pkb), sk(skb), mu(m * 64), sig_ref(m * sgb), sig(m * sgb);
    std::vector<int32_t> att_ref(m), att(m), verdict(m);
    for (auto& b : seed) b = (uint8_t)rand();
    for (auto& b : mu) b = (uint8_t)rand();
    CK(hipSetDevice(0));
    CK(dil_keygen_host(pk.data(), sk.data(), seed.data(), level, 1));
    CK(dil_sign_host(sig_ref.data(), att_ref.data(), sk.data(), mu.data(), level, m, /*shared_sk*/ 1, 512));
    std::vector<uint8_t*> d_sig(G), d_sk(G), d_mu(G), d_pk(G), d_sig_slice(G);
    std::vector<int32_t*> d_att(G), d_verdict(G);
    for (int g = 0; g < G; g++) {
        size_t lo, hi;
        dil_shard_range(m, g, G, &lo, &hi);
        CK(hipSetDevice(g));
        CK(hipMalloc(&d_sig[g], m * sgb));
        CK(hipMalloc(&d_att[g], m * 4));
        CK(hipMalloc(&d_verdict[g], m * 4));
        CK(hipMalloc(&d_sk[g], skb));
        CK(hipMalloc(&d_pk[g], pkb));
        CK(hipMalloc(&d_mu[g], (hi - lo + 1) * 64));
        CK(hipMalloc(&d_sig_slice[g], (hi - lo + 1) * sgb));
        CK(hipMemset(d_sig[g], 0, m * sgb));
        CK(hipMemset(d_verdict[g], 0x7F, m * 4));
        CK(hipMemcpy(d_sk[g], sk.data(), skb, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_pk[g], pk.data(), pkb, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_mu[g], mu.data() + lo * 64, (hi - lo) * 64, hipMemcpyHostToDevice));
    }
    CK(dil_sign_multi_dev(d_sig.data(), d_att.data(), d_sk.data(), d_mu.data(), level, m, 1, 512, -1, G));
    for (int g = 0; g < G; g++) {
        CK(hipSetDevice(g));
        CK(hipMemcpy(sig.data(), d_sig[g], m * sgb, hipMemcpyDeviceToHost));
        CK(hipMemcpy(att.data(), d_att[g], m * 4, hipMemcpyDeviceToHost));
        if (sig != sig_ref || att != att_ref) return printf("device %d: gathered signatures differ from the single-device call\nERROR\n", g), 1;
        size_t lo, hi;
        dil_shard_range(m, g, G, &lo, &hi);
        CK(hipMemcpy(d_sig_slice[g], d_sig[g] + lo * sgb, (hi - lo) * sgb, hipMemcpyDeviceToDevice));   // this device's slice as verify input
    }
    CK(dil_verify_sig_multi_dev(d_verdict.data(), d_pk.data(), d_sig_slice.data(), d_mu.data(), level, m, 1, /*root*/ 0, G));
    CK(hipSetDevice(0));
    CK(hipMemcpy(verdict.data(), d_verdict[0], m * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < m; i++)
        if (verdict[i] != 0) return printf("multi-device verify rejected signature %zu (verdict %d)\nERROR\n", i, verdict[i]), 1;
    CK(dil_multi_shutdown());
    CK(dil_shutdown());
    printf("%d device(s): NTT all-gather + gather-to-root, %zu signatures sharded / gathered / verified: identical to the single-device calls\nOK\n", G, m);
    return 0;
}
